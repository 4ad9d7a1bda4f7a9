// FlashAttention forward for gfx950 (MI355X / CDNA4) -- device code.
//
// Replaces (behaviourally, not textually) the reference's fused forward kernels:
//   _fwd_kernel                  code/triton_fa2/FA2-triton.py:25-93
//   flash_attn_cutlass_kernel<D> code/cutlass_cuda_fa1/run/flash_attn_cutlass.cu:346-453
//
// Structure (one workgroup = 8 wave64 = one 256-row query block of one (batch, head)):
//   * each wave owns 32 query rows; the workgroup walks 64-key K/V tiles.
//   * S^T = K . Q^T  with v_mfma_f32_32x32x16 (K fragment = A operand from LDS via
//     ds_read_b128, Q fragment = B operand, resident in VGPRs).  With this orientation the
//     accumulator holds, per lane, 16 keys x ONE query (query = lane & 31), so the online
//     softmax (running max / sum) is lane-local plus one exchange with lane ^ 32.
//   * O^T += V^T . P^T : the S^T accumulator, converted to 16-bit pairs, IS the B operand
//     (no LDS round trip for P); V^T fragments come from the row-major V tile in LDS through
//     ds_read_b64_tr_b16 (hardware transpose read).
//   * K/V tiles: global -> VGPR (buffer_load_dwordx4, coalesced along head_dim, OOB rows
//     read as zero) -> LDS (ds_write_b128, XOR-swizzled so both the row reads of K and the
//     transposed reads of V are bank-conflict free), double buffered, one barrier per tile;
//     the loads for tile j+2 are in flight while tile j is computed.
//   * exp2 with scale*log2(e) folded in, deferred 1/l normalisation, lazy O rescale
//     (only when some row's max grew by more than RESCALE_THR in log2 units),
//     causal tile skipping + diagonal-only masking, LSE output.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fa {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;

struct FwdParams {
    const void* q; const void* k; const void* v; void* o; float* lse;
    int B, H, S;
    int nqb;                 // ceil(S / 256)
    int bh;                  // B*H
    // element strides (innermost head_dim stride is 1)
    long long q_sb, q_sh, q_ss;
    long long k_sb, k_sh, k_ss;
    long long v_sb, v_sh, v_ss;
    long long o_sb, o_sh, o_ss;
    float scale;             // softmax scale (1/sqrt(D) by default: FA2-triton.py:183)
    float scale_log2;        // scale * log2(e)
};

template <class To, class From> __device__ __forceinline__ To bitcast(From f) { return __builtin_bit_cast(To, f); }

struct TypeBF16 {
    static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(bitcast<bf16x8>(a), bitcast<bf16x8>(b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ unsigned pack2(float a, float b) {
        f32x2 f = {a, b};
        return bitcast<unsigned>(__builtin_convertvector(f, bf16x2));
    }
};
struct TypeF16 {
    static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(bitcast<f16x8>(a), bitcast<f16x8>(b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ unsigned pack2(float a, float b) {
        f32x2 f = {a, b};
        return bitcast<unsigned>(__builtin_convertvector(f, f16x2));
    }
};

constexpr int kBM = 256;          // query rows per workgroup
constexpr int kBN = 64;           // keys per tile
constexpr int kThreads = 512;
constexpr float kRescaleThr = 8.0f;   // log2 units; 0 = rescale whenever any row max grows

template <int D> constexpr int lds_bytes() { return 2 /*buffers*/ * 2 /*K,V*/ * kBN * D * 2; }

typedef __attribute__((address_space(3))) char lds_char;

__device__ __forceinline__ u32x4 lds_read_b128(unsigned addr) {
    return *reinterpret_cast<__attribute__((address_space(3))) u32x4*>(addr);
}
__device__ __forceinline__ void lds_write_b128(unsigned addr, u32x4 v) {
    *reinterpret_cast<__attribute__((address_space(3))) u32x4*>(addr) = v;
}
__device__ __forceinline__ u32x2 lds_read_tr16_b64(unsigned addr) {
    s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<__attribute__((address_space(3))) s16x4*>(addr));
    return bitcast<u32x2>(t);
}

// 16-byte-chunk swizzles of the LDS tiles (row = key index inside the tile).
//  K is read by rows with ds_read_b128: the 16 lanes of a read group hold 16 different keys at the
//  same head_dim chunk -> spread them over the 16 slots of the 256-byte bank row.
//  V is read transposed (4 consecutive keys x 64 bytes per 32-lane half) -> put 4 consecutive
//  keys into the 4 different 64-byte quarters of the bank row.
template <int D> __device__ __forceinline__ int k_swz(int row, int ch) {
    if constexpr (D == 128) return ch ^ (row & 15);
    else return ch ^ ((row >> 1) & 7);
}
template <int D> __device__ __forceinline__ int v_swz(int row, int ch) {
    if constexpr (D == 128) return ch ^ ((row & 3) << 2);
    else return ch ^ (((row >> 1) & 1) << 2);
}

template <class T, int D, bool CAUSAL>
__global__ __launch_bounds__(kThreads, 2) void fa_fwd_kernel(const FwdParams p)
{
    constexpr int KS = D / 16;                 // k-steps of the QK^T product
    constexpr int DB = D / 32;                 // 32-wide head_dim blocks of O^T
    constexpr int ROWB = D * 2;                // bytes per K/V row in LDS
    constexpr int CH = D / 8;                  // 16-byte chunks per row
    constexpr int TILE = kBN * ROWB;           // bytes per K (or V) tile
    constexpr int CPT = (kBN * CH) / kThreads; // chunks per thread per tile (2 @ D=128, 1 @ D=64)
    static_assert(CPT >= 1, "tile too small");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_char*)smem;   // [buf][K|V][TILE]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31;
    const int hh = lane >> 5;

    // ---- workgroup -> (head, query block).  blockIdx % 8 labels the XCD group: all query blocks
    // of one head share an XCD (its L2 holds that head's K/V); heavy blocks first when causal.
    const int bid = blockIdx.x;
    const int xcd = bid & 7;
    const int slot = bid >> 3;
    const int hl = slot / p.nqb;
    const int qpos = slot - hl * p.nqb;
    const int head = hl * 8 + xcd;
    if (head >= p.bh) return;
    const int qb = CAUSAL ? (p.nqb - 1 - qpos) : qpos;
    const int b = head / p.H;
    const int h = head - b * p.H;
    const int S = p.S;

    // waves w and w+4 share a SIMD: give them row blocks that sum to 7 so that the causal
    // diagonal block is balanced across SIMDs.
    const int rowblk = CAUSAL ? (wave < 4 ? wave : 11 - wave) : wave;
    const int q0w = qb * kBM + rowblk * 32;

    const int kv_end_wg = CAUSAL ? min(S, qb * kBM + kBM) : S;
    const int nt = (kv_end_wg + kBN - 1) / kBN;
    const int kv_end_w = (q0w >= S) ? 0 : (CAUSAL ? min(S, q0w + 32) : S);
    const int my_nt = (kv_end_w + kBN - 1) / kBN;

    using elem_t = unsigned short;
    const elem_t* qh = reinterpret_cast<const elem_t*>(p.q) + b * p.q_sb + h * p.q_sh;
    const elem_t* kh = reinterpret_cast<const elem_t*>(p.k) + b * p.k_sb + h * p.k_sh;
    const elem_t* vh = reinterpret_cast<const elem_t*>(p.v) + b * p.v_sb + h * p.v_sh;
    elem_t* oh = reinterpret_cast<elem_t*>(p.o) + b * p.o_sb + h * p.o_sh;

    const unsigned q_bytes = (unsigned)(((long long)(S - 1) * p.q_ss + D) * 2);
    const unsigned k_bytes = (unsigned)(((long long)(S - 1) * p.k_ss + D) * 2);
    const unsigned v_bytes = (unsigned)(((long long)(S - 1) * p.v_ss + D) * 2);
    __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<elem_t*>(qh), 0, q_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<elem_t*>(kh), 0, k_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<elem_t*>(vh), 0, v_bytes, 0x00020000);

    // ---- Q fragments (B operand of S^T = K Q^T): lane (r,hh) holds Q[q0w + r][16 ks + 8 hh + 0..7]
    u32x4 qf[KS];
    {
        const int qrow = q0w + r;
        // rows past the end of the sequence get an offset outside the descriptor: they read as zero
        const unsigned qoff = (qrow < S) ? (unsigned)((long long)qrow * p.q_ss * 2 + hh * 16) : 0x80000000u;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            qf[ks] = __builtin_amdgcn_raw_buffer_load_b128(rq, qoff + ks * 32, 0, 0);
    }

    // ---- K/V staging: thread -> (row, chunk) of the tile.  LDS: [K0][K1][V0][V1], TILE bytes each.
    unsigned g_koff[CPT], g_voff[CPT], l_koff[CPT], l_voff[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int c = tid + kThreads * i;
        const int row = c / CH, ch = c % CH;
        g_koff[i] = (unsigned)(row * p.k_ss * 2 + ch * 16);
        g_voff[i] = (unsigned)(row * p.v_ss * 2 + ch * 16);
        l_koff[i] = row * ROWB + k_swz<D>(row, ch) * 16;
        l_voff[i] = 2 * TILE + row * ROWB + v_swz<D>(row, ch) * 16;
    }
    const unsigned k_tile_stride = (unsigned)(kBN * p.k_ss * 2);
    const unsigned v_tile_stride = (unsigned)(kBN * p.v_ss * 2);

    // rows past the end of the sequence fall outside the descriptor and read as zero
    // (the host guarantees (S + 256) * row_stride_bytes < 2^31: no 32-bit wrap)
    u32x4 kreg[CPT], vreg[CPT];
    auto load_k = [&](int j) {
#pragma unroll
        for (int i = 0; i < CPT; ++i)
            kreg[i] = __builtin_amdgcn_raw_buffer_load_b128(rk, (unsigned)j * k_tile_stride + g_koff[i], 0, 0);
    };
    auto load_v = [&](int j) {
#pragma unroll
        for (int i = 0; i < CPT; ++i)
            vreg[i] = __builtin_amdgcn_raw_buffer_load_b128(rv, (unsigned)j * v_tile_stride + g_voff[i], 0, 0);
    };
    auto store_k = [&](int buf) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) lds_write_b128(lds_base + buf * TILE + l_koff[i], kreg[i]);
    };
    auto store_v = [&](int buf) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) lds_write_b128(lds_base + buf * TILE + l_voff[i], vreg[i]);
    };

    // ---- LDS read addresses
    // K fragment (A operand): lane (r,hh) reads K[kb*32 + r][16 ks + 8 hh + 0..7] = chunk 2ks+hh of row
    unsigned ka[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) ka[ks] = lds_base + r * ROWB + k_swz<D>(r, 2 * ks + hh) * 16;   // + kb*32*ROWB
    // V^T fragment (A operand of O^T = V^T P^T): 4x16 transposed blocks.
    //   16-lane group g16 (0/1) covers head_dim columns 16*g16..+15 of the 32-wide block,
    //   lane i of the group supplies row (i>>2), columns 4*(i&3)..+3; lane half hh adds 4 keys.
    unsigned va[DB];
    {
        const int i16 = lane & 15, g16 = (lane >> 4) & 1;
        const int qq = i16 >> 2, pp = i16 & 3;
        const int row = 4 * hh + qq;                       // + 16 s + 8 a  (multiples of 8: swizzle-neutral)
#pragma unroll
        for (int db = 0; db < DB; ++db) {
            const int ch = 4 * db + 2 * g16 + (pp >> 1);
            va[db] = lds_base + 2 * TILE + row * ROWB + v_swz<D>(row, ch) * 16 + 8 * (pp & 1);
        }
    }

    // K fragments are read in groups of 4 k-steps, one group ahead of the MFMAs that use them.
    constexpr int GPB = KS / 4;           // groups per 32-key block
    constexpr int NG = 2 * GPB;           // groups per tile
    u32x4 kf[2][4];
    auto read_kgroup = [&](u32x4 (&dst)[4], unsigned koff, int g) {
        const int kb = g / GPB, ks0 = (g % GPB) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) dst[i] = lds_read_b128(ka[ks0 + i] + koff + kb * 32 * ROWB);
    };
    // V^T fragments of one 16-key k-step (DB blocks of 32 head_dim columns)
    u32x4 vf[2][DB];
    auto read_vstep = [&](u32x4 (&dst)[DB], unsigned voff, int s) {
#pragma unroll
        for (int db = 0; db < DB; ++db) {
            u32x2 lo = lds_read_tr16_b64(va[db] + voff + (16 * s) * ROWB);
            u32x2 hi = lds_read_tr16_b64(va[db] + voff + (16 * s + 8) * ROWB);
            dst[db] = u32x4{lo[0], lo[1], hi[0], hi[1]};
        }
    };

    f32x16 o_acc[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) o_acc[db][i] = 0.f;
    float m_c = -INFINITY;     // running max, already multiplied by scale*log2(e)
    float l_part = 0.f;        // this lane's share of the row sum (lanes r and r+32 are combined at the end)
    const float c = p.scale_log2;

    // ---- prologue: tile 0 -> LDS buffers 0, tile 1 in flight, first K fragments in registers
    load_k(0);
    load_v(0);
    store_k(0);
    store_v(0);
    load_k(1);
    load_v(1);
    __syncthreads();
    read_kgroup(kf[0], 0, 0);

    // One barrier per tile.  Buffer life times (b = j & 1):
    //   K[b]: written in iteration j-1 before barrier(j-1); read in QK(j) (first group already during PV(j-1),
    //         i.e. after barrier(j-1)); overwritten in iteration j+1 before barrier(j+1), after barrier(j).
    //   V[b]: written in iteration j-1 after barrier(j-1); read in PV(j) after barrier(j); overwritten in
    //         iteration j+1 after barrier(j+1).
    for (int j = 0; j < nt; ++j) {
        const unsigned koff = (j & 1) * TILE;
        const unsigned voff = (j & 1) * TILE;
        const unsigned koff_n = ((j + 1) & 1) * TILE;
        const bool active = j < my_nt;          // wave-uniform
        f32x16 s_acc[2];
        if (active) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) s_acc[kb][i] = 0.f;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + 1 < NG) read_kgroup(kf[(g + 1) & 1], koff, g + 1);
                __builtin_amdgcn_sched_barrier(0);
                const int kb = g / GPB, ks0 = (g % GPB) * 4;
#pragma unroll
                for (int i = 0; i < 4; ++i) s_acc[kb] = T::mfma(kf[g & 1][i], qf[ks0 + i], s_acc[kb]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // K tile j+1 -> LDS (its buffer was last read in QK(j-1), before barrier(j-1)); K tile j+2 in flight
        store_k((j + 1) & 1);
        load_k(j + 2);
        __syncthreads();                       // barrier(j)
        // V tile j+1 -> LDS (its buffer was last read in PV(j-1), before barrier(j)); V tile j+2 in flight
        store_v((j + 1) & 1);
        load_v(j + 2);

        if (active) {
            read_vstep(vf[0], voff, 0);
            const int k0 = j * kBN;
            const bool need_mask = (CAUSAL && (k0 + kBN - 1 > q0w)) || (k0 + kBN > S);
            if (need_mask) {
                const int qrow = q0w + r;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int key = k0 + kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
                        const bool dead = (key >= S) || (CAUSAL && key > qrow);
                        if (dead) s_acc[kb][i] = -INFINITY;
                    }
            }
            // row max over this lane's 32 keys, then with lane ^ 32
            float mx = s_acc[0][0];
#pragma unroll
            for (int i = 1; i < 16; ++i) mx = fmaxf(mx, s_acc[0][i]);
#pragma unroll
            for (int i = 0; i < 16; ++i) mx = fmaxf(mx, s_acc[1][i]);
            {
                auto sw = __builtin_amdgcn_permlane32_swap(bitcast<unsigned>(mx), bitcast<unsigned>(mx), false, false);
                mx = fmaxf(bitcast<float>(sw[0]), bitcast<float>(sw[1]));
            }
            const float mx_c = mx * c;
            // lazy rescale: only when some row's max grew by more than the threshold
            if (__builtin_amdgcn_ballot_w64(mx_c - m_c > kRescaleThr) != 0) {   // NaN (-inf - -inf) compares false
                const float m_new = fmaxf(m_c, mx_c);
                const float alpha = __builtin_amdgcn_exp2f(m_c - m_new);     // m_c = -inf -> 0
                l_part *= alpha;
#pragma unroll
                for (int db = 0; db < DB; ++db)
#pragma unroll
                    for (int i = 0; i < 16; ++i) o_acc[db][i] *= alpha;
                m_c = m_new;
            }
            const float m_sub = (m_c == -INFINITY) ? 0.f : m_c;   // fully masked so far: exp2(-inf - 0) = 0
            u32x4 pf[4];
            float rs = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                float pv[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    pv[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(s_acc[kb][i], c, -m_sub));
                    rs += pv[i];
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    u32x4 f;
#pragma unroll
                    for (int w = 0; w < 4; ++w) f[w] = T::pack2(pv[8 * s2 + 2 * w], pv[8 * s2 + 2 * w + 1]);
                    pf[kb * 2 + s2] = f;
                }
            }
            l_part += rs;
            // O^T += V^T P^T.  k-step s covers keys 16s..16s+15 of the tile in the permuted order
            // 16s + 8(jj>>2) + 4hh + (jj&3), which is exactly the order of the S^T accumulator registers.
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (s + 1 < 4) read_vstep(vf[(s + 1) & 1], voff, s + 1);
                else read_kgroup(kf[0], koff_n, 0);           // first K fragments of tile j+1
#pragma unroll
                for (int db = 0; db < DB; ++db) o_acc[db] = T::mfma(vf[s & 1][db], pf[s], o_acc[db]);
            }
        }
    }

    // ---- epilogue: combine the two lane halves' row sums, normalise, store O (and LSE)
    {
        auto sw = __builtin_amdgcn_permlane32_swap(bitcast<unsigned>(l_part), bitcast<unsigned>(l_part), false, false);
        const float l = bitcast<float>(sw[0]) + bitcast<float>(sw[1]);
        const float inv = (l > 0.f) ? 1.0f / l : 0.f;        // flash_attn_cutlass.cu:446-452 guard
        const int qrow = q0w + r;
        if (p.lse != nullptr && hh == 0 && qrow < S) {
            // lse = m*scale + ln(l) = (m_c + log2(l)) * ln(2)
            p.lse[((long long)b * p.H + h) * S + qrow] = (l > 0.f) ? (m_c + __builtin_amdgcn_logf(l)) * 0.6931471805599453f : -INFINITY;
        }
        // lane (r,hh) holds O[qrow][db*32 + 8g + 4hh + 0..3] in o_acc[db][4g..4g+3].
        // Pair column groups (g, g+1) with permlane32_swap so that each lane stores 16 contiguous bytes:
        // lanes 0-31 -> cols 8g..8g+7, lanes 32-63 -> cols 8(g+1)..8(g+1)+7.
        elem_t* orow = oh + (long long)qrow * p.o_ss;
#pragma unroll
        for (int db = 0; db < DB; ++db) {
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                unsigned a0 = T::pack2(o_acc[db][4 * g + 0] * inv, o_acc[db][4 * g + 1] * inv);
                unsigned a1 = T::pack2(o_acc[db][4 * g + 2] * inv, o_acc[db][4 * g + 3] * inv);
                unsigned b0 = T::pack2(o_acc[db][4 * g + 4] * inv, o_acc[db][4 * g + 5] * inv);
                unsigned b1 = T::pack2(o_acc[db][4 * g + 6] * inv, o_acc[db][4 * g + 7] * inv);
                auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                // after the swap: lanes 0-31: {own a, upper's a} ; lanes 32-63: {lower's b, own b}
                u32x4 outv = {s0[0], s1[0], s0[1], s1[1]};
                if (qrow < S) {
                    const int col = db * 32 + 8 * g + 8 * hh;
                    *reinterpret_cast<u32x4*>(orow + col) = outv;
                }
            }
        }
    }
}

}  // namespace fa
