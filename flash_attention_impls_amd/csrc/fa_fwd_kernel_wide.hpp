// FlashAttention forward for gfx950 -- wide heads: head_dim 144 .. 256 (multiples of 16), bf16 / fp16.
//
// SURVEY.md section 8(f) row N2 lists head_dim 256 among the "wider shapes"; the reference itself stops at 128
// (FA2-triton.py:178).  This kernel exists for coverage of that row, not for the headline: it is the plain form of the
// algorithm (FA2-triton.py:60-85 -- running maximum, rescale, deferred 1/l) on the fragment maps of fa_fwd_kernel16.hpp,
// without that kernel's software pipeline: at head_dim 256 the O^T accumulators alone are 128 registers per lane.
//
//   * workgroup = 128 query rows: 8 waves x 16 rows, two waves per SIMD within 256 registers (round 4; until then 4 waves x 32
//     rows, one wave per SIMD -- QT below); Q fragments (8 k-steps per query tile) stay in registers;
//   * 64-key tiles, K and V rows of 512 bytes staged by LDS-DMA into two stages each (2 x 2 x 32 KiB = 128 KiB), one
//     barrier per tile: tile j + 1 travels while tile j is computed.  Rows are 0 mod 256 bytes like head_dim 128's, so the
//     same 16-byte-chunk swizzles keep the row reads (K) and the transposed reads (V^T) conflict-free;
//   * per tile: S^T = K Q^T (64 MFMAs 16x16x32), the online softmax on the 64 scores of each query row (running maximum per
//     row, combined across the four lane groups by two shuffles; exponentials in base 2 with scale*log2(e) folded in), O^T
//     rescaled only when some row of the wave raised its maximum (a wave-uniform branch), O^T += V^T P^T (128 MFMAs);
//   * columns past the runtime head_dim are read as zeros through the buffer bounds (as in the other kernels) and not stored;
//     grouped key/value heads, S_q != S_k with the bottom-right aligned causal mask, strides and the LSE as everywhere.
#pragma once
#include "fa_fwd_kernel16.hpp"

namespace fa {

constexpr int kBMW = 128;                                  // query rows per workgroup
constexpr int kWideD = 256;                                // compiled head_dim
constexpr int kWideLds = 2 /*K, V*/ * 2 /*stages*/ * kBN * kWideD * 2;
#ifndef FA_WIDE_QT
#define FA_WIDE_QT 1
#endif
constexpr int kWideThreads = 512 / FA_WIDE_QT;            // 8 waves x 16 rows (default) or 4 waves x 32 rows

// QT = 16-row query tiles per wave: 2 (4 waves x 32 rows, one wave per SIMD, the whole register file) or 1 (8 waves x 16 rows,
// two waves per SIMD within 256 registers: each K / V^T fragment then feeds one MFMA instead of two -- twice the LDS reads, 2048
// cycles of the LDS array per tile against 3072 of the matrix pipe at head_dim 256 -- but a wave waiting for its fragments or at
// the tile's barrier has a partner that computes; the plain loop below has no other latency hiding).  Same arithmetic per row.
template <class T, bool CAUSAL, int QT = FA_WIDE_QT>
__global__ __launch_bounds__(512 / QT, QT == 1 ? 2 : 1) void fa_fwd_kernel_wide(const FwdParams p)
{
    static_assert(QT == 1 || QT == 2, "16 or 32 query rows per wave");
    constexpr int D = kWideD;
    constexpr int NWAVES = 8 / QT;
    constexpr int KS = D / 32;                 // k-steps of the QK^T product (8)
    constexpr int DT = D / 16;                 // 16-wide head_dim tiles of O^T (16)
    constexpr int ROWB = D * 2;                // bytes per K / V row in LDS
    constexpr int TILE = kBN * ROWB;           // bytes per K (or V) tile (32 KiB)
    constexpr int PIECE = 1024;
    constexpr int CPT = TILE / PIECE / NWAVES; // DMA pieces per wave per tile (8)
    constexpr int VBASE = 2 * TILE;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_char*)smem;   // K stages [2][TILE], then V stages

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int li = lane & 15, lg = lane >> 4;

    int head, tq;
    if (!wg_decode(blockIdx.x, p.bh, p.nqb, p.hsplit, head, tq)) return;
    const int b = head / p.H;
    const int h = head - b * p.H;
    const int S = p.S, Sk = p.Sk;
    const int coff = CAUSAL ? Sk - S : 0;
    const int qb = CAUSAL ? p.nqb - 1 - tq : tq;          // causal: the longest query blocks start first

    using elem_t = unsigned short;
    const char* qh = reinterpret_cast<const char*>(p.q) + (b * p.q_sb + h * p.q_sh) * 2;
    const int hk = h / p.G;
    const char* kh = reinterpret_cast<const char*>(p.k) + (b * p.k_sb + hk * p.k_sh) * 2;
    const elem_t* vh = reinterpret_cast<const elem_t*>(p.v) + b * p.v_sb + hk * p.v_sh;
    elem_t* oh = reinterpret_cast<elem_t*>(p.o) + b * p.o_sb + h * p.o_sh;

    const unsigned q_bytes = (unsigned)(((long long)(S - 1) * p.q_ss + p.dv) * 2);
    const unsigned k_bytes = (unsigned)(((long long)(Sk - 1) * p.k_ss + p.dv) * 2);
    const unsigned v_bytes = (unsigned)(((long long)(Sk - 1) * p.v_ss + p.dv) * 2);
    __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(qh), 0, q_bytes, 0x00020000);
    const u32x4 rk_w = make_rsrc(kh, k_bytes);
    const u32x4 rv_w = make_rsrc(vh, v_bytes);

    const int q0w = qb * kBMW + wave * 16 * QT;
    const int kv_end_wg = CAUSAL ? max(0, min(Sk, qb * kBMW + kBMW + coff)) : Sk;
    const int nt = (kv_end_wg + kBN - 1) / kBN;                       // tiles the workgroup stages
    const int kv_end_w = (q0w >= S) ? 0 : (CAUSAL ? max(0, min(Sk, q0w + 16 * QT + coff)) : Sk);
    const int my_nt = (kv_end_w + kBN - 1) / kBN;                     // tiles this wave computes on

    // ---- Q fragments: lane (li, lg) holds Q[q0w + 16 qt + li][32 ks + 8 lg + 0..7]; rows past S and columns past the
    // runtime head_dim get an offset outside the descriptor and read as zeros
    u32x4 qf[QT][KS];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int qrow = q0w + 16 * qt + li;
        const unsigned qoff = (qrow < S) ? (unsigned)((long long)qrow * p.q_ss * 2 + lg * 16) : 0x80000000u;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const unsigned off = (32 * ks + 8 * lg < p.dv) ? qoff + ks * 64 : 0x80000000u;
            qf[qt][ks] = __builtin_amdgcn_raw_buffer_load_b128(rq, off, 0, 0);
        }
    }

    // ---- staging: piece (wave, i) of a tile is 1 KiB = two 512-byte rows; lane l lands at chunk position l of the piece
    // and fetches the chunk the swizzle maps there
    unsigned g_koff[CPT], g_voff[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int byte = (wave * CPT + i) * PIECE + lane * 16;
        const int row = byte / ROWB, chp = (byte % ROWB) / 16;
        const int kc = k_swz<128>(row, chp), vc = v_swz16(row, chp);
        g_koff[i] = (kc * 8 < p.dv) ? (unsigned)(row * p.k_ss * 2 + kc * 16) : 0x80000000u;
        g_voff[i] = (vc * 8 < p.dv) ? (unsigned)(row * p.v_ss * 2 + vc * 16) : 0x80000000u;
    }
    const unsigned k_tile_stride = (unsigned)(kBN * p.k_ss * 2);
    const unsigned v_tile_stride = (unsigned)(kBN * p.v_ss * 2);
    const unsigned piece_base = lds_base + wave * CPT * PIECE;
    auto stage_tile = [&](int j) {
        const unsigned st = (j & 1) * TILE;
#pragma unroll
        for (int i = 0; i < CPT; ++i)
            dma16(rk_w, __builtin_amdgcn_readfirstlane(piece_base + st + i * PIECE), (unsigned)j * k_tile_stride + g_koff[i]);
#pragma unroll
        for (int i = 0; i < CPT; ++i)
            dma16(rv_w, __builtin_amdgcn_readfirstlane(piece_base + VBASE + st + i * PIECE), (unsigned)j * v_tile_stride + g_voff[i]);
    };

    // ---- LDS read addresses (fa_fwd_kernel16.hpp's maps with 512-byte rows)
    unsigned ka[KS], va[DT];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) ka[ks] = lds_base + li * ROWB + k_swz<128>(li, 4 * ks + lg) * 16;
    {
        const int qq = li >> 2, pp = li & 3;
        const int row = 4 * lg + qq;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
            va[dt] = lds_base + VBASE + row * ROWB + v_swz16(row, 2 * dt + (pp >> 1)) * 16 + 8 * (pp & 1);
    }

    f32x4 o_acc[DT][QT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) o_acc[dt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_r[QT], l_r[QT];                    // running row maxima, times scale*log2(e); this lane's share of the row sums
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) { m_r[qt] = -INFINITY; l_r[qt] = 0.f; }
    const float c = p.scale_log2;

    if (nt > 0) stage_tile(0);
    for (int j = 0; j < nt; ++j) {
        dma_wait<0>();
        __syncthreads();                       // tile j has landed and is visible; every wave is done with tile j - 1
        if (j + 1 < nt) stage_tile(j + 1);
        if (j >= my_nt) continue;              // (wave past its last tile: staging duty only)
        const unsigned st = (j & 1) * TILE;
        const int key0 = j * kBN;

        f32x4 s_acc[4][QT];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) s_acc[kt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const u32x4 kf = lds_read_b128(ka[ks] + st + kt * 16 * ROWB);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) s_acc[kt][qt] = T::mfma16(kf, qf[qt][ks], s_acc[kt][qt]);
            }
        }

        const bool need_mask = (key0 + kBN > Sk) || (CAUSAL && key0 + kBN - 1 > q0w + coff);
        u32x4 pf[QT][2];                       // P^T fragments [query tile][32-key half]
        float alpha[QT];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            const int qrow = q0w + 16 * qt + li;
            const int lim = (CAUSAL ? min(Sk - 1, qrow + coff) : Sk - 1) - key0 - 4 * lg;   // key 16 kt + e of this lane is kept iff 16 kt + e <= lim
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (need_mask && 16 * kt + e > lim) s_acc[kt][qt][e] = -INFINITY;
                    mx = fmaxf(mx, s_acc[kt][qt][e]);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_r[qt], mx * c);
            alpha[qt] = (m_new == -INFINITY) ? 1.f : __builtin_amdgcn_exp2f(m_r[qt] - m_new);
            const float m_sub = (m_new == -INFINITY) ? 0.f : m_new;
            float la = 0.f, lb = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s_acc[kt][qt][0], c, -m_sub));
                const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s_acc[kt][qt][1], c, -m_sub));
                const float p2 = __builtin_amdgcn_exp2f(__builtin_fmaf(s_acc[kt][qt][2], c, -m_sub));
                const float p3 = __builtin_amdgcn_exp2f(__builtin_fmaf(s_acc[kt][qt][3], c, -m_sub));
                la += p0 + p2;
                lb += p1 + p3;
                pf[qt][kt >> 1][2 * (kt & 1)] = T::pack2(p0, p1);
                pf[qt][kt >> 1][2 * (kt & 1) + 1] = T::pack2(p2, p3);
            }
            l_r[qt] = l_r[qt] * alpha[qt] + (la + lb);
            m_r[qt] = m_new;
        }
        if (__builtin_amdgcn_ballot_w64(alpha[0] != 1.f || alpha[QT - 1] != 1.f) != 0) {      // some row of the wave raised its maximum
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) o_acc[dt][qt] *= alpha[qt];
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const u32x2 lo = lds_read_tr16_b64(va[dt] + st + 32 * kb * ROWB);
                const u32x2 hi = lds_read_tr16_b64(va[dt] + st + (32 * kb + 16) * ROWB);
                const u32x4 vf = {lo[0], lo[1], hi[0], hi[1]};
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) o_acc[dt][qt] = T::mfma16(vf, pf[qt][kb], o_acc[dt][qt]);
            }
    }

    // ---- epilogue (as fa_fwd_kernel16.hpp): combine the lane groups' row sums, normalise, store O and the LSE
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        float l = l_r[qt];
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        const float inv = (l > 0.f) ? p.out_scale / l : 0.f;        // flash_attn_cutlass.cu:446-452 guard
        const int qrow = q0w + 16 * qt + li;
        if (p.lse != nullptr && lg == 0 && qrow < S)
            p.lse[((long long)b * p.H + h) * S + qrow] = (l > 0.f) ? (m_r[qt] + __builtin_amdgcn_logf(l)) * 0.6931471805599453f : -INFINITY;
        elem_t* orow = oh + (long long)qrow * p.o_ss;
#pragma unroll
        for (int dt = 0; dt < DT; dt += 2) {
            const f32x4 oa = o_acc[dt][qt], ob = o_acc[dt + 1][qt];
            unsigned a0 = T::pack2(oa[0] * inv, oa[1] * inv), a1 = T::pack2(oa[2] * inv, oa[3] * inv);
            unsigned b0 = T::pack2(ob[0] * inv, ob[1] * inv), b1 = T::pack2(ob[2] * inv, ob[3] * inv);
            auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
            auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
            u32x4 outv = {s0[0], s1[0], s0[1], s1[1]};
            const int col = (lg & 1) ? (16 * (dt + 1) + 4 * (lg - 1)) : (16 * dt + 4 * lg);
            if (qrow < S && col < p.dv) *reinterpret_cast<u32x4*>(orow + col) = outv;
        }
    }
}

}  // namespace fa
