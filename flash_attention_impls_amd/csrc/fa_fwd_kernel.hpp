// FlashAttention forward for gfx950 (MI355X / CDNA4) -- device code.
//
// Replaces (behaviourally, not textually) the reference's fused forward kernels:
//   _fwd_kernel                  code/triton_fa2/FA2-triton.py:25-93
//   flash_attn_cutlass_kernel<D> code/cutlass_cuda_fa1/run/flash_attn_cutlass.cu:346-453
//
// Structure (one workgroup = one 256-row query block of one (batch, head)):
//   * template parameter QB = 32-row query blocks per wave: QB = 1 -> 8 waves (two per SIMD, <= 256 registers),
//     QB = 2 -> 4 waves (one per SIMD, the whole 512-register file); with QB = 2 every K / V^T fragment read from
//     LDS feeds two MFMAs.  The workgroup walks 64-key K/V tiles; each wave processes them as two 32-key
//     half-tiles ("blocks").
//   * S^T = K . Q^T  with v_mfma_f32_32x32x16 (K fragment = A operand from LDS via ds_read_b128, Q fragment =
//     B operand, resident in registers).  With this orientation the accumulator holds, per lane, 16 keys x ONE
//     query (query = lane & 31), so the softmax is lane-local (one exchange with lane ^ 32 for row statistics).
//   * O^T += V^T . P^T : the S^T accumulator, converted to 16-bit pairs, IS the B operand (no LDS round trip
//     for P); V^T fragments come from the row-major V tile in LDS through ds_read_b64_tr_b16.
//   * software pipeline over blocks n: one straight-line region sequence issues the MFMAs of S(n) and of PV(n-2)
//     with the softmax VALU work of block n-1 in between, so exp/convert work hides in the MFMA shadows.
//   * fixed softmax reference: a row's reference is set once from its first key block (+ a type-dependent bias)
//     and the hot loop never tracks a running max or rescales O; if a row's scores later rise too far above it
//     (block sum of P >= kPLimit) the whole workgroup redoes its query block with an exact online-softmax loop.
//   * K/V tiles: global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds, coalesced along head_dim, OOB rows land
//     as zeros), XOR-swizzled through the per-lane SOURCE address so that both the row reads of K and the
//     transposed reads of V are bank-conflict free; 4-deep ring, ONE barrier per 64-key tile, each tile's DMA
//     has two iterations to land.
//   * exp2 with scale*log2(e) folded in, deferred 1/l normalisation, causal block skipping + diagonal-only
//     masking, LSE output.
#pragma once
#include "fa_build_guard.hpp"
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace fa {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
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
    int B, H, S;             // S: query rows
    int Sk;                  // keys (= S for the reference's operator); causal mask bottom-right aligned: key <= query + Sk - S
    int G;                   // query heads per key/value head (1 = the reference's case): K, V have H / G heads
    int dv;                  // valid head_dim (a multiple of 16, <= the kernel's compiled D): columns dv .. D-1 of every
                             // Q / K / V row are read as zeros (buffer offsets pushed out of range) and not stored in O
    int nqb;                 // ceil(S / 256)
    int unpaired;            // causal launches only: 1 = one query block per workgroup (small grids), 0 = block pairs
    int hsplit;              // virtual heads per head of the grid mapping (wg_decode): 1 unless B*H is small and not a multiple of 8
    int bh;                  // B*H
    // element strides (innermost head_dim stride is 1)
    long long q_sb, q_sh, q_ss;
    long long k_sb, k_sh, k_ss;
    long long v_sb, v_sh, v_ss;
    long long o_sb, o_sh, o_ss;
    float scale;             // softmax scale (1/sqrt(D) by default: FA2-triton.py:183)
    float scale_log2;        // scale * log2(e)
    float out_scale;         // multiplies O (V dequantisation scale of the fp8 entry; 1 otherwise)
};

template <class To, class From> __device__ __forceinline__ To bitcast(From f) { return __builtin_bit_cast(To, f); }

// Workgroup -> (head, block of the head).  blockIdx % 8 is the XCD a workgroup lands on (round-robin dispatch), and all
// blocks of a (batch, head) slice go to one XCD, so that its K / V stay in that XCD's L2: head = 8 * (slot / per_head) +
// xcd.  With a head count that is not a multiple of 8 that leaves XCDs idle (4 heads: half the chip); the host then
// sets hsplit = 2, 4 or 8 virtual heads per head (heads * hsplit a multiple of 8), virtual head r of a head owning its
// blocks t = r, r + hsplit, ...  Returns false for the padding workgroups of the grid.
__device__ __forceinline__ bool wg_decode(int bid, int heads, int per_head, int hsplit, int& head, int& t)
{
    const int xcd = bid & 7, slot = bid >> 3;
    const int pvh = (per_head + hsplit - 1) / hsplit;           // blocks per virtual head
    const int hl = slot / pvh;
    const int vhead = hl * 8 + xcd;
    head = vhead / hsplit;
    t = (slot - hl * pvh) * hsplit + (vhead - head * hsplit);
    return head < heads && t < per_head;
}

// kPBias : log2 offset added to the softmax reference so that P = exp2(score - ref) starts at 2^-kPBias and has
//          room to grow while the reference stays fixed (fp16 P saturates at 65504; bf16 has the fp32 range).
// kPLimit: a per-lane block sum of P at or above this value flags the fixed-reference pass as unusable.
struct TypeBF16 {
    static constexpr float kPBias = 0.0f;
    static constexpr float kPLimit = 1.152921504606846976e18f;   // 2^60: O stays far below the fp32 range
    static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
#if defined(FA_ABL_MFMA16)   // timing-only: two 16x16x32 MFMAs (same FLOPs, same operands) in place of one 32x32x16
        typedef __attribute__((ext_vector_type(4))) float f32x4;
        f32x4 c0 = {c[0], c[1], c[2], c[3]}, c1 = {c[4], c[5], c[6], c[7]};
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bitcast<bf16x8>(a), bitcast<bf16x8>(b), c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bitcast<bf16x8>(a), bitcast<bf16x8>(b), c1, 0, 0, 0);
        c[0] = c0[0]; c[1] = c0[1]; c[2] = c0[2]; c[3] = c0[3];
        c[4] = c1[0]; c[5] = c1[1]; c[6] = c1[2]; c[7] = c1[3];
        return c;
#else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(bitcast<bf16x8>(a), bitcast<bf16x8>(b), c, 0, 0, 0);
#endif
    }
    static __device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(bitcast<bf16x8>(a), bitcast<bf16x8>(b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ unsigned pack2(float a, float b) {
        f32x2 f = {a, b};
        return bitcast<unsigned>(__builtin_convertvector(f, bf16x2));
    }
    // 16x16x32 MFMA with an arch-VGPR destination and the B operand in the accumulator file, through inline asm:
    // in a 512-register kernel hipcc gives every builtin MFMA an AGPR destination, which costs one v_accvgpr_read
    // per element that VALU code consumes.  The caller keeps the result away from non-MFMA readers for the
    // MFMA -> VALU wait states (nothing inside or after the string is padded by the compiler).
    static __device__ __forceinline__ void mfma16_v_first(f32x4& d, u32x4 a, u32x4 b) {
        asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "a"(b));
    }
    static __device__ __forceinline__ void mfma16_v_acc(f32x4& d, u32x4 a, u32x4 b) {
        asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "a"(b));
    }
    static __device__ __forceinline__ void mfma16_v_init(f32x4& d, u32x4 a, u32x4 b, f32x4 c) {   // d = a.b + c
        asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "a"(b), "v"(c));
    }
};
struct TypeF16 {
    static constexpr float kPBias = 6.0f;
    static constexpr float kPLimit = 60000.0f;
    static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(bitcast<f16x8>(a), bitcast<f16x8>(b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(bitcast<f16x8>(a), bitcast<f16x8>(b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ unsigned pack2(float a, float b) {
        f32x2 f = {a, b};
        return bitcast<unsigned>(__builtin_convertvector(f, f16x2));
    }
    static __device__ __forceinline__ void mfma16_v_first(f32x4& d, u32x4 a, u32x4 b) {
        asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "a"(b));
    }
    static __device__ __forceinline__ void mfma16_v_acc(f32x4& d, u32x4 a, u32x4 b) {
        asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "a"(b));
    }
    static __device__ __forceinline__ void mfma16_v_init(f32x4& d, u32x4 a, u32x4 b, f32x4 c) {
        asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "a"(b), "v"(c));
    }
};

constexpr int kBM = 256;          // query rows per workgroup
constexpr int kBN = 64;           // keys per tile
constexpr int kStages = 4;        // LDS ring depth (tiles) for K and for V (a power of two)
template <int QB> constexpr int threads_per_wg() { return 64 * (8 / QB); }
template <int D> constexpr int lds_bytes() { return kStages * 2 /*K,V*/ * kBN * D * 2; }

typedef __attribute__((address_space(3))) char lds_char;

__device__ __forceinline__ u32x4 lds_read_b128(unsigned addr) {
    return *reinterpret_cast<__attribute__((address_space(3))) u32x4*>(addr);
}
__device__ __forceinline__ u32x2 lds_read_b64(unsigned addr) {
    return *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(addr);
}
// S^T tile on fp8 (OCP e4m3) operands: v_mfma_f32_16x16x32_fp8_fp8, lane (i, g) holds A[row i][k = 8 g + 0..7] /
// B[k = 8 g + 0..7][col i] in the eight bytes of a register pair (the bf16 map with one byte per element)
__device__ __forceinline__ f32x4 mfma16_fp8(u32x2 a, u32x2 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(bitcast<long>(a), bitcast<long>(b), c, 0, 0, 0);
}
// The same product over the whole head_dim in ONE instruction: v_mfma_scale_f32_16x16x128_f8f6f4 with fp8 A and B and unit
// block scales (E8M0 127), twice the K = 32 form's rate.  Its k index is only summed over, so any (lane group, byte) ->
// k map is right as long as both operands use the same one: the four 8-byte k-step fragments of the K = 32 form,
// concatenated, are such a map (tools/probes/mfma_f8f6f4_probe.cpp checks the row / column lanes and the unit scales).
typedef __attribute__((ext_vector_type(8))) int i32x8;
__device__ __forceinline__ f32x4 mfma16_fp8_k128(const u32x2 (&a)[4], const u32x2 (&b)[4], f32x4 c) {
    const i32x8 av = {(int)a[0][0], (int)a[0][1], (int)a[1][0], (int)a[1][1], (int)a[2][0], (int)a[2][1], (int)a[3][0], (int)a[3][1]};
    const i32x8 bv = {(int)b[0][0], (int)b[0][1], (int)b[1][0], (int)b[1][1], (int)b[2][0], (int)b[2][1], (int)b[3][0], (int)b[3][1]};
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
}
__device__ __forceinline__ void lds_write_b128(unsigned addr, u32x4 v) {
    *reinterpret_cast<__attribute__((address_space(3))) u32x4*>(addr) = v;
}
__device__ __forceinline__ void lds_write_b32(unsigned addr, unsigned v) {
    *reinterpret_cast<__attribute__((address_space(3))) unsigned*>(addr) = v;
}
// Workgroup-wide OR of a per-lane predicate through `nwaves` flag words at LDS address `flags` (which nothing else may
// touch until every wave has passed the second barrier).  Replaces __syncthreads_or: the device-library version keeps
// the packed work-item ids alive across the whole kernel for its linear-id computation (two VGPRs, spilled here).
__device__ __forceinline__ bool wg_any(bool pred, unsigned flags, int wave, int lane, int nwaves) {
    const bool wave_any = __builtin_amdgcn_ballot_w64(pred) != 0;
    __syncthreads();
    if (lane == 0) lds_write_b32(flags + 4 * wave, wave_any ? 1u : 0u);
    __syncthreads();
    unsigned any = 0;
    for (int w = 0; w < nwaves; w += 4) {
        const u32x4 f = lds_read_b128(flags + 4 * w);
        any |= f[0] | f[1] | f[2] | f[3];
    }
    return __builtin_amdgcn_readfirstlane(any) != 0;
}
__device__ __forceinline__ u32x2 lds_read_tr16_b64(unsigned addr) {
    s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<__attribute__((address_space(3))) s16x4*>(addr));
    return bitcast<u32x2>(t);
}

// LDS-DMA: 64 lanes x 16 bytes from per-lane buffer offsets `voff` to LDS [lds_addr, lds_addr + 1024)
// (wave-uniform destination, lane l lands at lds_addr + 16 l).  Issued through inline asm so that hipcc does
// not count it: its own bookkeeping would drain every DMA (s_waitcnt vmcnt(0)) in front of the next ds_read.
// The kernel waits for its DMAs itself (dma_wait) before the barrier that publishes a tile.
// M0 (LDS base of the transfer) is compiler-reserved: written in the same statement that uses it.
__device__ __forceinline__ void dma16(u32x4 rsrc, unsigned lds_addr, unsigned voff) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds"
                 :: "v"(voff), "s"(rsrc), "s"(lds_addr) : "memory");
}
// wait until at most N of this wave's DMA instructions are still in flight (vmcnt counts in issue order)
template <int N> __device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// buffer descriptor words for raw (stride 0) access to [base, base + bytes), built from wave-uniform values
__device__ __forceinline__ u32x4 make_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    u32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    r[2] = __builtin_amdgcn_readfirstlane(bytes);
    r[3] = 0x00020000u;
    return r;
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

// experiment helpers (FA_ASM_SMFMA): bf16 32x32x16 MFMA with an arch-VGPR destination, issued through inline asm
__device__ __forceinline__ void asm_mfma_bf16_first(f32x16& d, u32x4 a, u32x4 b) {
    asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b));
}
__device__ __forceinline__ void asm_mfma_bf16_acc(f32x16& d, u32x4 a, u32x4 b) {
    asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
}

// s_setprio PRIO executed by the waves with `late` != 0 only, without control flow the compiler can see (a branch of its own
// inside the statement: a C++ `if` at a region boundary of a pipelined loop splits the scheduling region)
template <int PRIO> __device__ __forceinline__ void setprio_if(int late) {
    int tmp;        // (the flag travels in a vector register: under scalar-register pressure hipcc parks uniform values there anyway, and an "s" input it has parked does not assemble)
    asm volatile("v_readfirstlane_b32 %0, %1\n\ts_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 .Lfa_prio_%=\n\ts_setprio %2\n.Lfa_prio_%=:"
                 : "=&s"(tmp) : "v"(late), "n"(PRIO) : "scc");
}

template <int V> using IC = std::integral_constant<int, V>;

// This lane's index (0..63), produced in place by an opaque instruction pair.  Nothing derived from it can be hoisted
// above the call, so lane coordinates needed again after a long loop (epilogue, fallback) are recomputed there instead of
// being kept alive -- i.e. spilled -- across it, and threadIdx.x need not stay live past the kernel's first lines.
__device__ __forceinline__ int lane_here() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

struct SmState {
    f32x16 sv;        // (masked) scores of one 32-key block
    float rs0, rs1;   // this lane's partial row sums (two independent chains)
};

// NWV = waves per workgroup: 8 / QB by default (a 256-row query block); NWV = 4 with QB = 1 gives 128-row workgroups for
// launches that would otherwise leave CUs without a workgroup (the host picks: fa_capi.hip, small_rows_for)
template <class T, int D, bool CAUSAL, int QB, int NWV = 8 / QB>
__global__ __launch_bounds__(64 * NWV, (NWV * QB == 8) ? 2 / QB : 2) void fa_fwd_kernel(const FwdParams p)
{
    constexpr int NWAVES = NWV;
    constexpr int kBM = 32 * QB * NWV;         // query rows per workgroup (shadows the namespace-level 256)
    constexpr bool SIMD_PAIRS = NWV == 8 && QB == 1;   // waves w and w + 4 share a SIMD
    constexpr int KS = D / 16;                 // k-steps of the QK^T product
    constexpr int DB = D / 32;                 // 32-wide head_dim blocks of O^T
    constexpr int ROWB = D * 2;                // bytes per K/V row in LDS
    constexpr int TILE = kBN * ROWB;           // bytes per K (or V) tile
    constexpr int PIECE = 1024;                // bytes one DMA wave-instruction moves
    constexpr int CPT = TILE / PIECE / NWAVES; // DMA pieces per wave per tile
    static_assert(CPT >= 1, "tile too small");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_char*)smem;   // K ring [kStages][TILE], then V ring

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- workgroup -> (head, query block(s)).  blockIdx % 8 labels the XCD group: all query blocks of one
    // head share an XCD (its L2 holds that head's K/V).  Causal: a workgroup processes the pair of query
    // blocks (nqb-1-t, t) one after the other, so every workgroup carries the same number of key tiles
    // (nqb+1 of them) and the grid is balanced however the dispatcher places it.
    // causal: a workgroup takes the query-block pair (nqb-1-t, t) -- equal work for every workgroup -- unless the launch is
    // so small that every query block can have a CU of its own (p.unpaired: then the longest block alone sets the time)
    const bool paired = CAUSAL && !p.unpaired;
    const int wg_per_head = paired ? (p.nqb + 1) / 2 : p.nqb;
    int head, tq;
    if (!wg_decode(blockIdx.x, p.bh, wg_per_head, p.hsplit, head, tq)) return;
    const int b = head / p.H;
    const int h = head - b * p.H;
    const int S = p.S;
    const int Sk = p.Sk;
    const int coff = CAUSAL ? Sk - S : 0;      // key <= query + coff
    const int n_pass = (paired && (p.nqb - 1 - tq != tq)) ? 2 : 1;

    using elem_t = unsigned short;
    const elem_t* qh = reinterpret_cast<const elem_t*>(p.q) + b * p.q_sb + h * p.q_sh;
    const elem_t* kh = reinterpret_cast<const elem_t*>(p.k) + b * p.k_sb + (h / p.G) * p.k_sh;
    const elem_t* vh = reinterpret_cast<const elem_t*>(p.v) + b * p.v_sb + (h / p.G) * p.v_sh;
    elem_t* oh = reinterpret_cast<elem_t*>(p.o) + b * p.o_sb + h * p.o_sh;

    const unsigned q_bytes = (unsigned)(((long long)(S - 1) * p.q_ss + p.dv) * 2);
    const unsigned k_bytes = (unsigned)(((long long)(Sk - 1) * p.k_ss + p.dv) * 2);
    const unsigned v_bytes = (unsigned)(((long long)(Sk - 1) * p.v_ss + p.dv) * 2);
    __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<elem_t*>(qh), 0, q_bytes, 0x00020000);
    const u32x4 rk_w = make_rsrc(kh, k_bytes);
    const u32x4 rv_w = make_rsrc(vh, v_bytes);

    u32x4 qf[QB][KS];          // Q fragments of the current pass (the next pass's are loaded behind its epilogue)

  for (int pass = 0; pass < n_pass; ++pass) {
    const int qb = CAUSAL ? ((pass == 0) ? p.nqb - 1 - tq : tq) : tq;      // (unpaired: one pass, longest blocks first)
    // lane coordinates, made opaque per pass: everything derived from them (addresses, mask indices) is then
    // recomputed inside the pass instead of being hoisted out of the pass loop and spilled around it
    int lane = tid & 63;
    asm volatile("" : "+v"(lane));
    const int r = lane & 31;
    const int hh = lane >> 5;

    // 32-row query blocks of this wave.  QB = 1: waves w and w+4 share a SIMD and get row blocks that sum to 7
    // so that the causal diagonal tile is balanced across SIMDs.  QB = 2: wave w owns row blocks 2w and 2w+1.
    int q0[QB];
    if constexpr (QB == 1) {
        q0[0] = qb * kBM + ((CAUSAL && SIMD_PAIRS) ? (wave < 4 ? wave : 11 - wave) : wave) * 32;
    } else {
#pragma unroll
        for (int qi = 0; qi < QB; ++qi) q0[qi] = qb * kBM + (wave * QB + qi) * 32;
    }
    const int q_first = q0[0], q_last = q0[QB - 1];

    const int kv_end_wg = CAUSAL ? max(0, min(Sk, qb * kBM + kBM + coff)) : Sk;   // (coff < 0: the first -coff queries see no key)
    const int nt = (kv_end_wg + kBN - 1) / kBN;                       // tiles the workgroup stages
    const int kv_end_w = (q_first >= S) ? 0 : (CAUSAL ? max(0, min(Sk, q_last + 32 + coff)) : Sk);
    const int my_nt = (kv_end_w + kBN - 1) / kBN;                     // tiles this wave computes on

    // ---- Q fragments (B operand of S^T = K Q^T): lane (r,hh) holds Q[row0 + r][16 ks + 8 hh + 0..7]
    auto load_q = [&](int qblk) {
#pragma unroll
        for (int qi = 0; qi < QB; ++qi) {
            int row0;
            if constexpr (QB == 1) row0 = qblk * kBM + ((CAUSAL && SIMD_PAIRS) ? (wave < 4 ? wave : 11 - wave) : wave) * 32;
            else row0 = qblk * kBM + (wave * QB + qi) * 32;
            const int qrow = row0 + r;
            // rows past the end of the sequence get an offset outside the descriptor: they read as zero
            const unsigned qoff = (qrow < S) ? (unsigned)((long long)qrow * p.q_ss * 2 + hh * 16) : 0x80000000u;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                qf[qi][ks] = __builtin_amdgcn_raw_buffer_load_b128(rq, (16 * ks + 8 * hh < p.dv) ? qoff + ks * 32 : 0x80000000u, 0, 0);
        }
    };
    if (pass == 0) load_q(qb);

    // ---- K/V staging by LDS-DMA.  One wave-instruction writes a 1-KiB piece linearly (M0 base + lane*16), so
    // the XOR swizzle is applied to the per-lane SOURCE address: the lane that lands on LDS chunk position c' of
    // row `row` fetches global chunk  c' ^ f(row)  of that row.  Rows past the end of the sequence fall outside
    // the descriptor and are written as zeros (the host guarantees (S + 256) * row_stride_bytes < 2^31).
    constexpr int VBASE = kStages * TILE;
    unsigned g_koff[CPT], g_voff[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int byte = (wave * CPT + i) * PIECE + lane * 16;
        const int row = byte / ROWB, chp = (byte % ROWB) / 16;
        // (chunks past the valid head_dim: an offset no tile index brings back into the descriptor -> zeros)
        g_koff[i] = (k_swz<D>(row, chp) * 8 < p.dv) ? (unsigned)(row * p.k_ss * 2 + k_swz<D>(row, chp) * 16) : 0x80000000u;
        g_voff[i] = (v_swz<D>(row, chp) * 8 < p.dv) ? (unsigned)(row * p.v_ss * 2 + v_swz<D>(row, chp) * 16) : 0x80000000u;
    }
    const unsigned k_tile_stride = (unsigned)(kBN * p.k_ss * 2);
    const unsigned v_tile_stride = (unsigned)(kBN * p.v_ss * 2);
    const unsigned piece_base = lds_base + wave * CPT * PIECE;       // wave-uniform
    auto dma_k = [&](int j, unsigned stage_off) {
#pragma unroll
        for (int i = 0; i < CPT; ++i)
            dma16(rk_w, __builtin_amdgcn_readfirstlane(piece_base + stage_off + i * PIECE), (unsigned)j * k_tile_stride + g_koff[i]);
    };
    auto dma_v = [&](int j, unsigned stage_off) {
#pragma unroll
        for (int i = 0; i < CPT; ++i)
            dma16(rv_w, __builtin_amdgcn_readfirstlane(piece_base + VBASE + stage_off + i * PIECE), (unsigned)j * v_tile_stride + g_voff[i]);
    };

    // ---- LDS read addresses (they carry the ring-stage offset of the tile currently being read)
    // K fragment (A operand): lane (r,hh) reads K[half*32 + r][16 ks + 8 hh + 0..7] = chunk 2ks+hh of row
    unsigned ka[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) ka[ks] = lds_base + r * ROWB + k_swz<D>(r, 2 * ks + hh) * 16;
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
            va[db] = lds_base + VBASE + row * ROWB + v_swz<D>(row, ch) * 16 + 8 * (pp & 1);
        }
    }

    // K fragments are read in groups of 4 k-steps; V^T fragments per 16-key k-step (DB blocks).
    // One register set each: a group is re-read right after the MFMAs that consumed the previous one
    // were issued, and lands while the MFMAs of the other product run.
    constexpr int GPB = KS / 4;           // K groups per 32-key block (2 @ D=128, 1 @ D=64)
    u32x4 kf[4];
    u32x4 vf[DB];
    auto read_kgroup = [&] __device__ (auto half_c, auto g_c) {
        constexpr int half = decltype(half_c)::value, g = decltype(g_c)::value;
#if defined(FA_ABL_NOLDS)     // timing-only ablation builds (wrong results): see tools/ab_bench.py
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(kf[i]));
        return;
#endif
#pragma unroll
        for (int i = 0; i < 4; ++i) kf[i] = lds_read_b128(ka[4 * g + i] + half * 32 * ROWB);
    };
    auto read_vstep = [&] __device__ (auto s_c) {
        constexpr int s = decltype(s_c)::value;
#if defined(FA_ABL_NOLDS)
#pragma unroll
        for (int db = 0; db < DB; ++db) asm volatile("" : "+v"(vf[db]));
        return;
#endif
#pragma unroll
        for (int db = 0; db < DB; ++db) {
            u32x2 lo = lds_read_tr16_b64(va[db] + (16 * s) * ROWB);
            u32x2 hi = lds_read_tr16_b64(va[db] + (16 * s + 8) * ROWB);
            vf[db] = u32x4{lo[0], lo[1], hi[0], hi[1]};
        }
    };

    f32x16 o_acc[QB][DB];
#pragma unroll
    for (int qi = 0; qi < QB; ++qi)
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int i = 0; i < 16; ++i) o_acc[qi][db][i] = 0.f;
    float m_c[QB];             // softmax reference per row, already multiplied by scale*log2(e)
    float l_part[QB];          // this lane's share of the row sum (lanes r and r+32 are combined at the end)
#pragma unroll
    for (int qi = 0; qi < QB; ++qi) { m_c[qi] = -INFINITY; l_part[qi] = 0.f; }
    const float c = p.scale_log2;

    f32x16 s_acc[QB][2];       // S^T of block n (being produced) and n-1 (being consumed)
    u32x4 pf[QB][2][2];        // P^T fragments (2 k-steps of 16 keys) of block n-1 (produced) and n-2 (consumed)

    // ---- softmax of one 32-key block, in pieces that the pipeline block interleaves with its MFMA groups.
    // FAST PATH: the softmax reference m_c of a row is fixed once, from the row maximum of key block 0
    // (+ kPBias), and never raised afterwards: P = exp2(score*c - m_c) may exceed 1 (bf16 / fp32 have the
    // range), O and l simply accumulate at that scale, and no running max / rescale work exists in the loop.
    // If a block sum ever reaches kPLimit (scores far above the first block's), `p_peak` records it and the
    // whole workgroup redoes its query block with the exact online-softmax loop further below.
    float p_peak = 0.f;
    auto sm_begin = [&] __device__ (auto mask_c, auto first_c, auto qi_c, const f32x16& s_in, int key0, SmState& st) {
        constexpr bool MASK = decltype(mask_c)::value, FIRST = decltype(first_c)::value;
        constexpr int qi = decltype(qi_c)::value;
        st.sv = s_in;
        st.rs0 = 0.f;
        st.rs1 = 0.f;
        if constexpr (MASK) {
            const int qrow = q0[qi] + r;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = key0 + (i & 3) + 8 * (i >> 2) + 4 * hh;
                const bool dead = (key >= Sk) || (CAUSAL && key > qrow + coff);
                if (dead) st.sv[i] = -INFINITY;
            }
        }
        if constexpr (FIRST) {            // key block 0 fixes the reference
            float mx = fmaxf(st.sv[0], st.sv[1]);
#pragma unroll
            for (int i = 2; i < 16; ++i) mx = fmaxf(mx, st.sv[i]);
            auto sw = __builtin_amdgcn_permlane32_swap(bitcast<unsigned>(mx), bitcast<unsigned>(mx), false, false);
            mx = fmaxf(bitcast<float>(sw[0]), bitcast<float>(sw[1]));
            m_c[qi] = (mx == -INFINITY) ? 0.f : __builtin_fmaf(mx, c, T::kPBias);
        }
    };
    // exponentials of scores [i0, i1) (i0, i1 even) against reference m_ref (already scaled); each pair is
    // converted to 16 bits at once and dropped into its word of the P^T fragments
    auto sm_exp = [&] __device__ (SmState& st, u32x4 (&pw)[2], auto i0_c, auto i1_c, float m_ref) {
        constexpr int i0 = decltype(i0_c)::value, i1 = decltype(i1_c)::value;
#pragma unroll
        for (int i = i0; i < i1; i += 2) {
#if defined(FA_ABL_NOVALU)
            if (i == i0) { asm volatile("" : "+v"(st.sv)); pw[i >> 3][(i & 7) >> 1] = bitcast<unsigned>(st.sv[i]); }
            continue;
#endif
#if defined(FA_ABL_NOEXP)
            const float p0 = __builtin_fmaf(st.sv[i], c, -m_ref);
            const float p1 = __builtin_fmaf(st.sv[i + 1], c, -m_ref);
#else
            const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(st.sv[i], c, -m_ref));
            const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(st.sv[i + 1], c, -m_ref));
#endif
#if !defined(FA_ABL_NOSUM)
#if defined(FA_ASM_ADD)
            // pinned single-instruction adds (experiment).  NOT safe as written: hipcc pads no hazards for an
            // asm consumer of a transcendental result (v_exp_f32 -> VALU read needs a wait state).
            asm volatile("s_nop 0\n\tv_add_f32 %0, %0, %1" : "+v"(st.rs0) : "v"(p0));
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(st.rs1) : "v"(p1));
#else
            st.rs0 += p0;           // two independent chains; the build uses -fno-slp-vectorize so that they
            st.rs1 += p1;           // stay single-issue adds next to their exponentials
#endif
#endif
            pw[i >> 3][(i & 7) >> 1] = T::pack2(p0, p1);
        }
    };
    auto sm_end = [&] __device__ (auto qi_c, SmState& st) {
        constexpr int qi = decltype(qi_c)::value;
        const float rs = st.rs0 + st.rs1;
        l_part[qi] += rs;
        p_peak = fmaxf(p_peak, rs);
    };

    // One pipeline block n (HALF = n & 1 selects the 32-key half of the 64-key tiles):
    //   MFMA: PV(n-2) (V tile j-1, rows 32*HALF..) and S(n) (K tile j, rows 32*HALF..), j = n >> 1
    //   VALU: softmax(n-1)
    //   LDS : each fragment group is re-read right after the MFMAs that consumed it were issued and lands
    //         while the other product's MFMAs run; the last two reads prefetch the next block's first groups.
    //         Odd blocks advance the address registers to the next tiles (dk/dv = ring-stage byte deltas).
    // The block is cut into regions (one MFMA group + a slice of the softmax each) by scheduling fences, so that
    // the compiler interleaves inside a region but keeps the fragment reads one group ahead.
    auto mfma_pv = [&] __device__ (auto half_c, auto kstep_c) {
        constexpr int HALFv = decltype(half_c)::value, kstep = decltype(kstep_c)::value;
#pragma unroll
        for (int qi = 0; qi < QB; ++qi)
#pragma unroll
            for (int db = 0; db < DB; ++db) o_acc[qi][db] = T::mfma(vf[db], pf[qi][HALFv][kstep], o_acc[qi][db]);
    };
    auto mfma_s = [&] __device__ (auto half_c, auto g_c) {
        constexpr int HALFv = decltype(half_c)::value, g = decltype(g_c)::value;
#pragma unroll
        for (int qi = 0; qi < QB; ++qi) {
#if !defined(FA_ASM_SMFMA)
            if constexpr (g == 0) {
#pragma unroll
                for (int i = 0; i < 16; ++i) s_acc[qi][HALFv][i] = 0.f;
            }
#endif
#if defined(FA_ASM_SMFMA)
            // experiment (QB = 2): S accumulates in arch VGPRs through inline-asm MFMAs, while the compiler's own
            // (AGPR-form) MFMAs keep O in the accumulator file -- no v_accvgpr_read per score.  bf16 only.
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (g == 0 && i == 0) asm_mfma_bf16_first(s_acc[qi][HALFv], kf[i], qf[qi][4 * g + i]);
                else asm_mfma_bf16_acc(s_acc[qi][HALFv], kf[i], qf[qi][4 * g + i]);
            }
#else
#pragma unroll
            for (int i = 0; i < 4; ++i) s_acc[qi][HALFv] = T::mfma(kf[i], qf[qi][4 * g + i], s_acc[qi][HALFv]);
#endif
        }
    };
    auto advance = [&](int dk, int dv) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) ka[ks] += dk;
#pragma unroll
        for (int db = 0; db < DB; ++db) va[db] += dv;
    };
    auto block = [&] __device__ (auto half_c, auto do_s_c, auto do_sm_c, auto do_pv_c, auto mask_c, auto first_c, int n, int dk, int dv) {
        constexpr int HALF = decltype(half_c)::value;
        constexpr bool DO_S = decltype(do_s_c)::value, DO_SM = decltype(do_sm_c)::value, DO_PV = decltype(do_pv_c)::value;
        SmState st[QB];
        const int key0 = (n - 1) * 32;
        // ---- region 0: PV k-step 0 | softmax head
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (DO_PV) {
            mfma_pv(IC<HALF>{}, IC<0>{});
            read_vstep(IC<2 * HALF + 1>{});
        }
        if constexpr (DO_SM) {
            sm_begin(mask_c, first_c, IC<0>{}, s_acc[0][HALF ^ 1], key0, st[0]);
            sm_exp(st[0], pf[0][HALF ^ 1], IC<0>{}, IC<4>{}, m_c[0]);
            if constexpr (QB == 2) {
                sm_begin(mask_c, first_c, IC<QB - 1>{}, s_acc[QB - 1][HALF ^ 1], key0, st[QB - 1]);
                sm_exp(st[QB - 1], pf[QB - 1][HALF ^ 1], IC<0>{}, IC<4>{}, m_c[QB - 1]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- region 1: S group 0 | exponentials
        if constexpr (DO_S) mfma_s(IC<HALF>{}, IC<0>{});
        if constexpr (GPB == 2) {
            if constexpr (DO_S) read_kgroup(IC<HALF>{}, IC<1>{});
            if constexpr (DO_SM) {
#pragma unroll
                for (int qi = 0; qi < QB; ++qi) sm_exp(st[qi], pf[qi][HALF ^ 1], IC<4>{}, IC<8>{}, m_c[qi]);
            }
        } else {
            if constexpr (HALF == 1) advance(dk, dv);
            read_kgroup(IC<HALF ^ 1>{}, IC<0>{});             // next block's K fragments
            if constexpr (DO_SM) {
#pragma unroll
                for (int qi = 0; qi < QB; ++qi) sm_exp(st[qi], pf[qi][HALF ^ 1], IC<4>{}, IC<12>{}, m_c[qi]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- region 2: PV k-step 1 | exponentials
        if constexpr (DO_PV) mfma_pv(IC<HALF>{}, IC<1>{});
        if constexpr (GPB == 2) {
            if constexpr (HALF == 1) advance(dk, dv);
            read_vstep(IC<2 * (HALF ^ 1)>{});                 // next block's first V^T fragments
            if constexpr (DO_SM) {
#pragma unroll
                for (int qi = 0; qi < QB; ++qi) sm_exp(st[qi], pf[qi][HALF ^ 1], IC<8>{}, IC<12>{}, m_c[qi]);
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- region 3: S group 1 | exponentials
            if constexpr (DO_S) mfma_s(IC<HALF>{}, IC<1>{});
            read_kgroup(IC<HALF ^ 1>{}, IC<0>{});             // next block's first K fragments
        } else {
            read_vstep(IC<2 * (HALF ^ 1)>{});
        }
        if constexpr (DO_SM) {
            sm_exp(st[0], pf[0][HALF ^ 1], IC<12>{}, IC<16>{}, m_c[0]);
            sm_end(IC<0>{}, st[0]);
            if constexpr (QB == 2) {
                sm_exp(st[QB - 1], pf[QB - 1][HALF ^ 1], IC<12>{}, IC<16>{}, m_c[QB - 1]);
                sm_end(IC<QB - 1>{}, st[QB - 1]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };

#if defined(FA_ASM_SMFMA)
    // ---- hand-ordered steady-state block for QB = 2, D = 128 (one wave per SIMD: the instruction order IS the
    // schedule).  32 slots of {1 MFMA, half a softmax pair-unit (3-4 VALU), <= 1 LDS read}, each pinned by a fence.
    //   slots  0.. 7: PV k-step 0           | softmax q-block 0, scores 0..7
    //   slots  8..15: S group 0 (asm MFMA)  | softmax q-block 0, scores 8..15 | reads: V^T fragments of k-step 1
    //   slots 16..23: PV k-step 1           | softmax q-block 1, scores 0..7  | reads: K fragments of group 1
    //   slots 24..31: S group 1 (asm MFMA)  | softmax q-block 1, scores 8..15 | reads: next block's V^T k-step 0
    //   tail        : next block's K fragments of group 0
    auto block_hs = [&] __device__ (auto half_c, int n, int dk, int dv) {
        static_assert(QB == 2 && D == 128, "hand-ordered block: QB = 2, head_dim 128");
        constexpr int HALF = decltype(half_c)::value;
        (void)n;
        float t1 = 0.f, p0 = 0.f;
        float rs0[2] = {0.f, 0.f}, rs1[2] = {0.f, 0.f};
        // first half of pair-unit u (u = 0..15): two scaled differences, one exponential
        auto unit_a = [&] __device__ (auto u_c) {
            constexpr int u = decltype(u_c)::value, qi = u >> 3, i = 2 * (u & 7);
            const f32x16& sv = s_acc[qi][HALF ^ 1];
            const float t0 = __builtin_fmaf(sv[i], c, -m_c[qi]);
            t1 = __builtin_fmaf(sv[i + 1], c, -m_c[qi]);
            p0 = __builtin_amdgcn_exp2f(t0);
        };
        // second half: the other exponential, row-sum adds, 16-bit pack
        auto unit_b = [&] __device__ (auto u_c) {
            constexpr int u = decltype(u_c)::value, qi = u >> 3, i = 2 * (u & 7);
            const float p1 = __builtin_amdgcn_exp2f(t1);
            rs0[qi] += p0;
            rs1[qi] += p1;
            pf[qi][HALF ^ 1][i >> 3][(i & 7) >> 1] = T::pack2(p0, p1);
        };
        auto vread = [&] __device__ (auto s_c, auto db_c) {      // one V^T fragment (2 transposed reads)
            constexpr int sidx = decltype(s_c)::value, db = decltype(db_c)::value;
            u32x2 lo = lds_read_tr16_b64(va[db] + (16 * sidx) * ROWB);
            u32x2 hi = lds_read_tr16_b64(va[db] + (16 * sidx + 8) * ROWB);
            vf[db] = u32x4{lo[0], lo[1], hi[0], hi[1]};
        };
        auto kread = [&] __device__ (auto half2_c, auto g_c, auto i_c) {
            constexpr int h2 = decltype(half2_c)::value, g = decltype(g_c)::value, i = decltype(i_c)::value;
            kf[i] = lds_read_b128(ka[4 * g + i] + h2 * 32 * ROWB);
        };
#define FA_FENCE __builtin_amdgcn_sched_barrier(0)
        FA_FENCE;
        // ---- slots 0..7: PV k-step 0
#define FA_PV_SLOT(S, KSTEP, U) \
        { constexpr int qi_ = (S) >> 2, db_ = (S) & 3; \
          o_acc[qi_][db_] = T::mfma(vf[db_], pf[qi_][HALF][KSTEP], o_acc[qi_][db_]); \
          if (((S) & 1) == 0) unit_a(IC<(U)>{}); else unit_b(IC<(U)>{}); }
        FA_PV_SLOT(0, 0, 0) FA_FENCE; FA_PV_SLOT(1, 0, 0) FA_FENCE; FA_PV_SLOT(2, 0, 1) FA_FENCE; FA_PV_SLOT(3, 0, 1) FA_FENCE;
        FA_PV_SLOT(4, 0, 2) FA_FENCE; FA_PV_SLOT(5, 0, 2) FA_FENCE; FA_PV_SLOT(6, 0, 3) FA_FENCE; FA_PV_SLOT(7, 0, 3) FA_FENCE;
        // ---- slots 8..15: S group 0; V^T fragments of k-step 1 (one fragment every other slot)
#define FA_S_SLOT(S, G, U) \
        { constexpr int qi_ = (S) >> 2, i_ = (S) & 3; \
          if ((G) == 0 && i_ == 0) asm_mfma_bf16_first(s_acc[qi_][HALF], kf[i_], qf[qi_][4 * (G) + i_]); \
          else asm_mfma_bf16_acc(s_acc[qi_][HALF], kf[i_], qf[qi_][4 * (G) + i_]); \
          if (((S) & 1) == 0) unit_a(IC<(U)>{}); else unit_b(IC<(U)>{}); }
        FA_S_SLOT(0, 0, 4) vread(IC<2 * HALF + 1>{}, IC<0>{}); FA_FENCE; FA_S_SLOT(1, 0, 4) FA_FENCE;
        FA_S_SLOT(2, 0, 5) vread(IC<2 * HALF + 1>{}, IC<1>{}); FA_FENCE; FA_S_SLOT(3, 0, 5) FA_FENCE;
        FA_S_SLOT(4, 0, 6) vread(IC<2 * HALF + 1>{}, IC<2>{}); FA_FENCE; FA_S_SLOT(5, 0, 6) FA_FENCE;
        FA_S_SLOT(6, 0, 7) vread(IC<2 * HALF + 1>{}, IC<3>{}); FA_FENCE; FA_S_SLOT(7, 0, 7) FA_FENCE;
        // ---- slots 16..23: PV k-step 1; K fragments of group 1
        FA_PV_SLOT(0, 1, 8)  kread(IC<HALF>{}, IC<1>{}, IC<0>{}); FA_FENCE; FA_PV_SLOT(1, 1, 8)  FA_FENCE;
        FA_PV_SLOT(2, 1, 9)  kread(IC<HALF>{}, IC<1>{}, IC<1>{}); FA_FENCE; FA_PV_SLOT(3, 1, 9)  FA_FENCE;
        FA_PV_SLOT(4, 1, 10) kread(IC<HALF>{}, IC<1>{}, IC<2>{}); FA_FENCE; FA_PV_SLOT(5, 1, 10) FA_FENCE;
        FA_PV_SLOT(6, 1, 11) kread(IC<HALF>{}, IC<1>{}, IC<3>{}); FA_FENCE; FA_PV_SLOT(7, 1, 11) FA_FENCE;
        // ---- slots 24..31: S group 1; next block's V^T fragments of its k-step 0
        if constexpr (HALF == 1) advance(dk, dv);
        FA_S_SLOT(0, 1, 12) vread(IC<2 * (HALF ^ 1)>{}, IC<0>{}); FA_FENCE; FA_S_SLOT(1, 1, 12) FA_FENCE;
        FA_S_SLOT(2, 1, 13) vread(IC<2 * (HALF ^ 1)>{}, IC<1>{}); FA_FENCE; FA_S_SLOT(3, 1, 13) FA_FENCE;
        FA_S_SLOT(4, 1, 14) vread(IC<2 * (HALF ^ 1)>{}, IC<2>{}); FA_FENCE; FA_S_SLOT(5, 1, 14) FA_FENCE;
        FA_S_SLOT(6, 1, 15) vread(IC<2 * (HALF ^ 1)>{}, IC<3>{}); FA_FENCE; FA_S_SLOT(7, 1, 15) FA_FENCE;
        // ---- tail: next block's first K fragments; commit the row sums
        kread(IC<HALF ^ 1>{}, IC<0>{}, IC<0>{}); kread(IC<HALF ^ 1>{}, IC<0>{}, IC<1>{});
        kread(IC<HALF ^ 1>{}, IC<0>{}, IC<2>{}); kread(IC<HALF ^ 1>{}, IC<0>{}, IC<3>{});
#pragma unroll
        for (int qi = 0; qi < QB; ++qi) {
            const float rs = rs0[qi] + rs1[qi];
            l_part[qi] += rs;
            p_peak = fmaxf(p_peak, rs);
        }
        FA_FENCE;
#undef FA_PV_SLOT
#undef FA_S_SLOT
#undef FA_FENCE
    };
#endif


    typedef std::true_type Y;
    typedef std::false_type N;
    typedef std::integral_constant<int, 0> half0_t;
    typedef std::integral_constant<int, 1> half1_t;

    // ---- prologue: K(0), V(0) must have landed before iteration 0; K(1), K(2), V(1) follow behind.
    // (for the second query block of a causal pair this was issued behind the first one's main loop)
    auto issue_prologue = [&]() {
        dma_k(0, 0);
        dma_v(0, 0);
        dma_k(1, TILE);
        dma_k(2, 2 * TILE);
        dma_v(1, TILE);
    };
    if (pass == 0) issue_prologue();
    dma_wait<3 * CPT>();        // this wave's pieces of K(0), V(0) have landed ...
    __syncthreads();            // ... and every wave's are visible

    // Iteration j: block 2j | wait, barrier(j) | DMA K(j+3), V(j+2) | block 2j+1.      (stage of tile t = t % 4)
    //   reads : K(j), V(j-1)                        -- published by barrier(j-1) or earlier
    //   block 2j+1 prefetches from K(j+1), V(j)     -- DMA issued in iteration j-2 (or the prologue); each wave
    //                                                  waits for its own pieces (counted vmcnt, leaving the
    //                                                  DMAs of iteration j-1 in flight) before barrier(j)
    //   DMA K(j+3) overwrites K(j-1), DMA V(j+2) overwrites V(j-2): last read in block 2j-1, before barrier(j).
    // The two-iteration flight time hides the HBM/L2 latency of a tile behind ~2 tiles of compute.
    // Every wave executes barrier(j) and its share of the DMA for j = 0..nt-1; the compute of a wave covers
    // its own NT = my_nt tiles: fill (iteration 0), steady loops, drain (iteration NT), then a DMA/barrier tail.
    int stage_k = 0;                               // ring stage of tile j
    int dk = 0, dv = 0;                            // address deltas used by the odd block of iteration j
    auto begin_iter = [&](int j) {
        dk = (stage_k == kStages - 1) ? -(kStages - 1) * TILE : TILE;          // K(j) -> K(j+1)
        dv = (j == 0) ? 0 : ((stage_k == 0) ? -(kStages - 1) * TILE : TILE);   // V(j-1) -> V(j)
    };
    auto sync_and_stage = [&](int j) {             // barrier(j) + DMA of K(j+3), V(j+2)
        dma_wait<2 * CPT>();                       // everything but the previous iteration's DMA has landed ...
#if !defined(FA_ABL_NOBARRIER)
        __syncthreads();                           // ... and is published; last iteration's reads are done
#endif
#if !defined(FA_ABL_NODMA)
        dma_k(j + 3, ((stage_k + 3) & (kStages - 1)) * TILE);
        dma_v(j + 2, ((stage_k + 2) & (kStages - 1)) * TILE);
#endif
    };
    auto end_iter = [&]() { stage_k = (stage_k + 1) & (kStages - 1); };

    const int NT = my_nt;                          // 64-key tiles this wave computes on
    // first block whose softmax needs the mask (causal diagonal of the wave's first row block, or ragged end)
    const int mb = min(CAUSAL ? (max(0, q_first + coff) >> 5) : 0x7fffffff, Sk >> 5);
    const int jm = (mb + 1) >> 1;                  // iteration j softmaxes blocks 2j-1 and 2j
    int j = 0;
    if (NT > 0) {
        read_kgroup(IC<0>{}, IC<0>{});
        // iteration 0 (pipeline fill); softmax(0) fixes the reference
        begin_iter(0);
        block(half0_t{}, Y{}, N{}, N{}, N{}, N{}, 0, 0, 0);
        sync_and_stage(0);
        block(half1_t{}, Y{}, Y{}, N{}, Y{}, Y{}, 1, dk, dv);
        end_iter();
        j = 1;
        const int ja = min(jm, NT);
#if defined(FA_ASM_SMFMA)
        if constexpr (QB == 2 && D == 128) {
            for (; j < ja; ++j) {                  // steady state, no masking: hand-ordered blocks
                begin_iter(j);
                block_hs(half0_t{}, 2 * j, 0, 0);
                sync_and_stage(j);
                block_hs(half1_t{}, 2 * j + 1, dk, dv);
                end_iter();
            }
        }
#endif
        for (; j < ja; ++j) {                      // steady state, no masking
            begin_iter(j);
            block(half0_t{}, Y{}, Y{}, Y{}, N{}, N{}, 2 * j, 0, 0);
            sync_and_stage(j);
            block(half1_t{}, Y{}, Y{}, Y{}, N{}, N{}, 2 * j + 1, dk, dv);
            end_iter();
        }
        for (; j < NT; ++j) {                      // steady state with masking (diagonal / ragged tiles)
            begin_iter(j);
            block(half0_t{}, Y{}, Y{}, Y{}, Y{}, N{}, 2 * j, 0, 0);
            sync_and_stage(j);
            block(half1_t{}, Y{}, Y{}, Y{}, Y{}, N{}, 2 * j + 1, dk, dv);
            end_iter();
        }
        // iteration NT (pipeline drain)
        begin_iter(j);
        block(half0_t{}, N{}, Y{}, Y{}, Y{}, N{}, 2 * j, 0, 0);
        if (j < nt) sync_and_stage(j);
        block(half1_t{}, N{}, N{}, Y{}, N{}, N{}, 2 * j + 1, dk, dv);
        end_iter();
        ++j;
    }
    for (; j < nt; ++j) {                          // remaining tiles of the workgroup: staging duty only
        sync_and_stage(j);
        end_iter();
    }

    // ---- exact fallback (rare): some row's scores rose too far above its first-block reference.
    // The whole workgroup redoes its query block with a plain per-tile online softmax (running max, rescale).
    dma_wait<0>();                                 // no DMA may still be writing LDS past this point
    // (flag words in the last V stage: the next pass's prologue does not write there, and its first later DMA sits
    // behind a barrier)
    if (wg_any(!(p_peak < T::kPLimit), lds_base + VBASE + (kStages - 1) * TILE, wave, tid & 63, NWAVES)) {
        constexpr int KO = 0, VO = 2 * TILE;                  // two stages each: K at KO, V at VO
#pragma unroll
        for (int qi = 0; qi < QB; ++qi) {
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int i = 0; i < 16; ++i) o_acc[qi][db][i] = 0.f;
            m_c[qi] = -INFINITY;
            l_part[qi] = 0.f;
        }
        unsigned ka2[KS], va2[DB];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) ka2[ks] = lds_base + KO + r * ROWB + k_swz<D>(r, 2 * ks + hh) * 16;
        {
            const int i16 = lane & 15, g16 = (lane >> 4) & 1;
            const int qq = i16 >> 2, pp = i16 & 3;
            const int row = 4 * hh + qq;
#pragma unroll
            for (int db = 0; db < DB; ++db)
                va2[db] = lds_base + VO + row * ROWB + v_swz<D>(row, 4 * db + 2 * g16 + (pp >> 1)) * 16 + 8 * (pp & 1);
        }
        auto dma2 = [&](int jj) {
            const unsigned st = (jj & 1) * TILE;
#pragma unroll
            for (int i = 0; i < CPT; ++i) {
                dma16(rk_w, __builtin_amdgcn_readfirstlane(piece_base + KO + st + i * PIECE), (unsigned)jj * k_tile_stride + g_koff[i]);
                dma16(rv_w, __builtin_amdgcn_readfirstlane(piece_base + VO + st + i * PIECE), (unsigned)jj * v_tile_stride + g_voff[i]);
            }
        };
        dma2(0);
        for (int jj = 0; jj < nt; ++jj) {
            dma_wait<0>();
            __syncthreads();                   // tile jj landed and is visible; tile jj-1's stage is free
            dma2(jj + 1);
            if (jj < my_nt) {
                const unsigned st = (jj & 1) * TILE;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
#pragma unroll
                    for (int qi = 0; qi < QB; ++qi) {
                        f32x16 sx;
#pragma unroll
                        for (int i = 0; i < 16; ++i) sx[i] = 0.f;
#pragma unroll
                        for (int ks = 0; ks < KS; ++ks)
                            sx = T::mfma(lds_read_b128(ka2[ks] + st + half * 32 * ROWB), qf[qi][ks], sx);
                        const int qrow = q0[qi] + r, key0 = jj * kBN + half * 32;
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const int key = key0 + (i & 3) + 8 * (i >> 2) + 4 * hh;
                            if ((key >= Sk) || (CAUSAL && key > qrow + coff)) sx[i] = -INFINITY;
                        }
                        float mx = fmaxf(sx[0], sx[1]);
#pragma unroll
                        for (int i = 2; i < 16; ++i) mx = fmaxf(mx, sx[i]);
                        auto sw = __builtin_amdgcn_permlane32_swap(bitcast<unsigned>(mx), bitcast<unsigned>(mx), false, false);
                        mx = fmaxf(bitcast<float>(sw[0]), bitcast<float>(sw[1]));
                        const float m_new = fmaxf(m_c[qi], mx * c);
                        const float alpha = (m_new == -INFINITY) ? 1.f : __builtin_amdgcn_exp2f(m_c[qi] - m_new);   // m_c = -inf -> 0
                        l_part[qi] *= alpha;
#pragma unroll
                        for (int db = 0; db < DB; ++db)
#pragma unroll
                            for (int i = 0; i < 16; ++i) o_acc[qi][db][i] *= alpha;
                        m_c[qi] = m_new;
                        const float m_sub = (m_new == -INFINITY) ? 0.f : m_new;       // row fully masked so far
                        u32x4 px[2];
#pragma unroll
                        for (int i = 0; i < 16; i += 2) {
                            const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sx[i], c, -m_sub));
                            const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sx[i + 1], c, -m_sub));
                            l_part[qi] += p0 + p1;
                            px[i >> 3][(i & 7) >> 1] = T::pack2(p0, p1);
                        }
#pragma unroll
                        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                            for (int db = 0; db < DB; ++db) {
                                const int sidx = 2 * half + s2;
                                u32x2 lo = lds_read_tr16_b64(va2[db] + st + (16 * sidx) * ROWB);
                                u32x2 hi = lds_read_tr16_b64(va2[db] + st + (16 * sidx + 8) * ROWB);
                                o_acc[qi][db] = T::mfma(u32x4{lo[0], lo[1], hi[0], hi[1]}, px[s2], o_acc[qi][db]);
                            }
                    }
                }
            }
        }
        dma_wait<0>();
        __syncthreads();                       // every wave is done with the fallback's LDS stages
    }

    // ---- the next query block of a causal pair: its first tiles and its Q fragments travel while this
    // block's output is normalised and stored (the rings are free: every wave passed the barrier above)
    if (pass + 1 < n_pass) {
        issue_prologue();
        load_q(tq);
    }

    // ---- epilogue: combine the two lane halves' row sums, normalise, store O (and LSE)
#pragma unroll
    for (int qi = 0; qi < QB; ++qi) {
        auto sw = __builtin_amdgcn_permlane32_swap(bitcast<unsigned>(l_part[qi]), bitcast<unsigned>(l_part[qi]), false, false);
        const float l = bitcast<float>(sw[0]) + bitcast<float>(sw[1]);
        const float inv = (l > 0.f) ? p.out_scale / l : 0.f;        // flash_attn_cutlass.cu:446-452 guard
        const int qrow = q0[qi] + r;
        if (p.lse != nullptr && hh == 0 && qrow < S) {
            // lse = m*scale + ln(l) = (m_c + log2(l)) * ln(2)
            p.lse[((long long)b * p.H + h) * S + qrow] = (l > 0.f) ? (m_c[qi] + __builtin_amdgcn_logf(l)) * 0.6931471805599453f : -INFINITY;
        }
        // lane (r,hh) holds O[qrow][db*32 + 8g + 4hh + 0..3] in o_acc[db][4g..4g+3].
        // Pair column groups (g, g+1) with permlane32_swap so that each lane stores 16 contiguous bytes:
        // lanes 0-31 -> cols 8g..8g+7, lanes 32-63 -> cols 8(g+1)..8(g+1)+7.
        elem_t* orow = oh + (long long)qrow * p.o_ss;
#pragma unroll
        for (int db = 0; db < DB; ++db) {
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                const f32x16& oa = o_acc[qi][db];
                unsigned a0 = T::pack2(oa[4 * g + 0] * inv, oa[4 * g + 1] * inv);
                unsigned a1 = T::pack2(oa[4 * g + 2] * inv, oa[4 * g + 3] * inv);
                unsigned b0 = T::pack2(oa[4 * g + 4] * inv, oa[4 * g + 5] * inv);
                unsigned b1 = T::pack2(oa[4 * g + 6] * inv, oa[4 * g + 7] * inv);
                auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                // after the swap: lanes 0-31: {own a, upper's a} ; lanes 32-63: {lower's b, own b}
                u32x4 outv = {s0[0], s1[0], s0[1], s1[1]};
                const int col = db * 32 + 8 * g + 8 * hh;
                if (qrow < S && col < p.dv) {
                    *reinterpret_cast<u32x4*>(orow + col) = outv;
                }
            }
        }
    }
  }  // pass
}

}  // namespace fa
