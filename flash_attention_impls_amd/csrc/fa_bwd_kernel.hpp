// FlashAttention backward for gfx950 (MI355X / CDNA4) -- device code.
//
// Replaces (behaviourally) the reference's recompute backward:
//   _bwd_kernel                  code/triton_fa2/FA2-triton.py:98-170
// with the softmax Jacobian in its correct form dS = P o (dP - delta) * scale, delta = rowsum(dO o O)
// (the reference's :160-161 is defective; its dV is reproduced, its dQ/dK are not).
//
// No atomics and no cross-workgroup sums: the gradient is computed by two launches of ONE kernel template,
//   MODE 0 (dQ)     : a workgroup owns 256 query rows (8 waves x 32) and streams the K / V tiles of its head;
//   MODE 1 (dK, dV) : a workgroup owns 128 keys (4 waves x 32, the 512-register file: two 32x128 f32
//                     accumulators per wave) and streams the Q / dO tiles of its head.
// Both recompute S and dP (7 matrix products for the 5 of the textbook count); in exchange dQ needs no f32
// atomics (their chip-wide rate, ~1.3 TB/s, would bound the kernel: MI355X_MICROARCH.md 'Global float
// atomics'), no f32 staging buffer and no conversion pass, and every output is bitwise reproducible.
//
// Common structure ("stationary" X rows live in registers as B operands, "streamed" Y tiles pass through LDS):
//   scores   T1[y][x] = Y1 . X1^T,  T2[y][x] = Y2 . X2^T      v_mfma_f32_16x16x32, A = Y rows (ds_read_b128)
//            MODE 0: Y1 = K, Y2 = V, X1 = Q, X2 = dO  (T1 = S^T, T2 = dP^T: key on the accumulator row)
//            MODE 1: Y1 = Q, Y2 = dO, X1 = K, X2 = V  (T1 = S,  T2 = dP : query on the accumulator row)
//   P  = exp2(T1 * scale*log2e - LSE*log2e)  (LSE from the forward; no running max, no rescale),
//   dS = P * T2', T2' = T2 - delta           (-delta from the pre-pass is the initial value of the T2 accumulator:
//                                             no subtraction in the loop; softmax scale applied in the epilogue)
//   the accumulators, packed to 16 bits in register order, ARE the B operands of the gradient products
//   (same trick as the forward's P): out^T[d][x] += Y^T[d][y] . W[y][x], A = Y^T through ds_read_b64_tr_b16:
//            MODE 0: dQ^T += K^T . dS^T          MODE 1: dV^T += dO^T . P,  dK^T += Q^T . dS
//   so every streamed tile is ONE LDS image read both by rows and transposed; the XOR swizzle below is
//   conflict-free for both read kinds (tools/lds_bank_sim.py).
//   Tiles arrive by LDS-DMA into a 3-stage (MODE 0) / 4-stage (MODE 1) ring (tile j+2 is issued right after the
//   barrier that publishes tile j); one barrier per 64-row tile.  MODE 0 runs two waves per SIMD and lets the
//   hardware interleave them; MODE 1 (one wave per SIMD) software-pipelines the 32-row blocks: scores(i) and
//   grads(i-2) are issued around the softmax of block i-1.  MODE 1 also streams the 64 rows' (LSE*log2e, delta) pairs.
//   Rows past the end of the sequence read as zeros (buffer bounds); the statistics planes are padded with
//   (+inf, 0) so that such rows give P = 0 without a mask.  Causal: fully masked 32-row blocks are skipped
//   per wave, the mask is applied only on diagonal blocks, workgroups are launched heaviest first.
#pragma once
#include "fa_fwd_kernel.hpp"

#ifndef FA_BWD_HALF_PRIO
#define FA_BWD_HALF_PRIO 0
#endif

namespace fa {

struct BwdParams {
    const void* x1; const void* x2;     // stationary operands: MODE 0: Q, dO ; MODE 1: K, V
    const void* y1; const void* y2;     // streamed operands:   MODE 0: K, V  ; MODE 1: Q, dO
    void* out1; void* out2;             // MODE 0: dQ, unused ; MODE 1: dK, dV
    const float* stats;                 // [2][B*H][Spad]: LSE*log2(e) (+inf past S), then -delta (0 past S)
    int B, H, S;                        // H, S: heads and rows of the stationary operands (= of the grid)
    int Sy;                             // rows of the streamed operands (MODE 0: keys, dK/dV kernel: queries)
    int coff;                           // keys - queries: the causal mask is bottom-right aligned, key <= query + coff
    int G;                              // grouped-query attention, query heads per key/value head (1 = equal head counts).
                                        // MODE 0: H query heads, the streamed K / V have H / G heads (head h reads h / G);
                                        // dK/dV kernel: H key/value heads, G query heads (h G .. h G + G-1) are streamed
                                        // one after the other into the same accumulators; stats cover B * H * G heads
    int dv;                             // valid head_dim (multiple of 16, <= the compiled D): columns past it read as zeros, are not stored
    int Spad;                           // S rounded up to a multiple of 64
    int bh;                             // B*H
    int nxb;                            // stationary blocks per head
    int unpaired;                       // MODE 0, causal: 1 = one query block per workgroup (small grids), 0 = block pairs
    int hsplit;                         // virtual heads per head of the grid mapping (wg_decode in fa_fwd_kernel.hpp)
    int xsplit;                         // dK/dV kernel: a key/value head's group of query heads is split over `xsplit` workgroups
                                        // (grid heads = key/value heads x xsplit, G = query heads per part); > 1: out1 / out2 are
                                        // fp32 partial sums [B][H][S][D] that fa_bwd_reduce_kernel adds up
    int qsplit;                         // dK/dV kernel: the visible query range of a key block is split over `qsplit` workgroups
                                        // (grid heads = key/value heads x xsplit x qsplit; partial sums whenever xsplit * qsplit > 1)
    long long x1_sb, x1_sh, x1_ss;      // element strides (head_dim stride is 1)
    long long x2_sb, x2_sh, x2_ss;
    long long y1_sb, y1_sh, y1_ss;
    long long y2_sb, y2_sh, y2_ss;
    long long o1_sb, o1_sh, o1_ss;
    long long o2_sb, o2_sh, o2_ss;
    float scale;                        // softmax scale
    float scale_log2;                   // scale * log2(e)
    // dS hand-off (fa_bwd_dq_gemm_kernel.hpp): workspace of 2 KiB units [query head][32-key slab][32-query block]
    void* ds;                           // null: recompute path
    long long ds_head_bytes;            // slabs (even count) x ds_row_bytes
    unsigned ds_row_bytes;              // blocks per slab (a multiple of 8) x 2048
};

template <int MODE> constexpr int bwd_waves() { return MODE == 0 ? 8 : 4; }
template <int MODE> constexpr int bwd_stages() { return MODE == 0 ? 3 : 4; }       // LDS ring depth (tiles)
template <int D, int MODE> constexpr int bwd_lds_bytes() { return 2 * bwd_stages<MODE>() * kBN * D * 2 + (MODE == 1 ? bwd_stages<MODE>() * 1024 : 0); }

// 16-byte-chunk swizzle of a streamed tile: serves the 16x16x32 row reads (16 lanes = 16 rows at one chunk)
// and the transposed reads (a 32-lane half = 8 consecutive rows x 32 bytes) without bank conflicts.
template <int D> __device__ __forceinline__ int bwd_swz(int row, int ch) {
    if constexpr (D == 128) return ch ^ ((row & 7) << 1);
    else return ch ^ (((row >> 1) & 3) << 1);
}

// pre-pass: -delta[row] = -sum_d dO[row][d] * O[row][d] (fp32), lse2[row] = LSE[row] * log2(e); padded rows get (+inf, 0).
// HBM-bound streaming kernel: D/8 lanes per row, 16-byte loads.
template <class T, int D>
__global__ __launch_bounds__(256) void fa_bwd_prep_kernel(const void* __restrict__ o, const void* __restrict__ d_o,
                                                          const float* __restrict__ lse, float* __restrict__ stats,
                                                          int H, int S, int Spad, int bh, int dv,
                                                          long long o_sb, long long o_sh, long long o_ss,
                                                          long long g_sb, long long g_sh, long long g_ss)
{
    constexpr int LPR = D / 8;                      // lanes per row
    constexpr int RPB = 256 / LPR;                  // rows per block
    const int nblk = (Spad + RPB - 1) / RPB;        // row blocks per head (flattened grid: no 65535 limit on B*H)
    const int head = blockIdx.x / nblk;
    const int b = head / H, h = head - b * H;
    const int row = (blockIdx.x - head * nblk) * RPB + threadIdx.x / LPR;
    const int sub = threadIdx.x % LPR;
    if (row >= Spad) return;
    float acc = 0.f;
    if (row < S && sub * 8 < dv) {
        const unsigned short* op = reinterpret_cast<const unsigned short*>(o) + b * o_sb + h * o_sh + (long long)row * o_ss + sub * 8;
        const unsigned short* gp = reinterpret_cast<const unsigned short*>(d_o) + b * g_sb + h * g_sh + (long long)row * g_ss + sub * 8;
        const u32x4 ov = *reinterpret_cast<const u32x4*>(op);
        const u32x4 gv = *reinterpret_cast<const u32x4*>(gp);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            float o0, o1, g0, g1;
            if constexpr (std::is_same<T, TypeBF16>::value) {
                o0 = bitcast<float>(ov[w] << 16); o1 = bitcast<float>(ov[w] & 0xffff0000u);
                g0 = bitcast<float>(gv[w] << 16); g1 = bitcast<float>(gv[w] & 0xffff0000u);
            } else {
                const f16x2 oh = bitcast<f16x2>(ov[w]), gh = bitcast<f16x2>(gv[w]);
                o0 = (float)oh[0]; o1 = (float)oh[1]; g0 = (float)gh[0]; g1 = (float)gh[1];
            }
            acc = __builtin_fmaf(o0, g0, acc);
            acc = __builtin_fmaf(o1, g1, acc);
        }
    }
#pragma unroll
    for (int m = LPR / 2; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
    if (sub == 0) {
        const long long idx = (long long)head * Spad + row;
        float l2 = INFINITY;
        if (row < S) {
            const float l = lse[(long long)head * S + row];
            l2 = (l == -INFINITY) ? INFINITY : l * 1.4426950408889634f;      // fully masked row: P = 0
        }
        stats[idx] = l2;
        stats[(long long)bh * Spad + idx] = -acc;        // stored negated: it seeds the dP accumulators
    }
}

// Sum of the `parts` fp32 partial dK (or dV) tensors of a key/value head, written once in the 16-bit output type:
// part[b][hk * parts + j][s][0..D) over j.  HBM-bound; one thread per 8 output elements (two 16-byte loads per part).
template <class T>
__global__ __launch_bounds__(256) void fa_bwd_reduce_kernel(const float* __restrict__ part, void* __restrict__ out,
                                                            int Hkv, int S, int dv, int parts, long long total8,
                                                            long long o_sb, long long o_sh, long long o_ss)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total8) return;
    const int c8 = dv / 8;
    const int col = (int)(t % c8) * 8;
    const long long row = t / c8;                    // (b * Hkv + hk) * S + s
    const int srow = (int)(row % S);
    const long long bh = row / S;
    const int hk = (int)(bh % Hkv);
    const long long b = bh / Hkv;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < parts; ++j) {
        const float* src = part + (((bh * parts + j) * S + srow) * (long long)dv + col);
        const f32x4 a = *reinterpret_cast<const f32x4*>(src), c = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[e] += a[e]; acc[4 + e] += c[e]; }
    }
    const u32x4 w = {T::pack2(acc[0], acc[1]), T::pack2(acc[2], acc[3]), T::pack2(acc[4], acc[5]), T::pack2(acc[6], acc[7])};
    unsigned short* dst = reinterpret_cast<unsigned short*>(out) + b * o_sb + hk * o_sh + (long long)srow * o_ss + col;
    *reinterpret_cast<u32x4*>(dst) = w;
}

// (MODE 0 keeps the 256-register budget at every head_dim.  Round 1 ran the head_dim-64 non-causal variant within 128 VGPRs --
// two workgroups per CU, +3 % -- but that build spilled 1-6 registers to scratch whatever was moved around; a scratch reload
// behind the hand-counted LDS-DMA waits drains the staging ring, and no default-path kernel may have a private segment.)
template <class T, int D, int MODE, bool CAUSAL>
__global__ __launch_bounds__(64 * bwd_waves<MODE>(), MODE == 0 ? 2 : 1) void fa_bwd_kernel(const BwdParams p)
{
    constexpr int NW = bwd_waves<MODE>();
    constexpr int XB = NW * 32;                // stationary rows per workgroup
    constexpr int KS = D / 32;                 // k-steps of the score products
    constexpr int DT = D / 16;                 // 16-wide head_dim tiles of the gradient accumulators
    constexpr int ROWB = D * 2;                // bytes per row of an LDS tile
    constexpr int TILE = kBN * ROWB;           // bytes per streamed tile
    constexpr int PIECE = 1024;                // bytes one DMA wave-instruction moves
    constexpr int CPT = TILE / PIECE / NW;     // DMA pieces per wave per tile
    constexpr bool PIPE = MODE == 1;           // one wave per SIMD: software-pipelined block loop
    constexpr int NS = bwd_stages<MODE>();
    constexpr int NBUF = PIPE ? 2 : 1;
    constexpr int Y2BASE = NS * TILE;
    constexpr int STBASE = 2 * NS * TILE;      // MODE 1: NS x 1 KiB of row statistics
    static_assert(CPT >= 1, "tile too small for the workgroup");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_char*)smem;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // workgroup -> (head, stationary block); blockIdx % 8 = XCD: all blocks of a head share one L2
    // MODE 0 under the causal mask: a workgroup takes the query-block pair (nxb-1-t, t), so that every workgroup
    // streams nxb+1 blocks' worth of tiles (a balanced grid, as in the forward); MODE 1 launches the heaviest key
    // blocks (the first ones) first
    constexpr bool PAIRABLE = MODE == 0 && CAUSAL;
    const bool PAIR = PAIRABLE && !p.unpaired;
    const int wg_per_head = PAIR ? (p.nxb + 1) / 2 : p.nxb;
    int head, t;
    if (!wg_decode(blockIdx.x, p.bh, wg_per_head, p.hsplit, head, t)) return;
    const int b = head / p.H;
    const int h = head - b * p.H;
    const int S = p.S;
    const int Sy = p.Sy;
    const int coff = (CAUSAL && MODE == 0) ? p.coff : 0;      // (MODE 1 is launched for equal lengths only)
    const int n_pass = (PAIR && p.nxb - 1 - t != t) ? 2 : 1;
  for (int pass = 0; pass < n_pass; ++pass) {
    const int xb = PAIRABLE ? (pass == 0 ? p.nxb - 1 - t : t) : t;      // (unpaired: one pass, longest blocks first)
    const int x0 = xb * XB;                    // first stationary row of the workgroup
    const int lane = lane_here();              // per-pass opaque lane id: nothing derived from it is hoisted out of the pass loop
    const int li = lane & 15;
    const int lg = lane >> 4;
    // MODE 0 under the causal mask: waves w and w+4 share a SIMD; row blocks that sum to 7 give every SIMD the same
    // number of blocks on the workgroup's diagonal tiles
    const int rowblk_of_wave = (MODE == 0 && CAUSAL && wave >= 4) ? 11 - wave : wave;
    const int x0w = x0 + rowblk_of_wave * 32;  // ... of this wave

    using elem_t = unsigned short;
    const elem_t* x1h = reinterpret_cast<const elem_t*>(p.x1) + b * p.x1_sb + h * p.x1_sh;
    const elem_t* x2h = reinterpret_cast<const elem_t*>(p.x2) + b * p.x2_sb + h * p.x2_sh;
    // MODE 0: the key/value head of a query head (grouped-query attention); MODE 1 (the single-wave dK/dV variant) is
    // launched for equal head counts only
    const int hy = MODE == 0 ? h / p.G : h;
    const elem_t* y1h = reinterpret_cast<const elem_t*>(p.y1) + b * p.y1_sb + hy * p.y1_sh;
    const elem_t* y2h = reinterpret_cast<const elem_t*>(p.y2) + b * p.y2_sb + hy * p.y2_sh;

    // ---- streamed range of the workgroup (64-row tiles) and of this wave (32-row blocks)
    const int nty = (Sy + kBN - 1) / kBN;
    int j_begin = 0, j_end = nty;
    int blk_begin_w = 0, blk_end_w = (Sy + 31) / 32;     // blocks this wave computes on
    int blk_mask_lo = 0x7fffffff, blk_mask_hi = -1;      // blocks in [lo, hi] need the causal mask
    if constexpr (CAUSAL) {
        if constexpr (MODE == 0) {                        // queries stationary, keys streamed: keys <= query
            j_end = min(nty, (max(0, min(Sy, min(S, x0 + XB) + coff)) + kBN - 1) / kBN);
            blk_end_w = (x0w >= S) ? 0 : (max(0, min(Sy, min(S, x0w + 32) + coff)) + 31) / 32;
            blk_mask_lo = max(0, x0w + coff) >> 5;        // first block containing a key > the wave's first query (+ coff)
            blk_mask_hi = 0x7fffffff;
        } else {                                          // keys stationary, queries streamed: queries >= key
            j_begin = min(nty, x0 / kBN);
            blk_begin_w = x0w >> 5;                       // blocks before it hold only queries < the wave's keys
            blk_mask_lo = blk_mask_hi = x0w >> 5;         // the diagonal block
        }
    }
    if (x0w >= S) { blk_begin_w = 0; blk_end_w = 0; }     // no stationary rows: staging duty only

    // ---- stationary fragments: lane (li, lg) holds X[x0w + 16 xt + li][32 ks + 8 lg .. +7]
    u32x4 xf1[2][KS], xf2[2][KS];
    {
        const unsigned x1_bytes = (unsigned)(((long long)(S - 1) * p.x1_ss + p.dv) * 2);
        const unsigned x2_bytes = (unsigned)(((long long)(S - 1) * p.x2_ss + p.dv) * 2);
        __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<elem_t*>(x1h), 0, x1_bytes, 0x00020000);
        __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<elem_t*>(x2h), 0, x2_bytes, 0x00020000);
#pragma unroll
        for (int xt = 0; xt < 2; ++xt) {
            const int xrow = x0w + 16 * xt + li;
            // rows past the end of the sequence get an offset outside the descriptor: they read as zero
            const unsigned o1 = (xrow < S) ? (unsigned)((long long)xrow * p.x1_ss * 2 + lg * 16) : 0x80000000u;
            const unsigned o2 = (xrow < S) ? (unsigned)((long long)xrow * p.x2_ss * 2 + lg * 16) : 0x80000000u;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bool live = 32 * ks + 8 * lg < p.dv;            // columns past the valid head_dim read as zero
                xf1[xt][ks] = __builtin_amdgcn_raw_buffer_load_b128(r1, live ? o1 + ks * 64 : 0x80000000u, 0, 0);
                xf2[xt][ks] = __builtin_amdgcn_raw_buffer_load_b128(r2, live ? o2 + ks * 64 : 0x80000000u, 0, 0);
            }
        }
    }
    // MODE 0: the statistics belong to the stationary rows (one pair per lane and x tile)
    float lse_x[2] = {INFINITY, INFINITY}, ndelta_x[2] = {0.f, 0.f};
    f32x4 nd4_x[2];                                 // -delta of the lane's stationary row, four times: initial T2 accumulator
    if constexpr (MODE == 0) {
#pragma unroll
        for (int xt = 0; xt < 2; ++xt) {
            const int xrow = x0w + 16 * xt + li;
            if (xrow < p.Spad) {
                lse_x[xt] = p.stats[(long long)head * p.Spad + xrow];
                ndelta_x[xt] = p.stats[(long long)(p.bh + head) * p.Spad + xrow];
            }
            nd4_x[xt] = f32x4{ndelta_x[xt], ndelta_x[xt], ndelta_x[xt], ndelta_x[xt]};
        }
    }

    // consume every register load HERE, before any LDS-DMA of this pass is in flight: hipcc does not see the asm DMAs,
    // and the vmcnt waits it would otherwise place at first uses inside the tile loop would drain them
#pragma unroll
    for (int xt = 0; xt < 2; ++xt) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if constexpr (PIPE) {
                // MODE 1 reads them as accumulator-file operands of the asm score MFMAs: pin them there (a copy made
                // right in front of an asm MFMA would not get its v_accvgpr_write -> MFMA wait states)
                asm volatile("" : "+a"(xf1[xt][ks]));
                asm volatile("" : "+a"(xf2[xt][ks]));
            } else {
                asm volatile("" : "+v"(xf1[xt][ks]));
                asm volatile("" : "+v"(xf2[xt][ks]));
            }
        }
        if constexpr (MODE == 0) {
            asm volatile("" : "+v"(lse_x[xt]));
            asm volatile("" : "+v"(nd4_x[xt]));
        }
    }

    // ---- staging by LDS-DMA: piece (wave * CPT + i) of a tile, swizzle applied on the source address
    const unsigned y1_bytes = (unsigned)(((long long)(Sy - 1) * p.y1_ss + p.dv) * 2);
    const unsigned y2_bytes = (unsigned)(((long long)(Sy - 1) * p.y2_ss + p.dv) * 2);
    const u32x4 ry1 = make_rsrc(y1h, y1_bytes);
    const u32x4 ry2 = make_rsrc(y2h, y2_bytes);
    unsigned g_y1[CPT], g_y2[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int byte = (wave * CPT + i) * PIECE + lane * 16;
        const int row = byte / ROWB, chp = (byte % ROWB) / 16;
        const bool live = bwd_swz<D>(row, chp) * 8 < p.dv;       // chunks past the valid head_dim: out of the descriptor -> zeros
        g_y1[i] = live ? (unsigned)(row * p.y1_ss * 2 + bwd_swz<D>(row, chp) * 16) : 0x80000000u;
        g_y2[i] = live ? (unsigned)(row * p.y2_ss * 2 + bwd_swz<D>(row, chp) * 16) : 0x80000000u;
    }
    const unsigned y1_tile_stride = (unsigned)(kBN * p.y1_ss * 2);
    const unsigned y2_tile_stride = (unsigned)(kBN * p.y2_ss * 2);
    const unsigned piece_base = lds_base + wave * CPT * PIECE;          // wave-uniform
    // statistics of a 64-row tile (MODE 1): lanes 0-15 fetch LSE*log2e, lanes 16-31 delta, 16 bytes each
    const u32x4 rst = make_rsrc(p.stats, (unsigned)((long long)2 * p.bh * p.Spad * 4));
    unsigned g_st = 0x80000000u;
    if constexpr (MODE == 1) {
        if (lane < 32) g_st = (unsigned)((((long long)(lane >> 4) * p.bh + head) * p.Spad + (lane & 15) * 4) * 4);
    }
    auto issue_tile = [&](int j, int stage) {
        if constexpr (MODE == 1) {
            // every wave fetches the same 1 KiB (identical bytes to one place): no branch in the tile loop, so a
            // whole unrolled trip stays one basic block for the scheduler.  Past the last tile the offset leaves the
            // descriptor only for the delta plane's tail; the LSE lanes may then read the next head's rows --
            // harmless, such tiles are never computed on.
            dma16(rst, __builtin_amdgcn_readfirstlane(lds_base + STBASE + stage * 1024), g_st + (unsigned)j * (kBN * 4));
        }
#pragma unroll
        for (int i = 0; i < CPT; ++i)
            dma16(ry1, __builtin_amdgcn_readfirstlane(piece_base + stage * TILE + i * PIECE), (unsigned)j * y1_tile_stride + g_y1[i]);
#pragma unroll
        for (int i = 0; i < CPT; ++i)
            dma16(ry2, __builtin_amdgcn_readfirstlane(piece_base + Y2BASE + stage * TILE + i * PIECE), (unsigned)j * y2_tile_stride + g_y2[i]);
    };

    // ---- LDS read addresses (stage 0, block 0)
    // (ring stage, block and tile offsets are added as instruction immediates: < 64 KiB per ring)
    unsigned ra[KS], ra2[KS];  // row reads: lane (li, lg) reads row li (+16 yt + 32 blk), chunk 4 ks + lg
    unsigned sta;              // statistics of rows 4 lg .. +3 (+16 yt + 32 blk)
    unsigned ta[DT], ta2[MODE == 1 ? DT : 1];   // transposed reads: lane 4 qq + pp of a 16-lane group supplies row 4 lg + qq, columns 16 dt + 4 pp .. +3
    // (filled in behind the prologue's DMA issue, from fresh lane coordinates: the address arithmetic then starts where the
    // stationary fragments' load registers are free again, instead of being carried -- spilled -- across them)
    auto setup_addresses = [&]() {
        const int lane_a = lane_here();
        const int li_a = lane_a & 15, lg_a = lane_a >> 4;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            ra[ks] = lds_base + li_a * ROWB + bwd_swz<D>(li_a, 4 * ks + lg_a) * 16;
            ra2[ks] = ra[ks] + Y2BASE;
            asm volatile("" : "+v"(ra2[ks]));                   // opaque: keeps it a register of its own (the sum would not fit an immediate)
        }
        sta = lds_base + STBASE + lg_a * 16;
        asm volatile("" : "+v"(sta));
        const int qq = li_a >> 2, pp = li_a & 3;
        const int row = 4 * lg_a + qq;                  // + 16 a + 32 blk: multiples of 16 rows, swizzle-neutral
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            ta[dt] = lds_base + row * ROWB + bwd_swz<D>(row, 2 * dt + (pp >> 1)) * 16 + 8 * (pp & 1);
            if constexpr (MODE == 1) {
                ta2[dt] = ta[dt] + Y2BASE;
                asm volatile("" : "+v"(ta2[dt]));
            }
        }
    };

    f32x4 acc1[DT][2];         // MODE 0: dQ^T ; MODE 1: dK^T   [head_dim tile][x tile]
    f32x4 acc2[MODE == 1 ? DT : 1][2];   // MODE 1: dV^T
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int xt = 0; xt < 2; ++xt) {
            acc1[dt][xt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (MODE == 1) acc2[dt][xt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    const float c = p.scale_log2;

    // ---- a 32-row block passes through three steps; PAR selects one of two register sets (software pipeline)
    f32x4 t1[NBUF][2][2], t2[NBUF][2][2];          // score accumulators [set][y tile][x tile]
    u32x4 pw[NBUF][2], dsw[NBUF][2];               // packed P and dS: B operands, element 4 yt + e <-> y = 16 yt + 4 lg + e
#if defined(FA_BWD_ABL_NOEW) || defined(FA_BWD_ABL_NOSC)
    for (int i = 0; i < NBUF; ++i)
        for (int xt = 0; xt < 2; ++xt) {
            pw[i][xt] = dsw[i][xt] = u32x4{(unsigned)lane, 1u, 2u, 3u};
            for (int yt = 0; yt < 2; ++yt) t1[i][yt][xt] = t2[i][yt][xt] = f32x4{0.f, 1.f, 2.f, 3.f};
        }
#endif

    // SC: T1 = Y1 . X1^T, T2 = Y2 . X2^T for block `blk` of the tile in ring stage `stage`
    // (`stage` and `blk` are ints or integral constants: constants turn every LDS offset into an immediate)
    auto step_scores = [&] __device__ (auto par_c, auto stage, auto blk) {
        constexpr int PAR = decltype(par_c)::value;
#if defined(FA_BWD_ABL_NOSC)      // timing-only build
        return;
#endif
        const unsigned so = stage * TILE + blk * 32 * ROWB;
#pragma unroll
        for (int yt = 0; yt < 2; ++yt) {
            // T2 starts at -delta: of the stationary row (MODE 0, a lane constant) or of the streamed rows 16 yt + 4 lg + 0..3
            f32x4 nd_y;
            if constexpr (MODE == 1) nd_y = bitcast<f32x4>(lds_read_b128(sta + (stage * 1024 + (blk * 32 + 16 * yt) * 4 + 256)));
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const u32x4 a1 = lds_read_b128(ra[ks] + (so + yt * 16 * ROWB));
                const u32x4 a2 = lds_read_b128(ra2[ks] + (so + yt * 16 * ROWB));
#pragma unroll
                for (int xt = 0; xt < 2; ++xt) {
                    if constexpr (PIPE) {
                        // 512-register kernel: scores go to arch VGPRs (see mfma16_v_first); the callers keep VALU
                        // readers of t1/t2 away from these MFMAs
                        if (ks == 0) {
                            T::mfma16_v_first(t1[PAR][yt][xt], a1, xf1[xt][ks]);
                            T::mfma16_v_init(t2[PAR][yt][xt], a2, xf2[xt][ks], nd_y);
                        } else {
                            T::mfma16_v_acc(t1[PAR][yt][xt], a1, xf1[xt][ks]);
                            T::mfma16_v_acc(t2[PAR][yt][xt], a2, xf2[xt][ks]);
                        }
                    } else {
                        // first k-step: C = 0 / the resident -delta vector (a different register than the destination)
                        t1[PAR][yt][xt] = T::mfma16(a1, xf1[xt][ks], ks == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : t1[PAR][yt][xt]);
                        t2[PAR][yt][xt] = T::mfma16(a2, xf2[xt][ks], ks == 0 ? nd4_x[xt] : t2[PAR][yt][xt]);
                    }
                }
            }
        }
    };
    // EW: P = exp2(T1 c - LSE2) (masked), dS = P (T2 - delta), packed to 16 bits
    auto step_softmax = [&] __device__ (auto mask_c, auto par_c, auto stage, auto blk, int y0) {
        constexpr bool MASK = decltype(mask_c)::value;
        constexpr int PAR = decltype(par_c)::value;
#if defined(FA_BWD_ABL_NOEW)      // timing-only build
        return;
#endif
        // row statistics of the streamed rows (MODE 1): rows 16 yt + 4 lg + 0..3 of the block
        f32x4 lse_y[2];
        if constexpr (MODE == 1) {
#pragma unroll
            for (int yt = 0; yt < 2; ++yt)
                lse_y[yt] = bitcast<f32x4>(lds_read_b128(sta + (stage * 1024 + (blk * 32 + 16 * yt) * 4)));
        }
#pragma unroll
        for (int xt = 0; xt < 2; ++xt) {
            const int xrow = x0w + 16 * xt + li;
#pragma unroll
            for (int yt = 0; yt < 2; ++yt) {
                float pv[4], dv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float l2 = (MODE == 0) ? lse_x[xt] : lse_y[yt][e];
                    float pe = __builtin_amdgcn_exp2f(__builtin_fmaf(t1[PAR][yt][xt][e], c, -l2));
                    if constexpr (MASK) {
                        const int yrow = y0 + 16 * yt + 4 * lg + e;
                        const bool dead = (MODE == 0) ? (yrow > xrow + coff) : (xrow > yrow);     // key > query (+ coff)
                        if (dead) pe = 0.f;
                    }
                    pv[e] = pe;
                    dv[e] = pe * t2[PAR][yt][xt][e];
                }
                if constexpr (MODE == 1) {
                    pw[PAR][xt][2 * yt] = T::pack2(pv[0], pv[1]);
                    pw[PAR][xt][2 * yt + 1] = T::pack2(pv[2], pv[3]);
                }
                dsw[PAR][xt][2 * yt] = T::pack2(dv[0], dv[1]);
                dsw[PAR][xt][2 * yt + 1] = T::pack2(dv[2], dv[3]);
            }
        }
    };
    // GR: gradient products, A = Y^T fragments through the transposed read (rows 4 lg .. +3 and 16 + 4 lg .. +3 of the block)
    auto grad_frag = [&] __device__ (unsigned (&base)[DT], int dt, unsigned so) {
        const u32x2 lo = lds_read_tr16_b64(base[dt] + so);
        const u32x2 hi = lds_read_tr16_b64(base[dt] + (so + 16 * ROWB));
        return u32x4{lo[0], lo[1], hi[0], hi[1]};
    };
    auto step_grads = [&] __device__ (auto par_c, auto stage, auto blk, auto dt_lo, auto dt_hi) {
        constexpr int PAR = decltype(par_c)::value;
#if defined(FA_BWD_ABL_NOGR)      // timing-only build
        return;
#endif
        const unsigned so = stage * TILE + blk * 32 * ROWB;
#pragma unroll
        for (int dt = decltype(dt_lo)::value; dt < decltype(dt_hi)::value; ++dt) {
            const u32x4 a1 = grad_frag(ta, dt, so);
#pragma unroll
            for (int xt = 0; xt < 2; ++xt) acc1[dt][xt] = T::mfma16(a1, dsw[PAR][xt], acc1[dt][xt]);
            if constexpr (MODE == 1) {
                const u32x4 a2 = grad_frag(ta2, dt, so);
#pragma unroll
                for (int xt = 0; xt < 2; ++xt) acc2[dt][xt] = T::mfma16(a2, pw[PAR][xt], acc2[dt][xt]);
            }
        }
    };
    auto needs_mask = [&](int bi) { return CAUSAL && bi >= blk_mask_lo && bi <= blk_mask_hi; };

    // ---- main loop: tile j lives in ring stage (j - j_begin) % NS; tile j + 2 is issued behind barrier j
    constexpr int OPS = 2 * CPT + (MODE == 1 ? 1 : 0);   // DMA instructions per tile and wave
    if (j_begin < j_end) {
        // MODE 1: the ring stage of the first tile is chosen so that the first tile whose blocks are all in the
        // steady state (for the workgroup's last wave; earlier waves reach it up to one tile sooner) sits in stage 0,
        // where the four-tile trips with immediate offsets start
        int stage = 0;
        if constexpr (PIPE) {
            const int bb_last = CAUSAL ? ((x0 + (NW - 1) * 32) >> 5) : 0;
            const int j_fast = (bb_last + 3) >> 1;
            stage = (NS - ((j_fast - j_begin) & (NS - 1))) & (NS - 1);
        }
        issue_tile(j_begin, stage);
        issue_tile(j_begin + 1, (stage + 1) % NS);
        setup_addresses();
        if constexpr (!PIPE) {
            // two waves per SIMD: the hardware interleaves one wave's matrix steps with its partner's softmax
#if FA_BWD_HALF_PRIO
            const int late_half = (NW == 8 && wave >= 4) ? 1 : 0;
#endif
            auto sync_and_issue0 = [&](int j, int dst_stage) {
                dma_wait<OPS>();                      // this wave's pieces of tile j have landed ...
                __syncthreads();                      // ... every wave's are visible, and tile j-1 is no longer read
                issue_tile(j + 2, dst_stage);
            };
            auto whole_block = [&] __device__ (auto mask_c, auto st, auto blk, int y0) {
                step_scores(IC<0>{}, st, blk);
                step_softmax(mask_c, IC<0>{}, st, blk, y0);
                step_grads(IC<0>{}, st, blk, IC<0>{}, IC<DT>{});
            };
            // one loop form only (a second, specialised loop made the register allocator shuffle and spill the
            // accumulators between the two): a trip = one turn of the ring, so ring stages and block offsets are
            // immediates; each block is guarded (in this wave's range? on the diagonal?) by scalar branches
            auto tile = [&] __device__ (auto st_c, int j) {
                constexpr int ST = decltype(st_c)::value;
                sync_and_issue0(j, (ST + 2) % NS);
                const int b0 = 2 * j, b1 = 2 * j + 1;
#if FA_BWD_HALF_PRIO
                // behind the barrier the two waves of a SIMD start the same block together and the older one wins every
                // arbitration, then waits at the next barrier: priority for the younger half in the first block of a tile
                // evens the two out (forward: fa_fwd_kernel8.hpp, profiles/r3_ab_runs.txt)
                setprio_if<1>(late_half);
#endif
                if (b0 >= blk_begin_w && b0 < blk_end_w) {
                    if (needs_mask(b0)) whole_block(std::true_type{}, IC<ST>{}, IC<0>{}, b0 * 32);
                    else whole_block(std::false_type{}, IC<ST>{}, IC<0>{}, b0 * 32);
                }
#if FA_BWD_HALF_PRIO
                setprio_if<0>(late_half);
#endif
                if (b1 >= blk_begin_w && b1 < blk_end_w) {
                    if (needs_mask(b1)) whole_block(std::true_type{}, IC<ST>{}, IC<1>{}, b1 * 32);
                    else whole_block(std::false_type{}, IC<ST>{}, IC<1>{}, b1 * 32);
                }
            };
            for (int j = j_begin; j < j_end; j += NS) {
                tile(IC<0>{}, j);
                if (j + 1 < j_end) tile(IC<1>{}, j + 1);
                if (j + 2 < j_end) tile(IC<2>{}, j + 2);
            }
            (void)stage;
        } else {
            // one wave per SIMD: software pipeline over blocks -- iteration i runs the matrix products of scores(i)
            // and grads(i-2) around the softmax of block i-1, all three independent of each other.  Block i reads
            // tile i/2 (rows), block i-2 tile i/2 - 1 (transposed): barrier j therefore retires tile j-2, whose ring
            // stage (of 4) receives tile j+2.  One extra trip drains the pipeline.
            auto iteration = [&] __device__ (auto par_c, int i, auto st, auto sp) {
                constexpr int PAR = decltype(par_c)::value;              // = i & 1 = block of the tile
                const bool do_sc = i >= blk_begin_w && i < blk_end_w;
                const bool do_ew = i - 1 >= blk_begin_w && i - 1 < blk_end_w;
                const bool do_gr = i - 2 >= blk_begin_w && i - 2 < blk_end_w;
                // scores(i): tile i/2 = stage st; softmax(i-1): tile (i-1)/2 = st (odd i) or sp (even i); grads(i-2): sp
                if (do_gr) step_grads(IC<PAR>{}, sp, IC<PAR>{}, IC<0>{}, IC<DT>{});
                if (do_ew) {
                    if constexpr (PAR == 1) {
                        if (needs_mask(i - 1)) step_softmax(std::true_type{}, IC<0>{}, st, IC<0>{}, (i - 1) * 32);
                        else step_softmax(std::false_type{}, IC<0>{}, st, IC<0>{}, (i - 1) * 32);
                    } else {
                        if (needs_mask(i - 1)) step_softmax(std::true_type{}, IC<1>{}, sp, IC<1>{}, (i - 1) * 32);
                        else step_softmax(std::false_type{}, IC<1>{}, sp, IC<1>{}, (i - 1) * 32);
                    }
                }
                if (do_sc) {
                    step_scores(IC<PAR>{}, st, IC<PAR>{});
                    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");      // MFMA -> VALU wait states of the asm score MFMAs
                }
            };
            // steady state (all three steps live, no mask): one straight-line region per block, ring stage known
            // at compile time, the scheduler interleaves the three independent steps
            auto fused = [&] __device__ (auto par_c, auto st_c) {
                constexpr int PAR = decltype(par_c)::value, ST = decltype(st_c)::value;
                typedef IC<ST> SC_ST;
                typedef IC<(ST + NS - 1) % NS> GR_ST;
                typedef IC<(PAR == 1 ? ST : (ST + NS - 1) % NS)> EW_ST;
                __builtin_amdgcn_sched_barrier(0);
                step_scores(IC<PAR>{}, SC_ST{}, IC<PAR>{});
                step_softmax(std::false_type{}, IC<PAR ^ 1>{}, EW_ST{}, IC<PAR ^ 1>{}, 0);
                step_grads(IC<PAR>{}, GR_ST{}, IC<PAR>{}, IC<0>{}, IC<DT - 1>{});
                // the region ends with four compiler-visible MFMAs (64 cycles): the next region's softmax reads the
                // asm score MFMAs' results no earlier than that (their MFMA -> VALU wait states)
                const unsigned so = GR_ST::value * TILE + PAR * 32 * ROWB;
                const u32x4 l1 = grad_frag(ta, DT - 1, so);
                const u32x4 l2 = grad_frag(ta2, DT - 1, so);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int xt = 0; xt < 2; ++xt) {
                    acc1[DT - 1][xt] = T::mfma16(l1, dsw[PAR][xt], acc1[DT - 1][xt]);
                    acc2[DT - 1][xt] = T::mfma16(l2, pw[PAR][xt], acc2[DT - 1][xt]);
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            auto sync_and_issue = [&](int j, int dst_stage) {
                dma_wait<OPS>();
#if !defined(FA_BWD_ABL_NOBAR)    // timing-only build without the barrier
                __syncthreads();
#endif
#if !defined(FA_BWD_ABL_NODMA)    // timing-only build without the staging DMA
                issue_tile(j + 2, dst_stage);
#endif
            };
            int j = j_begin;
            while (j <= j_end) {
                // four tiles whose eight blocks are all in the steady state, starting at ring stage 0
                if (stage == 0 && j + 3 < j_end && 2 * j - 2 >= blk_begin_w && 2 * j + 7 < blk_end_w &&
                    !(CAUSAL && 2 * j - 1 <= blk_mask_hi && 2 * j + 6 >= blk_mask_lo)) {
                    sync_and_issue(j, 2);
                    fused(IC<0>{}, IC<0>{});
                    fused(IC<1>{}, IC<0>{});
                    sync_and_issue(j + 1, 3);
                    fused(IC<0>{}, IC<1>{});
                    fused(IC<1>{}, IC<1>{});
                    sync_and_issue(j + 2, 0);
                    fused(IC<0>{}, IC<2>{});
                    fused(IC<1>{}, IC<2>{});
                    sync_and_issue(j + 3, 1);
                    fused(IC<0>{}, IC<3>{});
                    fused(IC<1>{}, IC<3>{});
                    j += 4;
                    continue;
                }
                if (j < j_end) sync_and_issue(j, (stage + 2) % NS);
                const int sp = (stage + NS - 1) % NS;
                iteration(IC<0>{}, 2 * j, stage, sp);
                iteration(IC<1>{}, 2 * j + 1, stage, sp);
                stage = (stage + 1) % NS;
                ++j;
            }
        }
        dma_wait<0>();                                // no DMA may still be writing LDS when the workgroup retires
    }

    // ---- epilogue: out[x][16 dt + 4 lg + 0..3] = acc[dt][xt] * mult, 16-byte stores through v_permlane16_swap pairs
    auto store_out = [&] __device__ (f32x4 (&acc)[DT][2], void* out, long long sb, long long sh, long long ss, float mult) {
        elem_t* oh = reinterpret_cast<elem_t*>(out) + b * sb + h * sh;
        const int lane_e = lane_here();               // fresh lane coordinates (not kept alive across the main loop)
        const int li = lane_e & 15, lg = lane_e >> 4;
#pragma unroll
        for (int xt = 0; xt < 2; ++xt) {
            const int xrow = x0w + 16 * xt + li;
            elem_t* orow = oh + (long long)xrow * ss;
#pragma unroll
            for (int dt = 0; dt < DT; dt += 2) {
                const f32x4 oa = acc[dt][xt], ob = acc[dt + 1][xt];
                unsigned a0 = T::pack2(oa[0] * mult, oa[1] * mult), a1 = T::pack2(oa[2] * mult, oa[3] * mult);
                unsigned b0 = T::pack2(ob[0] * mult, ob[1] * mult), b1 = T::pack2(ob[2] * mult, ob[3] * mult);
                auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
                auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
                u32x4 outv = {s0[0], s1[0], s0[1], s1[1]};
                const int col = (lg & 1) ? (16 * (dt + 1) + 4 * (lg - 1)) : (16 * dt + 4 * lg);
                if (xrow < S && col < p.dv) *reinterpret_cast<u32x4*>(orow + col) = outv;
            }
        }
    };
    store_out(acc1, p.out1, p.o1_sb, p.o1_sh, p.o1_ss, p.scale);
    if constexpr (MODE == 1) store_out(acc2, p.out2, p.o2_sb, p.o2_sh, p.o2_ss, 1.0f);
    if (pass + 1 < n_pass) __syncthreads();           // every wave is done with the ring before the next pass refills it
  }  // pass
}

}  // namespace fa
