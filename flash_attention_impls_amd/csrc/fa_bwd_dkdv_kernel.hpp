// FlashAttention backward for gfx950 -- dK / dV kernel with wave-specialised workgroups.
//
// Same products, fragment maps, LDS image and causal rules as MODE 1 of fa_bwd_kernel.hpp (read its header first):
// a workgroup owns 128 keys and streams the Q / dO tiles of its head.  The difference is who does what.  There, one
// wave per SIMD runs everything for its 32 keys (scores, softmax, both gradient products: 512 registers) and its
// instruction stream is effectively serial.  Here the work of a 32-key slice is split between TWO waves that share a
// SIMD, both within 256 registers, and what one of them issues is what the other one does not:
//
//   wave w   (0..3, "score wave")   : S = Q K^T, dP' = dO V^T - delta (K and V fragments in registers),
//                                     P = exp2(S c - LSE2) [masked], dS = P o dP'; publishes P and dS (16-bit, packed
//                                     in register order) in an LDS mailbox.  MFMA + all of the VALU work, row reads.
//   wave w+4 (4..7, "gradient wave"): dV^T += dO^T P, dK^T += Q^T dS (both accumulators in registers).
//                                     MFMA + transposed LDS reads, no VALU work at all.
//
// Nothing is recomputed.  Both roles keep the 16x16x32 fragment maps of fa_bwd_kernel.hpp, so the mailbox is a
// lane-to-lane hand-over: lane l of the score wave writes the sixteen packed words lane l of the gradient wave needs
// (four conflict-free 16-byte LDS writes / reads per 32-row block).  The gradient wave runs one 32-row block behind;
// one barrier per block publishes the mailbox (double-buffered by block parity) and, every second block, a staged
// tile.  The score wave is software-pipelined on its own: the MFMAs of scores(i+1) are issued beside the softmax of
// block i.  Ring: 3 stages of 64-row tiles, tile j+2 requested in step 2j+1 into the stage tile j-1 has just left.
#pragma once
#include "fa_bwd_kernel.hpp"

#ifndef FA_BWD_DS_STORE_AUX
#define FA_BWD_DS_STORE_AUX 2          // cache policy bits of the dS stores: 2 = nt (written once, read by the next kernel: +2.6 % over 0)
#endif

namespace fa {

#if defined(FA_BWD_STAMP)
// Diagnostic build only (-DFA_BWD_STAMP; never shipped, never timed; tools/stamps_bwd.py reads it): per-wave cycle sums by
// s_memtime stamps, added up over all workgroups of a launch: [wave][0] whole kernel, [1] DMA wait of a step, [2] the step's
// barrier, [3] staging issue behind it, [4] the step's work (scores + softmax, or the gradient products), [5] steps,
// [6] prologue of a head pass (first tiles, their wait, the barrier).  The stamps drain the LDS reads the real kernel keeps in
// flight across steps: read the SHARES, not the lengths.
__device__ unsigned long long g_fa_bwd_stamp[8][8];
__device__ __forceinline__ unsigned long long bwd_stamp_now() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define FA_BWD_ST(...) __VA_ARGS__
#else
#define FA_BWD_ST(...)
#endif

constexpr int kDkdvStages = 3;
template <int D> constexpr int dkdv_lds_bytes() { return 2 * kDkdvStages * kBN * D * 2 + kDkdvStages * 1024 + 2 * 4 * 4096; }

// WDS: the score waves also write each block's dS to the hand-off workspace (fa_bwd_dq_gemm_kernel.hpp), two 16-byte stores
// per lane and block in register order.
template <class T, int D, bool CAUSAL, bool WDS = false>
__global__ __launch_bounds__(512, 2) void fa_bwd_dkdv_kernel(const BwdParams p)
{
    constexpr int NW = 8;
    constexpr int XB = 128;                    // keys per workgroup (4 wave pairs x 32)
    constexpr int KS = D / 32, DT = D / 16, ROWB = D * 2, TILE = kBN * ROWB, PIECE = 1024;
    // staging duty goes to the four gradient waves only (macro FA_BWD_DMA_ALL: all eight): the score waves are the
    // pole of every step and a DMA piece costs its issuer about 100 cycles
#if defined(FA_BWD_DMA_ALL)
    constexpr int NDMA = NW;
#else
    constexpr int NDMA = NW / 2;
#endif
    constexpr int CPT = TILE / PIECE / NDMA;   // DMA pieces per staging wave, tile and operand
    constexpr int NS = kDkdvStages;
    constexpr int Y2BASE = NS * TILE;
    constexpr int STBASE = 2 * NS * TILE;      // NS x 1 KiB of row statistics
    constexpr int MBBASE = STBASE + NS * 1024; // mailbox: [block parity][pair][P x0, P x1, dS x0, dS x1][64 lanes x 16 B]
    static_assert(CPT >= 1, "tile too small for the workgroup");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_char*)smem;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = wave >> 2;                // 0: score wave, 1: gradient wave
    const int pair = wave & 3;
    const int lane = tid & 63;
    const int li = lane & 15;
    const int lg = lane >> 4;
    FA_BWD_ST(unsigned long long st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}; const unsigned long long st_t0 = bwd_stamp_now(); unsigned long long st_last = st_t0;)

    int head, xb;                              // key blocks in launch order: heaviest (first) first under the causal mask
    if (!wg_decode(blockIdx.x, p.bh, p.nxb, p.hsplit, head, xb)) return;
    const int b = head / p.H;
    const int hq = head - b * p.H;             // grid head = (key/value head * xsplit + group part) * qsplit + query-range part
    const int h = hq / p.qsplit;
    const int qpart = hq - h * p.qsplit;
    const int S = p.S;                         // keys
    const int Sy = p.Sy;                       // queries
    const int coff = CAUSAL ? p.coff : 0;      // key <= query + coff
    const int x0 = xb * XB;
    const int x0w = x0 + pair * 32;            // first key of this wave pair

    using elem_t = unsigned short;
    // (grid head h = key/value head * xsplit + part: the parts of a group share K and V and split its query heads)
    const int hx = h / p.xsplit;
    const elem_t* x1h = reinterpret_cast<const elem_t*>(p.x1) + b * p.x1_sb + hx * p.x1_sh;    // K
    const elem_t* x2h = reinterpret_cast<const elem_t*>(p.x2) + b * p.x2_sb + hx * p.x2_sh;    // V
    // Q and dO of the G query heads that share this key/value head (h G .. h G + G-1) are streamed one head after the other
    const int headq0 = (b * (p.H / p.qsplit) + h) * p.G;                                       // first of them, counted over all batches
    const int bhq = (p.bh / p.qsplit) * p.G;                                                   // query heads in all
    const elem_t* y1h = reinterpret_cast<const elem_t*>(p.y1) + b * p.y1_sb + (h * p.G) * p.y1_sh;   // Q
    const elem_t* y2h = reinterpret_cast<const elem_t*>(p.y2) + b * p.y2_sb + (h * p.G) * p.y2_sh;   // dO

    // ---- streamed range of the workgroup (tiles) and of this pair (blocks)
    const int nty = (Sy + kBN - 1) / kBN;
    int j_begin = 0;
    int j_end = nty;
    int blk_begin_w = 0, blk_end_w = (Sy + 31) / 32;
    int blk_mask = 0, blk_mask_hi = -1;                   // the diagonal block(s): two when coff is not a multiple of 32
    if constexpr (CAUSAL) {
        const int q_lo = x0w - coff;                      // first query that sees this pair's first key
        j_begin = min(nty, max(0, x0 - coff) / kBN);
        blk_begin_w = max(0, q_lo) >> 5;                  // earlier blocks hold only queries that see none of this pair's keys
        blk_mask = blk_begin_w;
        blk_mask_hi = (q_lo + 31 < 0) ? -1 : (q_lo + 31) >> 5;
    }
    if (p.qsplit > 1) {                                   // this workgroup's share of the visible query tiles
        const int per = (j_end - j_begin + p.qsplit - 1) / p.qsplit;
        j_begin = min(j_end, j_begin + qpart * per);
        j_end = min(j_end, j_begin + per);
        blk_begin_w = max(blk_begin_w, 2 * j_begin);
        blk_end_w = min(blk_end_w, 2 * j_end);
    }
    if (x0w >= S) { blk_begin_w = 0; blk_end_w = 0; }     // no keys: staging duty only

    // ---- staging by LDS-DMA (as fa_bwd_kernel.hpp), 8 waves
    const unsigned y1_bytes = (unsigned)(((long long)(Sy - 1) * p.y1_ss + p.dv) * 2);
    const unsigned y2_bytes = (unsigned)(((long long)(Sy - 1) * p.y2_ss + p.dv) * 2);
    u32x4 ry1, ry2;                        // descriptors of the current query head's Q and dO
    unsigned g_st = 0x80000000u;           // lanes 0-15: LSE*log2e of the tile's rows, lanes 16-31: -delta, 16 bytes each
    if (lane < 32) g_st = (unsigned)((((long long)(lane >> 4) * bhq + headq0) * p.Spad + (lane & 15) * 4) * 4);
    auto next_query_head = [&] __device__ () {          // (lanes >= 32 stay far outside the statistics descriptor)
        y1h += p.y1_sh;
        y2h += p.y2_sh;
        g_st += (unsigned)p.Spad * 4;
    };
    unsigned g_y1[CPT], g_y2[CPT];
    const int dwave = (NDMA == NW) ? wave : (wave & 3);          // staging index of this wave (gradient waves: 0..3)
    const bool stager = (NDMA == NW) || role == 1;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int byte = (dwave * CPT + i) * PIECE + lane * 16;
        const int row = byte / ROWB, chp = (byte % ROWB) / 16;
        const bool live = bwd_swz<D>(row, chp) * 8 < p.dv;       // chunks past the valid head_dim: out of the descriptor -> zeros
        g_y1[i] = live ? (unsigned)(row * p.y1_ss * 2 + bwd_swz<D>(row, chp) * 16) : 0x80000000u;
        g_y2[i] = live ? (unsigned)(row * p.y2_ss * 2 + bwd_swz<D>(row, chp) * 16) : 0x80000000u;
    }
    const unsigned y1_tile_stride = (unsigned)(kBN * p.y1_ss * 2);
    const unsigned y2_tile_stride = (unsigned)(kBN * p.y2_ss * 2);
    const unsigned piece_base = lds_base + dwave * CPT * PIECE;
    const u32x4 rst = make_rsrc(p.stats, (unsigned)((long long)2 * bhq * p.Spad * 4));
    auto issue_tile = [&] __device__ (int j, auto stage_c, auto role_c) {
        constexpr int ST = decltype(stage_c)::value;
        if constexpr (NDMA != NW && decltype(role_c)::value == 0) return;       // (known per role: the score waves' copy of the pipeline holds no staging code)
        if (!stager) return;
        if (dwave == 0) dma16(rst, __builtin_amdgcn_readfirstlane(lds_base + STBASE + ST * 1024), g_st + (unsigned)j * (kBN * 4));
#pragma unroll
        for (int i = 0; i < CPT; ++i)
            dma16(ry1, __builtin_amdgcn_readfirstlane(piece_base + ST * TILE + i * PIECE), (unsigned)j * y1_tile_stride + g_y1[i]);
#pragma unroll
        for (int i = 0; i < CPT; ++i)
            dma16(ry2, __builtin_amdgcn_readfirstlane(piece_base + Y2BASE + ST * TILE + i * PIECE), (unsigned)j * y2_tile_stride + g_y2[i]);
    };

    // ---- LDS addresses: row reads of Q / dO (score wave), transposed reads of Q / dO (gradient wave)
    unsigned ra[KS], ra2[KS], ta[DT], ta2[DT];
    {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            ra[ks] = lds_base + li * ROWB + bwd_swz<D>(li, 4 * ks + lg) * 16;
            ra2[ks] = ra[ks] + Y2BASE;
            asm volatile("" : "+v"(ra2[ks]));           // a register of its own: base + 48 KiB does not fit an immediate
        }
        const int qq = li >> 2, pp = li & 3;
        const int row = 4 * lg + qq;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            ta[dt] = lds_base + row * ROWB + bwd_swz<D>(row, 2 * dt + (pp >> 1)) * 16 + 8 * (pp & 1);
            ta2[dt] = ta[dt] + Y2BASE;
            asm volatile("" : "+v"(ta2[dt]));
        }
    }
    unsigned sta = lds_base + STBASE + lg * 16;            // statistics of rows 4 lg .. +3 (+16 yt + 32 blk): LSE2, -delta at +256
    asm volatile("" : "+v"(sta));
    unsigned mbox = lds_base + MBBASE + pair * 4096 + lane * 16;     // this lane's slot: + parity * 16 KiB + word group * 1 KiB
    asm volatile("" : "+v"(mbox));
    const float c = p.scale_log2;

    // every step starts the same way in both roles: HALF = block of the tile, ST = ring stage of tile i / 2.
    // The score waves run one block ahead with their matrix products (scores(i+1) beside softmax(i)), so tile j+1 has
    // to be visible from step 2j+1 on: the odd steps wait for it and, behind the barrier, send tile j+2 on its way into
    // the stage tile j-1 left when the gradient waves finished block 2j-1 in step 2j.
    constexpr int OPS = 2 * CPT;                                  // DMA instructions per tile and wave (wave 0: one more, issued first)
    // (WAITS: a wave that stages waits for its own DMAs; the score waves stage nothing -- with WDS they have stores in flight,
    // which nothing here waits for)
    auto open_step = [&] __device__ (auto st_c, auto half_c, int i, auto waits_c, auto role_c) {
        constexpr int ST = decltype(st_c)::value, HALF = decltype(half_c)::value;
        FA_BWD_ST(unsigned long long t_ = bwd_stamp_now(); st_sum[4] += t_ - st_last; st_last = t_; st_sum[5] += 1;)      // the previous step's work
        if constexpr (HALF == 1 && decltype(waits_c)::value) dma_wait<0>();      // tile i/2 + 1 (issued a tile ago) has landed
        FA_BWD_ST(t_ = bwd_stamp_now(); st_sum[1] += t_ - st_last; st_last = t_;)
#if !defined(FA_BWD_ABL_NOBAR)     // timing-only build without the barrier
        __syncthreads();                                          // publishes the mailbox of block i-1 (and tile i/2 + 1), retires block i-2
#endif
        FA_BWD_ST(t_ = bwd_stamp_now(); st_sum[2] += t_ - st_last; st_last = t_;)
        if constexpr (HALF == 1) {
            if ((i >> 1) + 2 < j_end) issue_tile((i >> 1) + 2, IC<(ST + 2) % NS>{}, role_c);
        }
        FA_BWD_ST(t_ = bwd_stamp_now(); st_sum[3] += t_ - st_last; st_last = t_;)
    };
    // one trip = one turn of the ring (6 blocks); step i = 2 j_end only drains the gradient waves.  The two roles run
    // separate loops (separate register sets) with the same sequence of barriers: one in the prologue (tile j_begin
    // published; the score waves compute their first scores behind it), then one per step.
    // Outer loop: the G query heads of the group, each a complete pass of the pipeline (ring and mailbox start over
    // behind a barrier; the gradient accumulators carry on).
    auto for_all_steps = [&] __device__ (auto role_c, auto&& prologue, auto&& body) {
        if (j_begin >= j_end) return;
      for (int g = 0; g < p.G; ++g) {
        if (g > 0) {
            __syncthreads();                                      // the previous head's last blocks have left the ring
            next_query_head();
        }
        ry1 = make_rsrc(y1h, y1_bytes);
        ry2 = make_rsrc(y2h, y2_bytes);
        issue_tile(j_begin, IC<0>{}, role_c);
        issue_tile(j_begin + 1, IC<1>{}, role_c);
        if constexpr (NDMA == NW || decltype(role_c)::value == 1) dma_wait<OPS>();                                          // tile j_begin has landed (tile j_begin + 1 stays in flight)
        __syncthreads();
        FA_BWD_ST({ const unsigned long long t_ = bwd_stamp_now(); st_sum[6] += t_ - st_last; st_last = t_; })
        prologue(g);
        for (int i = 2 * j_begin; i <= 2 * j_end; i += 2 * NS) {
            body(IC<0>{}, IC<0>{}, i);
            if (i + 1 <= 2 * j_end) body(IC<0>{}, IC<1>{}, i + 1);
            if (i + 2 <= 2 * j_end) body(IC<1>{}, IC<0>{}, i + 2);
            if (i + 3 <= 2 * j_end) body(IC<1>{}, IC<1>{}, i + 3);
            if (i + 4 <= 2 * j_end) body(IC<2>{}, IC<0>{}, i + 4);
            if (i + 5 <= 2 * j_end) body(IC<2>{}, IC<1>{}, i + 5);
        }
        FA_BWD_ST({ const unsigned long long t_ = bwd_stamp_now(); st_sum[4] += t_ - st_last; st_last = t_; })
      }
#if defined(FA_BWD_STAMP)
      st_sum[0] = bwd_stamp_now() - st_t0;          // (before the epilogue's stores)
      if (lane == 0)
          for (int k_ = 0; k_ < 8; ++k_) atomicAdd(&g_fa_bwd_stamp[wave][k_], st_sum[k_]);
#endif
    };

    if (role == 0) {
        // ================= score waves: S, dP - delta -> P, dS -> mailbox =================
        // stationary fragments: lane (li, lg) holds X[x0w + 16 xt + li][32 ks + 8 lg .. +7]; rows past S read as zero
        u32x4 xf1[2][KS], xf2[2][KS];
        {
            const unsigned x1_bytes = (unsigned)(((long long)(S - 1) * p.x1_ss + p.dv) * 2);
            const unsigned x2_bytes = (unsigned)(((long long)(S - 1) * p.x2_ss + p.dv) * 2);
            __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<elem_t*>(x1h), 0, x1_bytes, 0x00020000);
            __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<elem_t*>(x2h), 0, x2_bytes, 0x00020000);
#pragma unroll
            for (int xt = 0; xt < 2; ++xt) {
                const int xrow = x0w + 16 * xt + li;
                const unsigned o1 = (xrow < S) ? (unsigned)((long long)xrow * p.x1_ss * 2 + lg * 16) : 0x80000000u;
                const unsigned o2 = (xrow < S) ? (unsigned)((long long)xrow * p.x2_ss * 2 + lg * 16) : 0x80000000u;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bool live = 32 * ks + 8 * lg < p.dv;        // columns past the valid head_dim read as zero
                    xf1[xt][ks] = __builtin_amdgcn_raw_buffer_load_b128(r1, live ? o1 + ks * 64 : 0x80000000u, 0, 0);
                    xf2[xt][ks] = __builtin_amdgcn_raw_buffer_load_b128(r2, live ? o2 + ks * 64 : 0x80000000u, 0, 0);
                }
            }
            // consume the loads HERE, before any LDS-DMA is in flight: hipcc does not see the asm DMAs, and the vmcnt
            // waits it would otherwise place at the fragments' first uses inside the loop would drain them every block
#pragma unroll
            for (int xt = 0; xt < 2; ++xt)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    asm volatile("" : "+v"(xf1[xt][ks]));
                    asm volatile("" : "+v"(xf2[xt][ks]));
                }
        }
        // dS hand-off: this wave's row of units, [slab x0w / 32][NQ8 blocks][2 KiB] of the current query head
        __amdgpu_buffer_rsrc_t ds_rsrc;
        const unsigned ds_voff = (unsigned)lane * 16u;
        auto ds_descriptor = [&] __device__ (long long headq) {
            char* base = static_cast<char*>(p.ds) + headq * p.ds_head_bytes + (long long)(x0w >> 5) * p.ds_row_bytes;
            ds_rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, (x0w < S) ? p.ds_row_bytes : 0u, 0x00020000);
        };
        if constexpr (WDS) ds_descriptor(headq0);
        f32x4 t1[2][2][2], t2[2][2][2];            // score accumulators [block parity][y tile][x tile]
        // S and dP - delta of block BLK of the tile in ring stage ST, into register set BLK
        auto scores = [&] __device__ (auto st_c, auto blk_c) {
            constexpr int ST = decltype(st_c)::value, BLK = decltype(blk_c)::value;
            constexpr unsigned so = ST * TILE + BLK * 32 * ROWB;
#pragma unroll
            for (int yt = 0; yt < 2; ++yt) {
                const f32x4 nd = bitcast<f32x4>(lds_read_b128(sta + (ST * 1024 + (BLK * 32 + 16 * yt) * 4 + 256)));
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const u32x4 a1 = lds_read_b128(ra[ks] + (so + yt * 16 * ROWB));
                    const u32x4 a2 = lds_read_b128(ra2[ks] + (so + yt * 16 * ROWB));
#pragma unroll
                    for (int xt = 0; xt < 2; ++xt) {
                        t1[BLK][yt][xt] = T::mfma16(a1, xf1[xt][ks], ks == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : t1[BLK][yt][xt]);
                        t2[BLK][yt][xt] = T::mfma16(a2, xf2[xt][ks], ks == 0 ? nd : t2[BLK][yt][xt]);
                    }
                }
            }
        };
        // P, dS of block BLK (register set BLK) -> mailbox[BLK]
        auto softmax = [&] __device__ (auto mask_c, auto st_c, auto blk_c, int y0) {
            constexpr bool MASK = decltype(mask_c)::value;
            constexpr int ST = decltype(st_c)::value, BLK = decltype(blk_c)::value;
            f32x4 lse_y[2];
#pragma unroll
            for (int yt = 0; yt < 2; ++yt) lse_y[yt] = bitcast<f32x4>(lds_read_b128(sta + (ST * 1024 + (BLK * 32 + 16 * yt) * 4)));
            u32x4 pw[2], dsw[2];
#pragma unroll
            for (int xt = 0; xt < 2; ++xt) {
                const int xrow = x0w + 16 * xt + li - coff;
#pragma unroll
                for (int yt = 0; yt < 2; ++yt) {
                    float pv[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        pv[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(t1[BLK][yt][xt][e], c, -lse_y[yt][e]));
                        if constexpr (MASK) {
                            if (xrow > y0 + 16 * yt + 4 * lg + e) pv[e] = 0.f;          // key - coff > query
                        }
                    }
                    pw[xt][2 * yt] = T::pack2(pv[0], pv[1]);
                    pw[xt][2 * yt + 1] = T::pack2(pv[2], pv[3]);
                    dsw[xt][2 * yt] = T::pack2(pv[0] * t2[BLK][yt][xt][0], pv[1] * t2[BLK][yt][xt][1]);
                    dsw[xt][2 * yt + 1] = T::pack2(pv[2] * t2[BLK][yt][xt][2], pv[3] * t2[BLK][yt][xt][3]);
                }
            }
            constexpr unsigned mo = BLK * 16384;                        // mailbox parity = block of the tile
            lds_write_b128(mbox + mo, pw[0]);
            lds_write_b128(mbox + (mo + 1024), pw[1]);
            lds_write_b128(mbox + (mo + 2048), dsw[0]);
            lds_write_b128(mbox + (mo + 3072), dsw[1]);
            if constexpr (WDS) {                                        // unit (slab of this wave, block y0 / 32): [xt][lane][16 B]
                const unsigned soff = (unsigned)(y0 >> 5) * 2048u;
                __builtin_amdgcn_raw_buffer_store_b128(dsw[0], ds_rsrc, ds_voff, soff, FA_BWD_DS_STORE_AUX);
                __builtin_amdgcn_raw_buffer_store_b128(dsw[1], ds_rsrc, ds_voff + 1024u, soff, FA_BWD_DS_STORE_AUX);
            }
        };
        for_all_steps(IC<0>{},
            [&] __device__ (int g) {                                // first scores (block 2 j_begin, stage 0) behind the prologue barrier
                if constexpr (WDS) { if (g > 0) ds_descriptor(headq0 + g); }
                const int i0 = 2 * j_begin;
                if (i0 >= blk_begin_w && i0 < blk_end_w) scores(IC<0>{}, IC<0>{});
            },
            [&] __device__ (auto st_c, auto half_c, int i) {
                constexpr int ST = decltype(st_c)::value, HALF = decltype(half_c)::value;
                open_step(st_c, half_c, i, std::integral_constant<bool, NDMA == NW>{}, IC<0>{});
#if defined(FA_BWD_ABL_NOSCORE)    // timing-only build: idle score waves
                return;
#endif
                // scores of block i+1 (same tile, or the first block of the next one) beside the softmax of block i:
                // two independent instruction streams for the scheduler and the hardware
                const bool do_sc = i + 1 >= blk_begin_w && i + 1 < blk_end_w;
                const bool do_sm = i >= blk_begin_w && i < blk_end_w;
                auto next_scores = [&] __device__ () {
                    if constexpr (HALF == 0) scores(IC<ST>{}, IC<1>{});
                    else scores(IC<(ST + 1) % NS>{}, IC<0>{});
                };
                const bool diag = CAUSAL && i >= blk_mask && i <= blk_mask_hi;
                if (do_sc && do_sm && !diag) {          // steady state: one straight-line region
                    __builtin_amdgcn_sched_barrier(0);       // (also keeps the compiler from folding this path into the guarded one)
                    next_scores();
                    softmax(std::false_type{}, IC<ST>{}, IC<HALF>{}, i * 32);
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                    if (do_sc) next_scores();
                    if (do_sm) {
                        if (diag) softmax(std::true_type{}, IC<ST>{}, IC<HALF>{}, i * 32);
                        else softmax(std::false_type{}, IC<ST>{}, IC<HALF>{}, i * 32);
                    }
                }
            });
    } else {
        // ================= gradient waves: mailbox -> dV^T += dO^T P, dK^T += Q^T dS =================
        f32x4 acc1[DT][2], acc2[DT][2];            // dK^T, dV^T   [head_dim tile][key tile]
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) acc1[dt][0] = acc1[dt][1] = acc2[dt][0] = acc2[dt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto block_grads = [&] __device__ (auto st_c, auto blk_c) {
            constexpr int ST = decltype(st_c)::value, BLK = decltype(blk_c)::value;
            constexpr unsigned so = ST * TILE + BLK * 32 * ROWB;
            constexpr unsigned mo = BLK * 16384;
            u32x4 pw[2], dsw[2];
            pw[0] = lds_read_b128(mbox + mo);
            pw[1] = lds_read_b128(mbox + (mo + 1024));
            dsw[0] = lds_read_b128(mbox + (mo + 2048));
            dsw[1] = lds_read_b128(mbox + (mo + 3072));
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const u32x2 lo2 = lds_read_tr16_b64(ta2[dt] + so);
                const u32x2 hi2 = lds_read_tr16_b64(ta2[dt] + (so + 16 * ROWB));
                const u32x4 a2 = {lo2[0], lo2[1], hi2[0], hi2[1]};
                const u32x2 lo1 = lds_read_tr16_b64(ta[dt] + so);
                const u32x2 hi1 = lds_read_tr16_b64(ta[dt] + (so + 16 * ROWB));
                const u32x4 a1 = {lo1[0], lo1[1], hi1[0], hi1[1]};
#pragma unroll
                for (int xt = 0; xt < 2; ++xt) {
                    acc2[dt][xt] = T::mfma16(a2, pw[xt], acc2[dt][xt]);
                    acc1[dt][xt] = T::mfma16(a1, dsw[xt], acc1[dt][xt]);
                }
            }
        };
        for_all_steps(IC<1>{},
            [&] __device__ (int) {},
            [&] __device__ (auto st_c, auto half_c, int i) {
                constexpr int ST = decltype(st_c)::value, HALF = decltype(half_c)::value;
                open_step(st_c, half_c, i, std::true_type{}, IC<1>{});
#if defined(FA_BWD_ABL_NOGRAD)     // timing-only build: idle gradient waves
                return;
#endif
                if (i - 1 >= blk_begin_w && i - 1 < blk_end_w) {
                    // block i-1: the other half; of the previous tile when this step opens a tile
                    if constexpr (HALF == 0) block_grads(IC<(ST + NS - 1) % NS>{}, IC<1>{});
                    else block_grads(IC<ST>{}, IC<0>{});
                }
            });
        // epilogue: dK * scale, dV; layout and stores as fa_bwd_kernel.hpp -- or, when a group is split over several
        // workgroups, this part's fp32 partial sums (contiguous [B][grid heads][S][dv], 16 bytes per lane and tile)
        if (p.xsplit * p.qsplit > 1) {
#pragma unroll
            for (int which = 0; which < 2; ++which) {
                float* ph = reinterpret_cast<float*>(which == 0 ? p.out1 : p.out2) + ((long long)head * S) * p.dv;
                const float mult = which == 0 ? p.scale : 1.0f;
#pragma unroll
                for (int xt = 0; xt < 2; ++xt) {
                    const int xrow = x0w + 16 * xt + li;
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) {
                        const f32x4 a = (which == 0 ? acc1[dt][xt] : acc2[dt][xt]) * mult;
                        const int col = 16 * dt + 4 * lg;
                        if (xrow < S && col < p.dv) *reinterpret_cast<f32x4*>(ph + (long long)xrow * p.dv + col) = a;
                    }
                }
            }
            return;
        }
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            elem_t* oh = which == 0 ? reinterpret_cast<elem_t*>(p.out1) + b * p.o1_sb + h * p.o1_sh
                                    : reinterpret_cast<elem_t*>(p.out2) + b * p.o2_sb + h * p.o2_sh;
            const long long o_ss = which == 0 ? p.o1_ss : p.o2_ss;
            const float mult = which == 0 ? p.scale : 1.0f;
#pragma unroll
            for (int xt = 0; xt < 2; ++xt) {
                const int xrow = x0w + 16 * xt + li;
                elem_t* orow = oh + (long long)xrow * o_ss;
#pragma unroll
                for (int dt = 0; dt < DT; dt += 2) {
                    const f32x4 oa = which == 0 ? acc1[dt][xt] : acc2[dt][xt];
                    const f32x4 ob = which == 0 ? acc1[dt + 1][xt] : acc2[dt + 1][xt];
                    unsigned a0 = T::pack2(oa[0] * mult, oa[1] * mult), a1 = T::pack2(oa[2] * mult, oa[3] * mult);
                    unsigned b0 = T::pack2(ob[0] * mult, ob[1] * mult), b1 = T::pack2(ob[2] * mult, ob[3] * mult);
                    auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
                    auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
                    u32x4 outv = {s0[0], s1[0], s0[1], s1[1]};
                    const int col = (lg & 1) ? (16 * (dt + 1) + 4 * (lg - 1)) : (16 * dt + 4 * lg);
                    if (xrow < S && col < p.dv) *reinterpret_cast<u32x4*>(orow + col) = outv;
                }
            }
        }
    }
}

}  // namespace fa
