// FlashAttention backward for gfx950 -- dQ from a dS hand-off (the "five products" backward).
//
// The recompute backward (fa_bwd_kernel.hpp MODE 0 + fa_bwd_dkdv_kernel.hpp) forms S and dP twice: 7 matrix products
// for the 5 of the textbook count, and the softmax arithmetic twice.  This path trades HBM capacity and bandwidth --
// which an MI355X has to spare (288 GB, 8 TB/s; the backward is MFMA / issue bound) -- for that work:
//
//   fa_bwd_dkdv_kernel<.., WDS = true>  writes the 16-bit dS it hands to its gradient waves ALSO to a workspace,
//                                       one 16-byte store per lane and 16-key x 32-query unit, in register order;
//   fa_bwd_dq_gemm_kernel (this file)   dQ = scale * dS . K: ONE product, no exp, no statistics -- a streaming GEMM
//                                       over that workspace (read once: HBM-bound) and the K tiles (L2-resident).
//
// Workspace image (producer's register order, so that its stores are whole 1 KiB wave-instructions):
//   unit(head, s, i) = 2 KiB at ((head * NK2 + s) * NQ8 + i) * 2048   s = 32-key slab, i = 32-query block
//   unit[xt][L = 16 lg + li][16 B] = dS[key 32 s + 16 xt + li][queries 32 i + 16 yt + 4 lg + e],  byte 8 yt + 2 e
// (lane (li, lg) of the score wave holds exactly these eight values in dsw[xt]).  Units of (s, i) pairs with no visible
// element under the causal mask are never written and never read.
//
// Consumer: a workgroup owns 256 query rows (8 waves x 32 = 8 blocks i), streams 64-key tiles: the K tile (shared, as
// MODE 0 of fa_bwd_kernel.hpp) and, per wave, ITS OWN four 1 KiB pieces of dS (2 slabs x 2 xt) by LDS-DMA into a
// wave-private part of the ring stage.  dQ^T[d][q] += K^T[d][key] . dS^T[key][q] on 16x16x32 MFMAs:
//   A = K^T through ds_read_b64_tr_b16 of the row-major K tile (k slots 4 lg + e, 16 + 4 lg + e: fa_bwd_kernel.hpp),
//   B = dS^T through ds_read_b64_tr_b16 of the unit: a 16-lane group's lanes 4 qq + pp supply key 4 lg' + qq (+ 16 xt)
//       and the 8-byte chunk (lg_src, yt) = (pp >> 1 (+ 2 t), pp & 1), so one read pair (xt = 0, 1) gives lane li' the
//       eight keys of its k slots for query column li' of "query tile" t:  query = 16 (pp & 1) + 4 ((pp >> 1) + 2 t) + e,
//       li' = 4 pp + e.  The permutation of the queries inside a block is undone by the epilogue's row addresses.
//   LDS position of producer lane L inside its 1 KiB: L ^ (((L >> 4) & 1) << 3), applied on the DMA's source address:
//       the 32 lanes of a read half then cover the 64 banks exactly once (conflict-free).
// Per wave and 64-key tile: 32 MFMAs, 40 LDS reads, 4 + CPT DMA pieces, nothing else.
#pragma once
#include "fa_bwd_kernel.hpp"

#ifndef FA_BWD_DS_LOAD_NT
#define FA_BWD_DS_LOAD_NT 1            // the dS pieces (read once) are fetched with the nt cache policy: +1.5 % causal over 0
#endif

namespace fa {

// LDS-DMA as dma16, for bytes read exactly once
__device__ __forceinline__ void dma16_once(u32x4 rsrc, unsigned lds_addr, unsigned voff) {
#if FA_BWD_DS_LOAD_NT
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen nt lds"
                 :: "v"(voff), "s"(rsrc), "s"(lds_addr) : "memory");
#else
    dma16(rsrc, lds_addr, voff);
#endif
}

constexpr int kDqgStages = 3;
constexpr int kDqgDsTile = 8 * 4 * 1024;      // 8 waves x (2 slabs x 2 xt) x 1 KiB
template <int D> constexpr int dqg_lds_bytes() { return kDqgStages * (kBN * D * 2 + kDqgDsTile); }

// position <-> producer lane inside a 1 KiB piece (an involution)
__device__ __forceinline__ int dqg_pi(int L) { return L ^ (((L >> 4) & 1) << 3); }

template <class T, int D, bool CAUSAL>
__global__ __launch_bounds__(512, 2) void fa_bwd_dq_gemm_kernel(const BwdParams p)
{
    constexpr int NW = 8;
    constexpr int XB = NW * 32;
    constexpr int DT = D / 16;
    constexpr int ROWB = D * 2;
    constexpr int TILE = kBN * ROWB;
    constexpr int PIECE = 1024;
    constexpr int CPT = TILE / PIECE / NW;     // K pieces per wave and tile
    constexpr int NS = kDqgStages;
    constexpr int DSBASE = NS * TILE;          // K ring first; then per wave its own dS ring, NS x 4 KiB (all LDS offsets of a
    constexpr int DSW = NS * 4 * PIECE;        // read stay below the 64 KiB an instruction immediate can hold)
    constexpr int OPS = CPT + 4;               // DMA instructions per wave and tile
    static_assert(CPT >= 1, "tile too small for the workgroup");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_char*)smem;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const bool PAIR = CAUSAL && !p.unpaired;
    const int wg_per_head = PAIR ? (p.nxb + 1) / 2 : p.nxb;
    int head, t;
    if (!wg_decode(blockIdx.x, p.bh, wg_per_head, p.hsplit, head, t)) return;
    const int b = head / p.H;
    const int h = head - b * p.H;
    const int S = p.S;                         // queries
    const int Sy = p.Sy;                       // keys
    const int coff = CAUSAL ? p.coff : 0;
    const int n_pass = (PAIR && p.nxb - 1 - t != t) ? 2 : 1;
    using elem_t = unsigned short;
    const elem_t* y1h = reinterpret_cast<const elem_t*>(p.y1) + b * p.y1_sb + (h / p.G) * p.y1_sh;     // K of this query head
    const unsigned y1_bytes = (unsigned)(((long long)(Sy - 1) * p.y1_ss + p.dv) * 2);
    const u32x4 ry1 = make_rsrc(y1h, y1_bytes);
    const int nslab = (Sy + 31) >> 5;

  for (int pass = 0; pass < n_pass; ++pass) {
    const int xb = CAUSAL ? (pass == 0 ? p.nxb - 1 - t : t) : t;
    const int x0 = xb * XB;
    const int lane = lane_here();
    // waves w and w + 4 share a SIMD: under the causal mask row blocks that sum to 7 give every SIMD the same work
    const int rowblk_of_wave = (CAUSAL && wave >= 4) ? 11 - wave : wave;
    const int x0w = x0 + rowblk_of_wave * 32;
    const int blk = x0w >> 5;                  // this wave's 32-query block i

    // slabs (32 keys) this wave multiplies: those with a visible element, i.e. written by the producer
    int s_end_w = nslab;
    if constexpr (CAUSAL) s_end_w = max(0, min(nslab, ((32 * blk + 31 + coff) >> 5) + 1));
    if (x0w >= S) s_end_w = 0;
    int s_end_wg = nslab;
    if constexpr (CAUSAL) s_end_wg = max(0, min(nslab, ((min(S, x0 + XB) - 1 + coff) >> 5) + 1));
    const int j_end = (s_end_wg + 1) >> 1;     // 64-key tiles the workgroup stages

    // ---- staging: K pieces as MODE 0 of fa_bwd_kernel.hpp; this wave's own dS pieces (slab sg, xt)
    unsigned g_y1[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int byte = (wave * CPT + i) * PIECE + lane * 16;
        const int row = byte / ROWB, chp = (byte % ROWB) / 16;
        const bool live = bwd_swz<D>(row, chp) * 8 < p.dv;
        g_y1[i] = live ? (unsigned)(row * p.y1_ss * 2 + bwd_swz<D>(row, chp) * 16) : 0x80000000u;
    }
    const unsigned y1_tile_stride = (unsigned)(kBN * p.y1_ss * 2);
    const unsigned k_piece_base = lds_base + wave * CPT * PIECE;
    const unsigned ds_piece_base = lds_base + DSBASE + wave * DSW;
    // this wave's column of units, [slab][NQ8 blocks][2 KiB]: a descriptor per 64-key tile, from (slab 2 j, block blk) to the end of
    // slab 2 j + 1's row -- a head's image may pass the 4 GiB a 32-bit offset reaches (S_q S_k >= 2^31), two slab rows never do
    const unsigned ds_row = p.ds_row_bytes;                         // NQ8 * 2048
    const char* dsh = reinterpret_cast<const char*>(p.ds) + (long long)head * p.ds_head_bytes + (long long)blk * 2048;
    const unsigned ds_bytes = x0w < S ? 2u * ds_row - (unsigned)blk * 2048u : 0u;
    const unsigned g_ds = (unsigned)dqg_pi(lane) * 16;
    auto issue_tile = [&](int j, int stage) {
        const u32x4 rds = make_rsrc(dsh + (long long)j * (2ll * ds_row), ds_bytes);
#pragma unroll
        for (int i = 0; i < CPT; ++i)
            dma16(ry1, __builtin_amdgcn_readfirstlane(k_piece_base + stage * TILE + i * PIECE), (unsigned)j * y1_tile_stride + g_y1[i]);
#pragma unroll
        for (int sg = 0; sg < 2; ++sg) {
            // a slab past this wave's range has no unit written for this block: its pieces are pushed out of the descriptor
            // (zeros, no memory traffic); the DMA count per tile stays the same for the counted waits
            const unsigned off = (2 * j + sg < s_end_w) ? (unsigned)sg * ds_row + g_ds : 0x80000000u;
#pragma unroll
            for (int xt = 0; xt < 2; ++xt)
                dma16_once(rds, __builtin_amdgcn_readfirstlane(ds_piece_base + (stage * 4 + 2 * sg + xt) * PIECE), off + xt * 1024);
        }
    };

    f32x4 acc[DT][2];                          // dQ^T [head_dim tile][query tile t]
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) acc[dt][0] = acc[dt][1] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (j_end > 0) {
        issue_tile(0, 0);
        issue_tile(1, 1);
        // ---- LDS read addresses (stage 0, slab 0), formed behind the prologue's DMA issue
        unsigned ta[DT], tb[2];
        {
            const int lane_a = lane_here();
            const int li_a = lane_a & 15, lg_a = lane_a >> 4;
            const int qq = li_a >> 2, pp = li_a & 3;
            const int row = 4 * lg_a + qq;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
                ta[dt] = lds_base + row * ROWB + bwd_swz<D>(row, 2 * dt + (pp >> 1)) * 16 + 8 * (pp & 1);
#pragma unroll
            for (int tq = 0; tq < 2; ++tq) {
                const int L = ((pp >> 1) + 2 * tq) * 16 + row;            // producer lane: lg_src = (pp >> 1) + 2 t, li_src = 4 lg' + qq
                tb[tq] = lds_base + DSBASE + wave * DSW + dqg_pi(L) * 16 + 8 * (pp & 1);
            }
        }
        auto slab = [&] __device__ (auto st_c, auto sg_c) {
            constexpr int ST = decltype(st_c)::value, SG = decltype(sg_c)::value;
            constexpr unsigned ko = ST * TILE + SG * 32 * ROWB;
            constexpr unsigned dso = (ST * 4 + SG * 2) * PIECE;
            // every fragment of the slab is requested before the first product (DT + 2 register quads): a wave's LDS latency is
            // paid once per slab, not once per head_dim tile
            u32x4 bf[2], af[DT];
#pragma unroll
            for (int tq = 0; tq < 2; ++tq) {
                const u32x2 lo = lds_read_tr16_b64(tb[tq] + dso);
                const u32x2 hi = lds_read_tr16_b64(tb[tq] + (dso + PIECE));
                bf[tq] = u32x4{lo[0], lo[1], hi[0], hi[1]};
            }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const u32x2 lo = lds_read_tr16_b64(ta[dt] + ko);
                const u32x2 hi = lds_read_tr16_b64(ta[dt] + (ko + 16 * ROWB));
                af[dt] = u32x4{lo[0], lo[1], hi[0], hi[1]};
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int tq = 0; tq < 2; ++tq) acc[dt][tq] = T::mfma16(af[dt], bf[tq], acc[dt][tq]);
        };
        auto tile = [&] __device__ (auto st_c, int j) {
            constexpr int ST = decltype(st_c)::value;
            dma_wait<OPS>();                          // this wave's pieces of tile j have landed ...
            __syncthreads();                          // ... every wave's are visible, and tile j-1 is no longer read
            issue_tile(j + 2, (ST + 2) % NS);
            if (2 * j < s_end_w) slab(IC<ST>{}, IC<0>{});
            if (2 * j + 1 < s_end_w) slab(IC<ST>{}, IC<1>{});
        };
        for (int j = 0; j < j_end; j += NS) {
            tile(IC<0>{}, j);
            if (j + 1 < j_end) tile(IC<1>{}, j + 1);
            if (j + 2 < j_end) tile(IC<2>{}, j + 2);
        }
        dma_wait<0>();                                // no DMA may still be writing LDS when the ring is reused or the workgroup retires
    }

    // ---- epilogue: dQ[x][16 dt + 4 lg + 0..3] = acc[dt][t] * scale, x = the lane's query of tile t (see the header)
    {
        elem_t* oh = reinterpret_cast<elem_t*>(p.out1) + b * p.o1_sb + h * p.o1_sh;
        const int lane_e = lane_here();
        const int li = lane_e & 15, lg = lane_e >> 4;
        const int pp = li >> 2, e = li & 3;
        const float mult = p.scale;
#pragma unroll
        for (int tq = 0; tq < 2; ++tq) {
            const int xrow = x0w + 16 * (pp & 1) + 4 * ((pp >> 1) + 2 * tq) + e;
            elem_t* orow = oh + (long long)xrow * p.o1_ss;
#pragma unroll
            for (int dt = 0; dt < DT; dt += 2) {
                const f32x4 oa = acc[dt][tq], ob = acc[dt + 1][tq];
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
    if (pass + 1 < n_pass) __syncthreads();           // every wave is done with the ring before the next pass refills it
  }  // pass
}

}  // namespace fa
