// FlashAttention backward for gfx950 -- dK / dV kernel with wave-specialised workgroups.
//
// Same products, fragment maps, LDS image and causal rules as MODE 1 of fa_bwd_kernel.hpp (read its header first):
// a workgroup owns 128 keys and streams the Q / dO tiles of its head.  The difference is who holds what.  There, one
// wave per SIMD carries both 32x128 f32 accumulators (512 registers) and has no partner to overlap its softmax with.
// Here the work of a 32-key slice is split between TWO waves that share a SIMD, so both fit 256 registers and the
// hardware interleaves them, without recomputing anything:
//
//   wave w   (0..3, role P): S = Q K^T, P = exp2(S c - LSE2) [masked], publishes P (16-bit, packed in register order)
//                            in an LDS mailbox, accumulates dV^T += dO^T P           (K fragments + dV^T in registers)
//   wave w+4 (role S)      : dP' = dO V^T - delta, reads P from the mailbox, dS = P o dP', accumulates
//                            dK^T += Q^T dS                                          (V fragments + dK^T in registers)
//
// Both roles keep the 16x16x32 fragment maps of fa_bwd_kernel.hpp, so the mailbox is a lane-to-lane hand-over: lane l
// of the P wave writes the eight packed words that lane l of the S wave needs (two conflict-free ds_write_b128 /
// ds_read_b128 per 32-row block).  The S wave runs one tile behind the P wave; the one barrier per tile that publishes
// a staged tile also publishes the mailbox (double-buffered by tile parity).  Ring: 3 stages -- in phase f the P waves
// read tile f, the S waves tile f-1, tile f+1 is in flight into the stage tile f-2 just left.
// dS is formed from the 16-bit P (the P wave's fp32 P never leaves its registers): one more rounding of 2^-9 (bf16)
// on dK's operand, inside the stated tolerance.  The causal mask is applied by the P wave only (P = 0 gives dS = 0).
#pragma once
#include "fa_bwd_kernel.hpp"

namespace fa {

constexpr int kDkdvStages = 3;
template <int D> constexpr int dkdv_lds_bytes() { return 2 * kDkdvStages * kBN * D * 2 + kDkdvStages * 1024 + 2 * 4 * 2 * 2048; }

template <class T> __device__ __forceinline__ void unpack2(unsigned w, float& a, float& b) {
    if constexpr (std::is_same<T, TypeBF16>::value) {
        a = bitcast<float>(w << 16);
        b = bitcast<float>(w & 0xffff0000u);
    } else {
        const f16x2 h = bitcast<f16x2>(w);
        a = (float)h[0];
        b = (float)h[1];
    }
}

template <class T, int D, bool CAUSAL>
__global__ __launch_bounds__(512, 2) void fa_bwd_dkdv_kernel(const BwdParams p)
{
    constexpr int NW = 8;
    constexpr int XB = 128;                    // keys per workgroup (4 wave pairs x 32)
    constexpr int KS = D / 32, DT = D / 16, ROWB = D * 2, TILE = kBN * ROWB, PIECE = 1024;
    constexpr int CPT = TILE / PIECE / NW;     // DMA pieces per wave, tile and operand
    constexpr int NS = kDkdvStages;
    constexpr int Y2BASE = NS * TILE;
    constexpr int STBASE = 2 * NS * TILE;      // NS x 1 KiB of row statistics
    constexpr int MBBASE = STBASE + NS * 1024; // mailbox: [tile parity][pair][block][2 x 1 KiB]
    static_assert(CPT >= 1, "tile too small for the workgroup");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_char*)smem;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = wave >> 2;                // 0: P wave (S, P, dV)   1: S wave (dP, dS, dK)
    const int pair = wave & 3;
    const int lane = tid & 63;
    const int li = lane & 15;
    const int lg = lane >> 4;

    const int bid = blockIdx.x;
    const int xcd = bid & 7;
    const int slot = bid >> 3;
    const int hl = slot / p.nxb;
    const int xb = slot - hl * p.nxb;          // key blocks in launch order: heaviest (first) first under the causal mask
    const int head = hl * 8 + xcd;
    if (head >= p.bh) return;
    const int b = head / p.H;
    const int h = head - b * p.H;
    const int S = p.S;
    const int x0 = xb * XB;
    const int x0w = x0 + pair * 32;            // first key of this wave pair

    using elem_t = unsigned short;
    // role P holds K (x1), role S holds V (x2); both stream Q (y1) and dO (y2)
    const elem_t* xh = role == 0 ? reinterpret_cast<const elem_t*>(p.x1) + b * p.x1_sb + h * p.x1_sh
                                 : reinterpret_cast<const elem_t*>(p.x2) + b * p.x2_sb + h * p.x2_sh;
    const long long x_ss = role == 0 ? p.x1_ss : p.x2_ss;
    const elem_t* y1h = reinterpret_cast<const elem_t*>(p.y1) + b * p.y1_sb + h * p.y1_sh;
    const elem_t* y2h = reinterpret_cast<const elem_t*>(p.y2) + b * p.y2_sb + h * p.y2_sh;

    // ---- streamed range of the workgroup (tiles) and of this pair (blocks)
    const int nty = (S + kBN - 1) / kBN;
    int j_begin = 0;
    const int j_end = nty;
    int blk_begin_w = 0, blk_end_w = (S + 31) / 32;
    int blk_mask = -1;                                    // the diagonal block
    if constexpr (CAUSAL) {
        j_begin = min(nty, x0 / kBN);
        blk_begin_w = x0w >> 5;                           // earlier blocks hold only queries < this pair's keys
        blk_mask = x0w >> 5;
    }
    if (x0w >= S) { blk_begin_w = 0; blk_end_w = 0; }     // no keys: staging duty only

    // ---- stationary fragments: lane (li, lg) holds X[x0w + 16 xt + li][32 ks + 8 lg .. +7]
    u32x4 xf[2][KS];
    {
        const unsigned x_bytes = (unsigned)(((long long)(S - 1) * x_ss + D) * 2);
        __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<elem_t*>(xh), 0, x_bytes, 0x00020000);
#pragma unroll
        for (int xt = 0; xt < 2; ++xt) {
            const int xrow = x0w + 16 * xt + li;
            const unsigned off = (xrow < S) ? (unsigned)((long long)xrow * x_ss * 2 + lg * 16) : 0x80000000u;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) xf[xt][ks] = __builtin_amdgcn_raw_buffer_load_b128(rx, off + ks * 64, 0, 0);
        }
    }

    // ---- staging by LDS-DMA (as fa_bwd_kernel.hpp), 8 waves
    const unsigned y1_bytes = (unsigned)(((long long)(S - 1) * p.y1_ss + D) * 2);
    const unsigned y2_bytes = (unsigned)(((long long)(S - 1) * p.y2_ss + D) * 2);
    const u32x4 ry1 = make_rsrc(y1h, y1_bytes);
    const u32x4 ry2 = make_rsrc(y2h, y2_bytes);
    unsigned g_y1[CPT], g_y2[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int byte = (wave * CPT + i) * PIECE + lane * 16;
        const int row = byte / ROWB, chp = (byte % ROWB) / 16;
        g_y1[i] = (unsigned)(row * p.y1_ss * 2 + bwd_swz<D>(row, chp) * 16);
        g_y2[i] = (unsigned)(row * p.y2_ss * 2 + bwd_swz<D>(row, chp) * 16);
    }
    const unsigned y1_tile_stride = (unsigned)(kBN * p.y1_ss * 2);
    const unsigned y2_tile_stride = (unsigned)(kBN * p.y2_ss * 2);
    const unsigned piece_base = lds_base + wave * CPT * PIECE;
    const u32x4 rst = make_rsrc(p.stats, (unsigned)((long long)2 * p.bh * p.Spad * 4));
    unsigned g_st = 0x80000000u;           // lanes 0-15: LSE*log2e of the tile's rows, lanes 16-31: -delta, 16 bytes each
    if (lane < 32) g_st = (unsigned)((((long long)(lane >> 4) * p.bh + head) * p.Spad + (lane & 15) * 4) * 4);
    auto issue_tile = [&] __device__ (int j, auto stage_c) {
        constexpr int ST = decltype(stage_c)::value;
        if (wave == 0) dma16(rst, __builtin_amdgcn_readfirstlane(lds_base + STBASE + ST * 1024), g_st + (unsigned)j * (kBN * 4));
#pragma unroll
        for (int i = 0; i < CPT; ++i)
            dma16(ry1, __builtin_amdgcn_readfirstlane(piece_base + ST * TILE + i * PIECE), (unsigned)j * y1_tile_stride + g_y1[i]);
#pragma unroll
        for (int i = 0; i < CPT; ++i)
            dma16(ry2, __builtin_amdgcn_readfirstlane(piece_base + Y2BASE + ST * TILE + i * PIECE), (unsigned)j * y2_tile_stride + g_y2[i]);
    };

    // ---- LDS read addresses.  Role P reads Q by rows (scores) and dO transposed (dV); role S reads dO by rows (dP)
    //      and Q transposed (dK): the row-read bases point into one ring, the transposed-read bases into the other.
    unsigned ra[KS], ta[DT];
    {
        const unsigned row_ring = lds_base + (role == 0 ? 0 : Y2BASE);
        const unsigned tr_ring = lds_base + (role == 0 ? Y2BASE : 0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) ra[ks] = row_ring + li * ROWB + bwd_swz<D>(li, 4 * ks + lg) * 16;
        const int qq = li >> 2, pp = li & 3;
        const int row = 4 * lg + qq;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) ta[dt] = tr_ring + row * ROWB + bwd_swz<D>(row, 2 * dt + (pp >> 1)) * 16 + 8 * (pp & 1);
    }
    // statistics of rows 4 lg .. +3 (+16 yt + 32 blk): role P needs LSE2 (first 256 bytes of a slot), role S -delta
    const unsigned sta = lds_base + STBASE + lg * 16 + (role == 0 ? 0 : 256);
    // mailbox slot of this lane: [parity][pair][blk][half][lane]
    const unsigned mbox = lds_base + MBBASE + pair * 4096 + lane * 16;

    f32x4 acc[DT][2];          // role P: dV^T, role S: dK^T   [head_dim tile][key tile]
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) acc[dt][0] = acc[dt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float c = p.scale_log2;

    auto grads = [&] __device__ (u32x4 (&w)[2], unsigned so) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
#if defined(FA_BWD_ABL_NOTR)      // timing-only build: no transposed reads
            const u32x4 a = {ta[dt], so, (unsigned)dt, 1u};
#else
            const u32x2 lo = lds_read_tr16_b64(ta[dt] + so);
            const u32x2 hi = lds_read_tr16_b64(ta[dt] + (so + 16 * ROWB));
            const u32x4 a = {lo[0], lo[1], hi[0], hi[1]};
#endif
#pragma unroll
            for (int xt = 0; xt < 2; ++xt) acc[dt][xt] = T::mfma16(a, w[xt], acc[dt][xt]);
        }
    };
    // role P, one 32-row block: scores -> P -> mailbox -> dV
    auto block_p = [&] __device__ (auto mask_c, auto st_c, auto blk_c, unsigned mb, int y0) {
        constexpr bool MASK = decltype(mask_c)::value;
        constexpr int ST = decltype(st_c)::value, BLK = decltype(blk_c)::value;
        constexpr unsigned so = ST * TILE + BLK * 32 * ROWB;
        f32x4 t[2][2];
        f32x4 lse_y[2];
#pragma unroll
        for (int yt = 0; yt < 2; ++yt) {
            lse_y[yt] = bitcast<f32x4>(lds_read_b128(sta + (ST * 1024 + (BLK * 32 + 16 * yt) * 4)));
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#if defined(FA_BWD_ABL_NOROW)     // timing-only build: no row reads
                const u32x4 a = {ra[ks], so, (unsigned)yt, 1u};
#else
                const u32x4 a = lds_read_b128(ra[ks] + (so + yt * 16 * ROWB));
#endif
#pragma unroll
                for (int xt = 0; xt < 2; ++xt)
                    t[yt][xt] = T::mfma16(a, xf[xt][ks], ks == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : t[yt][xt]);
            }
        }
        u32x4 pw[2];
#pragma unroll
        for (int xt = 0; xt < 2; ++xt) {
            const int xrow = x0w + 16 * xt + li;
#pragma unroll
            for (int yt = 0; yt < 2; ++yt) {
                float pv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
#if defined(FA_BWD_ABL_NOEXP)     // timing-only build: no exponentials
                    pv[e] = __builtin_fmaf(t[yt][xt][e], c, -lse_y[yt][e]);
#else
                    pv[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(t[yt][xt][e], c, -lse_y[yt][e]));
#endif
                    if constexpr (MASK) {
                        if (xrow > y0 + 16 * yt + 4 * lg + e) pv[e] = 0.f;          // key > query
                    }
                }
                pw[xt][2 * yt] = T::pack2(pv[0], pv[1]);
                pw[xt][2 * yt + 1] = T::pack2(pv[2], pv[3]);
            }
        }
#if !defined(FA_BWD_ABL_NOMBOX)    // timing-only build: no mailbox traffic
        lds_write_b128(mb + BLK * 2048, pw[0]);
        lds_write_b128(mb + BLK * 2048 + 1024, pw[1]);
#endif
        grads(pw, so);
    };
    // role S, one 32-row block: dP - delta -> (mailbox P) -> dS -> dK
    auto block_s = [&] __device__ (auto st_c, auto blk_c, unsigned mb) {
        constexpr int ST = decltype(st_c)::value, BLK = decltype(blk_c)::value;
        constexpr unsigned so = ST * TILE + BLK * 32 * ROWB;
        f32x4 t[2][2];
#pragma unroll
        for (int yt = 0; yt < 2; ++yt) {
            const f32x4 nd = bitcast<f32x4>(lds_read_b128(sta + (ST * 1024 + (BLK * 32 + 16 * yt) * 4)));
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#if defined(FA_BWD_ABL_NOROW)
                const u32x4 a = {ra[ks], so, (unsigned)yt, 1u};
#else
                const u32x4 a = lds_read_b128(ra[ks] + (so + yt * 16 * ROWB));
#endif
#pragma unroll
                for (int xt = 0; xt < 2; ++xt) t[yt][xt] = T::mfma16(a, xf[xt][ks], ks == 0 ? nd : t[yt][xt]);
            }
        }
        u32x4 pin[2], dsw[2];
#if defined(FA_BWD_ABL_NOMBOX)
        pin[0] = pin[1] = u32x4{mb, 1u, 2u, 3u};
#else
        pin[0] = lds_read_b128(mb + BLK * 2048);
        pin[1] = lds_read_b128(mb + BLK * 2048 + 1024);
#endif
#pragma unroll
        for (int xt = 0; xt < 2; ++xt)
#pragma unroll
            for (int yt = 0; yt < 2; ++yt) {
                float p0, p1, p2, p3;
                unpack2<T>(pin[xt][2 * yt], p0, p1);
                unpack2<T>(pin[xt][2 * yt + 1], p2, p3);
                dsw[xt][2 * yt] = T::pack2(p0 * t[yt][xt][0], p1 * t[yt][xt][1]);
                dsw[xt][2 * yt + 1] = T::pack2(p2 * t[yt][xt][2], p3 * t[yt][xt][3]);
            }
        grads(dsw, so);
    };

    // ---- phases: in phase f (tile index) the P waves work on tile f, the S waves on tile f-1
    auto phase = [&] __device__ (auto st_c, int f) {
        constexpr int ST = decltype(st_c)::value;                 // ring stage of tile f
#if !defined(FA_BWD_ABL_NOWAIT)   // timing-only build: how much of the phase is DMA latency?
        dma_wait<0>();                                            // this wave's pieces of tile f (issued a phase ago) have landed
#endif
#if !defined(FA_BWD_ABL_NOBAR)
        __syncthreads();                                          // tile f and the mailbox of tile f-1 are published; tile f-2 is retired
#endif
        if (f + 1 < j_end) issue_tile(f + 1, IC<(ST + 1) % NS>{});
        if (role == 0) {
            if (f < j_end) {
                const unsigned mb = mbox + (f & 1) * 16384;
                const int b0 = 2 * f, b1 = 2 * f + 1;
                if (b0 >= blk_begin_w && b0 < blk_end_w) {
                    if (CAUSAL && b0 == blk_mask) block_p(std::true_type{}, IC<ST>{}, IC<0>{}, mb, b0 * 32);
                    else block_p(std::false_type{}, IC<ST>{}, IC<0>{}, mb, b0 * 32);
                }
                if (b1 >= blk_begin_w && b1 < blk_end_w) {
                    if (CAUSAL && b1 == blk_mask) block_p(std::true_type{}, IC<ST>{}, IC<1>{}, mb, b1 * 32);
                    else block_p(std::false_type{}, IC<ST>{}, IC<1>{}, mb, b1 * 32);
                }
            }
        } else {
            if (f > j_begin) {
                const unsigned mb = mbox + ((f - 1) & 1) * 16384;
                const int b0 = 2 * f - 2, b1 = 2 * f - 1;
                if (b0 >= blk_begin_w && b0 < blk_end_w) block_s(IC<(ST + NS - 1) % NS>{}, IC<0>{}, mb);
                if (b1 >= blk_begin_w && b1 < blk_end_w) block_s(IC<(ST + NS - 1) % NS>{}, IC<1>{}, mb);
            }
        }
    };
    if (j_begin < j_end) {
        issue_tile(j_begin, IC<0>{});
        for (int f = j_begin; f <= j_end; f += NS) {              // one trip = one turn of the ring; f = j_end: the S waves' last tile
            phase(IC<0>{}, f);
            if (f + 1 <= j_end) phase(IC<1>{}, f + 1);
            if (f + 2 <= j_end) phase(IC<2>{}, f + 2);
        }
    }

    // ---- epilogue: role P stores dV, role S stores dK * scale (layout and stores as fa_bwd_kernel.hpp)
    {
        elem_t* oh = role == 0 ? reinterpret_cast<elem_t*>(p.out2) + b * p.o2_sb + h * p.o2_sh
                               : reinterpret_cast<elem_t*>(p.out1) + b * p.o1_sb + h * p.o1_sh;
        const long long o_ss = role == 0 ? p.o2_ss : p.o1_ss;
        const float mult = role == 0 ? 1.0f : p.scale;
#pragma unroll
        for (int xt = 0; xt < 2; ++xt) {
            const int xrow = x0w + 16 * xt + li;
            elem_t* orow = oh + (long long)xrow * o_ss;
#pragma unroll
            for (int dt = 0; dt < DT; dt += 2) {
                const f32x4 oa = acc[dt][xt], ob = acc[dt + 1][xt];
                unsigned a0 = T::pack2(oa[0] * mult, oa[1] * mult), a1 = T::pack2(oa[2] * mult, oa[3] * mult);
                unsigned b0 = T::pack2(ob[0] * mult, ob[1] * mult), b1 = T::pack2(ob[2] * mult, ob[3] * mult);
                auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
                auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
                u32x4 outv = {s0[0], s1[0], s0[1], s1[1]};
                if (xrow < S) {
                    const int col = (lg & 1) ? (16 * (dt + 1) + 4 * (lg - 1)) : (16 * dt + 4 * lg);
                    *reinterpret_cast<u32x4*>(orow + col) = outv;
                }
            }
        }
    }
}

}  // namespace fa
