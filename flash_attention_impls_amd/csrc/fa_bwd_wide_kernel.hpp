// FlashAttention backward for gfx950 -- wide heads (head_dim 144 .. 256): the P / dS producer.
//
// SURVEY.md section 8(f) row N2 lists head_dim 256; the reference stops at 128 (FA2-triton.py:178), so like the forward of
// fa_fwd_kernel_wide.hpp this is coverage of that row, not a tuned kernel.  At head_dim 256 neither backward kernel of
// fa_bwd_kernel.hpp fits a register file: the dK and dV accumulators of 32 keys alone are 256 registers per lane.  The backward
// is therefore split where the head_dim stops mattering:
//
//   * this kernel forms, per 128-row query block and 64-key tile, S^T = K Q^T and dP^T = V dO^T (both contract over the whole
//     head_dim: 2 x 64 MFMAs 16x16x32, Q and dO fragments in registers as in the forward), then
//         P = exp2(S scale log2(e) - LSE log2(e))            (FA2-triton.py:128-131 with the saved statistics, masked -> 0)
//         dS = scale P (dP - delta)                           (FA2-triton.py:140-143; -delta seeds the dP accumulators)
//     and writes both as 16-bit row-major images [query head][query][key] (row stride `ld` elements) -- exactly the operands
//     the three products that remain are plain GEMMs over:  dV = P^T dO,  dK = dS^T Q,  dQ = dS K   (FA2-triton.py:134,146,150);
//   * those GEMMs carry no attention-specific structure any more and run on the vendor library (the host side of
//     flash_attn.py: torch.bmm -> hipBLASLt / rocBLAS; a C caller uses its own BLAS on the two images).
//
// Every (query row, key) of the images is written exactly once -- zeros where the causal mask or the end of the key range
// hides a key, also in the tiles the block never computes on -- so the workspace needs no clearing.
#pragma once
#include "fa_bwd_kernel.hpp"

namespace fa {

constexpr int kBwdWideRows = 128;                          // query rows per workgroup (as fa_fwd_kernel_wide.hpp)
constexpr int kBwdWideD = 256;                             // compiled head_dim
constexpr int kBwdWideLds = 2 /*K, V*/ * 2 /*stages*/ * kBN * kBwdWideD * 2;
#ifndef FA_BWD_WIDE_QT
#define FA_BWD_WIDE_QT 2
#endif
constexpr int kBwdWideThreads = 512 / FA_BWD_WIDE_QT;

// BwdParams as used here: x1 = Q, x2 = dO (stationary, H heads, S rows); y1 = K, y2 = V (streamed, H / G heads, Sy rows);
// out1 = P image, out2 = dS image (o?_sb / o?_sh / o?_ss: element strides of [batch][head][query row], o?_ss = ld >= Sy rounded
// up to 4); stats as everywhere ([2][B*H][Spad]: LSE log2(e), then -delta); nxb = 128-row query blocks per head.
// QT = 16-row query tiles per wave, as in fa_fwd_kernel_wide.hpp: 2 = 4 waves x 32 rows, one wave per SIMD (the default here);
// 1 = 8 waves x 16 rows, two waves per SIMD -- which the forward gains 17 - 28 % from and this kernel LOSES 5 - 20 % with
// ((4,16,4096,256): 3.82 / 4.08 ms causal / non-causal against 4.01 / 4.94): it is bound by its stores (two images of 8-byte pieces
// per lane), and halving the rows per wave doubles the fragment reads without adding anything the stores could hide behind.
template <class T, bool CAUSAL, int QT = FA_BWD_WIDE_QT>
__global__ __launch_bounds__(512 / QT, QT == 1 ? 2 : 1) void fa_bwd_wide_ds_kernel(const BwdParams p)
{
    static_assert(QT == 1 || QT == 2, "16 or 32 query rows per wave");
    constexpr int D = kBwdWideD;
    constexpr int NWAVES = 8 / QT;
    constexpr int KS = D / 32;
    constexpr int ROWB = D * 2;
    constexpr int TILE = kBN * ROWB;           // 32 KiB per K (or V) tile
    constexpr int PIECE = 1024;
    constexpr int CPT = TILE / PIECE / NWAVES;
    constexpr int VBASE = 2 * TILE;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_char*)smem;   // K stages [2][TILE], then V stages (both read by rows)

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int li = lane & 15, lg = lane >> 4;

    int head, tq;
    if (!wg_decode(blockIdx.x, p.bh, p.nxb, p.hsplit, head, tq)) return;
    const int b = head / p.H;
    const int h = head - b * p.H;
    const int S = p.S, Sk = p.Sy;
    const int coff = CAUSAL ? p.coff : 0;
    const int qb = CAUSAL ? p.nxb - 1 - tq : tq;          // causal: the longest query blocks start first

    using elem_t = unsigned short;
    const char* qh = reinterpret_cast<const char*>(p.x1) + (b * p.x1_sb + h * p.x1_sh) * 2;
    const char* gh = reinterpret_cast<const char*>(p.x2) + (b * p.x2_sb + h * p.x2_sh) * 2;
    const int hk = h / p.G;
    const char* kh = reinterpret_cast<const char*>(p.y1) + (b * p.y1_sb + hk * p.y1_sh) * 2;
    const char* vh = reinterpret_cast<const char*>(p.y2) + (b * p.y2_sb + hk * p.y2_sh) * 2;
    elem_t* ph = reinterpret_cast<elem_t*>(p.out1) + b * p.o1_sb + h * p.o1_sh;
    elem_t* dh = reinterpret_cast<elem_t*>(p.out2) + b * p.o2_sb + h * p.o2_sh;
    const int ld = (int)p.o1_ss;                           // row stride of both images (elements)

    const unsigned q_bytes = (unsigned)(((long long)(S - 1) * p.x1_ss + p.dv) * 2);
    const unsigned g_bytes = (unsigned)(((long long)(S - 1) * p.x2_ss + p.dv) * 2);
    const unsigned k_bytes = (unsigned)(((long long)(Sk - 1) * p.y1_ss + p.dv) * 2);
    const unsigned v_bytes = (unsigned)(((long long)(Sk - 1) * p.y2_ss + p.dv) * 2);
    __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(qh), 0, q_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(gh), 0, g_bytes, 0x00020000);
    const u32x4 rk_w = make_rsrc(kh, k_bytes);
    const u32x4 rv_w = make_rsrc(vh, v_bytes);

    const int q0w = qb * kBwdWideRows + wave * 16 * QT;
    const int kv_end_wg = CAUSAL ? max(0, min(Sk, qb * kBwdWideRows + kBwdWideRows + coff)) : Sk;
    const int nt = (kv_end_wg + kBN - 1) / kBN;                       // tiles the workgroup stages
    const int kv_end_w = (q0w >= S) ? 0 : (CAUSAL ? max(0, min(Sk, q0w + 16 * QT + coff)) : Sk);
    const int my_nt = (kv_end_w + kBN - 1) / kBN;                     // tiles this wave computes on
    const int ntk = (ld + kBN - 1) / kBN;                             // tiles of an image row

    // ---- Q and dO fragments (the forward's map): lane (li, lg) holds row q0w + 16 qt + li, columns 32 ks + 8 lg + 0..7
    u32x4 qf[QT][KS], gf[QT][KS];
    float lse2[QT], ndelta[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int qrow = q0w + 16 * qt + li;
        const unsigned qoff = (qrow < S) ? (unsigned)((long long)qrow * p.x1_ss * 2 + lg * 16) : 0x80000000u;
        const unsigned goff = (qrow < S) ? (unsigned)((long long)qrow * p.x2_ss * 2 + lg * 16) : 0x80000000u;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bool in = 32 * ks + 8 * lg < p.dv;
            qf[qt][ks] = __builtin_amdgcn_raw_buffer_load_b128(rq, in ? qoff + ks * 64 : 0x80000000u, 0, 0);
            gf[qt][ks] = __builtin_amdgcn_raw_buffer_load_b128(rg, in ? goff + ks * 64 : 0x80000000u, 0, 0);
        }
        // (rows past S: +inf / 0, as the pre-pass writes them for the padding rows: P = 0)
        const long long idx = (long long)head * p.Spad + min(qrow, p.Spad - 1);
        lse2[qt] = (qrow < S) ? p.stats[idx] : INFINITY;
        ndelta[qt] = (qrow < S) ? p.stats[(long long)p.bh * p.Spad + idx] : 0.f;
    }

    // ---- staging: K and V tiles alike, rows of 512 bytes under K's chunk swizzle (both are read by rows here)
    unsigned g_koff[CPT], g_voff[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int byte = (wave * CPT + i) * PIECE + lane * 16;
        const int row = byte / ROWB, chp = (byte % ROWB) / 16;
        const int kc = k_swz<128>(row, chp);
        g_koff[i] = (kc * 8 < p.dv) ? (unsigned)(row * p.y1_ss * 2 + kc * 16) : 0x80000000u;
        g_voff[i] = (kc * 8 < p.dv) ? (unsigned)(row * p.y2_ss * 2 + kc * 16) : 0x80000000u;
    }
    const unsigned k_tile_stride = (unsigned)(kBN * p.y1_ss * 2);
    const unsigned v_tile_stride = (unsigned)(kBN * p.y2_ss * 2);
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
    unsigned ka[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) ka[ks] = lds_base + li * ROWB + k_swz<128>(li, 4 * ks + lg) * 16;

    const float c = p.scale_log2;
    // lane (li, lg) owns, of a 64-key tile, the keys 16 kt + 4 lg + 0..3 of query rows q0w + 16 qt + li: two packed words per key
    // tile.  Key tiles are stored in pairs (kt, kt + 1) with the forward epilogue's lane exchange (v_permlane16_swap: odd lane groups
    // of the first tile's words <-> even groups of the second's), so that every lane stores 16 contiguous bytes -- even lg: keys
    // 16 kt + 4 lg .. + 7, odd lg: keys 16 (kt + 1) + 4 (lg - 1) .. + 7 -- and one instruction covers 64 contiguous bytes of each row
    // instead of 32 (with 8-byte stores the kernel spent most of its time on partial-line writes).
    const int key_in_pair = (lg & 1) ? (16 + 4 * (lg - 1)) : 4 * lg;           // + 16 kt (kt even) + the tile's first key
    auto store_pair = [&](elem_t* img, int qt, int key, unsigned a0, unsigned a1, unsigned b0, unsigned b1) {
        const auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
        const int qrow = q0w + 16 * qt + li;
        if (qrow < S && key < ld)                                              // (ld is a multiple of 8: a 16-byte piece is in or out whole)
            *reinterpret_cast<u32x4*>(img + (long long)qrow * ld + key) = u32x4{s0[0], s1[0], s0[1], s1[1]};
    };
    auto zero_tile = [&](int j) {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int kt = 0; kt < 4; kt += 2) {
                const int qrow = q0w + 16 * qt + li, key = j * kBN + 16 * kt + key_in_pair;
                if (qrow < S && key < ld) {
                    *reinterpret_cast<u32x4*>(ph + (long long)qrow * ld + key) = u32x4{0u, 0u, 0u, 0u};
                    *reinterpret_cast<u32x4*>(dh + (long long)qrow * ld + key) = u32x4{0u, 0u, 0u, 0u};
                }
            }
    };

    if (nt > 0) stage_tile(0);
    for (int j = 0; j < nt; ++j) {
        dma_wait<0>();
        __syncthreads();                       // tile j has landed and is visible; every wave is done with tile j - 1
        if (j + 1 < nt) stage_tile(j + 1);
        if (j >= my_nt) { zero_tile(j); continue; }        // (wave past its last tile: its rows see none of these keys)
        const unsigned st = (j & 1) * TILE;
        const int key0 = j * kBN;

        f32x4 s_acc[4][QT], d_acc[4][QT];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                s_acc[kt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
                d_acc[kt][qt] = f32x4{ndelta[qt], ndelta[qt], ndelta[qt], ndelta[qt]};
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const u32x4 kf = lds_read_b128(ka[ks] + st + kt * 16 * ROWB);
                const u32x4 vf = lds_read_b128(ka[ks] + VBASE + st + kt * 16 * ROWB);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    s_acc[kt][qt] = T::mfma16(kf, qf[qt][ks], s_acc[kt][qt]);
                    d_acc[kt][qt] = T::mfma16(vf, gf[qt][ks], d_acc[kt][qt]);
                }
            }
        }
        const bool need_mask = (key0 + kBN > Sk) || (CAUSAL && key0 + kBN - 1 > q0w + coff);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            const int qrow = q0w + 16 * qt + li;
            const int lim = (CAUSAL ? min(Sk - 1, qrow + coff) : Sk - 1) - key0 - 4 * lg;   // key 16 kt + e of this lane is kept iff 16 kt + e <= lim
#pragma unroll
            for (int kt = 0; kt < 4; kt += 2) {
                unsigned pw[2][2], dw[2][2];                                   // [tile of the pair][word]
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    float pv[4], dsv[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x = __builtin_amdgcn_exp2f(__builtin_fmaf(s_acc[kt + t][qt][e], c, -lse2[qt]));   // (a fully masked row has LSE = +inf here: 0)
                        if (need_mask && 16 * (kt + t) + e > lim) x = 0.f;
                        pv[e] = x;
                        dsv[e] = x * d_acc[kt + t][qt][e] * p.scale;
                    }
                    pw[t][0] = T::pack2(pv[0], pv[1]);   pw[t][1] = T::pack2(pv[2], pv[3]);
                    dw[t][0] = T::pack2(dsv[0], dsv[1]); dw[t][1] = T::pack2(dsv[2], dsv[3]);
                }
                const int key = key0 + 16 * kt + key_in_pair;
                store_pair(ph, qt, key, pw[0][0], pw[0][1], pw[1][0], pw[1][1]);
                store_pair(dh, qt, key, dw[0][0], dw[0][1], dw[1][0], dw[1][1]);
            }
        }
    }
    for (int j = nt; j < ntk; ++j) zero_tile(j);           // the tiles right of the block's diagonal: zeros
}

}  // namespace fa
