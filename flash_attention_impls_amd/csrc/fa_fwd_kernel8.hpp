// FlashAttention forward for gfx950 -- fp8 (OCP e4m3) Q, K, V with BOTH matrix products on MX-scaled fp8 MFMAs
// (v_mfma_scale_f32_16x16x128_f8f6f4, unit block scales: twice the 16-bit rate), head_dim 65..128, bf16 output.
//
// BASELINE.json config 5 ("fp8 Q/K/V (CDNA4 fp8 MFMA)").  Same workgroup shape, grid mapping, LDS-DMA ring, fixed softmax
// reference + exact fallback and epilogue as fa_fwd_kernel16.hpp (read that header and fa_fwd_kernel.hpp first).  What changes:
//
//   * a K/V tile is 128 keys x 128 bytes (fp8): the K = 128 instruction sums over the whole head_dim in the score product
//     and over 128 KEYS in the P V product, so a wave walks 128-key blocks, one workgroup barrier per block;
//   * S^T (16 keys x 16 queries) = K . Q^T, ONE MFMA per tile.  The instruction's k index is only summed over, so any
//     (lane group, byte) -> k map serves as long as both operands use the same one.  Here: lane (i, g) holds bytes
//     16 g .. 16 g + 15 and 64 + 16 g .. 64 + 16 g + 15 of its row (two ds_read_b128 of K, two 16-byte loads of Q);
//   * O^T (16 head_dim x 16 queries) += V^T[16 x 128 keys] . P^T[128 keys x 16].  B = P^T: byte 4 kt + e of lane (i, g) is
//     the S^T accumulator element e of key tile kt (key 16 kt + 4 g + e), converted to e4m3: again the accumulators in register
//     order ARE the operand.  A = V^T: byte 4 kt + e of lane (i', g) must be V[key 16 kt + 4 g + e][16 dt + i']: four
//     ds_read_b64_tr_b8 per head_dim tile, each gathering 8 key rows (key tiles 2 m and 2 m + 1) x 16 columns
//     (tools/probes/ds_read_tr8_probe.cpp: lane 2 q + p of a 16-lane group supplies the address of row q, lane i receives
//     column i of the 8 rows);
//   * P in e4m3: P = exp2(score - reference + b) with the reference fixed from the row's first 128-key block and b (0 .. 8) placing
//     e4m3's 2^-9 .. 448 window by how far that reference stands out of its own key tile (kGapFree8 below: ordinary rows get
//     b ~ 0.5 and typical P in 2^-7 .. 2^4, a row led by a dominant early key gets b = 8); a P beyond e4m3's
//     range converts to NaN, which poisons the row's sum on the matrix pipe and sends the
//     workgroup to the exact loop (running maximum, P <= 128).  O is normalised by the sum of the ROUNDED P (from a "ones"
//     head_dim tile on the matrix pipe); the LSE comes from the exact fp32 sum, formed only when an LSE is asked for.
//     The e4m3 rounding of P (3 mantissa bits) is the accuracy cost of this kernel: ~2.7 % relative Frobenius error
//     against the 5 % bound BASELINE.md states for fp8 inputs.
//   * software pipeline at key-tile granularity: region R of block n issues S(n) of key tile R, the softmax slice of key
//     tile R - 1, and the two P V MFMAs of head_dim tile R - 1 of block n - 1 (region 0: tile 7 of block n - 2): the
//     score accumulators live in two 8-register slots instead of a 64-register block.
#pragma once
#include "fa_fwd_kernel.hpp"

namespace fa {

constexpr int kBN8 = 128;                      // keys per tile (= per block) of the fp8 kernel
// Where a row's P sit in e4m3 (2^-9 .. 448, full precision from 2^-6 up).  The softmax reference of a row is the maximum mx of
// its first 128 keys (round 2: 16) and is shown to e4m3 as P(mx) = 2^b, b = clamp(gap - kGapFree8, 0, kPRefTop8) with gap = mx -
// (mean of those scores), in binades (the numbers that follow were measured with the 16-key sample): the top of the window stands 8.86 + kGapFree8 binades above the MEAN of the first key tile (7.5 sigma on
// N(0,1) data; measured on config 5's shard: kGapFree8 = 1 costs 2.6 % in workgroups sent to the exact loop, 2 costs 0.4 %, and
// a max-anchored b = 3 sends 14 % of the workgroups there: -25 %), but never lower than just above mx itself -- a dominant early key (an "attention sink") then sits at the top of e4m3
// and the many small terms under it, which together can weigh as much as the sink, stay in full precision down to 14.8 binades
// below it.  A score above the window converts to NaN and sends the workgroup to the exact loop; a term 9 + b binades below the
// reference is rounded to zero.  (The usual running-maximum fp8 softmax, P <= 1, drops terms 10 binades below the row maximum.)
#ifndef FA8_GAP_FREE
#define FA8_GAP_FREE 2.0f
#endif
constexpr float kGapFree8 = FA8_GAP_FREE;
#ifndef FA8_HALF_PRIO
#define FA8_HALF_PRIO 1
#endif
// FA8_EARLY_EPILOGUE / FA8_CONT_RING: the pass end of fa_fwd_kernel16.hpp (a wave stores its output as soon as it has finished its
// blocks, ONE barrier with per-wave verdict flags instead of three; the staging slots of a causal pair's first pass that would fetch
// tiles past its diagonal fetch the second pass's first tiles instead, so that it starts without a prologue and without a wait)
#ifndef FA8_EARLY_EPILOGUE
#define FA8_EARLY_EPILOGUE 1
#endif
#ifndef FA8_CONT_RING
#define FA8_CONT_RING 1
#endif
#ifndef FA8_ODD_UNMASKED
#define FA8_ODD_UNMASKED 1
#endif
constexpr float kPRefTop8 = 8.0f;
constexpr float kGapFloor8 = 12.0f;            // a score further than this below mx counts as mx - 12 in the mean (one very low key must not move the window)
constexpr float kSumFloor8 = 0.92f;            // WANT_LSE: rounded row sum below this share of the exact one -> exact fallback (underflowed weight)
constexpr float kPLimit8 = 1e30f;              // a rounded row sum not below this (i.e. NaN: some P left e4m3's range) -> exact fallback
// !WANT_LSE (no exact sums to compare with): a SAMPLED bound on the weight e4m3 cannot hold.  Behind a row's first 128-key block
// (whose keys the window was placed by) one score in 32 -- one element of one key tile per block, another one in odd blocks: 4 of
// a row's 128 keys -- adds min(P, 2^-9) to a per-lane sum u; 2^-9 is e4m3's smallest subnormal: everything below it is rounded to
// 0 or up to it.  32 u estimates U = sum over the keys of min(P, 2^-9) >= the weight below the window.  Ordinary rows have
// U <= 0.5 % of the row sum L (2^-9 over the mean P, which the window places at 2^-2 ... 2^1); a row whose tail sits under the
// window has U ~ the tail's whole weight.  32 u > L / 64 sends the workgroup to the exact loop, as the rounded-vs-exact
// comparison does where the exact sums exist.  Why so sparse a sample suffices: a full first block makes L >= 128 x 2^-2 (the mean
// score sits 2 binades under the top of the window or higher), so a tail that carries 5 % of L under the window is > 1600 keys:
// > 50 samples.  Cost: 4 of ~190 vector instructions per wave and block (measured: -0.6 %; one score in eight: -2.2 %).
#ifndef FA8_SAMPLED_CHECK
#define FA8_SAMPLED_CHECK_ON 1
#else
#define FA8_SAMPLED_CHECK_ON FA8_SAMPLED_CHECK
#endif
constexpr float kUnderCap8 = 0.001953125f;     // 2^-9
constexpr float kSampleGain8 = 32.0f * 64.0f;  // sampling rate x inverse trigger share

// 16-byte-chunk swizzles of the 128-byte rows (two rows per 256-byte bank row)
//  K, read by rows (ds_read_b128, chunks g and g + 4 of row i): chunk ^ ((row >> 1) & 7)
//  V, read transposed (8-byte pieces of keys 4 g + e (+ 16) per 16-lane group): chunk ^ (((key >> 1) & 3) | (((key >> 4) & 1) << 2))
__device__ __forceinline__ int k8_swz(int row, int ch) { return ch ^ ((row >> 1) & 7); }
__device__ __forceinline__ int v8_swz(int row, int ch) { return ch ^ (((row >> 1) & 3) | (((row >> 4) & 1) << 2)); }

__device__ __forceinline__ f32x4 mfma8(i32x8 a, i32x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
}
typedef __attribute__((ext_vector_type(2))) int i32x2;
__device__ __forceinline__ i32x2 lds_read_tr8_b64(unsigned addr) {
    return __builtin_amdgcn_ds_read_tr8_b64_v2i32(reinterpret_cast<__attribute__((address_space(3))) i32x2*>(addr));
}
// two fp32 -> two e4m3 bytes in the low (hi = false) or high half of `old`
__device__ __forceinline__ int pack_fp8(float a, float b, int old, bool hi) {
    return hi ? __builtin_amdgcn_cvt_pk_fp8_f32(a, b, old, true) : __builtin_amdgcn_cvt_pk_fp8_f32(a, b, old, false);
}

// WANT_LSE = false (the caller passed no LSE buffer): the exact fp32 row sums -- one v_add_f32 per score, a quarter of the
// kernel's vector instructions -- are not formed at all; the output only needs the rounded sums, which the matrix pipe delivers.
template <bool CAUSAL, bool WANT_LSE>
__global__ __launch_bounds__(512, 2) void fa_fwd_kernel8(const FwdParams p)
{
    constexpr int NWAVES = 8;
    constexpr int ROWB = 128;                  // bytes per K / V row in LDS
    constexpr int TILE = kBN8 * ROWB;          // 16 KiB per K (or V) tile
    constexpr int PIECE = 1024;
    constexpr int CPT = TILE / PIECE / NWAVES; // DMA pieces per wave and tile (2)
    constexpr int DT = 8;                      // 16-wide head_dim tiles
    constexpr int VBASE = kStages * TILE;
    typedef TypeBF16 T;                        // output type

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_char*)smem;   // K ring [kStages][TILE], then V ring
    __shared__ __attribute__((aligned(16))) unsigned fa_flags8[2][8];          // per-wave fallback verdicts, by pass parity (outside the ring)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#if defined(FA_STAMP)
    // diagnostic build only (tools/stamps.py --dtype fp8): cycle sums per wave, as in fa_fwd_kernel16.hpp -- (a) a steady-state
    // 128-key block split at its barrier: regions 0..3 | the counted wait + barrier + the four staging DMA pieces | regions 4..7;
    // (b) the phases of a pass.  g_fa_stamp and stamp_now() are those of fa_fwd_kernel16.hpp.
    unsigned long long st_e = 0, st_w = 0, st_o = 0, st_n = 0;
    const unsigned long long st_t0 = stamp_now();
    unsigned long long st_rt0;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_rt0) :: "memory");
    unsigned long long st_ph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_last = st_t0, st_a = 0, st_b = 0;
#define FA8_PHASE(K) do { const unsigned long long t_ = stamp_now(); st_ph[K] += t_ - st_last; st_last = t_; } while (0)
#else
#define FA8_PHASE(K) do {} while (0)
#endif

    const bool paired = CAUSAL && !p.unpaired;
    const int wg_per_head = paired ? (p.nqb + 1) / 2 : p.nqb;
    int head, tq;
    if (!wg_decode(blockIdx.x, p.bh, wg_per_head, p.hsplit, head, tq)) return;
    const int b = head / p.H;
    const int h = head - b * p.H;
    const int S = p.S, Sk = p.Sk;
    const int coff = CAUSAL ? Sk - S : 0;
    const int n_pass = (paired && (p.nqb - 1 - tq != tq)) ? 2 : 1;

    using elem_t = unsigned short;
    const int hk = h / p.G;
    const char* qh = reinterpret_cast<const char*>(p.q) + b * p.q_sb + h * p.q_sh;          // fp8: element strides = byte strides
    const char* kh = reinterpret_cast<const char*>(p.k) + b * p.k_sb + hk * p.k_sh;
    const char* vh = reinterpret_cast<const char*>(p.v) + b * p.v_sb + hk * p.v_sh;
    elem_t* oh = reinterpret_cast<elem_t*>(p.o) + b * p.o_sb + h * p.o_sh;
    const unsigned q_bytes = (unsigned)((long long)(S - 1) * p.q_ss + p.dv);
    const unsigned k_bytes = (unsigned)((long long)(Sk - 1) * p.k_ss + p.dv);
    const unsigned v_bytes = (unsigned)((long long)(Sk - 1) * p.v_ss + p.dv);
    __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(qh), 0, q_bytes, 0x00020000);
    const u32x4 rk_w = make_rsrc(kh, k_bytes);
    const u32x4 rv_w = make_rsrc(vh, v_bytes);

    i32x8 qf[2];               // Q fragments [query tile]: 32 bytes per lane (chunks g and g + 4 of the row)
    bool staged = false;       // the previous pass left this pass's first tiles in the ring, landed and published (cont)
    int stage0 = 0;            // ring stage of this pass's tile 0 (a staged pass continues where the ring stood)

  for (int pass = 0; pass < n_pass; ++pass) {
    const int qb = CAUSAL ? ((pass == 0) ? p.nqb - 1 - tq : tq) : tq;
    const int lane = lane_here();
    const int li = lane & 15;
    const int lg = lane >> 4;
    const int rowblk_of_wave = CAUSAL ? (wave < 4 ? wave : 11 - wave) : wave;
    const int q0w = qb * kBM + rowblk_of_wave * 32;

    const int kv_end_wg = CAUSAL ? max(0, min(Sk, qb * kBM + kBM + coff)) : Sk;
    const int nt = (kv_end_wg + kBN8 - 1) / kBN8;                     // 128-key tiles the workgroup stages
    const int kv_end_w = (q0w >= S) ? 0 : (CAUSAL ? max(0, min(Sk, q0w + 32 + coff)) : Sk);
    const int my_nt = (kv_end_w + kBN8 - 1) / kBN8;                   // tiles this wave computes on
    // Continuous ring (as fa_fwd_kernel16.hpp): the staging slots of the last three iterations fetch K(0..2), V(0..1) of the NEXT
    // pass -- the same head -- instead of tiles past this pass's diagonal; the next pass's tile 0 then sits at ring stage
    // (stage0 + nt) & 3 (128-key tiles: nt = 2 (qb + 1), so 0 or 2)
    const bool cont = FA8_CONT_RING != 0 && FA8_EARLY_EPILOGUE != 0 && CAUSAL && (pass + 1 < n_pass) && nt >= kStages;
    const int ntw = cont ? nt : 0x3fffffff;                           // tile index wrap of the staging DMAs
    auto tk = [&](int t) { return t >= ntw ? t - ntw : t; };

    auto load_q = [&](int qblk, int lane_q) {
        const int li_ = lane_q & 15, lg_ = lane_q >> 4;
        const int row0 = qblk * kBM + rowblk_of_wave * 32;
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const int qrow = row0 + 16 * qt + li_;
            const unsigned qoff = (qrow < S) ? (unsigned)((long long)qrow * p.q_ss) : 0x80000000u;   // rows past the end read as zero
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int ch = lg_ + 4 * hh;
                const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rq, (ch * 16 < p.dv) ? qoff + ch * 16 : 0x80000000u, 0, 0);
                qf[qt][4 * hh] = (int)w[0]; qf[qt][4 * hh + 1] = (int)w[1]; qf[qt][4 * hh + 2] = (int)w[2]; qf[qt][4 * hh + 3] = (int)w[3];
            }
        }
    };
    FA8_PHASE(8);               // (diagnostic) entry: parameters, workgroup decode, descriptors
    if (pass == 0) load_q(qb, lane);
    FA8_PHASE(9);               // (diagnostic) Q loads issued

    // ---- K/V staging by LDS-DMA (1-KiB pieces = 8 rows; swizzle on the per-lane source address)
    unsigned g_koff[CPT], g_voff[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int byte = (wave * CPT + i) * PIECE + lane * 16;
        const int row = byte / ROWB, chp = (byte % ROWB) / 16;
        const int ck = k8_swz(row, chp), cv = v8_swz(row, chp);
        g_koff[i] = (ck * 16 < p.dv) ? (unsigned)(row * p.k_ss + ck * 16) : 0x80000000u;
        g_voff[i] = (cv * 16 < p.dv) ? (unsigned)(row * p.v_ss + cv * 16) : 0x80000000u;
    }
    const unsigned k_tile_stride = (unsigned)(kBN8 * p.k_ss);
    const unsigned v_tile_stride = (unsigned)(kBN8 * p.v_ss);
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

    // ---- LDS read addresses; they carry the ring stage of the tile being read (K: tile j, V: tile j - 1)
    unsigned ka[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) ka[hh] = lds_base + stage0 * TILE + li * ROWB + (k8_swz(li, lg + 4 * hh) << 4);
    unsigned va[DT];
    {
        const int qq = li >> 1, pp = li & 1;
        const int key_in = 16 * (qq >> 2) + 4 * lg + (qq & 3);        // + 32 m: swizzle-neutral
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) va[dt] = lds_base + VBASE + stage0 * TILE + key_in * ROWB + (v8_swz(key_in, dt) << 4) + 8 * pp;
    }
    constexpr int NFB = 2;     // K and V^T fragment double buffers (by region parity): read two regions ahead of their MFMAs
    i32x8 kf[NFB], vf[NFB];
    auto read_k = [&] __device__ (auto kt_c, auto buf_c) {
        constexpr int kt = decltype(kt_c)::value, buf = decltype(buf_c)::value;
        const u32x4 w0 = lds_read_b128(ka[0] + kt * 16 * ROWB), w1 = lds_read_b128(ka[1] + kt * 16 * ROWB);
        kf[buf % NFB] = i32x8{(int)w0[0], (int)w0[1], (int)w0[2], (int)w0[3], (int)w1[0], (int)w1[1], (int)w1[2], (int)w1[3]};
    };
    auto read_v = [&] __device__ (auto dt_c, auto buf_c) {
        constexpr int dt = decltype(dt_c)::value, buf = decltype(buf_c)::value;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const i32x2 w = lds_read_tr8_b64(va[dt] + m * 32 * ROWB);
            vf[buf % NFB][2 * m] = w[0];
            vf[buf % NFB][2 * m + 1] = w[1];
        }
    };

    f32x4 o_acc[DT][2];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) o_acc[dt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // row sums of the ROUNDED P (what the P V product really weighs V with), from the matrix pipe: one more "head_dim tile"
    // whose V^T fragment is all ones.  Every lane (i, g) then holds the full sum of query i in all four elements: no lane
    // exchange, and O = sum(P~ V) / sum(P~) is a properly normalised average (V = 1 gives O = 1 exactly; a row with one
    // dominant key returns that key's V).  The LSE uses the exact fp32 sums below.
    f32x4 l_acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    const i32x8 ones8 = {0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838};   // e4m3 1.0
    float m_c[2] = {-INFINITY, -INFINITY};
    float l_a[2] = {0.f, 0.f}, l_b[2] = {0.f, 0.f};    // this lane's share of the exact row sums (LSE): two add chains per query tile
    float u_s[2] = {0.f, 0.f};                         // !WANT_LSE: this lane's sampled sum of min(P, 2^-9) (kUnderCap8 above)
    const float c = p.scale_log2;

    // mask: key (relative to the lane's first key 4 g of a tile) is dead iff it exceeds limq - (first key of the tile); the
    // per-element compares then run against immediates (no per-(tile, element) index vectors kept alive across the loops)
    int limq[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int qrow = q0w + 16 * qt + li;
        limq[qt] = (CAUSAL ? min(Sk - 1, qrow + coff) : Sk - 1) - 4 * lg;
    }

    f32x4 s_acc[2][2];         // S^T tiles [slot = key tile parity][query tile]
    i32x8 pf[2][2];            // P^T fragments [block parity][query tile], dword kt

    auto sm_slice = [&] __device__ (auto mask_c, auto par_c, auto kt_c, auto samp_c, int key0) {
        constexpr bool MASK = decltype(mask_c)::value, SAMP = decltype(samp_c)::value;
        constexpr int par = decltype(par_c)::value, kt = decltype(kt_c)::value;
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            f32x4 sv = s_acc[kt & 1][qt];
            if constexpr (MASK) {
                const int lim = limq[qt] - key0;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (16 * kt + e > lim) sv[e] = -INFINITY;
            }
            const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sv[0], c, -m_c[qt]));
            const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sv[1], c, -m_c[qt]));
            const float p2 = __builtin_amdgcn_exp2f(__builtin_fmaf(sv[2], c, -m_c[qt]));
            const float p3 = __builtin_amdgcn_exp2f(__builtin_fmaf(sv[3], c, -m_c[qt]));
            if constexpr (WANT_LSE) { l_a[qt] += p0; l_b[qt] += p1; l_a[qt] += p2; l_b[qt] += p3; }
            else if constexpr (FA8_SAMPLED_CHECK_ON != 0 && SAMP && kt == (par == 0 ? 2 : 5)) {
                u_s[qt] += __builtin_fminf(par == 0 ? p1 : p2, kUnderCap8);      // (with the buffer parity: another element of another key tile)
            }
            // (both halves of the word are overwritten: its old content serves as the conversions' pass-through operand,
            // which saves the v_mov a fresh zero would cost.)  A P beyond e4m3's range (> 464) converts to NaN.
            pf[par][qt][kt] = pack_fp8(p2, p3, pack_fp8(p0, p1, pf[par][qt][kt], false), true);
        }
    };
    // The row's FIRST 128-KEY BLOCK fixes its softmax reference and the place of its e4m3 window (kGapFree8 above): row maximum,
    // clamped sum and count of its 128 scores.  (Until round 3 the first 16 keys alone decided: a row whose first 16 keys were ALL
    // comparably dominant, with a broad tail 7 nats below them, got b = 0 and lost the tail -- a fifth of its weight at S = 4096 --
    // to e4m3's underflow, silently.  With 128 keys the tail is in the sample and opens the window downwards; a row whose first 128
    // keys are all dominant is caught at the pass end: the rounded against the exact row sums with an LSE, the sampled bound without.)
    // Two forms.  The common one (`first_block_folded`, a block no row of the wave needs the mask in) forms the block's sixteen score
    // tiles ONCE, into registers the O^T accumulators do not need yet, takes the statistics from them and then the eight softmax
    // slices: it is block 0 of the pipeline as well.  The masked one (the first 128 rows of a causal head, a key length under 128)
    // keeps the pre-pass of rounds 3 - 4: the tiles are formed for the statistics alone and once more by the pipeline's fill block.
    auto place_window = [&] __device__ (const float (&mx)[2], const float (&sm)[2], const float (&sm2)[2], const float (&cn)[2]) {
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            float m = mx[qt], u = sm[qt] + sm2[qt], n = cn[qt];
            m = fmaxf(m, __shfl_xor(m, 16));   u += __shfl_xor(u, 16);   n += __shfl_xor(n, 16);
            m = fmaxf(m, __shfl_xor(m, 32));   u += __shfl_xor(u, 32);   n += __shfl_xor(n, 32);
            const float gap = (m - u * __builtin_amdgcn_rcpf(fmaxf(n, 1.f))) * c;               // 0 .. 12 binades
            const float b = fminf(fmaxf(gap - kGapFree8, 0.f), kPRefTop8);                      // (a NaN gap gives 0)
            m_c[qt] = (m == -INFINITY) ? 0.f : __builtin_fmaf(m, c, -b);
        }
    };
    auto set_reference_from_first_block = [&] __device__ (bool masked) {
        float mx[2] = {-INFINITY, -INFINITY}, sm[2] = {0.f, 0.f}, sm2[2] = {0.f, 0.f}, cn[2] = {0.f, 0.f};
        const float span = kGapFloor8 * __builtin_amdgcn_rcpf(c);
        auto tile_stats = [&] __device__ (auto mask_c, int kt) {
            constexpr bool MASK = decltype(mask_c)::value;
            const u32x4 w0 = lds_read_b128(ka[0] + kt * 16 * ROWB), w1 = lds_read_b128(ka[1] + kt * 16 * ROWB);
            const i32x8 kx = {(int)w0[0], (int)w0[1], (int)w0[2], (int)w0[3], (int)w1[0], (int)w1[1], (int)w1[2], (int)w1[3]};
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                const f32x4 sx = mfma8(kx, qf[qt], f32x4{0.f, 0.f, 0.f, 0.f});
                if constexpr (MASK) {
                    const int lim = limq[qt] - 16 * kt;                   // (keys past the causal limit / the last key: not counted)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bool in = (e <= lim);
                        mx[qt] = in ? fmaxf(mx[qt], sx[e]) : mx[qt];
                        // a score far below the (running) maximum counts as maximum - 12 binades: one very low key must not move the window
                        sm[qt] += in ? fmaxf(sx[e], mx[qt] - span) : 0.f;
                        cn[qt] += in ? 1.f : 0.f;
                    }
                } else {
                    // the common form, kept short (96 vector instructions per wave and pass): maximum by v_max3, plain sum (a very
                    // low score only widens the gap, i.e. opens the window further down: the safe side)
                    mx[qt] = fmaxf(fmaxf(mx[qt], sx[0]), sx[1]);
                    mx[qt] = fmaxf(fmaxf(mx[qt], sx[2]), sx[3]);
                    sm[qt] += sx[0]; sm2[qt] += sx[1]; sm[qt] += sx[2]; sm2[qt] += sx[3];
                }
            }
        };
        if (masked) {                      // (the first 128 rows of a causal head, a key length under 128: rare -- a rolled loop)
#pragma unroll 1
            for (int kt = 0; kt < 8; ++kt) tile_stats(std::true_type{}, kt);
        } else {
#pragma unroll
            for (int kt = 0; kt < 8; ++kt) tile_stats(std::false_type{}, kt);
            cn[0] = cn[1] = 32.f;                                         // (8 tiles x 4 keys per lane)
        }
        place_window(mx, sm, sm2, cn);
    };

    // ---- ring protocol (as fa_fwd_kernel.hpp, with 128-key tiles): iteration j reads K(j) (and K(j+1)'s first fragment in its
    // last region) and V(j-1); barrier(j) sits in the middle of the iteration, behind it K(j+3) and V(j+2) are issued.
    auto issue_prologue = [&]() {
        dma_k(0, 0);
        dma_v(0, 0);
        dma_k(1, TILE);
        dma_k(2, 2 * TILE);
        dma_v(1, TILE);
    };
    FA8_PHASE(10);              // (diagnostic) offsets, addresses, accumulator init
    if (pass == 0) issue_prologue();
    FA8_PHASE(11);              // (diagnostic) prologue DMAs issued
    if (!staged) {
        // this wave's pieces of K(0) (and, older, its Q fragments) have landed; V(0) may still be in flight: its first reader is the
        // fragment read in region 7 of block 0, behind that block's own wait and barrier, which cover it
        dma_wait<4 * CPT>();
        __syncthreads();        // ... and every wave's are visible
    }
    FA8_PHASE(0);               // pass start -> first tiles visible

    int stage_k = stage0;                          // ring stage of tile j
    auto sync_and_stage = [&](int j) {
#if !defined(FA8_ABL_NOBARRIER)                    // (timing-only ablation builds: wrong results)
        dma_wait<2 * CPT>();                       // everything but the previous iteration's DMA has landed ...
        __syncthreads();                           // ... and is published; the stages refilled below are no longer read
#endif
#if !defined(FA8_ABL_NODMA)
        dma_k(tk(j + 3), ((stage_k + 3) & (kStages - 1)) * TILE);
        dma_v(tk(j + 2), ((stage_k + 2) & (kStages - 1)) * TILE);
#endif
    };
    auto advance_k = [&]() {                       // K(j) -> K(j+1)
        const int d = (stage_k == kStages - 1) ? -(kStages - 1) * TILE : TILE;
        ka[0] += d; ka[1] += d;
    };
    int stage_v = stage0;                          // ring stage the V^T read addresses point at (block 1 reads V(0))
    auto advance_v = [&]() {                       // V(t) -> V(t+1): in region 7 of every block but the first
        const int d = (stage_v == kStages - 1) ? -(kStages - 1) * TILE : TILE;
        stage_v = (stage_v + 1) & (kStages - 1);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) va[dt] += d;
    };

    // One block n (= iteration j = n), eight regions R.  PAR = the P^T buffer the block's own slices fill:
    //   DO_S : S(n), key tile R                       SL : softmax slice of key tile R - 1 (R >= 1) of block n -> pf[PAR]
    //   SL0  : region 0, slice of key tile 7 of block n - 1 -> pf[PAR ^ 1]
    //   PV   : region 0: P V of head_dim tile 7 of block n - 2 (P in pf[PAR]); regions 1..7: tile R - 1 of block n - 1 (pf[PAR ^ 1])
    //   LDS  : region R reads the K fragment of key tile R + 2 (regions 6, 7: tiles 0, 1 of block n + 1) and the V^T fragment
    //          that region R + 2 multiplies (regions 0..6: tile n - 1; region 7: the first fragment of tile n)
    //   sync : barrier(n) and the staging DMAs behind region 3 (every wave runs it for every tile of the workgroup)
    // Only two forms run more than a few times per pass: the unmasked pair (PAR 0 then PAR 1) of the steady state.  Every
    // other block (the second, an odd one out, diagonal / ragged blocks, the drain) runs the masked PAR = 0 form followed
    // by a swap of the two P^T buffers, so that block parity never has to be known at run time.  Blocks that have no
    // predecessor to finish multiply zeroed fragments (pf, vf start at zero), which keeps the forms few.
    // Fragments are read TWO regions ahead of their MFMAs (region R reads key tile R + 2 and the V^T fragment of region
    // R + 2's product into the buffer region R has just consumed): an LDS read has a whole region to land instead of a few instructions.
    // VRD = the V^T reads of regions 0..6 (tile n - 1), VRD7 = that of region 7 (first fragment of tile n, behind the address advance).
#if FA8_HALF_PRIO
    const int late_half = (wave >= 4) ? 1 : 0;
#endif
    auto block8 = [&] __device__ (auto par_c, auto do_s_c, auto sl0_c, auto sl_c, auto pv_c, auto mask_c, auto first_c, auto vrd_c, auto vrd7_c, bool do_sync, int n) {
        constexpr int PAR = decltype(par_c)::value;
        constexpr bool VRD = decltype(vrd_c)::value, VRD7 = decltype(vrd7_c)::value;
        constexpr bool DO_S = decltype(do_s_c)::value, SL0 = decltype(sl0_c)::value, SL = decltype(sl_c)::value;
        constexpr bool PV = decltype(pv_c)::value, FIRST = decltype(first_c)::value;
        const int key0 = n * kBN8;
        auto region = [&] __device__ (auto r_c) {
            constexpr int R = decltype(r_c)::value;
            __builtin_amdgcn_sched_barrier(0);
#if FA8_HALF_PRIO
            // behind the block's barrier (end of region 3) both waves of a SIMD start regions 4..7 together and the older one
            // wins every issue arbitration: it runs them in 1180 cycles, the younger in 2460, and then waits 1700 cycles at the
            // next barrier (profiles/r3_stamps_cfg5.txt).  Priority for the younger half in exactly those regions evens it out.
            if constexpr (R == 4) setprio_if<1>(late_half);
            if constexpr (R == 0) setprio_if<0>(late_half);
#endif
            // the score MFMAs go first: their results are the next region's first VALU operands
            if constexpr (DO_S) {
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) s_acc[R & 1][qt] = mfma8(kf[(R & 1) % NFB], qf[qt], f32x4{0.f, 0.f, 0.f, 0.f});
            }
            if constexpr (PV) {
                constexpr int dt = (R == 0) ? 7 : R - 1;
                constexpr int par = (R == 0) ? PAR : (PAR ^ 1);
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) o_acc[dt][qt] = mfma8(vf[(R & 1) % NFB], pf[par][qt], o_acc[dt][qt]);
                if constexpr (R == 4) {        // the ones tile of block n - 1
#pragma unroll
                    for (int qt = 0; qt < 2; ++qt) l_acc[qt] = mfma8(ones8, pf[PAR ^ 1][qt], l_acc[qt]);
                }
            }
            if constexpr (DO_S) {
                if constexpr (R == 6) advance_k();
                read_k(IC<(R + 2) & 7>{}, IC<R & 1>{});
            }
            if constexpr (R == 7 ? VRD7 : VRD) {
                if constexpr (R == 7 && VRD) advance_v();         // (the fill block has no predecessor: its addresses already point at tile 0)
                read_v(IC<(R + 1) & 7>{}, IC<R & 1>{});
            }
            if constexpr (R == 0) {
                if constexpr (SL0) sm_slice(mask_c, IC<PAR ^ 1>{}, IC<7>{}, std::false_type{}, key0 - kBN8);
            } else if constexpr (SL) {
                sm_slice(mask_c, IC<PAR>{}, IC<R - 1>{}, std::integral_constant<bool, !FIRST>{}, key0);      // (the first block is not sampled)
            }
#if !defined(FA8_NO_SGB)
            if constexpr (DO_S && PV && SL && SL0) {       // steady form: 4 x {1 MFMA, DS reads, 5 VALU}
#define FA8_GRP(NDS) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, NDS, 0); __builtin_amdgcn_sched_group_barrier(0x402, 5, 0)
                FA8_GRP(1); FA8_GRP(1); FA8_GRP(2); FA8_GRP(2);
#undef FA8_GRP
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (R == 3) {
#if defined(FA_STAMP)
                st_a = stamp_now();
#endif
                if (do_sync) sync_and_stage(n);
#if defined(FA_STAMP)
                st_b = stamp_now();
#endif
            }
        };
        region(IC<0>{}); region(IC<1>{}); region(IC<2>{}); region(IC<3>{});
        region(IC<4>{}); region(IC<5>{}); region(IC<6>{}); region(IC<7>{});
    };
    // Block 0 where no row of the wave needs the mask: what set_reference_from_first_block and the fill block do between them, with
    // the sixteen score tiles formed once.  Leaves the pipeline exactly where the fill block leaves it: P of key tiles 0..6 in
    // pf[0], the scores of key tile 7 in slot 1 (block 1's region 0 slices them), the first two K fragments of block 1 and the
    // first V^T fragment of block 0 read, barrier(0) passed and K(3), V(2) requested.
    auto first_block_folded = [&] __device__ () {
        f32x4 s0[8][2];
#pragma unroll
        for (int kt = 0; kt < 8; ++kt) {
            const u32x4 w0 = lds_read_b128(ka[0] + kt * 16 * ROWB), w1 = lds_read_b128(ka[1] + kt * 16 * ROWB);
            const i32x8 kx = {(int)w0[0], (int)w0[1], (int)w0[2], (int)w0[3], (int)w1[0], (int)w1[1], (int)w1[2], (int)w1[3]};
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) s0[kt][qt] = mfma8(kx, qf[qt], f32x4{0.f, 0.f, 0.f, 0.f});
        }
        float mx[2] = {-INFINITY, -INFINITY}, sm[2] = {0.f, 0.f}, sm2[2] = {0.f, 0.f}, cn[2] = {32.f, 32.f};   // (8 tiles x 4 keys per lane)
#pragma unroll
        for (int kt = 0; kt < 8; ++kt)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {                              // (the unmasked statistics of set_reference_from_first_block, same order)
                const f32x4 sx = s0[kt][qt];
                mx[qt] = fmaxf(fmaxf(mx[qt], sx[0]), sx[1]);
                mx[qt] = fmaxf(fmaxf(mx[qt], sx[2]), sx[3]);
                sm[qt] += sx[0]; sm2[qt] += sx[1]; sm[qt] += sx[2]; sm2[qt] += sx[3];
            }
        place_window(mx, sm, sm2, cn);
        sync_and_stage(0);
        auto slice_of = [&] __device__ (auto kt_c) {
            constexpr int kt = decltype(kt_c)::value;
            s_acc[kt & 1][0] = s0[kt][0];
            s_acc[kt & 1][1] = s0[kt][1];
            sm_slice(std::false_type{}, IC<0>{}, kt_c, std::false_type{}, 0);      // (the first block is not sampled)
        };
        slice_of(IC<0>{}); slice_of(IC<1>{}); slice_of(IC<2>{}); slice_of(IC<3>{});
        slice_of(IC<4>{}); slice_of(IC<5>{}); slice_of(IC<6>{});
        s_acc[1][0] = s0[7][0];
        s_acc[1][1] = s0[7][1];
        advance_k();
        read_k(IC<0>{}, IC<0>{});
        read_k(IC<1>{}, IC<1>{});
        read_v(IC<0>{}, IC<1>{});
    };
    auto swap_pf = [&]() {
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) { const i32x8 t = pf[0][qt]; pf[0][qt] = pf[1][qt]; pf[1][qt] = t; }
    };
    typedef std::true_type Y;
    typedef std::false_type N_;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) pf[0][qt] = pf[1][qt] = i32x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < NFB; ++i) vf[i] = i32x8{0, 0, 0, 0, 0, 0, 0, 0};

    const int NT = my_nt;
    // first block whose softmax needs the mask (the wave's diagonal block, or the ragged last one)
    const int mb = min(CAUSAL ? (max(0, q0w + coff) >> 7) : 0x7fffffff, Sk >> 7);
    int j = 0;
    if (NT > 0) {
        if (mb > 0) {           // block 0 needs no mask: scores once, then statistics and slices from them
            first_block_folded();
            FA8_PHASE(1);
        } else {
            set_reference_from_first_block(true);
            FA8_PHASE(1);       // reference pre-pass over the first block
            read_k(IC<0>{}, IC<0>{});
            read_k(IC<1>{}, IC<1>{});
            // block 0 (pipeline fill): scores and slices only (the reference is fixed: pre-pass above)
            block8(IC<0>{}, Y{}, N_{}, Y{}, N_{}, Y{}, Y{}, N_{}, Y{}, true, 0);
        }
        swap_pf();
        stage_k = (stage_k + 1) & (kStages - 1);
        j = 1;
        FA8_PHASE(2);           // fill block
        const int ja = min(mb, NT);
        for (; j + 1 < ja; j += 2) {                   // steady state: pairs of unmasked blocks
#if defined(FA_STAMP)
            const unsigned long long t0 = stamp_now();
#endif
            block8(IC<0>{}, Y{}, Y{}, Y{}, Y{}, N_{}, N_{}, Y{}, Y{}, true, j);
            stage_k = (stage_k + 1) & (kStages - 1);
#if defined(FA_STAMP)
            const unsigned long long t1 = stamp_now();
            st_e += st_a - t0; st_w += st_b - st_a; st_o += t1 - st_b;
#endif
            block8(IC<1>{}, Y{}, Y{}, Y{}, Y{}, N_{}, N_{}, Y{}, Y{}, true, j + 1);
            stage_k = (stage_k + 1) & (kStages - 1);
#if defined(FA_STAMP)
            const unsigned long long t2 = stamp_now();
            st_e += st_a - t1; st_w += st_b - st_a; st_o += t2 - st_b; st_n += 2;
#endif
        }
#if defined(FA_STAMP)
        st_last = stamp_now();  // (the steady loop's cycles are kept in the segment sums)
#endif
        if (FA8_ODD_UNMASKED != 0 && j < ja) {         // an odd unmasked block out: single form + buffer swap
            block8(IC<0>{}, Y{}, Y{}, Y{}, Y{}, N_{}, N_{}, Y{}, Y{}, true, j);
            swap_pf();
            stage_k = (stage_k + 1) & (kStages - 1);
            ++j;
        }
        for (; j < NT; ++j) {                          // diagonal / ragged blocks: masked form + buffer swap
            block8(IC<0>{}, Y{}, Y{}, Y{}, Y{}, Y{}, N_{}, Y{}, Y{}, true, j);
            swap_pf();
            stage_k = (stage_k + 1) & (kStages - 1);
        }
        FA8_PHASE(3);           // an odd block out, masked (diagonal / ragged) blocks
        // this wave's last score product is issued: the next query block's Q fragments take the registers now and travel under
        // the drain, the staging-only tiles and the normalisation
        if (cont) load_q(tq, lane_here());
        // drain, "block" NT: the last slice, P V of block NT - 1 (tiles 0..6) and of block NT - 2 (tile 7) ...
        const bool sync_d = j < nt;
        block8(IC<0>{}, N_{}, Y{}, N_{}, Y{}, Y{}, N_{}, Y{}, N_{}, sync_d, j);
        if (sync_d) stage_k = (stage_k + 1) & (kStages - 1);
        ++j;
        // ... and tile 7 of block NT - 1 (its V^T fragment was read in region 7 above)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) o_acc[7][qt] = mfma8(vf[0], pf[1][qt], o_acc[7][qt]);
    } else if (cont) {
        load_q(tq, lane_here());
    }
    for (; j < nt; ++j) {                              // remaining tiles of the workgroup: staging duty only
        sync_and_stage(j);
        stage_k = (stage_k + 1) & (kStages - 1);
    }

    // ---- exact fallback (rare): per-block online softmax with running maximum and rescale, P = exp2(score - max + 7) <= 128.
    // Trigger: a P beyond e4m3's range turns into NaN in the conversion (v_cvt_pk_fp8_f32 under the default float mode:
    // tools/probes/cvt_fp8_probe.cpp) and poisons the row's rounded sum on the matrix pipe -- no per-score bookkeeping.
#if FA8_HALF_PRIO
    setprio_if<0>(late_half);
#endif
    FA8_PHASE(5);               // drain + staging-only tiles
    float l_part[2] = {l_a[0] + l_b[0], l_a[1] + l_b[1]};
    bool bad_row = !(l_acc[0][0] < kPLimit8 && l_acc[1][0] < kPLimit8);
    if constexpr (WANT_LSE) {
        // second trigger, where the exact sums exist anyway: the rounded P of a row add up to clearly less than the exact P,
        // i.e. a share of the row's weight sat below the window and was rounded to zero (e4m3's own rounding moves a sum by at
        // most 2^-4, and in both directions; a dominant term rounded down alone can make 6.25 %)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            float le = l_part[qt];
            le += __shfl_xor(le, 16);
            le += __shfl_xor(le, 32);
            bad_row = bad_row || (l_acc[qt][0] < kSumFloor8 * le);
        }
    } else if constexpr (FA8_SAMPLED_CHECK_ON != 0) {
        // ... and where they do not: the sampled bound on the weight under the window (kUnderCap8 above)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            float u = u_s[qt];
            u += __shfl_xor(u, 16);
            u += __shfl_xor(u, 32);
            bad_row = bad_row || (l_acc[qt][0] < kSampleGain8 * u);
        }
    }
    // ---- epilogue (as fa_fwd_kernel16.hpp): everything that needs no memory first, `mid()` (the caller's wait for its staging DMAs
    // and the next Q fragments), then the stores, which stay in flight across the barrier that follows
    auto epilogue = [&] __device__ (auto mid) {
        const int lane_e = lane_here();
        const int li = lane_e & 15, lg = lane_e >> 4;
        u32x4 outv[2][DT / 2];
        float lsev[2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            float l = l_part[qt];                                       // exact row sum (LSE)
            l += __shfl_xor(l, 16);
            l += __shfl_xor(l, 32);
            const float lq = l_acc[qt][0];                              // sum of the rounded P of query li (normalisation)
            const float inv = (lq > 0.f) ? p.out_scale / lq : 0.f;      // flash_attn_cutlass.cu:446-452 guard
            lsev[qt] = (l > 0.f) ? (m_c[qt] + __builtin_amdgcn_logf(l)) * 0.6931471805599453f : -INFINITY;
#pragma unroll
            for (int dt = 0; dt < DT; dt += 2) {
                const f32x4 oa = o_acc[dt][qt], ob = o_acc[dt + 1][qt];
                unsigned a0 = T::pack2(oa[0] * inv, oa[1] * inv), a1 = T::pack2(oa[2] * inv, oa[3] * inv);
                unsigned b0 = T::pack2(ob[0] * inv, ob[1] * inv), b1 = T::pack2(ob[2] * inv, ob[3] * inv);
                auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
                auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
                outv[qt][dt / 2] = u32x4{s0[0], s1[0], s0[1], s1[1]};
            }
        }
        mid();
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const int qrow = q0w + 16 * qt + li;
            if (p.lse != nullptr && lg == 0 && qrow < S) p.lse[((long long)b * p.H + h) * S + qrow] = lsev[qt];
            elem_t* orow = oh + (long long)qrow * p.o_ss;
#pragma unroll
            for (int dt = 0; dt < DT; dt += 2) {
                const int col = (lg & 1) ? (16 * (dt + 1) + 4 * (lg - 1)) : (16 * dt + 4 * lg);
                if (qrow < S && col < p.dv) *reinterpret_cast<u32x4*>(orow + col) = outv[qt][dt / 2];
            }
        }
    };
    auto no_mid = [] __device__ () {};
#if FA8_EARLY_EPILOGUE
    // A wave that has finished its blocks stores its output at once -- the result is final unless some wave of the workgroup asks
    // for the exact loop, which then recomputes and stores every row again.  Each wave posts its verdict in a flag word of its own
    // before the ONE barrier that also retires the ring and, on a continuous ring, publishes the next query block's first tiles.
    bool redo;
    {
        const unsigned flags = (unsigned)(uintptr_t)(lds_char*)&fa_flags8[pass & 1][0];
        const bool wave_bad = __builtin_amdgcn_ballot_w64(bad_row) != 0;
        if (lane_here() == 0) lds_write_b32(flags + 4 * wave, wave_bad ? 1u : 0u);
        epilogue([&] __device__ () {
            dma_wait<0>();                         // no DMA may still be writing LDS past this point; the next Q fragments are in
            if (cont) {                            // (consumed here, so that hipcc places its own wait for them here, not behind the stores)
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) asm volatile("" :: "v"(qf[qt]));
            }
        });
        __syncthreads();
        const u32x4 f0 = lds_read_b128(flags), f1 = lds_read_b128(flags + 16);
        redo = __builtin_amdgcn_readfirstlane(f0[0] | f0[1] | f0[2] | f0[3] | f1[0] | f1[1] | f1[2] | f1[3]) != 0;
    }
    FA8_PHASE(4);               // epilogue: normalise, store; the pass-end barrier
    if (redo) {
        if (cont) load_q(qb, lane_here());         // (the registers already hold the next block's fragments)
#else
    dma_wait<0>();
    if (wg_any(bad_row, lds_base + VBASE + (kStages - 1) * TILE, wave, lane_here(), NWAVES)) {
#endif
        constexpr int KO = 0, VO = VBASE;
        const int lane_f = lane_here();
        const int li = lane_f & 15, lg = lane_f >> 4;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) o_acc[dt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
        float m_r[2] = {-INFINITY, -INFINITY};
        l_part[0] = l_part[1] = 0.f;
        l_acc[0] = l_acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        unsigned ka2[2], va2[DT];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) ka2[hh] = lds_base + KO + li * ROWB + (k8_swz(li, lg + 4 * hh) << 4);
        {
            const int qq = li >> 1, pp = li & 1;
            const int key_in = 16 * (qq >> 2) + 4 * lg + (qq & 3);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) va2[dt] = lds_base + VO + key_in * ROWB + (v8_swz(key_in, dt) << 4) + 8 * pp;
        }
        auto dma2 = [&](int jj) {
            const unsigned st = (jj & 1) * TILE;
#pragma unroll
            for (int i = 0; i < CPT; ++i)
                dma16(rk_w, __builtin_amdgcn_readfirstlane(piece_base + KO + st + i * PIECE), (unsigned)jj * k_tile_stride + g_koff[i]);
#pragma unroll
            for (int i = 0; i < CPT; ++i)
                dma16(rv_w, __builtin_amdgcn_readfirstlane(piece_base + VO + st + i * PIECE), (unsigned)jj * v_tile_stride + g_voff[i]);
        };
        dma2(0);
        for (int jj = 0; jj < nt; ++jj) {
            dma_wait<0>();
            __syncthreads();                   // tile jj landed and is visible; tile jj-1's stage is free
            dma2(jj + 1);
            if (jj < my_nt) {
                const unsigned st = (jj & 1) * TILE;
                const int key0 = jj * kBN8;
                f32x4 sx[8][2];
#pragma unroll
                for (int kt = 0; kt < 8; ++kt) {
                    const u32x4 w0 = lds_read_b128(ka2[0] + st + kt * 16 * ROWB), w1 = lds_read_b128(ka2[1] + st + kt * 16 * ROWB);
                    const i32x8 kx = {(int)w0[0], (int)w0[1], (int)w0[2], (int)w0[3], (int)w1[0], (int)w1[1], (int)w1[2], (int)w1[3]};
#pragma unroll
                    for (int qt = 0; qt < 2; ++qt) sx[kt][qt] = mfma8(kx, qf[qt], f32x4{0.f, 0.f, 0.f, 0.f});
                }
                i32x8 px[2];
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) {
                    const int qrow = q0w + 16 * qt + li;
                    const int lim = (CAUSAL ? min(Sk - 1, qrow + coff) : Sk - 1) - 4 * lg - key0;
                    float mx = -INFINITY;
#pragma unroll
                    for (int kt = 0; kt < 8; ++kt)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (16 * kt + e > lim) sx[kt][qt][e] = -INFINITY;
                            mx = fmaxf(mx, sx[kt][qt][e]);
                        }
                    mx = fmaxf(mx, __shfl_xor(mx, 16));
                    mx = fmaxf(mx, __shfl_xor(mx, 32));
                    const float m_new = fmaxf(m_r[qt], mx * c);
                    const float alpha = (m_new == -INFINITY) ? 1.f : __builtin_amdgcn_exp2f(m_r[qt] - m_new);
                    l_part[qt] *= alpha;
                    l_acc[qt] *= alpha;
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) o_acc[dt][qt] *= alpha;
                    m_r[qt] = m_new;
                    const float m_sub = (m_new == -INFINITY) ? 0.f : m_new - 7.0f;     // P <= 2^7 (row fully masked so far: anything)
#pragma unroll
                    for (int kt = 0; kt < 8; ++kt) {
                        const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sx[kt][qt][0], c, -m_sub));
                        const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sx[kt][qt][1], c, -m_sub));
                        const float p2 = __builtin_amdgcn_exp2f(__builtin_fmaf(sx[kt][qt][2], c, -m_sub));
                        const float p3 = __builtin_amdgcn_exp2f(__builtin_fmaf(sx[kt][qt][3], c, -m_sub));
                        l_part[qt] += (p0 + p1) + (p2 + p3);
                        px[qt][kt] = pack_fp8(p2, p3, pack_fp8(p0, p1, 0, false), true);
                    }
                }
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    i32x8 vx;
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        const i32x2 w = lds_read_tr8_b64(va2[dt] + st + m * 32 * ROWB);
                        vx[2 * m] = w[0];
                        vx[2 * m + 1] = w[1];
                    }
#pragma unroll
                    for (int qt = 0; qt < 2; ++qt) o_acc[dt][qt] = mfma8(vx, px[qt], o_acc[dt][qt]);
                }
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) l_acc[qt] = mfma8(ones8, px[qt], l_acc[qt]);
            }
        }
        // sums and outputs carry the factor 2^7: the reference the epilogue sees is max - 7
        m_c[0] = (m_r[0] == -INFINITY) ? 0.f : m_r[0] - 7.0f;
        m_c[1] = (m_r[1] == -INFINITY) ? 0.f : m_r[1] - 7.0f;
        dma_wait<0>();
        __syncthreads();
#if FA8_EARLY_EPILOGUE
        epilogue(no_mid);
#endif
    }

    FA8_PHASE(6);               // the exact fallback, where taken (FA8_EARLY_EPILOGUE = 0: the check's two barriers)
    // ---- the next query block of a causal pair
    staged = false;
    if (pass + 1 < n_pass) {
#if FA8_EARLY_EPILOGUE
        if (cont && !redo) {                       // its first tiles and Q fragments are in place (continuous ring)
            staged = true;
            stage0 = (stage0 + nt) & (kStages - 1);
        } else
#endif
        {
            stage0 = 0;
            issue_prologue();
            load_q(tq, lane_here());
        }
    }

    FA8_PHASE(7);               // (the next pass's prologue issue)
#if !FA8_EARLY_EPILOGUE
    epilogue(no_mid);           // (old order: output normalised and stored while the next block's first tiles travel)
    FA8_PHASE(4);               // epilogue: normalise, store
#endif
  }  // pass
#undef FA8_PHASE
#if defined(FA_STAMP)
    if (lane_here() == 0 && blockIdx.x < 8192) {
        unsigned long long* d = g_fa_stamp + ((size_t)blockIdx.x * 8 + wave) * 24;
        unsigned long long st_rt1;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_rt1) :: "memory");
        d[0] = st_e; d[1] = st_w; d[2] = st_o; d[3] = st_n; d[4] = stamp_now() - st_t0; d[5] = st_rt1 - st_rt0;
#pragma unroll
        for (int i = 0; i < 12; ++i) d[8 + i] = st_ph[i];
    }
#endif
}

}  // namespace fa
