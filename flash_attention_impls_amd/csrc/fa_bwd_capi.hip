// C-ABI launcher of the gfx950 FlashAttention backward (see include/fa_mi355.h).
//
// Host-side counterpart of the reference's _FlashAttnFn.backward (code/triton_fa2/FA2-triton.py:207-237):
// there, dQ/dK/dV are zero-filled (:211-213) and one kernel adds dK/dV with fp16 atomics; here three launches
// write every output exactly once (pre-pass, dQ kernel, dK/dV kernel), nothing needs zeroing.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstring>

#include "../../include/fa_mi355.h"
#include "fa_capi_common.hpp"
#include "fa_bwd_kernel.hpp"
#include "fa_bwd_dkdv_kernel.hpp"
#include "fa_bwd_dq_gemm_kernel.hpp"
#include "fa_bwd_wide_kernel.hpp"
#include <cstdlib>

namespace {

using fa_capi::fail;
using fa_capi::g_err;
using fa_capi::set_strides;

template <class T, int D, int MODE, bool CAUSAL>
int launch_bwd(const fa::BwdParams& p, int grid, hipStream_t stream)
{
    constexpr int lds = fa::bwd_lds_bytes<D, MODE>();
    auto* kernel = &fa::fa_bwd_kernel<T, D, MODE, CAUSAL>;
    struct Tag {};                                    // (local to this instantiation of the launcher)
    const hipError_t attr_err = fa_capi::ensure_dynamic_lds<Tag>(reinterpret_cast<const void*>(kernel), lds);
    if (attr_err != hipSuccess)
        return fail(FA_ERR_LAUNCH, "hipFuncSetAttribute(lds=%d): %s", lds, hipGetErrorString(attr_err));
    hipLaunchKernelGGL((fa::fa_bwd_kernel<T, D, MODE, CAUSAL>), dim3(grid), dim3(64 * fa::bwd_waves<MODE>()), lds, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FA_ERR_LAUNCH, "backward kernel launch failed: %s", hipGetErrorString(e));
    return FA_OK;
}

// dK / dV: wave-specialised workgroups (fa_bwd_dkdv_kernel.hpp: a score wave and a gradient wave per SIMD).
// -DFA_BWD_DKDV_SINGLE selects MODE 1 of fa_bwd_kernel.hpp instead (one wave per SIMD doing everything: 4-8 % slower
// on cfg3, kept as a tuning option)
template <class T, int D, bool CAUSAL, bool WDS = false>
int launch_dkdv(const fa::BwdParams& p, int grid, hipStream_t stream)
{
#if defined(FA_BWD_DKDV_SINGLE)
    return launch_bwd<T, D, 1, CAUSAL>(p, grid, stream);
#else
    constexpr int lds = fa::dkdv_lds_bytes<D>();
    auto* kernel = &fa::fa_bwd_dkdv_kernel<T, D, CAUSAL, WDS>;
    struct Tag {};                                    // (local to this instantiation of the launcher)
    const hipError_t attr_err = fa_capi::ensure_dynamic_lds<Tag>(reinterpret_cast<const void*>(kernel), lds);
    if (attr_err != hipSuccess)
        return fail(FA_ERR_LAUNCH, "hipFuncSetAttribute(lds=%d): %s", lds, hipGetErrorString(attr_err));
    hipLaunchKernelGGL((fa::fa_bwd_dkdv_kernel<T, D, CAUSAL, WDS>), dim3(grid), dim3(512), lds, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FA_ERR_LAUNCH, "dK/dV kernel launch failed: %s", hipGetErrorString(e));
    return FA_OK;
#endif
}

// dQ = scale * dS . K over the hand-off workspace (fa_bwd_dq_gemm_kernel.hpp)
template <class T, int D, bool CAUSAL>
int launch_dq_gemm(const fa::BwdParams& p, int grid, hipStream_t stream)
{
    constexpr int lds = fa::dqg_lds_bytes<D>();
    auto* kernel = &fa::fa_bwd_dq_gemm_kernel<T, D, CAUSAL>;
    struct Tag {};
    const hipError_t attr_err = fa_capi::ensure_dynamic_lds<Tag>(reinterpret_cast<const void*>(kernel), lds);
    if (attr_err != hipSuccess)
        return fail(FA_ERR_LAUNCH, "hipFuncSetAttribute(lds=%d): %s", lds, hipGetErrorString(attr_err));
    hipLaunchKernelGGL((fa::fa_bwd_dq_gemm_kernel<T, D, CAUSAL>), dim3(grid), dim3(512), lds, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FA_ERR_LAUNCH, "dQ GEMM kernel launch failed: %s", hipGetErrorString(e));
    return FA_OK;
}

// The hand-off backward: dK/dV kernel first (it writes dS), then the dQ GEMM over it.
// `phase`: 0 = both kernels; experiment builds only (-DFA_BWD_EXPERIMENTS, FA_MI355_BWD_PHASE, tools/exp_bwd_overlap.py): 1 = the dK/dV kernel alone, 2 = the
// dQ GEMM alone over a workspace an earlier phase-1 call filled
template <class T, int D>
int run_bwd_ds(const fa::BwdParams& pq, int grid_q, const fa::BwdParams& pk, int grid_k, bool causal, hipStream_t s, int phase)
{
#if defined(FA_BWD_DKDV_SINGLE)
    return fail(FA_ERR_LAUNCH, "this build has no dS hand-off");
#else
    int rc = FA_OK;
    if (phase != 2) rc = causal ? launch_dkdv<T, D, true, true>(pk, grid_k, s) : launch_dkdv<T, D, false, true>(pk, grid_k, s);
    if (rc != FA_OK || phase == 1) return rc;
    return causal ? launch_dq_gemm<T, D, true>(pq, grid_q, s) : launch_dq_gemm<T, D, false>(pq, grid_q, s);
#endif
}

int ds_phase()
{
#if defined(FA_BWD_EXPERIMENTS)         // only in experiment builds (tools/exp_bwd_overlap.py): the shipped library ignores the variable
    const char* e = std::getenv("FA_MI355_BWD_PHASE");
    return (e && (e[0] == '1' || e[0] == '2')) ? e[0] - '0' : 0;
#else
    return 0;
#endif
}

template <class T, int D>
int run_bwd(const fa::BwdParams& pq, int grid_q, const fa::BwdParams& pk, int grid_k, bool causal, hipStream_t s)
{
    int rc = FA_OK;
#if !defined(FA_BWD_ONLY) || FA_BWD_ONLY == 0      // FA_BWD_ONLY: timing-only builds that launch one of the two kernels
    rc = causal ? launch_bwd<T, D, 0, true>(pq, grid_q, s) : launch_bwd<T, D, 0, false>(pq, grid_q, s);
    if (rc != FA_OK) return rc;
#endif
#if !defined(FA_BWD_ONLY) || FA_BWD_ONLY == 1
    rc = causal ? launch_dkdv<T, D, true>(pk, grid_k, s) : launch_dkdv<T, D, false>(pk, grid_k, s);
#endif
    return rc;
}

template <class T, int D>
int run_prep(const void* o, const void* d_o, const float* lse, float* stats, int B, int H, int S, int Spad, int dv,
             long long o_sb, long long o_sh, long long o_ss, long long g_sb, long long g_sh, long long g_ss, hipStream_t s)
{
    constexpr int RPB = 256 / (D / 8);
    const long long blocks = (long long)((Spad + RPB - 1) / RPB) * B * H;
    if (blocks > 0x7FFFFFFFll) return fail(FA_ERR_TOO_LARGE, "backward pre-pass grid too large");
    hipLaunchKernelGGL((fa::fa_bwd_prep_kernel<T, D>), dim3((unsigned)blocks), dim3(256), 0, s,
                       o, d_o, lse, stats, H, S, Spad, B * H, dv, o_sb, o_sh, o_ss, g_sb, g_sh, g_ss);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FA_ERR_LAUNCH, "backward pre-pass launch failed: %s", hipGetErrorString(e));
    return FA_OK;
}

// Parts a key/value head's group of G query heads is split into for the dK/dV kernel.  One workgroup per (key block,
// key/value head) streams all G query heads; with few key/value heads and a small batch (multi-query attention above
// all) that is too few, too unequal workgroups for the 256 CUs.  Splitting the group over `parts` workgroups multiplies the
// grid; each part writes fp32 partial sums and fa_bwd_reduce_kernel adds them up (fixed order: still deterministic).
// Measured (cfg3-like shapes with 1 ... 8 key/value heads, S = 4096 ... 16384): under the causal mask, where workgroups
// are of very unequal length, splitting pays up to about 2048 workgroups; without it only until the chip is full.
// Where the heads of a group do not suffice (or there is no group: G = 1), the visible query range of a key block is
// split as well (`qparts` workgroups per key block, at least four 64-row tiles each).
void dkdv_parts(int B, int H_kv, int G, int S_q, int S_k, bool causal, int& gparts, int& qparts)
{
    gparts = qparts = 1;
    const long long cus = fa_capi::device_cus();     // (256 on a whole MI355X; the workspace query and the launch see the same device)
    const long long want = causal ? 8 * cus : 2 * cus;
    const long long wgs = (long long)B * H_kv * ((S_k + 127) / 128);
    const long long tiles = (S_q + fa::kBN - 1) / fa::kBN;
    if (wgs >= want) return;                         // enough workgroups for the dispatcher to balance
    if (tiles * G < 64) return;                      // a workgroup of < 64 tiles (~0.1 ms): the reduction pass would cost more
    for (int d = 2; d <= G; ++d)
        if (G % d == 0) { gparts = d; if (wgs * d >= want) break; }
    // the query range is split only where there are fewer workgroups than CUs: its partial sums are as large as dK and
    // dV themselves per part (measured: with 512 or more workgroups the extra traffic costs more than the balance gains)
    if (wgs * gparts >= cus) return;
    const long long more = (2 * cus + wgs * gparts - 1) / (wgs * gparts);
    qparts = (int)std::max(1ll, std::min(std::min(more, tiles / 4), 8ll));
}

size_t stats_bytes(int B, int H, int S)
{
    const size_t spad = ((size_t)S + fa::kBN - 1) / fa::kBN * fa::kBN;
    return (size_t)2 * B * H * spad * sizeof(float);
}

size_t partial_bytes(int B, int H_kv, int parts, int S_k, int D)       // one of the two fp32 partial tensors
{
    return parts > 1 ? (size_t)B * H_kv * parts * S_k * D * sizeof(float) : 0;
}

template <class T>
int run_reduce(const float* part, void* out, int B, int Hkv, int S, int dv, int parts, long long sb, long long sh, long long ss, hipStream_t s)
{
    const long long total8 = (long long)B * Hkv * S * (dv / 8);
    const long long blocks = (total8 + 255) / 256;
    if (blocks > 0x7FFFFFFFll) return fail(FA_ERR_TOO_LARGE, "reduction grid too large");
    hipLaunchKernelGGL((fa::fa_bwd_reduce_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, s, part, out, Hkv, S, dv, parts, total8, sb, sh, ss);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FA_ERR_LAUNCH, "dK/dV reduction launch failed: %s", hipGetErrorString(e));
    return FA_OK;
}

// Geometry of the dS hand-off workspace (fa_bwd_dq_gemm_kernel.hpp): 2 KiB units [query head][32-key slab][32-query block];
// slabs padded to whole 64-key tiles, blocks to whole 256-row workgroups.  Both kernels address it through descriptors of at
// most two slab rows, so one head's image may be of any size; 0: a slab row itself would pass the 32-bit range (S_q >= 2^24).
size_t ds_head_bytes(int S_q, int S_k, unsigned& row_bytes)
{
    const unsigned long long nq8 = ((unsigned long long)(S_q + 31) / 32 + 7) / 8 * 8;
    const unsigned long long nk2 = ((unsigned long long)(S_k + 31) / 32 + 1) / 2 * 2;
    const unsigned long long row = nq8 * 2048, head = nk2 * row;
    row_bytes = (unsigned)row;
    if (2 * row >= (1ull << 30)) return 0;
    return (size_t)head;
}

// Where the hand-off pays (profiles/r3_bwd_handoff_sweep.txt, r3_bwd_handoff_sweep_d64.txt): it moves 2 S_q S_k bytes per head
// -- half of that under the causal mask -- whatever the head_dim, and the work it saves is proportional to head_dim.
// head_dim > 64: +8 ... 14 % on large shapes, more on small ones: always.  head_dim <= 64: while the dS that is actually written
// and read stays within half of the 256 MiB Infinity Cache (+9 ... 34 % up to 64 MiB, -1 ... +25 % at 128 MiB); streamed through
// HBM it costs more than the recompute kernel it replaces ((8,32,4096,64): -8 % causal, -17 % non-causal).
bool ds_pays(int D, size_t image_bytes, bool causal)
{
#if defined(FA_BWD_DS_ALWAYS)           // (tuning builds: the hand-off wherever it qualifies, to measure where it stops paying)
    return true;
#endif
    if (D > 64) return true;
    return image_bytes / (causal ? 2 : 1) <= ((size_t)128 << 20);
}

bool ds_enabled()
{
#if defined(FA_BWD_DS_DISABLE)      // A/B arm: the recompute path whatever the workspace
    return false;
#endif
    const char* e = std::getenv("FA_MI355_BWD_DS");           // "0": never take the hand-off path (A/B, tests of the recompute path)
    return !(e && e[0] == '0');
}

int bwd_grid(long long bh, long long blocks_per_head)
{
    const long long g = fa_capi::grid_blocks(bh, blocks_per_head, fa_capi::head_split(bh, blocks_per_head));   // (virtual) heads padded to 8 XCD groups
    return g > 0x7FFFFFFFll ? -1 : (int)g;
}

// head_dim 144 .. 256: the P / dS producer of fa_bwd_wide_kernel.hpp
template <class T, bool CAUSAL>
int launch_wide_ds(const fa::BwdParams& p, int grid, hipStream_t stream)
{
    constexpr int lds = fa::kBwdWideLds;
    auto* kernel = &fa::fa_bwd_wide_ds_kernel<T, CAUSAL>;
    struct Tag {};
    const hipError_t attr_err = fa_capi::ensure_dynamic_lds<Tag>(reinterpret_cast<const void*>(kernel), lds);
    if (attr_err != hipSuccess)
        return fail(FA_ERR_LAUNCH, "hipFuncSetAttribute(lds=%d): %s", lds, hipGetErrorString(attr_err));
    hipLaunchKernelGGL((fa::fa_bwd_wide_ds_kernel<T, CAUSAL>), dim3(grid), dim3(fa::kBwdWideThreads), lds, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FA_ERR_LAUNCH, "kernel launch failed: %s", hipGetErrorString(e));
    return FA_OK;
}

}  // namespace

extern "C" {

size_t fa_bwd_workspace_bytes(int B, int H, int S)
{
    if (B <= 0 || H <= 0 || S <= 0) return 0;
    return stats_bytes(B, H, S);
}

size_t fa_bwd_ex_workspace_bytes(int B, int H, int H_kv, int S_q, int S_k, int D)
{
    if (B <= 0 || H <= 0 || H_kv <= 0 || S_q <= 0 || S_k <= 0 || D <= 0 || H % H_kv != 0) return 0;
    int gp, qp;
    dkdv_parts(B, H_kv, H / H_kv, S_q, S_k, /*causal=*/true, gp, qp);          // (the causal mask splits further)
    return ((stats_bytes(B, H, S_q) + 255) / 256) * 256 + 2 * partial_bytes(B, H_kv, gp * qp, S_k, D);
}

#if defined(FA_BWD_STAMP)
// diagnostic build (tools/stamps_bwd.py): read (and clear) the dK/dV kernel's stamp sums
int fa_debug_read_bwd_stamps(void* dst, size_t bytes, int clear)
{
    hipError_t e = hipMemcpyFromSymbol(dst, HIP_SYMBOL(fa::g_fa_bwd_stamp), bytes, 0, hipMemcpyDeviceToHost);
    if (e == hipSuccess && clear) {
        static const unsigned long long zeros[64] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(fa::g_fa_bwd_stamp), zeros, sizeof(zeros), 0, hipMemcpyHostToDevice);
    }
    return (int)e;
}
#endif

size_t fa_bwd_ds_workspace_bytes(int B, int H, int H_kv, int S_q, int S_k, int D)
{
#if defined(FA_BWD_DKDV_SINGLE)
    return 0;
#endif
    const size_t base = fa_bwd_ex_workspace_bytes(B, H, H_kv, S_q, S_k, D);
    if (base == 0 || D < 16 || D > 128 || D % 16 != 0) return 0;
    unsigned row;
    const size_t head = ds_head_bytes(S_q, S_k, row);
    if (head == 0) return 0;
    if (!ds_pays(D, (size_t)B * H * head, /*causal=*/true)) return 0;        // (the size is asked for without the mask: the laxer rule)
    return (base + 255) / 256 * 256 + (size_t)B * H * head;
}

int fa_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
           void* dq, void* dk, void* dv,
           int B, int H, int S, int D,
           const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
           const int64_t* o_strides, const int64_t* do_strides,
           const int64_t* dq_strides, const int64_t* dk_strides, const int64_t* dv_strides,
           int dtype, int causal, float softmax_scale,
           void* workspace, size_t workspace_bytes, void* stream)
{
    return fa_bwd_ex(q, k, v, o, d_o, lse, dq, dk, dv, B, H, H, S, S, D, q_strides, k_strides, v_strides, o_strides, do_strides,
                      dq_strides, dk_strides, dv_strides, dtype, causal, softmax_scale, workspace, workspace_bytes, stream);
}

int fa_bwd_ex(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
               void* dq, void* dk, void* dv,
               int B, int H, int H_kv, int S, int S_k, int D,
               const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
               const int64_t* o_strides, const int64_t* do_strides,
               const int64_t* dq_strides, const int64_t* dk_strides, const int64_t* dv_strides,
               int dtype, int causal, float softmax_scale,
               void* workspace, size_t workspace_bytes, void* stream)
{
    g_err[0] = 0;
    if (dtype != FA_DTYPE_BF16 && dtype != FA_DTYPE_FP16)
        return fail(FA_ERR_BAD_DTYPE, "backward supports bf16 and fp16 (dtype code %d)", dtype);
    if (D < 16 || D > 128 || D % 16 != 0)
        return fail(FA_ERR_BAD_HEAD_DIM, "head_dim %d not supported (need D %% 16 == 0 and 16 <= D <= 128)", D);
    const bool big = D > 64;                 // head_dim-128 kernels, otherwise the head_dim-64 ones (columns past D read as zeros)
    if (B < 0 || H < 0 || S < 0) return fail(FA_ERR_BAD_SHAPE, "negative shape B=%d H=%d S=%d", B, H, S);
    if (B == 0 || H == 0 || S == 0) return FA_OK;
    if (H_kv <= 0 || H % H_kv != 0)
        return fail(FA_ERR_BAD_SHAPE, "H=%d query heads are not a multiple of H_kv=%d key/value heads", H, H_kv);
    if (S_k <= 0) return fail(FA_ERR_BAD_SHAPE, "S_k=%d: no keys", S_k);
#if defined(FA_BWD_DKDV_SINGLE)
    if (H_kv != H || S_k != S) return fail(FA_ERR_BAD_SHAPE, "this build's dK/dV kernel serves equal head counts and lengths only");
#endif
    if (!q || !k || !v || !o || !d_o || !lse || !dq || !dk || !dv || !workspace)
        return fail(FA_ERR_NULL_PTR, "null tensor / workspace pointer");
    if (workspace_bytes < fa_bwd_workspace_bytes(B, H, S))
        return fail(FA_ERR_BAD_SHAPE, "workspace too small: %zu < %zu bytes", workspace_bytes, fa_bwd_workspace_bytes(B, H, S));

    long long st[8][3];
    const int64_t* given[8] = {q_strides, k_strides, v_strides, o_strides, do_strides, dq_strides, dk_strides, dv_strides};
    const void* ptrs[8] = {q, k, v, o, d_o, dq, dk, dv};
    long long max_ss = 0;
    const int heads[8] = {H, H_kv, H_kv, H, H, H, H_kv, H_kv};
    const int rows[8] = {S, S_k, S_k, S, S, S, S_k, S_k};
    for (int i = 0; i < 8; ++i) {
        if (!set_strides(given[i], heads[i], rows[i], D, st[i][0], st[i][1], st[i][2]))
            return fail(FA_ERR_BAD_STRIDE, "strides must be non-negative with seq stride >= head_dim");
        for (int c = 0; c < 3; ++c)
            if ((st[i][c] * 2) % 16 != 0) return fail(FA_ERR_BAD_STRIDE, "stride %lld elements is not 16-byte aligned", st[i][c]);
        if (reinterpret_cast<uintptr_t>(ptrs[i]) % 16 != 0) return fail(FA_ERR_BAD_STRIDE, "tensor base pointer not 16-byte aligned");
        max_ss = std::max(max_ss, st[i][2]);
    }
    if (reinterpret_cast<uintptr_t>(workspace) % 16 != 0) return fail(FA_ERR_BAD_STRIDE, "workspace not 16-byte aligned");
    // 32-bit buffer offsets inside one (batch, head) slice, with room for two prefetched tiles
    if (((long long)std::max(S, S_k) + 4 * fa::kBN) * max_ss * 2 >= (1ll << 31))
        return fail(FA_ERR_TOO_LARGE, "one (batch, head) slice spans >= 2 GiB (S=%d, seq stride=%lld)", std::max(S, S_k), max_ss);
    const int Spad = (S + fa::kBN - 1) / fa::kBN * fa::kBN;
    if ((long long)2 * B * H * Spad * 4 + 4 * fa::kBN * 4 >= (1ll << 31))
        return fail(FA_ERR_TOO_LARGE, "row statistics exceed 2 GiB (B*H*S = %lld)", (long long)B * H * S);

    const float scale = (softmax_scale > 0.f) ? softmax_scale : 1.0f / std::sqrt((float)D);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    float* stats = static_cast<float*>(workspace);

    int rc = FA_OK;
    const int phase = ds_phase();
    if (phase == 2) {}                        // (experiment: the statistics are those of the phase-1 call)
    else if (dtype == FA_DTYPE_BF16)
        rc = big ? run_prep<fa::TypeBF16, 128>(o, d_o, lse, stats, B, H, S, Spad, D, st[3][0], st[3][1], st[3][2], st[4][0], st[4][1], st[4][2], s)
                      : run_prep<fa::TypeBF16, 64>(o, d_o, lse, stats, B, H, S, Spad, D, st[3][0], st[3][1], st[3][2], st[4][0], st[4][1], st[4][2], s);
    else
        rc = big ? run_prep<fa::TypeF16, 128>(o, d_o, lse, stats, B, H, S, Spad, D, st[3][0], st[3][1], st[3][2], st[4][0], st[4][1], st[4][2], s)
                      : run_prep<fa::TypeF16, 64>(o, d_o, lse, stats, B, H, S, Spad, D, st[3][0], st[3][1], st[3][2], st[4][0], st[4][1], st[4][2], s);
    if (rc != FA_OK) return rc;

    fa::BwdParams pq;
    memset(&pq, 0, sizeof(pq));
    pq.stats = stats;
    pq.B = B; pq.H = H; pq.S = S; pq.dv = D; pq.Spad = Spad; pq.bh = B * H;
    pq.G = H / H_kv;
    pq.Sy = S_k;                        // dQ: queries stationary, keys streamed
    pq.coff = S_k - S;
    pq.scale = scale;
    pq.scale_log2 = scale * 1.4426950408889634f;
    fa::BwdParams pk = pq;
    // MODE 0 (dQ): stationary Q, dO; streamed K, V
    pq.x1 = q; pq.x2 = d_o; pq.y1 = k; pq.y2 = v; pq.out1 = dq; pq.out2 = nullptr;
    pq.x1_sb = st[0][0]; pq.x1_sh = st[0][1]; pq.x1_ss = st[0][2];
    pq.x2_sb = st[4][0]; pq.x2_sh = st[4][1]; pq.x2_ss = st[4][2];
    pq.y1_sb = st[1][0]; pq.y1_sh = st[1][1]; pq.y1_ss = st[1][2];
    pq.y2_sb = st[2][0]; pq.y2_sh = st[2][1]; pq.y2_ss = st[2][2];
    pq.o1_sb = st[5][0]; pq.o1_sh = st[5][1]; pq.o1_ss = st[5][2];
    pq.nxb = (S + 32 * fa::bwd_waves<0>() - 1) / (32 * fa::bwd_waves<0>());
    // dK, dV: stationary K, V (the grid runs over the key/value heads); streamed Q, dO of the group's query heads
    // a group split over several workgroups if that helps and the caller's workspace (fa_bwd_ex_workspace_bytes) has room
    // for the fp32 partial sums; with the smaller fa_bwd_workspace_bytes the unsplit kernel runs
    int gparts, qparts;
    dkdv_parts(B, H_kv, H / H_kv, S, S_k, causal != 0, gparts, qparts);
    const size_t part_off = ((stats_bytes(B, H, S) + 255) / 256) * 256;
    size_t part_one = partial_bytes(B, H_kv, gparts * qparts, S_k, D);
#if defined(FA_BWD_DKDV_SINGLE)
    gparts = qparts = 1;
#endif
    if (gparts * qparts > 1 && workspace_bytes < part_off + 2 * part_one) gparts = qparts = 1;
    const int parts = gparts * qparts;
    pk.xsplit = gparts;
    pk.qsplit = qparts;
    pk.G = (H / H_kv) / gparts;
    pk.H = H_kv * parts; pk.bh = B * H_kv * parts;
    pk.S = S_k; pk.Sy = S;
    pk.x1 = k; pk.x2 = v; pk.y1 = q; pk.y2 = d_o; pk.out1 = dk; pk.out2 = dv;
    float* part_dk = reinterpret_cast<float*>(static_cast<char*>(workspace) + part_off);
    float* part_dv = reinterpret_cast<float*>(static_cast<char*>(workspace) + part_off + part_one);
    if (parts > 1) { pk.out1 = part_dk; pk.out2 = part_dv; }
    pk.x1_sb = st[1][0]; pk.x1_sh = st[1][1]; pk.x1_ss = st[1][2];
    pk.x2_sb = st[2][0]; pk.x2_sh = st[2][1]; pk.x2_ss = st[2][2];
    pk.y1_sb = st[0][0]; pk.y1_sh = st[0][1]; pk.y1_ss = st[0][2];
    pk.y2_sb = st[4][0]; pk.y2_sh = st[4][1]; pk.y2_ss = st[4][2];
    pk.o1_sb = st[6][0]; pk.o1_sh = st[6][1]; pk.o1_ss = st[6][2];
    pk.o2_sb = st[7][0]; pk.o2_sh = st[7][1]; pk.o2_ss = st[7][2];
    pk.nxb = (S_k + 32 * fa::bwd_waves<1>() - 1) / (32 * fa::bwd_waves<1>());

    // under the causal mask a dQ workgroup takes a pair of query blocks (see fa_bwd_kernel.hpp), unless single blocks are
    // expected to finish earlier (fa_capi::causal_unpaired)
    pq.unpaired = (causal != 0 && fa_capi::causal_unpaired((long long)B * H, pq.nxb, fa_capi::device_cus())) ? 1 : 0;
    pq.hsplit = fa_capi::head_split((long long)B * H, (causal != 0 && !pq.unpaired) ? (pq.nxb + 1) / 2 : pq.nxb);
    pk.hsplit = fa_capi::head_split((long long)pk.bh, pk.nxb);
    const int grid_q = bwd_grid((long long)B * H, (causal != 0 && !pq.unpaired) ? (pq.nxb + 1) / 2 : pq.nxb);
    const int grid_k = bwd_grid((long long)B * H_kv * parts, pk.nxb);
    if (grid_q <= 0 || grid_k <= 0) return fail(FA_ERR_TOO_LARGE, "grid too large");
    const bool c = causal != 0;
    // a workspace of fa_bwd_ds_workspace_bytes selects the dS hand-off: 5 matrix products instead of 7
    const size_t ds_need = fa_bwd_ds_workspace_bytes(B, H, H_kv, S, S_k, D);
    unsigned ds_row_probe;
    const bool use_ds = ds_need != 0 && workspace_bytes >= ds_need && ds_enabled() &&
                        ds_pays(D, (size_t)B * H * ds_head_bytes(S, S_k, ds_row_probe), c);
    if (use_ds) {
        unsigned row;
        const size_t head_bytes = ds_head_bytes(S, S_k, row);
        char* ds = static_cast<char*>(workspace) + (fa_bwd_ex_workspace_bytes(B, H, H_kv, S, S_k, D) + 255) / 256 * 256;
        pq.ds = pk.ds = ds;
        pq.ds_head_bytes = pk.ds_head_bytes = (long long)head_bytes;
        pq.ds_row_bytes = pk.ds_row_bytes = row;
        if (dtype == FA_DTYPE_BF16)
            rc = big ? run_bwd_ds<fa::TypeBF16, 128>(pq, grid_q, pk, grid_k, c, s, phase) : run_bwd_ds<fa::TypeBF16, 64>(pq, grid_q, pk, grid_k, c, s, phase);
        else
            rc = big ? run_bwd_ds<fa::TypeF16, 128>(pq, grid_q, pk, grid_k, c, s, phase) : run_bwd_ds<fa::TypeF16, 64>(pq, grid_q, pk, grid_k, c, s, phase);
    } else if (dtype == FA_DTYPE_BF16)
        rc = big ? run_bwd<fa::TypeBF16, 128>(pq, grid_q, pk, grid_k, c, s) : run_bwd<fa::TypeBF16, 64>(pq, grid_q, pk, grid_k, c, s);
    else
        rc = big ? run_bwd<fa::TypeF16, 128>(pq, grid_q, pk, grid_k, c, s) : run_bwd<fa::TypeF16, 64>(pq, grid_q, pk, grid_k, c, s);
    if (rc != FA_OK || parts == 1) return rc;
    // the parts' fp32 partial sums -> dK, dV (16-bit, caller's strides)
    for (int which = 0; which < 2 && rc == FA_OK; ++which) {
        const float* src = which == 0 ? part_dk : part_dv;
        void* dst = which == 0 ? dk : dv;
        const long long* so = st[6 + which];
        rc = dtype == FA_DTYPE_BF16 ? run_reduce<fa::TypeBF16>(src, dst, B, H_kv, S_k, D, parts, so[0], so[1], so[2], s)
                                    : run_reduce<fa::TypeF16>(src, dst, B, H_kv, S_k, D, parts, so[0], so[1], so[2], s);
    }
    return rc;
}

size_t fa_bwd_wide_workspace_bytes(int B, int H, int S)
{
    if (B <= 0 || H <= 0 || S <= 0) return 0;
    return stats_bytes(B, H, S);
}

int fa_bwd_wide_ds(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
                   void* p_image, void* ds_image, long long ld,
                   int B, int H, int H_kv, int S, int S_k, int D,
                   const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                   const int64_t* o_strides, const int64_t* do_strides,
                   int dtype, int causal, float softmax_scale,
                   void* workspace, size_t workspace_bytes, void* stream)
{
    g_err[0] = 0;
    if (dtype != FA_DTYPE_BF16 && dtype != FA_DTYPE_FP16)
        return fail(FA_ERR_BAD_DTYPE, "backward supports bf16 and fp16 (dtype code %d)", dtype);
    if (D <= 128 || D > fa::kBwdWideD || D % 16 != 0)
        return fail(FA_ERR_BAD_HEAD_DIM, "fa_bwd_wide_ds serves head_dim 144 .. %d (multiples of 16); got %d (fa_bwd_ex serves 16 .. 128)", fa::kBwdWideD, D);
    if (B < 0 || H < 0 || S < 0) return fail(FA_ERR_BAD_SHAPE, "negative shape B=%d H=%d S=%d", B, H, S);
    if (B == 0 || H == 0 || S == 0) return FA_OK;
    if (H_kv <= 0 || H % H_kv != 0)
        return fail(FA_ERR_BAD_SHAPE, "H=%d query heads are not a multiple of H_kv=%d key/value heads", H, H_kv);
    if (S_k <= 0) return fail(FA_ERR_BAD_SHAPE, "S_k=%d: no keys", S_k);
    if (ld < S_k || ld % 8 != 0) return fail(FA_ERR_BAD_STRIDE, "image row stride ld=%lld must be a multiple of 8 and >= S_k=%d", ld, S_k);
    if (!q || !k || !v || !o || !d_o || !lse || !p_image || !ds_image || !workspace)
        return fail(FA_ERR_NULL_PTR, "null tensor / image / workspace pointer");
    if (workspace_bytes < fa_bwd_wide_workspace_bytes(B, H, S))
        return fail(FA_ERR_BAD_SHAPE, "workspace too small: %zu < %zu bytes", workspace_bytes, fa_bwd_wide_workspace_bytes(B, H, S));

    long long st[5][3];
    const int64_t* given[5] = {q_strides, k_strides, v_strides, o_strides, do_strides};
    const void* ptrs[7] = {q, k, v, o, d_o, p_image, ds_image};
    long long max_ss = 0;
    const int heads[5] = {H, H_kv, H_kv, H, H};
    const int rows[5] = {S, S_k, S_k, S, S};
    for (int i = 0; i < 5; ++i) {
        if (!set_strides(given[i], heads[i], rows[i], D, st[i][0], st[i][1], st[i][2]))
            return fail(FA_ERR_BAD_STRIDE, "strides must be non-negative with seq stride >= head_dim");
        for (int c = 0; c < 3; ++c)
            if ((st[i][c] * 2) % 16 != 0) return fail(FA_ERR_BAD_STRIDE, "stride %lld elements is not 16-byte aligned", st[i][c]);
        max_ss = std::max(max_ss, st[i][2]);
    }
    for (int i = 0; i < 7; ++i)
        if (reinterpret_cast<uintptr_t>(ptrs[i]) % 16 != 0) return fail(FA_ERR_BAD_STRIDE, "tensor / image base pointer not 16-byte aligned");
    if (reinterpret_cast<uintptr_t>(workspace) % 16 != 0) return fail(FA_ERR_BAD_STRIDE, "workspace not 16-byte aligned");
    if (((long long)std::max(S, S_k) + 4 * fa::kBN) * max_ss * 2 >= (1ll << 31))
        return fail(FA_ERR_TOO_LARGE, "one (batch, head) slice spans >= 2 GiB (S=%d, seq stride=%lld)", std::max(S, S_k), max_ss);
    const int Spad = (S + fa::kBN - 1) / fa::kBN * fa::kBN;
    if ((long long)2 * B * H * Spad * 4 + 4 * fa::kBN * 4 >= (1ll << 31))
        return fail(FA_ERR_TOO_LARGE, "row statistics exceed 2 GiB (B*H*S = %lld)", (long long)B * H * S);

    const float scale = (softmax_scale > 0.f) ? softmax_scale : 1.0f / std::sqrt((float)D);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    float* stats = static_cast<float*>(workspace);
    int rc = dtype == FA_DTYPE_BF16
                 ? run_prep<fa::TypeBF16, fa::kBwdWideD>(o, d_o, lse, stats, B, H, S, Spad, D, st[3][0], st[3][1], st[3][2], st[4][0], st[4][1], st[4][2], s)
                 : run_prep<fa::TypeF16, fa::kBwdWideD>(o, d_o, lse, stats, B, H, S, Spad, D, st[3][0], st[3][1], st[3][2], st[4][0], st[4][1], st[4][2], s);
    if (rc != FA_OK) return rc;

    fa::BwdParams p;
    memset(&p, 0, sizeof(p));
    p.stats = stats;
    p.B = B; p.H = H; p.S = S; p.dv = D; p.Spad = Spad; p.bh = B * H;
    p.G = H / H_kv;
    p.Sy = S_k;
    p.coff = S_k - S;
    p.scale = scale;
    p.scale_log2 = scale * 1.4426950408889634f;
    p.x1 = q; p.x2 = d_o; p.y1 = k; p.y2 = v; p.out1 = p_image; p.out2 = ds_image;
    p.x1_sb = st[0][0]; p.x1_sh = st[0][1]; p.x1_ss = st[0][2];
    p.x2_sb = st[4][0]; p.x2_sh = st[4][1]; p.x2_ss = st[4][2];
    p.y1_sb = st[1][0]; p.y1_sh = st[1][1]; p.y1_ss = st[1][2];
    p.y2_sb = st[2][0]; p.y2_sh = st[2][1]; p.y2_ss = st[2][2];
    p.o1_ss = p.o2_ss = ld;
    p.o1_sh = p.o2_sh = (long long)S * ld;
    p.o1_sb = p.o2_sb = (long long)H * S * ld;
    p.nxb = (S + fa::kBwdWideRows - 1) / fa::kBwdWideRows;
    p.unpaired = 1;
    p.hsplit = fa_capi::head_split((long long)B * H, p.nxb);
    const long long g = fa_capi::grid_blocks((long long)B * H, p.nxb, p.hsplit);
    if (g > 0x7FFFFFFFll) return fail(FA_ERR_TOO_LARGE, "grid too large");
    const bool c = causal != 0;
    if (dtype == FA_DTYPE_BF16) return c ? launch_wide_ds<fa::TypeBF16, true>(p, (int)g, s) : launch_wide_ds<fa::TypeBF16, false>(p, (int)g, s);
    return c ? launch_wide_ds<fa::TypeF16, true>(p, (int)g, s) : launch_wide_ds<fa::TypeF16, false>(p, (int)g, s);
}

}  // extern "C"
