// C-ABI launcher for the gfx950 FlashAttention forward kernels (see include/fa_mi355.h).
//
// Host-side counterpart of the reference's launchers:
//   _FlashAttnFn.forward                 code/triton_fa2/FA2-triton.py:175-205
//   flash_attn_cutlass_forward<D>        code/cutlass_cuda_fa1/run/flash_attn_cutlass.cu:457-515
//   flash_attention_cutlass_dispatch     code/cutlass_cuda_fa1/run/flash_attn_cutlass.cu:519-544
// Differences by design: int status + thread-local message instead of fprintf and a silent
// no-op on unsupported head_dim (:540-542); no first-call banner / static bool (:487-502).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <algorithm>
#include <cstring>
#include <cstdlib>

#include "../../include/fa_mi355.h"
#include "fa_capi_common.hpp"
#include "fa_fwd_kernel.hpp"
#include "fa_fwd_kernel16.hpp"
#include "fa_fwd_kernel8.hpp"
#include "fa_fwd_kernel_wide.hpp"

// 32-row query blocks per wave: 1 = 8 waves per workgroup (two per SIMD), 2 = 4 waves (one per SIMD, 512 registers)
#ifndef FA_QB
#define FA_QB 1
#endif

namespace {

constexpr int kQB = FA_QB;
constexpr int kThreads = fa::threads_per_wg<kQB>();

using fa_capi::fail;
using fa_capi::g_err;
using fa_capi::set_strides;

template <class T, int D, bool CAUSAL>
int launch(const fa::FwdParams& p, int grid, hipStream_t stream)
{
    constexpr int lds = fa::lds_bytes<D>();
    auto* kernel = &fa::fa_fwd_kernel<T, D, CAUSAL, kQB>;
    struct Tag {};                                    // (local to this instantiation of the launcher)
    const hipError_t attr_err = fa_capi::ensure_dynamic_lds<Tag>(reinterpret_cast<const void*>(kernel), lds);
    if (attr_err != hipSuccess)
        return fail(FA_ERR_LAUNCH, "hipFuncSetAttribute(lds=%d): %s", lds, hipGetErrorString(attr_err));
    hipLaunchKernelGGL((fa::fa_fwd_kernel<T, D, CAUSAL, kQB>), dim3(grid), dim3(kThreads), lds, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FA_ERR_LAUNCH, "kernel launch failed: %s", hipGetErrorString(e));
    return FA_OK;
}

// 128-row workgroups (4 waves x 32 rows) of the 32x32x16 kernel: launches whose 256-row grid would leave CUs without work
template <class T, int D, bool CAUSAL>
int launch_rows128(const fa::FwdParams& p, int grid, hipStream_t stream)
{
    constexpr int lds = fa::lds_bytes<D>();
    auto* kernel = &fa::fa_fwd_kernel<T, D, CAUSAL, 1, 4>;
    struct Tag {};
    const hipError_t attr_err = fa_capi::ensure_dynamic_lds<Tag>(reinterpret_cast<const void*>(kernel), lds);
    if (attr_err != hipSuccess)
        return fail(FA_ERR_LAUNCH, "hipFuncSetAttribute(lds=%d): %s", lds, hipGetErrorString(attr_err));
    hipLaunchKernelGGL((fa::fa_fwd_kernel<T, D, CAUSAL, 1, 4>), dim3(grid), dim3(256), lds, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FA_ERR_LAUNCH, "kernel launch failed: %s", hipGetErrorString(e));
    return FA_OK;
}

// kernel on 16x16x32 MFMA tiles (fa_fwd_kernel16.hpp), compiled head_dim D = 128 or 64
template <class T, bool CAUSAL, int D>
int launch16(const fa::FwdParams& p, int grid, hipStream_t stream)
{
    constexpr int lds = fa::lds_bytes<D>();
    auto* kernel = &fa::fa_fwd_kernel16<T, CAUSAL, false, D>;
    struct Tag {};                                    // (local to this instantiation of the launcher)
    const hipError_t attr_err = fa_capi::ensure_dynamic_lds<Tag>(reinterpret_cast<const void*>(kernel), lds);
    if (attr_err != hipSuccess)
        return fail(FA_ERR_LAUNCH, "hipFuncSetAttribute(lds=%d): %s", lds, hipGetErrorString(attr_err));
    hipLaunchKernelGGL((fa::fa_fwd_kernel16<T, CAUSAL, false, D>), dim3(grid), dim3(512), lds, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FA_ERR_LAUNCH, "kernel launch failed: %s", hipGetErrorString(e));
    return FA_OK;
}

// the same kernel with 128-row workgroups (4 waves, one per SIMD) at head_dim 128: bitwise the 8-wave form's output
template <class T, bool CAUSAL>
int launch16_rows128(const fa::FwdParams& p, int grid, hipStream_t stream)
{
    constexpr int lds = fa::lds_bytes<128>();
    auto* kernel = &fa::fa_fwd_kernel16<T, CAUSAL, false, 128, 4>;
    struct Tag {};
    const hipError_t attr_err = fa_capi::ensure_dynamic_lds<Tag>(reinterpret_cast<const void*>(kernel), lds);
    if (attr_err != hipSuccess)
        return fail(FA_ERR_LAUNCH, "hipFuncSetAttribute(lds=%d): %s", lds, hipGetErrorString(attr_err));
    hipLaunchKernelGGL((fa::fa_fwd_kernel16<T, CAUSAL, false, 128, 4>), dim3(grid), dim3(256), lds, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FA_ERR_LAUNCH, "kernel launch failed: %s", hipGetErrorString(e));
    return FA_OK;
}

// head_dim > 64, fp8 Q and K read natively by fp8 MFMAs, V and O bf16 (fa_fwd_kernel16<TypeBF16, CAUSAL, true>)
template <bool CAUSAL>
int launch16_qk8(const fa::FwdParams& p, int grid, hipStream_t stream)
{
    constexpr int lds = fa::lds_bytes<128>();
    auto* kernel = &fa::fa_fwd_kernel16<fa::TypeBF16, CAUSAL, true>;
    struct Tag {};                                    // (local to this instantiation of the launcher)
    const hipError_t attr_err = fa_capi::ensure_dynamic_lds<Tag>(reinterpret_cast<const void*>(kernel), lds);
    if (attr_err != hipSuccess)
        return fail(FA_ERR_LAUNCH, "hipFuncSetAttribute(lds=%d): %s", lds, hipGetErrorString(attr_err));
    hipLaunchKernelGGL((fa::fa_fwd_kernel16<fa::TypeBF16, CAUSAL, true>), dim3(grid), dim3(512), lds, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FA_ERR_LAUNCH, "kernel launch failed: %s", hipGetErrorString(e));
    return FA_OK;
}

// head_dim > 64, fp8 Q, K and V, both products on MX-scaled fp8 MFMAs (fa_fwd_kernel8.hpp)
template <bool CAUSAL, bool WANT_LSE>
int launch8(const fa::FwdParams& p, int grid, hipStream_t stream)
{
    constexpr int lds = fa::kStages * 2 * fa::kBN8 * 128;
    auto* kernel = &fa::fa_fwd_kernel8<CAUSAL, WANT_LSE>;
    struct Tag {};
    const hipError_t attr_err = fa_capi::ensure_dynamic_lds<Tag>(reinterpret_cast<const void*>(kernel), lds);
    if (attr_err != hipSuccess)
        return fail(FA_ERR_LAUNCH, "hipFuncSetAttribute(lds=%d): %s", lds, hipGetErrorString(attr_err));
    hipLaunchKernelGGL((fa::fa_fwd_kernel8<CAUSAL, WANT_LSE>), dim3(grid), dim3(512), lds, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FA_ERR_LAUNCH, "kernel launch failed: %s", hipGetErrorString(e));
    return FA_OK;
}

// head_dim 144 .. 256: the plain 128-row kernel of fa_fwd_kernel_wide.hpp
template <class T, bool CAUSAL>
int launch_wide(const fa::FwdParams& p, int grid, hipStream_t stream)
{
    constexpr int lds = fa::kWideLds;
    auto* kernel = &fa::fa_fwd_kernel_wide<T, CAUSAL>;
    struct Tag {};
    const hipError_t attr_err = fa_capi::ensure_dynamic_lds<Tag>(reinterpret_cast<const void*>(kernel), lds);
    if (attr_err != hipSuccess)
        return fail(FA_ERR_LAUNCH, "hipFuncSetAttribute(lds=%d): %s", lds, hipGetErrorString(attr_err));
    hipLaunchKernelGGL((fa::fa_fwd_kernel_wide<T, CAUSAL>), dim3(grid), dim3(fa::kWideThreads), lds, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FA_ERR_LAUNCH, "kernel launch failed: %s", hipGetErrorString(e));
    return FA_OK;
}

template <class T, int D>
int launch_c(const fa::FwdParams& p, int grid, bool causal, int bm, hipStream_t s)
{
#if !defined(FA_MFMA32)
    if constexpr (kQB == 1 && D == 64) {
        if (bm == 128) return causal ? launch_rows128<T, D, true>(p, grid, s) : launch_rows128<T, D, false>(p, grid, s);
    }
    if constexpr (kQB == 1 && D == 128) {
#if defined(FA_ROWS128_MFMA32)      // A/B arm: the 128-row form of the 32x32x16 kernel instead (other rounding than the 256-row launches)
        if (bm == 128) return causal ? launch_rows128<T, D, true>(p, grid, s) : launch_rows128<T, D, false>(p, grid, s);
#else
        if (bm == 128) return causal ? launch16_rows128<T, true>(p, grid, s) : launch16_rows128<T, false>(p, grid, s);
#endif
    }
#endif
    (void)bm;
    // default: the 16x16x32 kernel; -DFA_MFMA32 (and the FA_QB variants) select the 32x32x16 kernel of fa_fwd_kernel.hpp
#if !defined(FA_MFMA32)
    if constexpr (kQB == 1) {
        // head_dim 128: always the 16x16 kernel (the 32x32 kernel is then not even instantiated)
        if constexpr (D == 128) {
            return causal ? launch16<T, true, D>(p, grid, s) : launch16<T, false, D>(p, grid, s);
        } else {
            // head_dim 64: the 16x16 kernel fits two workgroups per CU (128 VGPRs, 64 KiB of LDS: +11-12 % at cfg3's
            // batch and sequence); launches that do not fill the chip twice (cfg2: 128 workgroups) are latency-bound and
            // 5 % faster on the 32x32 kernel
            if (grid > 2 * fa_capi::device_cus())
                return causal ? launch16<T, true, D>(p, grid, s) : launch16<T, false, D>(p, grid, s);
            return causal ? launch<T, D, true>(p, grid, s) : launch<T, D, false>(p, grid, s);
        }
    } else {
        return causal ? launch<T, D, true>(p, grid, s) : launch<T, D, false>(p, grid, s);
    }
#else
    return causal ? launch<T, D, true>(p, grid, s) : launch<T, D, false>(p, grid, s);
#endif
}

// fp8 (OCP e4m3fn) -> bf16, exact (every e4m3 value is representable in bf16).  HBM-bound streaming pass:
// `bx` consecutive blocks per (batch, head) slice (flattened grid: no 65535 limit on B*H), a thread converts 16 bytes
// of one sequence row per step (16-byte loads, 2 x 16-byte stores); rows are D contiguous bytes addressed through the
// source strides, written contiguously.
typedef __attribute__((ext_vector_type(2))) float cvt_f32x2;
__global__ __launch_bounds__(256) void fp8_to_bf16_kernel(const unsigned char* __restrict__ src, unsigned short* __restrict__ dst,
                                                           int D, int H, int S, int bx, long long sb, long long sh, long long ss)
{
    const int bh = blockIdx.x / bx;
    const int blk = blockIdx.x - bh * bx;
    const int b = bh / H, h = bh - b * H;
    const unsigned char* sp = src + b * sb + h * sh;
    unsigned short* dp = dst + (long long)bh * S * D;
    const int cpr = D / 16;                                   // 16-byte chunks per row (8 or 4)
    const int total = S * cpr;
    for (int t = blk * blockDim.x + threadIdx.x; t < total; t += bx * blockDim.x) {
        const int row = t / cpr, ch = t - row * cpr;
        const fa::u32x4 in = *reinterpret_cast<const fa::u32x4*>(sp + (long long)row * ss + ch * 16);
        fa::u32x4 out[2];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const cvt_f32x2 lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)in[w], false);
            const cvt_f32x2 hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)in[w], true);
            out[w >> 1][2 * (w & 1)] = fa::TypeBF16::pack2(lo[0], lo[1]);
            out[w >> 1][2 * (w & 1) + 1] = fa::TypeBF16::pack2(hi[0], hi[1]);
        }
        fa::u32x4* o = reinterpret_cast<fa::u32x4*>(dp + (long long)t * 16);
        o[0] = out[0];
        o[1] = out[1];
    }
}

// causal launches pair query blocks (nqb-1-t, t) per workgroup for equal work -- unless single blocks, longest first, are
// expected to finish earlier (fa_capi::causal_unpaired: small grids, or a mostly empty last round of pairs)
bool unpaired_for(int B, int H, int S, bool causal, int bm = fa::kBM, bool lone = false)
{
    const long long nqb = (S + bm - 1) / bm;
#if defined(FA_FWD_EXPERIMENTS)         // experiment builds only (tools/build_variant.sh ... -DFA_FWD_EXPERIMENTS): FA_MI355_FORCE_UNPAIRED=0/1 overrides the estimate
    if (const char* e = std::getenv("FA_MI355_FORCE_UNPAIRED")) {
        if (e[0] == '0' || e[0] == '1') return causal && nqb > 1 && e[0] == '1';
    }
#endif
    return causal && fa_capi::causal_unpaired((long long)B * H, nqb, fa_capi::device_cus(), lone);
}

int grid_for(int B, int H, int S, bool causal, int bm = fa::kBM, bool lone = false)
{
    const long long bh = (long long)B * H;
    const long long nqb = (S + bm - 1) / bm;
    const long long per_head = (causal && !unpaired_for(B, H, S, causal, bm, lone)) ? (nqb + 1) / 2 : nqb;   // causal: one workgroup per pair of query blocks
    const long long g = fa_capi::grid_blocks(bh, per_head, fa_capi::head_split(bh, per_head));   // (virtual) heads padded to a multiple of 8 XCD groups
    return g > 0x7FFFFFFFll ? -1 : (int)g;
}

// wide heads (head_dim > 128): one 128-row query block per workgroup, no pairing
int grid_wide(int B, int H, int S, int* hsplit)
{
    const long long bh = (long long)B * H;
    const long long nqb = (S + fa::kBMW - 1) / fa::kBMW;
    const int hs = fa_capi::head_split(bh, nqb);
    if (hsplit) *hsplit = hs;
    const long long g = fa_capi::grid_blocks(bh, nqb, hs);
    return g > 0x7FFFFFFFll ? -1 : (int)g;
}

// Query rows per workgroup of a 16-bit launch: 256 (8 waves), or 128 (4 waves, head_dim <= 64 only) when the 256-row grid
// would leave a quarter or more of the CUs (256 on a whole MI355X) without a workgroup -- small launches are latency-bound, and a CU that holds
// no workgroup contributes nothing (cfg2: 128 workgroups of 256 rows -> 256 of 128 rows)
int rows_per_wg(int B, int H, int S, int D, bool causal)
{
#if defined(FA_MFMA32) || FA_QB != 1 || defined(FA_NO_ROWS128)
    (void)B; (void)H; (void)S; (void)D; (void)causal;
    return fa::kBM;
#else
    if (S <= 128) return fa::kBM;
#if defined(FA_FWD_EXPERIMENTS)         // experiment builds only: FA_MI355_ROWS=128/256 overrides the rule
    if (const char* e = std::getenv("FA_MI355_ROWS")) {
        if (e[0] == '1') return 128;
        if (e[0] == '2') return fa::kBM;
    }
#endif
    const int g = grid_for(B, H, S, causal);
    if (D <= 64) return (g > 0 && 4 * g <= 3 * fa_capi::device_cus()) ? 128 : fa::kBM;
    // head_dim 128 (round 4: the few-head shards of a strong split, e.g. 2 of cfg4's 16 heads per GPU): 128-row workgroups of the
    // 32x32x16 kernel (4 waves, one per SIMD) once the 256-row grid leaves at least half of the CUs without a workgroup -- a lone
    // wave per SIMD runs a tile at ~0.6 of a pair's rate, so the chip must at least double its busy CUs for that to pay
    return (g > 0 && 2 * g <= fa_capi::device_cus()) ? 128 : fa::kBM;
#endif
}

}  // namespace

extern "C" {

int fa_version(void) { return FA_VERSION; }

const char* fa_last_error(void) { return g_err; }

int fa_supported(int dtype, int head_dim)
{
    if (dtype != FA_DTYPE_BF16 && dtype != FA_DTYPE_FP16 && dtype != FA_DTYPE_FP8_E4M3) return 0;
    // every head_dim the reference accepts (D % 16 == 0, D <= 128: FA2-triton.py:178; its native dispatcher: 32, 64, 128,
    // flash_attn_cutlass.cu:530-543): D <= 64 runs on the head_dim-64 kernel, larger on the head_dim-128 kernel, with the
    // columns past D read as zeros by the hardware's buffer bounds check and never stored
    // ... and, beyond the reference, 144 .. 256 for the 16-bit types (forward only; fa_fwd_kernel_wide.hpp)
    const int top = (dtype == FA_DTYPE_FP8_E4M3) ? 128 : fa::kWideD;
    return (head_dim >= 16 && head_dim <= top && head_dim % 16 == 0) ? 1 : 0;
}

int fa_fwd_launch_info(int B, int H, int S, int D, int dtype, int causal, int* grid, int* block, int* lds_bytes)
{
    if (!fa_supported(dtype, D)) return fail(FA_ERR_BAD_HEAD_DIM, "unsupported (dtype=%d, head_dim=%d)", dtype, D);
    if (B < 0 || H < 0 || S < 0) return fail(FA_ERR_BAD_SHAPE, "negative shape");
    if (D > 128) {
        if (grid) *grid = grid_wide(B, H, S, nullptr);
        if (block) *block = fa::kWideThreads;
        if (lds_bytes) *lds_bytes = fa::kWideLds;
        return FA_OK;
    }
    const int bm = (dtype == FA_DTYPE_FP8_E4M3 && D > 64) ? fa::kBM : rows_per_wg(B, H, S, D, causal != 0);
    if (grid) *grid = grid_for(B, H, S, causal != 0, bm, bm == 128 && D > 64);
    if (block) *block = bm == 128 ? 256 : kThreads;
    if (lds_bytes) *lds_bytes = (D > 64) ? fa::lds_bytes<128>() : fa::lds_bytes<64>();
    return FA_OK;
}

int fa_fwd(const void* q, const void* k, const void* v, void* o, float* lse,
           int B, int H, int S, int D,
           const int64_t* q_strides, const int64_t* k_strides,
           const int64_t* v_strides, const int64_t* o_strides,
           int dtype, int causal, float softmax_scale,
           const float* descale, void* stream)
{
    return fa_fwd_ex(q, k, v, o, lse, B, H, H, S, S, D, q_strides, k_strides, v_strides, o_strides,
                     dtype, causal, softmax_scale, descale, stream);
}

int fa_fwd_ex(const void* q, const void* k, const void* v, void* o, float* lse,
              int B, int H, int H_kv, int S, int S_k, int D,
              const int64_t* q_strides, const int64_t* k_strides,
              const int64_t* v_strides, const int64_t* o_strides,
              int dtype, int causal, float softmax_scale,
              const float* descale, void* stream)
{
    g_err[0] = 0;
    if (dtype == FA_DTYPE_FP8_E4M3)
        return fail(FA_ERR_BAD_DTYPE, "fp8 inputs need a workspace: call fa_fwd_fp8");
    if (dtype != FA_DTYPE_BF16 && dtype != FA_DTYPE_FP16 && dtype != FA_DTYPE_FP8_E4M3)
        return fail(FA_ERR_BAD_DTYPE, "unknown dtype code %d", dtype);
    if (!fa_supported(dtype, D))
        return fail(FA_ERR_BAD_HEAD_DIM, "head_dim %d not supported (need D %% 16 == 0 and 16 <= D <= 256)", D);
    if (B < 0 || H < 0 || S < 0) return fail(FA_ERR_BAD_SHAPE, "negative shape B=%d H=%d S=%d", B, H, S);
    if (B == 0 || H == 0 || S == 0) return FA_OK;        // empty problem: nothing to do
    if (H_kv <= 0 || H % H_kv != 0)
        return fail(FA_ERR_BAD_SHAPE, "H=%d query heads are not a multiple of H_kv=%d key/value heads", H, H_kv);
    if (S_k <= 0) return fail(FA_ERR_BAD_SHAPE, "S_k=%d: no keys", S_k);
    if (!q || !k || !v || !o) return fail(FA_ERR_NULL_PTR, "null tensor pointer");

    fa::FwdParams p;
    memset(&p, 0, sizeof(p));
    p.q = q; p.k = k; p.v = v; p.o = o; p.lse = lse;
    p.B = B; p.H = H; p.S = S; p.Sk = S_k;
    p.G = H / H_kv;
    p.dv = D;
    const bool wide = D > 128;
    const int bm = wide ? fa::kBMW : rows_per_wg(B, H, S, D, causal != 0);
    p.nqb = (S + bm - 1) / bm;
    const bool lone = bm == 128 && D > 64;           // 128-row form of the head_dim-128 kernel: one wave per SIMD
    p.unpaired = wide ? 1 : (unpaired_for(B, H, S, causal != 0, bm, lone) ? 1 : 0);
    {
        const long long nqb_ = (S + bm - 1) / bm;
        p.hsplit = fa_capi::head_split((long long)B * H, (causal != 0 && !p.unpaired) ? (nqb_ + 1) / 2 : nqb_);
    }
    p.bh = B * H;
    if (!set_strides(q_strides, H, S, D, p.q_sb, p.q_sh, p.q_ss) ||
        !set_strides(k_strides, H_kv, S_k, D, p.k_sb, p.k_sh, p.k_ss) ||
        !set_strides(v_strides, H_kv, S_k, D, p.v_sb, p.v_sh, p.v_ss) ||
        !set_strides(o_strides, H, S, D, p.o_sb, p.o_sh, p.o_ss))
        return fail(FA_ERR_BAD_STRIDE, "strides must be non-negative with seq stride >= head_dim");
    const long long esz = 2;
    const long long strides[] = {p.q_sb, p.q_sh, p.q_ss, p.k_sb, p.k_sh, p.k_ss,
                                 p.v_sb, p.v_sh, p.v_ss, p.o_sb, p.o_sh, p.o_ss};
    for (long long s : strides)
        if ((s * esz) % 16 != 0) return fail(FA_ERR_BAD_STRIDE, "stride %lld elements is not 16-byte aligned", s);
    const void* ptrs[] = {q, k, v, o};
    for (const void* ptr : ptrs)
        if (reinterpret_cast<uintptr_t>(ptr) % 16 != 0) return fail(FA_ERR_BAD_STRIDE, "tensor base pointer not 16-byte aligned");
    // 32-bit buffer offsets inside one (batch, head) slice, with room for two prefetched tiles
    const long long max_ss = std::max(std::max(p.q_ss, p.k_ss), std::max(p.v_ss, p.o_ss));
    if (((long long)std::max(S, S_k) + 4 * fa::kBN) * max_ss * esz >= (1ll << 31))
        return fail(FA_ERR_TOO_LARGE, "one (batch, head) slice spans >= 2 GiB (S=%d, seq stride=%lld)", std::max(S, S_k), max_ss);

    const float scale = (softmax_scale > 0.f) ? softmax_scale : 1.0f / std::sqrt((float)D);
    // descale (used by fa_fwd_fp8 on its converted tensors): q and k scales fold into the softmax scale,
    // the v scale multiplies the output
    const float dq = descale ? descale[0] : 1.f, dkk = descale ? descale[1] : 1.f, dvv = descale ? descale[2] : 1.f;
    p.scale = scale * dq * dkk;
    p.scale_log2 = p.scale * 1.4426950408889634f;
    p.out_scale = dvv;

    const int grid = wide ? grid_wide(B, H, S, nullptr) : grid_for(B, H, S, causal != 0, bm, lone);
    if (grid <= 0) return fail(FA_ERR_TOO_LARGE, "grid too large");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool c = causal != 0;
    if (wide) {
        if (dtype == FA_DTYPE_BF16) return c ? launch_wide<fa::TypeBF16, true>(p, grid, s) : launch_wide<fa::TypeBF16, false>(p, grid, s);
        return c ? launch_wide<fa::TypeF16, true>(p, grid, s) : launch_wide<fa::TypeF16, false>(p, grid, s);
    }
    if (dtype == FA_DTYPE_BF16)
        return D > 64 ? launch_c<fa::TypeBF16, 128>(p, grid, c, bm, s) : launch_c<fa::TypeBF16, 64>(p, grid, c, bm, s);
    if (dtype == FA_DTYPE_FP16)
        return D > 64 ? launch_c<fa::TypeF16, 128>(p, grid, c, bm, s) : launch_c<fa::TypeF16, 64>(p, grid, c, bm, s);
    return fail(FA_ERR_BAD_DTYPE, "dtype %d not compiled", dtype);
}

size_t fa_fp8_workspace_bytes(int B, int H, int S, int D)
{
    if (B <= 0 || H <= 0 || S <= 0 || D <= 0) return 0;
#if defined(FA_FP8_CONVERT_ALL) || defined(FA_MFMA32) || FA_QB != 1
    const size_t copies = 3;
#elif defined(FA_FP8_PV_BF16)
    const size_t copies = D > 64 ? 1 : 3;               // round-1 path: only V is converted (Q, K feed fp8 MFMAs)
#else
    const size_t copies = D > 64 ? 0 : 3;               // head_dim > 64: nothing is converted, all three tensors feed fp8 MFMAs
#endif
    return copies * B * H * S * D * 2;                  // bf16 copies
}

int fa_fwd_fp8(const void* q, const void* k, const void* v, void* o, float* lse,
               int B, int H, int S, int D,
               const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides, const int64_t* o_strides,
               int causal, float softmax_scale, const float* descale,
               void* workspace, size_t workspace_bytes, void* stream)
{
    return fa_fwd_fp8_ex(q, k, v, o, lse, B, H, H, S, S, D, q_strides, k_strides, v_strides, o_strides,
                         causal, softmax_scale, descale, workspace, workspace_bytes, stream);
}

int fa_fwd_fp8_ex(const void* q, const void* k, const void* v, void* o, float* lse,
                  int B, int H, int H_kv, int S, int S_k, int D,
                  const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides, const int64_t* o_strides,
                  int causal, float softmax_scale, const float* descale,
                  void* workspace, size_t workspace_bytes, void* stream)
{
    g_err[0] = 0;
    if (!fa_supported(FA_DTYPE_FP8_E4M3, D))
        return fail(FA_ERR_BAD_HEAD_DIM, "head_dim %d not supported (need D %% 16 == 0 and 16 <= D <= 128)", D);
    if (B < 0 || H < 0 || S < 0) return fail(FA_ERR_BAD_SHAPE, "negative shape B=%d H=%d S=%d", B, H, S);
    if (B == 0 || H == 0 || S == 0) return FA_OK;
    if (H_kv <= 0 || H % H_kv != 0)
        return fail(FA_ERR_BAD_SHAPE, "H=%d query heads are not a multiple of H_kv=%d key/value heads", H, H_kv);
    if (S_k <= 0) return fail(FA_ERR_BAD_SHAPE, "S_k=%d: no keys", S_k);
    if (!q || !k || !v || !o) return fail(FA_ERR_NULL_PTR, "null tensor pointer");
    // How the three fp8 tensors reach the matrix cores:
    //   head_dim > 64 (default): nothing is converted; Q K^T and P V both run on MX-scaled fp8 MFMAs (fa_fwd_kernel8.hpp),
    //                            no workspace is needed (fa_fp8_workspace_bytes = 0, `workspace` may be NULL);
    //   head_dim <= 64, or -DFA_FP8_CONVERT_ALL: an exact fp8 -> bf16 HIP pre-pass of all three, then the bf16 kernels;
    //   -DFA_FP8_PV_BF16 (round 1's path, kept for A/B): Q, K native on fp8 MFMAs, V converted, P V on bf16 MFMAs.
#if defined(FA_FP8_CONVERT_ALL) || defined(FA_MFMA32) || FA_QB != 1
    const bool native_qk = false, native_v = false;
#elif defined(FA_FP8_PV_BF16)
    const bool native_qk = D > 64, native_v = false;
#else
    const bool native_qk = D > 64, native_v = D > 64;
#endif
    const int heads[3] = {H, H_kv, H_kv};
    const int rows[3] = {S, S_k, S_k};
    const size_t one_q = (size_t)B * H * S * D * 2, one_kv = (size_t)B * H_kv * S_k * D * 2;   // bf16 copies
    const size_t need = native_v ? 0 : (native_qk ? one_kv : one_q + 2 * one_kv);
    if (need > 0) {
        if (!workspace) return fail(FA_ERR_NULL_PTR, "null workspace pointer (%zu bytes needed)", need);
        if (workspace_bytes < need)
            return fail(FA_ERR_BAD_SHAPE, "workspace too small: %zu < %zu bytes", workspace_bytes, need);
        if (reinterpret_cast<uintptr_t>(workspace) % 16 != 0) return fail(FA_ERR_BAD_STRIDE, "workspace not 16-byte aligned");
    }
    const void* src[3] = {q, k, v};
    const int64_t* strd[3] = {q_strides, k_strides, v_strides};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    long long st[3][3];
    for (int t = 0; t < 3; ++t) {
        if (!set_strides(strd[t], heads[t], rows[t], D, st[t][0], st[t][1], st[t][2])) return fail(FA_ERR_BAD_STRIDE, "bad strides");
        if (st[t][0] % 16 || st[t][1] % 16 || st[t][2] % 16 || reinterpret_cast<uintptr_t>(src[t]) % 16)
            return fail(FA_ERR_BAD_STRIDE, "fp8 tensors need 16-byte aligned rows");
    }
    char* w = static_cast<char*>(workspace);
    char* wt[3] = {w, w + one_q, w + one_q + one_kv};   // the workspace holds the converted tensors back to back
    if (native_qk) wt[2] = w;
    for (int t = native_v ? 3 : (native_qk ? 2 : 0); t < 3; ++t) {
        const int per_slice = rows[t] * (D / 16);
        const int bx = std::max(1, std::min((per_slice + 255) / 256, 64));
        const long long blocks = (long long)bx * B * heads[t];
        if (blocks > 0x7FFFFFFFll) return fail(FA_ERR_TOO_LARGE, "fp8 pre-pass grid too large");
        hipLaunchKernelGGL(fp8_to_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, s,
                           reinterpret_cast<const unsigned char*>(src[t]),
                           reinterpret_cast<unsigned short*>(wt[t]),
                           D, heads[t], rows[t], bx, st[t][0], st[t][1], st[t][2]);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(FA_ERR_LAUNCH, "fp8 conversion launch failed: %s", hipGetErrorString(e));
    }
    if (!native_qk)
        return fa_fwd_ex(wt[0], wt[1], wt[2], o, lse, B, H, H_kv, S, S_k, D, nullptr, nullptr, nullptr, o_strides,
                         FA_DTYPE_BF16, causal, softmax_scale, descale, stream);

    fa::FwdParams p;
    memset(&p, 0, sizeof(p));
    p.q = q; p.k = k; p.v = native_v ? v : w; p.o = o; p.lse = lse;
    p.B = B; p.H = H; p.S = S; p.Sk = S_k; p.dv = D;
    p.G = H / H_kv;
    p.nqb = (S + fa::kBM - 1) / fa::kBM;
    p.unpaired = unpaired_for(B, H, S, causal != 0) ? 1 : 0;
    {
        const long long nqb_ = (S + fa::kBM - 1) / fa::kBM;
        p.hsplit = fa_capi::head_split((long long)B * H, (causal != 0 && !p.unpaired) ? (nqb_ + 1) / 2 : nqb_);
    }
    p.bh = B * H;
    p.q_sb = st[0][0]; p.q_sh = st[0][1]; p.q_ss = st[0][2];
    p.k_sb = st[1][0]; p.k_sh = st[1][1]; p.k_ss = st[1][2];
    if (native_v) { p.v_sb = st[2][0]; p.v_sh = st[2][1]; p.v_ss = st[2][2]; }       // fp8 V in place: byte strides
    else { p.v_ss = D; p.v_sh = (long long)S_k * D; p.v_sb = (long long)H_kv * S_k * D; }   // the bf16 copy, contiguous (element strides)
    if (!set_strides(o_strides, H, S, D, p.o_sb, p.o_sh, p.o_ss)) return fail(FA_ERR_BAD_STRIDE, "bad output strides");
    if ((p.o_sb * 2) % 16 || (p.o_sh * 2) % 16 || (p.o_ss * 2) % 16 || reinterpret_cast<uintptr_t>(o) % 16)
        return fail(FA_ERR_BAD_STRIDE, "output rows must be 16-byte aligned");
    // 32-bit buffer offsets inside one (batch, head) slice, with room for the prefetched tiles (byte strides; V, O 16-bit
    // where they are not fp8)
    const long long max_ss = std::max(std::max(p.q_ss, p.k_ss), std::max((native_v ? 1 : 2) * p.v_ss, 2 * p.o_ss));
    if (((long long)std::max(S, S_k) + 4 * fa::kBN8) * max_ss >= (1ll << 31))
        return fail(FA_ERR_TOO_LARGE, "one (batch, head) slice spans >= 2 GiB (S=%d)", std::max(S, S_k));
    const float scale = (softmax_scale > 0.f) ? softmax_scale : 1.0f / std::sqrt((float)D);
    const float dq = descale ? descale[0] : 1.f, dkk = descale ? descale[1] : 1.f, dvv = descale ? descale[2] : 1.f;
    p.scale = scale * dq * dkk;
    p.scale_log2 = p.scale * 1.4426950408889634f;
    p.out_scale = dvv;
    const int grid = grid_for(B, H, S, causal != 0);
    if (grid <= 0) return fail(FA_ERR_TOO_LARGE, "grid too large");
    if (native_v) {
        if (lse) return causal ? launch8<true, true>(p, grid, s) : launch8<false, true>(p, grid, s);
        return causal ? launch8<true, false>(p, grid, s) : launch8<false, false>(p, grid, s);
    }
    return causal ? launch16_qk8<true>(p, grid, s) : launch16_qk8<false>(p, grid, s);
}

#if defined(FA_STAMP)
// diagnostic build (tools/stamps.py): copy the stamp sums of the last head_dim-128 forward launches to the host
extern "C" int fa_debug_read_stamps(void* dst, size_t bytes)
{
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(fa::g_fa_stamp), bytes, 0, hipMemcpyDeviceToHost);
}
#endif

// 1 if the fp8 entry runs P V on fp8 MFMAs too (head_dim > 64), 0 if that product runs on bf16 MFMAs
int fa_fp8_pv_native(void)
{
#if defined(FA_FP8_CONVERT_ALL) || defined(FA_MFMA32) || FA_QB != 1 || defined(FA_FP8_PV_BF16)
    return 0;
#else
    return 1;
#endif
}

int fa_fwd_dispatch(const void* Q, const void* K, const void* V, void* O,
                    int batch_size, int num_heads, int seq_len, int head_dim,
                    int dtype, void* stream)
{
    return fa_fwd(Q, K, V, O, nullptr, batch_size, num_heads, seq_len, head_dim,
                  nullptr, nullptr, nullptr, nullptr, dtype, /*causal=*/0, /*scale=*/0.f, nullptr, stream);
}

}  // extern "C"
