// Diagnostic entry points of the C ABI (include/fa_mi355.h): nothing here is on the attention path.
//
// fa_diag_mfma_loop: what the matrix cores of THIS device deliver on random operands when nothing else competes -- the
// attainable ceiling bench.py prints next to the nominal dense peak (`roofline.attainable`).  On MI355X a dense bf16 MFMA
// stream on N(0,1) data runs against the board power limit, not against the 2.4 GHz the nominal peak assumes
// (profiles/r3_power_*.txt), so the nominal peak cannot be reached by any kernel on such data.
// No reference counterpart: the reference's protocol only asks for the fraction of peak (code/README.md:41-43).
#include <hip/hip_runtime.h>
#include <cstdint>

#include "../../include/fa_mi355.h"
#include "fa_capi_common.hpp"

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) int i32x8;

// KIND 0: v_mfma_f32_16x16x32_bf16, 1: _f16, 2: v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 x e4m3, unit scales).
// One workgroup of 8 waves per CU (two waves per SIMD, as the attention kernels run); per wave 16 independent accumulator
// tiles (the 64 accumulator registers of the attention kernel's O^T), 64 MFMAs per loop trip over 4 + 4 operand fragments
// held in registers (random data from `ops`), nothing else in the loop.
template <int KIND>
__global__ __launch_bounds__(512, 2) void fa_diag_mfma_kernel(const u32x4* __restrict__ ops, float* __restrict__ sink, int iters)
{
    const int tid = threadIdx.x;
    u32x4 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = ops[(size_t)tid * 8 + i];
        b[i] = ops[(size_t)tid * 8 + 4 + i];
    }
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if constexpr (KIND == 0)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[(i + r) & 3]),
                                                                      __builtin_bit_cast(bf16x8, b[i & 3]), acc[i], 0, 0, 0);
                else if constexpr (KIND == 1)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[(i + r) & 3]),
                                                                     __builtin_bit_cast(f16x8, b[i & 3]), acc[i], 0, 0, 0);
                else {
                    const u32x4 a0 = a[(i + r) & 3], a1 = a[(i + r + 1) & 3], b0 = b[i & 3], b1 = b[(i + 1) & 3];
                    const i32x8 av = {(int)a0[0], (int)a0[1], (int)a0[2], (int)a0[3], (int)a1[0], (int)a1[1], (int)a1[2], (int)a1[3]};
                    const i32x8 bv = {(int)b0[0], (int)b0[1], (int)b0[2], (int)b0[3], (int)b1[0], (int)b1[1], (int)b1[2], (int)b1[3]};
                    acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, acc[i], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
                }
            }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 1.2345678e-30f) sink[0] = s;          // (keeps the loop alive; never true in practice)
}

}  // namespace

extern "C" {

int fa_device_cus(void) { return fa_capi::device_cus(); }

int fa_build_is_default(void) { return FA_BUILD_NON_DEFAULT ? 0 : 1; }

int fa_diag_mfma_loop(int dtype, int iters, const void* operands, float* sink, double* flops_out, void* stream)
{
    fa_capi::g_err[0] = 0;
    if (!operands || !sink) return fa_capi::fail(FA_ERR_NULL_PTR, "null operand / sink pointer");
    if (iters <= 0) return fa_capi::fail(FA_ERR_BAD_SHAPE, "iters=%d", iters);
    if (reinterpret_cast<uintptr_t>(operands) % 16 != 0) return fa_capi::fail(FA_ERR_BAD_STRIDE, "operands not 16-byte aligned");
    const int grid = fa_capi::device_cus();
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const u32x4* ops = static_cast<const u32x4*>(operands);
    double flops_per_mfma;
    if (dtype == FA_DTYPE_BF16) {
        hipLaunchKernelGGL(fa_diag_mfma_kernel<0>, dim3(grid), dim3(512), 0, s, ops, sink, iters);
        flops_per_mfma = 2.0 * 16 * 16 * 32;
    } else if (dtype == FA_DTYPE_FP16) {
        hipLaunchKernelGGL(fa_diag_mfma_kernel<1>, dim3(grid), dim3(512), 0, s, ops, sink, iters);
        flops_per_mfma = 2.0 * 16 * 16 * 32;
    } else if (dtype == FA_DTYPE_FP8_E4M3) {
        hipLaunchKernelGGL(fa_diag_mfma_kernel<2>, dim3(grid), dim3(512), 0, s, ops, sink, iters);
        flops_per_mfma = 2.0 * 16 * 16 * 128;
    } else {
        return fa_capi::fail(FA_ERR_BAD_DTYPE, "unknown dtype code %d", dtype);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fa_capi::fail(FA_ERR_LAUNCH, "kernel launch failed: %s", hipGetErrorString(e));
    if (flops_out) *flops_out = (double)grid * 8.0 * (double)iters * 64.0 * flops_per_mfma;
    return FA_OK;
}

}  // extern "C"
