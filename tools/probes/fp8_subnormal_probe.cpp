// Probe: does the fp8 path keep e4m3 subnormals (2^-9 .. 2^-7)?  (1) v_cvt_pk_fp8_f32 of values around them, (2) the scaled MFMA
// v_mfma_scale_f32_16x16x128_f8f6f4 fed subnormal bytes on either operand against ones on the other (expected: 128 x value).
// build: hipcc --offload-arch=gfx950 -O2 -o build/fp8_subnormal_probe tools/probes/fp8_subnormal_probe.cpp ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void probe(const float* in, int* out, int n, float* mf, const int* words)
{
    const int i = threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_cvt_pk_fp8_f32(in[i], in[i], 0, false) & 0xFF;
    for (int t = 0; t < 4; ++t) {
        const int w = words[t], one = words[4];        // (from memory: keeps the operands out of the constant folder)
        v8i a, b, c1;
        for (int j = 0; j < 8; ++j) { a[j] = one; b[j] = w; c1[j] = one; }
        v4f z = {0.f, 0.f, 0.f, 0.f};
        v4f r0 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, z, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);    // subnormal on B
        v4f r1 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b, c1, z, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);   // subnormal on A
        mf[8 * i + 2 * t] = r0[0]; mf[8 * i + 2 * t + 1] = r1[0];      // every lane stores: the MFMA must run under a full EXEC
    }
}

int main()
{
    std::vector<float> h;
    for (int e = -12; e <= -5; ++e) for (float m : {1.0f, 1.25f, 1.5f, 1.75f}) h.push_back(ldexpf(m, e));
    float *di, *dm; int* dout;
    hipMalloc(&di, h.size() * 4); hipMalloc(&dout, h.size() * 4); hipMalloc(&dm, 64 * 32);
    hipMemcpy(di, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const int words[5] = {0x01010101, 0x02020202, 0x04040404, 0x08080808, 0x38383838};   // 2^-9, 2^-8, 2^-7, 2^-6 (smallest normal), 1.0
    int* dw; hipMalloc(&dw, 20); hipMemcpy(dw, words, 20, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, di, dout, (int)h.size(), dm, dw);
    std::vector<int> o(h.size()); float m[8];
    hipMemcpy(o.data(), dout, h.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(m, dm, 32, hipMemcpyDeviceToHost);
    for (size_t i = 0; i < h.size(); ++i) printf("%g*2^%d -> 0x%02X\n", h[i] / ldexpf(1.f, (int)floorf(log2f(h[i]))), (int)floorf(log2f(h[i])), o[i]);
    const char* nm[4] = {"2^-9", "2^-8", "2^-7", "2^-6"};
    for (int t = 0; t < 4; ++t) printf("mfma sum of 128 x %s: on B %g, on A %g (expected %g)\n", nm[t], m[2 * t], m[2 * t + 1], 128.0 * ldexp(1.0, t - 9));
    return 0;
}
