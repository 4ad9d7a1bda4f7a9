// Probe of v_cvt_pk_fp8_f32 on gfx950 (OCP e4m3fn) at the edges: what do values past 448, infinities and tiny values turn into
// under the default float mode (FP16_OVFL = 0)?  The fp8 attention kernel relies on "too large -> NaN (0x7F)", which then
// poisons the MFMA row sums and sends the workgroup to its exact loop.
// build: hipcc --offload-arch=gfx950 -O2 -o build/cvt_fp8_probe tools/probes/cvt_fp8_probe.cpp ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>

__global__ void probe(const float* in, int* out, int n)
{
    const int i = threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_cvt_pk_fp8_f32(in[i], in[i], 0, false) & 0xFF;
}

int main()
{
    std::vector<float> h = {0.f, 1.f, 240.f, 448.f, 463.9f, 464.f, 465.f, 480.f, 512.f, 1e6f, INFINITY, NAN, -500.f,
                            0.001953125f /*2^-9*/, 0.0009765625f /*2^-10*/, 0.00146f, 0.0005f, 1.0625f, 1.1875f};
    float* di; int* dout;
    hipMalloc(&di, h.size() * 4); hipMalloc(&dout, h.size() * 4);
    hipMemcpy(di, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, di, dout, (int)h.size());
    std::vector<int> o(h.size());
    hipMemcpy(o.data(), dout, h.size() * 4, hipMemcpyDeviceToHost);
    for (size_t i = 0; i < h.size(); ++i) printf("%14g -> 0x%02X\n", h[i], o[i]);
    return 0;
}
