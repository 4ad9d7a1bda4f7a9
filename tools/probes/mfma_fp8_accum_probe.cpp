// Probe: how exactly does v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 e4m3 x e4m3, unit scales) add its 128 products?  One product is
// `big`, the other 127 are `small` (A = the values, B = 1.0): an exact fp32 sum gives big + 127 small.  A datapath that aligns
// the products to the largest one and keeps a limited number of bits below it shows up as a shortfall that grows with big/small.
// Also: the same sum split over two instructions (big alone, then the 127) to see whether the fp32 accumulator input is exact.
// build: hipcc --offload-arch=gfx950 -O2 -o build/mfma_fp8_accum_probe tools/probes/mfma_fp8_accum_probe.cpp ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>

typedef __attribute__((ext_vector_type(8))) int v8i;
typedef __attribute__((ext_vector_type(4))) float v4f;

__global__ void probe(const uint8_t* a_raw, const uint8_t* b_raw, float* d, float cin)
{
    const int lane = threadIdx.x;
    v8i a, b;
    memcpy(&a, a_raw + lane * 32, 32);
    memcpy(&b, b_raw + lane * 32, 32);
    v4f c = {cin, cin, cin, cin};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    for (int r = 0; r < 4; ++r) d[(4 * (lane >> 4) + r) * 16 + (lane & 15)] = c[r];
}

static uint8_t enc(double v)       // positive powers of two and small multiples: e4m3fn encoding by search
{
    uint8_t best = 0; double bd = 1e300;
    for (int byte = 0; byte < 0x7F; ++byte) {
        const int e = byte >> 3, m = byte & 7;
        const double x = e == 0 ? ldexp(m / 8.0, -6) : ldexp(1.0 + m / 8.0, e - 7);
        if (fabs(x - v) < bd) { bd = fabs(x - v); best = (uint8_t)byte; }
    }
    return best;
}

int main()
{
    uint8_t *da, *db; float* dd;
    hipMalloc(&da, 2048); hipMalloc(&db, 2048); hipMalloc(&dd, 1024);
    const double bigs[] = {256.0, 256.0, 256.0, 256.0, 256.0, 256.0, 448.0, 1.0, 1.0};
    const double smalls[] = {1.0, 0.125, 0.015625, 0.001953125, 0.0625, 0.03125, 0.001953125, 0.001953125, 0.015625};
    for (int t = 0; t < 9; ++t) for (int where = 0; where < 2; ++where) for (int split = 0; split < 2; ++split) {
        std::vector<uint8_t> ar(2048, enc(smalls[t])), br(2048, enc(1.0));
        // the big product sits at (lane group 0, byte 0) of every row -- or, `where` = 1, at (lane group 3, byte 31)
        for (int i = 0; i < 16; ++i) ar[(where ? 48 + i : i) * 32 + (where ? 31 : 0)] = enc(bigs[t]);
        float cin = 0.f;
        if (split) {                      // big goes in through the accumulator instead
            for (int i = 0; i < 16; ++i) ar[(where ? 48 + i : i) * 32 + (where ? 31 : 0)] = enc(smalls[t]);
            cin = (float)(bigs[t] - smalls[t]);
        }
        hipMemcpy(da, ar.data(), 2048, hipMemcpyHostToDevice);
        hipMemcpy(db, br.data(), 2048, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dd, cin);
        float got[256];
        hipMemcpy(got, dd, 1024, hipMemcpyDeviceToHost);
        const double want = bigs[t] + 127 * smalls[t];
        printf("big %g + 127 x %g (%s, %s): got %.9g, exact %.9g, shortfall of the small part %.3f %%\n", bigs[t], smalls[t],
               where ? "last k" : "first k", split ? "big via C" : "one instr", got[0], want, 100.0 * (want - got[0]) / (127 * smalls[t]));
    }
    return 0;
}
