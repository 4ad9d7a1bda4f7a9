// Probe of ds_read_b64_tr_b8 (8-bit transposed LDS read): which bytes does lane l receive when the lanes of a 16-lane group
// supply the addresses of an 8-row x 16-column byte block?  Hypothesis (by analogy with the 16-bit form): lane 2q+p of the
// group supplies the address of row q, columns 8p..8p+7; lane i receives column i, row r in byte r.
// build: hipcc --offload-arch=gfx950 -O2 -o build/tr8_probe tools/probes/ds_read_tr8_probe.cpp ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef int v2i __attribute__((ext_vector_type(2)));

__global__ void probe(uint8_t* out)
{
    __shared__ __attribute__((aligned(16))) uint8_t m[64 * 64];      // [row][64 columns]; element = (row << 4 | col & 15) for col < 16 ...
    const int lane = threadIdx.x;
    for (int i = lane; i < 64 * 64; i += 64) m[i] = (uint8_t)(((i / 64) & 15) << 4 | ((i % 64) & 15));
    __syncthreads();
    const int grp = lane >> 4, n = lane & 15;
    const int q = n >> 1, pp = n & 1;
    // group g reads rows 8g .. 8g+7, columns 16g' ... use columns 16*grp .. +15 to make groups distinguishable
    const uint8_t* addr = &m[(8 * grp + q) * 64 + 16 * grp + 8 * pp];
    v2i r = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i*)addr);
    uint8_t* o = out + lane * 8;
    for (int b = 0; b < 4; ++b) { o[b] = (uint8_t)(r[0] >> (8 * b)); o[4 + b] = (uint8_t)(r[1] >> (8 * b)); }
}

int main()
{
    uint8_t* d;
    hipMalloc(&d, 512);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    std::vector<uint8_t> h(512);
    hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane) {
        const int grp = lane >> 4, i = lane & 15;
        for (int b = 0; b < 8; ++b) {
            const int want_row = (8 * grp + b) & 15, want_col = i;          // column i of the block, row b
            const int got = h[lane * 8 + b];
            if (got != ((want_row << 4) | want_col)) ++bad;
        }
    }
    printf("hypothesis mismatches: %d / 512\n", bad);
    for (int lane : {0, 1, 2, 15, 16, 17, 33}) {
        printf("lane %2d:", lane);
        for (int b = 0; b < 8; ++b) printf(" (r%d,c%d)", h[lane * 8 + b] >> 4, h[lane * 8 + b] & 15);
        printf("\n");
    }
    return 0;
}
