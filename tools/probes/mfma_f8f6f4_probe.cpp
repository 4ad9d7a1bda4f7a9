// Probe of the operand lane map of v_mfma_scale_f32_16x16x128_f8f6f4 with fp8 (e4m3) A and B, unit scales.
// For each hypothesis H (how (lane, byte) of the 32-byte fragments map to (row, k) / (k, col)) the host fills A and B
// with small exact integers through H, runs one MFMA per wave and compares D (the standard 16x16 C/D map) with A.B.
// build: hipcc --offload-arch=gfx950 -O2 -o build/mfma_probe tools/probes/mfma_f8f6f4_probe.cpp ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) int v8i;
typedef __attribute__((ext_vector_type(4))) float v4f;

__global__ void probe(const uint8_t* a_raw, const uint8_t* b_raw, float* d)
{
    const int lane = threadIdx.x;
    v8i a, b;
    memcpy(&a, a_raw + lane * 32, 32);
    memcpy(&b, b_raw + lane * 32, 32);
    v4f c = {0.f, 0.f, 0.f, 0.f};
    // cbsz = 0 (A fp8 e4m3), blgp = 0 (B fp8 e4m3); scales: E8M0 127 = 1.0 in every byte
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    for (int r = 0; r < 4; ++r) d[(4 * (lane >> 4) + r) * 16 + (lane & 15)] = c[r];     // D[row 4g + r][col i]
}

static uint8_t e4m3(int v)   // exact encoding of small integers -8..8 (OCP e4m3fn)
{
    if (v == 0) return 0;
    const uint8_t sign = v < 0 ? 0x80 : 0;
    int m = v < 0 ? -v : v;
    int e = 0;
    while ((m >> (e + 1)) != 0) ++e;          // m in [2^e, 2^(e+1))
    const int frac = ((m << 3) >> e) & 7;     // 3 mantissa bits (exact for m <= 15)
    return sign | (uint8_t)((e + 7) << 3) | (uint8_t)frac;
}

int main()
{
    std::vector<int> A(16 * 128), B(128 * 16);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 128; ++k) A[i * 128 + k] = ((i * 7 + k * 3) % 9) - 4;
    for (int k = 0; k < 128; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = ((k * 5 + j * 11) % 7) - 3;
    std::vector<float> ref(256);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        long s = 0;
        for (int k = 0; k < 128; ++k) s += (long)A[i * 128 + k] * B[k * 16 + j];
        ref[i * 16 + j] = (float)s;
    }
    uint8_t *da, *db; float* dd;
    hipMalloc(&da, 2048); hipMalloc(&db, 2048); hipMalloc(&dd, 1024);
    const char* names[] = {"H1: k = 32*(lane>>4) + byte",
                           "H2: k = 64*(byte>>4) + 16*(lane>>4) + (byte&15)",
                           "H3: k = 16*(byte>>3)... k = 8*(lane>>4) + (byte&7) + 32*(byte>>3)",
                           "H4: k = 4*(lane>>4)... k = 16*(lane>>4) + (byte&15) + 64*(byte>>4) (same as H2)"};
    for (int h = 0; h < 3; ++h) {
        std::vector<uint8_t> ar(2048), br(2048);
        for (int lane = 0; lane < 64; ++lane) for (int byte = 0; byte < 32; ++byte) {
            const int g = lane >> 4, i = lane & 15;
            int k;
            if (h == 0) k = 32 * g + byte;
            else if (h == 1) k = 64 * (byte >> 4) + 16 * g + (byte & 15);
            else k = 8 * g + (byte & 7) + 32 * (byte >> 3);
            ar[lane * 32 + byte] = e4m3(A[i * 128 + k]);
            br[lane * 32 + byte] = e4m3(B[k * 16 + i]);
        }
        hipMemcpy(da, ar.data(), 2048, hipMemcpyHostToDevice);
        hipMemcpy(db, br.data(), 2048, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dd);
        std::vector<float> got(256);
        hipMemcpy(got.data(), dd, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int x = 0; x < 256; ++x) bad += got[x] != ref[x];
        printf("%-70s mismatches %d / 256 (got[0]=%g ref[0]=%g, got[17]=%g ref[17]=%g)\n", names[h], bad, got[0], ref[0], got[17], ref[17]);
    }
    return 0;
}
