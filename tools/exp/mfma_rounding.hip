// Probe (round 5): HOW does v_mfma_f32_16x16x32_bf16 round?  The bf16x6 split products chain 152 MFMAs per output at
// K = 4864; if the add into the fp32 accumulator truncates (instead of round-to-nearest-even) the error is BIASED and
// grows linearly with the chain length instead of as a random walk.
//   case 1: C = 1.0, ONE product of j * 2^-26 (j = -15..15): which way does 1 + j/8 ulp go?
//   case 2: C = 0, 32 products: 1.0 + 31 x (j * 2^-29): internal width of the 32-term sum
//   case 3: same as 1 for v_mfma_f32_16x16x4_f32
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_rounding tools/exp/mfma_rounding.hip && /tmp/mfma_rounding
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ short bf(float x) { return (short)(__float_as_uint(x) >> 16); }   // exact for the values used

// out[case][j+16]: D[0][col] for col = 0
__global__ void probe(float* out) {
    const int lane = threadIdx.x;
    const int r = lane & 15, kg = lane >> 4;          // A: row r, k = 8*kg..+7 ; B: col r, k = 8*kg..+7
    for (int j = -15; j <= 15; ++j) {
        // ---- case 1: one product j * 2^-26 on top of C = 1.0
        {
            bf16x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
            if (kg == 0) { a[0] = bf(ldexpf(1.0f, -13)); b[0] = bf(ldexpf((float)j, -13)); }
            f32x4 c = {1.0f, 1.0f, 1.0f, 1.0f};
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
            if (lane == 0) out[0 * 32 + j + 16] = c[0];
        }
        // ---- case 2: C = 0; products: k = 0 -> 1.0, k = 1..31 -> j * 2^-29 each (sum = 31 j 2^-29)
        {
            bf16x8 a, b;
            for (int q = 0; q < 8; ++q) {
                const int k = 8 * kg + q;
                a[q] = bf(k == 0 ? 1.0f : ldexpf(1.0f, -14));
                b[q] = bf(k == 0 ? 1.0f : ldexpf((float)j, -15));
            }
            f32x4 c = {0, 0, 0, 0};
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
            if (lane == 0) out[1 * 32 + j + 16] = c[0];
        }
        // ---- case 3: fp32 MFMA, one product j * 2^-26 on C = 1.0
        {
            float a = (kg == 0) ? ldexpf(1.0f, -13) : 0.0f, b = (kg == 0) ? ldexpf((float)j, -13) : 0.0f;
            f32x4 c = {1.0f, 1.0f, 1.0f, 1.0f};
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
            if (lane == 0) out[2 * 32 + j + 16] = c[0];
        }
        // ---- case 4: C = 1.0 and 32 products of j * 2^-31 each (sum = j * 2^-26): are the small terms summed before C?
        {
            bf16x8 a, b;
            for (int q = 0; q < 8; ++q) { a[q] = bf(ldexpf(1.0f, -15)); b[q] = bf(ldexpf((float)j, -16)); }
            f32x4 c = {1.0f, 1.0f, 1.0f, 1.0f};
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
            if (lane == 0) out[3 * 32 + j + 16] = c[0];
        }
    }
}

int main() {
    float* d;
    hipMalloc(&d, 4 * 32 * sizeof(float));
    hipMemset(d, 0, 4 * 32 * sizeof(float));
    probe<<<1, 64>>>(d);
    float h[4 * 32];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[4] = {"bf16 MFMA: C=1 + one product j*2^-26 (ulp = 2^-23 = 8 units)",
                            "bf16 MFMA: C=0, 1.0 + 31 products of j*2^-29 (exact = 1 + 31j/64 ulp)",
                            "fp32 MFMA: C=1 + one product j*2^-26",
                            "bf16 MFMA: C=1 + 32 products of j*2^-31 (sum j*2^-26)"};
    for (int c = 0; c < 4; ++c) {
        printf("%s\n   j: result-1 in ulps (2^-23) | round-to-nearest-even would give\n", names[c]);
        for (int j = -15; j <= 15; ++j) {
            double exact = (c == 1) ? 31.0 * j / 64.0 : j / 8.0;      // in ulps of 1.0 (upper binade)
            double got = ((double)h[c * 32 + j + 16] - 1.0) / ldexp(1.0, -23);
            // below 1.0 the ulp is 2^-24: express RN on the true grid
            float rn = (float)(1.0 + exact * ldexp(1.0, -23));
            printf("  %3d: exact %+8.4f  got %+6.2f  rn %+6.2f %s\n", j, exact, got, ((double)rn - 1.0) / ldexp(1.0, -23),
                   got == ((double)rn - 1.0) / ldexp(1.0, -23) ? "" : "  <-- differs");
        }
    }
    return 0;
}
