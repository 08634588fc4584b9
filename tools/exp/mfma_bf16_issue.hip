// Lab (round 5): issue rate of v_mfma_f32_16x16x32_bf16 against the distance between MFMAs that share an accumulator,
// and what independent VALU work between them costs.  One number per configuration: cycles per MFMA and SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
template <int NACC, int VALU>
__global__ __launch_bounds__(512) void k(float* out, int iters, unsigned seed) {
    unsigned s = seed + threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j) {
            s = s * 1664525u + 1013904223u; a[i][j] = (short)(0x3c00 + (s >> 24));
            s = s * 1664525u + 1013904223u; b[i][j] = (short)(0x3800 + (s >> 24));
        }
    f32x4 acc[NACC];
    for (int t = 0; t < NACC; ++t) acc[t] = f32x4{0, 0, 0, 0};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = 1.f + i + threadIdx.x * 1e-3f;
    const long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 48 / NACC; ++r)
#pragma unroll
            for (int t = 0; t < NACC; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(r + t) & 3], b[t & 3], acc[t], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < VALU; ++q) v[(t * VALU + q) & 7] = __builtin_fmaf(v[(t * VALU + q) & 7], 1.0001f, 0.5f);
                if (VALU) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, VALU, 0); }
            }
    }
    const long long c1 = clock64();
    float r = 0;
    for (int t = 0; t < NACC; ++t) r += acc[t][0] + acc[t][3];
    for (int i = 0; i < 8; ++i) r += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[256 * 512] = (float)(c1 - c0);
}
template <int NACC, int VALU>
void run(float* out, int threads) {
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<NACC, VALU>), dim3(256), dim3(threads), 0, 0, out, iters, 7u);
    hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<NACC, VALU>), dim3(256), dim3(threads), 0, 0, out, iters, 7u + i);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per_launch_us = ms * 1e3 / reps;
    const double mfma_per_simd = (double)iters * 48 * (threads / 256);      // waves per SIMD x MFMAs per wave
    const double tf = 2.0 * 16 * 16 * 32 * iters * 48.0 * (threads / 64) * 256 / (per_launch_us * 1e-6) / 1e12;
    printf("accumulators %2d  VALU per MFMA %d  waves/SIMD %d:  %8.1f us  %6.1f ns per MFMA and SIMD  %7.1f TFLOP/s bf16\n",
           NACC, VALU, threads / 256, per_launch_us, per_launch_us * 1e3 / mfma_per_simd, tf);
}
int main() {
    float* out; hipMalloc(&out, (256 * 512 + 16) * sizeof(float));
    for (int threads : {256, 512}) {
        run<16, 0>(out, threads); run<8, 0>(out, threads); run<4, 0>(out, threads); run<2, 0>(out, threads); run<1, 0>(out, threads);
        run<16, 1>(out, threads); run<16, 2>(out, threads); run<16, 4>(out, threads);
        run<2, 2>(out, threads); run<4, 2>(out, threads);
    }
    return 0;
}
