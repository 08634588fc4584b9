// Experiment: sustained fp32 MFMA rate vs operand data (DVFS) and instruction shape.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int SHAPE, int RANDOM>
__global__ __launch_bounds__(512) void k(float* out, int iters, unsigned seed) {
    unsigned s = seed + threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    float a[8], b[4];
    for (int i = 0; i < 8; ++i) { s = s * 1664525u + 1013904223u; a[i] = RANDOM ? ((int)(s >> 9) % 2001 - 1000) * 1e-3f : 1.0f; }
    for (int i = 0; i < 4; ++i) { s = s * 1664525u + 1013904223u; b[i] = RANDOM ? ((int)(s >> 9) % 2001 - 1000) * 2e-5f : 0.0f; }
    float r = 0;
    if (SHAPE == 16) {
        f32x4 acc[7];
        for (int t = 0; t < 7; ++t) acc[t] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < 7; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[j], acc[t], 0, 0, 0);
        for (int t = 0; t < 7; ++t) r += acc[t][0] + acc[t][3];
    } else {
        f32x16 acc[4];
        for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[j], acc[t], 0, 0, 0);
        for (int t = 0; t < 4; ++t) r += acc[t][0] + acc[t][15];
    }
    out[blockIdx.x * 512 + threadIdx.x] = r;
}
template <int SHAPE, int RANDOM>
void run(const char* name, float* out, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<SHAPE, RANDOM>), dim3(256), dim3(512), 0, 0, out, iters, 7u);
    hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<SHAPE, RANDOM>), dim3(256), dim3(512), 0, 0, out, iters, 7u + i);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = 256.0 * 8 * iters * (SHAPE == 16 ? 28 * 2048.0 : 16 * 4096.0);
    printf("%-40s iters %5d  %8.2f us/launch  %6.1f TFLOP/s\n", name, iters, ms * 1e3 / reps, flop / (ms * 1e-3 / reps) / 1e12);
}
int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    for (int iters : {40, 400}) {
        run<16, 0>("16x16x4 constant operands", out, iters);
        run<16, 1>("16x16x4 random operands", out, iters);
        run<32, 0>("32x32x2 constant operands", out, iters);
        run<32, 1>("32x32x2 random operands", out, iters);
    }
    return 0;
}
