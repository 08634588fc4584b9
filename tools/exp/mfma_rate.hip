// Experiment: what bounds the LSTM-gate GEMM structure?  Variants of the A-resident kernel with
// the W loads and/or the LDS fragment reads stubbed out.  hipcc --offload-arch=gfx950 -O3 mfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int MT = 7, CH = 19, LD = CH * 16 + 4, LDS_ = 320;
__device__ __forceinline__ float comp(const float4& v, int c) { return c == 0 ? v.x : c == 1 ? v.y : c == 2 ? v.z : v.w; }

template <int VAR, int REP>
__global__ __launch_bounds__(512) void k(const float* A, const float* W, float* out, int K, int N) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kk = lane >> 4;
    const int n = blockIdx.x * 128 + wave * 16 + li;
    const int k0 = blockIdx.y * CH * 16;
    for (int i = tid; i < MT * 16 * LDS_; i += 512) smem[i] = 0.001f * (i % 97);
    __syncthreads();
    f32x4 acc[MT];
    for (int t = 0; t < MT; ++t) acc[t] = f32x4{0, 0, 0, 0};
    const float* wp = W + (size_t)n * K + k0 + 4 * kk;
    float4 ring[4];
    for (int i = 0; i < 4; ++i) ring[i] = (VAR & 1) ? *reinterpret_cast<const float4*>(wp + 16 * i) : make_float4(1, 2, 3, 4);
    auto frag = [&](int t, int c) -> float4 {
        if (!(VAR & 2)) return make_float4(0.5f + t, 1, 2, 3);
        if (VAR & 4) {
            const int f = 4 * c + kk, row = t * 16 + li;
            const int pos = (f & ~15) | ((f + 2 * row) & 15);
            return *reinterpret_cast<const float4*>(smem + row * LDS_ + 4 * pos);
        }
        return *reinterpret_cast<const float4*>(smem + (t * 16 + li) * LD + c * 16 + 4 * kk);
    };
    float4 nxt[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) nxt[t] = frag(t, 0);
    for (int rep = 0; rep < REP; ++rep)
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const float4 b = ring[c & 3];
        if (VAR & 1) ring[c & 3] = *reinterpret_cast<const float4*>(wp + 16 * ((c + 4) % CH));
        float4 av[MT];
        if (VAR & 8) {
#pragma unroll
            for (int t = 0; t < MT; ++t) av[t] = nxt[t];
#pragma unroll
            for (int t = 0; t < MT; ++t) nxt[t] = frag(t, (c + 1) % CH);
        } else {
#pragma unroll
            for (int t = 0; t < MT; ++t) av[t] = frag(t, c);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < MT; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(comp(av[t], j), comp(b, j), acc[t], 0, 0, 0);
    }
    float s = 0;
    for (int t = 0; t < MT; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 512 + tid] = s;
}

template <int VAR, int REP>
void run(const char* name, const float* A, const float* W, float* out, int K, int N) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<VAR, REP>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    dim3 g(N / 128, K / (CH * 16));
    size_t lds = MT * 16 * LDS_ * 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<VAR, REP>), g, dim3(512), lds, 0, A, W, out, K, N);
    hipEventRecord(e0);
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL((k<VAR, REP>), g, dim3(512), lds, 0, A, W, out, K, N);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double mfma = (double)g.x * g.y * 8 * REP * CH * 28;
    double us = ms * 1e3 / 50;
    printf("%-34s %8.2f us  -> %6.1f TFLOP/s (mfma-only ideal %.2f us @2.4GHz)\n", name, us, mfma * 2048 / us / 1e6,
           mfma * 32 / 1024 / 2400.0);
}

int main() {
    const int K = 4864, N = 2048;
    float *A, *W, *out;
    hipMalloc(&A, 128 * K * 4); hipMalloc(&W, (size_t)N * K * 4); hipMalloc(&out, 16 * 16 * 512 * 4 * 4);
    hipMemset(A, 0, 128 * K * 4); hipMemset(W, 0, (size_t)N * K * 4);
    run<0, 1>("pure mfma (no W, no LDS) x1", A, W, out, K, N);
    run<0, 8>("pure mfma (no W, no LDS) x8", A, W, out, K, N);
    run<2, 1>("LDS frag reads only x1", A, W, out, K, N);
    run<2, 8>("LDS frag reads only x8", A, W, out, K, N);
    run<1, 1>("W global loads only x1", A, W, out, K, N);
    run<3, 1>("W loads + LDS reads x1", A, W, out, K, N);
    run<3, 8>("W loads + LDS reads x8 (W cached)", A, W, out, K, N);
    run<6, 8>("LDS swizzled x8", A, W, out, K, N);
    run<10, 8>("LDS pipelined x8", A, W, out, K, N);
    run<14, 8>("LDS swizzled+pipelined x8", A, W, out, K, N);
    run<15, 8>("all: W + LDS swz + pipelined x8", A, W, out, K, N);
    run<15, 1>("all: W + LDS swz + pipelined x1", A, W, out, K, N);
    return 0;
}
