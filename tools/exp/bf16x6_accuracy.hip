// Experiment (round 4): fp32 products on the bf16 matrix cores by error-free splitting.
//   a = a1 + a2 + a3 (bf16 pieces, round-to-nearest: exact), b likewise;
//   a b ~= a1 b1 + (a1 b2 + a2 b1) + (a2 b2 + a1 b3 + a3 b1)      [6 bf16 MFMAs, fp32 accumulate]
// The dropped terms are < 2^-25 |a b|.  Question 1: is the accumulate inside v_mfma_f32_32x32x16_bf16 good enough
// (round-to-nearest fp32, not truncation) for the sum to stay in the fp32-roundoff class at K = 4864?
// Compares, against a float64 host reference on the gate-product shape [100 x 4864] x [2048 x 4864]^T:
//   (a) v_mfma_f32_16x16x4_f32 (what gemm_nt_tiled_kernel runs), (b) bf16 x 6, (c) bf16 x 3 (a1b1 + a1b2 + a2b1).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/bf16x6 tools/exp/bf16x6_accuracy.hip && /tmp/bf16x6
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short bf16_rn(float x) {      // round to nearest even
    unsigned u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ void split3(float a, unsigned short* p) {
    p[0] = bf16_rn(a);
    const float r1 = a - bf16_f(p[0]);
    p[1] = bf16_rn(r1);
    const float r2 = r1 - bf16_f(p[1]);
    p[2] = bf16_rn(r2);
}

// one wave per 32 x 32 output tile, whole K; A [M,K], W [N,K] row-major
template <int NPROD>
__global__ __launch_bounds__(64) void split_kernel(const float* A, const float* W, float* C, int M, int N, int K) {
    const int lane = threadIdx.x, i = lane & 31, kg = lane >> 5;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const int arow = min(m0 + i, M - 1), wrow = n0 + i;
    f32x16 hi = {0}, lo = {0};
    for (int k0 = 0; k0 < K; k0 += 16) {
        bf16x8 a[3], b[3];
        for (int j = 0; j < 8; ++j) {
            unsigned short pa[3], pb[3];
            split3(A[(size_t)arow * K + k0 + 8 * kg + j], pa);
            split3(W[(size_t)wrow * K + k0 + 8 * kg + j], pb);
            for (int p = 0; p < 3; ++p) { a[p][j] = (short)pa[p]; b[p][j] = (short)pb[p]; }
        }
        if (NPROD == 7) {            // six products, ONE accumulator (small terms first)
            hi = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], hi, 0, 0, 0);
            hi = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], hi, 0, 0, 0);
            hi = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], hi, 0, 0, 0);
            hi = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], hi, 0, 0, 0);
            hi = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], hi, 0, 0, 0);
            hi = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], hi, 0, 0, 0);
            continue;
        }
        hi = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], hi, 0, 0, 0);
        lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], lo, 0, 0, 0);
        lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], lo, 0, 0, 0);
        if (NPROD >= 6) {
            lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], lo, 0, 0, 0);
            lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], lo, 0, 0, 0);
            lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], lo, 0, 0, 0);
        }
    }
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * kg, col = n0 + i;
        if (row < M) C[(size_t)row * N + col] = hi[r] + lo[r];
    }
}

__global__ __launch_bounds__(64) void f32_kernel(const float* A, const float* W, float* C, int M, int N, int K) {
    const int lane = threadIdx.x, li = lane & 15, kk = lane >> 4;
    const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 16;
    const int arow = min(m0 + li, M - 1), wrow = n0 + li;
    f32x4 acc = {0, 0, 0, 0};
    for (int k0 = 0; k0 < K; k0 += 4)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(size_t)arow * K + k0 + kk], W[(size_t)wrow * K + k0 + kk], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) {
        const int row = m0 + 4 * kk + r;
        if (row < M) C[(size_t)row * N + n0 + li] = acc[r];
    }
}

int main() {
    const int M = 100, N = 2048, K = 4864;
    std::mt19937 g(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> A((size_t)M * K), W((size_t)N * K);
    // A: [u | feature | h]: post-ReLU features (non-negative, mean 0.4) with dropout x2, h in (-1, 1)
    for (int m = 0; m < M; ++m)
        for (int k = 0; k < K; ++k) {
            float v = nd(g);
            if (k < 4352) v = (g() & 1) ? 2.f * fmaxf(0.f, 0.5f * v + 0.4f) : 0.f;
            else v = tanhf(v);
            A[(size_t)m * K + k] = v;
        }
    for (auto& w : W) w = nd(g) * 0.03f;
    std::vector<double> ref((size_t)M * N), mag((size_t)M * N);
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            double s = 0, a = 0;
            for (int k = 0; k < K; ++k) {
                const double p = (double)A[(size_t)m * K + k] * (double)W[(size_t)n * K + k];
                s += p;
                a += fabs(p);
            }
            ref[(size_t)m * N + n] = s;
            mag[(size_t)m * N + n] = a;
        }
    float *dA, *dW, *dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dW, W.size() * 4); hipMalloc(&dC, (size_t)M * N * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> C((size_t)M * N);
    auto report = [&](const char* name) {
        hipDeviceSynchronize();
        hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
        double worst = 0, worst_rel = 0, sum = 0, bias = 0;
        for (size_t i = 0; i < C.size(); ++i) {
            const double e = (double)C[i] - ref[i];
            worst = fmax(worst, fabs(e));
            worst_rel = fmax(worst_rel, fabs(e) / mag[i]);
            sum += e * e;
            bias += e / mag[i];
        }
        printf("%-28s max|err| %.3e   max err/sum|ab| %.3e   rms %.3e   mean signed err/sum|ab| %+.3e\n", name, worst,
               worst_rel, sqrt(sum / C.size()), bias / C.size());
    };
    double vmax = 0;
    for (double v : ref) vmax = fmax(vmax, fabs(v));
    printf("shape [%d x %d] x [%d x %d]^T, max|C| %.3f\n", M, K, N, K, vmax);
    f32_kernel<<<dim3(N / 16, (M + 15) / 16), 64>>>(dA, dW, dC, M, N, K);
    report("mfma_f32_16x16x4_f32");
    split_kernel<6><<<dim3(N / 32, (M + 31) / 32), 64>>>(dA, dW, dC, M, N, K);
    report("bf16 x 6 (32x32x16)");
    split_kernel<3><<<dim3(N / 32, (M + 31) / 32), 64>>>(dA, dW, dC, M, N, K);
    report("bf16 x 3 (32x32x16)");
    split_kernel<7><<<dim3(N / 32, (M + 31) / 32), 64>>>(dA, dW, dC, M, N, K);
    report("bf16 x 6, one accumulator");
    // float rounding of the exact result, for scale
    for (size_t i = 0; i < C.size(); ++i) C[i] = (float)ref[i];
    hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
    report("(float)(exact)");
    return 0;
}
