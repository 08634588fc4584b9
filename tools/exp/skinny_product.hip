// Lab (round 5): the folded inference products [100 x 512] x [2180 x 512]^T (q' = M_v h1 + c_v, [r | c] = M_a h~ + c_a) as
// K-split slabs whose consumer adds them up: how fast can one launch be?  (gemm_nt_small_kernel<4, 4> takes 11.5 us: 272
// blocks x 160 KB, A re-read by every 16-column block.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float comp(const float4& v, int c) { return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w)); }

// block = NT n-tiles (16 columns each) x KW K-groups of 64; grid = (ceil(N / (16 NT)), K / (64 KW)); slab s = blockIdx.y
template <int NT, int KW, int MT, int MODE = 0>
__global__ __launch_bounds__(NT * KW * 64) void skinny(const float* A, int lda, const float* W, int ldw, const float* bias,
                                                      float* out, int M, int N) {
    __shared__ float red[KW > 1 ? (KW - 1) * NT * MT * 256 : 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nt = wave % NT, kw = wave / NT;
    const int li = lane & 15, kk = lane >> 4;
    const int n = min((int)blockIdx.x * 16 * NT + nt * 16 + li, N - 1);
    const int k0 = ((int)blockIdx.y * KW + kw) * 64 + 4 * kk;
    float4 b[4], a[MT][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = ld4(W + (size_t)n * ldw + k0 + 16 * j);
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int row = min(16 * t + li, M - 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) a[t][j] = ld4(A + (size_t)row * lda + k0 + 16 * j);
    }
    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                if (MODE == 1) acc[t][c] += comp(a[t][j], c) * comp(b[j], c);
                else acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(comp(a[t][j], c), comp(b[j], c), acc[t], 0, 0, 0);
            }
    if (KW > 1) {
        if (kw > 0) {
#pragma unroll
            for (int t = 0; t < MT; ++t)
                reinterpret_cast<f32x4*>(red)[(((kw - 1) * NT + nt) * MT + t) * 64 + lane] = acc[t];
        }
        __syncthreads();
        if (kw > 0) return;
#pragma unroll
        for (int q = 1; q < KW; ++q)
#pragma unroll
            for (int t = 0; t < MT; ++t) acc[t] += reinterpret_cast<f32x4*>(red)[(((q - 1) * NT + nt) * MT + t) * 64 + lane];
    }
    const int col = blockIdx.x * 16 * NT + nt * 16 + li;
    if (col >= N) return;
    const float bs = (blockIdx.y == 0 && bias) ? bias[col] : 0.f;
    float* slab = out + (size_t)blockIdx.y * M * N;
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * t + kk * 4 + r;
            if (row < M && (MODE != 2 || acc[t][r] == 12345.f)) slab[(size_t)row * N + col] = acc[t][r] + bs;
        }
}

template <int NT, int KW, int MT = 7, int MODE = 0>
void run(const float* A, const float* W, const float* bias, float* out, int M, int N, int K, float* junk, size_t junk_n, int lda = 512) {
    dim3 grid((N + 16 * NT - 1) / (16 * NT), K / (64 * KW));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9, sum = 0;
    for (int i = 0; i < 40; ++i) {
        (void)hipMemsetAsync(junk, 0, junk_n, 0);                       // push the operands out of the L2s (MALL keeps them)
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((skinny<NT, KW, MT, MODE>), grid, dim3(NT * KW * 64), 0, 0, A, lda, W, K, bias, out, M, N);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (i >= 5) { best = ms < best ? ms : best; sum += ms; }
    }
    // back to back (graph-like): 50 launches between one event pair
    (void)hipEventRecord(e0);
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL((skinny<NT, KW, MT, MODE>), grid, dim3(NT * KW * 64), 0, 0, A, lda, W, K, bias, out, M, N);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms50; (void)hipEventElapsedTime(&ms50, e0, e1);
    printf("lda %d m-tiles %d mode %d | n-tiles %d  K-groups %d  grid %3d x %d (%4d blocks of %3d threads, %d slabs): cold-L2 single %.1f us (min %.1f)   back to back %.2f us\n",
           lda, MT, MODE, NT, KW, grid.x, grid.y, grid.x * grid.y, NT * KW * 64, grid.y, sum / 35 * 1e3, best * 1e3, ms50 * 1e3 / 50);
}

int main() {
    const int M = 100, N = 2180, K = 512;
    std::vector<float> hA((size_t)M * K), hW((size_t)N * K), hb(N);
    unsigned s = 1;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((int)(s >> 9) % 2001 - 1000) * 1e-3f; };
    for (auto& v : hA) v = rnd();
    for (auto& v : hW) v = rnd() * 0.05f;
    for (auto& v : hb) v = rnd();
    float *A, *A2, *W, *b, *out, *junk;
    const size_t junk_n = (size_t)64 << 20;
    (void)hipMalloc(&A, hA.size() * 4); (void)hipMalloc(&A2, (size_t)M * 544 * 4); (void)hipMemset(A2, 0, (size_t)M * 544 * 4); (void)hipMalloc(&W, hW.size() * 4); (void)hipMalloc(&b, hb.size() * 4);
    (void)hipMalloc(&out, (size_t)8 * M * N * 4); (void)hipMalloc(&junk, junk_n);
    (void)hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
    run<2, 2>(A, W, b, out, M, N, K, junk, junk_n);
    run<2, 2>(A2, W, b, out, M, N, K, junk, junk_n, 528);
    run<2, 2>(A2, W, b, out, M, N, K, junk, junk_n, 544);
    run<2, 2, 7, 2>(A2, W, b, out, M, N, K, junk, junk_n, 528);
    run<2, 2, 1>(A, W, b, out, M, N, K, junk, junk_n);
    run<2, 2, 3>(A, W, b, out, M, N, K, junk, junk_n);
    run<2, 2, 7, 1>(A, W, b, out, M, N, K, junk, junk_n);
    run<2, 2, 7, 2>(A, W, b, out, M, N, K, junk, junk_n);
    run<4, 1>(A, W, b, out, M, N, K, junk, junk_n);
    run<2, 1>(A, W, b, out, M, N, K, junk, junk_n);
    run<4, 2>(A, W, b, out, M, N, K, junk, junk_n);
    run<1, 2>(A, W, b, out, M, N, K, junk, junk_n);
    run<2, 4>(A, W, b, out, M, N, K, junk, junk_n);
    // correctness of the last variant against the host
    std::vector<float> ho((size_t)2 * M * N);
    (void)hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int r = 0; r < M; r += 7)
        for (int c = 0; c < N; c += 13) {
            double ref = hb[c];
            for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)r * K + k] * hW[(size_t)c * K + k];
            const double got = (double)ho[(size_t)r * N + c] + ho[(size_t)M * N + (size_t)r * N + c];
            worst = fabs(got - ref) > worst ? fabs(got - ref) : worst;
        }
    printf("max |error| of the two-slab variant against float64: %.2e\n", worst);
    return 0;
}
