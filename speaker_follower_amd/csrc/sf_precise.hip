// Small Linears on the FLOAT64 matrix cores (v_mfma_f64_16x16x4_f64), round 5.
//
// Why: the speaker's path encoder (model.py:429-457) forms its visual-attention scores as
// x_v . (W_v^T (W_h h + b_h)) over K = 512 -> 256 -> 2176.  With the reference's own "peaky" weights those scores
// reach +-80, where one fp32 ulp is 7.6e-6: every fp32 evaluation of the chain -- the reference's CPU path as much
// as the fp32 kernels here -- carries 2e-6 .. 9e-6 of roundoff PER STAGE into the softmax weights, the 7-step context
// amplifies it and it arrives as 1e-4 .. 3e-4 in the word logits (tools/speaker_drift.py: the word loop itself adds
// 1e-5).  Rounding the three intermediates once instead (t_v and q kept in float64, the score accumulated in float64)
// leaves 3e-7 in the softmax weights.  The products are tiny (26 + 111 MFLOP per path step); the f64 matrix pipe
// (78 TFLOP/s) does them in the time of the launch ramp.
//
//   y64[M,N] = A[M,K] W[N,K]^T + bias      A fp32 or f64, W / bias fp32, accumulate f64;  optional fp32 copy of y
//
// One workgroup (4 waves) per 16 x 16 output tile; the waves split K in interleaved 16-deep chunks (lane (r, kq)
// pulls 4 consecutive k of row r: 16-byte loads, the 4 values feed 4 consecutive MFMAs as k-slot kq), partial tiles
// meet in LDS, the bias is added once.
#include "sf_kernels.h"

namespace sf {

int g_precise_attention = 1;      // sf_debug_precise_attention

namespace {

typedef double f64x4 __attribute__((ext_vector_type(4)));

template <typename TA>
__device__ __forceinline__ void load4(const TA* p, double (&v)[4]);
template <>
__device__ __forceinline__ void load4<float>(const float* p, double (&v)[4]) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = (double)t.x; v[1] = (double)t.y; v[2] = (double)t.z; v[3] = (double)t.w;
}
template <>
__device__ __forceinline__ void load4<double>(const double* p, double (&v)[4]) {
    const double2 t0 = *reinterpret_cast<const double2*>(p), t1 = *reinterpret_cast<const double2*>(p + 2);
    v[0] = t0.x; v[1] = t0.y; v[2] = t1.x; v[3] = t1.y;
}

template <typename TA, typename TW = float>
__global__ __launch_bounds__(256) void linear_f64_kernel(const TA* __restrict__ A, int lda, const TW* __restrict__ W,
                                                         int ldw, const TW* __restrict__ bias, int M, int N, int K,
                                                         double* __restrict__ y64, int ldy64, float* __restrict__ y32,
                                                         int ldy32) {
    __shared__ double red[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, kq = lane >> 4;
    const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 16;
    const TA* arow = A + (size_t)min(m0 + r, M - 1) * lda;
    const TW* wrow = W + (size_t)min(n0 + r, N - 1) * ldw;
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
    for (int kc = wave * 16; kc < K; kc += 64) {
        const int k = kc + 4 * kq;
        const bool ok = k < K;                       // K % 4 == 0 (checked by the launcher): a partial last chunk
        double a[4], w[4];
        load4<TA>(arow + (ok ? k : 0), a);
        load4<TW>(wrow + (ok ? k : 0), w);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ok ? a[j] : 0.0, ok ? w[j] : 0.0, acc, 0, 0, 0);
    }
    // lane (r, kq) holds D[4 v + kq][r] in register pair v (measured: tools/exp/mfma_f64_layout.hip -- NOT the fp32
    // 16x16x4 layout, whose lane group kq holds rows 4 kq + v)
#pragma unroll
    for (int v = 0; v < 4; ++v) red[wave][(4 * v + kq) * 16 + r] = acc[v];
    __syncthreads();
    const int t = threadIdx.x, row = m0 + (t >> 4), col = n0 + (t & 15);
    if (row < M && col < N) {
        double s = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
        if (bias) s += (double)bias[col];
        if (y64) y64[(size_t)row * ldy64 + col] = s;
        if (y32) y32[(size_t)row * ldy32 + col] = (float)s;
    }
}

}  // namespace

int linear_f64(const float* A32, const double* A64, int lda, const float* W, int ldw, const float* bias, int M, int N,
               int K, double* y64, int ldy64, float* y32, int ldy32, hipStream_t st) {
    if ((K & 3) || (lda & 3) || (ldw & 3) || M <= 0 || N <= 0 || (!A32 == !A64)) return SF_ERR_UNSUPPORTED;
    const dim3 grid(ceil_div(N, 16), ceil_div(M, 16));
    if (A32)
        SF_LAUNCH_AS("linear_f64_kernel<f32 in>", linear_f64_kernel<float>, grid, dim3(256), 0, st, A32, lda, W, ldw, bias, M,
                     N, K, y64, ldy64, y32, ldy32);
    else
        SF_LAUNCH_AS("linear_f64_kernel<f64 in>", linear_f64_kernel<double>, grid, dim3(256), 0, st, A64, lda, W, ldw, bias,
                     M, N, K, y64, ldy64, y32, ldy32);
    return launch_status();
}

// The same with FLOAT64 weights and bias (A fp32): the query of the visual attention through the folded matrix
// M_v = W_v^T W_h [F,H] and c_v = W_v^T b_h [F], both formed and kept in float64 (sf_visual_query_fold_f64) -- one
// product per path step instead of two dependent ones, and one rounding fewer.
int linear_f64_w64(const float* A32, int lda, const double* W, int ldw, const double* bias, int M, int N, int K,
                   double* y64, int ldy64, float* y32, int ldy32, hipStream_t st) {
    if ((K & 3) || (lda & 3) || (ldw & 3) || M <= 0 || N <= 0 || !A32 || !W) return SF_ERR_UNSUPPORTED;
    const dim3 grid(ceil_div(N, 16), ceil_div(M, 16));
    SF_LAUNCH_AS("linear_f64_kernel<f32 in, f64 w>", (linear_f64_kernel<float, double>), grid, dim3(256), 0, st, A32, lda, W,
                 ldw, bias, M, N, K, y64, ldy64, y32, ldy32);
    return launch_status();
}

}  // namespace sf
