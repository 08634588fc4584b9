// fp32 GEMMs on the gfx950 matrix cores (v_mfma_f32_16x16x4_f32: exact f32 FMA chain).
//
// Every Linear / LSTMCell product on the speaker/follower path has a skinny batch dimension
// (M = 100 samples, or M = batch*time for the hoisted encoder input product), so the kernels
// stream the weight operand once, straight from HBM into registers (no LDS round trip: a
// weight tile is used by exactly one wave), keep a column of 16x16 accumulator tiles per wave
// and get their parallelism from N-tiles x K-splits.  The small activation operand is re-read
// by every wave and is served by L1/L2.
//
// Fragment trick: mfma_f32_16x16x4 wants A[i][k] from lane (i = lane&15, k = lane>>4) and
// B[k][j] from lane (j = lane&15, k = lane>>4).  A sum over k is order-independent, so a lane
// loads a float4 of four CONSECUTIVE k (k0 + 4*(lane>>4) + c, c = 0..3) for its row and feeds
// component c to MFMA c: A and B use the same permutation of k, loads are 16 B wide and each
// 16-lane group reads 64 contiguous bytes per row.  For operands whose contiguous dimension is
// the row/column index instead (W[K,N] in dX = dY*W, both operands of dW = dY^T*X) the float4
// runs along that index and defines four "virtual" 16-wide tiles with stride-4 columns, which
// the epilogue writes back as float4.
#include "sf_gemm.h"

namespace sf {

namespace {

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float comp(const float4& v, int c) {
    return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w));
}

// ------------------------------------------------------------------------------------------------
// NT: C[M,N] = sum_s A_s[M,K_s] * W_s[N,K_s]^T        (forward Linear; both operands K-contiguous)
// grid (ceil(N/64), ksplit, mblocks), block 256 = 4 waves, wave = MT m-tiles x one 16-col n-tile.
// ------------------------------------------------------------------------------------------------
struct NtArgs {
    Seg seg[3];
    int nseg;
    int M, N;
    int chunks_total;      // sum over segments of ceil(K_s / 16)
    int ksplit;
    float* out;            // slab base ([ksplit][M][N], ld = N) or final y when ksplit == 1
    int ldo;               // N for slabs, ldy for direct
    const float* bias;     // direct mode only
    const float* bias2;
};

template <int MT>
__global__ __launch_bounds__(256) void gemm_nt_kernel(NtArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int n0 = (blockIdx.x * 4 + wave) * 16;
    if (n0 >= a.N) return;
    const int split = blockIdx.y;
    const int m0 = blockIdx.z * (16 * MT);
    const int li = lane & 15, kk = lane >> 4;

    const int n = min(n0 + li, a.N - 1);
    int mrow[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) mrow[t] = min(m0 + 16 * t + li, a.M - 1);

    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int c0 = (int)(((long)split * a.chunks_total) / a.ksplit);
    const int c1 = (int)(((long)(split + 1) * a.chunks_total) / a.ksplit);

    int cs = 0;
    for (int s = 0; s < a.nseg; ++s) {
        const Seg sg = a.seg[s];
        const int nch = (sg.K + 15) >> 4;
        const int lo = max(c0, cs) - cs, hi = min(c1, cs + nch) - cs;
        cs += nch;
        if (lo >= hi) continue;
        const int full = sg.K >> 4;
        const float* pb = sg.W + (size_t)n * sg.ldw + lo * 16 + 4 * kk;
        const float* pa[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t) pa[t] = sg.A + (size_t)mrow[t] * sg.lda + lo * 16 + 4 * kk;
        const int nfull = min(hi, full) - lo;
#pragma unroll 2
        for (int i = 0; i < nfull; ++i) {
            const float4 b = ld4(pb);
            float4 av[MT];
#pragma unroll
            for (int t = 0; t < MT; ++t) av[t] = ld4(pa[t]);
            pb += 16;
#pragma unroll
            for (int t = 0; t < MT; ++t) pa[t] += 16;
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int t = 0; t < MT; ++t) acc[t] = mfma16(comp(av[t], c), comp(b, c), acc[t]);
        }
        if (hi > full) {   // K_s % 16 != 0: one partial chunk, float4 granularity (K_s % 4 == 0)
            const bool ok = full * 16 + 4 * kk < sg.K;
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 b = ok ? ld4(pb) : z;
            float4 av[MT];
#pragma unroll
            for (int t = 0; t < MT; ++t) av[t] = ok ? ld4(pa[t]) : z;
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int t = 0; t < MT; ++t) acc[t] = mfma16(comp(av[t], c), comp(b, c), acc[t]);
        }
    }

    // D layout: col = lane & 15, row = (lane >> 4) * 4 + r
    const int col = n0 + li;
    if (col >= a.N) return;
    float* out = a.out + (a.ksplit > 1 ? (size_t)split * a.M * a.N : 0);
    float bsum = 0.f;
    if (a.ksplit == 1) {
        if (a.bias) bsum += a.bias[col];
        if (a.bias2) bsum += a.bias2[col];
    }
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + 16 * t + kk * 4 + r;
            if (row < a.M) out[(size_t)row * a.ldo + col] = acc[t][r] + bsum;
        }
}

// ------------------------------------------------------------------------------------------------
// NN: C[M,N] = A[M,K] * W[K,N]     (dX = dY * W; A K-contiguous, W N-contiguous)
// grid (ceil(N/256), ksplit, mblocks); wave = MT m-tiles x 64 columns (4 virtual tiles).
// ------------------------------------------------------------------------------------------------
struct NnArgs {
    const float* A;
    int lda;
    const float* W;
    int ldw;
    int M, N, K;
    int ksplit;
    float* out;      // slabs [ksplit][M][N] or final
    int ldo;
    int accumulate;  // direct mode only
};

template <int MT>
__global__ __launch_bounds__(256) void gemm_nn_kernel(NnArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int n0 = (blockIdx.x * 4 + wave) * 64;
    if (n0 >= a.N) return;
    const int split = blockIdx.y;
    const int m0 = blockIdx.z * (16 * MT);
    const int li = lane & 15, kk = lane >> 4;
    const int ncol = n0 + 4 * li;                 // this lane's 4 consecutive columns
    const bool colok = ncol < a.N;                // N % 4 == 0
    const int ncl = colok ? ncol : 0;

    int mrow[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) mrow[t] = min(m0 + 16 * t + li, a.M - 1);

    f32x4 acc[MT][4];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[t][v] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int chunks = (a.K + 15) >> 4;
    const int c0 = (int)(((long)split * chunks) / a.ksplit);
    const int c1 = (int)(((long)(split + 1) * chunks) / a.ksplit);
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);

    for (int ch = c0; ch < c1; ++ch) {
        const int k = ch * 16 + 4 * kk;           // this lane's 4 consecutive k
        float4 bw[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
            bw[c] = (k + c < a.K) ? ld4(a.W + (size_t)(k + c) * a.ldw + ncl) : z;
        float4 av[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t) av[t] = (k < a.K) ? ld4(a.A + (size_t)mrow[t] * a.lda + k) : z;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int v = 0; v < 4; ++v)
#pragma unroll
                for (int t = 0; t < MT; ++t)
                    acc[t][v] = mfma16(comp(av[t], c), comp(bw[c], v), acc[t][v]);
    }

    if (!colok) return;
    float* out = a.out + (a.ksplit > 1 ? (size_t)split * a.M * a.N : 0);
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + 16 * t + kk * 4 + r;
            if (row >= a.M) continue;
            float4 v = make_float4(acc[t][0][r], acc[t][1][r], acc[t][2][r], acc[t][3][r]);
            float4* p = reinterpret_cast<float4*>(out + (size_t)row * a.ldo + ncol);
            if (a.ksplit == 1 && a.accumulate) {
                const float4 o = *p;
                v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
            }
            *p = v;
        }
}

// ------------------------------------------------------------------------------------------------
// TN: out[P,Q] (+)= Y[M,P]^T * X[M,Q]   (weight gradients; reduction over the batch rows)
// grid (ceil(Q/256), ceil(P/64)); wave = 64 P-rows x 64 Q-cols (4 x 4 virtual tiles).
// ------------------------------------------------------------------------------------------------
struct TnArgs {
    const float* Y;
    int ldy;
    const float* X;
    int ldx;
    int M, P, Q;
    float* out;
    int ldo;
    int accumulate;
};

__global__ __launch_bounds__(256) void gemm_tn_kernel(TnArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int q0 = (blockIdx.x * 4 + wave) * 64;
    if (q0 >= a.Q) return;
    const int p0 = blockIdx.y * 64;
    const int li = lane & 15, kk = lane >> 4;
    const int pc = p0 + 4 * li;                 // Y columns pc..pc+3 (must be readable: ldy padded)
    const int qc = q0 + 4 * li;
    const bool pok = pc < a.ldy;                // stay inside the row (caller pads P to %4 via ldy)
    const bool qok = qc < a.Q;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll 2
    for (int m = kk; m < a.M + kk; m += 4) {    // every lane runs the same trip count
        const bool ok = m < a.M;
        const float4 yv = (ok && pok) ? ld4(a.Y + (size_t)m * a.ldy + pc) : z;
        const float4 xv = (ok && qok) ? ld4(a.X + (size_t)m * a.ldx + qc) : z;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(comp(yv, i), comp(xv, j), acc[i][j]);
    }

    if (!qok) return;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int p = p0 + 4 * (kk * 4 + r) + i;     // tile row (kk*4+r) is virtual: stride-4 rows
            if (p >= a.P) continue;
            float4 v = make_float4(acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]);
            float4* o = reinterpret_cast<float4*>(a.out + (size_t)p * a.ldo + qc);
            if (a.accumulate) {
                const float4 w = *o;
                v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
            }
            *o = v;
        }
}

// ------------------------------------------------------------------------------------------------
// split-K reduce + epilogue
// ------------------------------------------------------------------------------------------------
struct RedArgs {
    const float* slabs;
    int ks;
    int M, N;
    float* y;
    int ldy;
    const float* bias;
    const float* bias2;
    const float* mul;
    float* y_pre;
    int ldy_pre;
    int epi;
    int accumulate;
};

__global__ __launch_bounds__(256) void reduce_slabs_kernel(RedArgs a) {
    const size_t total = (size_t)a.M * a.N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / a.N), col = (int)(i % a.N);
        float v = 0.f;
        for (int s = 0; s < a.ks; ++s) v += a.slabs[(size_t)s * total + i];
        if (a.bias) v += a.bias[col];
        if (a.bias2) v += a.bias2[col];
        if (a.epi == EPI_TANH) v = tanhf(v);
        if (a.epi == EPI_MUL) {
            if (a.y_pre) a.y_pre[(size_t)row * a.ldy_pre + col] = v;
            v *= a.mul[col];
        }
        float* o = a.y + (size_t)row * a.ldy + col;
        *o = a.accumulate ? *o + v : v;
    }
}

__global__ __launch_bounds__(256) void epilogue_inplace_kernel(RedArgs a) {
    // single-split direct output already holds acc + bias: apply the non-linear part in place
    const size_t total = (size_t)a.M * a.N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / a.N), col = (int)(i % a.N);
        float* o = a.y + (size_t)row * a.ldy + col;
        float v = *o;
        if (a.epi == EPI_TANH) v = tanhf(v);
        if (a.epi == EPI_MUL) {
            if (a.y_pre) a.y_pre[(size_t)row * a.ldy_pre + col] = v;
            v *= a.mul[col];
        }
        *o = v;
    }
}

__global__ __launch_bounds__(256) void colsum_kernel(const float* Y, int ldy, int M, int N,
                                                     float* out, int accumulate) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float s = 0.f;
    for (int m = 0; m < M; ++m) s += Y[(size_t)m * ldy + n];
    out[n] = accumulate ? out[n] + s : s;
}

int pick_ksplit(int waves_per_split, int chunks) {
    // aim for ~2 waves per SIMD over the chip (1024 SIMDs), at least 4 16-deep chunks per split,
    // and at most 8 partial slabs (each split costs one extra write + read of the output)
    int ks = (2048 + waves_per_split - 1) / waves_per_split;
    ks = std::min(ks, std::max(1, chunks / 4));
    return std::max(1, std::min(ks, 8));
}

template <int MT>
void launch_nt(const NtArgs& a, dim3 grid, hipStream_t st) {
    hipLaunchKernelGGL(gemm_nt_kernel<MT>, grid, dim3(256), 0, st, a);
}

template <int MT>
void launch_nn(const NnArgs& a, dim3 grid, hipStream_t st) {
    hipLaunchKernelGGL(gemm_nn_kernel<MT>, grid, dim3(256), 0, st, a);
}

inline int red_grid(size_t total) { return (int)std::min<size_t>((total + 255) / 256, 2048); }

}  // namespace

static void nt_shape(int M, int N, int Ktot_chunks, int* mt, int* mblocks, int* ks) {
    const int mtiles = ceil_div(M, 16);
    *mt = mtiles <= 8 ? mtiles : 8;
    *mblocks = ceil_div(mtiles, *mt);
    const int waves = ceil_div(N, 16) * *mblocks;
    *ks = pick_ksplit(waves, Ktot_chunks);
}

int linear_ksplit(int M, int N, int Ktot) {
    int mt, mb, ks;
    nt_shape(M, N, ceil_div(Ktot, 16), &mt, &mb, &ks);
    return ks;
}

size_t linear_ws_floats(int M, int N, int Ktot) {
    const int ks = linear_ksplit(M, N, Ktot);
    return ks > 1 ? (size_t)ks * M * N : 0;
}

int linear_nt(const Seg* segs, int nseg, int M, int N, const LinearOut& out, float* ws,
              size_t ws_floats, hipStream_t st, float** raw_slabs, int* ksplit_out) {
    SF_CHECK_ARG(nseg >= 1 && nseg <= 3 && M > 0 && N > 0);
    NtArgs a{};
    a.nseg = nseg;
    int chunks = 0;
    for (int s = 0; s < nseg; ++s) {
        SF_CHECK_ARG(segs[s].K > 0 && segs[s].K % 4 == 0 && segs[s].lda % 4 == 0 &&
                     segs[s].ldw % 4 == 0);
        a.seg[s] = segs[s];
        chunks += ceil_div(segs[s].K, 16);
    }
    int mt, mblocks, ks;
    nt_shape(M, N, chunks, &mt, &mblocks, &ks);
    const bool slabs = ks > 1 || raw_slabs;
    if (slabs) {
        if (!ws || ws_floats < (size_t)ks * M * N) return SF_ERR_WORKSPACE;
    }
    a.M = M;
    a.N = N;
    a.chunks_total = chunks;
    a.ksplit = ks;
    if (slabs) {
        a.out = ws;
        a.ldo = N;
        a.bias = nullptr;
        a.bias2 = nullptr;
    } else {
        a.out = out.y;
        a.ldo = out.ldy;
        a.bias = out.bias;
        a.bias2 = out.bias2;
    }
    const NtArgs& k = a;   // raw slabs with ks == 1: slab 0 is written without bias
    dim3 grid(ceil_div(N, 64), ks, mblocks);
    switch (mt) {
        case 1: launch_nt<1>(k, grid, st); break;
        case 2: launch_nt<2>(k, grid, st); break;
        case 3: launch_nt<3>(k, grid, st); break;
        case 4: launch_nt<4>(k, grid, st); break;
        case 5: launch_nt<5>(k, grid, st); break;
        case 6: launch_nt<6>(k, grid, st); break;
        case 7: launch_nt<7>(k, grid, st); break;
        default: launch_nt<8>(k, grid, st); break;
    }
    if (ksplit_out) *ksplit_out = ks;
    if (raw_slabs) {
        *raw_slabs = ws;
        return launch_status();
    }
    RedArgs r{};
    r.M = M;
    r.N = N;
    r.y = out.y;
    r.ldy = out.ldy;
    r.mul = out.mul;
    r.y_pre = out.y_pre;
    r.ldy_pre = out.ldy_pre;
    r.epi = out.epi;
    if (ks > 1) {
        r.slabs = ws;
        r.ks = ks;
        r.bias = out.bias;
        r.bias2 = out.bias2;
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3(red_grid((size_t)M * N)), dim3(256), 0, st, r);
    } else if (out.epi != EPI_NONE) {
        hipLaunchKernelGGL(epilogue_inplace_kernel, dim3(red_grid((size_t)M * N)), dim3(256), 0,
                           st, r);
    }
    return launch_status();
}

// NN split-K workspace is provided by a per-stream scratch owned by the API layer.
int gemm_nn_ws(const float* A, int lda, const float* W, int ldw, int M, int N, int K, float* y,
               int ldy, int accumulate, float* ws, size_t ws_floats, hipStream_t st) {
    SF_CHECK_ARG(M > 0 && N > 0 && K > 0 && N % 4 == 0 && lda % 4 == 0 && ldw % 4 == 0 &&
                 ldy % 4 == 0);
    const int mtiles = ceil_div(M, 16);
    int mt = mtiles <= 1 ? 1 : (mtiles <= 2 ? 2 : (mtiles <= 4 ? 4 : 7));
    if (mtiles > 7) mt = 7;
    const int mblocks = ceil_div(mtiles, mt);
    const int chunks = ceil_div(K, 16);
    int ks = pick_ksplit(ceil_div(N, 64) * mblocks, chunks);
    if (ks > 1 && (!ws || ws_floats < (size_t)ks * M * N)) ks = 1;   // degrade, still correct
    NnArgs a{A, lda, W, ldw, M, N, K, ks, ks > 1 ? ws : y, ks > 1 ? N : ldy, accumulate};
    dim3 grid(ceil_div(N, 256), ks, mblocks);
    switch (mt) {
        case 1: launch_nn<1>(a, grid, st); break;
        case 2: launch_nn<2>(a, grid, st); break;
        case 4: launch_nn<4>(a, grid, st); break;
        default: launch_nn<7>(a, grid, st); break;
    }
    if (ks > 1) {
        RedArgs r{};
        r.slabs = ws;
        r.ks = ks;
        r.M = M;
        r.N = N;
        r.y = y;
        r.ldy = ldy;
        r.epi = EPI_NONE;
        r.accumulate = accumulate;
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3(red_grid((size_t)M * N)), dim3(256), 0, st, r);
    }
    return launch_status();
}

size_t gemm_nn_ws_floats(int M, int N, int K) {
    const int mtiles = ceil_div(M, 16);
    int mt = mtiles <= 1 ? 1 : (mtiles <= 2 ? 2 : (mtiles <= 4 ? 4 : 7));
    const int mblocks = ceil_div(mtiles, mt);
    const int ks = pick_ksplit(ceil_div(N, 64) * mblocks, ceil_div(K, 16));
    return ks > 1 ? (size_t)ks * M * N : 0;
}

int gemm_nn(const float* A, int lda, const float* W, int ldw, int M, int N, int K, float* y,
            int ldy, int accumulate, hipStream_t st) {
    return gemm_nn_ws(A, lda, W, ldw, M, N, K, y, ldy, accumulate, nullptr, 0, st);
}

int gemm_tn(const float* Y, int ldy, const float* X, int ldx, int M, int P, int Q, float* out,
            int ldo, int accumulate, hipStream_t st) {
    SF_CHECK_ARG(M > 0 && P > 0 && Q > 0 && Q % 4 == 0 && ldy % 4 == 0 && ldx % 4 == 0 &&
                 ldo % 4 == 0);
    TnArgs a{Y, ldy, X, ldx, M, P, Q, out, ldo, accumulate};
    dim3 grid(ceil_div(Q, 256), ceil_div(P, 64));
    hipLaunchKernelGGL(gemm_tn_kernel, grid, dim3(256), 0, st, a);
    return launch_status();
}

int colsum(const float* Y, int ldy, int M, int N, float* out, int accumulate, hipStream_t st) {
    SF_CHECK_ARG(M > 0 && N > 0);
    hipLaunchKernelGGL(colsum_kernel, dim3(ceil_div(N, 256)), dim3(256), 0, st, Y, ldy, M, N, out,
                       accumulate);
    return launch_status();
}

}  // namespace sf
