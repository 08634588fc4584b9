// Streaming (HBM-bound) kernels of the path: LSTM gate math, dropout, gathers from the feature
// table, the follower/speaker per-step glue, and the small vector helpers the backward needs.
#include <atomic>
#include "sf_kernels.h"

#include <algorithm>
#include <cmath>
#include "sf_glue.h"
#include "sf_sampling.h"

namespace sf {

namespace {

constexpr int TPB = 256;
inline int grid1d(size_t n) { return (int)std::min<size_t>((n + TPB - 1) / TPB, 4096); }

// -------------------------------------------------------------------------------------------------
// LSTM pointwise forward: one thread per (row, hidden unit)
// -------------------------------------------------------------------------------------------------
// KS = number of split-K slabs (compile time: every slab load is issued before the first add, in
// straight-line code).  The slabs were just written by the GEMM and sit in L2 / Infinity Cache:
// this kernel is latency-bound, so one round trip instead of one per slab is what matters.
template <int KS>
__global__ __launch_bounds__(TPB) void lstm_pw_fwd_kernel(LstmPwFwd a) {
    const int H = a.H, B = a.B;
    const size_t slab = (size_t)B * 4 * H;
    for (int idx = blockIdx.x * TPB + threadIdx.x; idx < B * H; idx += gridDim.x * TPB) {
        const int b = idx / H, j = idx - b * H;
        float part[4][KS > 0 ? KS : 1];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int s = 0; s < KS; ++s)
                part[g][s] = a.slabs[s * slab + (size_t)b * 4 * H + g * H + j];
        float bi[4], bh[4], xv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bi[g] = a.b_ih[g * H + j];
            bh[g] = a.b_hh[g * H + j];
        }
        const float c0 = a.c0[idx];
        const LstmLive lv = lstm_live_load(a, b, j);
        if (a.xg) {                                              // block-uniform
#pragma unroll
            for (int g = 0; g < 4; ++g) xv[g] = a.xg[(size_t)b * 4 * H + g * H + j];
        }
        float g4[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v = bi[g] + bh[g] + xv[g];
#pragma unroll
            for (int s = 0; s < KS; ++s) v += part[g][s];
            g4[g] = v;
        }
        lstm_cell_update(a, b, j, g4, c0, lv);
    }
}

// LSTM pointwise backward: dgates (pre-activation), dc0
__global__ __launch_bounds__(TPB) void lstm_pw_bwd_kernel(LstmPwBwd a) {
    const int H = a.H, B = a.B;
    for (int idx = blockIdx.x * TPB + threadIdx.x; idx < B * H; idx += gridDim.x * TPB) {
        const int b = idx / H, j = idx - b * H;
        lstm_pw_bwd_elem(a, b, j, a.dh1 ? a.dh1[idx] : 0.f);         // (sf_lstm_pw.h)
    }
}

__global__ __launch_bounds__(TPB) void dropout_copy_kernel(const float* src, int lds, int B, int N,
                                                           float* dst, int ldd, Dropout d,
                                                           int col0) {
    const size_t total = (size_t)B * N;
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < total; i += (size_t)gridDim.x * TPB) {
        const int b = (int)(i / N), n = (int)(i % N);
        float v = src[(size_t)b * lds + n];
        if (d.on()) {
            const uint32_t rk = drop_key(d, (uint32_t)(d.row0 + b));
            v = dropout_keep(rk, (uint32_t)(col0 + n), d.thresh) ? v * d.scale : 0.f;
        }
        dst[(size_t)b * ldd + n] = v;
    }
}

// The same over S stacked steps of B rows each: row m = t * B + b is masked with site `d.stream + stream_step * t`, row
// key b -- what S per-step launches with streams 2 (step0 + t) + 1 would apply (the speaker's dropout(h1), model.py:516,
// for all word steps of a teacher-forced pass at once, and its backward).
__global__ __launch_bounds__(TPB) void dropout_steps_kernel(const float* src, int lds, int S, int B, int N, float* dst,
                                                            int ldd, Dropout d, uint32_t stream_step) {
    const size_t total = (size_t)S * B * N;
    const uint32_t seed = drop_seed(d);
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < total; i += (size_t)gridDim.x * TPB) {
        const int m = (int)(i / N), n = (int)(i % N);
        float v = src[(size_t)m * lds + n];
        if (d.on()) {
            const int t = m / B, b = m - t * B;
            const uint32_t rk = dropout_row_key(seed, d.stream + stream_step * (uint32_t)t, (uint32_t)(d.row0 + b));
            v = dropout_keep(rk, (uint32_t)n, d.thresh) ? v * d.scale : 0.f;
        }
        dst[(size_t)m * ldd + n] = v;
    }
}
__global__ void row_mod_kernel(int* out, int M, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < M) out[i] = i % B;
}

// enc: dctx for the encoder arrives per (b, t, :) and must be dropout-masked with col = t*H + j
__global__ __launch_bounds__(TPB) void ctx_grad_slice_kernel(const float* dctx, int T, int H, int B,
                                                             int t, Dropout d, float* out) {
    for (int idx = blockIdx.x * TPB + threadIdx.x; idx < B * H; idx += gridDim.x * TPB) {
        const int b = idx / H, j = idx - b * H;
        float v = dctx[((size_t)b * T + t) * H + j];
        if (d.on()) {
            const uint32_t rk = drop_key(d, (uint32_t)(d.row0 + b));
            v = dropout_keep(rk, (uint32_t)(t * H + j), d.thresh) ? v * d.scale : 0.f;
        }
        out[idx] = v;
    }
}

enum EwOp { EW_ADD2, EW_TANH_BWD, EW_SCALE_COLS, EW_RANK1_ADD };
struct EwArgs {
    const float* a; int lda;
    const float* b; int ldb;
    int M, N;
    float* dst; int ldd;
};
template <int OP>
__global__ __launch_bounds__(TPB) void ew_kernel(EwArgs e) {
    const size_t total = (size_t)e.M * e.N;
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < total; i += (size_t)gridDim.x * TPB) {
        const int m = (int)(i / e.N), n = (int)(i % e.N);
        float* o = e.dst + (size_t)m * e.ldd + n;
        if (OP == EW_ADD2) {
            float v = 0.f;
            if (e.a) v += e.a[(size_t)m * e.lda + n];
            if (e.b) v += e.b[(size_t)m * e.ldb + n];
            *o = v;
        } else if (OP == EW_TANH_BWD) {
            const float y = e.a[(size_t)m * e.lda + n];
            *o = e.b[(size_t)m * e.ldb + n] * (1.f - y * y);
        } else if (OP == EW_SCALE_COLS) {
            *o = e.a[(size_t)m * e.lda + n] * e.b[n];
        } else {                                   // dst[m,n] += a[m] * b[n]
            *o += e.a[m] * e.b[n];
        }
    }
}

// out[n] += sum_m a[m,n] * b[m,n]  /  out[n] += sum_m s[m] * x[m,n]: block = 64 columns x 16 row
// groups (coalesced 256-B row segments), partials meet in LDS.
__global__ __launch_bounds__(1024) void colsum_prod_kernel(const float* a, int lda, const float* b,
                                                           int ldb, int M, int N, float* out) {
    __shared__ float s_p[16][64];
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + c;
    const int nc = min(n, N - 1);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;          // 8 independent loads in flight per thread
    int m = g;
    for (; m + 48 < M; m += 64) {
        s0 += a[(size_t)m * lda + nc] * b[(size_t)m * ldb + nc];
        s1 += a[(size_t)(m + 16) * lda + nc] * b[(size_t)(m + 16) * ldb + nc];
        s2 += a[(size_t)(m + 32) * lda + nc] * b[(size_t)(m + 32) * ldb + nc];
        s3 += a[(size_t)(m + 48) * lda + nc] * b[(size_t)(m + 48) * ldb + nc];
    }
    for (; m < M; m += 16) s0 += a[(size_t)m * lda + nc] * b[(size_t)m * ldb + nc];
    s_p[g][c] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && n < N) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += s_p[k][c];
        out[n] += t;
    }
}
__global__ __launch_bounds__(1024) void dot_rows_kernel(const float* sv, const float* x, int ldx,
                                                        int M, int N, float* out) {
    __shared__ float s_p[16][64];
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + c;
    const int nc = min(n, N - 1);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int m = g;
    for (; m + 48 < M; m += 64) {
        s0 += sv[m] * x[(size_t)m * ldx + nc];
        s1 += sv[m + 16] * x[(size_t)(m + 16) * ldx + nc];
        s2 += sv[m + 32] * x[(size_t)(m + 32) * ldx + nc];
        s3 += sv[m + 48] * x[(size_t)(m + 48) * ldx + nc];
    }
    for (; m < M; m += 16) s0 += sv[m] * x[(size_t)m * ldx + nc];
    s_p[g][c] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && n < N) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += s_p[k][c];
        out[n] += t;
    }
}
__global__ void sum_accum_kernel(const float* s, int M, float* out) {     // one wave, fixed order
    const int lane = threadIdx.x;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int m = lane;
    for (; m + 192 < M; m += 256) {
        a0 += s[m]; a1 += s[m + 64]; a2 += s[m + 128]; a3 += s[m + 192];
    }
    for (; m < M; m += 64) a0 += s[m];
    const float t = wave_sum((a0 + a1) + (a2 + a3));
    if (lane == 0) out[0] += t;
}
// 32x32 LDS-tiled transpose (both sides coalesced)
__global__ __launch_bounds__(256) void transpose_kernel(const float* src, int lds, int R, int C, float* dst) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8)
        if (r0 + i < R && c0 + tx < C) tile[i][tx] = src[(size_t)(r0 + i) * lds + c0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < C && r0 + tx < R) dst[(size_t)(c0 + i) * R + r0 + tx] = tile[tx][i];
}
__global__ __launch_bounds__(TPB) void fill_kernel(float* p, size_t n, float v) {
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (size_t)gridDim.x * TPB)
        p[i] = v;
}

// out[t, b, :] = table[seq[b, t], :]      (time-major so that step t is one contiguous [B,E] block)
__global__ __launch_bounds__(TPB) void embedding_tm_kernel(const float* table, int E,
                                                           const int64_t* seq, int B, int Lpad,
                                                           int T, float* out) {
    const int e4 = E >> 2;
    const size_t total = (size_t)T * B * e4;
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < total; i += (size_t)gridDim.x * TPB) {
        const int c = (int)(i % e4);
        const size_t tb = i / e4;
        const int b = (int)(tb % B), t = (int)(tb / B);
        const int64_t w = seq[(size_t)b * Lpad + t];
        reinterpret_cast<float4*>(out)[i] = reinterpret_cast<const float4*>(table)[(size_t)w * e4 + c];
    }
}
__global__ __launch_bounds__(TPB) void embedding_rows_kernel(const float* table, int E,
                                                             const int64_t* idx, int B, float* out) {
    const int e4 = E >> 2;
    const size_t total = (size_t)B * e4;
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < total; i += (size_t)gridDim.x * TPB) {
        const int c = (int)(i % e4), b = (int)(i / e4);
        reinterpret_cast<float4*>(out)[i] =
            reinterpret_cast<const float4*>(table)[(size_t)idx[b] * e4 + c];
    }
}

// ---- batched gathers from the HBM feature table (a11) -------------------------------------------
__global__ __launch_bounds__(TPB) void gather_pano_kernel(PanoSrc s, int B, float* out) {
    const int n4 = (s.IMG + s.LOC) >> 2;
    const size_t total = (size_t)B * s.V * n4;
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < total; i += (size_t)gridDim.x * TPB) {
        const int c = (int)(i % n4);
        const size_t bv = i / n4;
        const int v = (int)(bv % s.V), b = (int)(bv / s.V);
        reinterpret_cast<float4*>(out)[i] = pano_chunk(s, b, v, c);
    }
}
__global__ __launch_bounds__(TPB) void gather_cand_kernel(CandSrc s, int B, float* all_u,
                                                          float* is_valid) {
    const int n4 = (s.IMG + s.LOC) >> 2;
    const size_t total = (size_t)B * s.A * n4;
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < total; i += (size_t)gridDim.x * TPB) {
        const int c = (int)(i % n4);
        const size_t ba = i / n4;
        const int a = (int)(ba % s.A), b = (int)(ba / s.A);
        reinterpret_cast<float4*>(all_u)[i] = cand_chunk(s, b, a, c);
        if (c == 0 && is_valid) is_valid[ba] = a < s.a_num[b] ? 1.f : 0.f;
    }
}
__global__ __launch_bounds__(TPB) void gather_action_kernel(CandSrc s, int B, const int* act,
                                                            float* out, int ldo4) {     // ldo4: row stride in float4
    const int n4 = (s.IMG + s.LOC) >> 2;
    const size_t total = (size_t)B * n4;
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < total; i += (size_t)gridDim.x * TPB) {
        const int c = (int)(i % n4), b = (int)(i / n4);
        const int a = act[b];
        const CandRow row = cand_row(s, b, a);                   // a <= 0, padding or vp < 0: zeros
        reinterpret_cast<float4*>(out)[(size_t)b * ldo4 + c] = cand_load(row, c, a > 0 && !row.zero, n4);
    }
}

// The chosen-action embeddings of ALL path steps of a speaker batch (speaker.py:87-104: `action_embedding[a]` of
// every (step, path), zeros for stop actions and padded steps) in one launch, written with a row stride -- i.e.
// straight into the first half of the encoder's LSTM inputs.  Row n: table[vp[n], act_view[n]] || sin/cos groups.
__global__ __launch_bounds__(TPB) void gather_path_actions_kernel(const float* table, int V, int IMG, int LOC, const int* vp,
                                                                  const int* act_view, const float* sincos, const int* act,
                                                                  int N, float* out, int ldo) {
    const int n4 = (IMG + LOC) >> 2;
    const size_t total = (size_t)N * n4;
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < total; i += (size_t)gridDim.x * TPB) {
        const int c = (int)(i % n4), n = (int)(i / n4);
        const int v = vp[n], view = act_view[n];
        const float4 sc = reinterpret_cast<const float4*>(sincos)[n];
        CandRow r;
        r.I4 = IMG >> 2;
        r.g4 = max(LOC >> 4, 1);
        r.img = reinterpret_cast<const float4*>(table) + ((size_t)max(v, 0) * V + min(max(view, 0), V - 1)) * r.I4;
        r.s0 = sc.x; r.s1 = sc.y; r.s2 = sc.z; r.s3 = sc.w;
        r.zero = false;
        reinterpret_cast<float4*>(out + (size_t)n * ldo)[c] = cand_load(r, c, act[n] > 0 && v >= 0, n4);
    }
}

// ---- search helpers (follower.py:541-980, speaker.py:211-318) ---------------------------------------
// dst[i, :w] = src[idx[i], :w]  (idx < 0 => zeros): `h_t[flat_indices]`, `c_t[flat_indices]`
__global__ __launch_bounds__(TPB) void gather_rows_kernel(const float* src, int lds, const int* idx,
                                                          int n, int w, float* dst, int ldd) {
    const int w4 = w >> 2;
    const size_t total = (size_t)n * w4;
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < total; i += (size_t)gridDim.x * TPB) {
        const int c = (int)(i % w4), r = (int)(i / w4);
        const int s = idx[r];
        float4 v = f4zero();
        if (s >= 0) v = reinterpret_cast<const float4*>(src + (size_t)s * lds)[c];
        reinterpret_cast<float4*>(dst + (size_t)r * ldd)[c] = v;
    }
}

// dst[idx[i], :w] = src[i, :w]  (idx < 0: row skipped): new search states into the rows of the state pool
__global__ __launch_bounds__(TPB) void scatter_rows_kernel(const float* src, int lds, const int* idx,
                                                           int n, int w, float* dst, int ldd) {
    const int w4 = w >> 2;
    const size_t total = (size_t)n * w4;
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < total; i += (size_t)gridDim.x * TPB) {
        const int c = (int)(i % w4), r = (int)(i / w4);
        const int d = idx[r];
        if (d >= 0)
            reinterpret_cast<float4*>(dst + (size_t)d * ldd)[c] = reinterpret_cast<const float4*>(src + (size_t)r * lds)[c];
    }
}

// sf_move_rows: several gathers / scatters over n rows in one launch (blockIdx.y = the move): h and c of the expanded
// search states out of the state pool, h / c / attention of the new states into it.
__global__ __launch_bounds__(TPB) void move_rows_kernel(RowMoves mv, int n) {
    const RowMoves::M m = mv.m[blockIdx.y];
    const int w4 = m.w >> 2;
    const size_t total = (size_t)n * w4;
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < total; i += (size_t)gridDim.x * TPB) {
        const int c = (int)(i % w4), r = (int)(i / w4);
        const int k = m.idx[r];
        if (m.scatter) {
            if (k >= 0)
                reinterpret_cast<float4*>(m.dst + (size_t)k * m.ldd)[c] =
                    reinterpret_cast<const float4*>(m.src + (size_t)r * m.lds)[c];
        } else {
            float4 v = f4zero();
            if (k >= 0) v = reinterpret_cast<const float4*>(m.src + (size_t)k * m.lds)[c];
            reinterpret_cast<float4*>(m.dst + (size_t)r * m.ldd)[c] = v;
        }
    }
}

// Masked log-softmax + the k best columns of every row in descending order (ties: lower column
// first).  One block per row, up to TOPK_E * TPB columns, k rounds of a block-wide arg-max.
constexpr int TOPK_E = 4;
__global__ __launch_bounds__(TPB) void logprob_topk_kernel(float* logit, int ld, int n,
                                                           const int* n_valid, int k, int* idx,
                                                           float* logp) {
    __shared__ float s_v[TPB / 64];
    __shared__ int s_i[TPB / 64];
    __shared__ float s_red[TPB / 64];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* x = logit + (size_t)row * ld;
    const int nv = n_valid ? min(n_valid[row], n) : n;
    float v[TOPK_E];
    float m = -INFINITY;
#pragma unroll
    for (int e = 0; e < TOPK_E; ++e) {
        const int c = tid + TPB * e;
        v[e] = c < nv ? x[c] : -INFINITY;
        if (n_valid && c >= nv && c < n) x[c] = -INFINITY;        // follower.py:585 logit[is_valid == 0] = -inf
        m = fmaxf(m, v[e]);
    }
    m = wave_max(m);
    if (lane == 0) s_red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
    __syncthreads();
    float se = 0.f;
#pragma unroll
    for (int e = 0; e < TOPK_E; ++e) se += (tid + TPB * e < nv) ? expf(v[e] - m) : 0.f;
    se = wave_sum(se);
    if (lane == 0) s_red[wave] = se;
    __syncthreads();
    const float lse = logf((s_red[0] + s_red[1]) + (s_red[2] + s_red[3]));
    if (!idx) {                                  // the whole row in COLUMN order (k == n): log_softmax, -inf beyond n_valid
#pragma unroll
        for (int e = 0; e < TOPK_E; ++e) {
            const int c = tid + TPB * e;
            if (c < n) logp[(size_t)row * k + c] = c < nv ? (v[e] - m) - lse : -INFINITY;
        }
        return;
    }
    unsigned taken = 0;
    for (int r = 0; r < k; ++r) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int e = 0; e < TOPK_E; ++e) {
            const int c = tid + TPB * e;
            if (c < n && !((taken >> e) & 1u) && (v[e] > bv || bi == 0x7fffffff)) {
                bv = v[e];
                bi = c;
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(bv, off, WAVE);
            const int oi = __shfl_xor(bi, off, WAVE);
            if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > bv || (ov == bv && oi < bi))) {
                bv = ov;
                bi = oi;
            }
        }
        __syncthreads();
        if (lane == 0) {
            s_v[wave] = bv;
            s_i[wave] = bi;
        }
        __syncthreads();
        bv = s_v[0];
        bi = s_i[0];
#pragma unroll
        for (int w = 1; w < TPB / 64; ++w) {
            const float ov = s_v[w];
            const int oi = s_i[w];
            if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > bv || (ov == bv && oi < bi))) {
                bv = ov;
                bi = oi;
            }
        }
        if (bi != 0x7fffffff && (bi % TPB) == tid) taken |= 1u << (bi / TPB);
        if (tid == 0) {
            idx[(size_t)row * k + r] = bi == 0x7fffffff ? -1 : bi;
            logp[(size_t)row * k + r] = bi == 0x7fffffff ? -INFINITY : (bv - m) - lse;
        }
    }
}

// ---- follower per-step glue (follower.py:476-505): one wave per sample -----------------------------
__global__ __launch_bounds__(TPB) void follower_glue_kernel(FGlue g) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * (TPB / 64) + (threadIdx.x >> 6);
    if (b >= g.B) return;
    const int A = g.src.A;
    const FGlueIn gin = follower_glue_load(g, b);
    const float raw = g.logit[(size_t)b * A + min(lane, A - 1)];
    const int at = follower_glue_row(g, b, raw, gin);
    if (g.nav.on && lane < A) nav_advance_slot(g.nav, b, lane, at, gin.was_ended || at == 0);
    if (g.u_next) {                                              // follower.py:502
        const int n4 = (g.src.IMG + g.src.LOC) >> 2;
        const CandRow row = cand_row(g.src, b, at);
        for (int c0 = 0; c0 < n4; c0 += 64 * 9) {               // 9 straight-line loads per pass
            float4 x[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) x[i] = cand_load(row, c0 + lane + 64 * i, !row.zero, n4);
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                const int c = c0 + lane + 64 * i;
                if (c < n4) store_u_next(g, b, c, x[i]);
            }
        }
    }
}

// softmax - onehot, scaled; rows with target == ignore get zeros.  One wave per row, any N.
__global__ __launch_bounds__(TPB) void softmax_ce_bwd_kernel(int B, int N, int ld,
                                                             const float* logit,
                                                             const int64_t* target, int ignore,
                                                             const float* gscale, float* dlogit, int rps) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * (TPB / 64) + (threadIdx.x >> 6);
    if (b >= B) return;
    const int64_t tgt = target[b];
    float* out = dlogit + (size_t)b * ld;
    if (tgt == ignore) {
        for (int n = lane; n < ld; n += 64) out[n] = 0.f;
        return;
    }
    const float* row = logit + (size_t)b * ld;
    float m = -INFINITY;
    for (int n = lane; n < N; n += 64) m = fmaxf(m, row[n]);
    m = wave_max(m);
    float s = 0.f;
    for (int n = lane; n < N; n += 64) s += expf(row[n] - m);
    s = wave_sum(s);
    const float gs = gscale[rps ? b / rps : 0];          // (stacked steps: rows [t rps, +rps) belong to step t)
    for (int n = lane; n < ld; n += 64) {
        float v = 0.f;
        if (n < N) v = gs * (expf(row[n] - m) / s - (n == tgt ? 1.f : 0.f));
        out[n] = v;
    }
}

// speaker.py:163-191: one wave per sample over the vocabulary
struct SGlue {
    int B, vocab, ldv;
    const float* logit;
    const int64_t* target;
    int feedback, pad_idx, eos_idx;
    uint8_t* ended;
    int64_t* w_t;
    float* score;
    float* nll_term;
    float* live;
    uint32_t sample_seed, sample_stream; int sample_row0;     // feedback 2
    const uint32_t* sample_site;                              // device-side stream offset (never null), see Dropout.site
    int rps;                                                  // > 0: S stacked steps of rps rows: row m sets ended[m % rps]
};
// feedback 2 (speaker.py:170-174): the two-level draw of sf_sampling.h by one wave.  Lane l holds columns
// [16 l, 16 l + 16); slot s = lanes 2 s, 2 s + 1.  m = the row's max, am its arg max.  Returns the word.
__device__ __forceinline__ int speaker_sample_row(const SGlue& g, const float* row, int b, float m, int am) {
    const int lane = threadIdx.x & 63;
    float x[16], e[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) x[j] = row[min(16 * lane + j, g.vocab - 1)];
    float ml = -INFINITY;
#pragma unroll
    for (int j = 0; j < 16; ++j) ml = 16 * lane + j < g.vocab ? fmaxf(ml, x[j]) : ml;
    const float ms = fmaxf(ml, __shfl_xor(ml, 1, WAVE));
    float sl = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        e[j] = 16 * lane + j < g.vocab ? expf(x[j] - ms) : 0.f;
        sl += e[j];
    }
    const float so = __shfl_xor(sl, 1, WAVE);
    const float zs = (lane & 1) ? so + sl : sl + so;
    float u1, u2;
    sample_uniforms(g.sample_seed + 0x9E3779B9u * site_value(g.sample_site), g.sample_stream, (uint32_t)(g.sample_row0 + b), &u1, &u2);
    // level 2: this slot's column
    const float thr2 = u2 * zs;
    float cum = (lane & 1) ? so : 0.f;
    int pick = 0x7FFFFFFF;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        cum += e[j];
        if (pick == 0x7FFFFFFF && cum > thr2 && e[j] > 0.f) pick = 16 * lane + j;
    }
    pick = min(pick, __shfl_xor(pick, 1, WAVE));
    if (pick == 0x7FFFFFFF) pick = min(32 * (lane >> 1) + 31, g.vocab - 1);
    // level 1: the slot
    const float ps = (lane & 1) ? 0.f : zs * wexp(ms, m);
    float cdf = ps;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float v = __shfl_up(cdf, off, WAVE);
        if (lane >= off) cdf += v;
    }
    const float Z = __shfl(cdf, 63, WAVE);
    const unsigned long long hit = __ballot(ps > 0.f && cdf > u1 * Z);
    const int sl_lane = hit ? (int)__ffsll((long long)hit) - 1 : 2 * (am >> 5);
    return __shfl(pick, sl_lane, WAVE);
}
__global__ __launch_bounds__(TPB) void speaker_glue_kernel(SGlue g) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * (TPB / 64) + (threadIdx.x >> 6);
    if (b >= g.B) return;
    const float* row = g.logit + (size_t)b * g.ldv;
    // the whole row in registers with straight-line loads (vocab <= 1024 on this path), the target
    // and its logit fetched alongside: one memory round trip instead of ~35 dependent ones
    constexpr int NV = 16;
    float v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = row[min(lane + 64 * i, g.vocab - 1)];
    const int64_t tgt = g.target[b];
    const float ltgt = row[min(max((int)tgt, 0), g.vocab - 1)];
    float m = -INFINITY;
    int am = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int n = lane + 64 * i;
        if (n < g.vocab && v[i] > m) { m = v[i]; am = n; }      // strict > keeps the lowest index per lane
    }
    for (int n = lane + 64 * NV; n < g.vocab; n += 64) {       // (larger vocabularies: the slow way)
        const float x = row[n];
        if (x > m) { m = x; am = n; }
    }
    // wave arg-max with lowest-index tie break (torch.max semantics, speaker.py:169)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float om = __shfl_xor(m, off, WAVE);
        const int oa = __shfl_xor(am, off, WAVE);
        if (om > m || (om == m && oa < am)) { m = om; am = oa; }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += (lane + 64 * i < g.vocab) ? expf(v[i] - m) : 0.f;
    for (int n = lane + 64 * NV; n < g.vocab; n += 64) s += expf(row[n] - m);
    const float lse = m + logf(wave_sum(s));
    int ws = 0;
    float lws = 0.f;
    if (g.feedback == 2) {                                       // block-uniform
        ws = speaker_sample_row(g, row, b, m, am);
        lws = row[ws];
    }
    if (lane == 0) {
        const int64_t w = g.feedback == 0 ? tgt : (g.feedback == 1 ? (int64_t)am : (int64_t)ws);
        const float lw = g.feedback == 0 ? ltgt : (g.feedback == 1 ? m : lws);   // logit of the chosen word
        g.w_t[b] = w;
        g.score[b] = (w != g.pad_idx) ? lw - lse : 0.f;          // speaker.py:179-180 (per-step term)
        const bool lv = tgt != g.pad_idx;
        g.nll_term[b] = lv ? lse - ltgt : 0.f;                   // speaker.py:182
        g.live[b] = lv ? 1.f : 0.f;
        if (w == g.eos_idx) g.ended[g.rps ? b % g.rps : b] = 1;  // speaker.py:190-191
    }
}

// deterministic per-step reduction of the loss terms: sum_cnt[t] = (sum_b term, sum_b live)
__global__ __launch_bounds__(64) void reduce_terms_kernel(const float* term, const float* live,
                                                          int B, float* sum_cnt) {
    const int t = blockIdx.x, lane = threadIdx.x;
    float s = 0.f, c = 0.f;
    for (int b = lane; b < B; b += 64) {
        s += term[(size_t)t * B + b];
        c += live[(size_t)t * B + b];
    }
    s = wave_sum(s);
    c = wave_sum(c);
    if (lane == 0) {
        sum_cnt[2 * t] = s;
        sum_cnt[2 * t + 1] = c;
    }
}
// One wave.  Lane t loads step t's (sum, count) -- one round trip for 64 steps instead of one per step -- and the
// per-step means are then added in step order through shuffles: the same float32 sum as a serial loop.
__global__ __launch_bounds__(64) void loss_finalize_kernel(const float* sum_cnt, int T, float* loss, float* gscale) {
    const int lane = threadIdx.x;
    float acc = 0.f;
    for (int base = 0; base < T; base += 64) {
        const int t = base + lane;
        const float2 sc = t < T ? reinterpret_cast<const float2*>(sum_cnt)[t] : make_float2(0.f, 0.f);
        const float v = sc.y > 0.f ? sc.x / sc.y : 0.f;          // follower.py:481 / speaker.py:182
        if (t < T) gscale[t] = sc.y > 0.f ? 1.f / sc.y : 0.f;
        const int n = min(64, T - base);
        for (int j = 0; j < n; ++j) acc += __shfl(v, j, WAVE);
    }
    if (lane == 0) loss[0] = acc;
}

// The speaker's loss (speaker.py:182, 192-197): the per-step means are added only up to and including the first step
// at which EVERY row has produced EOS (`if ended.all(): break` sits behind the step's loss).  words [T+1,B] (row 0 =
// the start tokens).  One block of 1024 threads: thread (b mod .., t) finds each row's first EOS step by a min over
// steps, the block takes the max over rows, wave 0 adds the step means in step order (the float32 sum of a serial
// loop) and zeroes gscale behind the last step (no gradient flows from steps the reference never adds).
__global__ __launch_bounds__(1024) void speaker_loss_finalize_kernel(const float* sum_cnt, const int64_t* words, int eos,
                                                                     int T, int B, float* loss, float* gscale) {
    __shared__ int s_first[1024];              // per row (the B <= 1024 form): first step whose word is EOS
    __shared__ int s_end;
    const int tid = threadIdx.x;
    s_first[tid] = T - 1;                      // never ended: the reference's loop runs all T steps
    if (tid == 0) s_end = 0;
    __syncthreads();
    // every (step, row) is looked at once; the loads of a pass are issued together, straight-line on clamped indices
    // (a load behind the EOS test would cost one memory round trip per iteration)
    const int n = T * B;
    if (B <= 1024) {
        for (int base = 0; base < n; base += 8 * 1024) {
            int64_t w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) w[u] = words[(size_t)B + min(base + u * 1024 + tid, n - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = base + u * 1024 + tid;
                if (i < n && w[u] == eos) atomicMin(&s_first[i % B], i / B);
            }
        }
        __syncthreads();
        if (tid < B) atomicMax(&s_end, s_first[tid]);
    } else {
        // more rows than threads (the pragmatic re-ranking scores ~2 500 candidate routes as ONE batch,
        // rational_follower.py:67-69): every thread walks the steps of its own rows, eight loads in flight
        int mine = 0;
        for (int b = tid; b < B; b += 1024) {
            int first = T - 1;
            for (int t0 = 0; t0 < T; t0 += 8) {
                int64_t w[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) w[u] = words[(size_t)B * (size_t)(min(t0 + u, T - 1) + 1) + b];
#pragma unroll
                for (int u = 7; u >= 0; --u)
                    if (t0 + u < T && w[u] == eos) first = min(first, t0 + u);
            }
            mine = max(mine, first);
        }
        atomicMax(&s_end, mine);
    }
    __syncthreads();
    const int last = s_end;                                      // steps 0 .. last are added
    if (tid < 64) {
        const int lane = tid;
        float acc = 0.f;
        for (int base = 0; base < T; base += 64) {
            const int t = base + lane;
            const float2 sc = t < T ? reinterpret_cast<const float2*>(sum_cnt)[t] : make_float2(0.f, 0.f);
            const bool in = t <= last;
            const float v = (in && sc.y > 0.f) ? sc.x / sc.y : 0.f;
            if (t < T) gscale[t] = (in && sc.y > 0.f) ? 1.f / sc.y : 0.f;
            const int n = min(64, T - base);
            for (int j = 0; j < n; ++j) acc += __shfl(v, j, WAVE);
        }
        if (lane == 0) loss[0] = acc;
    }
}

// torch.optim.Adam.step() (train.py:263-268: lr 1e-4, weight_decay 5e-4 as L2-in-gradient, default
// betas / eps, no amsgrad) over one flat parameter range, in the operation order of torch's own
// implementation (lerp for exp_avg, mul + addcmul for exp_avg_sq, sqrt / sqrt(bc2) + eps, addcdiv).
struct AdamArgs {
    float* p; const float* g; float* m; float* v; size_t n;
    float beta2, om_beta1, om_beta2, eps, wd, step_size, inv_sqrt_bc2;   // om = 1 - beta, rounded from double
    const float* coef;          // device-side {step_size, inv_sqrt_bc2} (sf_adam_step_dev), or null
};
__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamArgs& a) {
    if (a.wd != 0.f) g = g + a.wd * p;
    m = m + a.om_beta1 * (g - m);
    v = v * a.beta2 + (a.om_beta2 * g) * g;
    const float denom = sqrtf(v) * a.inv_sqrt_bc2 + a.eps;
    p = p - a.step_size * (m / denom);
}
// the step counter of a CAPTURED optimizer step lives on the device: ++step, then the two step-dependent constants in
// double, rounded to float once (what adam_step does on the host)
// `guard` (optional): a device word -- the fault word of the persistent launches -- that, when non-zero, turns this
// optimizer step into a no-op (coef[2] = 1: adam_kernel returns, the counter stays): a captured training iteration must
// not step on gradients a starved launch has poisoned, and nobody can look at the word between two nodes of a graph.
__global__ void adam_coef_kernel(int* step, double lr, double beta1, double beta2, float* coef, const unsigned* guard) {
    if (threadIdx.x != 0) return;
    if (guard && *guard != 0u) {
        coef[2] = 1.f;
        return;
    }
    const int s = *step + 1;
    *step = s;
    const double bc1 = 1.0 - pow(beta1, (double)s), bc2 = 1.0 - pow(beta2, (double)s);
    coef[0] = (float)(lr / bc1);
    coef[1] = (float)(1.0 / sqrt(bc2));
    coef[2] = 0.f;
}
__global__ __launch_bounds__(TPB) void adam_kernel(AdamArgs a) {
    if (a.coef) {
        if (a.coef[2] != 0.f) return;                            // guarded step: see adam_coef_kernel
        a.step_size = a.coef[0];
        a.inv_sqrt_bc2 = a.coef[1];
    }
    const size_t n4 = a.n >> 2;
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < n4; i += (size_t)gridDim.x * TPB) {
        float4 p = reinterpret_cast<float4*>(a.p)[i];
        const float4 g = reinterpret_cast<const float4*>(a.g)[i];
        float4 m = reinterpret_cast<float4*>(a.m)[i];
        float4 v = reinterpret_cast<float4*>(a.v)[i];
        adam_one(p.x, g.x, m.x, v.x, a);
        adam_one(p.y, g.y, m.y, v.y, a);
        adam_one(p.z, g.z, m.z, v.z, a);
        adam_one(p.w, g.w, m.w, v.w, a);
        reinterpret_cast<float4*>(a.p)[i] = p;
        reinterpret_cast<float4*>(a.m)[i] = m;
        reinterpret_cast<float4*>(a.v)[i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {            // tail (n % 4 elements)
        const size_t i = (n4 << 2) + threadIdx.x;
        adam_one(a.p[i], a.g[i], a.m[i], a.v[i], a);
    }
}

}  // namespace

int lstm_pointwise_fwd(const LstmPwFwd& a, hipStream_t st) {
    const dim3 grid(grid1d((size_t)a.B * a.H));
    switch (a.ks) {
#define SF_PW(KS) case KS: SF_LAUNCH(lstm_pw_fwd_kernel<KS>, grid, dim3(TPB), 0, st, a); break;
        SF_PW(0) SF_PW(1) SF_PW(2) SF_PW(3) SF_PW(4) SF_PW(5) SF_PW(6) SF_PW(7) SF_PW(8)
        SF_PW(9) SF_PW(10) SF_PW(11) SF_PW(12) SF_PW(13) SF_PW(14) SF_PW(15) SF_PW(16)
#undef SF_PW
        default: return SF_ERR_UNSUPPORTED;
    }
    return launch_status();
}
int lstm_pointwise_bwd(const LstmPwBwd& a, hipStream_t st) {
    SF_LAUNCH(lstm_pw_bwd_kernel, dim3(grid1d((size_t)a.B * a.H)), dim3(TPB), 0, st, a);
    return launch_status();
}
int dropout_copy(const float* src, int lds, int B, int N, float* dst, int ldd, const Dropout& d,
                 int col0, hipStream_t st) {
    SF_LAUNCH(dropout_copy_kernel, dim3(grid1d((size_t)B * N)), dim3(TPB), 0, st, src, lds,
                       B, N, dst, ldd, d, col0);
    return launch_status();
}
int dropout_steps(const float* src, int lds, int S, int B, int N, float* dst, int ldd, const Dropout& d,
                  uint32_t stream_step, hipStream_t st) {
    SF_LAUNCH(dropout_steps_kernel, dim3(grid1d((size_t)S * B * N)), dim3(TPB), 0, st, src, lds, S, B, N, dst, ldd, d,
              stream_step);
    return launch_status();
}
int row_mod(int* out, int M, int B, hipStream_t st) {
    SF_LAUNCH(row_mod_kernel, dim3(ceil_div(M, 256)), dim3(256), 0, st, out, M, B);
    return launch_status();
}
int ctx_grad_slice(const float* dctx, int T, int H, int B, int t, const Dropout& d, float* out,
                   hipStream_t st) {
    SF_LAUNCH(ctx_grad_slice_kernel, dim3(grid1d((size_t)B * H)), dim3(TPB), 0, st, dctx, T,
                       H, B, t, d, out);
    return launch_status();
}
template <int OP>
static int ew(const float* a, int lda, const float* b, int ldb, int M, int N, float* dst, int ldd,
              hipStream_t st) {
    EwArgs e{a, lda, b, ldb, M, N, dst, ldd};
    SF_LAUNCH(ew_kernel<OP>, dim3(grid1d((size_t)M * N)), dim3(TPB), 0, st, e);
    return launch_status();
}
int add2(const float* a, int lda, const float* b, int ldb, int M, int N, float* dst, int ldd,
         hipStream_t st) {
    return ew<EW_ADD2>(a, lda, b, ldb, M, N, dst, ldd, st);
}
int tanh_bwd(const float* y, int ldy, const float* dy, int lddy, int M, int N, float* dpre, int ldp,
             hipStream_t st) {
    return ew<EW_TANH_BWD>(y, ldy, dy, lddy, M, N, dpre, ldp, st);
}
int scale_cols(const float* src, int lds, const float* v, int M, int N, float* dst, int ldd,
               hipStream_t st) {
    return ew<EW_SCALE_COLS>(src, lds, v, 0, M, N, dst, ldd, st);
}
int rank1_add(const float* s, const float* v, int M, int N, float* dst, int ldd, hipStream_t st) {
    return ew<EW_RANK1_ADD>(s, 0, v, 0, M, N, dst, ldd, st);
}
int colsum_prod(const float* a, int lda, const float* b, int ldb, int M, int N, float* out,
                hipStream_t st) {
    SF_LAUNCH(colsum_prod_kernel, dim3(ceil_div(N, 64)), dim3(1024), 0, st, a, lda, b, ldb,
                       M, N, out);
    return launch_status();
}
int dot_rows_accum(const float* s, const float* x, int ldx, int M, int N, float* out,
                   hipStream_t st) {
    SF_LAUNCH(dot_rows_kernel, dim3(ceil_div(N, 64)), dim3(1024), 0, st, s, x, ldx, M, N,
                       out);
    return launch_status();
}
int adam_step(float* p, const float* g, float* m, float* v, size_t n, double lr, double beta1,
              double beta2, double eps, double wd, int step, hipStream_t st) {
    // every derived constant in double, rounded to float once -- what torch does with its Python
    // floats (1 - 0.999 evaluated in float would be off by 5e-5 relative)
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    AdamArgs a{p, g, m, v, n, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps,
               (float)wd, (float)(lr / bc1), (float)(1.0 / sqrt(bc2)), nullptr};
    const size_t n4 = (n + 3) >> 2;
    SF_LAUNCH(adam_kernel, dim3((unsigned)std::min<size_t>((n4 + TPB - 1) / TPB, 4096)), dim3(TPB), 0,
                       st, a);
    return launch_status();
}
int adam_step_dev(float* p, const float* g, float* m, float* v, size_t n, double lr, double beta1, double beta2,
                  double eps, double wd, int* step_dev, float* coef, const unsigned* guard, hipStream_t st) {
    SF_LAUNCH(adam_coef_kernel, dim3(1), dim3(64), 0, st, step_dev, lr, beta1, beta2, coef, guard);
    AdamArgs a{p, g, m, v, n, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps,
               (float)wd, 0.f, 0.f, coef};
    const size_t n4 = (n + 3) >> 2;
    SF_LAUNCH(adam_kernel, dim3((unsigned)std::min<size_t>((n4 + TPB - 1) / TPB, 4096)), dim3(TPB), 0,
                       st, a);
    return launch_status();
}
int sum_accum(const float* s, int M, float* out, hipStream_t st) {
    SF_LAUNCH(sum_accum_kernel, dim3(1), dim3(64), 0, st, s, M, out);
    return launch_status();
}
// Cross-stream ordering without events: a one-wave kernel that spins on a device word until it
// reaches `target` (bounded by the wall clock), and a one-thread kernel that publishes a value.  A
// stream that launches flag_wait ahead of a consumer cannot start the consumer before another stream's
// flag_set (launched behind the producer) has run; kernel boundaries do the release / acquire of the
// data itself.  An event-based fork / join per decode step costs several times more on this stack.
__global__ void flag_wait_kernel(const unsigned* flag, unsigned target) {
    if (threadIdx.x != 0) return;
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(8);
        if (wall_clock64() - t0 > 50000000LL) break;      // 0.5 s: never hang a stream
    }
}
__global__ void flag_set_kernel(unsigned* flag, unsigned value) {
    if (threadIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Consumer side of a one-shot flag: spin until the word is non-zero, then clear it for the next use.  A wait that gives
// up (0.5 s) raises `fault_code` in the fault word: the host re-issues the pass (runtime.take_fault).
__global__ void flag_wait_clear_kernel(unsigned* flag, unsigned* fault, unsigned fault_code) {
    if (threadIdx.x != 0) return;
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
        __builtin_amdgcn_s_sleep(4);
        if (wall_clock64() - t0 > 50000000LL) {           // never hang a stream
            if (fault) atomicOr(fault, fault_code);
            break;
        }
    }
    __hip_atomic_store(flag, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
int flag_wait_clear(unsigned* flag, unsigned* fault, unsigned fault_code, hipStream_t st) {
    SF_LAUNCH(flag_wait_clear_kernel, dim3(1), dim3(64), 0, st, flag, fault, fault_code);
    return launch_status();
}
int flag_wait(const unsigned* flag, unsigned target, hipStream_t st) {
    SF_LAUNCH(flag_wait_kernel, dim3(1), dim3(64), 0, st, flag, target);
    return launch_status();
}
int flag_set(unsigned* flag, unsigned value, hipStream_t st) {
    SF_LAUNCH(flag_set_kernel, dim3(1), dim3(64), 0, st, flag, value);
    return launch_status();
}

// ---- device-side site counters (include/sf_hip.h: sf_dropout.site_dev, sf_site_advance) -----------------------------
__device__ uint32_t g_site_zero_word = 0;
const uint32_t* site_zero() {
    // the symbol has one address PER DEVICE: cached by the current device's ordinal (a process that drives several
    // GPUs must not hand device 0's word to a kernel on device 1).  A failed lookup yields null, which the kernels read
    // as site 0 (site_value) -- never a wild pointer.
    constexpr int kMaxDev = 64;
    static std::atomic<const uint32_t*> cache[kMaxDev];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return nullptr;
    const uint32_t* p = cache[dev].load(std::memory_order_acquire);
    if (p) return p;
    void* q = nullptr;
    if (hipGetSymbolAddress(&q, HIP_SYMBOL(g_site_zero_word)) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    cache[dev].store(static_cast<const uint32_t*>(q), std::memory_order_release);
    return static_cast<const uint32_t*>(q);
}
__global__ void site_advance_kernel(uint32_t* word, uint32_t by) {
    if (threadIdx.x == 0) *word += by;
}
__global__ void store_u32x4_kernel(uint32_t* dst, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    if (threadIdx.x == 0) { dst[0] = a; dst[1] = b; dst[2] = c; dst[3] = d; }
}
int store_u32x4(uint32_t* dst, uint32_t a, uint32_t b, uint32_t c, uint32_t d, hipStream_t st) {
    SF_LAUNCH(store_u32x4_kernel, dim3(1), dim3(64), 0, st, dst, a, b, c, d);
    return launch_status();
}
// sf_fill_regions: the initial conditions of a pass (zero states, BOS words, cleared flags) in ONE launch instead of
// one fill per tensor.  blockIdx.y = region; elements of 1, 4 or 8 bytes.
__global__ __launch_bounds__(256) void fill_regions_kernel(FillRegions fr) {
    const FillRegions::R r = fr.r[blockIdx.y];
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r.width == 4)
        for (size_t i = i0; i < r.count; i += stride) ((uint32_t*)r.ptr)[i] = (uint32_t)r.value;
    else if (r.width == 8)
        for (size_t i = i0; i < r.count; i += stride) ((unsigned long long*)r.ptr)[i] = r.value;
    else
        for (size_t i = i0; i < r.count; i += stride) ((unsigned char*)r.ptr)[i] = (unsigned char)r.value;
}
int fill_regions(const FillRegions& fr, hipStream_t st) {
    size_t most = 1;
    for (int i = 0; i < fr.n; ++i) most = fr.r[i].count > most ? fr.r[i].count : most;
    const unsigned gx = (unsigned)((most + 256 * 4 - 1) / (256 * 4));
    SF_LAUNCH(fill_regions_kernel, dim3(gx < 1 ? 1 : (gx > 256 ? 256 : gx), fr.n), dim3(256), 0, st, fr);
    return launch_status();
}
int site_advance(uint32_t* word, uint32_t by, hipStream_t st) {
    SF_LAUNCH(site_advance_kernel, dim3(1), dim3(64), 0, st, word, by);
    return launch_status();
}

// Test co-tenant (sf_debug_cotenant): a kernel shaped like a collective's channel workgroups -- a few dozen large
// workgroups with a big LDS allocation that stay resident for a given time, touching LDS and a little global memory --
// to run beside the persistent launches on another stream (tests/test_gpu_cotenancy.py).
__global__ void cotenant_kernel(long long ticks, float* sink) {
    extern __shared__ float co_lds[];
    const long long t0 = wall_clock64();
    const int i = threadIdx.x, n = blockDim.x;
    float acc = (float)i;
    while (wall_clock64() - t0 < ticks) {
        co_lds[i] = acc;
        __syncthreads();
        acc = acc * 0.5f + co_lds[(i * 33 + 7) % n];
        __syncthreads();
        if (sink && i == 0) sink[blockIdx.x] = acc;
        __builtin_amdgcn_s_sleep(4);
    }
}
int cotenant(int blocks, int threads, int lds_bytes, long long ticks, float* sink, hipStream_t st) {
    if (blocks <= 0 || threads <= 0 || threads > 1024 || lds_bytes < threads * 4 || lds_bytes > 65536) return SF_ERR_ARG;
    SF_LAUNCH(cotenant_kernel, dim3(blocks), dim3(threads), (size_t)lds_bytes, st, ticks, sink);
    return launch_status();
}

int fill(float* p, size_t n, float v, hipStream_t st) {
    if (n == 0) return SF_OK;
    SF_LAUNCH(fill_kernel, dim3(grid1d(n)), dim3(TPB), 0, st, p, n, v);
    return launch_status();
}
int transpose(const float* src, int R, int C, float* dst, hipStream_t st) {
    SF_LAUNCH(transpose_kernel, dim3(ceil_div(C, 32), ceil_div(R, 32)), dim3(256), 0, st, src, C,
                       R, C, dst);
    return launch_status();
}
int transpose_ld(const float* src, int lds, int R, int C, float* dst, hipStream_t st) {     // dst [C,R] contiguous
    SF_LAUNCH(transpose_kernel, dim3(ceil_div(C, 32), ceil_div(R, 32)), dim3(256), 0, st, src, lds,
                       R, C, dst);
    return launch_status();
}
// Dropout of the embedded tokens of a trainable embedding (model.py:86-87), in place on the time-major tape
// x [T,B,E]: key (global row b, column t*E + e).
// `rev` (per-row lengths or null): step t of row b holds the token of position len_b - 1 - t (the reverse direction
// of a bidirectional encoder); the mask is keyed on the POSITION, so both directions drop the same embedded tokens
// (model.py:86-87 drops them once, ahead of the LSTM).
__device__ __forceinline__ int tm_position(const int* rev, int b, int t) {
    if (!rev) return t;
    const int len = rev[b];
    return t < len ? len - 1 - t : t;
}
__global__ __launch_bounds__(TPB) void dropout_tm_kernel(float* x, int T, int B, int E, Dropout d, const int* rev) {
    const size_t total = (size_t)T * B * E;
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < total; i += (size_t)gridDim.x * TPB) {
        const int e = (int)(i % E), b = (int)((i / E) % B), t = (int)(i / ((size_t)E * B));
        const uint32_t rk = drop_key(d, (uint32_t)(d.row0 + b));
        x[i] = dropout_keep(rk, (uint32_t)(tm_position(rev, b, t) * E + e), d.thresh) ? x[i] * d.scale : 0.f;
    }
}
// Embedding gradient (nn.Embedding backward): grad[tok(n), :] += mask x demb[n, :], n = (t, b) time-major (tok =
// seq[b*Lpad + t]) or n = b with step == 0 (tok = seq[b], Lpad = 1).  fp32 atomics: token rows collide.
__global__ __launch_bounds__(TPB) void embedding_bwd_kernel(const float* demb, int ldd, const int64_t* seq, int Lpad, int T,
                                                            int B, int E, int padding_idx, Dropout d, const int* rev,
                                                            float* grad) {
    const size_t total = (size_t)T * B * E;
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < total; i += (size_t)gridDim.x * TPB) {
        const int e = (int)(i % E), b = (int)((i / E) % B), t = (int)(i / ((size_t)E * B));
        const int64_t tok = seq[(size_t)b * Lpad + t];
        float v = demb[((size_t)t * B + b) * ldd + e];
        if (d.on()) {
            const uint32_t rk = drop_key(d, (uint32_t)(d.row0 + b));
            v = dropout_keep(rk, (uint32_t)(tm_position(rev, b, t) * E + e), d.thresh) ? v * d.scale : 0.f;
        }
        if (tok != padding_idx && v != 0.f) atomicAdd(grad + (size_t)tok * E + e, v);
    }
}
int dropout_tm(float* x, int T, int B, int E, const Dropout& d, const int* rev, hipStream_t st) {
    if (!d.on()) return SF_OK;
    SF_LAUNCH(dropout_tm_kernel, dim3(grid1d((size_t)T * B * E)), dim3(TPB), 0, st, x, T, B, E, d, rev);
    return launch_status();
}
int embedding_bwd(const float* demb, int ldd, const int64_t* seq, int Lpad, int T, int B, int E, int padding_idx,
                  const Dropout& d, const int* rev, float* grad, hipStream_t st) {
    SF_LAUNCH(embedding_bwd_kernel, dim3(grid1d((size_t)T * B * E)), dim3(TPB), 0, st, demb, ldd, seq, Lpad, T, B, E,
              padding_idx, d, rev, grad);
    return launch_status();
}
int embedding_tm(const float* table, int E, const int64_t* seq, int B, int Lpad, int T, float* out,
                 hipStream_t st) {
    if (E & 3) return SF_ERR_UNSUPPORTED;
    SF_LAUNCH(embedding_tm_kernel, dim3(grid1d((size_t)T * B * (E >> 2))), dim3(TPB), 0, st,
                       table, E, seq, B, Lpad, T, out);
    return launch_status();
}
int embedding_rows(const float* table, int E, const int64_t* idx, int B, float* out,
                   hipStream_t st) {
    if (E & 3) return SF_ERR_UNSUPPORTED;
    SF_LAUNCH(embedding_rows_kernel, dim3(grid1d((size_t)B * (E >> 2))), dim3(TPB), 0, st,
                       table, E, idx, B, out);
    return launch_status();
}
int gather_panorama(const PanoSrc& s, int B, float* out, hipStream_t st) {
    SF_LAUNCH(gather_pano_kernel,
                       dim3(grid1d((size_t)B * s.V * ((s.IMG + s.LOC) >> 2))), dim3(TPB), 0, st, s, B,
                       out);
    return launch_status();
}
int gather_candidates(const CandSrc& s, int B, float* all_u, float* is_valid, hipStream_t st) {
    SF_LAUNCH(gather_cand_kernel,
                       dim3(grid1d((size_t)B * s.A * ((s.IMG + s.LOC) >> 2))), dim3(TPB), 0, st, s, B,
                       all_u, is_valid);
    return launch_status();
}
int gather_actions(const CandSrc& s, int B, const int* a, float* out, hipStream_t st, int ldo) {
    const int F = s.IMG + s.LOC;
    if (ldo == 0) ldo = F;
    if ((ldo & 3) || ldo < F) return SF_ERR_UNSUPPORTED;
    SF_LAUNCH(gather_action_kernel, dim3(grid1d((size_t)B * (F >> 2))),
                       dim3(TPB), 0, st, s, B, a, out, ldo >> 2);
    return launch_status();
}
int gather_path_actions(const float* table, int V, int IMG, int LOC, const int* vp, const int* act_view,
                        const float* sincos, const int* act, int N, float* out, int ldo, hipStream_t st) {
    if ((IMG & 3) || (LOC & 15) || (ldo & 3)) return SF_ERR_UNSUPPORTED;
    SF_LAUNCH(gather_path_actions_kernel, dim3(grid1d((size_t)N * ((IMG + LOC) >> 2))), dim3(TPB), 0, st, table, V, IMG,
              LOC, vp, act_view, sincos, act, N, out, ldo);
    return launch_status();
}
int gather_rows(const float* src, int lds, const int* idx, int n, int w, float* dst, int ldd,
                hipStream_t st) {
    if ((w & 3) || (lds & 3) || (ldd & 3)) return SF_ERR_UNSUPPORTED;
    SF_LAUNCH(gather_rows_kernel, dim3(grid1d((size_t)n * (w >> 2))), dim3(TPB), 0, st, src,
                       lds, idx, n, w, dst, ldd);
    return launch_status();
}
int scatter_rows(const float* src, int lds, const int* idx, int n, int w, float* dst, int ldd,
                 hipStream_t st) {
    if ((w & 3) || (lds & 3) || (ldd & 3)) return SF_ERR_UNSUPPORTED;
    SF_LAUNCH(scatter_rows_kernel, dim3(grid1d((size_t)n * (w >> 2))), dim3(TPB), 0, st, src,
                       lds, idx, n, w, dst, ldd);
    return launch_status();
}
int move_rows(const RowMoves& mv, int n, hipStream_t st) {
    int widest = 4;
    for (int i = 0; i < mv.n; ++i) {
        if ((mv.m[i].w & 3) || (mv.m[i].lds & 3) || (mv.m[i].ldd & 3)) return SF_ERR_UNSUPPORTED;
        widest = mv.m[i].w > widest ? mv.m[i].w : widest;
    }
    SF_LAUNCH(move_rows_kernel, dim3(grid1d((size_t)n * (widest >> 2)), mv.n), dim3(TPB), 0, st, mv, n);
    return launch_status();
}
int logprob_topk(float* logit, int ld, int N, int n, const int* n_valid, int k, int* idx,
                 float* logp, hipStream_t st) {
    if (n > TOPK_E * TPB || k < 1 || k > n || (!idx && k != n)) return SF_ERR_UNSUPPORTED;
    SF_LAUNCH(logprob_topk_kernel, dim3(N), dim3(TPB), 0, st, logit, ld, n, n_valid, k, idx,
                       logp);
    return launch_status();
}
int follower_glue_fwd(const FGlue& g, hipStream_t st) {
    if (g.src.A > 64) return SF_ERR_UNSUPPORTED;
    SF_LAUNCH(follower_glue_kernel, dim3(ceil_div(g.B, TPB / 64)), dim3(TPB), 0, st, g);
    return launch_status();
}
int softmax_ce_bwd(int B, int N, int ld, const float* logit, const int64_t* target, int ignore,
                   const float* gscale, float* dlogit, hipStream_t st, int rps) {
    SF_LAUNCH(softmax_ce_bwd_kernel, dim3(ceil_div(B, TPB / 64)), dim3(TPB), 0, st, B, N,
                       ld, logit, target, ignore, gscale, dlogit, rps);
    return launch_status();
}
int speaker_glue_fwd(int B, int vocab, int ldv, const float* logit, const int64_t* target,
                     int feedback, int pad_idx, int eos_idx, uint8_t* ended, int64_t* w_t,
                     float* score, float* nll_term, float* live, hipStream_t st, const sf_sample* sample, int rps) {
    if (feedback == 2 && (!sample || vocab > 1024)) return feedback == 2 && !sample ? SF_ERR_ARG : SF_ERR_UNSUPPORTED;
    SGlue g{B, vocab, ldv, logit, target, feedback, pad_idx, eos_idx, ended, w_t, score, nll_term,
            live, sample ? sample->seed : 0u, sample ? sample->stream : 0u, sample ? sample->row0 : 0,
            sample && sample->stream_dev ? sample->stream_dev : site_zero(), rps};
    SF_LAUNCH(speaker_glue_kernel, dim3(ceil_div(B, TPB / 64)), dim3(TPB), 0, st, g);
    return launch_status();
}
int reduce_terms(const float* term, const float* live, int T, int B, float* sum_cnt,
                 hipStream_t st) {
    SF_LAUNCH(reduce_terms_kernel, dim3(T), dim3(64), 0, st, term, live, B, sum_cnt);
    return launch_status();
}
int speaker_loss_finalize(const float* sum_cnt, const int64_t* words, int eos, int T, int B, float* loss, float* gscale,
                          hipStream_t st) {
    if (T > 1024) return SF_ERR_UNSUPPORTED;
    SF_LAUNCH(speaker_loss_finalize_kernel, dim3(1), dim3(1024), 0, st, sum_cnt, words, eos, T, B, loss, gscale);
    return launch_status();
}
int loss_finalize(const float* sum_cnt, int T, float* loss, float* gscale, hipStream_t st) {
    SF_LAUNCH(loss_finalize_kernel, dim3(1), dim3(64), 0, st, sum_cnt, T, loss, gscale);
    return launch_status();
}

}  // namespace sf
