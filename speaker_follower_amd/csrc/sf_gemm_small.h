// The "all loads up front" small-K NT GEMM as a device-side body, shared by the stand-alone kernel
// (sf_gemm.hip) and the paired launches (sf_attention.hip) that run it side by side with an
// attention body in one grid.
#pragma once
#include "sf_gemm.h"
#include "sf_lstm_pw.h"

namespace sf {

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float comp(const float4& v, int c) {
    return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w));
}

struct Seg2 {             // up to two K segments, resolved per chunk with uniform selects
    Seg s0, s1;
    int n0;               // chunks in s0
    int total;            // chunks in s0 + s1
};

template <int MT, int CPW>
struct Frags {
    float4 b[CPW];
    float4 a[CPW][MT];
};

// a = tanh(z + y) of one float4 (the A-prologue, see SmallArgs::apro_part)
__device__ __forceinline__ float4 apro_form(const float4& y, const float4& z) {
    return make_float4(tanhf(z.x + y.x), tanhf(z.y + y.y), tanhf(z.z + y.z), tanhf(z.w + y.w));
}

// The same loads with the A-prologue: one segment (K = sg.s0.K), A = y [M, lda], z [M, stride].
template <int MT, int CPW>
__device__ __forceinline__ void upfront_load_apro(Frags<MT, CPW>& f, const Seg2& sg, int c_lo, int c_hi, int n,
                                                  const int (&mrow)[MT], int kk, const float* zbuf, int stride) {
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const int K = sg.s0.K;
    float4 bv[CPW], yv[CPW][MT], zv[CPW][MT];
    bool ok[CPW];
    // every load first (straight line), the transcendental arithmetic behind them
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
        const int c = min(c_lo + i, c_hi - 1);
        const int k = c * 16 + 4 * kk;
        ok[i] = k < K;
        const int kc = ok[i] ? k : 0;
        bv[i] = ld4(sg.s0.W + (size_t)n * sg.s0.ldw + kc);
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            yv[i][t] = ld4(sg.s0.A + (size_t)mrow[t] * sg.s0.lda + kc);
            zv[i][t] = ld4(zbuf + (size_t)mrow[t] * stride + kc);
        }
    }
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
        f.b[i] = ok[i] ? bv[i] : zero;
#pragma unroll
        for (int t = 0; t < MT; ++t) f.a[i][t] = ok[i] ? apro_form(yv[i][t], zv[i][t]) : zero;
    }
}

template <int MT, int CPW>
__device__ __forceinline__ void upfront_load(Frags<MT, CPW>& f, const Seg2& sg, int c_lo, int c_hi,
                                             int n, const int (&mrow)[MT], int kk) {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
        const int c = min(c_lo + i, c_hi - 1);           // clamped: surplus slots re-load a valid chunk
        const bool second = c >= sg.n0;
        const int lc = second ? c - sg.n0 : c;
        const float* W = second ? sg.s1.W : sg.s0.W;
        const float* A = second ? sg.s1.A : sg.s0.A;
        const int ldw = second ? sg.s1.ldw : sg.s0.ldw;
        const int lda = second ? sg.s1.lda : sg.s0.lda;
        const int K = second ? sg.s1.K : sg.s0.K;
        const int k = lc * 16 + 4 * kk;
        const bool ok = k < K;                           // partial last chunk (K % 16 != 0)
        const int kc = ok ? k : 0;
        const float4 bv = ld4(W + (size_t)n * ldw + kc);
        f.b[i] = ok ? bv : z;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const float4 av = ld4(A + (size_t)mrow[t] * lda + kc);
            f.a[i][t] = ok ? av : z;
        }
    }
}

template <int MT, int CPW>
__device__ __forceinline__ void upfront_mma(const Frags<MT, CPW>& f, int cnt, f32x4 (&acc)[MT]) {
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
        if (i < cnt) {                                   // wave-uniform
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int t = 0; t < MT; ++t)
                    acc[t] = mfma16(comp(f.a[i][t], c), comp(f.b[i], c), acc[t]);
        }
    }
}

constexpr int SMALL_WAVES = 8;

struct SmallArgs {
    Seg2 sg;
    int M, N;
    float* y;
    int ldy;
    const float* bias;
    const float* bias2;
    int epi;
    const float* mul;
    float* y_pre;
    int ldy_pre;
    int accumulate;
    const float* aux; int ld_aux;          // see LinearOut
    const float* addend; int ld_addend;
    const float* r1_s; const float* r1_v;
    // A-PROLOGUE (APRO bodies only; the folded inference text stage, sf_attention.hip: text_fold_body): the A operand is
    // not read but FORMED on the fly, a[row, k] = tanh(z[row, k] + y[row, k]), where y = the `A` of segment 0 and
    // z = `apro_part` [M, apro_stride]: the merged attention sum the text_fold launch left.
    const float* apro_part; int apro_stride;
};

// Block (bx, by) of the grid (ceil(N/16), ceil(mtiles/MT)); 512 threads = 8 waves = 8 K-slices of one
// 16-col n-tile x MT m-tiles.  Threads >= 512 of a larger (paired) block must not enter.
// EXTRA: the fused backward epilogues (aux / addend / rank-1 operands); a separate instantiation so
// that the forward products carry none of their registers or branches.
// PW: the LSTM cell's pointwise backward as the epilogue (sf_lstm_pw.h): the product completes dh1 of a decoder step
// (y += ...), and element (row, col) -- all it needs of dh1 -- goes straight through the cell's backward: dgates and dc0
// of that step without the stand-alone launch.
// APRO: the A operand is formed by the A-prologue (SmallArgs::apro_part) instead of loaded.
template <int MT, int CPW, bool EXTRA = false, bool PW = false, bool APRO = false>
__device__ __forceinline__ void small_gemm_body(const SmallArgs& a, int bx, int by, const LstmPwBwd* pw = nullptr) {
    __shared__ float s_part[SMALL_WAVES][MT][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = bx * 16, m0 = by * (16 * MT);
    const int li = lane & 15, kk = lane >> 4;
    const int n = min(n0 + li, a.N - 1);
    int mrow[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) mrow[t] = min(m0 + 16 * t + li, a.M - 1);

    // epilogue operands of this thread's column, fetched with the fragments (not after the MFMAs)
    const int ecol = min(n0 + (int)(threadIdx.x & 15), a.N - 1);
    float e_bias = 0.f, e_mul = 1.f;
    if (a.bias) e_bias = a.bias[ecol];                 // block-uniform branches, one load each
    if (a.bias2) e_bias += a.bias2[ecol];
    if (a.epi == EPI_MUL) e_mul = a.mul[ecol];
    // element-wise operands of the fused backward epilogues: this thread's elements are known now,
    // so their loads travel with the fragments instead of sitting behind the MFMAs
    constexpr int EPT = (MT * 256 + SMALL_WAVES * 64 - 1) / (SMALL_WAVES * 64);
    float e_add[EPT], e_aux[EPT], e_r1[EPT];
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const int e = min((int)threadIdx.x + i * SMALL_WAVES * 64, MT * 256 - 1);
        const int row = min(m0 + 16 * (e >> 8) + ((e & 255) >> 4), a.M - 1);
        e_add[i] = e_aux[i] = e_r1[i] = 0.f;
        if (!EXTRA) continue;
        if (a.addend) e_add[i] = a.addend[(size_t)row * a.ld_addend + ecol];     // block-uniform
        if (a.epi == EPI_TANHBWD) e_aux[i] = a.aux[(size_t)row * a.ld_aux + ecol];
        if (a.r1_s) e_r1[i] = a.r1_s[row] * a.r1_v[ecol];
    }

    const int c_lo = (wave * a.sg.total) / SMALL_WAVES;
    const int c_hi = ((wave + 1) * a.sg.total) / SMALL_WAVES;
    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c_hi > c_lo) {
        Frags<MT, CPW> f;
        if (APRO)
            upfront_load_apro<MT, CPW>(f, a.sg, c_lo, c_hi, n, mrow, kk, a.apro_part, a.apro_stride);
        else
            upfront_load<MT, CPW>(f, a.sg, c_lo, c_hi, n, mrow, kk);
        upfront_mma<MT, CPW>(f, c_hi - c_lo, acc);
    }
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) s_part[wave][t][(kk * 4 + r) * 16 + li] = acc[t][r];
    __syncthreads();

#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const int e = threadIdx.x + i * SMALL_WAVES * 64;
        if (e >= MT * 256) break;
        const int t = e >> 8, rc = e & 255;
        const int row = m0 + 16 * t + (rc >> 4), col = n0 + (rc & 15);
        if (row >= a.M || col >= a.N) continue;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < SMALL_WAVES; ++w) v += s_part[w][t][rc];
        v += e_bias;                                   // (col == ecol: 512 % 16 == 0)
        if (EXTRA) v += e_add[i] + e_r1[i];
        if (a.epi == EPI_TANH) v = tanhf(v);
        if (EXTRA && a.epi == EPI_TANHBWD) v *= 1.f - e_aux[i] * e_aux[i];
        if (a.epi == EPI_MUL) {
            if (a.y_pre) a.y_pre[(size_t)row * a.ldy_pre + col] = v;
            v *= e_mul;
        }
        float* o = a.y + (size_t)row * a.ldy + col;
        const float out = a.accumulate ? *o + v : v;
        *o = out;
        if (PW) lstm_pw_bwd_elem(*pw, row, col, out);
    }
}

// Host side: the launch plan of a small product (null plan => not a "small" shape).
struct SmallPlan {
    SmallArgs args;
    int mt, cpw;          // template parameters of the body
    int gx, gy;           // grid
};
bool linear_small_plan(const Seg* segs, int nseg, int M, int N, const LinearOut& out, SmallPlan* plan);
int launch_small_plan_x(const SmallPlan& p, hipStream_t st);      // launch a plan on its own
// ... with the LSTM pointwise backward as its epilogue (N == pw.H columns, M == pw.B rows); SF_ERR_UNSUPPORTED for
// shapes without that instantiation
int launch_small_plan_pw(const SmallPlan& p, const LstmPwBwd& pw, hipStream_t st);

}  // namespace sf
