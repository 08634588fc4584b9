// The three soft attentions of the speaker/follower path, forward and backward.
// One workgroup per sample, row set resident in registers (see sf_rows.h).
#include "sf_kernels.h"
#include "sf_rows.h"
#include "sf_glue.h"

namespace sf {

namespace {

// =================================================================================================
// Visual attention (model.py:310-326), V rows of F floats; CPL = 9 covers F <= 2304.
// MODE 0 (forward):  w_v = softmax_v(x_v . vec)             out = sum_v w_v x_v   (vec = q)
// MODE 1 (backward): d_v = x_v . vec, w_v = alpha_v (d_v - sum_u alpha_u d_u)      (vec = dout)
//                    out = dq = sum_v w_v x_v
// =================================================================================================
constexpr int VIS_CPL = 9, VIS_RPW = 3, VIS_NW = 12, VIS_SLOTS = 6;

struct VisArgs {
    PanoSrc src;
    const float* vec;      // [B, ldvec]
    int ldvec;
    float* alpha;          // fwd: out [B,V]; bwd: in
    float* out;            // [B, ldo]
    int ldo;
    Dropout drop;          // fwd: applied to out; bwd: applied to vec (same mask)
    int drop_col0;
};

template <int MODE>
__global__ __launch_bounds__(VIS_NW * 64) void visual_attn_kernel(VisArgs a) {
    __shared__ float4 slots[VIS_SLOTS][VIS_CPL * 64];
    __shared__ float s_score[64];
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int V = a.src.V;
    const int n4 = (a.src.IMG + a.src.LOC) >> 2;

    float4 x[VIS_RPW][VIS_CPL];
#pragma unroll
    for (int r = 0; r < VIS_RPW; ++r) {
        const int v = wave * VIS_RPW + r;
#pragma unroll
        for (int i = 0; i < VIS_CPL; ++i) {
            const int c = lane + 64 * i;
            x[r][i] = (v < V && c < n4) ? pano_chunk(a.src, b, v, c) : f4zero();
        }
    }

    const uint32_t rkey = dropout_row_key(a.drop.seed, a.drop.stream, (uint32_t)(a.drop.row0 + b));
    float dot[VIS_RPW];
#pragma unroll
    for (int r = 0; r < VIS_RPW; ++r) dot[r] = 0.f;
#pragma unroll
    for (int i = 0; i < VIS_CPL; ++i) {
        const int c = lane + 64 * i;
        float4 q = f4zero();
        if (c < n4) {
            q = reinterpret_cast<const float4*>(a.vec + (size_t)b * a.ldvec)[c];
            if (MODE == 1 && a.drop.on()) {
                const uint32_t col = (uint32_t)(a.drop_col0 + 4 * c);
                q.x = dropout_keep(rkey, col + 0, a.drop.thresh) ? q.x * a.drop.scale : 0.f;
                q.y = dropout_keep(rkey, col + 1, a.drop.thresh) ? q.y * a.drop.scale : 0.f;
                q.z = dropout_keep(rkey, col + 2, a.drop.thresh) ? q.z * a.drop.scale : 0.f;
                q.w = dropout_keep(rkey, col + 3, a.drop.thresh) ? q.w * a.drop.scale : 0.f;
            }
        }
#pragma unroll
        for (int r = 0; r < VIS_RPW; ++r) dot[r] += dot4(x[r][i], q);
    }
#pragma unroll
    for (int r = 0; r < VIS_RPW; ++r) {
        const float s = wave_sum(dot[r]);
        const int v = wave * VIS_RPW + r;
        if (lane == 0 && v < V) s_score[v] = s;
    }
    __syncthreads();

    // every wave redoes the V-wide softmax with lane v holding score v (V <= 64)
    const float s = lane < V ? s_score[lane] : -INFINITY;
    float w;
    if (MODE == 0) {
        const float m = wave_max(s);
        const float e = lane < V ? expf(s - m) : 0.f;
        w = e / wave_sum(e);
        if (wave == 0 && lane < V) a.alpha[(size_t)b * V + lane] = w;
    } else {
        const float al = lane < V ? a.alpha[(size_t)b * V + lane] : 0.f;
        const float d = lane < V ? s : 0.f;
        w = al * (d - wave_sum(al * d));
    }

    float4 p[VIS_CPL];
#pragma unroll
    for (int i = 0; i < VIS_CPL; ++i) p[i] = f4zero();
#pragma unroll
    for (int r = 0; r < VIS_RPW; ++r) {
        const int v = wave * VIS_RPW + r;
        const float wr = __shfl(w, v < V ? v : 0, WAVE);
        if (v < V) {
#pragma unroll
            for (int i = 0; i < VIS_CPL; ++i) f4fma(p[i], wr, x[r][i]);
        }
    }

    float* orow = a.out + (size_t)b * a.ldo;
    const Dropout dr = a.drop;
    const int col0 = a.drop_col0;
    block_row_sum<VIS_CPL, VIS_NW, VIS_SLOTS>(p, slots, n4, [&](int c, float4 t) {
        if (MODE == 0 && dr.on()) {
            const uint32_t col = (uint32_t)(col0 + 4 * c);
            t.x = dropout_keep(rkey, col + 0, dr.thresh) ? t.x * dr.scale : 0.f;
            t.y = dropout_keep(rkey, col + 1, dr.thresh) ? t.y * dr.scale : 0.f;
            t.z = dropout_keep(rkey, col + 2, dr.thresh) ? t.z * dr.scale : 0.f;
            t.w = dropout_keep(rkey, col + 3, dr.thresh) ? t.w * dr.scale : 0.f;
        }
        reinterpret_cast<float4*>(orow)[c] = t;
    });
}

// =================================================================================================
// Text / path-context attention core (model.py:129-139): L rows of H floats, CPL = 2 (H <= 512).
// forward : s_l = ctx_l . t (masked -> -inf), alpha = softmax, wc = sum alpha_l ctx_l
// backward: d_l = ctx_l . dwc, ds_l = alpha_l (d_l - sum alpha d), dt = sum ds_l ctx_l,
//           dctx_l += alpha_l dwc + ds_l t
// =================================================================================================
constexpr int TXT_CPL = 2, TXT_NW = 16, TXT_SLOTS = 8;

struct TxtArgs {
    const float* ctx;      // [B, L, H]
    const uint8_t* mask;   // [B, L] or null
    int L, H;
    const float* vec;      // fwd: t [B, ldvec]; bwd: dwc [B, ldvec]
    int ldvec;
    const float* vec2;     // bwd: t [B, ldvec2]
    int ldvec2;
    float* alpha;          // [B, L]
    float* out;            // fwd: wc [B, ldo]; bwd: dt [B, ldo]
    int ldo;
    float* dctx;           // bwd, accumulated; may be null
    const int32_t* ctx_row;  // fwd: sample b attends over ctx / mask row ctx_row[b] (null = b)
};

template <int RPW, int MODE>
__global__ __launch_bounds__(TXT_NW * 64) void text_attn_kernel(TxtArgs a) {
    __shared__ float4 slots[TXT_SLOTS][TXT_CPL * 64];
    __shared__ float s_score[TXT_NW * RPW];
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int L = a.L, n4 = a.H >> 2;
    const int bc = a.ctx_row ? a.ctx_row[b] : b;
    const float4* ctx = reinterpret_cast<const float4*>(a.ctx) + (size_t)bc * L * n4;

    float4 x[RPW][TXT_CPL];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int l = wave * RPW + r;
#pragma unroll
        for (int i = 0; i < TXT_CPL; ++i) {
            const int c = lane + 64 * i;
            x[r][i] = (l < L && c < n4) ? ctx[(size_t)l * n4 + c] : f4zero();
        }
    }
    float4 v1[TXT_CPL], v2[TXT_CPL];
#pragma unroll
    for (int i = 0; i < TXT_CPL; ++i) {
        const int c = lane + 64 * i;
        v1[i] = c < n4 ? reinterpret_cast<const float4*>(a.vec + (size_t)b * a.ldvec)[c] : f4zero();
        v2[i] = (MODE == 1 && c < n4)
                    ? reinterpret_cast<const float4*>(a.vec2 + (size_t)b * a.ldvec2)[c]
                    : f4zero();
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        float d = 0.f;
#pragma unroll
        for (int i = 0; i < TXT_CPL; ++i) d += dot4(x[r][i], v1[i]);
        d = wave_sum(d);
        const int l = wave * RPW + r;
        if (lane == 0 && l < L) {
            if (MODE == 0 && a.mask && a.mask[(size_t)bc * L + l]) d = -INFINITY;
            s_score[l] = d;
        }
    }
    __syncthreads();

    // L <= 128: lane holds entries lane and lane + 64
    const int l0 = lane, l1 = lane + 64;
    float w0, w1;
    if (MODE == 0) {
        const float s0 = l0 < L ? s_score[l0] : -INFINITY;
        const float s1 = l1 < L ? s_score[l1] : -INFINITY;
        const float m = wave_max(fmaxf(s0, s1));
        const float e0 = l0 < L ? expf(s0 - m) : 0.f;
        const float e1 = l1 < L ? expf(s1 - m) : 0.f;
        const float inv = 1.0f / wave_sum(e0 + e1);
        w0 = e0 * inv;
        w1 = e1 * inv;
        if (wave == 0) {
            if (l0 < L) a.alpha[(size_t)b * L + l0] = w0;
            if (l1 < L) a.alpha[(size_t)b * L + l1] = w1;
        }
    } else {
        const float a0 = l0 < L ? a.alpha[(size_t)b * L + l0] : 0.f;
        const float a1 = l1 < L ? a.alpha[(size_t)b * L + l1] : 0.f;
        const float d0 = l0 < L ? s_score[l0] : 0.f;
        const float d1 = l1 < L ? s_score[l1] : 0.f;
        const float tot = wave_sum(a0 * d0 + a1 * d1);
        w0 = a0 * (d0 - tot);
        w1 = a1 * (d1 - tot);
    }

    float4 p[TXT_CPL];
#pragma unroll
    for (int i = 0; i < TXT_CPL; ++i) p[i] = f4zero();
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int l = wave * RPW + r;
        const int src = (l < L ? l : 0) & 63;
        const float lo = __shfl(w0, src, WAVE), hi = __shfl(w1, src, WAVE);
        const float wr = (l < 64) ? lo : hi;
        if (l < L) {
#pragma unroll
            for (int i = 0; i < TXT_CPL; ++i) f4fma(p[i], wr, x[r][i]);
            if (MODE == 1 && a.dctx) {
                const float al = a.alpha[(size_t)b * L + l];
                float4* drow = reinterpret_cast<float4*>(a.dctx) + ((size_t)b * L + l) * n4;
#pragma unroll
                for (int i = 0; i < TXT_CPL; ++i) {
                    const int c = lane + 64 * i;
                    if (c < n4) {
                        float4 g = drow[c];
                        f4fma(g, al, v1[i]);   // alpha_l * dwc
                        f4fma(g, wr, v2[i]);   // ds_l * t
                        drow[c] = g;
                    }
                }
            }
        }
    }
    float* orow = a.out + (size_t)b * a.ldo;
    block_row_sum<TXT_CPL, TXT_NW, TXT_SLOTS>(p, slots, n4, [&](int c, float4 t) {
        reinterpret_cast<float4*>(orow)[c] = t;
    });
}

// =================================================================================================
// Candidate scoring (model.py:342-352 after folding): one wave per candidate, A <= 16.
// forward : logit[b,a] = u_a . r[b] + (wt[b] . b_a + b_out)
// backward: dr[b] = sum_a dlogit[b,a] u_a ;  dc[b] = sum_a dlogit[b,a]
// =================================================================================================
constexpr int SC_CPL = 9, SC_NW = 16, SC_SLOTS = 4;

struct ScoreArgs {
    CandSrc src;
    int ldr;               // row stride of r
    const float* cst;      // optional per-row constant (stride ldr); null -> wt.b_a + b_out
    const float* r;        // fwd [B,ldr]
    const float* wt;       // fwd [B,D]
    const float* b_a;      // [D]
    const float* b_out;    // [1]
    int D;
    float* logit;          // fwd out [B,A]; bwd: dlogit in
    float* dr;             // bwd out [B,F]
    float* dc;             // bwd out [B]
};

__global__ __launch_bounds__(SC_NW * 64) void score_fwd_kernel(ScoreArgs a) {
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int A = a.src.A;
    if (wave >= A) return;
    const int n4 = (a.src.IMG + a.src.LOC) >> 2;
    float d = 0.f;
#pragma unroll
    for (int i = 0; i < SC_CPL; ++i) {
        const int c = lane + 64 * i;
        if (c < n4)
            d += dot4(cand_chunk(a.src, b, wave, c),
                      reinterpret_cast<const float4*>(a.r + (size_t)b * a.ldr)[c]);
    }
    float cst = 0.f;
    if (!a.cst)
        for (int k = lane; k < a.D; k += 64) cst += a.wt[(size_t)b * a.D + k] * a.b_a[k];
    d = wave_sum(d);
    cst = a.cst ? a.cst[(size_t)b * a.ldr] : wave_sum(cst) + a.b_out[0];
    if (lane == 0) a.logit[(size_t)b * A + wave] = d + cst;
}

// Scoring + per-step glue fused (one dependent stage instead of two): wave a keeps candidate a's row
// in registers, the 16 logits meet in LDS, wave 0 masks / soft-maxes / picks the action, and the wave
// that owns the chosen row writes dropout(u_next) straight into the next step's LSTM input.
__global__ __launch_bounds__(SC_NW * 64) void score_glue_kernel(ScoreArgs a, FGlue g) {
    __shared__ float s_logit[64];
    __shared__ int s_at;
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int A = a.src.A;
    const int n4 = (a.src.IMG + a.src.LOC) >> 2;
    float4 x[SC_CPL];
    float d = 0.f;
#pragma unroll
    for (int i = 0; i < SC_CPL; ++i) {
        const int c = lane + 64 * i;
        x[i] = (wave < A && c < n4) ? cand_chunk(a.src, b, wave, c) : f4zero();
        if (c < n4) d += dot4(x[i], reinterpret_cast<const float4*>(a.r + (size_t)b * a.ldr)[c]);
    }
    float cst = 0.f;
    if (wave < A && !a.cst)
        for (int k = lane; k < a.D; k += 64) cst += a.wt[(size_t)b * a.D + k] * a.b_a[k];
    d = wave_sum(d);
    cst = a.cst ? a.cst[(size_t)b * a.ldr] : wave_sum(cst) + a.b_out[0];
    if (lane == 0 && wave < A) s_logit[wave] = d + cst;
    __syncthreads();
    if (wave == 0) {
        const int at = follower_glue_row(g, b, lane < A ? s_logit[lane] : 0.f);
        if (lane == 0) s_at = at;
    }
    __syncthreads();
    if (g.u_next && wave == s_at) {
#pragma unroll
        for (int i = 0; i < SC_CPL; ++i) {
            const int c = lane + 64 * i;
            if (c < n4) store_u_next(g, b, c, x[i]);
        }
    }
}

__global__ __launch_bounds__(SC_NW * 64) void score_bwd_kernel(ScoreArgs a) {
    __shared__ float4 slots[SC_SLOTS][SC_CPL * 64];
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int A = a.src.A;
    const int n4 = (a.src.IMG + a.src.LOC) >> 2;
    const float w = wave < A ? a.logit[(size_t)b * A + wave] : 0.f;
    float4 p[SC_CPL];
#pragma unroll
    for (int i = 0; i < SC_CPL; ++i) {
        const int c = lane + 64 * i;
        p[i] = f4zero();
        if (wave < A && c < n4 && w != 0.f) f4fma(p[i], w, cand_chunk(a.src, b, wave, c));
    }
    if (wave == 0) {
        const float dl = lane < A ? a.logit[(size_t)b * A + lane] : 0.f;
        const float tot = wave_sum(dl);
        if (lane == 0) a.dc[b] = tot;
    }
    float* orow = a.dr + (size_t)b * (n4 << 2);
    block_row_sum<SC_CPL, SC_NW, SC_SLOTS>(p, slots, n4, [&](int c, float4 t) {
        reinterpret_cast<float4*>(orow)[c] = t;
    });
}

}  // namespace

int visual_attn(int mode, const PanoSrc& src, int B, const float* vec, int ldvec, float* alpha,
                float* out, int ldo, const Dropout& drop, int drop_col0, hipStream_t st) {
    const int F = src.IMG + src.LOC;
    if (src.V > VIS_RPW * VIS_NW || src.V > 64 || F > VIS_CPL * 256 || (F & 3) ||
        (!src.dense && ((src.IMG & 3) || (src.LOC & 3))) || (ldvec & 3) || (ldo & 3))
        return SF_ERR_UNSUPPORTED;
    VisArgs a{src, vec, ldvec, alpha, out, ldo, drop, drop_col0};
    if (mode == 0)
        hipLaunchKernelGGL(visual_attn_kernel<0>, dim3(B), dim3(VIS_NW * 64), 0, st, a);
    else
        hipLaunchKernelGGL(visual_attn_kernel<1>, dim3(B), dim3(VIS_NW * 64), 0, st, a);
    return launch_status();
}

template <int MODE>
static int text_attn_launch(const TxtArgs& a, int B, hipStream_t st) {
    if (a.L <= TXT_NW)
        hipLaunchKernelGGL((text_attn_kernel<1, MODE>), dim3(B), dim3(TXT_NW * 64), 0, st, a);
    else if (a.L <= TXT_NW * 5)
        hipLaunchKernelGGL((text_attn_kernel<5, MODE>), dim3(B), dim3(TXT_NW * 64), 0, st, a);
    else if (a.L <= TXT_NW * 8)
        hipLaunchKernelGGL((text_attn_kernel<8, MODE>), dim3(B), dim3(TXT_NW * 64), 0, st, a);
    else
        return SF_ERR_UNSUPPORTED;
    return launch_status();
}

int text_attn_fwd(const float* ctx, const uint8_t* mask, int B, int L, int H, const float* t,
                  int ldt, float* alpha, float* wc, int ldwc, hipStream_t st,
                  const int32_t* ctx_row) {
    if (H > TXT_CPL * 256 || (H & 3) || (ldt & 3) || (ldwc & 3) || L < 1) return SF_ERR_UNSUPPORTED;
    TxtArgs a{ctx, mask, L, H, t, ldt, nullptr, 0, alpha, wc, ldwc, nullptr, ctx_row};
    return text_attn_launch<0>(a, B, st);
}

int text_attn_bwd(const float* ctx, int B, int L, int H, const float* dwc, int lddwc,
                  const float* t, int ldt, const float* alpha, float* dt, int lddt, float* dctx,
                  hipStream_t st) {
    if (H > TXT_CPL * 256 || (H & 3) || (ldt & 3) || (lddwc & 3) || (lddt & 3) || L < 1)
        return SF_ERR_UNSUPPORTED;
    TxtArgs a{ctx, nullptr, L, H, dwc, lddwc, t, ldt, const_cast<float*>(alpha), dt, lddt, dctx, nullptr};
    return text_attn_launch<1>(a, B, st);
}

int score_fwd(const CandSrc& src, int B, int D, const float* r, const float* wt, const float* b_a,
              const float* b_out, float* logit, hipStream_t st, int ldr, const float* cst) {
    const int F = src.IMG + src.LOC;
    if (ldr <= 0) ldr = F;
    if (src.A > SC_NW || src.A < 1 || F > SC_CPL * 256 || (F & 3) ||
        (!src.dense && ((src.IMG & 3) || (src.LOC & 15))))
        return SF_ERR_UNSUPPORTED;
    ScoreArgs a{src, ldr, cst, r, wt, b_a, b_out, D, logit, nullptr, nullptr};
    hipLaunchKernelGGL(score_fwd_kernel, dim3(B), dim3(SC_NW * 64), 0, st, a);
    return launch_status();
}

int score_glue_fwd(const CandSrc& src, int B, int D, const float* r, const float* wt,
                   const float* b_a, const float* b_out, const FGlue& g, hipStream_t st, int ldr,
                   const float* cst) {
    const int F = src.IMG + src.LOC;
    if (ldr <= 0) ldr = F;
    if (src.A > SC_NW || src.A < 1 || F > SC_CPL * 256 || (F & 3) ||
        (!src.dense && ((src.IMG & 3) || (src.LOC & 15))))
        return SF_ERR_UNSUPPORTED;
    ScoreArgs a{src, ldr, cst, r, wt, b_a, b_out, D, g.logit, nullptr, nullptr};
    hipLaunchKernelGGL(score_glue_kernel, dim3(B), dim3(SC_NW * 64), 0, st, a, g);
    return launch_status();
}

int score_bwd(const CandSrc& src, int B, const float* dlogit, float* dr, float* dc,
              hipStream_t st) {
    const int F = src.IMG + src.LOC;
    if (src.A > SC_NW || src.A < 1 || F > SC_CPL * 256 || (F & 3) ||
        (!src.dense && ((src.IMG & 3) || (src.LOC & 15))))
        return SF_ERR_UNSUPPORTED;
    ScoreArgs a{src, F, nullptr, nullptr, nullptr, nullptr, nullptr, 0, const_cast<float*>(dlogit), dr, dc};
    hipLaunchKernelGGL(score_bwd_kernel, dim3(B), dim3(SC_NW * 64), 0, st, a);
    return launch_status();
}

}  // namespace sf
