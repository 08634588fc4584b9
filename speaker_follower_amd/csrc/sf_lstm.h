// LSTM cell update shared by the split-K pointwise kernel and the fused recurrent step kernel.
#pragma once
#include "sf_common.h"

namespace sf {

// LSTM gates -> state.  live (lengths != null) = per-row "t < length" flag of the
// packed-sequence encoder: dead rows copy h0/c0 through and write zeros to ctx_out.
struct LstmPwFwd {
    const float* slabs; int ks;  // split-K partial slabs [ks][B][4H] (pointwise kernel only)
    const float* xg;             // [B,4H] hoisted input product or null
    const float* b_ih; const float* b_hh;
    const float* c0; const float* h0;
    int B, H;
    float* gates;                // [B,4H] activated (may be null)
    float* h1; float* c1;        // [B,H]
    float* h1_drop; int ld_h1_drop; Dropout drop;   // optional dropped copy
    const int* lengths; int t;   // encoder only (lengths null otherwise)
    float* ctx_out; int ld_ctx;  // encoder only: ctx[b, t, :], row stride T*H
    Dropout ctx_drop;
};

struct LstmStepArgs {            // fused recurrent step (sf_gemm.hip: lstm_step_fused_kernel)
    const float* h0; const float* w_hh;             // [B,H], [4H,H]
    const float* x; int ldx; const float* w_ih; int I;   // optional input segment (null if xg)
    const float* xg;                                // [B,4H] hoisted x*W_ih^T or null
    const int64_t* xg_index;                        // optional: row b reads xg[xg_index[b * xg_index_ld]]
    int xg_index_ld;                                // (word lookup; 0 is read as 1)
    const float* b_ih; const float* b_hh;
    int B, H;
    LstmPwFwd pw;                                   // outputs / state (slabs, xg, biases unused)
};

// Encoder-only operands of the cell update (row liveness at step t, previous h), fetched by the
// caller up front with its other tail operands so that no load sits behind the gate reduction.
struct LstmLive {
    bool live;
    float h0;
};
__device__ __forceinline__ LstmLive lstm_live_load(const LstmPwFwd& a, int b, int j) {
    LstmLive l{true, 0.f};
    if (a.lengths) {                                             // block-uniform
        l.live = a.t < a.lengths[b];
        l.h0 = a.h0[b * a.H + j];
    }
    return l;
}

// g4 = pre-activation gates (i,f,g,o) of element (b, j), biases already added.
__device__ __forceinline__ void lstm_cell_update(const LstmPwFwd& a, int b, int j,
                                                 const float (&g4)[4], float c0,
                                                 const LstmLive& lv) {
    const int H = a.H;
    const int idx = b * H + j;
    const float ig = sigmoidf_(g4[0]), fg = sigmoidf_(g4[1]), gg = tanhf(g4[2]),
                og = sigmoidf_(g4[3]);
    float c1 = fg * c0 + ig * gg;
    float h1 = og * tanhf(c1);
    if (a.gates) {
        float* gp = a.gates + (size_t)b * 4 * H + j;
        gp[0] = ig; gp[H] = fg; gp[2 * H] = gg; gp[3 * H] = og;
    }
    if (a.lengths) {
        const bool live = lv.live;
        if (!live) { c1 = c0; h1 = lv.h0; }
        float cv = live ? h1 : 0.f;
        if (live && a.ctx_drop.on()) {
            const uint32_t rk = drop_key(a.ctx_drop, (uint32_t)(a.ctx_drop.row0 + b));
            cv = dropout_keep(rk, (uint32_t)(a.t * H + j), a.ctx_drop.thresh)
                     ? cv * a.ctx_drop.scale : 0.f;
        }
        a.ctx_out[(size_t)b * a.ld_ctx + (size_t)a.t * H + j] = cv;
    }
    a.c1[idx] = c1;
    a.h1[idx] = h1;
    if (a.h1_drop) {
        float hd = h1;
        if (a.drop.on()) {
            const uint32_t rk = drop_key(a.drop, (uint32_t)(a.drop.row0 + b));
            hd = dropout_keep(rk, (uint32_t)j, a.drop.thresh) ? hd * a.drop.scale : 0.f;
        }
        a.h1_drop[(size_t)b * a.ld_h1_drop + j] = hd;
    }
}

// Fused backward time step of the encoder LSTM (sf_gemm.hip: lstm_bwd_step_fused_kernel):
// dh_{t+1} = pass_{t+1} + dgates_{t+1} W_hh (skipped for the last step) and, in the same launch, the
// cell backward of step t on that tile -> dgates_t, dc_t, pass_t.
struct LstmBwdStepArgs {
    const float* dgates_next;    // [B,4H] pre-activation gate gradients of step t+1, or null (t = T-1)
    const float* w_hh_t;         // [H,4H] transposed recurrent weight (K-contiguous rows)
    const float* dh_in;          // [B,H] pass-through part of dh_{t+1} (dead rows) / the incoming dh_T
    const float* dc_in;          // [B,H] dc_{t+1}
    const float* gates;          // [B,4H] activated gates of step t
    const float* c0; const float* c1;   // [B,H] cell state before / after step t
    const float* dctx; int T; int t;    // dctx[b, t, :] (row stride T*H) or null
    Dropout ctx_drop;
    const int* lengths;
    int B, H;
    float* dgates;               // [B,4H] out
    float* dc_out;               // [B,H] out
    float* dh_out;               // [B,H] out: pass-through of dead rows (0 for live rows)
};
int lstm_bwd_step_fused(const LstmBwdStepArgs& p, hipStream_t st);

}  // namespace sf
