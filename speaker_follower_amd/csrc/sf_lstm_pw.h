// The LSTM cell's pointwise backward (model.py:393 / LSTMCell): its parameter block and the arithmetic of ONE element,
// shared by the stand-alone kernel (sf_pointwise.hip: lstm_pw_bwd_kernel) and the epilogue of the small product that
// completes dh1 in the decoder's backward through time (sf_gemm.hip: gemm_nt_small_pw_kernel).
#pragma once
#include "sf_common.h"

namespace sf {

struct LstmPwBwd {
    const float* gates; const float* c0; const float* c1;
    const float* dh1; const float* dh1_b;   // two optional contributions to dh1 (either may be null)
    const float* dc1;
    int B, H;
    float* dgates;               // [B,4H] pre-activation gate gradients
    float* dc0;                  // [B,H]
    const int* lengths; int t;   // encoder: dead rows pass dh1/dc1 through untouched, dgates = 0
    float* dh0_pass;             // encoder: for dead rows dh0 = dh1 (written here), live rows 0
                                 // (may alias dh1: element-wise in place)
    const float* dctx; int T;    // encoder: + dropout-masked dctx[b, t, :] (row stride T*H), or null
    Dropout ctx_drop;
    Dropout dh1b_drop;           // mask applied to dh1_b on load (column = j): the backward of the
                                 // dropout between h1 and the text attention (model.py:394)
};

// element (row b, hidden unit j); `dh` = the dh1 term (a.dh1 is NOT read here)
__device__ __forceinline__ void lstm_pw_bwd_elem(const LstmPwBwd& a, int b, int j, float dh) {
    const int H = a.H;
    const int idx = b * H + j;
    // all operands first (block-uniform branches around optional ones), then the arithmetic
    const float* gp = a.gates + (size_t)b * 4 * H + j;
    const float ig = gp[0], fg = gp[H], gg = gp[2 * H], og = gp[3 * H];
    const float c1 = a.c1[idx], c0 = a.c0[idx];
    float dc = 0.f;
    if (a.dh1_b) {
        float v = a.dh1_b[idx];
        if (a.dh1b_drop.on()) {
            const uint32_t rk = drop_key(a.dh1b_drop, (uint32_t)(a.dh1b_drop.row0 + b));
            v = dropout_keep(rk, (uint32_t)j, a.dh1b_drop.thresh) ? v * a.dh1b_drop.scale : 0.f;
        }
        dh += v;
    }
    if (a.dc1) dc = a.dc1[idx];
    bool dead = false;
    if (a.lengths) dead = a.t >= a.lengths[b];
    if (a.dctx) {                                     // encoder: gradient of ctx[b, t, :]
        float v = a.dctx[((size_t)b * a.T + a.t) * H + j];
        if (a.ctx_drop.on()) {
            const uint32_t rk = drop_key(a.ctx_drop, (uint32_t)(a.ctx_drop.row0 + b));
            v = dropout_keep(rk, (uint32_t)(a.t * H + j), a.ctx_drop.thresh)
                    ? v * a.ctx_drop.scale : 0.f;
        }
        dh += v;
    }
    float* dg = a.dgates + (size_t)b * 4 * H + j;
    const float tc = tanhf(c1);
    const float dout = dh * tc;
    const float dcl = dc + dh * og * (1.f - tc * tc);
    // packed sequence: a step that did not happen passes dh / dc through, dgates = 0
    dg[0] = dead ? 0.f : dcl * gg * ig * (1.f - ig);
    dg[H] = dead ? 0.f : dcl * c0 * fg * (1.f - fg);
    dg[2 * H] = dead ? 0.f : dcl * ig * (1.f - gg * gg);
    dg[3 * H] = dead ? 0.f : dout * og * (1.f - og);
    a.dc0[idx] = dead ? dc : dcl * fg;
    if (a.dh0_pass) a.dh0_pass[idx] = dead ? dh : 0.f;
}

}  // namespace sf
