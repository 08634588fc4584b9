// Host interface of experimental/sf_mega.hip (libsf_experimental.so only).
#pragma once
#include "sf_kernels.h"

namespace sf {

// ---- sf_mega.hip: the follower's decode loop as one persistent launch (inference) ----------------
struct MegaHost {
    const float *w_ih, *w_hh, *b_ih, *b_hh;             // LSTMCell [4H,2F], [4H,H], [4H] x2
    const float *w_in, *w_out;                          // text attention [H,H], [H,2H]
    const float *m_v, *c_v, *m_a, *c_a;                 // sf_decoder_fold
    const float *h_init, *c_init;                       // [B,H]
    const float* feat0; int ld_feat0;                   // attended feature of step 0 (per-stage head), row stride
    const float* ctx; const uint8_t* mask; int L;       // [B,L,H], [B,L]
    PanoSrc X;                                          // step 0 of the stacked [S][B] index arrays
    CandSrc U;
    const int64_t* target;                              // [S,B]
    int feedback;
    uint32_t sample_seed, sample_stream0;
    int row0;
    uint8_t* ended;                                     // [B] in / out
    float* logit; int64_t* a_t; int64_t* target_used; float* score; float* ce_term; float* live;   // [S][B][..]
    float* h1_tape; float* c1_tape;                     // [S,B,H]
    float *dbg_t_text, *dbg_cat2, *dbg_h_tilde, *dbg_q, *dbg_xin;   // optional copies for the tests
    int B, S;
    unsigned* xchg;                                     // mega_xchg_dwords() dwords of scratch
    unsigned* done;                                     // one zero-initialised ticket word
};
size_t mega_xchg_dwords();
bool mega_supported(int B, int H, int L, int A, const PanoSrc& X, const CandSrc& U);
int mega_decode(const MegaHost& h, hipStream_t st);

}  // namespace sf
