// EXPERIMENT, not part of libsf_hip.so: the decode loop of an inference rollout as ONE persistent launch
// (sf_mega.hip).  Correct (tests/test_gpu_mega.py) and slower than the per-stage path on MI355X (94 vs
// 80 us per decode step), so the product library and its ABI do not carry it.
// `python -m speaker_follower_amd.build --experimental` links this translation unit -- which textually
// includes the product's entry points for their file-local helpers (step views, workspace arena, head of
// step 0) -- with sf_mega.o and the product's kernel objects into libsf_experimental.so.
// Declaration: include/sf_hip_experimental.h.
#include "../sf_api.hip"
#include "sf_mega.h"
#include "../../../include/sf_hip_experimental.h"

extern "C" int sf_follower_decode_persistent(const sf_decoder_w* w, const sf_follower_episode* e, int debug_tapes,
                                             void* ws, size_t ws_bytes, sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(w && e && e->S > 0 && e->B > 0 && e->h_init && e->c_init && e->ctx && e->ctx_mask && e->tape.xin && e->tape.h1 &&
                 e->tape.c1 && e->tape.logit && glue_ok(&e->U, &e->glue));
    const PanoSrc xs = pano(&e->X);
    const CandSrc us = cands(&e->U);
    if (!w->fold || e->drop.p > 0.f || e->glue.is_valid || !mega_supported(e->B, e->H, e->L, e->A, xs, us))
        return SF_ERR_UNSUPPORTED;
    const int F = xs.IMG + xs.LOC;
    // head(0): the visual attention of step 0 on h_init, by the per-stage kernels, into tape.xin[0][:, F:2F]
    StepView v0 = step_view(e, 0);
    TRY(decoder_head_i(w, &v0.X, e->B, e->H, e->D, e->h_init, &v0.tp, nullptr, e->step0, ws, ws_bytes, stream));
    Arena ar = arena(ws, ws_bytes);
    float* xchg = ar.take(mega_xchg_dwords());
    NEED(xchg && ar.tickets());
    MegaHost h{};
    h.w_ih = w->lstm.w_ih; h.w_hh = w->lstm.w_hh; h.b_ih = w->lstm.b_ih; h.b_hh = w->lstm.b_hh;
    h.w_in = w->text.w_in; h.w_out = w->text.w_out;
    h.m_v = w->fold->m_v; h.c_v = w->fold->c_v; h.m_a = w->fold->m_a; h.c_a = w->fold->c_a;
    h.h_init = e->h_init; h.c_init = e->c_init; h.feat0 = e->tape.xin + F; h.ld_feat0 = 2 * F;
    h.ctx = e->ctx; h.mask = e->ctx_mask; h.L = e->L;
    h.X = xs; h.U = us;
    h.target = e->glue.target; h.feedback = e->glue.feedback; h.sample_seed = e->glue.sample_seed;
    h.sample_stream0 = e->step0; h.row0 = e->glue.row0; h.ended = e->glue.ended;
    h.logit = e->tape.logit; h.a_t = e->glue.a_t; h.target_used = e->glue.target_used; h.score = e->glue.score;
    h.ce_term = e->glue.ce_term; h.live = e->glue.live; h.h1_tape = e->tape.h1; h.c1_tape = e->tape.c1;
    if (debug_tapes) {
        h.dbg_t_text = e->tape.t_text; h.dbg_cat2 = e->tape.cat2; h.dbg_h_tilde = e->tape.h_tilde; h.dbg_q = e->tape.q;
        h.dbg_xin = e->tape.xin;
    }
    h.B = e->B; h.S = e->S;
    h.xchg = reinterpret_cast<unsigned*>(xchg);
    h.done = ar.tickets() + PERSIST_TICKET;
    return mega_decode(h, S(stream));
}

