// extern "C" entry points of libsf_hip.so (see include/sf_hip.h) and the host-side sequencing of
// kernels for the composite operators.  Nothing here allocates or synchronises.
#include "sf_kernels.h"
#include "sf_gemm_small.h"
#include "sf_glue.h"

using namespace sf;

#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace sf {
thread_local hipError_t g_last_hip_error = hipSuccess;

// ---- in-process kernel timing (SF_LAUNCH, sf_common.h) ----
namespace {
struct ProfRec { const char* name; hipEvent_t e0, e1; };
struct Prof {
    std::atomic<bool> active{false};   // read by every SF_LAUNCH on any host thread, flipped by begin / end
    std::vector<ProfRec> recs;
    std::vector<hipEvent_t> pool;      // events are created once and reused by later sessions
    size_t used = 0;
};
Prof g_prof;                 // process-wide: torch runs backward() on its own host thread
std::mutex g_prof_mutex;
}  // namespace
bool prof_active() { return g_prof.active.load(std::memory_order_acquire); }
void prof_events(const char* name, hipEvent_t* e0, hipEvent_t* e1) {
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    Prof& p = g_prof;
    *e0 = *e1 = nullptr;
    if (!p.active.load(std::memory_order_relaxed)) return;     // sf_profile_end won the race: plain launch
    while (p.pool.size() < p.used + 2) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) { e = nullptr; }
        p.pool.push_back(e);
    }
    if (!p.pool[p.used] || !p.pool[p.used + 1]) return;        // event creation failed: untimed launch
    *e0 = p.pool[p.used++];
    *e1 = p.pool[p.used++];
    p.recs.push_back(ProfRec{name, *e0, *e1});
}
}  // namespace sf

namespace {

constexpr size_t WORKSPACE_BYTES = 64u << 20;
// the weight-gradient entry points take a LARGER scratch buffer when the caller has one (the grouped weight gradients keep
// ~135 MB of transposed operands + slabs); kept apart from the workspace of the decode steps: measured in round 5, a
// 256 MB workspace costs the headline rollout 2 % (1.02 vs 1.04 M agent-steps/s, same kernels, same box)
constexpr size_t WGRAD_WORKSPACE_BYTES = 256u << 20;

// Bump allocator over the caller's workspace.  Passed BY VALUE into helpers so that their
// temporaries are released on return; whatever is left is handed to the GEMMs for split-K slabs.
struct Arena {
    float* base;
    size_t cap, off;
    unsigned* tk = nullptr;   // the workspace's ticket region (set by arena(); sub-arenas carry it along)
    float* take(size_t n) {
        n = (n + 63) & ~(size_t)63;
        if (off + n > cap) return nullptr;
        float* p = base + off;
        off += n;
        return p;
    }
    float* rest() const { return base + off; }
    size_t rest_n() const { return cap - off; }
    // the last SYNC_WORDS dwords of the workspace: monotonic ticket counters (zero-initialised by
    // the caller once, see sf_workspace_bytes); never handed out by take()
    unsigned* tickets() const { return tk; }
};
constexpr size_t SYNC_WORDS = 2048;        // [0, 1024): per-sample tickets of the split attention kernels
constexpr size_t PERSIST_TICKET = 1100;   // [1100, 1140): arrival counter + placement record of the persistent launches
constexpr size_t TEXT_TICKET = 1200;      // [1200, 1712): per-sample tickets of the folded text attention (B <= 512)

inline Arena arena(void* ws, size_t bytes) {
    const size_t n = ws ? bytes / 4 : 0;
    const size_t usable = n > SYNC_WORDS ? n - SYNC_WORDS : 0;
    return Arena{(float*)ws, usable, 0, usable ? reinterpret_cast<unsigned*>((float*)ws + usable) : nullptr};
}
inline hipStream_t S(sf_stream s) { return (hipStream_t)s; }

inline PanoSrc pano(const sf_pano* p) {
    return PanoSrc{p->dense, p->table, p->loc_table, p->vp, p->view, p->V, p->IMG, p->LOC};
}
inline CandSrc cands(const sf_cands* c) {
    return CandSrc{c->dense, c->table, c->vp, c->cand_view, c->cand_sincos, c->a_num,
                   c->A, c->V, c->IMG, c->LOC};
}

inline FGlue make_glue(const CandSrc& src, int B, float* logit, const sf_follower_glue* g) {
    FGlue f{};
    f.src = src; f.B = B; f.logit = logit; f.is_valid = g->is_valid; f.target = g->target;
    f.feedback = g->feedback; f.ended = g->ended; f.a_t = g->a_t; f.target_used = g->target_used;
    f.score = g->score; f.u_next = g->u_next; f.ld_u = g->ld_u_next;
    f.u_drop = make_dropout(g->u_drop, g->u_drop_stream, 2);     // (numbered 2 * (step + 1))
    f.ce_term = g->ce_term; f.live = g->live; f.sample_seed = g->sample_seed;
    f.sample_stream = g->sample_stream; f.row0 = g->row0;
    f.sample_site = g->sample_site_dev ? g->sample_site_dev : site_zero();
    if (g->nav) {
        const sf_nav_io& n = *g->nav;
        f.nav = NavIO{n.nav, n.row, n.view, n.goal_hop, n.ld_hop, n.hop_base, n.row_next, n.vp_next, n.view_next,
                      n.a_num_next, n.cand_view_next, n.sincos_next, n.target_next, true};
    }
    return f;
}
inline bool glue_ok(const sf_cands* U, const sf_follower_glue* g) {
    return g->target && g->ended && g->a_t && g->target_used && g->score && g->ce_term && g->live &&
           g->feedback >= 0 && g->feedback <= 2 && (g->is_valid || U->a_num) &&
           (!g->u_next || g->ld_u_next % 4 == 0) &&
           (!g->nav || (g->nav->nav.A == U->A && g->nav->nav.a_num && g->nav->nav.next_row && g->nav->nav.cand_view &&
                        g->nav->nav.cand_sincos && g->nav->nav.feat_row && g->nav->row && g->nav->view &&
                        g->nav->row_next && g->nav->vp_next && g->nav->view_next && g->nav->a_num_next &&
                        g->nav->cand_view_next && g->nav->sincos_next &&
                        (!g->nav->target_next || (g->nav->goal_hop && g->nav->hop_base && g->nav->ld_hop > 0))));
}

#define TRY(expr)                     \
    do {                              \
        int _st = (expr);             \
        if (_st != SF_OK) return _st; \
    } while (0)
#define NEED(ptr) \
    if (!(ptr)) return SF_ERR_WORKSPACE

// b_ih and b_hh of an LSTM have the same gradient (the column sums of dgates): one pass, two outputs
int colsum_pair(const float* Y, int ldy, int M, int N, float* a, float* b, Arena ar, hipStream_t st) {
    float* first = a ? a : b;
    if (!first) return SF_OK;
    return colsum(Y, ldy, M, N, first, 1, st, (a && b) ? b : nullptr, ar.rest(), ar.rest_n());
}

int linear_plain(const float* x, int ldx, const float* w, int ldw, const float* b, int M, int N,
                 int K, Epi epi, float* y, int ldy, Arena ar, hipStream_t st) {
    Seg sg{x, ldx, w, ldw, K};
    LinearOut o{};
    o.y = y;
    o.ldy = ldy;
    o.bias = b;
    o.epi = epi;
    return linear_nt(&sg, 1, M, N, o, ar.rest(), ar.rest_n(), st);
}

// dx[M,K] (+)= dy[M,N] * W[N,K].  With W^T ([K,N]) at hand this is a K-contiguous NT product.
int data_grad(const float* dy, int lddy, const float* w, const float* w_t, int M, int N, int K,
              float* dx, int lddx, int accumulate, Arena ar, hipStream_t st) {
    if (w_t && N % 4 == 0) {
        Seg sg{dy, lddy, w_t, N, N};
        LinearOut o{};
        o.y = dx; o.ldy = lddx; o.epi = EPI_NONE; o.accumulate = accumulate;
        return linear_nt(&sg, 1, M, K, o, ar.rest(), ar.rest_n(), st);
    }
    return gemm_nn_ws(dy, lddy, w, K, M, K, N, dx, lddx, accumulate, ar.rest(), ar.rest_n(), st);
}

// ---- a2 LSTMCell --------------------------------------------------------------------------------
int lstm_fwd_i(const sf_lstm_w* w, int B, int I, int H, const float* x, int ldx, const float* h0,
               const float* c0, float* h1, float* c1, float* gates, float* h1_drop, int ld_h1_drop,
               const Dropout& drop, Arena ar, hipStream_t st) {
    LstmPwFwd p{};
    p.xg = nullptr; p.b_ih = w->b_ih; p.b_hh = w->b_hh;
    p.c0 = c0; p.h0 = h0; p.B = B; p.H = H; p.gates = gates; p.h1 = h1; p.c1 = c1;
    p.h1_drop = h1_drop; p.ld_h1_drop = ld_h1_drop; p.drop = drop; p.lengths = nullptr;
    if (I + H <= 1024 && H % 16 == 0) {
        // short reduction (speaker decoder LSTMCell(300 -> 512)): one fused launch
        LstmStepArgs f{};
        f.h0 = h0; f.w_hh = w->w_hh; f.x = x; f.ldx = ldx; f.w_ih = w->w_ih; f.I = I; f.xg = nullptr;
        f.b_ih = w->b_ih; f.b_hh = w->b_hh; f.B = B; f.H = H; f.pw = p;
        return lstm_step_fused(f, st);
    }
    Seg segs[2] = {{x, ldx, w->w_ih, I, I}, {h0, H, w->w_hh, H, H}};
    LinearOut o{};
    float* slabs = nullptr;
    int ks = 1;
    TRY(linear_nt(segs, 2, B, 4 * H, o, ar.rest(), ar.rest_n(), st, &slabs, &ks));
    p.slabs = slabs; p.ks = ks;
    return lstm_pointwise_fwd(p, st);
}

// the cell's pointwise backward of one decoder / speaker step (no packed-sequence or ctx terms)
static LstmPwBwd cell_pw_bwd(const float* gates, const float* c0, const float* c1, const float* dh1, const float* dh1_b,
                             const float* dc1, int B, int H, float* dgates, float* dc0, const Dropout* dh1b_drop) {
    LstmPwBwd p{};
    p.gates = gates; p.c0 = c0; p.c1 = c1; p.dh1 = dh1; p.dh1_b = dh1_b; p.dc1 = dc1;
    p.B = B; p.H = H; p.dgates = dgates; p.dc0 = dc0; p.lengths = nullptr; p.dh0_pass = nullptr;
    if (dh1b_drop) p.dh1b_drop = *dh1b_drop;
    return p;
}

static int g_fold_build_overlap = 1;        // sf_debug_fold_build_overlap (0: the fold products on the caller's stream, in front of step 0's attention)
static int g_fold_chain3 = 1;               // sf_debug_fold_chain3 (0: the four-launch folded chain even with chain_fold)
static int g_fold_merge_with_glue = 1;      // sf_debug_fold_merge_with_glue (0: partials + merge in ONE launch, phase 0)
static int g_bptt_part = 0;        // EXPERIMENT (sf_debug_bptt_part): 1 = issue the heads only, 2 = the tails only (no waits)
static int g_bptt_flags = 0;       // sf_debug_bptt_flags: per-step device flags between the two chains of the backward instead of events
                                   // (measured equal; off by default)
static int g_bptt_lookahead = -1;  // sf_debug_bptt_lookahead (< 0: every head first)
static int g_fuse_cell_bwd = 0;    // sf_debug_fused_cell_backward: the cell's pointwise backward as the epilogue of the product
                                   // that completes dh1 instead of its own launch.  Bit-identical; measured 4.81 vs 4.79 ms per
                                   // iteration (one launch fewer per step, no gain): off by default
static int g_slab_consumers = 1;     // sf_debug_slab_consumers: consumers add up K-split slabs themselves (0: a reduce launch in between)

int lstm_bwd_i(const sf_lstm_w* w, const sf_lstm_g* g, int B, int I, int H, const float* x, int ldx,
               const float* h0, const float* c0, const float* c1, const float* gates,
               const float* dh1, const float* dh1_b, const float* dc1, float* dx, int lddx,
               float* dh0, float* dc0, Arena ar, hipStream_t st, float* dgates_out = nullptr,
               int dx_col0 = 0,            // dx is only formed for input columns >= dx_col0
               const Dropout* dh1b_drop = nullptr,     // mask still to be applied to dh1_b
               SmallPlan* dh0_plan = nullptr, bool* dh0_deferred = nullptr,
               const float** dx_slabs = nullptr, int* dx_ks = nullptr,     // (with dx_col0 > 0) leave d(x[:, dx_col0:]) as the
               bool pointwise_done = false) {   // dgates / dc0 were already formed (the epilogue of the product that completed dh1)
    // K-split slabs of its product: *dx_slabs -> [*dx_ks][B, I - dx_col0] (the consumer adds them up: one launch fewer);
    // *dx_ks == 0: dx was formed as usual
    // dh0_plan: do not launch dh0 = dgates W_hh; hand its launch plan to the caller (who pairs it with
    // an independent kernel) when the short-reduction kernel covers the shape
    float* dgates = dgates_out ? dgates_out : ar.take((size_t)B * 4 * H);
    NEED(dgates);
    if (!pointwise_done) {
        const LstmPwBwd p = cell_pw_bwd(gates, c0, c1, dh1, dh1_b, dc1, B, H, dgates, dc0, dh1b_drop);
        TRY(lstm_pointwise_bwd(p, st));
    }
    if (dx && dx_col0 == 0) {
        TRY(data_grad(dgates, 4 * H, w->w_ih, w->w_ih_t, B, 4 * H, I, dx, lddx, 0, ar, st));
    } else if (dx) {
        const int I2 = I - dx_col0;
        if (dx_ks) *dx_ks = 0;
        if (w->w_ih_t) {
            Seg sg{dgates, 4 * H, w->w_ih_t + (size_t)dx_col0 * 4 * H, 4 * H, 4 * H};
            LinearOut o{};
            o.y = dx + dx_col0; o.ldy = lddx; o.epi = EPI_NONE;
            float* slabs = (dx_slabs && dx_ks && g_slab_consumers) ? ar.take((size_t)16 * B * I2) : nullptr;
            int ks = 0;
            float* raw = nullptr;
            if (slabs && linear_nt(&sg, 1, B, I2, o, slabs, (size_t)16 * B * I2, st, &raw, &ks) == SF_OK && raw == slabs && ks >= 1) {
                *dx_slabs = slabs;
                *dx_ks = ks;
            } else {
                TRY(linear_nt(&sg, 1, B, I2, o, ar.rest(), ar.rest_n(), st));
            }
        } else {
            TRY(gemm_nn_ws(dgates, 4 * H, w->w_ih + dx_col0, I, B, I2, 4 * H, dx + dx_col0, lddx, 0,
                           ar.rest(), ar.rest_n(), st));
        }
    }
    if (dh0_deferred) *dh0_deferred = false;
    if (dh0 && dh0_plan && dh0_deferred && w->w_hh_t) {
        Seg sg{dgates, 4 * H, w->w_hh_t, 4 * H, 4 * H};
        LinearOut o{};
        o.y = dh0; o.ldy = H; o.epi = EPI_NONE;
        *dh0_deferred = linear_small_plan(&sg, 1, B, H, o, dh0_plan);
    }
    if (dh0 && !(dh0_deferred && *dh0_deferred))
        TRY(data_grad(dgates, 4 * H, w->w_hh, w->w_hh_t, B, 4 * H, H, dh0, H, 0, ar, st));
    if (g) {
        if (g->w_ih) TRY(gemm_tn(dgates, 4 * H, x, ldx, B, 4 * H, I, g->w_ih, I, 1, st, ar.rest(), ar.rest_n()));
        if (g->w_hh) TRY(gemm_tn(dgates, 4 * H, h0, H, B, 4 * H, H, g->w_hh, H, 1, st, ar.rest(), ar.rest_n()));
        TRY(colsum_pair(dgates, 4 * H, B, 4 * H, g->b_ih, g->b_hh, ar, st));
    }
    return SF_OK;
}

// ---- a1 VisualSoftDotAttention ---------------------------------------------------------------------
int visual_fwd_i(const sf_visual_w* w, const PanoSrc& X, int B, int H, int D, const float* h,
                 float* out, int ldo, float* alpha, float* t_v, float* q, const Dropout& drop,
                 int col0, Arena ar, hipStream_t st, const sf_decoder_fold* fold = nullptr, bool precise = false,
                 const sf_visual_fold64* fold64 = nullptr) {
    const int F = X.IMG + X.LOC;
    if (precise && !fold && fold64 && fold64->m_v && fold64->c_v && visual_attn_f64_supported(X, B) && !(H & 3)) {
        // inference through the float64 fold: q = M_v h + c_v in ONE product (t_v is not formed: no backward follows)
        double* q64 = reinterpret_cast<double*>(ar.take((size_t)B * F * 2));
        float* part = ar.take(visual_attn_split_floats(B, F));
        if (q64 && part && ar.tickets()) {
            TRY(linear_f64_w64(h, H, fold64->m_v, H, fold64->c_v, B, F, H, q64, F, q, F, st));
            return visual_attn(0, X, B, q, F, alpha, out, ldo, drop, col0, st, part, ar.tickets(), q64);
        }
    }
    if (precise && !fold && w->w_v_t && visual_attn_f64_supported(X, B) && !(H & 3) && !(D & 3)) {
        // The speaker's path encoder (csrc/sf_precise.hip): t_v, q and the scores in float64 -- each intermediate is
        // rounded ONCE.  t_v / q also land in their fp32 tapes (the backward reads those).
        double* t64 = reinterpret_cast<double*>(ar.take((size_t)B * D * 2));
        double* q64 = reinterpret_cast<double*>(ar.take((size_t)B * F * 2));
        float* part = ar.take(visual_attn_split_floats(B, F));
        if (t64 && q64 && part && ar.tickets()) {
            TRY(linear_f64(h, nullptr, H, w->w_h, H, w->b_h, B, D, H, t64, D, t_v, D, st));
            TRY(linear_f64(nullptr, t64, D, w->w_v_t, D, nullptr, B, F, D, q64, F, q, F, st));
            return visual_attn(0, X, B, q, F, alpha, out, ldo, drop, col0, st, part, ar.tickets(), q64);
        }
    }
    if (fold) {      // inference: q = M_v h + c_v in one product
        TRY(linear_plain(h, H, fold->m_v, H, fold->c_v, B, F, H, EPI_NONE, q, F, ar, st));
        float* part = B <= 1024 ? ar.take(visual_attn_split_floats(B, F)) : nullptr;
        return visual_attn(0, X, B, q, F, alpha, out, ldo, drop, col0, st, part,
                           part ? ar.tickets() : nullptr);
    }
    TRY(linear_plain(h, H, w->w_h, H, w->b_h, B, D, H, EPI_NONE, t_v, D, ar, st));
    if (w->w_v_t)   // q = t_v W_v as a K-contiguous product against the transposed copy
        TRY(linear_plain(t_v, D, w->w_v_t, D, nullptr, B, F, D, EPI_NONE, q, F, ar, st));
    else
        TRY(gemm_nn_ws(t_v, D, w->w_v, F, B, F, D, q, F, 0, ar.rest(), ar.rest_n(), st));
    // (the products above are done with their slabs by the time the attention kernel runs: the
    // partials may reuse that part of the workspace)
    float* part = B <= 1024 ? ar.take(visual_attn_split_floats(B, F)) : nullptr;
    return visual_attn(0, X, B, q, F, alpha, out, ldo, drop, col0, st, part,
                       part ? ar.tickets() : nullptr);
}

int visual_bwd_i(const sf_visual_w* w, const sf_visual_g* g, const PanoSrc& X, int B, int H, int D,
                 const float* h, const float* alpha, const float* t_v, const float* dout, int lddo,
                 const Dropout& drop, int col0, float* dh, Arena ar, hipStream_t st,
                 float* dq_out = nullptr, float* dt_out = nullptr,
                 const SmallPlan* beside = nullptr,     // an independent small product to launch with
                 int dout_slabs = 0, long dout_slab_stride = 0,     // dout = the sum of K-split slabs (added up in the kernel)
                 const LstmPwBwd* next_pw = nullptr, bool* next_fused = nullptr) {   // dh completes the NEXT backward step's dh1:
                                                        // run that step's pointwise backward as the product's epilogue
    const int F = X.IMG + X.LOC;                        // the attention backward (must run either way)
    float* dq = dq_out ? dq_out : ar.take((size_t)B * F);
    float* dt = dt_out ? dt_out : ar.take((size_t)B * D);
    NEED(dq && dt);
    bool paired = false;
    if (beside) {
        const int rc = pair_visbwd_small(X, B, dout, lddo, const_cast<float*>(alpha), dq, F, drop, col0,
                                         *beside, st, dout_slabs, dout_slab_stride);
        if (rc == SF_OK) paired = true;
        else if (rc != SF_ERR_UNSUPPORTED) return rc;
        else TRY(launch_small_plan_x(*beside, st));
    }
    if (!paired) TRY(visual_attn(1, X, B, dout, lddo, const_cast<float*>(alpha), dq, F, drop, col0, st, nullptr, nullptr,
                                 nullptr, dout_slabs, dout_slab_stride));
    TRY(linear_plain(dq, F, w->w_v, F, nullptr, B, D, F, EPI_NONE, dt, D, ar, st));
    if (g && g->w_v) TRY(gemm_tn(t_v, D, dq, F, B, D, F, g->w_v, F, 1, st, ar.rest(), ar.rest_n()));
    // g->b_v: the bias shifts all V scores of a row equally; its gradient is identically zero.
    if (next_fused) *next_fused = false;
    if (dh) {
        bool done = false;
        if (next_pw && next_fused && g_fuse_cell_bwd && w->w_h_t && H % 4 == 0) {
            Seg sg{dt, D, w->w_h_t, D, D};
            LinearOut o{};
            o.y = dh; o.ldy = H; o.epi = EPI_NONE; o.accumulate = 1;
            SmallPlan sp;
            if (linear_small_plan(&sg, 1, B, H, o, &sp)) {
                const int rc = launch_small_plan_pw(sp, *next_pw, st);
                if (rc == SF_OK) done = *next_fused = true;
                else if (rc != SF_ERR_UNSUPPORTED) return rc;
            }
        }
        if (!done) TRY(data_grad(dt, D, w->w_h, w->w_h_t, B, D, H, dh, H, 1, ar, st));
    }
    if (g && g->w_h) TRY(gemm_tn(dt, D, h, H, B, D, H, g->w_h, H, 1, st, ar.rest(), ar.rest_n()));
    if (g && g->b_h) TRY(colsum(dt, D, B, D, g->b_h, 1, st, nullptr, ar.rest(), ar.rest_n()));
    return SF_OK;
}

// ---- a3 SoftDotAttention ----------------------------------------------------------------------------
// h lives in cat2[:, H:2H] already (copy_from != null copies it there first).
int softdot_fwd_i(const sf_softdot_w* w, int B, int L, int H, const float* copy_from, int ldh,
                  const float* ctx, const uint8_t* mask, float* h_tilde, float* alpha, float* cat2,
                  float* t_text, Arena ar, hipStream_t st, const int32_t* ctx_row = nullptr) {
    if (copy_from) {
        Dropout none = make_dropout(nullptr, 0);
        TRY(dropout_copy(copy_from, ldh, B, H, cat2 + H, 2 * H, none, 0, st));
    }
    TRY(linear_plain(cat2 + H, 2 * H, w->w_in, H, nullptr, B, H, H, EPI_NONE, t_text, H, ar, st));
    TRY(text_attn_fwd(ctx, mask, B, L, H, t_text, H, alpha, cat2, 2 * H, st, ctx_row));
    return linear_plain(cat2, 2 * H, w->w_out, 2 * H, nullptr, B, H, 2 * H, EPI_TANH, h_tilde, H, ar,
                        st);
}

int softdot_bwd_i(const sf_softdot_w* w, const sf_softdot_g* g, int B, int L, int H,
                  const float* ctx, const float* alpha, const float* cat2, const float* t_text,
                  const float* h_tilde, const float* dh_tilde, float* dh, int lddh, float* dctx,
                  Arena ar, hipStream_t st, float* dpre_out = nullptr, float* dt_out = nullptr,
                  bool dpre_ready = false,     // dh_tilde already IS dpre (written to dpre_out)
                  float* dcat2_out = nullptr, float* ds_out = nullptr,     // both: dctx is deferred
                  const int32_t* ctx_row = nullptr) {                      // (deferred only) row b reads ctx row ctx_row[b]
    float* dpre = dpre_out ? dpre_out : ar.take((size_t)B * H);
    float* dt = dt_out ? dt_out : ar.take((size_t)B * H);
    float* dcat2 = dcat2_out ? dcat2_out : ar.take((size_t)B * 2 * H);
    const bool defer = dcat2_out && ds_out;
    NEED(dpre && dcat2 && dt);
    if (dpre_ready) dpre = const_cast<float*>(dh_tilde);
    else TRY(tanh_bwd(h_tilde, H, dh_tilde, H, B, H, dpre, H, st));
    TRY(data_grad(dpre, H, w->w_out, w->w_out_t, B, H, 2 * H, dcat2, 2 * H, 0, ar, st));
    if (g && g->w_out) TRY(gemm_tn(dpre, H, cat2, 2 * H, B, H, 2 * H, g->w_out, 2 * H, 1, st, ar.rest(), ar.rest_n()));
    TRY(text_attn_bwd(ctx, B, L, H, dcat2, 2 * H, t_text, H, alpha, dt, H, defer ? nullptr : dctx, st,
                      defer ? ds_out : nullptr, ctx_row));
    // dh = dcat2[:, H:] + dt W_in: the addend rides in the epilogue of the product
    bool dh_done = false;
    if (w->w_in_t) {
        Seg sg{dt, H, w->w_in_t, H, H};
        LinearOut o{};
        o.y = dh; o.ldy = lddh; o.epi = EPI_NONE; o.addend = dcat2 + H; o.ld_addend = 2 * H;
        const int rc = linear_nt(&sg, 1, B, H, o, ar.rest(), ar.rest_n(), st);
        if (rc == SF_OK) dh_done = true;
        else if (rc != SF_ERR_UNSUPPORTED) return rc;
    }
    if (!dh_done) {
        TRY(add2(dcat2 + H, 2 * H, nullptr, 0, B, H, dh, lddh, st));
        TRY(data_grad(dt, H, w->w_in, w->w_in_t, B, H, H, dh, lddh, 1, ar, st));
    }
    if (g && g->w_in) TRY(gemm_tn(dt, H, cat2 + H, 2 * H, B, H, H, g->w_in, H, 1, st, ar.rest(), ar.rest_n()));
    return SF_OK;
}

// ---- a4 EltwiseProdScoring --------------------------------------------------------------------------
int scoring_fwd_i(const sf_scoring_w* w, const CandSrc& U, int B, int H, int D, const float* h,
                  float* logit, float* t_a, float* wt, float* r, Arena ar, hipStream_t st,
                  const sf_follower_glue* glue = nullptr, const sf_decoder_fold* fold = nullptr,
                  bool t_a_done = false) {     // t_a / wt already formed by the caller (paired launch)
    const int F = U.IMG + U.LOC;
    if (fold) {      // inference: [r | c] = M_a h~ + c_a in one product ([B, F+4] scratch)
        float* rext = ar.take((size_t)B * (F + 4));
        NEED(rext);
        TRY(linear_plain(h, H, fold->m_a, H, fold->c_a, B, F + 4, H, EPI_NONE, rext, F + 4, ar, st));
        if (glue)
            return score_glue_fwd(U, B, D, rext, nullptr, nullptr, nullptr,
                                  make_glue(U, B, logit, glue), st, F + 4, rext + F);
        return score_fwd(U, B, D, rext, nullptr, nullptr, nullptr, logit, st, F + 4, rext + F);
    }
    Seg sg{h, H, w->w_h, H, H};
    LinearOut o{};
    o.y = wt; o.ldy = D; o.bias = w->b_h; o.mul = w->w_out; o.y_pre = t_a; o.ldy_pre = D;
    o.epi = EPI_MUL;
    if (!t_a_done) TRY(linear_nt(&sg, 1, B, D, o, ar.rest(), ar.rest_n(), st));
    if (w->w_a_t)
        TRY(linear_plain(wt, D, w->w_a_t, D, nullptr, B, F, D, EPI_NONE, r, F, ar, st));
    else
        TRY(gemm_nn_ws(wt, D, w->w_a, F, B, F, D, r, F, 0, ar.rest(), ar.rest_n(), st));
    if (glue) return score_glue_fwd(U, B, D, r, wt, w->b_a, w->b_out, make_glue(U, B, logit, glue), st);
    return score_fwd(U, B, D, r, wt, w->b_a, w->b_out, logit, st);
}

int scoring_bwd_i(const sf_scoring_w* w, const sf_scoring_g* g, const CandSrc& U, int B, int H,
                  int D, const float* h, const float* t_a, const float* wt, const float* dlogit,
                  float* dh, Arena ar, hipStream_t st, const sf_decoder_gtape* gt = nullptr,
                  const float* tanh_of = nullptr, bool* tanh_done = nullptr,
                  const CeSrc* ce = nullptr) {
    // tanh_of (= h, the tanh output that fed the scoring): when given and the shape allows, dh is
    // returned already multiplied by (1 - tanh_of^2) and *tanh_done is set
    const int F = U.IMG + U.LOC;
    float* dr = gt ? gt->dr : ar.take((size_t)B * F);
    float* dc = gt ? gt->dc : ar.take((size_t)B);
    float* dwt = gt ? gt->dwt : ar.take((size_t)B * D);
    float* dta = gt ? gt->dta : ar.take((size_t)B * D);
    NEED(dr && dc && dwt && dta);
    TRY(score_bwd(U, B, dlogit, dr, dc, st, ce));
    // dwt = dr W_a^T + dc (x) b_a,  dta = dwt * w_out: one launch (rank-1 term and column scale in
    // the epilogue of the product) where the short-reduction kernel covers the shape
    bool dta_done = false;
    {
        Seg sg{dr, F, w->w_a, F, F};
        LinearOut o{};
        o.y = dta; o.ldy = D; o.y_pre = dwt; o.ldy_pre = D; o.epi = EPI_MUL; o.mul = w->w_out;
        o.r1_s = dc; o.r1_v = w->b_a;
        const int rc = linear_nt(&sg, 1, B, D, o, ar.rest(), ar.rest_n(), st);
        if (rc == SF_OK) dta_done = true;
        else if (rc != SF_ERR_UNSUPPORTED) return rc;
    }
    if (!dta_done) {
        TRY(linear_plain(dr, F, w->w_a, F, nullptr, B, D, F, EPI_NONE, dwt, D, ar, st));
        TRY(rank1_add(dc, w->b_a, B, D, dwt, D, st));
    }
    if (g) {
        if (g->w_a) TRY(gemm_tn(wt, D, dr, F, B, D, F, g->w_a, F, 1, st, ar.rest(), ar.rest_n()));
        if (g->b_a) TRY(dot_rows_accum(dc, wt, D, B, D, g->b_a, st));
        if (g->b_out) TRY(sum_accum(dc, B, g->b_out, st));
        if (g->w_out) TRY(colsum_prod(dwt, D, t_a, D, B, D, g->w_out, st));
    }
    if (!dta_done) TRY(scale_cols(dwt, D, w->w_out, B, D, dta, D, st));
    if (dh && tanh_of && w->w_h_t) {
        // dh <- (dta W_h) * (1 - tanh_of^2): the backward through h~ = tanh(.) rides in the epilogue
        Seg sg{dta, D, w->w_h_t, D, D};
        LinearOut o{};
        o.y = dh; o.ldy = H; o.epi = EPI_TANHBWD; o.aux = tanh_of; o.ld_aux = H;
        const int rc = linear_nt(&sg, 1, B, H, o, ar.rest(), ar.rest_n(), st);
        if (rc == SF_OK) { if (tanh_done) *tanh_done = true; }
        else if (rc != SF_ERR_UNSUPPORTED) return rc;
        else TRY(data_grad(dta, D, w->w_h, w->w_h_t, B, D, H, dh, H, 0, ar, st));
    } else if (dh) {
        TRY(data_grad(dta, D, w->w_h, w->w_h_t, B, D, H, dh, H, 0, ar, st));
    }
    if (g && g->w_h) TRY(gemm_tn(dta, D, h, H, B, D, H, g->w_h, H, 1, st, ar.rest(), ar.rest_n()));
    if (g && g->b_h) TRY(colsum(dta, D, B, D, g->b_h, 1, st, nullptr, ar.rest(), ar.rest_n()));
    return SF_OK;
}

}  // namespace

extern "C" {

size_t sf_workspace_bytes(void) { return WORKSPACE_BYTES; }
size_t sf_wgrad_workspace_bytes(void) { return WGRAD_WORKSPACE_BYTES; }
int sf_abi_version(void) { return SF_ABI_VERSION; }
#ifndef SF_BUILD_ID
#define SF_BUILD_ID "unknown"
#endif
// (the marker in front of the id lets build.py read it from the file without loading the library)
const char* sf_build_id(void) { static const char id[] = "SF_BUILD_ID=" SF_BUILD_ID; return id + 12; }
void sf_debug_persist_timeout(long long ticks) { sf::g_persist_timeout = ticks; }
void sf_debug_gate_product_f32(int on) { sf::g_nt_force_f32 = on; }
void sf_debug_many_row_product(int on) { sf::g_nt_big = on & 1; sf::g_nt_big_ksplit = (on & 2) ? 0 : 1; }
void sf_debug_grouped_weight_gradients(int on) { sf::g_tn_group = on; }
void sf_debug_slab_consumers(int on) { g_slab_consumers = on; }
void sf_debug_fused_cell_backward(int on) { g_fuse_cell_bwd = on; }
void sf_debug_bptt_lookahead(int steps) { g_bptt_lookahead = steps; }
void sf_debug_bptt_flags(int on) { g_bptt_flags = on; }
void sf_debug_bptt_part(int part) { g_bptt_part = part; }
int sf_debug_cotenant(int blocks, int threads, int lds_bytes, long long ticks, float* sink, sf_stream stream) {
    SF_ENTER();
    return sf::cotenant(blocks, threads, lds_bytes, ticks, sink, S(stream));
}
void sf_gate_product_strict(int on) { sf::g_nt_force_f32 = on ? 1 : 0; }
int sf_gate_product_is_strict(void) { return sf::g_nt_force_f32 != 0; }
void sf_debug_fold_merge_with_glue(int on) { g_fold_merge_with_glue = on; }
void sf_debug_fold_chain3(int on) { g_fold_chain3 = on; }
void sf_debug_fold_build_overlap(int on) { g_fold_build_overlap = on; }
void sf_debug_precise_attention(int on) { sf::g_precise_attention = on; }
void sf_debug_tn_split_min_rows(int rows) { sf::g_tn_split_min_rows = rows < 0 ? 4096 : rows; }
size_t sf_workspace_fault_offset(size_t ws_bytes) {
    const size_t n = ws_bytes / 4;
    if (n <= SYNC_WORDS) return 0;
    return (n - SYNC_WORDS + PERSIST_TICKET + persistent_fault_word()) * 4;
}
const char* sf_last_error_string(void) { return hipGetErrorString(g_last_hip_error); }
const char* sf_status_string(int s) {
    switch (s) {
        case SF_OK: return "ok";
        case SF_ERR_ARG: return "invalid argument";
        case SF_ERR_UNSUPPORTED: return "unsupported shape";
        case SF_ERR_LAUNCH: return "kernel launch failed";
        case SF_ERR_WORKSPACE: return "workspace too small";
        default: return "unknown status";
    }
}

int sf_linear_fwd(const float* x, int ldx, const float* w, const float* b, int M, int N, int K,
                  int act, float* y, int ldy, void* ws, size_t ws_bytes, sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(x && w && y && M > 0 && N > 0 && K > 0 && (act == 0 || act == 1));
    return linear_plain(x, ldx, w, K, b, M, N, K, act ? EPI_TANH : EPI_NONE, y, ldy,
                        arena(ws, ws_bytes), S(stream));
}

int sf_linear_slabs_fwd(const float* x, int ldx, const float* w, int K1, const float* h, int ldh,
                        const float* u, int K2, int M, int N, int* ksplit, void* ws, size_t ws_bytes,
                        sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(x && w && ksplit && ws && M > 0 && N > 0 && K1 > 0 && (!h || (u && K2 > 0)));
    Arena ar = arena(ws, ws_bytes);
    Seg segs[2] = {{x, ldx, w, K1, K1}, {h, ldh, u, K2, K2}};
    LinearOut o{};
    float* slabs = nullptr;
    return linear_nt(segs, h ? 2 : 1, M, N, o, ar.rest(), ar.rest_n(), S(stream), &slabs, ksplit);
}

int sf_linear_bwd(const float* x, int ldx, const float* w, const float* y, int ldy, const float* dy,
                  int lddy, int M, int N, int K, int act, float* dx, int lddx, int accumulate_dx,
                  float* dw, float* db, void* ws, size_t ws_bytes, sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(x && w && dy && M > 0 && N > 0 && K > 0 && (N % 4 == 0) && (K % 4 == 0));
    Arena ar = arena(ws, ws_bytes);
    hipStream_t st = S(stream);
    const float* dpre = dy;
    int ldp = lddy;
    if (act == 1) {
        SF_CHECK_ARG(y);
        float* t = ar.take((size_t)M * N);
        NEED(t);
        TRY(tanh_bwd(y, ldy, dy, lddy, M, N, t, N, st));
        dpre = t;
        ldp = N;
    }
    if (dx) TRY(gemm_nn_ws(dpre, ldp, w, K, M, K, N, dx, lddx, accumulate_dx, ar.rest(), ar.rest_n(), st));
    if (dw) TRY(gemm_tn(dpre, ldp, x, ldx, M, N, K, dw, K, 1, st, ar.rest(), ar.rest_n()));
    if (db) TRY(colsum(dpre, ldp, M, N, db, 1, st, nullptr, ar.rest(), ar.rest_n()));
    return SF_OK;
}

int sf_lstm_cell_fwd(const sf_lstm_w* w, int B, int I, int H, const float* x, int ldx,
                     const float* h0, const float* c0, float* h1, float* c1, float* gates,
                     float* h1_drop, int ld_h1_drop, const sf_dropout* drop, uint32_t drop_stream,
                     void* ws, size_t ws_bytes, sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(w && x && h0 && c0 && h1 && c1 && B > 0 && I > 0 && H > 0 && H % 4 == 0);
    return lstm_fwd_i(w, B, I, H, x, ldx, h0, c0, h1, c1, gates, h1_drop, ld_h1_drop,
                      make_dropout(drop, drop_stream), arena(ws, ws_bytes), S(stream));
}

int sf_lstm_cell_bwd(const sf_lstm_w* w, const sf_lstm_g* g, int B, int I, int H, const float* x,
                     int ldx, const float* h0, const float* c0, const float* c1, const float* gates,
                     const float* dh1, const float* dc1, float* dx, int lddx, float* dh0, float* dc0,
                     void* ws, size_t ws_bytes, sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(w && x && h0 && c0 && c1 && gates && dc0 && B > 0 && H % 4 == 0 && I % 4 == 0);
    return lstm_bwd_i(w, g, B, I, H, x, ldx, h0, c0, c1, gates, dh1, nullptr, dc1, dx, lddx, dh0,
                      dc0, arena(ws, ws_bytes), S(stream));
}

int sf_visual_attention_fwd(const sf_visual_w* w, const sf_pano* X, int B, int H, int D,
                            const float* h, float* out, int ldo, float* alpha, float* t_v, float* q,
                            const sf_dropout* drop, uint32_t drop_stream, int drop_col0, void* ws,
                            size_t ws_bytes, sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(w && X && h && out && alpha && t_v && q && B > 0);
    return visual_fwd_i(w, pano(X), B, H, D, h, out, ldo, alpha, t_v, q,
                        make_dropout(drop, drop_stream), drop_col0, arena(ws, ws_bytes), S(stream));
}

int sf_visual_attention_fwd_f64(const sf_visual_w* w, const sf_pano* X, int B, int H, int D,
                                const float* h, float* out, int ldo, float* alpha, float* t_v, float* q,
                                const sf_dropout* drop, uint32_t drop_stream, int drop_col0, void* ws,
                                size_t ws_bytes, sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(w && X && h && out && alpha && t_v && q && B > 0);
    const PanoSrc src = pano(X);
    const int F = src.IMG + src.LOC;
    Arena ar = arena(ws, ws_bytes);
    if (!w->w_v_t || !visual_attn_f64_supported(src, B) || (H & 3) || (D & 3) || !ar.tickets() ||
        ar.rest_n() < (size_t)B * (D + F) * 2 + visual_attn_split_floats(B, F) + 256)
        return SF_ERR_UNSUPPORTED;
    return visual_fwd_i(w, src, B, H, D, h, out, ldo, alpha, t_v, q, make_dropout(drop, drop_stream), drop_col0, ar,
                        S(stream), nullptr, true);
}

int sf_linear_f64(const float* x, int ldx, const float* w, int ldw, const float* b, int M, int N, int K, double* y64,
                  float* y32, sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(x && w && (y64 || y32) && M > 0 && N > 0 && K > 0);
    return linear_f64(x, nullptr, ldx, w, ldw, b, M, N, K, y64, N, y32, N, S(stream));
}

int sf_visual_attention_bwd(const sf_visual_w* w, const sf_visual_g* g, const sf_pano* X, int B,
                            int H, int D, const float* h, const float* alpha, const float* t_v,
                            const float* dout, int lddo, const sf_dropout* drop,
                            uint32_t drop_stream, int drop_col0, float* dh, void* ws,
                            size_t ws_bytes, sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(w && X && h && alpha && t_v && dout && B > 0);
    return visual_bwd_i(w, g, pano(X), B, H, D, h, alpha, t_v, dout, lddo,
                        make_dropout(drop, drop_stream), drop_col0, dh, arena(ws, ws_bytes),
                        S(stream));
}

int sf_soft_dot_attention_fwd(const sf_softdot_w* w, int B, int L, int H, const float* h, int ldh,
                              const float* ctx, const uint8_t* mask, const int32_t* ctx_row,
                              float* h_tilde, float* alpha, float* cat2, float* t_text, void* ws,
                              size_t ws_bytes, sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(w && h && ctx && h_tilde && alpha && cat2 && t_text && B > 0 && L > 0);
    return softdot_fwd_i(w, B, L, H, h, ldh, ctx, mask, h_tilde, alpha, cat2, t_text,
                         arena(ws, ws_bytes), S(stream), ctx_row);
}

int sf_soft_dot_attention_bwd(const sf_softdot_w* w, const sf_softdot_g* g, int B, int L, int H,
                              const float* ctx, const float* alpha, const float* cat2,
                              const float* t_text, const float* h_tilde, const float* dh_tilde,
                              float* dh, int lddh, float* dctx, void* ws, size_t ws_bytes,
                              sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(w && ctx && alpha && cat2 && t_text && h_tilde && dh_tilde && dh && B > 0);
    return softdot_bwd_i(w, g, B, L, H, ctx, alpha, cat2, t_text, h_tilde, dh_tilde, dh, lddh, dctx,
                         arena(ws, ws_bytes), S(stream));
}

// ContextOnlySoftDotAttention core (model.py:166-177 behind its linear_in): scores, masked softmax, weighted context
int sf_text_attention_fwd(const float* ctx, const uint8_t* mask, int B, int L, int H, const float* t, int ldt,
                          float* alpha, float* wc, int ldwc, sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(ctx && t && alpha && wc && B > 0 && L > 0 && H > 0);
    return text_attn_fwd(ctx, mask, B, L, H, t, ldt, alpha, wc, ldwc, S(stream));
}

int sf_text_attention_bwd(const float* ctx, int B, int L, int H, const float* dwc, int lddwc, const float* t, int ldt,
                          const float* alpha, float* dt, int lddt, float* dctx, sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(ctx && dwc && t && alpha && dt && B > 0 && L > 0 && H > 0);
    return text_attn_bwd(ctx, B, L, H, dwc, lddwc, t, ldt, alpha, dt, lddt, dctx, S(stream));
}

int sf_eltwise_prod_scoring_fwd(const sf_scoring_w* w, const sf_cands* U, int B, int H, int D,
                                const float* h, float* logit, float* t_a, float* wt, float* r,
                                void* ws, size_t ws_bytes, sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(w && U && h && logit && t_a && wt && r && B > 0);
    return scoring_fwd_i(w, cands(U), B, H, D, h, logit, t_a, wt, r, arena(ws, ws_bytes), S(stream));
}

int sf_eltwise_prod_scoring_bwd(const sf_scoring_w* w, const sf_scoring_g* g, const sf_cands* U,
                                int B, int H, int D, const float* h, const float* t_a,
                                const float* wt, const float* dlogit, float* dh, void* ws,
                                size_t ws_bytes, sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(w && U && h && t_a && wt && dlogit && B > 0);
    return scoring_bwd_i(w, g, cands(U), B, H, D, h, t_a, wt, dlogit, dh, arena(ws, ws_bytes),
                         S(stream));
}

// ---- a6 AttnDecoderLSTM.forward (model.py:377-397) -------------------------------------------------
int sf_attn_decoder_fwd(const sf_decoder_w* w, const sf_pano* X, const sf_cands* U, int B, int H,
                        int D, int L, const float* u_prev, const float* h0, const float* c0,
                        const float* ctx, const uint8_t* ctx_mask, const int32_t* ctx_row,
                        const sf_decoder_tape* tp, const sf_follower_glue* glue,
                        const sf_dropout* drop, uint32_t step_id,
                        void* ws, size_t ws_bytes, sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(w && X && U && h0 && c0 && ctx && tp && B > 0 && L > 0 && (!glue || glue_ok(U, glue)));
    Arena ar = arena(ws, ws_bytes);
    hipStream_t st = S(stream);
    const PanoSrc xs = pano(X);
    const int F = xs.IMG + xs.LOC;
    const Dropout d_in = make_dropout(drop, 2 * step_id, 2), d_h = make_dropout(drop, 2 * step_id + 1, 2);
    // model.py:389  feature, alpha_v = visual_attention(h_0, X)  -> straight into xin[:, F:2F]
    TRY(visual_fwd_i(&w->visual, xs, B, H, D, h0, tp->xin + F, 2 * F, tp->alpha_v, tp->t_v, tp->q,
                     d_in, F, ar, st, w->fold));
    // model.py:391-392  drop(cat(u_prev, feature))
    if (u_prev) TRY(dropout_copy(u_prev, F, B, F, tp->xin, 2 * F, d_in, 0, st));
    // model.py:393-394  LSTMCell; dropout(h_1) lands in cat2[:, H:2H]
    TRY(lstm_fwd_i(&w->lstm, B, 2 * F, H, tp->xin, 2 * F, h0, c0, tp->h1, tp->c1, tp->gates,
                   tp->cat2 + H, 2 * H, d_h, ar, st));
    // model.py:395  text attention
    TRY(softdot_fwd_i(&w->text, B, L, H, nullptr, 0, ctx, ctx_mask, tp->h_tilde, tp->alpha, tp->cat2,
                      tp->t_text, ar, st, ctx_row));
    // model.py:396  action logits
    return scoring_fwd_i(&w->action, cands(U), B, H, D, tp->h_tilde, tp->logit, tp->t_a, tp->wt,
                         tp->r, ar, st, glue, w->fold);
}

// ---- a6, software-pipelined across steps ---------------------------------------------------------------
// head(t)   = visual half of step t: t_v, q, visual attention -> tape->xin[:, F:2F], alpha_v
// tail(t)   = LSTM cell, text attention, scoring (+ glue) of step t, and -- when X_next is given --
//             head(t+1) on h1 of step t, run SIDE BY SIDE with the text / scoring half in paired
//             launches (sf_attention.hip): the two halves are independent given h1.
static int decoder_head_a(const sf_decoder_w* w, const sf_pano* X, int B, int H, int D,
                          const float* h0, const sf_decoder_tape* tp, const sf_dropout* drop,
                          uint32_t step_id, Arena ar, sf_stream stream) {
    SF_CHECK_ARG(w && X && h0 && tp && B > 0);
    const PanoSrc xs = pano(X);
    const int F = xs.IMG + xs.LOC;
    return visual_fwd_i(&w->visual, xs, B, H, D, h0, tp->xin + F, 2 * F, tp->alpha_v, tp->t_v, tp->q,
                        make_dropout(drop, 2 * step_id, 2), F, ar, S(stream), w->fold);
}
static int decoder_head_i(const sf_decoder_w* w, const sf_pano* X, int B, int H, int D,
                          const float* h0, const sf_decoder_tape* tp, const sf_dropout* drop,
                          uint32_t step_id, void* ws, size_t ws_bytes, sf_stream stream) {
    return decoder_head_a(w, X, B, H, D, h0, tp, drop, step_id, arena(ws, ws_bytes), stream);
}

int sf_attn_decoder_head_fwd(const sf_decoder_w* w, const sf_pano* X, int B, int H, int D,
                             const float* h0, const sf_decoder_tape* tp, const sf_dropout* drop,
                             uint32_t step_id, void* ws, size_t ws_bytes, sf_stream stream) {
    SF_ENTER();
    return decoder_head_i(w, X, B, H, D, h0, tp, drop, step_id, ws, ws_bytes, stream);
}

static int plan_linear(const float* x, int ldx, const float* wgt, int ldw, const float* b, int M,
                       int N, int K, Epi epi, float* y, int ldy, SmallPlan* p) {
    Seg sg{x, ldx, wgt, ldw, K};
    LinearOut o{};
    o.y = y; o.ldy = ldy; o.bias = b; o.epi = epi;
    return linear_small_plan(&sg, 1, M, N, o, p) ? SF_OK : SF_ERR_UNSUPPORTED;
}

// the episode's folded context tensors (sf_follower_episode.ctx_q / ctx_o; sf_attention.hip: text_fold_body)
struct TextFold {
    const float* ctx_q;
    const float* ctx_o;
    const sf_decoder_fold* mats;      // optional (sf_follower_episode.chain_fold): the THREE-launch chain
};

static int decoder_tail_i(const sf_decoder_w* w, const sf_cands* U, int B, int H, int D, int L,
                          const float* u_prev, const float* h0, const float* c0, const float* ctx,
                          const uint8_t* ctx_mask, const int32_t* ctx_row, const sf_decoder_tape* tp,
                          const sf_follower_glue* glue, const sf_dropout* drop, uint32_t step_id,
                          const sf_pano* X_next, const sf_decoder_tape* tn, void* ws, size_t ws_bytes,
                          sf_stream stream, const TextFold* tf = nullptr) {
    SF_CHECK_ARG(w && U && h0 && c0 && ctx && tp && B > 0 && L > 0 && (!glue || glue_ok(U, glue)) &&
                 (!X_next || tn));
    Arena ar = arena(ws, ws_bytes);
    hipStream_t st = S(stream);
    const CandSrc us = cands(U);
    // tape_next WITHOUT X_next: the next panorama is not known yet (it depends on this step's action: a
    // device-resident environment).  Only the query of the next step's visual attention (t_v', q': they need
    // nothing but h1) rides beside the text stages; the attention itself follows the environment step
    // (sf_attn_decoder_attend_fwd).
    const bool query_only = !X_next && tn && w->visual.w_v_t && !w->fold;
    const int F = us.IMG + us.LOC;
    const Dropout d_in = make_dropout(drop, 2 * step_id, 2), d_h = make_dropout(drop, 2 * step_id + 1, 2);
    if (u_prev) TRY(dropout_copy(u_prev, F, B, F, tp->xin, 2 * F, d_in, 0, st));
    TRY(lstm_fwd_i(&w->lstm, B, 2 * F, H, tp->xin, 2 * F, h0, c0, tp->h1, tp->c1, tp->gates,
                   tp->cat2 + H, 2 * H, d_h, ar, st));
    const sf_softdot_w* tw = &w->text;
    const sf_visual_w* vw = &w->visual;
    bool paired = X_next && vw->w_v_t;     // (a scoring fold, if any, is applied by scoring_fwd_i)
    const bool last_step = !X_next && !tn;      // (nothing of a next step to prepare: the text chain alone)
    if ((paired || query_only || last_step) && tf && tf->mats && g_fold_chain3 && !w->fold && !ctx_row && !d_h.on() &&
        tw->w_out) {
        // Folded text stage + folded query / scoring products (sf_decoder_fold through sf_follower_episode.chain_fold):
        // THREE dependent launches behind the cell --
        //   (1) folded text attention  ||  y = W_out[:, H:] h1  ||  q' = M_v h1 + c_v      (t_v' is never formed)
        //   (2) [r | c] = M_a tanh(z + y) + c_a (A-prologue)   ||  visual-attention partials of step t+1
        //   (3) scoring + glue (logit = u . r + c)             ||  merge of the partials
        // (a device-resident environment: (2) is the product alone, the attention follows the environment step)
        const sf_decoder_fold* fm = tf->mats;
        const PanoSrc xn = paired ? pano(X_next) : PanoSrc{};
        const Dropout dn_in = make_dropout(drop, 2 * (step_id + 1), 2);
        Arena af = ar;
        float* tpart = af.take(text_fold_part_floats(B, H));
        // (leading dimension H + 16: rows of a power-of-two stride share a few cache sets, CHANGELOG round 5 "2c")
        const int ldp = H + 16;
        float* ybuf = af.take((size_t)B * ldp);
        float* zbuf = af.take((size_t)B * ldp);
        float* rext = af.take((size_t)B * (F + 4));
        unsigned* tcount = af.tickets() ? af.tickets() + TEXT_TICKET : nullptr;
        float* part = (paired && B <= 1024) ? af.take(visual_attn_split_floats(B, F)) : nullptr;
        SmallPlan py, pq, pm;
        bool ok = tpart && ybuf && zbuf && rext && (part || !paired) && tcount && B <= 256 && fm->m_v && fm->c_v && fm->m_a &&
                  fm->c_a &&
                  plan_linear(tp->cat2 + H, 2 * H, tw->w_out + H, 2 * H, nullptr, B, H, H, EPI_NONE, ybuf, ldp, &py) == SF_OK;
        if (ok && !last_step) {
            ok = plan_linear(tp->h1, H, fm->m_v, H, fm->c_v, B, F, H, EPI_NONE, tn->q, F, &pq) == SF_OK;
        } else if (ok) {
            pq = py;
            pq.gx = pq.gy = 0;
        }
        if (ok) {
            Seg sg{ybuf, ldp, fm->m_a, H, H};
            LinearOut o{};
            o.y = rext; o.ldy = F + 4; o.bias = fm->c_a; o.epi = EPI_NONE;
            ok = linear_small_plan(&sg, 1, B, F + 4, o, &pm) && pm.cpw == 4 && py.mt == 1 && py.cpw == 4 && pq.cpw == 4;
            pm.args.apro_part = zbuf;
            pm.args.apro_stride = ldp;
        }
        if (ok) {
            const int rc = pair_textfold_small_small(tf->ctx_q, tf->ctx_o, ctx_mask, B, L, H, tp->cat2 + H, 2 * H, tpart,
                                                     tcount, zbuf, ldp, tp->alpha, py, pq, st);
            if (rc == SF_OK) {
                TRY(pair_vis_apro(paired ? &xn : nullptr, B, paired ? tn->q : nullptr, F, part, pm, st));
                if (paired && glue)
                    return pair_score_merge(us, B, D, rext, nullptr, nullptr, nullptr, make_glue(us, B, tp->logit, glue), xn,
                                            tn->alpha_v, tn->xin + F, 2 * F, dn_in, F, part, st, F + 4, rext + F);
                if (paired) {                               // (no glue: the module-API step; merge by its own launch)
                    SmallPlan none = py;
                    none.gx = none.gy = 0;
                    TRY(pair_vis_small(xn, B, tn->q, F, tn->alpha_v, tn->xin + F, 2 * F, dn_in, F, part, nullptr, none, st, 2));
                }
                if (glue)
                    return score_glue_fwd(us, B, D, rext, nullptr, nullptr, nullptr, make_glue(us, B, tp->logit, glue), st,
                                          F + 4, rext + F);
                return score_fwd(us, B, D, rext, nullptr, nullptr, nullptr, tp->logit, st, F + 4, rext + F);
            }
            if (rc != SF_ERR_UNSUPPORTED) return rc;
        }
        // (shapes outside the instantiations: the four-launch chain below)
    }
    if ((paired || query_only || last_step) && tf && !w->fold && !ctx_row && !d_h.on() && tw->w_out && w->action.w_a_t) {
        // Folded text stage (inference; sf_attention.hip: text_fold_body): FOUR dependent launches behind the cell
        // instead of six:
        //   (1) folded text attention (4 groups per sample, merged by the last arriver)  ||  y = W_out[:, H:] h1  ||  t_v' = W_h h1 + b_h
        //   (2) t_a = W_h tanh(z + y) + b_h, wt = t_a * w_out (A-prologue)   ||  q' = W_v^T t_v'
        //   (3) r = W_a^T wt            ||  visual attention of step t+1 (partials, ticket, merge by the last arriver)
        //   (4) scoring + glue
        // (query_only -- a device-resident environment: the panorama of step t+1 is not known yet -- stage (3) is the
        // r product alone and the attention follows the environment step, sf_attn_decoder_attend_fwd)
        const PanoSrc xn = paired ? pano(X_next) : PanoSrc{};
        const Dropout dn_in = make_dropout(drop, 2 * (step_id + 1), 2);
        Arena af = ar;                                           // (released when this branch is left)
        float* tpart = af.take(text_fold_part_floats(B, H));
        // (leading dimension H + 16: rows of a power-of-two stride share a few cache sets, CHANGELOG round 5 "2c")
        const int ldp = H + 16;
        float* ybuf = af.take((size_t)B * ldp);
        float* zbuf = af.take((size_t)B * ldp);
        unsigned* tcount = af.tickets() ? af.tickets() + TEXT_TICKET : nullptr;
        float* part = (paired && B <= 1024) ? af.take(visual_attn_split_floats(B, F)) : nullptr;
        SmallPlan py, pv, pta, pq, pr;
        bool ok = tpart && ybuf && zbuf && (part || !paired) && tcount && B <= 256 &&
            plan_linear(tp->cat2 + H, 2 * H, tw->w_out + H, 2 * H, nullptr, B, H, H, EPI_NONE, ybuf, ldp, &py) == SF_OK &&
            plan_linear(tp->wt, D, w->action.w_a_t, D, nullptr, B, F, D, EPI_NONE, tp->r, F, &pr) == SF_OK;
        if (ok && !last_step) {
            ok = plan_linear(tp->h1, H, vw->w_h, H, vw->b_h, B, D, H, EPI_NONE, tn->t_v, D, &pv) == SF_OK &&
                 plan_linear(tn->t_v, D, vw->w_v_t, D, nullptr, B, F, D, EPI_NONE, tn->q, F, &pq) == SF_OK;
        } else if (ok) {                                       // no next step: the second bodies of (1) and (2) are empty
            pv = py;
            pv.gx = pv.gy = 0;
            pq = pr;
            pq.gx = pq.gy = 0;
        }
        if (ok) {
            Seg sg{ybuf, ldp, w->action.w_h, H, H};
            LinearOut o{};
            o.y = tp->wt; o.ldy = D; o.bias = w->action.b_h; o.mul = w->action.w_out; o.y_pre = tp->t_a;
            o.ldy_pre = D; o.epi = EPI_MUL;
            ok = linear_small_plan(&sg, 1, B, D, o, &pta) && pta.mt == 1 && pta.cpw == 4 && pq.cpw == 2 && pr.cpw == 2;
            pta.args.apro_part = zbuf;                            // (the merged attention sum of launch (1))
            pta.args.apro_stride = ldp;
        }
        if (ok) {
            const int rc = pair_textfold_small_small(tf->ctx_q, tf->ctx_o, ctx_mask, B, L, H, tp->cat2 + H, 2 * H, tpart,
                                                     tcount, zbuf, ldp, tp->alpha, py, pv, st);
            if (rc == SF_OK) {
                TRY(pair_apro_small(pta, pq, st));
                if (paired && glue && g_fold_merge_with_glue) {
                    // partials beside r, their merge beside the scoring + glue launch (same depth, no in-launch ticket)
                    TRY(pair_vis_small(xn, B, tn->q, F, tn->alpha_v, tn->xin + F, 2 * F, dn_in, F, part, nullptr, pr, st, 1));
                    const int rc2 = pair_score_merge(us, B, D, tp->r, tp->wt, w->action.b_a, w->action.b_out,
                                                     make_glue(us, B, tp->logit, glue), xn, tn->alpha_v, tn->xin + F, 2 * F,
                                                     dn_in, F, part, st);
                    return rc2;                 // (its shape limits are score_glue_fwd's and pair_vis_small's)
                } else if (paired) {
                    TRY(pair_vis_small(xn, B, tn->q, F, tn->alpha_v, tn->xin + F, 2 * F, dn_in, F, part, af.tickets(), pr, st, 0));
                } else {
                    TRY(launch_small_plan_x(pr, st));
                }
                if (glue)
                    return score_glue_fwd(us, B, D, tp->r, tp->wt, w->action.b_a, w->action.b_out,
                                          make_glue(us, B, tp->logit, glue), st);
                return score_fwd(us, B, D, tp->r, tp->wt, w->action.b_a, w->action.b_out, tp->logit, st);
            }
            if (rc != SF_ERR_UNSUPPORTED) return rc;
        }
        // (shapes outside the instantiations: the unfolded stages below)
    }
    if (paired && w->fold) {
        // Folded inference step (sf_decoder_fold): two dependent stages fewer, and the attention
        // partials of step t+1 ride beside the text attention (the longest small stage) instead of
        // stretching the h~ product:
        //   (1) t_text = W_in h1                  ||  q' = M_v h1 + c_v
        //   (2) text attention                    ||  visual-attention partials of step t+1
        //   (3) h~ = tanh(W_out [wc ; h1])        ||  merge of the partials
        //   (4) [r | c] = M_a h~ + c_a            (5) scoring + glue
        const PanoSrc xn = pano(X_next);
        const Dropout dn_in = make_dropout(drop, 2 * (step_id + 1), 2);
        SmallPlan pa, pb;
        const bool ok1 =
            plan_linear(tp->cat2 + H, 2 * H, tw->w_in, H, nullptr, B, H, H, EPI_NONE, tp->t_text, H, &pa) == SF_OK &&
            plan_linear(tp->h1, H, w->fold->m_v, H, w->fold->c_v, B, F, H, EPI_NONE, tn->q, F, &pb) == SF_OK;
        if (!ok1 || pair_small_small(pa, pb, st) != SF_OK) {
            TRY(linear_plain(tp->cat2 + H, 2 * H, tw->w_in, H, nullptr, B, H, H, EPI_NONE, tp->t_text, H, ar, st));
            TRY(linear_plain(tp->h1, H, w->fold->m_v, H, w->fold->c_v, B, F, H, EPI_NONE, tn->q, F, ar, st));
        }
        float* part = B <= 1024 ? ar.take(visual_attn_split_floats(B, F)) : nullptr;
        const bool ok3 = part && plan_linear(tp->cat2, 2 * H, tw->w_out, 2 * H, nullptr, B, H, 2 * H, EPI_TANH,
                                             tp->h_tilde, H, &pb) == SF_OK;
        bool split = false;
        if (ok3) {
            const int rc = pair_vis_text(xn, B, tn->q, F, tn->alpha_v, tn->xin + F, 2 * F, dn_in, F, part, ctx,
                                         ctx_mask, L, H, tp->t_text, H, tp->alpha, tp->cat2, 2 * H, ctx_row, st);
            if (rc == SF_OK) split = true;
            else if (rc != SF_ERR_UNSUPPORTED) return rc;
        }
        if (split && pair_vis_small(xn, B, tn->q, F, tn->alpha_v, tn->xin + F, 2 * F, dn_in, F, part, nullptr, pb,
                                    st, 2) != SF_OK)
            return SF_ERR_LAUNCH;            // the partials exist: the merge must not be skipped
        if (!split) {
            TRY(text_attn_fwd(ctx, ctx_mask, B, L, H, tp->t_text, H, tp->alpha, tp->cat2, 2 * H, st, ctx_row));
            TRY(linear_plain(tp->cat2, 2 * H, tw->w_out, 2 * H, nullptr, B, H, 2 * H, EPI_TANH, tp->h_tilde, H, ar, st));
            TRY(visual_attn(0, xn, B, tn->q, F, tn->alpha_v, tn->xin + F, 2 * F, dn_in, F, st, part,
                            part ? ar.tickets() : nullptr));
        }
    } else if (paired) {
        const PanoSrc xn = pano(X_next);
        const Dropout dn_in = make_dropout(drop, 2 * (step_id + 1), 2);
        SmallPlan pa, pb;
        // (1) t_text = W_in dropout(h1)   ||   t_v' = W_h h1 + b_h
        const bool ok1 =
            plan_linear(tp->cat2 + H, 2 * H, tw->w_in, H, nullptr, B, H, H, EPI_NONE, tp->t_text, H, &pa) == SF_OK &&
            plan_linear(tp->h1, H, vw->w_h, H, vw->b_h, B, D, H, EPI_NONE, tn->t_v, D, &pb) == SF_OK;
        if (!ok1 || pair_small_small(pa, pb, st) != SF_OK) {
            TRY(linear_plain(tp->cat2 + H, 2 * H, tw->w_in, H, nullptr, B, H, H, EPI_NONE, tp->t_text, H, ar, st));
            TRY(linear_plain(tp->h1, H, vw->w_h, H, vw->b_h, B, D, H, EPI_NONE, tn->t_v, D, ar, st));
        }
        // (2) text attention   ||   q' = W_v^T t_v'
        const bool ok2 = plan_linear(tn->t_v, D, vw->w_v_t, D, nullptr, B, F, D, EPI_NONE, tn->q, F, &pa) == SF_OK;
        if (!ok2 || pair_small_text(pa, ctx, ctx_mask, B, L, H, tp->t_text, H, tp->alpha, tp->cat2,
                                    2 * H, ctx_row, st) != SF_OK) {
            TRY(text_attn_fwd(ctx, ctx_mask, B, L, H, tp->t_text, H, tp->alpha, tp->cat2, 2 * H, st, ctx_row));
            TRY(linear_plain(tn->t_v, D, vw->w_v_t, D, nullptr, B, F, D, EPI_NONE, tn->q, F, ar, st));
        }
        // (3) h~ = tanh(W_out [wc ; h1])   ||   visual attention of step t+1, per-group partials
        // (4) t_a = W_h h~ + b_h, wt = t_a * w_out   ||   ... merge of the partials
        // (the attention is not needed before the next gate product: its two halves ride with two
        // stages of the text / scoring chain instead of stretching one of them)
        float* part = B <= 1024 ? ar.take(visual_attn_split_floats(B, F)) : nullptr;
        const bool ok3 = part && plan_linear(tp->cat2, 2 * H, tw->w_out, 2 * H, nullptr, B, H, 2 * H,
                                             EPI_TANH, tp->h_tilde, H, &pb) == SF_OK;
        SmallPlan pc;
        bool ok4 = false;
        if (ok3 && !w->fold) {
            Seg sg{tp->h_tilde, H, w->action.w_h, H, H};
            LinearOut o{};
            o.y = tp->wt; o.ldy = D; o.bias = w->action.b_h; o.mul = w->action.w_out; o.y_pre = tp->t_a;
            o.ldy_pre = D; o.epi = EPI_MUL;
            ok4 = linear_small_plan(&sg, 1, B, D, o, &pc) && pc.mt == 1 && (pc.cpw == 4 || pc.cpw == 8);
        }
        if (ok4 && pair_vis_small(xn, B, tn->q, F, tn->alpha_v, tn->xin + F, 2 * F, dn_in, F, part,
                                  nullptr, pb, st, 1) == SF_OK) {
            TRY(pair_vis_small(xn, B, tn->q, F, tn->alpha_v, tn->xin + F, 2 * F, dn_in, F, part, nullptr,
                               pc, st, 2));
            return scoring_fwd_i(&w->action, us, B, H, D, tp->h_tilde, tp->logit, tp->t_a, tp->wt, tp->r,
                                 ar, st, glue, nullptr, true);
        }
        if (!ok3 || pair_vis_small(xn, B, tn->q, F, tn->alpha_v, tn->xin + F, 2 * F, dn_in, F, part,
                                   ar.tickets(), pb, st) != SF_OK) {
            TRY(linear_plain(tp->cat2, 2 * H, tw->w_out, 2 * H, nullptr, B, H, 2 * H, EPI_TANH, tp->h_tilde, H, ar, st));
            TRY(visual_attn(0, xn, B, tn->q, F, tn->alpha_v, tn->xin + F, 2 * F, dn_in, F, st, part,
                            part ? ar.tickets() : nullptr));
        }
    } else if (query_only) {
        SmallPlan pa, pb;
        const bool ok1 =
            plan_linear(tp->cat2 + H, 2 * H, tw->w_in, H, nullptr, B, H, H, EPI_NONE, tp->t_text, H, &pa) == SF_OK &&
            plan_linear(tp->h1, H, vw->w_h, H, vw->b_h, B, D, H, EPI_NONE, tn->t_v, D, &pb) == SF_OK;
        if (!ok1 || pair_small_small(pa, pb, st) != SF_OK) {
            TRY(linear_plain(tp->cat2 + H, 2 * H, tw->w_in, H, nullptr, B, H, H, EPI_NONE, tp->t_text, H, ar, st));
            TRY(linear_plain(tp->h1, H, vw->w_h, H, vw->b_h, B, D, H, EPI_NONE, tn->t_v, D, ar, st));
        }
        const bool ok2 = plan_linear(tn->t_v, D, vw->w_v_t, D, nullptr, B, F, D, EPI_NONE, tn->q, F, &pa) == SF_OK;
        if (!ok2 || pair_small_text(pa, ctx, ctx_mask, B, L, H, tp->t_text, H, tp->alpha, tp->cat2,
                                    2 * H, ctx_row, st) != SF_OK) {
            TRY(text_attn_fwd(ctx, ctx_mask, B, L, H, tp->t_text, H, tp->alpha, tp->cat2, 2 * H, st, ctx_row));
            TRY(linear_plain(tn->t_v, D, vw->w_v_t, D, nullptr, B, F, D, EPI_NONE, tn->q, F, ar, st));
        }
        TRY(linear_plain(tp->cat2, 2 * H, tw->w_out, 2 * H, nullptr, B, H, 2 * H, EPI_TANH, tp->h_tilde, H, ar, st));
    } else {
        TRY(softdot_fwd_i(tw, B, L, H, nullptr, 0, ctx, ctx_mask, tp->h_tilde, tp->alpha, tp->cat2,
                          tp->t_text, ar, st, ctx_row));
        if (X_next)
            TRY(visual_fwd_i(vw, pano(X_next), B, H, D, tp->h1, tn->xin + F, 2 * F, tn->alpha_v,
                             tn->t_v, tn->q, make_dropout(drop, 2 * (step_id + 1), 2), F, ar, st, w->fold));
    }
    return scoring_fwd_i(&w->action, us, B, H, D, tp->h_tilde, tp->logit, tp->t_a, tp->wt, tp->r, ar,
                         st, glue, w->fold);
}

// tail(t) WITHOUT the head of step t+1 (that runs on another stream): LSTM cell, then -- after
// publishing "h1 of this step is complete" -- the text attention and scoring chain, unpaired.
static int decoder_tail_split_i(const sf_decoder_w* w, const sf_cands* U, int B, int H, int D, int L,
                                const float* h0, const float* c0, const float* ctx, const uint8_t* ctx_mask,
                                const sf_decoder_tape* tp, const sf_follower_glue* glue, const sf_dropout* drop,
                                uint32_t step_id, unsigned* flag_h1, unsigned flag_value, void* ws,
                                size_t ws_bytes, unsigned* tickets, hipStream_t st) {
    SF_CHECK_ARG(w && U && h0 && c0 && ctx && tp && glue && glue_ok(U, glue));
    const size_t n = ws_bytes / 4;
    Arena ar{(float*)ws, n, 0, tickets};
    const CandSrc us = cands(U);
    const int F = us.IMG + us.LOC;
    const Dropout d_h = make_dropout(drop, 2 * step_id + 1, 2);
    TRY(lstm_fwd_i(&w->lstm, B, 2 * F, H, tp->xin, 2 * F, h0, c0, tp->h1, tp->c1, tp->gates, tp->cat2 + H, 2 * H,
                   d_h, ar, st));
    if (flag_h1) TRY(flag_set(flag_h1, flag_value, st));
    TRY(softdot_fwd_i(&w->text, B, L, H, nullptr, 0, ctx, ctx_mask, tp->h_tilde, tp->alpha, tp->cat2, tp->t_text,
                      ar, st, nullptr));
    return scoring_fwd_i(&w->action, us, B, H, D, tp->h_tilde, tp->logit, tp->t_a, tp->wt, tp->r, ar, st, glue,
                         nullptr);
}

int sf_attn_decoder_attend_fwd(const sf_pano* X, int B, const sf_decoder_tape* tp, const sf_dropout* drop,
                               uint32_t step_id, void* ws, size_t ws_bytes, sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(X && tp && tp->q && tp->xin && tp->alpha_v && B > 0);
    Arena ar = arena(ws, ws_bytes);
    const PanoSrc xs = pano(X);
    const int F = xs.IMG + xs.LOC;
    float* part = B <= 1024 ? ar.take(visual_attn_split_floats(B, F)) : nullptr;
    return visual_attn(0, xs, B, tp->q, F, tp->alpha_v, tp->xin + F, 2 * F, make_dropout(drop, 2 * step_id, 2), F,
                       S(stream), part, part ? ar.tickets() : nullptr);
}

int sf_attn_decoder_tail_fwd(const sf_decoder_w* w, const sf_cands* U, int B, int H, int D, int L,
                             const float* u_prev, const float* h0, const float* c0,
                             const float* ctx, const uint8_t* ctx_mask, const int32_t* ctx_row,
                             const sf_decoder_tape* tp, const sf_follower_glue* glue,
                             const sf_dropout* drop, uint32_t step_id, const sf_pano* X_next,
                             const sf_decoder_tape* tn, void* ws, size_t ws_bytes,
                             sf_stream stream) {
    SF_ENTER();
    return decoder_tail_i(w, U, B, H, D, L, u_prev, h0, c0, ctx, ctx_mask, ctx_row, tp, glue, drop,
                          step_id, X_next, tn, ws, ws_bytes, stream);
}

int sf_decoder_fold_build(const sf_decoder_w* w, int H, int D, int F, float* m_v, float* c_v,
                          float* m_a, float* c_a, void* ws, size_t ws_bytes, sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(w && m_v && c_v && m_a && c_a && w->visual.w_v_t && w->visual.w_h_t &&
                 w->action.w_a_t && w->action.w_h_t && F % 4 == 0 && D % 4 == 0 && H % 4 == 0);
    Arena ar = arena(ws, ws_bytes);
    hipStream_t st = S(stream);
    float* wa_s = ar.take((size_t)F * D);      // W_a^T with columns scaled by w_out
    float* wo_ba = ar.take(D);                 // w_out * b_a
    float* wo_bh = ar.take(D);                 // w_out * b_h
    NEED(wa_s && wo_ba && wo_bh);
    // visual: M_v = W_v^T W_h  ([F,D] x [D,H]),  c_v = W_v^T b_h
    TRY(linear_plain(w->visual.w_v_t, D, w->visual.w_h_t, D, nullptr, F, H, D, EPI_NONE, m_v, H, ar, st));
    TRY(linear_plain(w->visual.b_h, D, w->visual.w_v_t, D, nullptr, 1, F, D, EPI_NONE, c_v, F, ar, st));
    // scoring
    TRY(scale_cols(w->action.w_a_t, D, w->action.w_out, F, D, wa_s, D, st));
    TRY(scale_cols(w->action.b_a, D, w->action.w_out, 1, D, wo_ba, D, st));
    TRY(scale_cols(w->action.b_h, D, w->action.w_out, 1, D, wo_bh, D, st));
    TRY(fill(m_a + (size_t)F * H, (size_t)4 * H, 0.f, st));
    TRY(fill(c_a + F, 4, 0.f, st));
    TRY(linear_plain(wa_s, D, w->action.w_h_t, D, nullptr, F, H, D, EPI_NONE, m_a, H, ar, st));
    TRY(linear_plain(wo_ba, D, w->action.w_h_t, D, nullptr, 1, H, D, EPI_NONE, m_a + (size_t)F * H, H, ar, st));
    TRY(linear_plain(wo_bh, D, w->action.w_a_t, D, nullptr, 1, F, D, EPI_NONE, c_a, F + 4, ar, st));
    return linear_plain(wo_bh, D, w->action.b_a, D, w->action.b_out, 1, 1, D, EPI_NONE, c_a + F, 4, ar, st);
}

// The backward of one decode step is two chains that only meet at the LSTM pointwise backward:
//   head: scoring -> h~ -> text attention  =>  dh1d (the attention path's share of d h1), and the dY
//         operands of that half; needs only the forward tape and d(logit) of ITS step;
//   tail: LSTM cell -> input gradient -> visual attention  =>  dh0, dc0; needs dh1 / dc1 of step t+1.
// So head(t-1) can run while tail(t) does (sf_follower_episode_bwd puts the heads on a side stream).
static int decoder_bwd_head_i(const sf_decoder_w* w, const sf_decoder_g* g, const sf_cands* U, int B,
                              int H, int D, int L, const float* ctx, const sf_decoder_tape* tp,
                              const sf_decoder_gtape* gt, const float* dlogit, float* dh1d, float* dctx,
                              Arena ar, hipStream_t st, const CeSrc* ce) {
    float* dht = ar.take((size_t)B * H);       // d h_tilde
    NEED(dht && dh1d);
    // with a gradient tape the scoring backward writes d(pre-tanh) straight into gt->dpre
    bool dpre_ready = false;
    float* dht_out = gt ? gt->dpre : dht;
    TRY(scoring_bwd_i(&w->action, g ? &g->action : nullptr, cands(U), B, H, D, tp->h_tilde, tp->t_a,
                      tp->wt, dlogit, dht_out, ar, st, gt, gt ? tp->h_tilde : nullptr, &dpre_ready, ce));
    if (gt && !dpre_ready) {      // not fused: dht_out holds d h~; keep it apart from gt->dpre
        TRY(add2(dht_out, H, nullptr, 0, B, H, dht, H, st));
        dht_out = dht;
    }
    return softdot_bwd_i(&w->text, g ? &g->text : nullptr, B, L, H, ctx, tp->alpha, tp->cat2, tp->t_text,
                         tp->h_tilde, dht_out, dh1d, H, dctx, ar, st, gt ? gt->dpre : nullptr,
                         gt ? gt->dt_text : nullptr, dpre_ready, gt ? gt->dcat2 : nullptr,
                         gt ? gt->ds : nullptr);
}

static int decoder_bwd_tail_i(const sf_decoder_w* w, const sf_decoder_g* g, const sf_pano* X, int B,
                              int H, int D, const float* h0, const float* c0,
                              const sf_decoder_tape* tp, const sf_decoder_gtape* gt, const float* dh1,
                              const float* dh1d, const float* dc1, float* dh0, float* dc0,
                              const sf_dropout* drop, uint32_t step_id, Arena ar, hipStream_t st,
                              const LstmPwBwd* next_pw = nullptr, bool* next_fused = nullptr,   // see visual_bwd_i
                              bool pointwise_done = false) {        // this step's dgates / dc0 exist already
    const PanoSrc xs = pano(X);
    const int F = xs.IMG + xs.LOC;
    const Dropout d_in = make_dropout(drop, 2 * step_id, 2), d_h = make_dropout(drop, 2 * step_id + 1, 2);
    float* dxin = ar.take((size_t)B * 2 * F);  // d LSTM input
    NEED(dxin);
    SmallPlan dh0_plan;
    bool dh0_deferred = false;
    const float* df_slabs = nullptr;
    int df_ks = 0;
    // (the dropout between h1 and the text attention is undone inside the LSTM pointwise backward)
    TRY(lstm_bwd_i(&w->lstm, g ? &g->lstm : nullptr, B, 2 * F, H, tp->xin, 2 * F, h0, c0, tp->c1,
                   tp->gates, dh1, dh1d, dc1, dxin, 2 * F, dh0, dc0, ar, st,
                   gt ? gt->dgates : nullptr, F, &d_h,   // u_prev is detached (follower.py:502): only
                   &dh0_plan, &dh0_deferred,            // the feature half of d(LSTM input) is needed
                   &df_slabs, &df_ks, pointwise_done));
    // dh0 = dgates W_hh rides beside the visual-attention backward (both only need the LSTM backward); the feature
    // half of d(LSTM input) reaches it as the K-split slabs of its product (no reduce launch in between)
    if (df_ks >= 1)
        return visual_bwd_i(&w->visual, g ? &g->visual : nullptr, xs, B, H, D, h0, tp->alpha_v, tp->t_v,
                            df_slabs, F, d_in, F, dh0, ar, st, gt ? gt->dq : nullptr,
                            gt ? gt->dt_v : nullptr, dh0_deferred ? &dh0_plan : nullptr, df_ks, (long)B * F, next_pw,
                            next_fused);
    return visual_bwd_i(&w->visual, g ? &g->visual : nullptr, xs, B, H, D, h0, tp->alpha_v, tp->t_v,
                        dxin + F, 2 * F, d_in, F, dh0, ar, st, gt ? gt->dq : nullptr,
                        gt ? gt->dt_v : nullptr, dh0_deferred ? &dh0_plan : nullptr, 0, 0, next_pw, next_fused);
}

static int decoder_bwd_i(const sf_decoder_w* w, const sf_decoder_g* g, const sf_pano* X,
                         const sf_cands* U, int B, int H, int D, int L, const float* h0,
                         const float* c0, const float* ctx, const sf_decoder_tape* tp,
                         const sf_decoder_gtape* gt, const float* dlogit, const float* dh1,
                         const float* dc1, float* dh0, float* dc0, float* dctx,
                         const sf_dropout* drop, uint32_t step_id, void* ws, size_t ws_bytes,
                         sf_stream stream, const CeSrc* ce = nullptr) {
    SF_CHECK_ARG(w && X && U && h0 && c0 && ctx && tp && dlogit && dh0 && dc0 && B > 0 && L > 0);
    Arena ar = arena(ws, ws_bytes);
    float* dh1d = ar.take((size_t)B * H);      // d dropout(h1)
    NEED(dh1d);
    TRY(decoder_bwd_head_i(w, g, U, B, H, D, L, ctx, tp, gt, dlogit, dh1d, dctx, ar, S(stream), ce));
    return decoder_bwd_tail_i(w, g, X, B, H, D, h0, c0, tp, gt, dh1, dh1d, dc1, dh0, dc0, drop, step_id,
                              ar, S(stream));
}

int sf_attn_decoder_bwd(const sf_decoder_w* w, const sf_decoder_g* g, const sf_pano* X,
                        const sf_cands* U, int B, int H, int D, int L, const float* h0,
                        const float* c0, const float* ctx, const sf_decoder_tape* tp,
                        const sf_decoder_gtape* gt, const float* dlogit, const float* dh1,
                        const float* dc1, float* dh0, float* dc0, float* dctx,
                        const sf_dropout* drop, uint32_t step_id, void* ws, size_t ws_bytes,
                        sf_stream stream) {
    SF_ENTER();
    return decoder_bwd_i(w, g, X, U, B, H, D, L, h0, c0, ctx, tp, gt, dlogit, dh1, dc1, dh0, dc0, dctx,
                         drop, step_id, ws, ws_bytes, stream);
}

// ---- a whole follower episode (follower.py:430-539 / 1001-1020) in one call ----------------------------
// Every per-step tensor of an index-form episode is a stacked [S][...] array, so the decode loop and
// its backward need no host work between steps: the host mirror makes ONE call instead of 2 S (its
// Python + ctypes cost per step was exposed as idle gaps between the short kernels of the backward).
extern "C++" {
namespace {
// events of the two-stream episode backward (created once per host thread, timing disabled)
// (keyed by device: an event belongs to the device that was current when it was created)
std::vector<hipEvent_t>& event_pool(size_t n) {
    static thread_local std::map<int, std::vector<hipEvent_t>> pools;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::vector<hipEvent_t>& pool = pools[dev];
    while (pool.size() < n) {
        hipEvent_t e;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) break;
        pool.push_back(e);
    }
    return pool;
}

struct StepView {
    sf_pano X;
    sf_cands U;
    sf_decoder_tape tp;
    sf_follower_glue glue;
};

template <class T>
inline T* adv(T* p, size_t n) { return p ? p + n : nullptr; }

StepView step_view(const sf_follower_episode* e, int t) {
    const size_t B = e->B, H = e->H, D = e->D, L = e->L, A = e->A;
    const size_t F = e->X.IMG + e->X.LOC, V = e->X.V;
    StepView v;
    v.X = e->X;
    v.X.vp = adv(e->X.vp, t * B);
    v.X.view = adv(e->X.view, t * B);
    v.X.dense = adv(e->X.dense, t * B * V * F);
    v.U = e->U;
    v.U.vp = adv(e->U.vp, t * B);
    v.U.cand_view = adv(e->U.cand_view, t * B * A);
    v.U.cand_sincos = adv(e->U.cand_sincos, t * B * A * 4);
    v.U.a_num = adv(e->U.a_num, t * B);
    v.U.dense = adv(e->U.dense, t * B * A * F);
    const sf_decoder_tape& p = e->tape;
    v.tp.t_v = adv(p.t_v, t * B * D);
    v.tp.q = adv(p.q, t * B * F);
    v.tp.alpha_v = adv(p.alpha_v, t * B * V);
    v.tp.xin = adv(p.xin, t * B * 2 * F);
    v.tp.gates = adv(p.gates, t * B * 4 * H);
    v.tp.c1 = adv(p.c1, t * B * H);
    v.tp.h1 = adv(p.h1, t * B * H);
    v.tp.cat2 = adv(p.cat2, t * B * 2 * H);
    v.tp.t_text = adv(p.t_text, t * B * H);
    v.tp.alpha = adv(p.alpha, t * B * L);
    v.tp.h_tilde = adv(p.h_tilde, t * B * H);
    v.tp.t_a = adv(p.t_a, t * B * D);
    v.tp.wt = adv(p.wt, t * B * D);
    v.tp.r = adv(p.r, t * B * F);
    v.tp.logit = adv(p.logit, t * B * A);
    v.glue = e->glue;
    v.glue.is_valid = adv(e->glue.is_valid, t * B * A);
    v.glue.target = adv(e->glue.target, t * B);
    v.glue.a_t = adv(e->glue.a_t, t * B);
    v.glue.target_used = adv(e->glue.target_used, t * B);
    v.glue.score = adv(e->glue.score, t * B);
    v.glue.ce_term = adv(e->glue.ce_term, t * B);
    v.glue.live = adv(e->glue.live, t * B);
    v.glue.u_next = adv(p.xin, (t + 1) * B * 2 * F);        // straight into the next step's LSTM input
    v.glue.ld_u_next = (int32_t)(2 * F);
    v.glue.u_drop = e->drop.p > 0.f ? &e->drop : nullptr;
    v.glue.u_drop_stream = 2 * (e->step0 + t + 1);
    v.glue.sample_stream = e->step0 + t;
    return v;
}

// step t of a device-resident environment: state slot t -> slot t + 1 of the stacked [S + 1][...] buffers
sf_nav_io nav_view(const sf_nav_io* n, const sf_follower_episode* e, int t) {
    const size_t B = e->B, A = e->A;
    sf_nav_io v = *n;
    v.row = adv(n->row, t * B);
    v.view = adv(n->view, t * B);
    v.row_next = adv(n->row_next, t * B);
    v.vp_next = adv(n->vp_next, t * B);
    v.view_next = adv(n->view_next, t * B);
    v.a_num_next = adv(n->a_num_next, t * B);
    v.cand_view_next = adv(n->cand_view_next, t * B * A);
    v.sincos_next = adv(n->sincos_next, t * B * A * 4);
    v.target_next = adv(n->target_next, t * B);
    return v;
}

sf_decoder_gtape gtape_view(const sf_decoder_gtape* g, const sf_follower_episode* e, int t) {
    const size_t B = e->B, H = e->H, D = e->D, F = e->X.IMG + e->X.LOC;
    sf_decoder_gtape v;
    v.dgates = adv(g->dgates, t * B * 4 * H);
    v.dpre = adv(g->dpre, t * B * H);
    v.dt_text = adv(g->dt_text, t * B * H);
    v.dt_v = adv(g->dt_v, t * B * D);
    v.dq = adv(g->dq, t * B * F);
    v.dwt = adv(g->dwt, t * B * D);
    v.dta = adv(g->dta, t * B * D);
    v.dr = adv(g->dr, t * B * F);
    v.dc = adv(g->dc, t * B);
    const bool defer = g->dcat2 && g->ds;
    v.dcat2 = defer ? adv(g->dcat2, t * B * 2 * H) : nullptr;
    v.ds = defer ? adv(g->ds, t * B * e->L) : nullptr;
    v.dh1d = adv(g->dh1d, t * B * H);
    return v;
}
}  // namespace
}  // extern "C++"

int sf_follower_episode_fwd(const sf_decoder_w* w, const sf_follower_episode* e, void* ws,
                            size_t ws_bytes, sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(w && e && e->S > 0 && e->B > 0 && e->h_init && e->c_init && e->ctx && e->tape.xin &&
                 e->tape.h1 && e->tape.c1 && e->glue.target && e->glue.ended);
    const sf_dropout* drop = e->drop.p > 0.f ? &e->drop : nullptr;
    const size_t BH = (size_t)e->B * e->H;
    StepView cur = step_view(e, 0);
    hipEvent_t fold_join = nullptr;
    hipStream_t fold_side = nullptr;
    // the folded text attention (ABI 9): ctx_q = ctx W_in, ctx_o = ctx W_out[:, :H]^T, once per episode
    TextFold tfold{e->ctx_q, e->ctx_o, e->chain_fold};
    const TextFold* tf = nullptr;
    if (e->ctx_q && e->ctx_o && !drop && !w->fold && e->S > 1 && w->text.w_in_t && w->text.w_out && w->action.w_a_t) {
        const int M = e->B * e->L, H = e->H;
        Arena ar = arena(ws, ws_bytes);
        hipStream_t ms = S(stream), bs = ms;
        std::vector<hipEvent_t>* ev = nullptr;
        if (e->side_stream && e->side_stream != stream && g_fold_build_overlap) {
            // the two products need nothing but the encoder's context: on the caller's SECOND stream beside step 0's
            // attention (one fork here, one join in front of the first decode step; the last quarter of the workspace is
            // theirs, the head keeps the front)
            ev = &event_pool(2);
            if (ev->size() >= 2 && ar.cap > ((size_t)8 << 20)) {
                bs = S(e->side_stream);
                const size_t side_n = ar.cap / 4;
                ar = Arena{(float*)ws + (ar.cap - side_n), side_n, 0, ar.tk};
                if (hipEventRecord((*ev)[0], ms) != hipSuccess || hipStreamWaitEvent(bs, (*ev)[0], 0) != hipSuccess)
                    return SF_ERR_LAUNCH;
            }
        }
        TRY(linear_plain(e->ctx, H, w->text.w_in_t, H, nullptr, M, H, H, EPI_NONE, e->ctx_q, H, ar, bs));
        TRY(linear_plain(e->ctx, H, w->text.w_out, 2 * H, nullptr, M, H, H, EPI_NONE, e->ctx_o, H, ar, bs));
        tf = &tfold;
        if (bs != ms) {
            // (the head -- issued below on `stream` -- runs beside them; join behind it)
            fold_join = (*ev)[1];
            fold_side = bs;
        }
    }
    {
        // (beside the fold products the head keeps the FRONT three quarters of the workspace -- and the same ticket words)
        Arena ha = arena(ws, ws_bytes);
        if (fold_side) ha.cap -= ha.cap / 4;
        TRY(decoder_head_a(w, &cur.X, e->B, e->H, e->D, e->h_init, &cur.tp, drop, e->step0, ha, stream));
    }
    if (fold_side) {
        if (hipEventRecord(fold_join, fold_side) != hipSuccess || hipStreamWaitEvent(S(stream), fold_join, 0) != hipSuccess)
            return SF_ERR_LAUNCH;
    }
    if (e->glue.nav) {
        // A device-resident environment (sf_nav_io of step 0 in glue.nav; state buffers stacked [S + 1][...]): the
        // panorama of step t + 1 is only known once the glue of step t has chosen its action and stepped the
        // environment (inside the scoring + glue launch), so its attention cannot ride beside tail(t); its QUERY can
        // (tape_next without X_next), and the attention follows as its own launch -- the schedule the host loop of
        // FollowerEngine.rollout issues call by call, here without host work between the launches.
        SF_CHECK_ARG(!w->fold && w->visual.w_v_t);
        for (int t = 0; t < e->S; ++t) {
            const bool more = t + 1 < e->S;
            StepView nxt = more ? step_view(e, t + 1) : cur;
            const sf_nav_io nv = nav_view(e->glue.nav, e, t);
            cur.glue.nav = &nv;
            const float* h0 = t == 0 ? e->h_init : e->tape.h1 + (size_t)(t - 1) * BH;
            const float* c0 = t == 0 ? e->c_init : e->tape.c1 + (size_t)(t - 1) * BH;
            TRY(decoder_tail_i(w, &cur.U, e->B, e->H, e->D, e->L, nullptr, h0, c0, e->ctx, e->ctx_mask,
                               nullptr, &cur.tp, &cur.glue, drop, e->step0 + t, nullptr, more ? &nxt.tp : nullptr, ws,
                               ws_bytes, stream, tf));
            if (more)
                TRY(sf_attn_decoder_attend_fwd(&nxt.X, e->B, &nxt.tp, drop, e->step0 + t + 1, ws, ws_bytes, stream));
            cur = nxt;
        }
        return SF_OK;
    }
    if (e->side_stream && e->side_stream != stream && !w->fold && e->S > 1 && !tf) {
        // Two chains, ONE fork and ONE join per episode, ordered per step by device flags
        // (flag_wait / flag_set kernels) instead of events:
        //   main:  [wait feat(t)] gate product, cell, [set h1(t)], t_text, text attention, h~, scoring + glue
        //   side:  [wait h1(t)]  t_v', q', visual attention of step t+1, [set feat(t+1)]
        hipStream_t ms = S(stream), ss = S(e->side_stream);
        std::vector<hipEvent_t>& ev = event_pool(2);
        if (ev.size() < 2) return SF_ERR_LAUNCH;
        const Arena whole = arena(ws, ws_bytes);
        if (!whole.tk) return SF_ERR_WORKSPACE;
        const size_t side_n = std::min<size_t>(whole.cap / 4, (size_t)4 << 20);
        unsigned* flag_h1 = whole.tk + PERSIST_TICKET + 48;
        unsigned* flag_ft = whole.tk + PERSIST_TICKET + 49;
        void* side_ws = (float*)ws + (whole.cap - side_n);
        const size_t main_bytes = (whole.cap - side_n) * sizeof(float);
        TRY(flag_set(flag_h1, 0u, ms));
        TRY(flag_set(flag_ft, 0u, ms));
        if (hipEventRecord(ev[0], ms) != hipSuccess || hipStreamWaitEvent(ss, ev[0], 0) != hipSuccess) return SF_ERR_LAUNCH;
        for (int t = 0; t < e->S; ++t) {
            const bool more = t + 1 < e->S;
            StepView nxt = more ? step_view(e, t + 1) : cur;
            const float* h0 = t == 0 ? e->h_init : e->tape.h1 + (size_t)(t - 1) * BH;
            const float* c0 = t == 0 ? e->c_init : e->tape.c1 + (size_t)(t - 1) * BH;
            if (t > 0) TRY(flag_wait(flag_ft, (unsigned)t, ms));
            TRY(decoder_tail_split_i(w, &cur.U, e->B, e->H, e->D, e->L, h0, c0, e->ctx, e->ctx_mask, &cur.tp,
                                     &cur.glue, drop, e->step0 + t, more ? flag_h1 : nullptr, (unsigned)(t + 1), ws,
                                     main_bytes, whole.tk, ms));
            if (more) {
                TRY(flag_wait(flag_h1, (unsigned)(t + 1), ss));
                Arena sar{(float*)side_ws, side_n, 0, whole.tk};
                const PanoSrc xs = pano(&nxt.X);
                const int F = xs.IMG + xs.LOC;
                TRY(visual_fwd_i(&w->visual, xs, e->B, e->H, e->D, cur.tp.h1, nxt.tp.xin + F, 2 * F, nxt.tp.alpha_v,
                                 nxt.tp.t_v, nxt.tp.q, make_dropout(drop, 2 * (e->step0 + t + 1), 2), F, sar, ss, nullptr));
                TRY(flag_set(flag_ft, (unsigned)(t + 1), ss));
            }
            cur = nxt;
        }
        if (hipEventRecord(ev[1], ss) != hipSuccess || hipStreamWaitEvent(ms, ev[1], 0) != hipSuccess) return SF_ERR_LAUNCH;
        return SF_OK;
    }
    for (int t = 0; t < e->S; ++t) {
        const bool more = t + 1 < e->S;
        StepView nxt = more ? step_view(e, t + 1) : cur;
        const float* h0 = t == 0 ? e->h_init : e->tape.h1 + (size_t)(t - 1) * BH;
        const float* c0 = t == 0 ? e->c_init : e->tape.c1 + (size_t)(t - 1) * BH;
        TRY(decoder_tail_i(w, &cur.U, e->B, e->H, e->D, e->L, nullptr, h0, c0, e->ctx, e->ctx_mask,
                           nullptr, &cur.tp, &cur.glue, drop, e->step0 + t, more ? &nxt.X : nullptr,
                           more ? &nxt.tp : nullptr, ws, ws_bytes, stream, tf));
        cur = nxt;
    }
    return SF_OK;
}

int sf_follower_episode_bwd(const sf_decoder_w* w, const sf_follower_episode* e,
                            const sf_decoder_gtape* gtape, const float* gscale, float* dlogit,
                            float* dh_a, float* dc_a, float* dh_b, float* dc_b, float* dctx,
                            int* result_in_b, void* ws, size_t ws_bytes, sf_stream stream) {
    return sf_follower_episode_bwd_range(w, e, gtape, gscale, dlogit, dh_a, dc_a, dh_b, dc_b, dctx, result_in_b,
                                         0, e ? e->S : 0, nullptr, nullptr, ws, ws_bytes, stream);
}

int sf_follower_episode_bwd_range(const sf_decoder_w* w, const sf_follower_episode* e,
                                  const sf_decoder_gtape* gtape, const float* gscale, float* dlogit,
                                  float* dh_a, float* dc_a, float* dh_b, float* dc_b, float* dctx,
                                  int* result_in_b, int t_lo, int t_hi, const float* dh_in, const float* dc_in,
                                  void* ws, size_t ws_bytes, sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(w && e && gtape && gscale && dlogit && dh_a && dc_a && dh_b && dc_b && result_in_b &&
                 e->S > 0 && e->B > 0 && 0 <= t_lo && t_lo < t_hi && t_hi <= e->S &&
                 (t_hi == e->S || (dh_in && dc_in)) && (!dh_in) == (!dc_in));
    const sf_dropout* drop = e->drop.p > 0.f ? &e->drop : nullptr;
    const size_t BH = (size_t)e->B * e->H;
    sf_decoder_gtape gt_all = *gtape;          // the deferred context gradient only where it is covered
    if (!(gt_all.dcat2 && gt_all.ds && dctx && ctx_grad_supported(e->S, e->L, e->H)))
        gt_all.dcat2 = gt_all.ds = nullptr;
    gtape = &gt_all;
    const float *dh1 = dh_in, *dc1 = dc_in;           // gradient arriving from step t_hi (null: the last step)
    float *dho = dh_a, *dco = dc_a, *dhn = dh_b, *dcn = dc_b;
    if (dh_in == dh_a) {                              // a chained call: the first output must not be its own input
        std::swap(dho, dhn);
        std::swap(dco, dcn);
    }
    // Software-pipelined over two streams when the caller gives a side stream and per-step dh1d
    // storage: the heads (scoring / text-attention backward, ~45 us of small dependent launches per
    // step) run ahead on the side stream, the tails (LSTM / visual backward, ~65 us) follow on the main
    // stream, each behind the event of its own head.  The heads write only per-step tape slots and
    // their own arena region, so nothing is shared but the events.
    hipStream_t main_st = S(stream), side_st = S(e->side_stream);
    const bool two = side_st && side_st != main_st && gtape->dh1d && gtape->dcat2 && gtape->ds;
    if (two) {
        SF_CHECK_ARG(e->ctx && dctx);
        std::vector<hipEvent_t>& ev = event_pool(e->S + 1);
        if (ev.size() < (size_t)e->S + 1) return SF_ERR_LAUNCH;      // event creation failed
        const Arena whole = arena(ws, ws_bytes);
        const size_t usable = whole.cap;
        const size_t head_n = std::min<size_t>(usable / 4, (size_t)4 << 20);
        // two disjoint regions of the one workspace; both keep the real ticket region
        Arena tail_ar{(float*)ws, usable - head_n, 0, whole.tk};
        Arena head_ar{(float*)ws + (usable - head_n), head_n, 0, whole.tk};
        // (only the first chunk orders the side stream behind the caller's stream -- the forward pass; the heads
        // of a later chunk need nothing from the tails of the chunk before it and keep running ahead)
        if (t_hi == e->S &&
            (hipEventRecord(ev[e->S], main_st) != hipSuccess || hipStreamWaitEvent(side_st, ev[e->S], 0) != hipSuccess))
            return SF_ERR_LAUNCH;
        // ALL heads are issued first (they depend on nothing the tails produce), each followed by
        // its event; then the tails, each behind the event of its head.  (Measured on MI355X: work of
        // two queues overlaps only where a kernel leaves CUs unoccupied -- tools/overlap_probe.py: an
        // 0.73 ms chain of recurrent steps beside 0.87 ms of gate products takes 1.35 ms, beside
        // chip-filling library GEMMs the plain sum, stream priority changes nothing -- so the gain of
        // the second stream is the small kernels of the heads filling the gaps of the tails.)
        const bool use_flags = g_bptt_flags && whole.tk && e->S <= 512;
        unsigned* bflags = whole.tk ? whole.tk + 1200 : nullptr;           // [S] one-shot flags (zero between uses)
        unsigned* fault_word = whole.tk ? whole.tk + PERSIST_TICKET + persistent_fault_word() : nullptr;
        // (g_bptt_lookahead: heads are issued this many steps ahead of their tails instead of all first)
        int th = t_hi - 1;               // the next head to issue
        auto issue_heads_down_to = [&](int t_stop) -> int {
            for (; th >= t_lo && th >= t_stop; --th) {
                StepView v = step_view(e, th);
                const sf_decoder_gtape g = gtape_view(gtape, e, th);
                const CeSrc ce{v.tp.logit, v.glue.target_used, gscale + th, -1, (int)e->A};
                TRY(decoder_bwd_head_i(w, nullptr, &v.U, e->B, e->H, e->D, e->L, e->ctx, &v.tp, &g, dlogit,
                                       g.dh1d, dctx, head_ar, side_st, &ce));
                if (use_flags) TRY(flag_set(bflags + th, 1u, side_st));
                else if (hipEventRecord(ev[th], side_st) != hipSuccess) return SF_ERR_LAUNCH;
            }
            return SF_OK;
        };
        // tail t may run once head t is complete: an event per step, or (sf_debug_bptt_flags) a one-shot device flag set
        // by a one-thread kernel behind the head and waited for / cleared by a one-wave kernel in front of the tail.
        // Measured by wall clock (tools/bptt_overlap_probe.py): heads alone 0.93 ms, tails alone 1.14 ms, both chains in
        // one graph 1.46 ms, as two graphs on two streams 1.43 ms -- they overlap either way; events and flags are equal.
        // (rocprofv3 --kernel-trace serialises the queues: its timelines show every head before the first tail.)
        int waited_down_to = t_hi;       // heads >= this index have been waited for
        auto wait_heads_down_to = [&](int t_need) -> int {
            for (int k = waited_down_to - 1; k >= t_need; --k) {
                if (use_flags) TRY(flag_wait_clear(bflags + k, fault_word, 16u /* FAULT_BPTT */, main_st));
                else if (hipStreamWaitEvent(main_st, ev[k], 0) != hipSuccess) return SF_ERR_LAUNCH;
            }
            if (t_need < waited_down_to) waited_down_to = t_need;
            return SF_OK;
        };
        bool cell_done = false;          // this step's pointwise backward ran as the epilogue of the step before
        for (int t = t_hi - 1; t >= t_lo; --t) {
            if (g_bptt_part != 2) TRY(issue_heads_down_to(g_bptt_lookahead < 0 ? t_lo : t - g_bptt_lookahead));
            if (g_bptt_part == 1) continue;
            StepView v = step_view(e, t);
            const sf_decoder_gtape g = gtape_view(gtape, e, t);
            const float* h0 = t == 0 ? e->h_init : e->tape.h1 + (size_t)(t - 1) * BH;
            const float* c0 = t == 0 ? e->c_init : e->tape.c1 + (size_t)(t - 1) * BH;
            // the last product of this tail completes dh1 of step t - 1: that step's pointwise backward (it needs the
            // head of step t - 1 as well) rides in its epilogue
            LstmPwBwd next_pw{};
            bool next_fused = false;
            const bool try_fuse = g_fuse_cell_bwd && t > t_lo && g.dgates;
            Dropout d_next;
            if (g_bptt_part != 2) TRY(wait_heads_down_to(try_fuse ? t - 1 : t));
            if (try_fuse) {
                StepView vn = step_view(e, t - 1);
                const sf_decoder_gtape gn = gtape_view(gtape, e, t - 1);
                const float* c0n = t - 1 == 0 ? e->c_init : e->tape.c1 + (size_t)(t - 2) * BH;
                d_next = make_dropout(drop, 2 * (e->step0 + t - 1) + 1, 2);
                next_pw = cell_pw_bwd(vn.tp.gates, c0n, vn.tp.c1, nullptr, gn.dh1d, dco, e->B, e->H, gn.dgates, dcn, &d_next);
            }
            TRY(decoder_bwd_tail_i(w, nullptr, &v.X, e->B, e->H, e->D, h0, c0, &v.tp, &g, dh1, g.dh1d, dc1,
                                   dho, dco, drop, e->step0 + t, tail_ar, main_st, try_fuse ? &next_pw : nullptr,
                                   &next_fused, cell_done));
            cell_done = next_fused;
            dh1 = dho;
            dc1 = dco;
            std::swap(dho, dhn);
            std::swap(dco, dcn);
        }
    } else {
        for (int t = t_hi - 1; t >= t_lo; --t) {
            StepView v = step_view(e, t);
            const sf_decoder_gtape g = gtape_view(gtape, e, t);
            const float* h0 = t == 0 ? e->h_init : e->tape.h1 + (size_t)(t - 1) * BH;
            const float* c0 = t == 0 ? e->c_init : e->tape.c1 + (size_t)(t - 1) * BH;
            // the cross-entropy backward (softmax - onehot, scaled by 1 / live rows of the step) is
            // formed inside the scoring backward: one dependent launch less per step
            const CeSrc ce{v.tp.logit, v.glue.target_used, gscale + t, -1, (int)e->A};
            TRY(decoder_bwd_i(w, nullptr, &v.X, &v.U, e->B, e->H, e->D, e->L, h0, c0, e->ctx, &v.tp, &g,
                              dlogit, dh1, dc1, dho, dco, dctx, drop, e->step0 + t, ws, ws_bytes, stream,
                              &ce));
            dh1 = dho;
            dc1 = dco;
            std::swap(dho, dhn);
            std::swap(dco, dcn);
        }
    }
    *result_in_b = (dh1 == dh_b) ? 1 : 0;
    if (gtape->dcat2 && gtape->ds && dctx && t_lo == 0)     // the deferred context gradient, once for the episode
        TRY(ctx_grad_accum(e->tape.alpha, gtape->ds, gtape->dcat2, 2 * e->H, e->tape.t_text, e->S, e->B,
                           e->L, e->H, dctx, S(stream)));
    return SF_OK;
}

// All weight gradients of S stacked decoder steps, each as ONE product of reduction depth M = S*B.
int sf_attn_decoder_wgrad(const sf_decoder_w* w, const sf_decoder_g* g, int M, int H, int D, int F,
                          const float* h0_all, const sf_decoder_tape* tp, const sf_decoder_gtape* gt,
                          void* ws, size_t ws_bytes, sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(w && g && h0_all && tp && gt && M > 0);
    hipStream_t st = S(stream);
    Arena ar = arena(ws, ws_bytes);
    // LSTMCell (model.py:393): dW_ih's leading columns (whole rounds of the chip) as one large product, its narrow tail
    // with the small matrices below
    const int q_main = gemm_tn_main_columns(M, 4 * H, 2 * F);
    if (g->lstm.w_ih) TRY(gemm_tn(gt->dgates, 4 * H, tp->xin, 2 * F, M, 4 * H, q_main, g->lstm.w_ih, 2 * F, 1, st, ar.rest(), ar.rest_n()));
    // the other matrices (a few dozen output tiles each over the same M stacked rows): one grouped launch sequence
    TnJob jobs[8];
    int nj = 0;
    auto job = [&](const float* Y, int ldy, const float* X, int ldx, int P, int Q, float* out, int ldo = 0) {
        if (out) jobs[nj++] = TnJob{Y, ldy, X, ldx, M, P, Q, out, ldo ? ldo : Q, 1};
    };
    if (g->lstm.w_ih && q_main < 2 * F)
        job(gt->dgates, 4 * H, tp->xin + q_main, 2 * F, 4 * H, 2 * F - q_main, g->lstm.w_ih + q_main, 2 * F);
    job(gt->dgates, 4 * H, h0_all, H, 4 * H, H, g->lstm.w_hh);
    job(tp->t_v, D, gt->dq, F, D, F, g->visual.w_v);                     // visual attention (model.py:389)
    job(gt->dt_v, D, h0_all, H, D, H, g->visual.w_h);
    job(gt->dpre, H, tp->cat2, 2 * H, H, 2 * H, g->text.w_out);          // text attention (model.py:395)
    job(gt->dt_text, H, tp->cat2 + H, 2 * H, H, H, g->text.w_in);
    job(tp->wt, D, gt->dr, F, D, F, g->action.w_a);                      // action scoring (model.py:396)
    job(gt->dta, D, tp->h_tilde, H, D, H, g->action.w_h);
    if (nj) TRY(gemm_tn_group(jobs, nj, st, ar.rest(), ar.rest_n()));
    if (g->lstm.w_ih || g->lstm.w_hh || g->lstm.b_ih || g->lstm.b_hh)
        TRY(colsum_pair(gt->dgates, 4 * H, M, 4 * H, g->lstm.b_ih, g->lstm.b_hh, ar, st));
    if (g->visual.b_h) TRY(colsum(gt->dt_v, D, M, D, g->visual.b_h, 1, st, nullptr, ar.rest(), ar.rest_n()));
    if (g->action.b_a) TRY(dot_rows_accum(gt->dc, tp->wt, D, M, D, g->action.b_a, st));
    if (g->action.b_out) TRY(sum_accum(gt->dc, M, g->action.b_out, st));
    if (g->action.w_out) TRY(colsum_prod(gt->dwt, D, tp->t_a, D, M, D, g->action.w_out, st));
    if (g->action.b_h) TRY(colsum(gt->dta, D, M, D, g->action.b_h, 1, st, nullptr, ar.rest(), ar.rest_n()));
    return SF_OK;
}

int sf_follower_glue_fwd(const sf_cands* U, int B, float* logit, const sf_follower_glue* glue,
                         sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(U && logit && glue && B > 0 && glue_ok(U, glue));
    return follower_glue_fwd(make_glue(cands(U), B, logit, glue), S(stream));
}

int sf_follower_glue_bwd(int B, int A, const float* logit, const int64_t* target_used,
                         const float* gscale, float* dlogit, sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(logit && target_used && gscale && dlogit && B > 0 && A > 0);
    return softmax_ce_bwd(B, A, A, logit, target_used, -1, gscale, dlogit, S(stream));
}

int sf_reduce_terms(const float* term, const float* live, int T, int B, float* sum_cnt,
                    sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(term && live && sum_cnt && T > 0 && B > 0);
    return reduce_terms(term, live, T, B, sum_cnt, S(stream));
}

int sf_loss_finalize(const float* sum_cnt, int T, float* loss, float* gscale, sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(sum_cnt && loss && gscale && T > 0);
    return loss_finalize(sum_cnt, T, loss, gscale, S(stream));
}

// ---- a5 EncoderLSTM.forward (model.py:81-104) --------------------------------------------------------
int sf_encoder_lstm_fwd(const sf_encoder_w* w, int B, int Lpad, int T, int E, int H,
                        const int64_t* seq, const int32_t* lengths, float* ctx, float* decoder_init,
                        float* c_t, const sf_encoder_tape* tp, const sf_dropout* drop,
                        uint32_t drop_stream, void* ws, size_t ws_bytes, sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(w && seq && lengths && ctx && decoder_init && c_t && tp && B > 0 && T > 0 &&
                 T <= Lpad && E % 4 == 0 && H % 16 == 0);
    Arena ar = arena(ws, ws_bytes);
    hipStream_t st = S(stream);
    const size_t BH = (size_t)B * H;
    // model.py:85  embedding, time-major so each step reads one contiguous [B,E] block (kept for the
    // backward's dW_ih; an inference call with the input-product table passes tape->emb = NULL)
    SF_CHECK_ARG(tp->emb || w->xw_table);
    if (tp->emb) TRY(embedding_tm(w->embedding, E, seq, B, Lpad, T, tp->emb, st));
    if (w->flags & SF_ENC_EMB_DROPOUT) {         // trainable embedding: model.py:86-87 drops the embedded tokens
        SF_CHECK_ARG(tp->emb && !w->xw_table);
        TRY(dropout_tm(tp->emb, T, B, E, make_dropout(drop, drop_stream ^ 0x40000000u),
                       (w->flags & SF_ENC_REVERSED) ? lengths : nullptr, st));
    }
    // input product: a row of the host's [vocab,4H] table per token, or hoisted for all steps at
    // once ([T*B,E] x [E,4H])
    if (!w->xw_table) SF_CHECK_ARG(tp->emb && tp->xg);
    if (!w->xw_table)
        TRY(linear_plain(tp->emb, E, w->lstm.w_ih, E, nullptr, T * B, 4 * H, E, EPI_NONE, tp->xg,
                         4 * H, ar, st));
    // (SF_ENC_RAW_STATE: ctx is handed over raw as well -- the caller drops the assembled [forward | reverse] rows)
    const Dropout dctx = (w->flags & SF_ENC_RAW_STATE) ? make_dropout(nullptr, 0) : make_dropout(drop, drop_stream);
    // all T steps as ONE persistent launch (sf_persist.hip) where it applies; bit-identical results
    bool persistent = false;
    if (w->xw_table && !(w->flags & SF_ENC_PER_STEP) && encoder_persistent_supported(B, H, T)) {
        Arena pa = ar;
        float* xchg = pa.take(encoder_persistent_xchg_floats(H));
        if (xchg && ar.tickets()) {
            TRY(encoder_persistent(w->lstm.w_hh, w->lstm.b_ih, w->lstm.b_hh, w->xw_table, seq, Lpad, lengths, B,
                                   H, T, tp->gates, tp->hs, tp->cs, ctx, dctx, xchg,
                                   ar.tickets() + PERSIST_TICKET, st, c_t));
            persistent = true;
        }
    }
    if (!persistent) {
        TRY(fill(tp->hs, BH, 0.f, st));   // model.py:67-79 init_state
        TRY(fill(tp->cs, BH, 0.f, st));
    }
    for (int t = 0; t < T && !persistent; ++t) {
        // one fused launch per time step: h_t W_hh^T on the matrix cores + gates + cell update
        LstmStepArgs f{};
        f.h0 = tp->hs + t * BH; f.w_hh = w->lstm.w_hh; f.x = nullptr; f.xg = tp->xg + (size_t)t * B * 4 * H;
        if (w->xw_table) {                       // token lookup: seq is [B, Lpad], step t = column t
            f.xg = w->xw_table;
            f.xg_index = seq + t;
            f.xg_index_ld = Lpad;
        }
        f.b_ih = w->lstm.b_ih; f.b_hh = w->lstm.b_hh; f.B = B; f.H = H;
        LstmPwFwd& p = f.pw;
        p.c0 = tp->cs + t * BH; p.h0 = tp->hs + t * BH; p.B = B; p.H = H;
        p.gates = tp->gates ? tp->gates + (size_t)t * B * 4 * H : nullptr;
        p.h1 = tp->hs + (t + 1) * BH; p.c1 = tp->cs + (t + 1) * BH;
        p.h1_drop = nullptr; p.drop = make_dropout(nullptr, 0);
        p.lengths = lengths; p.t = t; p.ctx_out = ctx; p.ld_ctx = T * H; p.ctx_drop = dctx;
        TRY(lstm_step_fused(f, st));
    }
    // model.py:96-99  decoder_init = tanh(encoder2decoder(h_T)); c_T raw
    if (w->flags & SF_ENC_RAW_STATE)               // one direction of a bidirectional encoder: the raw h_T
        TRY(add2(tp->hs + T * BH, H, nullptr, 0, B, H, decoder_init, H, st));
    else
        TRY(linear_plain(tp->hs + T * BH, H, w->w_e2d, H, w->b_e2d, B, H, H, EPI_TANH, decoder_init, H,
                         ar, st));
    if (persistent) return SF_OK;                 // (the persistent launch wrote c_T itself)
    return add2(tp->cs + T * BH, H, nullptr, 0, B, H, c_t, H, st);
}

int sf_encoder_lstm_bwd(const sf_encoder_w* w, const sf_encoder_g* g, int B, int T, int E, int H,
                        const int32_t* lengths, const float* decoder_init, const float* dctx,
                        const float* d_init, const float* d_ct, const sf_encoder_tape* tp,
                        const sf_dropout* drop, uint32_t drop_stream, void* ws, size_t ws_bytes,
                        sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(w && lengths && decoder_init && tp && B > 0 && T > 0);
    Arena ar = arena(ws, ws_bytes);
    hipStream_t st = S(stream);
    const size_t BH = (size_t)B * H, BG = (size_t)B * 4 * H;
    float* dh = ar.take(BH);        // gradient wrt h after step t
    float* dc = ar.take(BH);
    float* dcn = ar.take(BH);
    float* dctx_t = ar.take(BH);
    float* dpass = ar.take(BH);
    float* dpre = ar.take(BH);
    NEED(dh && dc && dcn && dctx_t && dpass && dpre);
    // through decoder_init = tanh(W h_T + b)
    if (d_init && (w->flags & SF_ENC_RAW_STATE)) {
        TRY(add2(d_init, H, nullptr, 0, B, H, dh, H, st));          // gradient wrt the raw h_T
    } else if (d_init) {
        TRY(tanh_bwd(decoder_init, H, d_init, H, B, H, dpre, H, st));
        TRY(data_grad(dpre, H, w->w_e2d, w->w_e2d_t, B, H, H, dh, H, 0, ar, st));
        if (g && g->w_e2d) TRY(gemm_tn(dpre, H, tp->hs + T * BH, H, B, H, H, g->w_e2d, H, 1, st, ar.rest(), ar.rest_n()));
        if (g && g->b_e2d) TRY(colsum(dpre, H, B, H, g->b_e2d, 1, st, nullptr, ar.rest(), ar.rest_n()));
    } else {
        TRY(fill(dh, BH, 0.f, st));
    }
    TRY(add2(d_ct, H, nullptr, 0, B, H, dc, H, st));
    const Dropout dd = (w->flags & SF_ENC_RAW_STATE) ? make_dropout(nullptr, 0) : make_dropout(drop, drop_stream);
    // dgates for step t overwrite tp->xg[t] (the hoisted product is dead after the forward)
    bool persistent = false;
    if (!(w->flags & SF_ENC_PER_STEP) && encoder_persistent_supported(B, H, T) && ar.tickets()) {
        // ALL T backward steps as one persistent launch (sf_persist.hip)
        Arena pa = ar;
        float* xchg = pa.take(encoder_bwd_persistent_xchg_floats());
        if (xchg) {
            TRY(encoder_bwd_persistent(w->lstm.w_hh, lengths, B, H, T, tp->gates, tp->cs, dctx, dd, dh, dc, tp->xg,
                                       xchg, ar.tickets() + PERSIST_TICKET, st));
            persistent = true;
        }
    }
    if (persistent) {
    } else if (w->lstm.w_hh_t && H % 16 == 0 && H <= 512) {
        // ONE launch per step: dh_{t+1} = pass + dgates_{t+1} W_hh on the matrix cores and the cell
        // backward of step t on the same tile (lstm_bwd_step_fused_kernel)
        float* dh_b = dpass;                              // ping-pong partners of dh / dc
        for (int t = T - 1; t >= 0; --t) {
            LstmBwdStepArgs f{};
            f.dgates_next = t + 1 < T ? tp->xg + (size_t)(t + 1) * BG : nullptr;
            f.w_hh_t = w->lstm.w_hh_t;
            f.dh_in = dh; f.dc_in = dc;
            f.gates = tp->gates + t * BG; f.c0 = tp->cs + t * BH; f.c1 = tp->cs + (t + 1) * BH;
            f.dctx = dctx; f.T = T; f.t = t; f.ctx_drop = dd; f.lengths = lengths; f.B = B; f.H = H;
            f.dgates = tp->xg + t * BG; f.dc_out = dcn; f.dh_out = dh_b;
            TRY(lstm_bwd_step_fused(f, st));
            std::swap(dc, dcn);
            std::swap(dh, dh_b);
        }
    } else
    for (int t = T - 1; t >= 0; --t) {
        // two dependent launches per step: the cell backward reads dctx[:, t, :] itself and leaves
        // the pass-through of dead rows IN dh (element-wise in place), then dh += dgates W_hh
        LstmPwBwd p{};
        p.gates = tp->gates + t * BG; p.c0 = tp->cs + t * BH; p.c1 = tp->cs + (t + 1) * BH;
        p.dh1 = dh; p.dh1_b = nullptr; p.dc1 = dc; p.B = B; p.H = H;
        p.dgates = tp->xg + t * BG; p.dc0 = dcn; p.lengths = lengths; p.t = t; p.dh0_pass = dh;
        p.dctx = dctx; p.T = T; p.ctx_drop = dd;
        TRY(lstm_pointwise_bwd(p, st));
        TRY(data_grad(tp->xg + t * BG, 4 * H, w->lstm.w_hh, w->lstm.w_hh_t, B, 4 * H, H, dh, H, 1, ar, st));
        std::swap(dc, dcn);
    }
    if (g) {
        // all time steps in one product each: reduction depth T*B
        if (g->lstm.w_hh) TRY(gemm_tn(tp->xg, 4 * H, tp->hs, H, T * B, 4 * H, H, g->lstm.w_hh, H, 1, st, ar.rest(), ar.rest_n()));
        if (g->lstm.w_ih) TRY(gemm_tn(tp->xg, 4 * H, tp->emb, E, T * B, 4 * H, E, g->lstm.w_ih, E, 1, st, ar.rest(), ar.rest_n()));
        TRY(colsum_pair(tp->xg, 4 * H, T * B, 4 * H, g->lstm.b_ih, g->lstm.b_hh, ar, st));
        if (g->embedding) {
            // trainable embedding (model.py:57-60): d emb = dgates W_ih for all steps at once, through the dropout of
            // the embedded tokens, scattered into the rows of their tokens
            SF_CHECK_ARG(g->seq && g->Lpad >= T);
            Arena ea = ar;
            float* demb = ea.take((size_t)T * B * E);
            NEED(demb);
            TRY(data_grad(tp->xg, 4 * H, w->lstm.w_ih, w->lstm.w_ih_t, T * B, 4 * H, E, demb, E, 0, ea, st));
            const Dropout de = (w->flags & SF_ENC_EMB_DROPOUT) ? make_dropout(drop, drop_stream ^ 0x40000000u)
                                                               : make_dropout(nullptr, 0);
            TRY(embedding_bwd(demb, E, g->seq, g->Lpad, T, B, E, g->padding_idx, de,
                              (w->flags & SF_ENC_REVERSED) ? lengths : nullptr, g->embedding, st));
        }
    }
    return SF_OK;
}

// ---- a11 gathers ---------------------------------------------------------------------------------------
int sf_gather_panorama(const sf_pano* X, int B, float* out, sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(X && out && B > 0 && X->table && X->loc_table && X->vp && X->view);
    return gather_panorama(pano(X), B, out, S(stream));
}
int sf_gather_candidates(const sf_cands* U, int B, float* all_u, float* is_valid, sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(U && all_u && B > 0 && U->table && U->vp && U->cand_view && U->cand_sincos && U->a_num);
    return gather_candidates(cands(U), B, all_u, is_valid, S(stream));
}
int sf_gather_actions(const sf_cands* U, int B, const int32_t* a, float* out, sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(U && a && out && B > 0);
    return gather_actions(cands(U), B, a, out, S(stream));
}

int sf_gather_actions_ld(const sf_cands* U, int B, const int32_t* a, float* out, int ld_out, sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(U && a && out && B > 0 && ld_out > 0);
    return gather_actions(cands(U), B, a, out, S(stream), ld_out);
}

int sf_gather_path_actions(const float* table, int V, int IMG, int LOC, const int32_t* vp, const int32_t* act_view,
                           const float* act_sincos, const int32_t* act, int N, float* out, int ld_out,
                           sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(table && vp && act_view && act_sincos && act && out && N > 0 && V > 0 && ld_out >= IMG + LOC);
    return gather_path_actions(table, V, IMG, LOC, vp, act_view, act_sincos, act, N, out, ld_out, S(stream));
}

// ---- search helpers -----------------------------------------------------------------------
int sf_gather_rows(const float* src, int ld_src, const int32_t* idx, int n, int width, float* dst,
                   int ld_dst, sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(src && idx && dst && n > 0 && width > 0);
    return gather_rows(src, ld_src, idx, n, width, dst, ld_dst, S(stream));
}

int sf_logprob_topk(float* logit, int ld, int N, int n, const int32_t* n_valid, int k, int32_t* idx,
                    float* logp, sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(logit && logp && N > 0 && n > 0 && ld >= n && (idx || k == n));
    return logprob_topk(logit, ld, N, n, n_valid, k, idx, logp, S(stream));
}
int sf_scatter_rows(const float* src, int ld_src, const int32_t* idx, int n, int width, float* dst,
                    int ld_dst, sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(src && idx && dst && n > 0 && width > 0);
    return scatter_rows(src, ld_src, idx, n, width, dst, ld_dst, S(stream));
}

// ---- a9 SpeakerDecoderLSTM.forward (model.py:497-519) -----------------------------------------------------
int sf_speaker_decoder_fwd(const sf_spk_decoder_w* w, int B, int E, int H, int Tp, int vocab,
                           const int64_t* prev_word, const float* h0, const float* c0,
                           const float* ctx, const uint8_t* ctx_mask, const int32_t* ctx_row,
                           const sf_spk_decoder_tape* tp, const sf_dropout* drop, uint32_t step_id,
                           void* ws, size_t ws_bytes, sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(w && prev_word && h0 && c0 && ctx && tp && B > 0 && Tp > 0 && vocab > 0);
    Arena ar = arena(ws, ws_bytes);
    hipStream_t st = S(stream);
    const int ldv = (vocab + 3) & ~3;
    const Dropout d_h = make_dropout(drop, 2 * step_id + 1, 2);
    if (tp->emb) TRY(embedding_rows(w->embedding, E, prev_word, B, tp->emb, st));   // :497-498
    if (w->flags & SF_SPK_EMB_DROPOUT) {          // trainable embedding: :499-500 drops the embedded word
        SF_CHECK_ARG(tp->emb && !w->xw_table);
        TRY(dropout_tm(tp->emb, 1, B, E, make_dropout(drop, 2 * step_id, 2), nullptr, st));
    }
    if (w->xw_table && H % 16 == 0 && H <= 1024) {
        // x W_ih^T is a row of the precomputed [vocab,4H] table: recurrent half only
        LstmStepArgs f{};
        f.h0 = h0; f.w_hh = w->lstm.w_hh; f.x = nullptr; f.xg = w->xw_table; f.xg_index = prev_word;
        f.b_ih = w->lstm.b_ih; f.b_hh = w->lstm.b_hh; f.B = B; f.H = H;
        LstmPwFwd& p = f.pw;
        p.c0 = c0; p.h0 = h0; p.B = B; p.H = H; p.gates = tp->gates; p.h1 = tp->h1; p.c1 = tp->c1;
        p.h1_drop = tp->cat2 + H; p.ld_h1_drop = 2 * H; p.drop = d_h; p.lengths = nullptr;
        TRY(lstm_step_fused(f, st));                                              // :515-516
    } else {
        SF_CHECK_ARG(tp->emb);
        TRY(lstm_fwd_i(&w->lstm, B, E, H, tp->emb, E, h0, c0, tp->h1, tp->c1, tp->gates,
                       tp->cat2 + H, 2 * H, d_h, ar, st));                        // :515-516
    }
    TRY(softdot_fwd_i(&w->attn, B, Tp, H, nullptr, 0, ctx, ctx_mask, tp->h_tilde, tp->alpha, tp->cat2,
                      tp->t_text, ar, st, ctx_row));                              // :517
    // (columns vocab..ldv-1 of tape->logit are padding: never read by the glue; callers slice)
    return linear_plain(tp->h_tilde, H, w->w_out, H, w->b_out, B, vocab, H, EPI_NONE, tp->logit, ldv,
                        ar, st);                                                  // :518
}

// The whole word loop of an inference pass in one persistent launch (sf_persist.hip)
int sf_speaker_decode(const sf_spk_decoder_w* w, int B, int H, int Tp, int vocab, int n_steps, int feedback,
                      int pad_idx, int eos_idx, const int64_t* targets, const float* h_init,
                      const float* c_init, const float* ctx, const uint8_t* ctx_mask, int64_t* words,
                      uint8_t* ended, float* step_scores, float* nll_term, float* live, float* logits,
                      float* alpha, float* h1_tape, float* c1_tape, const sf_sample* sample, void* ws,
                      size_t ws_bytes, sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(w && targets && h_init && c_init && ctx && words && ended && step_scores && nll_term && live &&
                 B > 0 && n_steps > 0 && Tp > 0 && feedback >= 0 && feedback <= 2 && (feedback != 2 || sample) &&
                 (!h1_tape == !c1_tape));
    if (!w->xw_table || !w->attn.w_in_t || !speaker_persistent_supported(B, H, Tp, vocab)) return SF_ERR_UNSUPPORTED;
    Arena ar = arena(ws, ws_bytes);
    hipStream_t st = S(stream);
    float* cq = ar.take((size_t)B * Tp * H);
    float* cw = ar.take((size_t)B * Tp * H);
    float* xchg = ar.take(speaker_persistent_xchg_floats());
    NEED(cq && cw && xchg && ar.tickets());
    // cq = ctx W_in (so that ctx_l . (W_in h) = cq_l . h), cw = ctx W_c^T with W_c = linear_out[:, :H]
    TRY(linear_plain(ctx, H, w->attn.w_in_t, H, nullptr, B * Tp, H, H, EPI_NONE, cq, H, ar, st));
    TRY(linear_plain(ctx, H, w->attn.w_out, 2 * H, nullptr, B * Tp, H, H, EPI_NONE, cw, H, ar, st));
    const int ldv = (vocab + 3) & ~3;
    return speaker_persistent(w->lstm.w_hh, w->lstm.b_ih, w->lstm.b_hh, w->xw_table, w->attn.w_out, 2 * H, w->w_out,
                              w->b_out, vocab, ldv, cq, cw, ctx_mask, h_init, c_init, targets, feedback, pad_idx,
                              eos_idx, B, H, Tp, n_steps, words, step_scores, nll_term, live, logits, alpha, h1_tape,
                              c1_tape, ended, xchg, ar.tickets() + PERSIST_TICKET, st, sample);
}

// d h~ = dlogit W_out (dlogit [B,ldv] with zero padding columns): K-contiguous through decoder2action^T when the host
// keeps it (sf_spk_decoder_w.w_out_t [H,ldv]), else the strided NN kernel
static int spk_dlogit_to_dht(const sf_spk_decoder_w* w, const float* dlogit, int ldv, int B, int H, int vocab, float* dht,
                             Arena& ar, hipStream_t st) {
    if (w->w_out_t) {
        Seg sg{dlogit, ldv, w->w_out_t, ldv, ldv};
        LinearOut o{};
        o.y = dht; o.ldy = H; o.epi = EPI_NONE;
        const int rc = linear_nt(&sg, 1, B, H, o, ar.rest(), ar.rest_n(), st);
        if (rc != SF_ERR_UNSUPPORTED) return rc;
    }
    return gemm_nn_ws(dlogit, ldv, w->w_out, H, B, H, vocab, dht, H, 0, ar.rest(), ar.rest_n(), st);
}

int sf_speaker_decoder_bwd(const sf_spk_decoder_w* w, const sf_spk_decoder_g* g, int B, int E, int H,
                           int Tp, int vocab, const int64_t* prev_word, const float* h0, const float* c0, const float* ctx,
                           const sf_spk_decoder_tape* tp, const float* dlogit, const float* dh1,
                           const float* dc1, float* dh0, float* dc0, float* dctx,
                           const sf_dropout* drop, uint32_t step_id, void* ws, size_t ws_bytes,
                           sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(w && h0 && c0 && ctx && tp && dlogit && dh0 && dc0 && B > 0);
    Arena ar = arena(ws, ws_bytes);
    hipStream_t st = S(stream);
    const int ldv = (vocab + 3) & ~3;
    const Dropout d_h = make_dropout(drop, 2 * step_id + 1, 2);
    float* dht = ar.take((size_t)B * H);
    float* dh1d = ar.take((size_t)B * H);
    float* dh1m = ar.take((size_t)B * H);
    NEED(dht && dh1d && dh1m);
    // dlogit [B,ldv] has zeros in its padding columns
    TRY(spk_dlogit_to_dht(w, dlogit, ldv, B, H, vocab, dht, ar, st));
    if (g && g->w_out) TRY(gemm_tn(dlogit, ldv, tp->h_tilde, H, B, vocab, H, g->w_out, H, 1, st, ar.rest(), ar.rest_n()));
    if (g && g->b_out) TRY(colsum(dlogit, ldv, B, vocab, g->b_out, 1, st, nullptr, ar.rest(), ar.rest_n()));
    TRY(softdot_bwd_i(&w->attn, g ? &g->attn : nullptr, B, Tp, H, ctx, tp->alpha, tp->cat2,
                      tp->t_text, tp->h_tilde, dht, dh1d, H, dctx, ar, st));
    TRY(dropout_copy(dh1d, H, B, H, dh1m, H, d_h, 0, st));
    // a frozen (GloVe) embedding needs no gradient wrt the LSTM input (model.py:472); a trainable one gets
    // d emb = dgates W_ih through its dropout, scattered into the rows of the previous words
    if (!(g && g->embedding))
        return lstm_bwd_i(&w->lstm, g ? &g->lstm : nullptr, B, E, H, tp->emb, E, h0, c0, tp->c1,
                          tp->gates, dh1, dh1m, dc1, nullptr, 0, dh0, dc0, ar, st);
    SF_CHECK_ARG(prev_word && tp->emb);
    float* demb = ar.take((size_t)B * E);
    NEED(demb);
    TRY(lstm_bwd_i(&w->lstm, &g->lstm, B, E, H, tp->emb, E, h0, c0, tp->c1, tp->gates, dh1, dh1m, dc1, demb, E, dh0, dc0,
                   ar, st));
    const Dropout de = (w->flags & SF_SPK_EMB_DROPOUT) ? make_dropout(drop, 2 * step_id, 2) : make_dropout(nullptr, 0);
    return embedding_bwd(demb, E, prev_word, 1, 1, B, E, -1, de, nullptr, g->embedding, st);
}

int sf_speaker_glue_fwd(int B, int vocab, int ldv, const float* logit, const int64_t* target,
                        int feedback, int pad_idx, int eos_idx, uint8_t* ended, int64_t* w_t,
                        float* score, float* nll_term, float* live, const sf_sample* sample, sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(logit && target && ended && w_t && score && nll_term && live && B > 0 &&
                 vocab > 0 && ldv >= vocab && feedback >= 0 && feedback <= 2 && (feedback != 2 || sample));
    return speaker_glue_fwd(B, vocab, ldv, logit, target, feedback, pad_idx, eos_idx, ended, w_t,
                            score, nll_term, live, S(stream), sample);
}

// ---- a8 SpeakerEncoderLSTM.forward (model.py:437-457), all path steps in one call -------------------------------------
static int speaker_encoder_fwd_i(const sf_visual_fold64* fold64, const sf_visual_w* vw, const sf_lstm_w* lw, const float* w_e2d, const float* b_e2d,
                           const sf_pano* X0, int Tp, int B, int H, int D, float* xin, float* alpha, float* t_v,
                           float* q, float* gates, float* hs, float* cs, float* ctx, const float* act_emb,
                           float* h_init, const sf_dropout* drop, uint32_t step0, void* ws, size_t ws_bytes,
                           sf_stream stream) {
    SF_CHECK_ARG(vw && lw && w_e2d && X0 && X0->vp && X0->view && Tp > 0 && B > 0 && xin && alpha && t_v && q && gates && hs &&
                 cs && h_init && (!act_emb || drop) && (!ctx || !drop));
    const int F = X0->IMG + X0->LOC, V = X0->V;
    const size_t BH = (size_t)B * H;
    for (int t = 0; t < Tp; ++t) {
        sf_pano X = *X0;
        X.vp = adv(X0->vp, (size_t)t * B);
        X.view = adv(X0->view, (size_t)t * B);
        float* x_t = xin + (size_t)t * B * 2 * F;
        // (float64 query and scores: see visual_fwd_i / csrc/sf_precise.hip; sf_debug_precise_attention(0) = fp32)
        TRY(visual_fwd_i(vw, pano(&X), B, H, D, hs + t * BH, x_t + F, 2 * F, alpha + (size_t)t * B * V,
                         t_v + (size_t)t * B * D, q + (size_t)t * B * F, make_dropout(drop, 2 * (step0 + t), 2), F,
                         arena(ws, ws_bytes), S(stream), nullptr, g_precise_attention != 0, fold64));
        if (act_emb)
            TRY(dropout_copy(act_emb + (size_t)t * B * F, F, B, F, x_t, 2 * F, make_dropout(drop, 2 * (step0 + t), 2), 0,
                             S(stream)));
        TRY(sf_lstm_cell_fwd(lw, B, 2 * F, H, x_t, 2 * F, hs + t * BH, cs + t * BH, hs + (t + 1) * BH, cs + (t + 1) * BH,
                             gates + (size_t)t * B * 4 * H, ctx ? ctx + (size_t)t * H : nullptr, ctx ? Tp * H : 0, nullptr, 0,
                             ws, ws_bytes, stream));
    }
    return sf_linear_fwd(hs + Tp * BH, H, w_e2d, b_e2d, B, H, H, 1, h_init, H, ws, ws_bytes, stream);
}

int sf_speaker_encoder_fwd(const sf_visual_w* vw, const sf_lstm_w* lw, const float* w_e2d, const float* b_e2d,
                           const sf_pano* X0, int Tp, int B, int H, int D, float* xin, float* alpha, float* t_v,
                           float* q, float* gates, float* hs, float* cs, float* ctx, const float* act_emb,
                           float* h_init, const sf_dropout* drop, uint32_t step0, void* ws, size_t ws_bytes,
                           sf_stream stream) {
    SF_ENTER();
    return speaker_encoder_fwd_i(nullptr, vw, lw, w_e2d, b_e2d, X0, Tp, B, H, D, xin, alpha, t_v, q, gates, hs, cs, ctx, act_emb,
                                 h_init, drop, step0, ws, ws_bytes, stream);
}

int sf_speaker_encoder_fwd_folded(const sf_visual_fold64* fold, const sf_visual_w* vw, const sf_lstm_w* lw, const float* w_e2d,
                                  const float* b_e2d, const sf_pano* X0, int Tp, int B, int H, int D, float* xin, float* alpha,
                                  float* t_v, float* q, float* gates, float* hs, float* cs, float* ctx, const float* act_emb,
                                  float* h_init, const sf_dropout* drop, uint32_t step0, void* ws, size_t ws_bytes,
                                  sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(fold && fold->m_v && fold->c_v);
    return speaker_encoder_fwd_i(fold, vw, lw, w_e2d, b_e2d, X0, Tp, B, H, D, xin, alpha, t_v, q, gates, hs, cs, ctx, act_emb,
                                 h_init, drop, step0, ws, ws_bytes, stream);
}

int sf_visual_query_fold_f64(const sf_visual_w* w, int H, int D, int F, double* m_v, double* c_v, sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(w && w->w_v_t && w->w_h_t && w->b_h && m_v && c_v && H > 0 && D > 0 && F > 0 && !(D & 3));
    // M_v[F,H] = W_v^T[F,D] (W_h^T[H,D])^T, c_v[1,F] = b_h[1,D] (W_v^T[F,D])^T -- both accumulated and kept in float64
    TRY(linear_f64(w->w_v_t, nullptr, D, w->w_h_t, D, nullptr, F, H, D, m_v, H, nullptr, 0, S(stream)));
    return linear_f64(w->b_h, nullptr, D, w->w_v_t, D, nullptr, 1, F, D, c_v, F, nullptr, 0, S(stream));
}

// ---- the speaker's word loop with its tape, one call each way (speaker.py:158-197 and its backward) ------------------
namespace {
sf_spk_decoder_tape spk_tape_view(const sf_spk_decoder_tape* p, int t, int B, int E, int H, int Tp, int ldv) {
    sf_spk_decoder_tape v;
    const size_t n = (size_t)t * B;
    v.emb = adv(p->emb, n * E);
    v.gates = adv(p->gates, n * 4 * H);
    v.c1 = adv(p->c1, n * H);
    v.h1 = adv(p->h1, n * H);
    v.cat2 = adv(p->cat2, n * 2 * H);
    v.t_text = adv(p->t_text, n * H);
    v.alpha = adv(p->alpha, n * Tp);
    v.h_tilde = adv(p->h_tilde, n * H);
    v.logit = adv(p->logit, n * ldv);
    return v;
}
}  // namespace

int sf_speaker_words_fwd(const sf_spk_decoder_w* w, int B, int E, int H, int Tp, int vocab, int S, int feedback,
                         int pad_idx, int eos_idx, const int64_t* targets, const float* h_init, const float* c_init,
                         const float* ctx, const uint8_t* ctx_mask, int64_t* words, uint8_t* ended, float* step_scores,
                         float* nll_term, float* live, const sf_spk_decoder_tape* tape0, const sf_dropout* drop,
                         uint32_t step0, const sf_sample* sample, void* ws, size_t ws_bytes, sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(w && tape0 && tape0->h1 && tape0->c1 && tape0->logit && targets && h_init && c_init && ctx && words &&
                 ended && step_scores && nll_term && live && B > 0 && S > 0 && (feedback != 2 || sample));
    const int ldv = (vocab + 3) & ~3;
    const size_t BH = (size_t)B * H;
    // with the input-product table the steps never read the embedded words: they are only kept for the backward's
    // dW_ih, so ALL S*B of them are gathered by one launch behind the loop (words [S+1,B] is complete by then)
    const bool emb_late = tape0->emb && w->xw_table && !(w->flags & SF_SPK_EMB_DROPOUT) && H % 16 == 0 && H <= 1024;
    for (int t = 0; t < S; ++t) {
        sf_spk_decoder_tape tp = spk_tape_view(tape0, t, B, E, H, Tp, ldv);
        if (emb_late) tp.emb = nullptr;
        const float* h0 = t == 0 ? h_init : tape0->h1 + (size_t)(t - 1) * BH;
        const float* c0 = t == 0 ? c_init : tape0->c1 + (size_t)(t - 1) * BH;
        TRY(sf_speaker_decoder_fwd(w, B, E, H, Tp, vocab, words + (size_t)t * B, h0, c0, ctx, ctx_mask, nullptr, &tp, drop,
                                   step0 + t, ws, ws_bytes, stream));
        sf_sample smp{};
        if (sample) {
            smp = *sample;
            smp.stream = sample->stream + (uint32_t)t;
        }
        TRY(sf_speaker_glue_fwd(B, vocab, ldv, tp.logit, targets + (size_t)t * B, feedback, pad_idx, eos_idx, ended,
                                words + (size_t)(t + 1) * B, step_scores + (size_t)t * B, nll_term + (size_t)t * B,
                                live + (size_t)t * B, sample ? &smp : nullptr, stream));
    }
    if (emb_late) TRY(embedding_rows(w->embedding, E, words, S * B, tape0->emb, (hipStream_t)stream));
    return SF_OK;
}

int sf_speaker_words_bwd(const sf_spk_decoder_w* w, const sf_spk_decoder_g* g, int B, int E, int H, int Tp, int vocab,
                         int S, int pad_idx, const int64_t* words, const int64_t* targets, const float* h_init,
                         const float* c_init, const float* ctx, const sf_spk_decoder_tape* tape0, const float* gscale,
                         float* dlogit, float* dh_a, float* dc_a, float* dh_b, float* dc_b, float* dctx,
                         int* result_in_b, const sf_dropout* drop, uint32_t step0, const sf_spk_decoder_gtape* gtape,
                         const float* h0_all, void* ws, size_t ws_bytes, sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(w && tape0 && tape0->h1 && tape0->c1 && tape0->logit && words && targets && h_init && c_init && ctx &&
                 gscale && (dlogit || gtape) && dh_a && dc_a && dh_b && dc_b && dctx && result_in_b && B > 0 && S > 0 &&
                 (!gtape || (gtape->dlogit && gtape->dpre && gtape->dt_text && gtape->dgates && h0_all)));
    const int ldv = (vocab + 3) & ~3;
    const size_t BH = (size_t)B * H;
    const float *dh1 = nullptr, *dc1 = nullptr;
    float *dho = dh_a, *dco = dc_a, *dhn = dh_b, *dcn = dc_b;
    hipStream_t st = (hipStream_t)stream;        // (the parameter S shadows the S() cast helper here)
    for (int t = S - 1; t >= 0; --t) {
        const sf_spk_decoder_tape tp = spk_tape_view(tape0, t, B, E, H, Tp, ldv);
        const float* h0 = t == 0 ? h_init : tape0->h1 + (size_t)(t - 1) * BH;
        const float* c0 = t == 0 ? c_init : tape0->c1 + (size_t)(t - 1) * BH;
        float* dl = gtape ? gtape->dlogit + (size_t)t * B * ldv : dlogit;
        TRY(sf_speaker_glue_bwd(B, vocab, ldv, tp.logit, targets + (size_t)t * B, pad_idx, gscale + t, dl, stream));
        if (!gtape) {
            TRY(sf_speaker_decoder_bwd(w, g, B, E, H, Tp, vocab, words + (size_t)t * B, h0, c0, ctx, &tp, dl, dh1, dc1, dho,
                                       dco, dctx, drop, step0 + t, ws, ws_bytes, stream));
        } else {
            // data gradients only; dY operands into the stacked gtape (sf_speaker_decoder_bwd with g = NULL)
            Arena ar = arena(ws, ws_bytes);
            const Dropout d_h = make_dropout(drop, 2 * (step0 + t) + 1, 2);
            float* dht = ar.take(BH);
            float* dh1d = ar.take(BH);
            float* dh1m = ar.take(BH);
            NEED(dht && dh1d && dh1m);
            TRY(spk_dlogit_to_dht(w, dl, ldv, B, H, vocab, dht, ar, st));
            TRY(softdot_bwd_i(&w->attn, nullptr, B, Tp, H, ctx, tp.alpha, tp.cat2, tp.t_text, tp.h_tilde, dht, dh1d, H, dctx,
                              ar, st, gtape->dpre + (size_t)t * BH, gtape->dt_text + (size_t)t * BH));
            // (the mask of dropout(h1) is applied to dh1d inside the LSTM pointwise backward: no copy launch)
            (void)dh1m;
            float* dgt = gtape->dgates + (size_t)t * B * 4 * H;
            if (!(g && g->embedding)) {
                TRY(lstm_bwd_i(&w->lstm, nullptr, B, E, H, tp.emb, E, h0, c0, tp.c1, tp.gates, dh1, dh1d, dc1, nullptr, 0,
                               dho, dco, ar, st, dgt, 0, &d_h));
            } else {
                SF_CHECK_ARG(tp.emb);
                float* demb = ar.take((size_t)B * E);
                NEED(demb);
                TRY(lstm_bwd_i(&w->lstm, nullptr, B, E, H, tp.emb, E, h0, c0, tp.c1, tp.gates, dh1, dh1d, dc1, demb, E, dho,
                               dco, ar, st, dgt, 0, &d_h));
                const Dropout de = (w->flags & SF_SPK_EMB_DROPOUT) ? make_dropout(drop, 2 * (step0 + t), 2)
                                                                   : make_dropout(nullptr, 0);
                TRY(embedding_bwd(demb, E, words + (size_t)t * B, 1, 1, B, E, -1, de, nullptr, g->embedding, st));
            }
        }
        dh1 = dho;
        dc1 = dco;
        std::swap(dho, dhn);
        std::swap(dco, dcn);
    }
    *result_in_b = (dh1 == dh_b) ? 1 : 0;
    if (gtape && g) {
        // every weight gradient as ONE product over the S*B stacked rows
        Arena ar = arena(ws, ws_bytes);
        const int M = S * B;
        if (g->w_out) TRY(gemm_tn(gtape->dlogit, ldv, tape0->h_tilde, H, M, vocab, H, g->w_out, H, 1, st, ar.rest(), ar.rest_n()));
        if (g->b_out) TRY(colsum(gtape->dlogit, ldv, M, vocab, g->b_out, 1, st, nullptr, ar.rest(), ar.rest_n()));
        if (g->attn.w_out) TRY(gemm_tn(gtape->dpre, H, tape0->cat2, 2 * H, M, H, 2 * H, g->attn.w_out, 2 * H, 1, st, ar.rest(), ar.rest_n()));
        if (g->attn.w_in) TRY(gemm_tn(gtape->dt_text, H, tape0->cat2 + H, 2 * H, M, H, H, g->attn.w_in, H, 1, st, ar.rest(), ar.rest_n()));
        if (g->lstm.w_ih) {
            SF_CHECK_ARG(tape0->emb);
            TRY(gemm_tn(gtape->dgates, 4 * H, tape0->emb, E, M, 4 * H, E, g->lstm.w_ih, E, 1, st, ar.rest(), ar.rest_n()));
        }
        if (g->lstm.w_hh) TRY(gemm_tn(gtape->dgates, 4 * H, h0_all, H, M, 4 * H, H, g->lstm.w_hh, H, 1, st, ar.rest(), ar.rest_n()));
        TRY(colsum_pair(gtape->dgates, 4 * H, M, 4 * H, g->lstm.b_ih, g->lstm.b_hh, ar, st));
    }
    return SF_OK;
}

// ---- TEACHER-FORCED speaker passes (round 5) ------------------------------------------------------------------------
// With teacher forcing the next input word is the target (speaker.py:166-167): the recurrence h_t = LSTM(emb(w_t), h_{t-1})
// does not depend on the attention, the vocabulary projection or the glue of any step.  So the S word steps split into
//   (1) the recurrence alone -- structurally the follower's EncoderLSTM with a given initial state: ONE persistent launch
//       (enc_persist_kernel, 4.5 us per step instead of 11 us for the full persistent word loop and ~50 us per-step), and
//   (2) everything else -- dropout(h1), attention over the path context, h~, vocabulary projection, log-soft-max / NLL /
//       score -- for all S*B rows AT ONCE: five products with M = S*B instead of 5 S launch-bound ones with M = B.
// The backward likewise: the head's backward for all rows at once, then enc_bwd_persist_kernel with the head's dh1 as the
// per-step external gradient.  Same arithmetic per element as the per-step entry points (sf_speaker_words_fwd / _bwd are
// the checkers in tests/test_gpu_speaker_teacher.py).
int sf_speaker_teacher_fwd(const sf_spk_decoder_w* w, int B, int E, int H, int Tp, int vocab, int S, int pad_idx,
                           int eos_idx, const int64_t* targets, float* hs_all, float* cs_all, const float* ctx,
                           const uint8_t* ctx_mask, int64_t* words, uint8_t* ended, float* step_scores, float* nll_term,
                           float* live, const sf_spk_decoder_tape* tape0, const sf_dropout* drop, uint32_t step0, void* ws,
                           size_t ws_bytes, sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(w && targets && hs_all && cs_all && ctx && words && ended && step_scores && nll_term && live && tape0 &&
                 tape0->gates && tape0->cat2 && tape0->t_text && tape0->alpha && tape0->h_tilde && tape0->logit && B > 0 &&
                 S > 0 && Tp > 0 && vocab > 0);
    if (!w->xw_table || (w->flags & SF_SPK_EMB_DROPOUT) || !encoder_persistent_supported(B, H, S)) return SF_ERR_UNSUPPORTED;
    Arena ar = arena(ws, ws_bytes);
    hipStream_t st = (hipStream_t)stream;        // (the parameter S shadows the S() cast helper here)
    const size_t BH = (size_t)B * H;
    const int M = S * B, ldv = (vocab + 3) & ~3;
    SF_CHECK_ARG(tape0->h1 == hs_all + BH && tape0->c1 == cs_all + BH);
    float* xchg = ar.take(encoder_persistent_xchg_floats(H));
    int* crow = reinterpret_cast<int*>(ar.take((size_t)M));
    NEED(xchg && crow && ar.tickets());
    // teacher forcing: words[t + 1] = targets[t] (speaker.py:167) -- known before the first step
    if (hipMemcpyAsync(words + B, targets, (size_t)M * sizeof(int64_t), hipMemcpyDeviceToDevice, st) != hipSuccess)
        return SF_ERR_LAUNCH;
    // (1) the recurrence: tokens words[t][b] (time-major: stride B per step), initial state = slot 0 of the tapes
    TRY(encoder_persistent(w->lstm.w_hh, w->lstm.b_ih, w->lstm.b_hh, w->xw_table, words, 1, nullptr, B, H, S,
                           tape0->gates, hs_all, cs_all, nullptr, make_dropout(nullptr, 0), xchg,
                           ar.tickets() + PERSIST_TICKET, st, nullptr, hs_all, cs_all, (long)B));
    if (tape0->emb) TRY(embedding_rows(w->embedding, E, words, M, tape0->emb, st));      // (only the backward's dW_ih reads them)
    // (2) the head over all S*B rows: dropout(h1) with the per-step sites 2 (step0 + t) + 1 (model.py:516) ...
    TRY(dropout_steps(hs_all + BH, H, S, B, H, tape0->cat2 + H, 2 * H, make_dropout(drop, 2 * step0 + 1, 2), 2, st));
    // ... attention over the path context of row m % B, h~ (model.py:517), vocabulary projection (:518)
    TRY(row_mod(crow, M, B, st));
    TRY(softdot_fwd_i(&w->attn, M, Tp, H, nullptr, 0, ctx, ctx_mask, tape0->h_tilde, tape0->alpha, tape0->cat2,
                      tape0->t_text, ar, st, crow));
    TRY(linear_plain(tape0->h_tilde, H, w->w_out, H, w->b_out, M, vocab, H, EPI_NONE, tape0->logit, ldv, ar, st));
    // glue of every step (speaker.py:163-191; teacher feedback: rows are independent, `ended` is an OR)
    return speaker_glue_fwd(M, vocab, ldv, tape0->logit, targets, 0, pad_idx, eos_idx, ended, words + B, step_scores,
                            nll_term, live, st, nullptr, B);
}

int sf_speaker_teacher_bwd(const sf_spk_decoder_w* w, const sf_spk_decoder_g* g, int B, int E, int H, int Tp, int vocab,
                           int S, int pad_idx, const int64_t* words, const int64_t* targets, const float* hs_all,
                           const float* cs_all, const float* ctx, const sf_spk_decoder_tape* tape0, const float* gscale,
                           float* dh_init, float* dc_init, float* dctx, const sf_dropout* drop, uint32_t step0,
                           const sf_spk_decoder_gtape* gtape, float* dcat2, float* ds, float* dh1_ext, void* ws,
                           size_t ws_bytes, sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(w && words && targets && hs_all && cs_all && ctx && tape0 && gscale && dh_init && dc_init && dctx && gtape &&
                 gtape->dlogit && gtape->dpre && gtape->dt_text && gtape->dgates && dcat2 && ds && dh1_ext && B > 0 && S > 0);
    if (!w->xw_table || (w->flags & SF_SPK_EMB_DROPOUT) || (g && g->embedding) || !encoder_persistent_supported(B, H, S) ||
        !ctx_grad_supported(S, Tp, H))
        return SF_ERR_UNSUPPORTED;
    Arena ar = arena(ws, ws_bytes);
    hipStream_t st = (hipStream_t)stream;        // (the parameter S shadows the S() cast helper here)
    const size_t BH = (size_t)B * H;
    const int M = S * B, ldv = (vocab + 3) & ~3;
    float* xchg = ar.take(encoder_bwd_persistent_xchg_floats());
    int* crow = reinterpret_cast<int*>(ar.take((size_t)M));
    NEED(xchg && crow && ar.tickets());
    // ---- the head's backward for all S*B rows at once
    TRY(softmax_ce_bwd(M, vocab, ldv, tape0->logit, targets, pad_idx, gscale, gtape->dlogit, st, B));
    TRY(spk_dlogit_to_dht(w, gtape->dlogit, ldv, M, H, vocab, gtape->dpre, ar, st));             // d h~ ...
    TRY(tanh_bwd(tape0->h_tilde, H, gtape->dpre, H, M, H, gtape->dpre, H, st));                  // ... -> d pre, in place
    TRY(row_mod(crow, M, B, st));
    // (dh1_ext receives d dropout(h1) = dcat2[:, H:] + dt W_in; the context gradient is deferred: dcat2 / ds)
    TRY(softdot_bwd_i(&w->attn, nullptr, M, Tp, H, ctx, tape0->alpha, tape0->cat2, tape0->t_text, tape0->h_tilde,
                      gtape->dpre, dh1_ext, H, nullptr, ar, st, gtape->dpre, gtape->dt_text, true, dcat2, ds, crow));
    TRY(ctx_grad_accum(tape0->alpha, ds, dcat2, 2 * H, tape0->t_text, S, B, Tp, H, dctx, st));
    // through dropout(h1): the forward's masks, in place
    TRY(dropout_steps(dh1_ext, H, S, B, H, dh1_ext, H, make_dropout(drop, 2 * step0 + 1, 2), 2, st));
    // ---- the recurrence's backward: all S steps in one persistent launch, dh1_ext[t] the external gradient of h_t
    TRY(encoder_bwd_persistent(w->lstm.w_hh, nullptr, B, H, S, tape0->gates, cs_all, dh1_ext, make_dropout(nullptr, 0),
                               nullptr, nullptr, gtape->dgates, xchg, ar.tickets() + PERSIST_TICKET, st, (long)H,
                               (long)BH, dc_init));
    // d h_init = dgates_0 W_hh
    TRY(data_grad(gtape->dgates, 4 * H, w->lstm.w_hh, w->lstm.w_hh_t, B, 4 * H, H, dh_init, H, 0, ar, st));
    if (g) {
        // every weight gradient as ONE product over the S*B stacked rows (as sf_speaker_words_bwd with a gtape)
        if (g->w_out) TRY(gemm_tn(gtape->dlogit, ldv, tape0->h_tilde, H, M, vocab, H, g->w_out, H, 1, st, ar.rest(), ar.rest_n()));
        if (g->b_out) TRY(colsum(gtape->dlogit, ldv, M, vocab, g->b_out, 1, st, nullptr, ar.rest(), ar.rest_n()));
        if (g->attn.w_out) TRY(gemm_tn(gtape->dpre, H, tape0->cat2, 2 * H, M, H, 2 * H, g->attn.w_out, 2 * H, 1, st, ar.rest(), ar.rest_n()));
        if (g->attn.w_in) TRY(gemm_tn(gtape->dt_text, H, tape0->cat2 + H, 2 * H, M, H, H, g->attn.w_in, H, 1, st, ar.rest(), ar.rest_n()));
        if (g->lstm.w_ih) {
            SF_CHECK_ARG(tape0->emb);
            TRY(gemm_tn(gtape->dgates, 4 * H, tape0->emb, E, M, 4 * H, E, g->lstm.w_ih, E, 1, st, ar.rest(), ar.rest_n()));
        }
        if (g->lstm.w_hh) TRY(gemm_tn(gtape->dgates, 4 * H, hs_all, H, M, 4 * H, H, g->lstm.w_hh, H, 1, st, ar.rest(), ar.rest_n()));
        TRY(colsum_pair(gtape->dgates, 4 * H, M, 4 * H, g->lstm.b_ih, g->lstm.b_hh, ar, st));
    }
    return SF_OK;
}

int sf_speaker_loss_finalize(const float* sum_cnt, const int64_t* words, int eos_idx, int T, int B, float* loss,
                             float* gscale, sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(sum_cnt && words && loss && gscale && T > 0 && B > 0);
    return speaker_loss_finalize(sum_cnt, words, eos_idx, T, B, loss, gscale, S(stream));
}

int sf_speaker_glue_bwd(int B, int vocab, int ldv, const float* logit, const int64_t* target,
                        int pad_idx, const float* gscale, float* dlogit, sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(logit && target && gscale && dlogit && B > 0 && vocab > 0 && ldv >= vocab);
    return softmax_ce_bwd(B, vocab, ldv, logit, target, pad_idx, gscale, dlogit, S(stream));
}

int sf_fill_f32(float* p, size_t n, float v, sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(p || n == 0);
    return fill(p, n, v, S(stream));
}

int sf_move_rows(const sf_row_move* moves, int n_moves, int n, sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(moves && n_moves >= 1 && n_moves <= SF_ROW_MOVES_MAX && n >= 0);
    if (n == 0) return SF_OK;
    RowMoves mv;
    mv.n = n_moves;
    for (int i = 0; i < n_moves; ++i) {
        const sf_row_move& m = moves[i];
        SF_CHECK_ARG(m.src && m.dst && m.idx && m.width > 0 && m.ld_src >= m.width && m.ld_dst >= m.width);
        mv.m[i] = RowMoves::M{m.src, m.dst, m.idx, m.ld_src, m.ld_dst, m.width, m.scatter ? 1 : 0};
    }
    return move_rows(mv, n, S(stream));
}

int sf_fill_regions(const sf_fill_region* regions, int n, sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(n >= 0 && n <= SF_FILL_MAX_REGIONS && (regions || n == 0));
    FillRegions fr;
    fr.n = 0;
    for (int i = 0; i < n; ++i) {
        const sf_fill_region& r = regions[i];
        SF_CHECK_ARG(r.width == 1 || r.width == 4 || r.width == 8);
        SF_CHECK_ARG(r.ptr || r.count == 0);
        SF_CHECK_ARG(((uintptr_t)r.ptr & (uintptr_t)(r.width - 1)) == 0);
        if (r.count == 0) continue;
        fr.r[fr.n++] = FillRegions::R{r.ptr, (unsigned long long)r.count, (unsigned long long)r.value, r.width};
    }
    if (fr.n == 0) return SF_OK;
    return fill_regions(fr, S(stream));
}

int sf_add_f32(float* dst, const float* src, size_t n, sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG((dst && src) || n == 0);
    if (n == 0) return SF_OK;
    return add2(dst, 0, src, 0, 1, (int)n, dst, 0, S(stream));
}

int sf_adam_step(float* p, const float* g, float* m, float* v, size_t n, double lr, double beta1,
                 double beta2, double eps, double weight_decay, int step, sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG((p && g && m && v) || n == 0);
    SF_CHECK_ARG(step >= 1 && beta1 >= 0. && beta1 < 1. && beta2 >= 0. && beta2 < 1.);
    if (n == 0) return SF_OK;
    return adam_step(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, S(stream));
}

int sf_adam_step_dev(float* p, const float* g, float* m, float* v, size_t n, double lr, double beta1, double beta2,
                     double eps, double weight_decay, int32_t* step_dev, float* coef, const uint32_t* skip_if_nonzero, sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(((p && g && m && v) || n == 0) && step_dev && coef);
    SF_CHECK_ARG(beta1 >= 0. && beta1 < 1. && beta2 >= 0. && beta2 < 1.);
    return adam_step_dev(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step_dev, coef, skip_if_nonzero, S(stream));
}

int sf_store_u32x4(uint32_t* dst, uint32_t a, uint32_t b, uint32_t c, uint32_t d, sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(dst != nullptr);
    return store_u32x4(dst, a, b, c, d, S(stream));
}

int sf_site_advance(uint32_t* word, uint32_t by, sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(word != nullptr);
    return site_advance(word, by, S(stream));
}

int sf_dropout_copy(const float* src, int lds, int B, int N, float* dst, int ldd,
                    const sf_dropout* drop, uint32_t drop_stream, int col0, sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(src && dst && B > 0 && N > 0);
    return dropout_copy(src, lds, B, N, dst, ldd, make_dropout(drop, drop_stream), col0, S(stream));
}

int sf_transpose(const float* src, int R, int Ccols, float* dst, sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(src && dst && R > 0 && Ccols > 0);
    return transpose(src, R, Ccols, dst, S(stream));
}

int sf_embedding_fwd(const float* table, int E, const int64_t* idx, int B, float* out,
                     sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(table && idx && out && B > 0 && E > 0);
    return embedding_rows(table, E, idx, B, out, S(stream));
}

// ---- in-process kernel timing ---------------------------------------------------------------------
int sf_profile_begin(void) {
    SF_ENTER();
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    if (g_prof.active) return SF_ERR_ARG;
    g_prof.recs.clear();
    g_prof.used = 0;
    g_prof.active = true;
    return SF_OK;
}

long sf_profile_end(char* buf, size_t cap) {
    SF_ENTER();
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    if (!g_prof.active) return -1;
    g_prof.active = false;
    struct Row { long calls = 0; double total = 0, mn = 1e30, mx = 0; };
    std::map<std::string, Row> rows;
    for (const ProfRec& r : g_prof.recs) {
        if (!r.e0 || !r.e1) return -1;
        if (hipEventSynchronize(r.e1) != hipSuccess) return -1;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) return -1;
        Row& w = rows[r.name];
        const double us = 1e3 * (double)ms;
        w.calls += 1; w.total += us; w.mn = std::min(w.mn, us); w.mx = std::max(w.mx, us);
    }
    g_prof.recs.clear();
    g_prof.used = 0;
    std::string out;
    char line[512];
    for (const auto& kv : rows) {
        snprintf(line, sizeof line, "%s\t%ld\t%.3f\t%.3f\t%.3f\n", kv.first.c_str(), kv.second.calls,
                 kv.second.total, kv.second.mn, kv.second.mx);
        out += line;
    }
    if (buf && cap) {
        const size_t n = std::min(cap - 1, out.size());
        std::copy(out.begin(), out.begin() + (long)n, buf);
        buf[n] = 0;
    }
    return (long)out.size() + 1;
}

}  // extern "C"
