/* sf_hip.h -- C ABI of libsf_hip.so: the speaker/follower hot path on MI355X (gfx950).
 *
 * The reference (ronghanghu/speaker_follower) has no C FFI for this path: its operator API is
 * the set of torch.nn.Module classes in tasks/R2R/model.py plus the per-step tensor glue in
 * tasks/R2R/follower.py and speaker.py.  Each entry point below replaces the arithmetic of one of
 * those functions (cited as file:line under /root/reference) and is what a binding written
 * against the reference would call -- see INTEGRATION.md for the ctypes stub.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless stated; fp32 row-major; leading dimensions in
 *     elements; index arrays int32 unless stated; masks uint8 (1 = masked / padded);
 *   - nothing allocates, frees or synchronises: outputs, saved activations and the scratch
 *     workspace are passed in; all work is enqueued on `stream` (a hipStream_t; NULL = default);
 *     calls are safe inside hipGraph stream capture;
 *   - return value: SF_OK or an SF_ERR_* code; no global state, thread-safe per stream as long as
 *     concurrent calls use different workspaces;
 *   - "accumulate" outputs (weight gradients, dctx) are read-modify-write: the caller zeroes them
 *     once per backward pass.  Gradient structs mirror the weight structs; a NULL member skips
 *     that gradient.
 *   - dropout is counter based: keep(seed, stream_id, global_row, col) -- see sf_dropout.
 */
#ifndef SF_HIP_H_
#define SF_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SF_ABI_VERSION 9

enum {
    SF_OK = 0,
    SF_ERR_ARG = 1,         /* null pointer, non-positive size, misaligned leading dimension */
    SF_ERR_UNSUPPORTED = 2, /* shape outside what the kernels are built for */
    SF_ERR_LAUNCH = 3,      /* hipGetLastError() reported a launch failure */
    SF_ERR_WORKSPACE = 4    /* workspace too small (see sf_workspace_bytes) */
};

typedef void* sf_stream;

/* nn.Dropout(p) replacement (model.py:52, 370, 414, 473).  p == 0 => eval mode.  The mask for
 * element (row, col) of dropout site `stream_id` is a pure function of (seed, stream_id,
 * row0 + row, col); oracle/rng.py mirrors it bit for bit. */
typedef struct sf_dropout {
    float p;
    uint32_t seed;
    int32_t row0; /* global id of local row 0 (data-parallel shards pass their offset) */
    /* Device-side site offset (ABI 8, optional: NULL = none).  A kernel forms its mask for site
     * `stream_id + site_mul * *site_dev`: the stream ids the host passes are then RELATIVE to a counter that lives in
     * device memory, and a hipGraph that captured a training iteration draws fresh masks on every replay once
     * sf_site_advance (captured at the end of the iteration) has moved the counter -- kernel arguments are frozen in a
     * graph, device memory is not.  Entry points that number their sites 2 * step_id + k internally (the decoder
     * steps) scale the offset by 2 themselves; site_mul (0 = 1) scales it for entry points that take a raw stream
     * id, so that host and device numbering agree: a replay draws exactly the masks of the eager iteration whose
     * host-side site counter equals the device word. */
    const uint32_t* site_dev;
    uint32_t site_mul;
} sf_dropout;
/* *word += by, on the stream (one thread): advances a device-side site counter (sf_dropout.site_dev,
 * sf_sample.stream_dev, sf_follower_glue.sample_site_dev). */
int sf_site_advance(uint32_t* word, uint32_t by, sf_stream stream);
/* dst[0..3] = a, b, c, d on the stream (values travel as kernel arguments: no host buffer to keep alive).  How the host
 * hands a replayed training graph its per-replay inputs -- site counter and optimizer step counters -- in ONE tiny
 * launch in front of the graph launch (runtime.TrainingGraph). */
int sf_store_u32x4(uint32_t* dst, uint32_t a, uint32_t b, uint32_t c, uint32_t d, sf_stream stream);

/* Panorama rows of a batch: either the dense [B,V,F] tensor the reference's
 * Seq2SeqAgent._feature_variables builds (follower.py:291-298, env.py:330-332, 771-773) or the
 * HBM-resident feature table addressed by index (replaces env.py:380-383 dict lookup + np.stack +
 * H2D copy): row v of sample b = table[vp[b], v, :] || loc_table[view[b], v, :]. */
typedef struct sf_pano {
    const float* dense;     /* [B,V,IMG+LOC] or NULL */
    const float* table;     /* [n_viewpoints, V, IMG] */
    const float* loc_table; /* [V, V, LOC]  (env.py:78-101) */
    const int32_t* vp;      /* [B]; < 0 => all-zero panorama (speaker.py:93-95 padded step) */
    const int32_t* view;    /* [B] */
    int32_t V, IMG, LOC;
} sf_pano;

/* Candidate-action rows: dense all_u_t [B,A,F] (follower.py:300-320) or by index
 * (env.py:60-75): row a>0 = table[vp[b], cand_view[b,a], :] || sin/cos(rel_heading, rel_elevation)
 * each repeated LOC/4 times; row 0 (stop) and rows >= a_num[b] are zero. */
typedef struct sf_cands {
    const float* dense;       /* [B,A,IMG+LOC] or NULL */
    const float* table;
    const int32_t* vp;        /* [B] */
    const int32_t* cand_view; /* [B,A] */
    const float* cand_sincos; /* [B,A,4]: sin h, cos h, sin e, cos e */
    const int32_t* a_num;     /* [B] */
    int32_t A, V, IMG, LOC;
} sf_cands;

/* ---- weights (pointer sets keyed like the reference state_dicts) -------------------------- */
/* Members ending in _t are OPTIONAL transposed copies (sf_transpose) of the weight of the same
 * name: with them every product on the path, forward and backward, is K-contiguous (288 GB of HBM
 * make a second layout of the hot weights free); NULL falls back to the slower in-place form.
 * Gradient structs mirror the weight structs member for member (the _t slots are ignored). */
typedef struct sf_lstm_w { /* nn.LSTMCell / nn.LSTM layer 0: weight_ih [4H,I], weight_hh [4H,H] */
    const float *w_ih, *w_hh, *b_ih, *b_hh;
    const float *w_ih_t, *w_hh_t; /* [I,4H], [H,4H] */
} sf_lstm_w;
typedef struct sf_lstm_g { float *w_ih, *w_hh, *b_ih, *b_hh, *unused0, *unused1; } sf_lstm_g;

typedef struct sf_visual_w { /* VisualSoftDotAttention model.py:303-308 */
    const float *w_h, *b_h; /* linear_in_h [D,H],[D] */
    const float *w_v, *b_v; /* linear_in_v [D,F],[D]; b_v cannot change the output (softmax shift) */
    const float *w_v_t;     /* [F,D] */
    const float *w_h_t;     /* [H,D] */
} sf_visual_w;
typedef struct sf_visual_g { float *w_h, *b_h, *w_v, *b_v, *unused0, *unused1; } sf_visual_g;

typedef struct sf_softdot_w { /* SoftDotAttention model.py:114-120 */
    const float *w_in;  /* linear_in  [H,H]  (no bias) */
    const float *w_out; /* linear_out [H,2H] (no bias) */
    const float *w_in_t, *w_out_t; /* [H,H], [2H,H] */
} sf_softdot_w;
typedef struct sf_softdot_g { float *w_in, *w_out, *unused0, *unused1; } sf_softdot_g;

typedef struct sf_scoring_w { /* EltwiseProdScoring model.py:335-340 */
    const float *w_h, *b_h;     /* linear_in_h [D,H],[D] */
    const float *w_a, *b_a;     /* linear_in_a [D,F],[D] */
    const float *w_out, *b_out; /* linear_out  [1,D],[1] */
    const float *w_a_t;         /* [F,D] */
    const float *w_h_t;         /* [H,D] */
} sf_scoring_w;
typedef struct sf_scoring_g { float *w_h, *b_h, *w_a, *b_a, *w_out, *b_out, *unused0, *unused1; } sf_scoring_g;

/* Optional INFERENCE-ONLY folding of consecutive Linears (built by sf_decoder_fold_build once per
 * weight version): the visual query q = W_v^T (W_h h + b_h) becomes ONE product q = M_v h + c_v, and
 * the scoring vector / constant r = W_a^T (w_out * (W_h h~ + b_h)), c = wt.b_a + b_out become ONE
 * product [r | c] = M_a h~ + c_a.  Two dependent stages fewer per decode step.  The backward needs
 * the unfolded intermediates, so training calls leave `fold` NULL. */
typedef struct sf_decoder_fold {
    const float* m_v; /* [F,H]   = W_v^T W_h */
    const float* c_v; /* [F]     = W_v^T b_h */
    const float* m_a; /* [F+4,H]: rows < F = W_a^T diag(w_out) W_h; row F = W_h^T (w_out*b_a); rest 0 */
    const float* c_a; /* [F+4]:   [< F] = W_a^T (w_out*b_h); [F] = (w_out*b_h).b_a + b_out; rest 0 */
} sf_decoder_fold;
typedef struct sf_decoder_w { /* AttnDecoderLSTM model.py:361-375 */
    sf_lstm_w lstm;       /* LSTMCell(2F -> H) */
    sf_visual_w visual;
    sf_softdot_w text;
    sf_scoring_w action;
    const sf_decoder_fold* fold; /* NULL = unfolded (required for training) */
} sf_decoder_w;
typedef struct sf_decoder_g { sf_lstm_g lstm; sf_visual_g visual; sf_softdot_g text; sf_scoring_g action; void* unused; } sf_decoder_g;

/* ---- saved activations of one AttnDecoderLSTM step (all written by fwd, read by bwd) ------- */
typedef struct sf_decoder_tape {
    float* t_v;     /* [B,D]   linear_in_h(h0) */
    float* q;       /* [B,F]   t_v folded through linear_in_v */
    float* alpha_v; /* [B,V] */
    float* xin;     /* [B,2F]  dropout(cat(u_prev, feature)) -- LSTM input */
    float* gates;   /* [B,4H]  activated gates i,f,g,o */
    float* c1;      /* [B,H] */
    float* h1;      /* [B,H]   (un-dropped, returned) */
    float* cat2;    /* [B,2H]  cat(weighted_ctx, dropout(h1)) -- linear_out input */
    float* t_text;  /* [B,H]   linear_in(dropout(h1)) */
    float* alpha;   /* [B,L] */
    float* h_tilde; /* [B,H] */
    float* t_a;     /* [B,D]   linear_in_h(h_tilde) (before the w_out product) */
    float* wt;      /* [B,D]   t_a * linear_out.weight */
    float* r;       /* [B,F]   wt folded through linear_in_a */
    float* logit;   /* [B,A]   raw logits (unmasked) */
} sf_decoder_tape;

/* Scratch any call may need (split-K partial slabs etc.); a constant, independent of batch. */
/* Size of the scratch buffer every composite entry point takes.  The caller ZERO-FILLS it once
 * after allocation (its last 4 KB hold self-maintaining ticket counters of multi-workgroup
 * kernels); after that the library owns its contents.  One workspace per concurrently used stream. */
size_t sf_workspace_bytes(void);
/* Recommended size of the scratch buffer handed to the WEIGHT-GRADIENT entry points (sf_attn_decoder_wgrad): with this
 * much they form the decoder's small weight gradients in three grouped launches (transposed operands + K-split slabs of
 * all of them at once); with sf_workspace_bytes() they fall back to one product at a time.  No zero-fill needed; a
 * buffer of its own -- not a bigger workspace for every call: the decode steps run 2 % slower over a 256 MB workspace. */
size_t sf_wgrad_workspace_bytes(void);
int sf_abi_version(void);
/* Hash (16 hex digits) of the sources, headers and compiler flags this library was built from
 * (speaker_follower_amd/build.py: build_id()).  The Python binding compares it with the sources on disk at import
 * and refuses a stale library. */
const char* sf_build_id(void);
const char* sf_status_string(int status);
/* hipGetErrorString of the HIP error behind the calling thread's last SF_ERR_LAUNCH */
const char* sf_last_error_string(void);

/* ---- nn.Linear (model.py:64, 99, 117-119, 306-307, 338-340, 419, 485) ------------------------
 * y[M,N] = act(x[M,K] w[N,K]^T + b);  act: 0 none, 1 tanh.  K % 4 == 0, ldx % 4 == 0. */
int sf_linear_fwd(const float* x, int ldx, const float* w, const float* b, int M, int N, int K,
                  int act, float* y, int ldy, void* ws, size_t ws_bytes, sf_stream stream);
/* The product alone, as split-K partial slabs (what the LSTM cell consumes: its pointwise kernel
 * adds the slabs, the biases and applies the gates -- model.py:393): x [M,K1] w [N,K1]^T +
 * h [M,K2] u [N,K2]^T (h, u may be NULL) -> *ksplit slabs [ksplit][M][N] at the START of the
 * workspace (ksplit >= 1 is returned; the caller sums them).  This is the entry point bench.py
 * times for its roofline object: exactly one launch of the gate-product kernel. */
int sf_linear_slabs_fwd(const float* x, int ldx, const float* w, int K1, const float* h, int ldh,
                        const float* u, int K2, int M, int N, int* ksplit, void* ws, size_t ws_bytes,
                        sf_stream stream);
/* dy is the gradient wrt the activation output; y is the saved output (needed for act = tanh).
 * dx [M,K] overwritten (accumulate_dx = 0) or added to; dw [N,K], db [N] accumulated. */
int sf_linear_bwd(const float* x, int ldx, const float* w, const float* y, int ldy,
                  const float* dy, int lddy, int M, int N, int K, int act, float* dx, int lddx,
                  int accumulate_dx, float* dw, float* db, void* ws, size_t ws_bytes,
                  sf_stream stream);

/* ---- nn.LSTMCell (model.py:371/393, 417-418/434, 483/515), gate order i,f,g,o -----------------
 * x [B,I] (I % 4 == 0).  Writes h1,c1 [B,H] and the activated gates [B,4H] for the backward.
 * If h1_drop != NULL also writes dropout(h1) there (row stride ld_h1_drop; site `drop_stream`). */
int sf_lstm_cell_fwd(const sf_lstm_w* w, int B, int I, int H, const float* x, int ldx,
                     const float* h0, const float* c0, float* h1, float* c1, float* gates,
                     float* h1_drop, int ld_h1_drop, const sf_dropout* drop, uint32_t drop_stream,
                     void* ws, size_t ws_bytes, sf_stream stream);
/* dh1,dc1 [B,H] in (NULL = zero); dx [B,I] (lddx), dh0, dc0 out (overwritten); weight grads
 * accumulated. */
int sf_lstm_cell_bwd(const sf_lstm_w* w, const sf_lstm_g* g, int B, int I, int H, const float* x,
                     int ldx, const float* h0, const float* c0, const float* c1,
                     const float* gates, const float* dh1, const float* dc1, float* dx, int lddx,
                     float* dh0, float* dc0, void* ws, size_t ws_bytes, sf_stream stream);

/* ---- VisualSoftDotAttention.forward (model.py:310-326) ---------------------------------------
 * out[b, :F] (row stride ldo) = sum_v alpha[b,v] X[b,v,:], alpha = softmax_v((W_v x_v + b_v).t),
 * t = W_h h + b_h.  Implemented as softmax_v(x_v . q), q = W_v^T t (b_v.t is a per-row constant and
 * cancels in the softmax), so X is read once and the [B*V,F]x[F,D] product disappears.
 * Optional dropout on `out` (site drop_stream, column offset drop_col0 -- the fused decoder writes
 * straight into the LSTM input buffer).  t_v [B,D] and q [B,F] are saved for the backward. */
int sf_visual_attention_fwd(const sf_visual_w* w, const sf_pano* X, int B, int H, int D,
                            const float* h, float* out, int ldo, float* alpha, float* t_v,
                            float* q, const sf_dropout* drop, uint32_t drop_stream,
                            int drop_col0, void* ws, size_t ws_bytes, sf_stream stream);
/* The same function with every intermediate rounded ONCE (round 5, csrc/sf_precise.hip): t = W_h h + b_h and
 * q = W_v^T t are formed on the float64 matrix cores (v_mfma_f64_16x16x4_f64) and kept in float64, the scores x_v . q
 * are accumulated in float64, the softmax sees them relative to their maximum.  What the speaker's path encoder runs
 * (sf_speaker_encoder_fwd): with the reference's "peaky" weights its scores reach +-80, where an fp32 evaluation of
 * the chain -- the reference's own included -- leaves 2e-6 .. 9e-6 of roundoff per stage in the softmax weights and
 * 1e-4 .. 3e-4 in the word logits behind the 7-step context.  t_v / q receive fp32 copies (the backward's tapes).
 * SF_ERR_UNSUPPORTED outside V in (18, 36], B <= 256, no transposed W_v copy: call sf_visual_attention_fwd. */
int sf_visual_attention_fwd_f64(const sf_visual_w* w, const sf_pano* X, int B, int H, int D,
                                const float* h, float* out, int ldo, float* alpha, float* t_v,
                                float* q, const sf_dropout* drop, uint32_t drop_stream,
                                int drop_col0, void* ws, size_t ws_bytes, sf_stream stream);
/* y = x W^T + b with float64 accumulation on the float64 matrix cores: x [M,K] fp32 (row stride ldx), W [N,K] fp32
 * (row stride ldw), b [N] or NULL; y64 [M,N] float64 and / or y32 [M,N] fp32 (either may be NULL).  K, ldx, ldw
 * multiples of 4. */
int sf_linear_f64(const float* x, int ldx, const float* w, int ldw, const float* b, int M, int N, int K,
                  double* y64, float* y32, sf_stream stream);
/* dout [B,F] (row stride lddo) is the gradient wrt the (dropped) output; dh [B,H] is ADDED to. */
int sf_visual_attention_bwd(const sf_visual_w* w, const sf_visual_g* g, const sf_pano* X, int B,
                            int H, int D, const float* h, const float* alpha, const float* t_v,
                            const float* dout, int lddo, const sf_dropout* drop,
                            uint32_t drop_stream, int drop_col0, float* dh, void* ws,
                            size_t ws_bytes, sf_stream stream);

/* ---- SoftDotAttention.forward (model.py:122-143) ---------------------------------------------
 * h [B,H] (row stride ldh); ctx [B,L,H]; mask [B,L] uint8 or NULL.  Writes alpha [B,L], h_tilde
 * [B,H] and the saved cat2 = [weighted_ctx ; h] [B,2H], t_text = linear_in(h) [B,H].
 * ctx_row (int32 [B] or NULL): sample b attends over ctx[ctx_row[b]] / mask[ctx_row[b]] -- the
 * search procedures' `ctx[beam_indices]` (follower.py:580, 790; speaker.py:252) without the copy.
 * Forward only: the backward entry points expect ctx_row == NULL in the forward pass. */
int sf_soft_dot_attention_fwd(const sf_softdot_w* w, int B, int L, int H, const float* h, int ldh,
                              const float* ctx, const uint8_t* mask, const int32_t* ctx_row,
                              float* h_tilde, float* alpha, float* cat2, float* t_text, void* ws,
                              size_t ws_bytes, sf_stream stream);
/* dh_tilde [B,H] in; dh [B,H] (row stride lddh) overwritten; dctx [B,L,H] ADDED to (NULL = skip). */
int sf_soft_dot_attention_bwd(const sf_softdot_w* w, const sf_softdot_g* g, int B, int L, int H,
                              const float* ctx, const float* alpha, const float* cat2,
                              const float* t_text, const float* h_tilde, const float* dh_tilde,
                              float* dh, int lddh, float* dctx, void* ws, size_t ws_bytes,
                              sf_stream stream);

/* ---- ContextOnlySoftDotAttention.forward behind its linear_in (model.py:166-177; SpeakerDecoderLSTM with
 * use_input_att_feed, model.py:500-503): attn = softmax_l(ctx[b,l,:] . t[b,:]) with mask[b,l] != 0 -> -inf,
 * wc[b,:] = sum_l attn[b,l] ctx[b,l,:].  ctx [B,L,H], mask [B,L] uint8 or NULL, t [B,H] (row stride ldt) = linear_in(h),
 * alpha [B,L], wc [B,H] (row stride ldwc). */
int sf_text_attention_fwd(const float* ctx, const uint8_t* mask, int B, int L, int H, const float* t, int ldt,
                          float* alpha, float* wc, int ldwc, sf_stream stream);
/* dwc [B,H] in; dt [B,H] out (overwritten); dctx [B,L,H] accumulated (NULL: not formed). */
int sf_text_attention_bwd(const float* ctx, int B, int L, int H, const float* dwc, int lddwc, const float* t, int ldt,
                          const float* alpha, float* dt, int lddt, float* dctx, sf_stream stream);

/* ---- EltwiseProdScoring.forward (model.py:342-352) -------------------------------------------
 * logit[b,a] = w_out . (t_a[b] * (W_a u[b,a] + b_a)) + b_out,  t_a = W_h h + b_h; implemented as
 * u[b,a] . r[b] + wt[b] . b_a + b_out with wt = w_out * t_a, r = W_a^T wt.  Saves t_a, wt [B,D], r. */
int sf_eltwise_prod_scoring_fwd(const sf_scoring_w* w, const sf_cands* U, int B, int H, int D,
                                const float* h, float* logit, float* t_a, float* wt, float* r,
                                void* ws, size_t ws_bytes, sf_stream stream);
/* dlogit [B,A] in; dh [B,H] overwritten. */
int sf_eltwise_prod_scoring_bwd(const sf_scoring_w* w, const sf_scoring_g* g, const sf_cands* U,
                                int B, int H, int D, const float* h, const float* t_a,
                                const float* wt, const float* dlogit, float* dh, void* ws,
                                size_t ws_bytes, sf_stream stream);

/* ---- follower per-step glue (follower.py:476-505) ---------------------------------------------
 * Masks logits of invalid candidates to -inf in place (valid = a < a_num[b], or is_valid[b,a] != 0
 * when is_valid is given), computes the cross-entropy term against target[b] (int64, -1 = ignore;
 * rows with ended[b] != 0 are forced to -1), chooses the next action (feedback 0 = teacher:
 * max(target,0); 1 = argmax, first maximum; 2 = sample from softmax with a counter-based uniform),
 * writes this step's score[b] = log p(a_t) (the caller sums steps: follower.py:504-505), writes
 * u_next[b,:] = U[b, a_t, :] (optionally through the NEXT step's input dropout, so that it can land
 * directly in that step's LSTM input buffer), updates ended[b] |= (a_t == 0).  ce_term[b] / live[b]
 * receive the row's CE term and 0/1 liveness; target_used the effective targets for the backward. */
typedef struct sf_follower_glue {
    const float* is_valid;    /* [B,A] or NULL */
    const int64_t* target;    /* [B] */
    int32_t feedback;
    uint8_t* ended;           /* [B] in/out */
    int64_t* a_t;             /* [B] out */
    int64_t* target_used;     /* [B] out */
    float* score;             /* [B] out */
    float* u_next;            /* [B, ld_u_next] out, or NULL */
    int32_t ld_u_next;
    const sf_dropout* u_drop; /* NULL = no dropout on u_next */
    uint32_t u_drop_stream;
    float* ce_term;           /* [B] out */
    float* live;              /* [B] out */
    uint32_t sample_seed, sample_stream; /* feedback 2 only */
    int32_t row0;                        /* global id of row 0 (sampling stream) */
    const uint32_t* sample_site_dev;     /* ABI 8, optional: device word added to sample_stream (sf_dropout.site_dev) */
    /* optional (scoring + glue launch only): env.step + env.observe + teacher of a device-resident
     * environment right behind the action choice, instead of a separate sf_nav_step launch */
    const struct sf_nav_io* nav;
} sf_follower_glue;
int sf_follower_glue_fwd(const sf_cands* U, int B, float* logit, const sf_follower_glue* glue,
                         sf_stream stream);
/* dlogit[b,a] = gscale[0] * (softmax(logit)[b,a] - [a == target_used[b]]) for live rows, else 0
 * (gscale = dloss / live count: a device scalar, so no host sync is needed). */
int sf_follower_glue_bwd(int B, int A, const float* logit, const int64_t* target_used,
                         const float* gscale, float* dlogit, sf_stream stream);

/* ---- AttnDecoderLSTM.forward (model.py:377-397): one follower decode step --------------------
 * u_prev [B,F] (NULL = tape->xin[:, :F] already holds dropout(u_prev), e.g. written by the
 * previous step's glue); X panorama; U candidates; h0,c0 [B,H]; ctx [B,L,H]; ctx_mask [B,L].
 * Returns h1,c1 (tape->h1, tape->c1), alpha (tape->alpha), logit (tape->logit), alpha_v.
 * With glue != NULL the per-step glue runs fused with the scoring kernel and tape->logit holds
 * the MASKED logits (what sf_follower_glue_bwd wants); otherwise the raw logits.
 * ctx_row: see sf_soft_dot_attention_fwd (search: many states share one instruction context).
 * Dropout sites: 2*step for the LSTM input, 2*step+1 for h1 (step = `step_id`). */
int sf_attn_decoder_fwd(const sf_decoder_w* w, const sf_pano* X, const sf_cands* U, int B, int H,
                        int D, int L, const float* u_prev, const float* h0, const float* c0,
                        const float* ctx, const uint8_t* ctx_mask, const int32_t* ctx_row,
                        const sf_decoder_tape* tape, const sf_follower_glue* glue,
                        const sf_dropout* drop, uint32_t step_id, void* ws, size_t ws_bytes,
                        sf_stream stream);
/* The same step, software-pipelined across steps (what a rollout over index-form observations
 * uses).  Given h1 of step t, the visual half of step t+1 (linear_in_h, W_v^T t, visual attention:
 * model.py:389 of the NEXT call) does not depend on the text attention / scoring half of step t,
 * so the two run side by side in paired launches:
 *   head(t):  t_v, q, visual attention           -> tape->xin[:, F:2F], alpha_v
 *   tail(t):  LSTMCell, text attention, scoring (+glue) of step t; if X_next != NULL also head(t+1)
 *             from h1 of step t into tape_next.
 * A rollout is head(0), tail(0; X_1), tail(1; X_2), ..., tail(S-1; NULL); results are identical to
 * S calls of sf_attn_decoder_fwd (same kernels, same tapes: the backward entry points apply). */
int sf_attn_decoder_head_fwd(const sf_decoder_w* w, const sf_pano* X, int B, int H, int D,
                             const float* h0, const sf_decoder_tape* tape, const sf_dropout* drop,
                             uint32_t step_id, void* ws, size_t ws_bytes, sf_stream stream);
/* When the next panorama depends on this step's action (a device-resident environment: nav.py), call the
 * tail with X_next = NULL and tape_next != NULL: only the QUERY of step t+1's visual attention (t_v', q':
 * functions of h1) is formed beside the text stages; after the environment step, sf_attn_decoder_attend_fwd
 * runs the attention of step t+1 from tape->q into tape->xin[:, F:2F] / tape->alpha_v (model.py:310-326). */
int sf_attn_decoder_attend_fwd(const sf_pano* X, int B, const sf_decoder_tape* tape, const sf_dropout* drop,
                               uint32_t step_id, void* ws, size_t ws_bytes, sf_stream stream);
int sf_attn_decoder_tail_fwd(const sf_decoder_w* w, const sf_cands* U, int B, int H, int D, int L,
                             const float* u_prev, const float* h0, const float* c0,
                             const float* ctx, const uint8_t* ctx_mask, const int32_t* ctx_row,
                             const sf_decoder_tape* tape, const sf_follower_glue* glue,
                             const sf_dropout* drop, uint32_t step_id, const sf_pano* X_next,
                             const sf_decoder_tape* tape_next, void* ws, size_t ws_bytes,
                             sf_stream stream);

/* Per-step gradient tape: the "dY" operands of every weight-gradient product of one decoder step.
 * BPTT over S steps keeps them stacked [S][B][..]; sf_attn_decoder_wgrad then forms each weight
 * gradient ONCE with reduction depth S*B instead of S read-modify-write passes over 12 M weights. */
typedef struct sf_decoder_gtape {
    float* dgates;  /* [B,4H] LSTM pre-activation gate gradients */
    float* dpre;    /* [B,H]  linear_out pre-activation gradient */
    float* dt_text; /* [B,H]  gradient of linear_in(h1_drop) */
    float* dt_v;    /* [B,D]  gradient of visual linear_in_h(h0) */
    float* dq;      /* [B,F]  gradient of the folded visual query */
    float* dwt;     /* [B,D]  gradient of wt = t_a * w_out */
    float* dta;     /* [B,D]  gradient of t_a */
    float* dr;      /* [B,F]  gradient of the folded scoring vector */
    float* dc;      /* [B]    gradient of the per-row scoring constant */
    /* optional (NULL = dctx is updated by every step's text-attention backward): with both given,
     * sf_follower_episode_bwd keeps each step's d[wc ; h1_drop] and d(score) here and adds
     * dctx += sum_t (alpha_t (x) dwc_t + ds_t (x) t_text_t) ONCE after the loop, instead of a
     * read-modify-write of the [B,L,H] gradient per step */
    float* dcat2;   /* [B,2H] gradient of [weighted context ; dropout(h1)] */
    float* ds;      /* [B,L]  gradient of the text-attention scores */
    /* optional (episode backward with a side stream): the attention path's share of d h1, per step */
    float* dh1d;    /* [B,H] */
} sf_decoder_gtape;
/* Builds the sf_decoder_fold matrices from the (transposed copies of the) decoder weights:
 * m_v [F,H], c_v [F], m_a [F+4,H], c_a [F+4] are caller-allocated device buffers. */
int sf_decoder_fold_build(const sf_decoder_w* w, int H, int D, int F, float* m_v, float* c_v,
                          float* m_a, float* c_a, void* ws, size_t ws_bytes, sf_stream stream);

/* Gradients in: dlogit [B,A], dh1, dc1 [B,H] (NULL = zero).  Out: dh0, dc0 [B,H] overwritten,
 * dctx [B,L,H] ADDED to.  u_prev is detached in the reference (follower.py:502): no du_prev.
 * g != NULL: weight gradients of this step are accumulated immediately; gtape != NULL: the dY
 * operands are saved there for a later sf_attn_decoder_wgrad (pass g = NULL then). */
int sf_attn_decoder_bwd(const sf_decoder_w* w, const sf_decoder_g* g, const sf_pano* X,
                        const sf_cands* U, int B, int H, int D, int L, const float* h0,
                        const float* c0, const float* ctx, const sf_decoder_tape* tape,
                        const sf_decoder_gtape* gtape, const float* dlogit, const float* dh1,
                        const float* dc1, float* dh0, float* dc0, float* dctx,
                        const sf_dropout* drop, uint32_t step_id, void* ws, size_t ws_bytes,
                        sf_stream stream);
/* Weight gradients of M = S*B stacked rows: `tape` / `gtape` point at step 0 of stacked
 * [S][B][..] tensors, h0_all [M,H] holds every step's incoming hidden state. */
int sf_attn_decoder_wgrad(const sf_decoder_w* w, const sf_decoder_g* g, int M, int H, int D, int F,
                          const float* h0_all, const sf_decoder_tape* tape,
                          const sf_decoder_gtape* gtape, void* ws, size_t ws_bytes,
                          sf_stream stream);

/* ---- a whole follower episode in one call (follower.py:430-539 forward, :1014-1016 backward) ----
 * All per-step tensors of an index-form (or dense) episode are stacked [S][...] arrays: `X`, `U`,
 * `tape`, `glue` hold the pointers of STEP 0 and step t lives t * (per-step size) further on (sizes
 * follow from S, B, H, D, L, A and the feature dims; tape.xin has S+1 steps, glue.ended is [B]).
 * glue.u_next / ld_u_next / u_drop / u_drop_stream / sample_stream are ignored: u_next of step t
 * always goes, through the next step's input dropout, into tape.xin of step t+1.
 * fwd  = sf_attn_decoder_head_fwd(0) then S x sf_attn_decoder_tail_fwd, with no host work between.
 *        glue.nav != NULL (sf_nav_io of STEP 0 of a device-resident environment, state buffers stacked
 *        [S + 1][...]): per step sf_attn_decoder_tail_fwd with tape_next but no X_next -- the environment steps
 *        inside its scoring + glue launch -- followed by sf_attn_decoder_attend_fwd over the panorama just chosen
 *        (needs w->visual.w_v_t, no folded weights).
 * bwd  = per step t = S-1..0: sf_follower_glue_bwd (gscale[t]) + sf_attn_decoder_bwd (g = NULL, dY
 *        operands into the stacked gtape), ping-ponging (dh, dc) between the two buffer pairs;
 *        *result_in_b tells which pair holds d h_init / d c_init; dctx [B,L,H] is ADDED to.
 *        Follow with sf_attn_decoder_wgrad. */
typedef struct sf_follower_episode {
    int32_t S, B, H, D, L, A;
    sf_pano X;
    sf_cands U;
    const float *h_init, *c_init; /* [B,H] */
    const float* ctx;             /* [B,L,H] */
    const uint8_t* ctx_mask;      /* [B,L] */
    sf_decoder_tape tape;
    sf_follower_glue glue;
    sf_dropout drop;              /* p == 0: no dropout */
    uint32_t step0;               /* dropout / sampling site of step 0 */
    /* optional, backward only: a second stream.  With it (and gtape.dcat2 / ds / dh1d) the scoring and
     * text-attention backward of step t-1 run on it while the LSTM / visual backward of step t runs on
     * `stream` (they only meet at the LSTM pointwise backward); the two are ordered with events and the
     * call leaves `stream` behind all of the side stream's work. */
    sf_stream side_stream;
    /* ABI 9, optional, INFERENCE ONLY (drop.p == 0, no backward through this pass: t_text / cat2[:H] / h_tilde of the
     * tape are NOT written; alpha is): two [B,L,H] scratch tensors.  With them sf_follower_episode_fwd forms
     * ctx_q = ctx W_in and ctx_o = ctx W_out[:, :H]^T ONCE (the context is constant over an episode) and runs the text
     * attention of every step but the last in folded form -- scores ctx_q[l] . h1, h~ = tanh(sum alpha_l ctx_o[l] +
     * W_out[:, H:] h1): the same function (fp32 re-association, like q = W_v^T t_v), two dependent launches fewer per
     * decode step.  Needs w->text.w_in_t and w->action.w_a_t; pre-drawn observations or a device-resident environment
     * (glue.nav).  With ctx_q / ctx_o, `side_stream` (if given) only carries the two fold products beside step 0's
     * attention (one fork, one join); the two-chain forward schedule is not taken. */
    float *ctx_q, *ctx_o;
    /* ... and, with the folded matrices of sf_decoder_fold (built once per weight version, sf_decoder_fold_build): the next
     * step's visual query as ONE product q' = M_v h1 + c_v beside the folded attention, the scoring vector / constant as
     * ONE product [r | c] = M_a h~ + c_a whose operand h~ the A-prologue forms -- THREE dependent launches behind the cell
     * (t_v, t_a, wt, r of the tape are not written either).  NULL: the four-launch chain.  `w->fold` stays NULL. */
    const sf_decoder_fold* chain_fold;
} sf_follower_episode;
int sf_follower_episode_fwd(const sf_decoder_w* w, const sf_follower_episode* e, void* ws,
                            size_t ws_bytes, sf_stream stream);
int sf_follower_episode_bwd(const sf_decoder_w* w, const sf_follower_episode* e,
                            const sf_decoder_gtape* gtape, const float* gscale, float* dlogit,
                            float* dh_a, float* dc_a, float* dh_b, float* dc_b, float* dctx,
                            int* result_in_b, void* ws, size_t ws_bytes, sf_stream stream);
/* The same over the decode steps [t_lo, t_hi) only (t_hi - 1 first): backpropagation through time in chunks,
 * so that the caller can start the weight-gradient products of the steps that are done (sf_attn_decoder_wgrad
 * over their stacked rows, on another stream) while the earlier steps are still being walked -- the slot is
 * `loss.backward()` of follower.py:1014-1018.  dh_in / dc_in: the gradient arriving from step t_hi (the
 * previous call's result; NULL when t_hi == S).  The call whose t_lo == 0 also forms the deferred context
 * gradient of the whole episode. */
int sf_follower_episode_bwd_range(const sf_decoder_w* w, const sf_follower_episode* e,
                                  const sf_decoder_gtape* gtape, const float* gscale, float* dlogit,
                                  float* dh_a, float* dc_a, float* dh_b, float* dc_b, float* dctx,
                                  int* result_in_b, int t_lo, int t_hi, const float* dh_in, const float* dc_in,
                                  void* ws, size_t ws_bytes, sf_stream stream);

/* Loss bookkeeping without host syncs or atomics (deterministic order):
 * sum_cnt[t] = (sum_b term[t,b], sum_b live[t,b]) for t < T;  then, optionally after a
 * data-parallel all-reduce of sum_cnt, loss[0] = sum_t sum/cnt (0 where cnt == 0) which is the
 * reference's "sum over steps of per-step means" (follower.py:481, speaker.py:182), and
 * gscale[t] = 1/cnt[t] (0 where cnt == 0) for the backward. */
int sf_reduce_terms(const float* term, const float* live, int T, int B, float* sum_cnt,
                    sf_stream stream);
int sf_loss_finalize(const float* sum_cnt, int T, float* loss, float* gscale, sf_stream stream);

/* ---- EncoderLSTM.forward (model.py:81-104) ----------------------------------------------------
 * seq [B,Lpad] int64 (PAD = 0 after each row's length), lengths [B] int32, T = max length.
 * Packed-sequence semantics: row b advances for t < lengths[b]; ctx [B,T,H] is zero beyond.
 * Writes ctx (dropout site `drop_stream` when training), decoder_init = tanh(W h_T + b), c_T.
 * tape: emb_t [T,B,E], xg [T,B,4H] (hoisted input product), gates [T,B,4H], hs [T+1,B,H],
 * cs [T+1,B,H] (state before/after every step).  An INFERENCE call (no backward to follow) with
 * w->xw_table may pass emb = NULL and gates = NULL: the embedded tokens and the gate tape are then not
 * written (hs / cs are still needed as working storage). */
typedef struct sf_encoder_w {
    const float* embedding; /* [vocab,E] */
    sf_lstm_w lstm;         /* weight_ih_l0 [4H,E] ... */
    const float *w_e2d, *b_e2d; /* encoder2decoder [H,H],[H] */
    const float *w_e2d_t;       /* optional [H,H] transposed */
    /* optional [vocab,4H] = embedding weight_ih_l0^T (refreshed by the host when either changes):
     * the input half of the gates of step t is then the table row of token seq[b,t] -- no
     * [T*B,E]x[E,4H] product, no [T,B,4H] intermediate.  NULL = the product is formed every call. */
    const float* xw_table;
    /* SF_ENC_* bits.  By default, when xw_table is given, H == 512, B <= 128, T <= 128 and the device
     * has >= 256 CUs, the T recurrent steps run as ONE persistent launch (csrc/sf_persist.hip: W_hh in
     * registers, the batch partitioned across the XCDs, h exchanged inside a partition); results are
     * bit-identical to the one-launch-per-step path for B > 16 (same summation order as its
     * 32-row kernel; within 2e-6 of its 16-row kernel below that).  SF_ENC_PER_STEP forces per-step. */
    int32_t flags;
} sf_encoder_w;
#define SF_ENC_PER_STEP 1
/* The embedding is TRAINABLE (glove=None, model.py:57-60): the reference then also applies its dropout module to
 * the embedded tokens (model.py:86-87), with the probability of `drop` at site drop_stream ^ 0x40000000, keyed on
 * (global row b, column t*E + e).  Needs tape->emb and tape->xg and xw_table == NULL (the input product of dropped
 * embeddings is not a table row). */
#define SF_ENC_EMB_DROPOUT 2
/* One DIRECTION of a bidirectional encoder (model.py:47-66, 92-94: nn.LSTM(bidirectional=True), hidden_size // 2 per
 * direction): the call neither applies encoder2decoder nor tanh -- decoder_init receives the raw h_T, and in the
 * backward d_init is the gradient wrt that raw h_T -- and ctx is written WITHOUT dropout (`drop` then only feeds
 * SF_ENC_EMB_DROPOUT).  The host mirror runs the two directions (the reverse one over per-row reversed tokens),
 * concatenates, drops, and applies encoder2decoder to [h_reverse ; h_forward] itself. */
#define SF_ENC_RAW_STATE 4
/* seq holds every row's tokens in REVERSED order (step t = position lengths[b] - 1 - t): the SF_ENC_EMB_DROPOUT mask is
 * keyed on the position, so that the two directions of a bidirectional encoder drop the same embedded tokens. */
#define SF_ENC_REVERSED 8
/* embedding (optional): gradient of embedding.weight [vocab,E], accumulated: row seq[b,t] += dropout-mask x
 * (dgates[t,b] W_ih); rows of token `padding_idx` receive nothing (nn.Embedding(padding_idx), model.py:55).  Needs
 * seq / Lpad as given to the forward. */
typedef struct sf_encoder_g {
    sf_lstm_g lstm;
    float *w_e2d, *b_e2d;
    float* embedding;
    const int64_t* seq;
    int32_t Lpad, padding_idx;
} sf_encoder_g;
typedef struct sf_encoder_tape { float *emb, *xg, *gates, *hs, *cs; } sf_encoder_tape;
int sf_encoder_lstm_fwd(const sf_encoder_w* w, int B, int Lpad, int T, int E, int H,
                        const int64_t* seq, const int32_t* lengths, float* ctx,
                        float* decoder_init, float* c_t, const sf_encoder_tape* tape,
                        const sf_dropout* drop, uint32_t drop_stream, void* ws, size_t ws_bytes,
                        sf_stream stream);
/* dctx [B,T,H] (gradient wrt the dropped ctx), d_init, d_ct [B,H] in (NULL = zero). */
int sf_encoder_lstm_bwd(const sf_encoder_w* w, const sf_encoder_g* g, int B, int T, int E, int H,
                        const int32_t* lengths, const float* decoder_init, const float* dctx,
                        const float* d_init, const float* d_ct, const sf_encoder_tape* tape,
                        const sf_dropout* drop, uint32_t drop_stream, void* ws, size_t ws_bytes,
                        sf_stream stream);

/* ---- batched feature gathers (env.py:380-383, 771-774, 60-75; follower.py:291-320) ------------
 * Materialise the dense tensors the reference builds on the host, from the HBM table. */
int sf_gather_panorama(const sf_pano* X, int B, float* out /* [B,V,F] */, sf_stream stream);
int sf_gather_candidates(const sf_cands* U, int B, float* all_u /* [B,A,F] */,
                         float* is_valid /* [B,A] */, sf_stream stream);
/* rows [B,F] of single chosen actions (speaker.py:104): a == 0 / vp < 0 => zeros */
int sf_gather_actions(const sf_cands* U, int B, const int32_t* a, float* out, sf_stream stream);
/* The same with a row stride on the output (a multiple of 4, >= F): the previous action's embedding straight into the
 * first half of a decoder step's LSTM input rows (sf_decoder_tape.xin, ld 2F; follower.py:588-590 `u_t_prev`), after which
 * sf_attn_decoder_fwd is called with u_prev = NULL -- one copy launch fewer per search step. */
int sf_gather_actions_ld(const sf_cands* U, int B, const int32_t* a, float* out, int ld_out, sf_stream stream);
/* The chosen-action embeddings of all N = Tp*B (path step, path) pairs of a speaker batch (speaker.py:87-104:
 * `ob['action_embedding'][a]`, zeros for a stop action or a padded step) in one launch: row n of `out` (row stride
 * ld_out floats >= IMG + LOC, so the rows can be the first half of the encoder's LSTM inputs) =
 * table[vp[n], act_view[n]] || [sin h]xg,[cos h]xg,[sin e]xg,[cos e]xg (g = LOC/4; env.py:60-75) where act[n] > 0 and
 * vp[n] >= 0, zeros elsewhere.  table [n_vp,V,IMG]; act_sincos [N,4]. */
int sf_gather_path_actions(const float* table, int V, int IMG, int LOC, const int32_t* vp, const int32_t* act_view,
                           const float* act_sincos, const int32_t* act, int N, float* out, int ld_out,
                           sf_stream stream);

/* ---- speaker (model.py:429-457, 487-519; speaker.py:158-197) ---------------------------------- */
typedef struct sf_spk_decoder_w {
    const float* embedding; /* [vocab,E] */
    sf_lstm_w lstm;         /* LSTMCell(E -> H) */
    sf_softdot_w attn;
    const float *w_out, *b_out; /* decoder2action [vocab,H],[vocab] */
    /* optional [vocab,4H] = embedding W_ih^T, refreshed by the host whenever either changes: the
     * input half of the LSTM gates becomes a row lookup by the previous word (model.py:497 + :515)
     * and the recurrent step is as short as the encoder's.  NULL = multiply every step. */
    const float* xw_table;
    /* SF_SPK_EMB_DROPOUT: the embedding is trainable (glove=None): model.py:499-500 applies dropout to the embedded
     * word in train mode -- site 2*step_id of `drop`; needs tape->emb and xw_table == NULL. */
    int32_t flags;
    /* optional, backward only: decoder2action^T as [H, ldv] (ldv = vocab rounded up to 4, padding columns zero): the
     * gradient wrt h~ = dlogit W_out becomes a K-contiguous product (NULL: the strided NN kernel, 26 us instead of 7). */
    const float* w_out_t;
} sf_spk_decoder_w;
#define SF_SPK_EMB_DROPOUT 1
/* embedding (optional): gradient of embedding.weight [vocab,E], accumulated: row prev_word[b] += mask x (dgates W_ih)
 * (no padding row: model.py:467 builds the speaker's nn.Embedding without padding_idx). */
typedef struct sf_spk_decoder_g { sf_lstm_g lstm; sf_softdot_g attn; float *w_out, *b_out; float* embedding; } sf_spk_decoder_g;
typedef struct sf_spk_decoder_tape {
    float *emb;     /* [B,E] */
    float *gates, *c1, *h1, *cat2, *t_text, *alpha, *h_tilde;
    float *logit;   /* [B,ldv] raw vocabulary logits, ldv = vocab rounded up to 4 */
} sf_spk_decoder_tape;
/* SpeakerDecoderLSTM.forward, non-att-feed branch (model.py:514-518).  prev_word [B] int64. */
int sf_speaker_decoder_fwd(const sf_spk_decoder_w* w, int B, int E, int H, int Tp, int vocab,
                           const int64_t* prev_word, const float* h0, const float* c0,
                           const float* ctx, const uint8_t* ctx_mask, const int32_t* ctx_row,
                           const sf_spk_decoder_tape* tape, const sf_dropout* drop,
                           uint32_t step_id, void* ws, size_t ws_bytes, sf_stream stream);
/* `sample` feedback of the speaker (speaker.py:170-174, D.Categorical(probs).sample()): counter-based draw keyed on
 * (seed, stream, row0 + b) -- csrc/sf_sampling.h; oracle/rng.py mirrors it.  `stream` names the word step (callers
 * pass site + t; sf_speaker_decode uses stream + t for its step t); row0 = global id of local row 0 (data-parallel
 * shards draw what the unsharded batch would).
 * LIMIT: the draw is two-level over at most 32 x 32 probabilities -- feedback 2 (`sample`) needs vocab <= 1024 in
 * sf_speaker_glue_fwd and sf_speaker_decode (SF_ERR_UNSUPPORTED above; the host classes raise NotImplementedError).
 * The live vocabularies fit (train_vocab.txt 991, sub_train_vocab.txt 935); trainval_vocab.txt (1 086) does not:
 * teacher / argmax passes over it run on the per-step entry points (tests/test_gpu_speaker.py). */
typedef struct sf_sample {
    uint32_t seed, stream;
    int32_t row0;
    const uint32_t* stream_dev; /* ABI 8, optional: device word added to `stream` (see sf_dropout.site_dev) -- a captured
                                 * `sample` pass draws new words on every replay */
} sf_sample;
/* The WHOLE word loop of an inference pass (speaker.py:158-197: S times SpeakerDecoderLSTM.forward +
 * sf_speaker_glue_fwd, eval mode, no tapes for a backward) as one persistent launch
 * (csrc/sf_persist.hip: weights register-resident, rows partitioned across XCDs, three in-kernel
 * exchanges per word).  targets [S,B] int64; words [S+1,B] with words[0] = the start tokens;
 * feedback 0 = teacher, 1 = argmax, 2 = sample (needs `sample`); step_scores / nll_term / live [S,B] as
 * sf_speaker_glue_fwd;
 * optional outputs (NULL = skip): logits [S,B,ldv], alpha [S,B,Tp], h1_tape / c1_tape [S,B,H].
 * Needs w->xw_table and w->attn.w_in_t.  The attention is evaluated in the folded form
 * cq = ctx W_in, cw = ctx W_c^T (same function, fp32 re-association).  SF_ERR_UNSUPPORTED (H != 512,
 * B > 128, Tp > 12, vocab > 1024, fewer than 256 CUs): run the per-step entry points instead. */
int sf_speaker_decode(const sf_spk_decoder_w* w, int B, int H, int Tp, int vocab, int S, int feedback,
                      int pad_idx, int eos_idx, const int64_t* targets, const float* h_init,
                      const float* c_init, const float* ctx, const uint8_t* ctx_mask, int64_t* words,
                      uint8_t* ended, float* step_scores, float* nll_term, float* live, float* logits,
                      float* alpha, float* h1_tape, float* c1_tape, const sf_sample* sample, void* ws,
                      size_t ws_bytes, sf_stream stream);
/* prev_word [B]: the words the forward embedded (only read when g->embedding is set; NULL otherwise). */
int sf_speaker_decoder_bwd(const sf_spk_decoder_w* w, const sf_spk_decoder_g* g, int B, int E,
                           int H, int Tp, int vocab, const int64_t* prev_word, const float* h0, const float* c0,
                           const float* ctx, const sf_spk_decoder_tape* tape, const float* dlogit,
                           const float* dh1, const float* dc1, float* dh0, float* dc0, float* dctx,
                           const sf_dropout* drop, uint32_t step_id, void* ws, size_t ws_bytes,
                           sf_stream stream);
/* speaker.py:163-191: log-softmax over the vocabulary, NLL terms against target (PAD ignored),
 * next word (feedback 0 = teacher, 1 = argmax, 2 = sample from softmax(logit): needs `sample`, else NULL),
 * score[b] = log p(w_t) (0 if w_t == PAD), ended[b] |= (w_t == EOS).  nll_term/live as in sf_follower_glue_fwd. */
int sf_speaker_glue_fwd(int B, int vocab, int ldv, const float* logit, const int64_t* target,
                        int feedback, int pad_idx, int eos_idx, uint8_t* ended, int64_t* w_t,
                        float* score, float* nll_term, float* live, const sf_sample* sample, sf_stream stream);
/* The speaker's loss from the per-step (NLL sum, live count) table (speaker.py:182, 192-197): the step means are added
 * only up to and including the first step at which EVERY row has produced EOS (the reference leaves its word loop
 * there); gscale[t] = 1 / count for those steps, 0 behind them.  words [T+1,B] as written by the glue / the decode
 * launch (row 0 = start tokens).  T <= 1024. */
int sf_speaker_loss_finalize(const float* sum_cnt, const int64_t* words, int eos_idx, int T, int B, float* loss,
                             float* gscale, sf_stream stream);
int sf_speaker_glue_bwd(int B, int vocab, int ldv, const float* logit, const int64_t* target,
                        int pad_idx, const float* gscale, float* dlogit, sf_stream stream);

/* ---- SpeakerEncoderLSTM.forward (model.py:437-457) in one call: for t in 0..Tp-1: visual attention over the panorama
 * of path step t with the previous hidden state (a1) -> [action embedding | attended feature] -> dropout -> LSTMCell (a2);
 * then decoder_init = tanh(encoder2decoder(h_Tp)).  X0 = the sf_pano of path step 0 of stacked [Tp][B] vp / view index
 * arrays; xin [Tp,B,2F] holds the action embeddings in its first halves on entry (sf_gather_path_actions) unless
 * act_emb [Tp,B,F] is given (train mode: they are dropped into xin here, site 2*(step0+t)); alpha [Tp,B,V], t_v
 * [Tp,B,D], q [Tp,B,F], gates [Tp,B,4H] are the backward's tape; hs, cs [Tp+1,B,H] with row 0 = zeros on entry.
 * ctx [B,Tp,H] (optional, eval mode only: h_t written straight into ctx[:, t]); h_init [B,H].
 * The loop this replaces issued 3 library calls per path step from Python. */
int sf_speaker_encoder_fwd(const sf_visual_w* vw, const sf_lstm_w* lw, const float* w_e2d, const float* b_e2d,
                           const sf_pano* X0, int Tp, int B, int H, int D, float* xin, float* alpha, float* t_v,
                           float* q, float* gates, float* hs, float* cs, float* ctx, const float* act_emb,
                           float* h_init, const sf_dropout* drop, uint32_t step0, void* ws, size_t ws_bytes,
                           sf_stream stream);
/* The visual attention's query through ONE float64 product (inference; ABI 8 addition): M_v = W_v^T W_h [F,H] and
 * c_v = W_v^T b_h [F], formed and kept in float64 by sf_visual_query_fold_f64 (needs w_v_t, w_h_t, b_h; redo it when a
 * weight changes), turn t_v = W_h h + b_h, q = W_v^T t_v (model.py:310-316) into q = M_v h + c_v: one dependent launch
 * fewer per path step and one rounding fewer.  sf_speaker_encoder_fwd_folded is sf_speaker_encoder_fwd with that query
 * (the t_v tape is not written: no backward can follow; `fold` must not be NULL). */
typedef struct sf_visual_fold64 {
    const double* m_v; /* [F,H] */
    const double* c_v; /* [F] */
} sf_visual_fold64;
int sf_visual_query_fold_f64(const sf_visual_w* w, int H, int D, int F, double* m_v, double* c_v, sf_stream stream);
int sf_speaker_encoder_fwd_folded(const sf_visual_fold64* fold, const sf_visual_w* vw, const sf_lstm_w* lw, const float* w_e2d,
                                  const float* b_e2d, const sf_pano* X0, int Tp, int B, int H, int D, float* xin, float* alpha,
                                  float* t_v, float* q, float* gates, float* hs, float* cs, float* ctx, const float* act_emb,
                                  float* h_init, const sf_dropout* drop, uint32_t step0, void* ws, size_t ws_bytes,
                                  sf_stream stream);

/* ---- the speaker's word loop with its tape, in one call each way (speaker.py:158-197 forward, the backward
 * `loss.backward()` of speaker.py:385 walks) -- what a TRAINING iteration runs (sf_speaker_decode keeps no tape):
 * fwd = for t in 0..S-1: sf_speaker_decoder_fwd(words[t], state t-1 -> tape[t], dropout / sampling site step0 + t) then
 *       sf_speaker_glue_fwd(tape[t].logit, targets[t] -> words[t+1], step_scores[t], nll_term[t], live[t]);
 * bwd = for t in S-1..0: sf_speaker_glue_bwd(gscale[t]) then sf_speaker_decoder_bwd, ping-ponging (dh, dc) between the
 *       two buffer pairs; *result_in_b tells which pair holds d h_init / d c_init; dctx [B,Tp,H] is ADDED to.
 * `tape0` holds the pointers of STEP 0 of stacked [S][B][..] tensors (emb may be NULL when nothing will run
 * backward); words [S+1,B] (row 0 = start tokens), targets / step_scores / nll_term / live [S,B], gscale [S].
 * gtape != NULL (needs h0_all [S,B,H]: every step's incoming hidden state, i.e. h_init followed by tape h1 of steps
 * 0..S-2 in ONE array): per step only the DATA gradients are formed and the dY operands kept in the stacked gtape;
 * every weight gradient of `g` is then ONE product over all S*B rows at the end (reduction depth 8 000 instead of
 * 80 products of depth 100 each) -- the follower's scheme (sf_attn_decoder_wgrad).  A trainable embedding
 * (g->embedding) still scatters per step.  dlogit [B,ldv] is scratch and may be NULL with a gtape. */
typedef struct sf_spk_decoder_gtape {
    float *dlogit;   /* [S,B,ldv] */
    float *dpre;     /* [S,B,H]  d(pre-tanh) of attention.linear_out */
    float *dt_text;  /* [S,B,H]  d(linear_in output) */
    float *dgates;   /* [S,B,4H] pre-activation gate gradients */
} sf_spk_decoder_gtape;
int sf_speaker_words_fwd(const sf_spk_decoder_w* w, int B, int E, int H, int Tp, int vocab, int S, int feedback,
                         int pad_idx, int eos_idx, const int64_t* targets, const float* h_init, const float* c_init,
                         const float* ctx, const uint8_t* ctx_mask, int64_t* words, uint8_t* ended, float* step_scores,
                         float* nll_term, float* live, const sf_spk_decoder_tape* tape0, const sf_dropout* drop,
                         uint32_t step0, const sf_sample* sample, void* ws, size_t ws_bytes, sf_stream stream);
int sf_speaker_words_bwd(const sf_spk_decoder_w* w, const sf_spk_decoder_g* g, int B, int E, int H, int Tp, int vocab,
                         int S, int pad_idx, const int64_t* words, const int64_t* targets, const float* h_init,
                         const float* c_init, const float* ctx, const sf_spk_decoder_tape* tape0, const float* gscale,
                         float* dlogit, float* dh_a, float* dc_a, float* dh_b, float* dc_b, float* dctx,
                         int* result_in_b, const sf_dropout* drop, uint32_t step0, const sf_spk_decoder_gtape* gtape,
                         const float* h0_all, void* ws, size_t ws_bytes, sf_stream stream);
/* TEACHER-FORCED pass over S word steps (speaker.py:158-197 with feedback = teacher; ABI 8).  The next input word is
 * the target, so the recurrence does not depend on attention / projection / glue: it runs alone as ONE persistent
 * launch (the encoder's recurrence kernel with a given initial state) and everything else -- dropout(h1), attention over
 * the path context, h~, vocabulary projection, log-soft-max / NLL / score (model.py:516-518, speaker.py:163-191) -- runs
 * for all S*B rows at once.  hs_all / cs_all [S+1,B,H]: slot 0 = the initial state (in), slots 1..S = h1 / c1 of the
 * steps (out; tape0->h1 / c1 must point at slot 1); tape0->gates / cat2 / t_text / alpha / h_tilde / logit are the
 * stacked [S,B,...] tapes of sf_speaker_decoder_fwd (emb optional: all S*B embedded words, for the backward's dW_ih);
 * words [S+1,B] receives words[t+1] = targets[t]; ended / step_scores / nll_term / live as sf_speaker_glue_fwd.
 * Needs w->xw_table, no embedding dropout.  SF_ERR_UNSUPPORTED (H != 512, B > 128, S > 128, < 256 CUs): run
 * sf_speaker_words_fwd. */
int sf_speaker_teacher_fwd(const sf_spk_decoder_w* w, int B, int E, int H, int Tp, int vocab, int S, int pad_idx,
                           int eos_idx, const int64_t* targets, float* hs_all, float* cs_all, const float* ctx,
                           const uint8_t* ctx_mask, int64_t* words, uint8_t* ended, float* step_scores, float* nll_term,
                           float* live, const sf_spk_decoder_tape* tape0, const sf_dropout* drop, uint32_t step0, void* ws,
                           size_t ws_bytes, sf_stream stream);
/* Its backward: the head's backward for all S*B rows at once, the recurrence's backward as one persistent launch with the
 * head's d h1 as per-step external gradient, every weight gradient as ONE product over the stacked rows (g may be NULL:
 * data gradients only).  gscale [S]; dh_init / dc_init [B,H] out (gradient wrt slot 0 of hs_all / cs_all); dctx
 * [B,Tp,H] accumulated; gtape as sf_speaker_words_bwd; dcat2 [S*B,2H], ds [S*B,Tp], dh1_ext [S*B,H]: caller-owned
 * scratch.  SF_ERR_UNSUPPORTED as the forward (and with a trainable embedding): run sf_speaker_words_bwd. */
int sf_speaker_teacher_bwd(const sf_spk_decoder_w* w, const sf_spk_decoder_g* g, int B, int E, int H, int Tp, int vocab,
                           int S, int pad_idx, const int64_t* words, const int64_t* targets, const float* hs_all,
                           const float* cs_all, const float* ctx, const sf_spk_decoder_tape* tape0, const float* gscale,
                           float* dh_init, float* dc_init, float* dctx, const sf_dropout* drop, uint32_t step0,
                           const sf_spk_decoder_gtape* gtape, float* dcat2, float* ds, float* dh1_ext, void* ws,
                           size_t ws_bytes, sf_stream stream);

/* ---- search helpers (follower.py:541-980 beam / state-factored search, speaker.py:211-318) ------
 * dst[i, :width] = src[idx[i], :width] (idx < 0 => zeros): `h_t[flat_indices]`, `c_t[flat_indices]`
 * (follower.py:580, speaker.py:252) as one gather; width, ld_src, ld_dst multiples of 4. */
int sf_gather_rows(const float* src, int ld_src, const int32_t* idx, int n, int width, float* dst,
                   int ld_dst, sf_stream stream);
/* follower.py:585-603 / speaker.py:254-257 on the device: per row of logit [N, ld] (n columns,
 * n <= 1024): columns >= n_valid[row] are set to -inf in place when n_valid is given
 * (`logit[is_valid == 0] = -inf`), then log_softmax, then the k best columns in descending order
 * (ties: lower column first): idx [N,k] int32, logp [N,k] = log_softmax(logit)[row, idx].  k == n
 * returns the whole row sorted (follower.py:802).  idx == NULL (k == n required): no selection -- logp [N,n] =
 * log_softmax(logit) in COLUMN order, -inf beyond n_valid: what state_factored_search consumes (it walks every
 * successor of a state, follower.py:802-836). */
int sf_logprob_topk(float* logit, int ld, int N, int n, const int32_t* n_valid, int k, int32_t* idx,
                    float* logp, sf_stream stream);
/* dst[idx[i], :width] = src[i, :width] (idx < 0: row i skipped): the h / c / attention rows of newly expanded
 * search states into the state pool (the reference keeps them on its InferenceState tuples, follower.py:826-836);
 * width, ld_src, ld_dst multiples of 4; the idx >= 0 must be distinct. */
int sf_scatter_rows(const float* src, int ld_src, const int32_t* idx, int n, int width, float* dst,
                    int ld_dst, sf_stream stream);

/* Up to SF_ROW_MOVES_MAX row gathers (scatter = 0: dst[i, :width] = src[idx[i], :width], idx < 0 => zeros) and row
 * scatters (scatter = 1: dst[idx[i], :width] = src[i, :width], idx < 0 => row skipped; the idx >= 0 distinct) over the
 * same n rows in ONE launch: `h_t[flat_indices]` and `c_t[flat_indices]` of a search step (follower.py:588-589, 826-827)
 * together, and the h / c / attention rows of its new states into the state pool together.  width, ld_src, ld_dst
 * multiples of 4.  The moves must not overlap one another. */
#define SF_ROW_MOVES_MAX 4
typedef struct sf_row_move {
    const float* src;
    float* dst;
    const int32_t* idx;
    int32_t ld_src, ld_dst, width;
    int32_t scatter;
} sf_row_move;
int sf_move_rows(const sf_row_move* moves, int n_moves, int n, sf_stream stream);

/* Small utilities used by the host mirror (kept on the stream so rollouts never sync). */
int sf_fill_f32(float* p, size_t n, float v, sf_stream stream);
int sf_add_f32(float* dst, const float* src, size_t n, sf_stream stream); /* dst += src */
/* Up to SF_FILL_MAX_REGIONS buffers set to a constant each in ONE launch: the initial conditions of a pass -- zero
 * states (model.py:67-79 init_state, :368 u_begin), the <BOS> word of every row (speaker.py:137), cleared `ended`
 * flags (follower.py:380, speaker.py:136) -- which a host mirror would otherwise issue as one fill per tensor.
 * `count` elements of `width` bytes (1, 4 or 8; ptr aligned to it) are set to the low `width` bytes of `value`
 * (a float travels as its bit pattern).  Regions with count 0 are skipped. */
#define SF_FILL_MAX_REGIONS 8
typedef struct sf_fill_region {
    void* ptr;
    uint64_t count;
    uint64_t value;
    int32_t width;
} sf_fill_region;
int sf_fill_regions(const sf_fill_region* regions, int n, sf_stream stream);
/* dst[b, :N] (row stride ldd) = dropout(src[b, :N]) at site `drop_stream`, columns col0.. */
int sf_dropout_copy(const float* src, int lds, int B, int N, float* dst, int ldd,
                    const sf_dropout* drop, uint32_t drop_stream, int col0, sf_stream stream);
/* dst[C,R] = src[R,C]^T -- builds the transposed weight copies (refresh after each optimizer step) */
int sf_transpose(const float* src, int R, int Ccols, float* dst, sf_stream stream);
/* embedding rows: out[b,:] = table[idx[b],:]  (model.py:497) */
int sf_embedding_fwd(const float* table, int E, const int64_t* idx, int B, float* out,
                     sf_stream stream);

/* ---- a13 the optimizer step (train.py:263-268: optim.Adam(lr=1e-4, weight_decay=5e-4) for the
 * encoder and for the decoder; follower.py:1014-1020 / speaker.py:389-395 call .step() after
 * loss.backward()).  torch.optim.Adam semantics (L2 weight decay added to the gradient, bias
 * correction with the 1-based `step`, no amsgrad) over ONE flat range of n fp32 parameters:
 * p, m (exp_avg), v (exp_avg_sq) are updated in place, g is read.  Hyper-parameters are doubles (as
 * in Python): 1 - beta and the bias corrections are formed in double and rounded to float once.  One launch for a whole
 * optimizer when its parameters, gradients and moments are laid out flat (optim.FusedAdam). */
int sf_adam_step(float* p, const float* g, float* m, float* v, size_t n, double lr, double beta1,
                 double beta2, double eps, double weight_decay, int step, sf_stream stream);
/* The same step with its 1-based step counter IN DEVICE MEMORY (ABI 8): `*step_dev` is incremented by the call and the
 * bias corrections are formed from it on the device (double, rounded once, as on the host), so that a hipGraph which
 * captured the call performs step n + 1 on its n-th replay.  coef: 4 floats of device scratch owned by the optimizer.
 * skip_if_nonzero (optional): a device word read on the stream; non-zero turns the call into a no-op (parameters,
 * moments and `*step_dev` untouched).  Given the fault word of the persistent launches (sf_workspace_fault_offset) a
 * captured iteration never steps on gradients a starved launch has poisoned; the host sees the word at its next
 * sync and re-issues the iteration (agents.Seq2SeqAgent.train). */
int sf_adam_step_dev(float* p, const float* g, float* m, float* v, size_t n, double lr, double beta1, double beta2,
                     double eps, double weight_decay, int32_t* step_dev, float* coef, const uint32_t* skip_if_nonzero,
                     sf_stream stream);

/* Development aid (no reference counterpart): while `buf` is non-null, the visual-attention body of
 * the pipelined decode step stamps wall_clock64() (100 MHz) per workgroup into buf[block * 8 + k]
 * (k = 0 start, 1 rows loaded and scored, 2 partials stored); buf = device memory of >= 512 * 8
 * uint64, NULL switches it off.  Used by tools/vis_trace.py to read a kernel's inner timeline. */
void sf_debug_trace(unsigned long long* buf);
/* Development aid: on != 0 makes the persistent launches (csrc/sf_persist.hip) use their
 * placement-independent exchange -- write-through (sc1) stores -- even when a row group's workgroups
 * share an XCD and would keep the exchange inside that XCD's L2.  Lets the tests exercise the protocol
 * the kernels fall back to when the observed workgroup -> XCD placement does not hold. */
void sf_debug_force_write_through(int on);
/* Development aid / test hook: bound of every in-kernel wait of the persistent launches in ticks of 10 ns
 * (< 0 restores the default, 0.25 s).  0 makes the first unsatisfied poll give up: the launch poisons its outputs
 * and raises its fault bit -- how the tests exercise the host's fallback to the per-step kernels. */
void sf_debug_persist_timeout(long long ticks);
/* Test co-tenant: `blocks` workgroups of `threads` threads with `lds_bytes` (<= 65536) of LDS each that stay resident
 * for `ticks` x 10 ns, touching LDS and (sink != NULL: [blocks] floats) a little global memory -- the shape of a
 * collective's channel workgroups (RCCL: a few dozen workgroups of 256-512 threads) -- to be launched on ANOTHER stream
 * beside the persistent launches (tests/test_gpu_cotenancy.py: data-parallel training launches the first gradient
 * bucket's all-reduce beside the persistent encoder backward). */
int sf_debug_cotenant(int blocks, int threads, int lds_bytes, long long ticks, float* sink, sf_stream stream);
/* STRICT summation order for the LSTM gate products (supported runtime switch, round 5).  on != 0: the large gate
 * products (K >= 2048, M <= 128: sf_lstm_cell_fwd, the decode step, the search step) run on the fp32 MFMA
 * (v_mfma_f32_16x16x4_f32, the kernel of rounds 1-3) instead of the bf16 matrix cores with three-way error-free operand
 * splitting.  Both are fp32-accurate (the split form is measured closer to the exact sum); they differ in the last
 * bit or two, and a best-first search that meets two frontier states whose scores are EQUAL to that last bit expands
 * them in the order the arithmetic happens to give.  With the strict order the state-factored search walks the
 * reference's own traversal for 64 of 64 instructions of golden G7b (tests/test_gpu_search.py); the default order
 * re-orders one exact tie (identical completions, order and scores either way).  5.5 us per decode step slower.
 * Process-wide; a captured hipGraph keeps the kernels it was captured with (search.graph_step_for keys its cache on
 * this switch).  sf_gate_product_is_strict() reads it back. */
void sf_gate_product_strict(int on);
int sf_gate_product_is_strict(void);
/* Older name of the same switch (tools/, A/B timing): on != 0 runs the large LSTM gate products (K >= 2048, M <= 128: sf_lstm_cell_fwd, the decode
 * step) on the fp32 MFMA (v_mfma_f32_16x16x4_f32, rounds 1-3) instead of the bf16 matrix cores with three-way
 * error-free operand splitting (csrc/sf_gemm.hip: gemm_nt_split_kernel; same fp32 accuracy class, measured closer
 * to the exact sum, 6/16 of the matrix-pipe time).  For A/B timing and for the accuracy tests. */
void sf_debug_gate_product_f32(int on);
/* folded inference chain (ABI 9): 1 (default) = attention partials beside r and their merge beside scoring + glue;
 * 0 = partials, ticket and merge in one launch beside r */
void sf_debug_fold_merge_with_glue(int on);
/* 0: the four-launch folded chain even when sf_follower_episode.chain_fold is given (A/B switch) */
void sf_debug_fold_chain3(int on);
/* 1 (default): with sf_follower_episode.side_stream the two fold products run on it beside step 0's attention */
void sf_debug_fold_build_overlap(int on);
/* A/B switch (round 5): on == 0 sends the many-row products (M >= 512: the speaker's teacher-forced head over all S*B rows,
 * the beam search's flat steps) back to the register-streaming kernel of rounds 1-4 instead of the LDS-tiled 128 x 128
 * bf16x6 kernel (csrc/sf_gemm.hip: gemm_nt_big_kernel; the default).  Bit 1 of `on` (on == 3) keeps the kernel but turns off
 * the K splits it takes where the consumer sums slabs anyway and the tile count wastes a round of the CUs (the beam
 * step's gate product: 320 tiles -> 3 splits). */
void sf_debug_many_row_product(int on);
/* A/B switch (round 5): on == 0 forms the decoder's small weight gradients one product at a time (two transposes, the
 * many-row product, the slab sum: four dependent launches each) instead of three grouped launches for all of them
 * (csrc/sf_gemm.hip: gemm_tn_group; the default). */
void sf_debug_grouped_weight_gradients(int on);
/* A/B switch (round 5): on == 0 puts the slab-sum launch back between the LSTM's data gradient and the visual-attention
 * backward of a decoder step (the default: the attention backward adds up the K-split slabs of d(feature) itself --
 * same order, same bits). */
void sf_debug_slab_consumers(int on);
/* Experiment switch (round 5): on != 0 runs the LSTM cell's pointwise backward of a decoder step as the epilogue of the
 * small product that completes that step's dh1 (the last launch of the backward step before it) instead of its own
 * launch: same arithmetic, same bits (tested), one launch fewer per step -- and no faster (4.81 vs 4.79 ms). */
void sf_debug_fused_cell_backward(int on);
/* Experiment switch: the two-stream backward through time issues the head of step t - steps right before the tail of
 * step t (steps >= 1) instead of every head first (steps < 0, the default). */
void sf_debug_bptt_lookahead(int steps);
/* Experiment switch (round 5): on != 0 orders the two chains of the two-stream backward through time (heads: scoring /
 * text attention; tails: LSTM / visual attention) with one-shot device flags instead of events (a flag wait that gives
 * up raises bit 16 of the fault word).  Measured equal. */
void sf_debug_bptt_flags(int on);
/* Experiment switch (tools/bptt_overlap_probe.py): 1 = the two-stream backward issues its heads only, 2 = its tails only
 * (no waits): the two chains as separately captured graphs. */
void sf_debug_bptt_part(int part);
/* A/B switch: on == 0 makes sf_speaker_encoder_fwd run its visual attention on the fp32 kernels (rounds 1-4) instead
 * of the float64 query / score path (sf_visual_attention_fwd_f64; the default). */
void sf_debug_precise_attention(int on);
/* Test switch: the weight-gradient products dW += dY^T X whose shape is a whole number of 128 x 128 tiles run as bf16x6
 * split products (gemm_tn_split_kernel) from `rows` reduction rows on (default 4096: where it is faster than the
 * fp32-MFMA kernels; rows < 0 restores the default). */
void sf_debug_tn_split_min_rows(int rows);
/* Byte offset, inside a workspace of `ws_bytes` bytes, of the FAULT WORD (uint32, zero in a healthy process): a
 * persistent launch whose bounded wait gave up (co-residency lost to another process) ORs its bit into it -- 1 encoder
 * forward (sf_encoder_lstm_fwd), 2 encoder backward, 4 speaker word loop (sf_speaker_decode), 8 device-wide lock not
 * obtained -- and poisons its outputs with NaN.  The library never reads or clears the word: the host reads it at a
 * sync it already has, clears it, and re-issues the pass with the per-step kernels (SF_ENC_PER_STEP / the per-step
 * speaker entry points).  speaker_follower_amd.runtime.take_fault does exactly that. */
size_t sf_workspace_fault_offset(size_t ws_bytes);
/* ---- device-resident navigation (env.py:126-146 step, :149-224 panorama sweep, :742-761 teacher,
 * :763-804 observe) ---------------------------------------------------------------------------------
 * The candidate list of a state is a pure function of (viewpoint, view index): the host tabulates it
 * once per connectivity graph (speaker_follower_amd/nav.py) over its own contiguous "nav rows" (one
 * per viewpoint, scans back to back).  State s = nav_row * V + view; candidate a of state s leads to
 * nav row next_row[s,a] facing view cand_view[s,a] (a = 0: stop, next_row = the state's own row). */
typedef struct sf_nav_table {
    const int32_t* a_num;       /* [n_rows*V]      1 + number of neighbours */
    const int32_t* next_row;    /* [n_rows*V, A] */
    const int32_t* cand_view;   /* [n_rows*V, A]   absViewIndex of the candidate = the view after the move */
    const float* cand_sincos;   /* [n_rows*V, A,4] sin/cos of rel_heading, rel_elevation */
    const int32_t* feat_row;    /* [n_rows]        feature-table row of a nav row */
    int32_t A, V;
} sf_nav_table;
/* One env.step + env.observe (+ teacher) for a batch: from state (row[b], view[b]) and the chosen
 * candidate a_t[b] (NULL = no move: the initial observation) to the next state, written as the NEXT
 * decode step's index-form observation (what sf_pano / sf_cands / sf_follower_glue read): row_next,
 * vp_next (feature rows), view_next, a_num_next [B], cand_view_next [B,A], sincos_next [B,A,4] and,
 * when target_next != NULL, the shortest-path teacher action (-1 for rows with ended[b] != 0): the
 * candidate whose next_row equals goal_hop[b, row - hop_base[b]] (the next nav row on the shortest
 * path to the sample's goal, tabulated per sample over its scan's rows), 0 at the goal. */
/* The same step as the tail of the scoring + glue launch (sf_follower_glue.nav): state and outputs as in
 * sf_nav_step; the action and the `ended` flag are the ones the glue has just produced. */
typedef struct sf_nav_io {
    sf_nav_table nav;
    const int32_t *row, *view;   /* [B] current state */
    const int32_t* goal_hop;     /* [B, ld_hop] or NULL with target_next == NULL */
    int32_t ld_hop;
    const int32_t* hop_base;     /* [B] */
    int32_t *row_next, *vp_next, *view_next, *a_num_next; /* [B] */
    int32_t* cand_view_next;     /* [B,A] */
    float* sincos_next;          /* [B,A,4] */
    int64_t* target_next;        /* [B] or NULL */
} sf_nav_io;
int sf_nav_step(const sf_nav_table* nav, int B, const int32_t* row, const int32_t* view,
                const int64_t* a_t, const uint8_t* ended, const int32_t* goal_hop, int ld_hop,
                const int32_t* hop_base, int32_t* row_next, int32_t* vp_next, int32_t* view_next,
                int32_t* a_num_next, int32_t* cand_view_next, float* sincos_next,
                int64_t* target_next, sf_stream stream);

/* In-process kernel timing (no reference counterpart; what bench.py's `roofline.kernels` table is
 * measured with).  Between sf_profile_begin() and sf_profile_end() every kernel ANY host thread of the
 * process launches through this library (torch runs backward() on its own thread) carries a start and a stop event on its own dispatch, so a
 * pair's elapsed time is that kernel's execution time on the stream it ran on (the figure rocprofv3
 * --kernel-trace reports).  Not usable during hipGraph stream capture.  sf_profile_end waits for the
 * recorded kernels, writes one text line per kernel name -- "name\tcalls\ttotal_us\tmin_us\tmax_us\n"
 * -- into buf (NUL-terminated, truncated to cap) and returns the bytes the full text needs, or -1. */
int sf_profile_begin(void);
long sf_profile_end(char* buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* SF_HIP_H_ */
