// Internal kernel-launch interface between the .hip translation units.
#pragma once
#include "sf_common.h"
#include "sf_gemm.h"
#include "sf_lstm.h"
#include "sf_lstm_pw.h"
#include "sf_rows.h"

namespace sf {

// ---- sf_attention.hip ---------------------------------------------------------------------------
// split_part / split_counter (optional): scratch ([visual_attn_split_floats] floats) and the
// per-sample ticket counters that let the forward pass use two workgroups per sample.
size_t visual_attn_split_floats(int B, int F);
int visual_attn(int mode, const PanoSrc& src, int B, const float* vec, int ldvec, float* alpha,
                float* out, int ldo, const Dropout& drop, int drop_col0, hipStream_t st,
                float* split_part = nullptr, unsigned* split_counter = nullptr, const double* vec64 = nullptr,
                int vec_slabs = 0, long vec_slab_stride = 0);   // (mode 1: vec = sum of K-split slabs, added up in the kernel)
// vec64: the forward with its scores accumulated in float64 from a float64 query (csrc/sf_precise.hip)
bool visual_attn_f64_supported(const PanoSrc& src, int B);

// ---- sf_precise.hip: y64 = A W^T + bias on the float64 matrix cores (A fp32 or f64; optional fp32 copy) --------
int linear_f64(const float* A32, const double* A64, int lda, const float* W, int ldw, const float* bias, int M, int N,
               int K, double* y64, int ldy64, float* y32, int ldy32, hipStream_t st);
int linear_f64_w64(const float* A32, int lda, const double* W, int ldw, const double* bias, int M, int N, int K,
                   double* y64, int ldy64, float* y32, int ldy32, hipStream_t st);
extern int g_precise_attention;       // sf_debug_precise_attention (default 1)
int text_attn_fwd(const float* ctx, const uint8_t* mask, int B, int L, int H, const float* t,
                  int ldt, float* alpha, float* wc, int ldwc, hipStream_t st,
                  const int32_t* ctx_row = nullptr);
int text_attn_bwd(const float* ctx, int B, int L, int H, const float* dwc, int lddwc,
                  const float* t, int ldt, const float* alpha, float* dt, int lddt, float* dctx,
                  hipStream_t st, float* ds_out = nullptr,    // dctx null + ds_out: deferred update
                  const int32_t* ctx_row = nullptr);          // (deferred form only) row b reads ctx row ctx_row[b]
// dctx[b,l,:] += sum_t (alpha[t,b,l] dcat2[t,b,:H] + ds[t,b,l] tt[t,b,:]) over S stacked steps
bool ctx_grad_supported(int S, int L, int H);
int ctx_grad_accum(const float* alpha, const float* ds, const float* dcat2, int lddc, const float* tt,
                   int S, int B, int L, int H, float* dctx, hipStream_t st);
int score_fwd(const CandSrc& src, int B, int D, const float* r, const float* wt, const float* b_a,
              const float* b_out, float* logit, hipStream_t st, int ldr = 0, const float* cst = nullptr);
// ce (optional): d(logit) = gscale (softmax(ce.logit) - onehot(ce.target)) is formed inside the
// kernel (and written to dlogit) instead of being read from dlogit
int score_bwd(const CandSrc& src, int B, const float* dlogit, float* dr, float* dc, hipStream_t st,
              const CeSrc* ce = nullptr);

// ---- sf_gemm.hip (workspace-aware NN) -------------------------------------------------------------
int gemm_nn_ws(const float* A, int lda, const float* W, int ldw, int M, int N, int K, float* y,
               int ldy, int accumulate, float* ws, size_t ws_floats, hipStream_t st);
size_t gemm_nn_ws_floats(int M, int N, int K);

// ---- sf_pointwise.hip ---------------------------------------------------------------------------
// LSTM gates: reduce `ks` split-K slabs [ks][B][4H] + biases (+ hoisted xg), activate, update state.
int lstm_pointwise_fwd(const LstmPwFwd& a, hipStream_t st);
int lstm_step_fused(const LstmStepArgs& p, hipStream_t st);      // sf_gemm.hip

extern unsigned long long* g_trace;   // sf_debug_trace buffer (development aid), null = off
extern int g_force_sc1;               // sf_debug_force_write_through
extern int g_nt_force_f32;            // sf_debug_gate_product_f32
extern int g_nt_big;                  // sf_debug_many_row_product (bit 0)
extern int g_nt_big_ksplit;           // ... (bit 1 clear): K splits of its raw-slab products
extern int g_tn_split_min_rows;       // sf_debug_tn_split_min_rows
extern long long g_persist_timeout;   // sf_debug_persist_timeout (ticks of 10 ns; < 0 = default 0.25 s)
size_t persistent_fault_word();       // dword index, behind the persistent launches' ticket word, of the fault word

unsigned* persist_lock_addr();         // the device-wide lock of every persistent launch (sf_persist.hip)

// ---- sf_persist.hip: the T recurrent steps of a table-input LSTM as one persistent launch ---------
size_t encoder_persistent_xchg_floats(int H);
bool encoder_persistent_supported(int B, int H, int T);
// hs / cs: [T+1,B,H] tapes (slot 0 is zeroed here); xchg: scratch of encoder_persistent_xchg_floats
// floats; done: one zero-initialised ticket word.  SF_ERR_UNSUPPORTED = use the per-step path.
int encoder_persistent(const float* w_hh, const float* b_ih, const float* b_hh, const float* xw_table,
                       const int64_t* seq, int Lpad, const int* lengths, int B, int H, int T, float* gates,
                       float* hs, float* cs, float* ctx, const Dropout& ctx_drop, float* xchg, unsigned* done,
                       hipStream_t st, float* c_out = nullptr,
                       // round 5 (the speaker's teacher-forced word recurrence): a given initial state (hs / cs slot 0 are
                       // then the caller's), tokens at seq[b * Lpad + t * seq_st], lengths / ctx may be null
                       const float* h_init = nullptr, const float* c_init = nullptr, long seq_st = 1);
// the speaker's S word steps (inference) as one persistent launch; cq = ctx W_in, cw = ctx W_c^T [B,Tp,H]
size_t speaker_persistent_xchg_floats();
bool speaker_persistent_supported(int B, int H, int Tp, int vocab);
int speaker_persistent(const float* w_hh, const float* b_ih, const float* b_hh, const float* xw_table,
                       const float* w_out, int ld_wout, const float* w_d2a, const float* b_d2a, int vocab, int ldv,
                       const float* cq, const float* cw, const uint8_t* mask, const float* h_init,
                       const float* c_init, const int64_t* targets, int feedback, int pad, int eos, int B, int H,
                       int Tp, int S, int64_t* words, float* step_scores, float* nll_term, float* live,
                       float* logits, float* alpha, float* h1_tape, float* c1_tape, uint8_t* ended, float* xchg,
                       unsigned* done, hipStream_t st, const sf_sample* sample = nullptr);
// the backward recurrence (all T steps of lstm_bwd_step_fused): dgates [T,B,4H] out
size_t encoder_bwd_persistent_xchg_floats();
int encoder_bwd_persistent(const float* w_hh, const int* lengths, int B, int H, int T, const float* gates,
                           const float* cs, const float* dctx, const Dropout& ctx_drop, const float* dh_in,
                           const float* dc_in, float* dgates, float* xchg, unsigned* done, hipStream_t st,
                           // round 5: the external gradient of h_t at dctx[b * dctx_sb + t * dctx_st + j] (0, 0 = [B,T,H]);
                           // dc0_out [B,H] = gradient wrt a given initial cell state; lengths may be null
                           long dctx_sb = 0, long dctx_st = 0, float* dc0_out = nullptr);

int lstm_pointwise_bwd(const LstmPwBwd& a, hipStream_t st);

int dropout_copy(const float* src, int lds, int B, int N, float* dst, int ldd, const Dropout& d,
                 int col0, hipStream_t st);
// S stacked steps of B rows: row t * B + b masked with site d.stream + stream_step * t, row key b
int dropout_steps(const float* src, int lds, int S, int B, int N, float* dst, int ldd, const Dropout& d,
                  uint32_t stream_step, hipStream_t st);
int row_mod(int* out, int M, int B, hipStream_t st);              // out[i] = i % B
// dst = (a ? a : 0) + (b ? b : 0)   [M,N] with row strides
int add2(const float* a, int lda, const float* b, int ldb, int M, int N, float* dst, int ldd,
         hipStream_t st);
// dpre = dy * (1 - y^2)
int tanh_bwd(const float* y, int ldy, const float* dy, int lddy, int M, int N, float* dpre,
             int ldp, hipStream_t st);
// dst[m, n] = src[m, n] * v[n]   and optionally accum[n] += sum_m src[m,n] * other[m,n]
int scale_cols(const float* src, int lds, const float* v, int M, int N, float* dst, int ldd,
               hipStream_t st);
int colsum_prod(const float* a, int lda, const float* b, int ldb, int M, int N, float* out,
                hipStream_t st);           // out[n] += sum_m a[m,n]*b[m,n]
int rank1_add(const float* s, const float* v, int M, int N, float* dst, int ldd,
              hipStream_t st);             // dst[m,n] += s[m] * v[n]
int dot_rows_accum(const float* s, const float* x, int ldx, int M, int N, float* out,
                   hipStream_t st);        // out[n] += sum_m s[m] * x[m,n]
int sum_accum(const float* s, int M, float* out, hipStream_t st);   // out[0] += sum_m s[m]
int fill(float* p, size_t n, float v, hipStream_t st);
int cotenant(int blocks, int threads, int lds_bytes, long long ticks, float* sink, hipStream_t st);   // sf_debug_cotenant
// device-flag ordering between two streams (sf_pointwise.hip)
int flag_wait(const unsigned* flag, unsigned target, hipStream_t st);
int flag_set(unsigned* flag, unsigned value, hipStream_t st);
int flag_wait_clear(unsigned* flag, unsigned* fault, unsigned fault_code, hipStream_t st);   // one-shot: wait for != 0, then clear
int adam_step(float* p, const float* g, float* m, float* v, size_t n, double lr, double beta1,
              double beta2, double eps, double wd, int step, hipStream_t st);
// the same with the 1-based step counter ON THE DEVICE (incremented by the call; coef: 2 floats of scratch)
int adam_step_dev(float* p, const float* g, float* m, float* v, size_t n, double lr, double beta1, double beta2,
                  double eps, double wd, int* step_dev, float* coef, const unsigned* guard, hipStream_t st);
struct RowMoves {                         // sf_move_rows
    struct M { const float* src; float* dst; const int* idx; int lds, ldd, w, scatter; } m[4];
    int n;
};
int move_rows(const RowMoves& mv, int n, hipStream_t st);
struct FillRegions {                      // sf_fill_regions
    struct R { void* ptr; unsigned long long count, value; int width; } r[8];
    int n;
};
int fill_regions(const FillRegions& fr, hipStream_t st);
int site_advance(uint32_t* word, uint32_t by, hipStream_t st);      // *word += by (sf_site_advance)
int store_u32x4(uint32_t* dst, uint32_t a, uint32_t b, uint32_t c, uint32_t d, hipStream_t st);   // sf_store_u32x4
int transpose(const float* src, int R, int C, float* dst, hipStream_t st);   // dst[C,R] = src^T
int transpose_ld(const float* src, int lds, int R, int C, float* dst, hipStream_t st);     // the same, src row stride lds
int dropout_tm(float* x, int T, int B, int E, const Dropout& d, const int* rev, hipStream_t st);
int embedding_bwd(const float* demb, int ldd, const int64_t* seq, int Lpad, int T, int B, int E, int padding_idx,
                  const Dropout& d, const int* rev, float* grad, hipStream_t st);
int embedding_tm(const float* table, int E, const int64_t* seq, int B, int Lpad, int T, float* out,
                 hipStream_t st);          // out[t, b, :] = table[seq[b, t], :]
int embedding_rows(const float* table, int E, const int64_t* idx, int B, float* out,
                   hipStream_t st);        // out[b, :] = table[idx[b], :]
int ctx_grad_slice(const float* dctx, int T, int H, int B, int t, const Dropout& d, float* out,
                   hipStream_t st);        // out[b,:] = dropout_mask(dctx[b,t,:])

int gather_panorama(const PanoSrc& s, int B, float* out, hipStream_t st);
int gather_candidates(const CandSrc& s, int B, float* all_u, float* is_valid, hipStream_t st);
int gather_actions(const CandSrc& s, int B, const int* a, float* out, hipStream_t st, int ldo = 0);   // ldo: row stride of out (0 = F)
int gather_path_actions(const float* table, int V, int IMG, int LOC, const int* vp, const int* act_view,
                        const float* sincos, const int* act, int N, float* out, int ldo, hipStream_t st);

int gather_rows(const float* src, int lds, const int* idx, int n, int w, float* dst, int ldd,
                hipStream_t st);
int scatter_rows(const float* src, int lds, const int* idx, int n, int w, float* dst, int ldd,
                 hipStream_t st);
int logprob_topk(float* logit, int ld, int N, int n, const int* n_valid, int k, int* idx,
                 float* logp, hipStream_t st);

// paired launches (see sf_attention.hip); SF_ERR_UNSUPPORTED = not pairable, launch separately
struct SmallPlan;
int pair_small_small(const SmallPlan& a, const SmallPlan& b, hipStream_t st);
int pair_small_text(const SmallPlan& a, const float* ctx, const uint8_t* mask, int B, int L, int H,
                    const float* t, int ldt, float* alpha, float* wc, int ldwc,
                    const int32_t* ctx_row, hipStream_t st);
int pair_visbwd_small(const PanoSrc& src, int B, const float* vec, int ldvec, float* alpha, float* out,
                      int ldo, const Dropout& drop, int drop_col0, const SmallPlan& b, hipStream_t st, int vec_slabs = 0,
                      long vec_slab_stride = 0);
int pair_vis_small(const PanoSrc& src, int B, const float* vec, int ldvec, float* alpha, float* out,
                   int ldo, const Dropout& drop, int drop_col0, float* split_part,
                   unsigned* split_counter, const SmallPlan& b, hipStream_t st, int phase = 0);
// phase 0: the whole split attention (ticket inside the launch); 1: per-group partials only;
// 2: merge of the partials written by a phase-1 launch (alpha, out)
// phase-1 partials beside the text attention (folded inference step)
int pair_vis_text(const PanoSrc& src, int B, const float* vec, int ldvec, float* alpha, float* out, int ldo,
                  const Dropout& drop, int drop_col0, float* split_part, const float* ctx, const uint8_t* mask,
                  int L, int H, const float* t, int ldt, float* talpha, float* wc, int ldwc,
                  const int32_t* ctx_row, hipStream_t st);

// the folded text stage of an inference decode step (sf_attention.hip: text_fold_body) and its consumer
int pair_textfold_small_small(const float* ctx_q, const float* ctx_o, const uint8_t* mask, int B, int L, int H,
                              const float* vec, int ldvec, float* part, unsigned* counter, float* z, int ldz, float* alpha,
                              const SmallPlan& a, const SmallPlan& b, hipStream_t st);
size_t text_fold_part_floats(int B, int H);
int pair_apro_small(const SmallPlan& a, const SmallPlan& b, hipStream_t st);
struct FGlue;
int pair_score_merge(const CandSrc& src, int B, int D, const float* r, const float* wt, const float* b_a,
                     const float* b_out, const FGlue& g, const PanoSrc& psrc, float* alpha, float* out, int ldo,
                     const Dropout& drop, int drop_col0, float* split_part, hipStream_t st, int ldr = 0,
                     const float* cst = nullptr);
int pair_vis_apro(const PanoSrc* src, int B, const float* vec, int ldvec, float* split_part, const SmallPlan& b,
                  hipStream_t st);

struct FGlue;
int follower_glue_fwd(const FGlue& g, hipStream_t st);
// scoring + glue in one launch (sf_attention.hip); g.logit receives the masked logits
int score_glue_fwd(const CandSrc& src, int B, int D, const float* r, const float* wt,
                   const float* b_a, const float* b_out, const FGlue& g, hipStream_t st, int ldr = 0,
                   const float* cst = nullptr);
// rps > 0: S stacked steps of rps rows each -- row m takes gscale[m / rps]
int softmax_ce_bwd(int B, int N, int ld, const float* logit, const int64_t* target, int ignore,
                   const float* gscale, float* dlogit, hipStream_t st, int rps = 0);
int speaker_glue_fwd(int B, int vocab, int ldv, const float* logit, const int64_t* target,
                     int feedback, int pad_idx, int eos_idx, uint8_t* ended, int64_t* w_t,
                     float* score, float* nll_term, float* live, hipStream_t st, const sf_sample* sample = nullptr,
                     int rps = 0);      // rps > 0: S stacked steps of rps rows each (row m sets ended[m % rps])
int speaker_loss_finalize(const float* sum_cnt, const int64_t* words, int eos, int T, int B, float* loss, float* gscale,
                          hipStream_t st);
int reduce_terms(const float* term, const float* live, int T, int B, float* sum_cnt,
                 hipStream_t st);
int loss_finalize(const float* sum_cnt, int T, float* loss, float* gscale, hipStream_t st);

}  // namespace sf
