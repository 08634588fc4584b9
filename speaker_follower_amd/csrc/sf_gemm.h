// Internal GEMM interface (fp32 in / fp32 accumulate on the gfx950 matrix cores).
#pragma once
#include "sf_common.h"

namespace sf {

// One K-segment of a "linear" product: contributes A[M,K] * W[N,K]^T.
struct Seg {
    const float* A;
    int lda;
    const float* W;
    int ldw;
    int K;
};

// Epilogue applied by the split-K reduce pass (or inline when there is a single split).
enum Epi : int {
    EPI_NONE = 0,    // y = acc + bias
    EPI_TANH = 1,    // y = tanh(acc + bias)
    EPI_MUL = 2,     // y = (acc + bias) * mul[n]         (EltwiseProdScoring fold)
    EPI_TANHBWD = 3, // y = (acc + bias) * (1 - aux[m,n]^2)   (backward through h~ = tanh(.))
};

struct LinearOut {
    float* y;          // [M, ldy]
    int ldy;
    const float* bias;   // [N] or null
    const float* bias2;  // [N] or null (LSTM: b_ih + b_hh)
    const float* mul;    // [N] for EPI_MUL
    float* y_pre;        // optional: un-multiplied value for EPI_MUL (saved for backward), ld = ldy_pre
    int ldy_pre;
    Epi epi;
    int accumulate;      // y += result (EPI_NONE only)
    // Fused element-wise neighbours of the backward pass (short-reduction kernel only; linear_nt
    // returns SF_ERR_UNSUPPORTED when the shape needs another kernel and the caller launches the
    // element-wise kernels separately):
    const float* aux;    // EPI_TANHBWD operand [M, ld_aux]
    int ld_aux;
    const float* addend; // v += addend[m, n] (row stride ld_addend) before the epilogue
    int ld_addend;
    const float* r1_s;   // v += r1_s[m] * r1_v[n] before the epilogue (rank-1 term)
    const float* r1_v;
};

// Workspace needed (in floats) for sf::linear_nt on an [M,N] output with total depth K.
size_t linear_ws_floats(int M, int N, int Ktot);
// How many K-splits linear_nt will use (1 => no workspace traffic).
int linear_ksplit(int M, int N, int Ktot);

// y[M,N] = epi( sum_s A_s[M,K_s] * W_s[N,K_s]^T + bias (+bias2) ).  K_s % 4 == 0, lda/ldw % 4 == 0.
// If `raw_slabs` is non-null the reduce pass is skipped and the caller receives the split-K
// partial slabs ([ksplit][M][N]) -- used by the LSTM cell whose pointwise kernel reduces them.
int linear_nt(const Seg* segs, int nseg, int M, int N, const LinearOut& out, float* ws,
              size_t ws_floats, hipStream_t st, float** raw_slabs = nullptr, int* ksplit_out = nullptr);

// y[M,N] (+)= A[M,K] * W[K,N]      (W row-major [K,N]; N % 4 == 0; lda % 4 == 0, A zero-padded to K%4)
int gemm_nn(const float* A, int lda, const float* W, int ldw, int M, int N, int K, float* y,
            int ldy, int accumulate, hipStream_t st);

// out[P,Q] (+)= Y[M,P]^T * X[M,Q]   (weight gradients; ldy % 4 == 0, ldx % 4 == 0, Q % 4 == 0)
// ws (optional): scratch for splitting a deep reduction over M into up to 16 slabs of [P,Q]
struct TnJob {                        // one weight gradient out[P,Q] (+)= Y[M,P]^T X[M,Q] of gemm_tn_group
    const float* Y; int ldy;
    const float* X; int ldx;
    int M, P, Q;
    float* out; int ldo;
    int accumulate;
};
int gemm_tn_group(const TnJob* jobs, int n, hipStream_t st, float* ws, size_t ws_floats);
int gemm_tn_main_columns(int M, int P, int Q);
extern int g_tn_group;                // sf_debug_grouped_weight_gradients
int gemm_tn(const float* Y, int ldy, const float* X, int ldx, int M, int P, int Q, float* out,
            int ldo, int accumulate, hipStream_t st, float* ws = nullptr, size_t ws_floats = 0);

// out[N] (+)= sum_m Y[m, n]         (bias gradients)
// out2 (optional) receives the same sums; ws (optional) lets a deep reduction be split over M
int colsum(const float* Y, int ldy, int M, int N, float* out, int accumulate, hipStream_t st,
           float* out2 = nullptr, float* ws = nullptr, size_t ws_floats = 0);

}  // namespace sf
