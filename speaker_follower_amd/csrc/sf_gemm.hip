// fp32 GEMMs on the gfx950 matrix cores (v_mfma_f32_16x16x4_f32: exact f32 FMA chain).
//
// Every Linear / LSTMCell product on the speaker/follower path has a skinny batch dimension
// (M = 100 samples, or M = batch*time for the hoisted encoder input product), so the kernels
// stream the weight operand once, straight from HBM into registers (no LDS round trip: a
// weight tile is used by exactly one wave), keep a column of 16x16 accumulator tiles per wave
// and get their parallelism from N-tiles x K-splits.  The small activation operand is re-read
// by every wave and is served by L1/L2.
//
// Fragment trick: mfma_f32_16x16x4 wants A[i][k] from lane (i = lane&15, k = lane>>4) and
// B[k][j] from lane (j = lane&15, k = lane>>4).  A sum over k is order-independent, so a lane
// loads a float4 of four CONSECUTIVE k (k0 + 4*(lane>>4) + c, c = 0..3) for its row and feeds
// component c to MFMA c: A and B use the same permutation of k, loads are 16 B wide and each
// 16-lane group reads 64 contiguous bytes per row.  For operands whose contiguous dimension is
// the row/column index instead (W[K,N] in dX = dY*W, both operands of dW = dY^T*X) the float4
// runs along that index and defines four "virtual" 16-wide tiles with stride-4 columns, which
// the epilogue writes back as float4.
#include "sf_gemm.h"
#include "sf_gemm_small.h"
#include "sf_lstm.h"
#include "sf_split.h"

#include <cstdlib>

namespace sf {

int g_tn_split_min_rows = 4096;   // sf_debug_tn_split_min_rows (tests run the split weight-gradient kernel on small shapes)
int g_nt_big_ksplit = 1;     // sf_debug_many_row_product(on | 2 * no K splits of the raw-slab many-row products)
int g_nt_big = 1;            // sf_debug_many_row_product: M >= 512 products on gemm_nt_big_kernel (0: gemm_nt_kernel, rounds 1-4)
int g_nt_force_f32 = 0;      // sf_debug_gate_product_f32: run the LSTM gate product on v_mfma_f32_16x16x4_f32 (round 1-3 kernel)

namespace {

// ------------------------------------------------------------------------------------------------
// NT: C[M,N] = sum_s A_s[M,K_s] * W_s[N,K_s]^T        (forward Linear; both operands K-contiguous)
// grid (ceil(N/64), ksplit, mblocks), block 256 = 4 waves, wave = MT m-tiles x one 16-col n-tile.
// The k loop runs a 3-deep register prefetch ring: a weight chunk comes straight from HBM
// (~900 cycles) while one chunk's MFMAs take MT*4*32 cycles, so two chunks stay in flight.
// ------------------------------------------------------------------------------------------------
struct NtArgs {
    Seg seg[3];
    int nseg;
    int M, N;
    int chunks_total;      // sum over segments of ceil(K_s / 16)
    int ksplit;
    float* out;            // slab base ([ksplit][M][N], ld = N) or final y when ksplit == 1
    int ldo;               // N for slabs, ldy for direct
    const float* bias;     // direct mode only
    const float* bias2;
    int epi;               // direct mode only: EPI_*
    const float* mul;
    float* y_pre;
    int ldy_pre;
    int accumulate;        // direct mode only
};

template <int MT>
struct Frag {
    float4 b;
    float4 a[MT];
};

template <int MT>
__device__ __forceinline__ void nt_load(Frag<MT>& f, const float* pb, const float* const (&pa)[MT],
                                        int chunk) {
    f.b = ld4(pb + chunk * 16);
#pragma unroll
    for (int t = 0; t < MT; ++t) f.a[t] = ld4(pa[t] + chunk * 16);
}

template <int MT>
__device__ __forceinline__ void nt_mma(const Frag<MT>& f, f32x4 (&acc)[MT]) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[t] = mfma16(comp(f.a[t], c), comp(f.b, c), acc[t]);
}

// Chunks [lo, hi) of one segment.  The steady-state loop body is branch-free (prefetch indices are
// clamped instead of predicated) so that the compiler can keep two chunks of loads in flight behind
// counted s_waitcnt vmcnt(N); a conditional load would force vmcnt(0) at every join.
template <int MT>
__device__ __forceinline__ void nt_segment(const Seg& sg, int lo, int hi, int n,
                                           const int (&mrow)[MT], int kk, f32x4 (&acc)[MT]) {
    const int full = sg.K >> 4;
    const float* pb = sg.W + (size_t)n * sg.ldw + 4 * kk;
    const float* pa[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) pa[t] = sg.A + (size_t)mrow[t] * sg.lda + 4 * kk;
    const int end = min(hi, full);          // full chunks are [lo, end)
    if (lo < end) {
        const int last = end - 1;
        Frag<MT> f0, f1, f2;
        nt_load<MT>(f0, pb, pa, lo);
        nt_load<MT>(f1, pb, pa, min(lo + 1, last));
        int i = lo;
        for (; i + 3 <= end; i += 3) {
            nt_load<MT>(f2, pb, pa, min(i + 2, last));
            nt_mma<MT>(f0, acc);
            nt_load<MT>(f0, pb, pa, min(i + 3, last));
            nt_mma<MT>(f1, acc);
            nt_load<MT>(f1, pb, pa, min(i + 4, last));
            nt_mma<MT>(f2, acc);
        }
        if (i < end) nt_mma<MT>(f0, acc);
        if (i + 1 < end) nt_mma<MT>(f1, acc);
    }
    if (hi > full) {   // K_s % 16 != 0: one partial chunk, float4 granularity (K_s % 4 == 0)
        const bool ok = full * 16 + 4 * kk < sg.K;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        Frag<MT> f;
        f.b = ok ? ld4(pb + full * 16) : z;
#pragma unroll
        for (int t = 0; t < MT; ++t) f.a[t] = ok ? ld4(pa[t] + full * 16) : z;
        nt_mma<MT>(f, acc);
    }
}

template <int MT>
__device__ __forceinline__ void nt_mainloop(const NtArgs& a, int c0, int c1, int n,
                                            const int (&mrow)[MT], int kk, f32x4 (&acc)[MT]) {
    int cs = 0;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        if (s >= a.nseg) break;
        const Seg sg = a.seg[s];
        const int nch = (sg.K + 15) >> 4;
        const int lo = max(c0, cs) - cs, hi = min(c1, cs + nch) - cs;
        cs += nch;
        if (lo < hi) nt_segment<MT>(sg, lo, hi, n, mrow, kk, acc);
    }
}

template <int MT>
__global__ __launch_bounds__(256) void gemm_nt_kernel(NtArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int n0 = (blockIdx.x * 4 + wave) * 16;
    if (n0 >= a.N) return;
    const int split = blockIdx.y;
    const int m0 = blockIdx.z * (16 * MT);
    const int li = lane & 15, kk = lane >> 4;

    const int n = min(n0 + li, a.N - 1);
    int mrow[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) mrow[t] = min(m0 + 16 * t + li, a.M - 1);

    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int c0 = (int)(((long)split * a.chunks_total) / a.ksplit);
    const int c1 = (int)(((long)(split + 1) * a.chunks_total) / a.ksplit);
    nt_mainloop<MT>(a, c0, c1, n, mrow, kk, acc);

    // D layout: col = lane & 15, row = (lane >> 4) * 4 + r
    const int col = n0 + li;
    if (col >= a.N) return;
    float* out = a.out + (a.ksplit > 1 ? (size_t)split * a.M * a.N : 0);
    float bsum = 0.f;
    if (a.bias) bsum += a.bias[col];
    if (a.bias2) bsum += a.bias2[col];
    const float mulv = a.epi == EPI_MUL ? a.mul[col] : 1.f;
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + 16 * t + kk * 4 + r;
            if (row >= a.M) continue;
            float v = acc[t][r] + bsum;
            if (a.epi == EPI_TANH) v = tanhf(v);
            if (a.epi == EPI_MUL) {
                if (a.y_pre) a.y_pre[(size_t)row * a.ldy_pre + col] = v;
                v *= mulv;
            }
            float* o = out + (size_t)row * a.ldo + col;
            *o = (a.accumulate && a.ksplit == 1) ? *o + v : v;
        }
}

// ------------------------------------------------------------------------------------------------
// LDS-tiled NT GEMM for the one MFMA-bound product of the path: the decoder LSTM gates
// [B<=128, 4864] x [2048, 4864]^T.  The register-streaming kernel above feeds the matrix cores with
// "fragment shaped" loads (16 rows x 64 B per wave instruction), which keeps the texture addresser
// busy twice as long as full lines would; here a block stages BK = 64 deep tiles of A (MT*16 rows)
// and W (64 rows) into LDS with full 256-B row segments (16 lanes per row), double buffered through
// registers (next stage's global loads are in flight while the current stage's 16*MT MFMAs per wave
// run), and the four waves read their fragments with conflict-free ds_read_b128 (row stride
// 68 dwords: 16 rows x 4 banks tile the 64 banks exactly).
// grid (N/64, ksplit), block 256; a block's stages lie inside one K segment each (K_s % 64 == 0).
// ------------------------------------------------------------------------------------------------
// Row stride 72 dwords: conflict-free for the ds_read_b128 lane groups of gfx950 ({0-3,12-15,20-27},
// ...: MI355X_MICROARCH.md, LDS table).  68 gave a 2-way conflict in every group
// (SQ_LDS_BANK_CONFLICT = 35 % of SQ_LDS_IDX_ACTIVE).
constexpr int TBK = 64, TLD = TBK + 8;

// 8 waves per block: waves 0-3 take the first half of every 64-deep stage, waves 4-7 the second
// half (two waves per SIMD hide each other's barrier and LDS latencies); the halves meet in LDS.
template <int MT>
__global__ __launch_bounds__(512) void gemm_nt_tiled_kernel(NtArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int AROWS = MT * 16, WROWS = 64;
    constexpr int BUF = (AROWS + WROWS) * TLD;           // floats per stage buffer
    constexpr int APASS = (MT + 1) / 2;                  // 32 rows x 16 float4 per staging pass
    const int tid = threadIdx.x, lane = tid & 63, wave8 = tid >> 6;
    const int wave = wave8 & 3, khalf = wave8 >> 2;
    const int li = lane & 15, kk = lane >> 4;
    // XCD-aware tile map: consecutive workgroup ids go round-robin to the 8 XCDs (each with its own
    // L2).  Give every XCD a contiguous range of (split, n-tile) pairs, so that its L2 only ever
    // holds ITS k-slices of the activation operand (M x K/8) instead of all of A.
    int n0, split;
    {
        const int total = gridDim.x * gridDim.y;
        const int b = blockIdx.x + gridDim.x * blockIdx.y;
        int g = b;
        if ((total & 7) == 0) g = (b & 7) * (total >> 3) + (b >> 3);
        split = g / (int)gridDim.x;
        n0 = (g % (int)gridDim.x) * 64;
    }
    const int ldrow = tid >> 4, ldc4 = tid & 15;         // staging: 32 rows x 16 float4 per pass

    // stage range of this split (stages of 64 k over the concatenated segments)
    const int st0n = a.seg[0].K / TBK;
    const int st1n = a.nseg > 1 ? a.seg[1].K / TBK : 0;
    const int st2n = a.nseg > 2 ? a.seg[2].K / TBK : 0;
    const int stages = st0n + st1n + st2n;
    const int s_lo = (int)(((long)split * stages) / a.ksplit);
    const int s_hi = (int)(((long)(split + 1) * stages) / a.ksplit);

    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    struct Regs {
        float4 a[APASS];
        float4 w[2];
    };
    auto gload = [&](Regs& r, int s) {
        const float* A;
        const float* W;
        int lda, ldw, k0;
        if (s < st0n) {
            A = a.seg[0].A; W = a.seg[0].W; lda = a.seg[0].lda; ldw = a.seg[0].ldw; k0 = s * TBK;
        } else if (s < st0n + st1n) {
            A = a.seg[1].A; W = a.seg[1].W; lda = a.seg[1].lda; ldw = a.seg[1].ldw; k0 = (s - st0n) * TBK;
        } else {
            A = a.seg[2].A; W = a.seg[2].W; lda = a.seg[2].lda; ldw = a.seg[2].ldw;
            k0 = (s - st0n - st1n) * TBK;
        }
#pragma unroll
        for (int p = 0; p < APASS; ++p) {
            const int row = min(p * 32 + ldrow, a.M - 1);
            r.a[p] = ld4(A + (size_t)row * lda + k0 + 4 * ldc4);
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int row = min(n0 + p * 32 + ldrow, a.N - 1);
            r.w[p] = ld4(W + (size_t)row * ldw + k0 + 4 * ldc4);
        }
    };
    auto lstore = [&](const Regs& r, int buf) {
        float* As = smem + buf * BUF;
        float* Ws = As + AROWS * TLD;
#pragma unroll
        for (int p = 0; p < APASS; ++p)
            if (p * 32 + ldrow < AROWS)
                *reinterpret_cast<float4*>(As + (p * 32 + ldrow) * TLD + 4 * ldc4) = r.a[p];
#pragma unroll
        for (int p = 0; p < 2; ++p)
            *reinterpret_cast<float4*>(Ws + (p * 32 + ldrow) * TLD + 4 * ldc4) = r.w[p];
    };
    // one global load of the staging set (piece 0..APASS+1), issued between groups of 7 MFMAs: six
    // back-to-back loads from all 8 waves right after the barrier stall every wave on the address
    // path with the matrix pipe idle (timestamped: ~0.4 us per stage)
    struct StageSrc {
        const float* A;
        const float* W;
        int lda, ldw;
    };
    auto stage_src = [&](int s) {
        const float* A;
        const float* W;
        int lda, ldw, k0;
        if (s < st0n) {
            A = a.seg[0].A; W = a.seg[0].W; lda = a.seg[0].lda; ldw = a.seg[0].ldw; k0 = s * TBK;
        } else if (s < st0n + st1n) {
            A = a.seg[1].A; W = a.seg[1].W; lda = a.seg[1].lda; ldw = a.seg[1].ldw; k0 = (s - st0n) * TBK;
        } else {
            A = a.seg[2].A; W = a.seg[2].W; lda = a.seg[2].lda; ldw = a.seg[2].ldw;
            k0 = (s - st0n - st1n) * TBK;
        }
        return StageSrc{A + k0 + 4 * ldc4, W + k0 + 4 * ldc4, lda, ldw};
    };
    auto gpiece = [&](Regs& r, const StageSrc& ss, int piece) {
        if (piece < APASS) {
            const int row = min(piece * 32 + ldrow, a.M - 1);
            r.a[piece < APASS ? piece : 0] = ld4(ss.A + (size_t)row * ss.lda);
        } else if (piece < APASS + 2) {
            const int p = piece - APASS;
            const int row = min(n0 + p * 32 + ldrow, a.N - 1);
            r.w[p & 1] = ld4(ss.W + (size_t)row * ss.ldw);
        }
    };
    auto compute = [&](int buf, Regs* nx, int s_next) {
        const float* As = smem + buf * BUF;
        const float* Ws = As + AROWS * TLD + (wave * 16 + li) * TLD;
        StageSrc ss{};
        if (nx) ss = stage_src(s_next);
#pragma unroll
        for (int cc = 0; cc < TBK / 32; ++cc) {
            const int c = khalf * (TBK / 32) + cc;
            const float4 b = *reinterpret_cast<const float4*>(Ws + 16 * c + 4 * kk);
            float4 av[MT];
#pragma unroll
            for (int t = 0; t < MT; ++t)
                av[t] = *reinterpret_cast<const float4*>(As + (t * 16 + li) * TLD + 16 * c + 4 * kk);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int t = 0; t < MT; ++t) acc[t] = mfma16(comp(av[t], j), comp(b, j), acc[t]);
                if (nx) gpiece(*nx, ss, cc * 4 + j);             // 8 slots >= APASS + 2 pieces
            }
        }
    };

    // Two register sets alternate over two LDS buffers; the refill of a set (its six loads) is
    // interleaved with the MFMAs of the stage that follows its store.  Look-ahead: rb is loaded
    // during stage s and stored after stage s+1, ra during stage s+1 and stored after stage s+2.
    // Prefetch indices are clamped, not predicated.
    if (s_lo < s_hi) {
        const int last = s_hi - 1;
        Regs ra, rb;
        gload(ra, s_lo);
        lstore(ra, 0);
        gload(ra, min(s_lo + 1, last));
        gload(rb, min(s_lo + 2, last));
        __syncthreads();
        bool first = true;
        for (int s = s_lo; s < s_hi; s += 2) {
            compute(0, first ? nullptr : &rb, min(s + 2, last));   // stage s; rb <- s+2 (after the 1st)
            first = false;
            lstore(ra, 1);
            __syncthreads();
            if (s + 1 >= s_hi) break;
            compute(1, &ra, min(s + 3, last));                     // stage s+1; ra <- s+3
            lstore(rb, 0);
            __syncthreads();
        }
    }
    __syncthreads();

    // the two K halves meet in LDS (the stage buffers are free after the loop's last barrier)
    f32x4* red = reinterpret_cast<f32x4*>(smem);
    if (khalf == 1) {
#pragma unroll
        for (int t = 0; t < MT; ++t) red[(wave * MT + t) * 64 + lane] = acc[t];
    }
    __syncthreads();
    if (khalf == 1) return;
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] += red[(wave * MT + t) * 64 + lane];

    const int col = n0 + wave * 16 + li;
    if (col >= a.N) return;
    float* out = a.out + (a.ksplit > 1 ? (size_t)split * a.M * a.N : 0);
    float bsum = 0.f;
    if (a.bias) bsum += a.bias[col];
    if (a.bias2) bsum += a.bias2[col];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * t + kk * 4 + r;
            if (row < a.M) {
                float* o = out + (size_t)row * a.ldo + col;
                const float v = acc[t][r] + bsum;
                *o = (a.accumulate && a.ksplit == 1) ? *o + v : v;
            }
        }
}

// ------------------------------------------------------------------------------------------------
// The same product on the BF16 matrix cores, at fp32 accuracy, by error-free operand splitting (round 4; the
// arithmetic, its error analysis and the measurements against float64 are in sf_split.h).  The gate product was the
// one MFMA-bound kernel of the path (fp32 MFMA loop at 92 % of the f32 vector rate); six bf16 MFMAs per product are
// 6/16 of that time, and what bounds the kernel now is moving its operands (timed with the MFMAs removed: 16 of its
// 20 us).
//
// Block = 64 columns x all rows (<= 128, as MT m-tiles of 16) x one K split; 8 waves = 4 n-tiles of 16
// columns x the 2 K-halves of every 64-deep stage -- the decomposition of gemm_nt_tiled_kernel, with
// v_mfma_f32_16x16x32_bf16 (one instruction per 32-deep half).  Operands are fetched as fp32 with full
// 256-B row segments (as the fp32 kernel does), split by the thread that fetched them and stored to LDS
// as three bf16 planes per operand ([row][64 k] = 128 B per row and plane, 16-B chunks XOR-swizzled with
// (row >> 1) & 7: the ds_read_b128 lane groups of gfx950 -- 16 rows, chunk c for 8 of them and c + 1 for
// the others -- then hit 16 distinct 16-B bank groups without padding: 2 x 3 x 176 rows x 128 B = 132 KB of
// the 160 KB at 112 rows).  The split + LDS stores of stage s+1 and the global loads of stage s+3 sit between
// the MFMAs of stage s: one barrier per stage.  The K halves meet in LDS as in the fp32 kernel.
// ------------------------------------------------------------------------------------------------
constexpr int SPL_ROWB = 128;             // bytes per row and plane (64 bf16)

template <int MT>
__global__ __launch_bounds__(512) void gemm_nt_split_kernel(NtArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    constexpr int APASS = (MT + 1) / 2;                  // 32 rows x 16 float4 per staging pass
    constexpr int AROWS = APASS * 32;                    // (whole passes: every staging store is unconditional)
    constexpr int PLANE_A = AROWS * SPL_ROWB;
    constexpr int BUF = 3 * PLANE_A;                     // bytes per stage buffer (A only: W never touches LDS)
    constexpr int NPIECE = APASS + 2;                    // loads per thread and stage: APASS of A, 2 of W
    const int tid = threadIdx.x, lane = tid & 63, wave8 = tid >> 6;
    const int wave = wave8 & 3, khalf = wave8 >> 2;      // n-tile of 16 columns, K-half of a stage
    const int li = lane & 15, kk = lane >> 4;
    int n0, split;
    {   // XCD-aware (split, n-block) map, as gemm_nt_tiled_kernel
        const int total = gridDim.x * gridDim.y;
        const int b = blockIdx.x + gridDim.x * blockIdx.y;
        int g = b;
        if ((total & 7) == 0) g = (b & 7) * (total >> 3) + (b >> 3);
        split = g / (int)gridDim.x;
        n0 = (g % (int)gridDim.x) * 64;
    }
    const int ldrow = tid >> 4, ldc4 = tid & 15;         // A staging: 32 rows x 16 float4 per pass
    const int st0n = a.seg[0].K / TBK;
    const int st1n = a.nseg > 1 ? a.seg[1].K / TBK : 0;
    const int st2n = a.nseg > 2 ? a.seg[2].K / TBK : 0;
    const int stages = st0n + st1n + st2n;
    const int s_lo = (int)(((long)split * stages) / a.ksplit);
    const int s_hi = (int)(((long)(split + 1) * stages) / a.ksplit);

    f32x4 hi[MT], lo[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) hi[t] = lo[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Staging registers are filled by INLINE-ASM loads and released by counted `s_waitcnt vmcnt(N)`: the compiler's own
    // bookkeeping loses the count across the loop's back edge and waits vmcnt(0) at the first use in every stage,
    // i.e. for the loads issued one stage earlier -- a prefetch distance of one stage (~1 us) against an HBM latency
    // of 1-2 us under load.  Loads return in order, so "at most N newer loads outstanding" means this one has landed.
    struct Regs {
        f32x4 a[APASS];          // A: this thread's float4 of rows 32 p + ldrow (shared through LDS)
        f32x4 w[2];              // W: THIS WAVE's fragment rows (n-tile, K-half), never shared: registers only
    };
    struct StageSrc {
        const float* A;          // wave-uniform bases (SGPR pairs): the row offset travels in one VGPR
        const float* W;
        int lda, ldw;
    };
    // The segment of a stage is picked with mask arithmetic on values that sit in SGPRs: no branch (a stage's whole
    // body stays ONE basic block the scheduler can interleave), no load (a select of loaded values is turned into a
    // load from a selected address, i.e. a memory round trip at the head of every stage).
    const unsigned long long pa0 = (unsigned long long)a.seg[0].A, pw0 = (unsigned long long)a.seg[0].W,
                             pa1 = (unsigned long long)a.seg[1].A, pw1 = (unsigned long long)a.seg[1].W,
                             pa2 = (unsigned long long)a.seg[2].A, pw2 = (unsigned long long)a.seg[2].W;
    const unsigned lda0 = a.seg[0].lda, ldw0 = a.seg[0].ldw, lda1 = a.seg[1].lda, ldw1 = a.seg[1].ldw, lda2 = a.seg[2].lda,
                   ldw2 = a.seg[2].ldw;
    auto stage_src = [&](int s) {
        const unsigned long long m0 = s < st0n ? ~0ull : 0ull, m2 = s >= st0n + st1n ? ~0ull : 0ull, m1 = ~(m0 | m2);
        const int k0 = (s - (int)((unsigned)st0n & (unsigned)~m0) - (int)((unsigned)st1n & (unsigned)m2)) * TBK;
        return StageSrc{reinterpret_cast<const float*>((pa0 & m0) | (pa1 & m1) | (pa2 & m2)) + k0,
                        reinterpret_cast<const float*>((pw0 & m0) | (pw1 & m1) | (pw2 & m2)) + k0,
                        (int)((lda0 & (unsigned)m0) | (lda1 & (unsigned)m1) | (lda2 & (unsigned)m2)),
                        (int)((ldw0 & (unsigned)m0) | (ldw1 & (unsigned)m1) | (ldw2 & (unsigned)m2))};
    };
    auto gld = [&](f32x4& dst, const float* base, unsigned byte_off) {
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(byte_off), "s"(base) : "memory");
    };
    // The K index inside one MFMA is free as long as A and B agree.  Lane (li, kk) holds, of its row's 32-deep half,
    // k = 4 kk .. 4 kk + 3 and 16 + 4 kk .. 16 + 4 kk + 3: a W fragment is then two 16-byte loads per lane whose 16-lane
    // groups read FULL 64-byte lines of a row (the straight 8-consecutive-k mapping reads half of every line twice).
    const int wrow = min(n0 + wave * 16 + li, a.N - 1);
    auto gpiece = [&](Regs& r, const StageSrc& ss, int piece) {       // one global load of the staging set
        if (piece < APASS) {
            const int row = min(piece * 32 + ldrow, a.M - 1);
            gld(r.a[piece < APASS ? piece : 0], ss.A, (unsigned)(row * ss.lda + 4 * ldc4) * 4u);
        } else if (piece < NPIECE) {
            const int e = piece - APASS;
            gld(r.w[e & 1], ss.W, (unsigned)(wrow * ss.ldw + 32 * khalf + 16 * e + 4 * kk) * 4u);
        }
    };
    // piece `piece` of `r` has landed once at most `newer` younger loads are outstanding
#define SPL_LANDED(reg, newer) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(reg) : "n"(newer))
    // A planes in LDS: [plane][row][64 k bf16]; inside a row the 16-byte chunk (half g, lane group kk) holds
    // k = 32 g + 4 kk .. + 3 (first 8 bytes) and 32 g + 16 + 4 kk .. + 3 (last 8 bytes); chunks XOR-swizzled with (row >> 1) & 7
    auto st_off = [&](int row, int c4) {
        const int chunk = ((c4 >> 3) << 2) | (c4 & 3), e = (c4 >> 2) & 1;
        return row * SPL_ROWB + (((chunk ^ (row >> 1)) & 7) << 4) + (e << 3);
    };
    auto spiece_a = [&](const Regs& r, int buf, int piece) {          // split one float4 of A, three 8-byte LDS stores
        uint2 p1, p2, p3;
        const f32x4 v = r.a[piece < APASS ? piece : 0];
        split3_f4(make_float4(v[0], v[1], v[2], v[3]), p1, p2, p3);
        unsigned char* d = smem_b + buf * BUF + st_off(piece * 32 + ldrow, ldc4);
        *reinterpret_cast<uint2*>(d) = p1;
        *reinterpret_cast<uint2*>(d + PLANE_A) = p2;
        *reinterpret_cast<uint2*>(d + 2 * PLANE_A) = p3;
    };
    struct BFrag {
        bf16x8 p[3];
    };
    auto split_w = [&](const Regs& r, BFrag& f) {                     // this wave's W fragment of the next stage
        uint2 x1, x2, x3, y1, y2, y3;
        split3_f4(make_float4(r.w[0][0], r.w[0][1], r.w[0][2], r.w[0][3]), x1, x2, x3);
        split3_f4(make_float4(r.w[1][0], r.w[1][1], r.w[1][2], r.w[1][3]), y1, y2, y3);
        f.p[0] = __builtin_bit_cast(bf16x8, uint4{x1.x, x1.y, y1.x, y1.y});
        f.p[1] = __builtin_bit_cast(bf16x8, uint4{x2.x, x2.y, y2.x, y2.y});
        f.p[2] = __builtin_bit_cast(bf16x8, uint4{x3.x, x3.y, y3.x, y3.y});
    };
    const int fchunk = 4 * khalf + kk;
    auto frag = [&](const unsigned char* plane, int row) {
        return *reinterpret_cast<const bf16x8*>(plane + row * SPL_ROWB + (((fchunk ^ (row >> 1)) & 7) << 4));
    };
    // One stage: 6 MT MFMAs on buffer `buf` with the W fragment `bc`.  `rs` holds the raw operands of the NEXT stage:
    // its A pieces are split into LDS buffer `sbuf`, its W pieces into the fragment `bn`, and every register is
    // reloaded with its piece of stage `s_next` right behind its use.
    auto compute = [&](int buf, const BFrag& bc, Regs& rs, BFrag& bn, int s_next, int sbuf) {
        const unsigned char* Ab = smem_b + buf * BUF;
        const StageSrc ss = stage_src(s_next);
        // the staging work of slot t (between the MFMAs of tile t)
        auto stage_work_a = [&](int t) {
            if (t < APASS) {
                // younger than this piece's load: the other set's NPIECE loads of the previous stage and this set's t
                // reloads of this stage
                SPL_LANDED(rs.a[t < APASS ? t : 0], NPIECE + t);
                spiece_a(rs, sbuf, t);
            } else if (t == APASS) {
                SPL_LANDED(rs.w[0], NPIECE + APASS);
                SPL_LANDED(rs.w[1], NPIECE + APASS);
                split_w(rs, bn);
            }
        };
        auto stage_work_b = [&](int t) {
            if (t < APASS) gpiece(rs, ss, t);
            else if (t == APASS) { gpiece(rs, ss, APASS); gpiece(rs, ss, APASS + 1); }
        };
        // Tiles go in PAIRS with their MFMAs interleaved (consecutive MFMAs never share an accumulator) and the
        // fragments of the next pair are requested before the current pair's MFMAs are issued.
        bf16x8 fa[2][2][3];                                           // [parity][tile of the pair][plane]
        auto fetch = [&](int pr, int t0) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (t0 + u < MT) {
                    const int arow = (t0 + u) * 16 + li;
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) fa[pr][u][pl] = frag(Ab + pl * PLANE_A, arow);
                }
        };
        fetch(0, 0);
#pragma unroll
        for (int t0 = 0; t0 < MT; t0 += 2) {
            const int pr = (t0 >> 1) & 1;
            if (t0 + 2 < MT) fetch(pr ^ 1, t0 + 2);
            const bool two = t0 + 1 < MT;
            const int t1 = two ? t0 + 1 : t0;
#define SPL_MM(acc, t, pa, pb) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[pr][(t) - t0][pa], bc.p[pb], acc[t], 0, 0, 0)
            SPL_MM(lo, t0, 2, 0);
            if (two) SPL_MM(lo, t1, 2, 0);
            SPL_MM(lo, t0, 0, 2);
            if (two) SPL_MM(lo, t1, 0, 2);
            stage_work_a(t0);
            SPL_MM(lo, t0, 1, 1);
            if (two) SPL_MM(lo, t1, 1, 1);
            SPL_MM(lo, t0, 1, 0);
            if (two) SPL_MM(lo, t1, 1, 0);
            stage_work_b(t0);
            if (two) stage_work_a(t1);
            SPL_MM(lo, t0, 0, 1);
            if (two) SPL_MM(lo, t1, 0, 1);
            SPL_MM(hi, t0, 0, 0);
            if (two) SPL_MM(hi, t1, 0, 0);
            if (two) stage_work_b(t1);
#undef SPL_MM
        }
        if (MT <= APASS) {                                            // (MT = 1: the W pieces did not fit above)
            SPL_LANDED(rs.w[0], NPIECE + APASS);
            SPL_LANDED(rs.w[1], NPIECE + APASS);
            split_w(rs, bn);
            gpiece(rs, ss, APASS);
            gpiece(rs, ss, APASS + 1);
        }
    };
    auto gload = [&](Regs& r, int s) {
        const StageSrc ss = stage_src(s);
#pragma unroll
        for (int piece = 0; piece < NPIECE; ++piece) gpiece(r, ss, piece);
    };

    // Register sets ra / rb alternate: at the top of stage s the buffer of s holds its A planes and one fragment its W
    // pieces; one set holds the raw operands of stage s+1 (landed) and the other those of stage s+2 (in flight).
    // Stage indices are clamped, not predicated.
    if (s_lo < s_hi) {
        const int last = s_hi - 1;
        Regs ra, rb;
        BFrag b0, b1;
        gload(ra, s_lo);
        gload(rb, min(s_lo + 1, last));
#pragma unroll
        for (int piece = 0; piece < APASS; ++piece) {
            SPL_LANDED(ra.a[piece], NPIECE);                          // (rb's loads are younger)
            spiece_a(ra, 0, piece);
        }
        SPL_LANDED(ra.w[0], NPIECE);
        SPL_LANDED(ra.w[1], NPIECE);
        split_w(ra, b0);
        gload(ra, min(s_lo + 2, last));
        __syncthreads();
        for (int s = s_lo; s < s_hi; s += 2) {
            compute(0, b0, rb, b1, min(s + 3, last), 1);      // stage s on buffer 0; rb (s+1) -> buffer 1 / b1; rb <- s+3
            __syncthreads();
            if (s + 1 >= s_hi) break;
            compute(1, b1, ra, b0, min(s + 4, last), 0);      // stage s+1 on buffer 1; ra (s+2) -> buffer 0 / b0; ra <- s+4
            __syncthreads();
        }
    }
    // (the clamped prefetches of the last stages are still in flight: nothing below may reuse their registers yet)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#undef SPL_LANDED

    // the two K halves meet in LDS (the stage buffers are free behind the loop's last barrier)
    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = hi[t] + lo[t];
    f32x4* red = reinterpret_cast<f32x4*>(smem_b);
    if (khalf == 1) {
#pragma unroll
        for (int t = 0; t < MT; ++t) red[(wave * MT + t) * 64 + lane] = acc[t];
    }
    __syncthreads();
    if (khalf == 1) return;
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] += red[(wave * MT + t) * 64 + lane];

    const int col = n0 + wave * 16 + li;
    if (col >= a.N) return;
    float* out = a.out + (a.ksplit > 1 ? (size_t)split * a.M * a.N : 0);
    float bsum = 0.f;
    if (a.bias) bsum += a.bias[col];
    if (a.bias2) bsum += a.bias2[col];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * t + kk * 4 + r;
            if (row < a.M) {
                float* o = out + (size_t)row * a.ldo + col;
                const float v = acc[t][r] + bsum;
                *o = (a.accumulate && a.ksplit == 1) ? *o + v : v;
            }
        }
}

// ------------------------------------------------------------------------------------------------
// MANY-ROW NT product (round 5): C[M,N] = A[M,K] W[N,K]^T (+ bias, + addend, tanh) for M >= 512 -- the speaker's
// teacher-forced head over all S*B = 8 000 rows (t_text, h~, vocabulary projection and their data gradients), the beam
// search's flat decoder steps over 2 560 states.  gemm_nt_kernel streams its operands straight from global memory (7
// FLOPs per operand byte: 38-55 TFLOP/s at M = 8 000); here a workgroup forms a 128 x 128 tile from 32-deep stages of
// both operands staged through LDS as three bf16 planes each (error-free three-way split, sf_split.h: fp32 accuracy on
// the bf16 matrix cores, 32 FLOPs per operand byte).  512 threads = 2 x 4 waves of 64 x 32 outputs (8 MFMA tiles, hi / lo
// accumulators); per stage a thread fetches 2 + 2 float4, splits them once and stores 12 x 8 bytes; double-buffered LDS
// (96 KB), the next stage's global loads in flight under the current stage's 48 MFMAs per wave; one barrier per stage.
// Row layout of a plane: 64 bytes per row and stage = four 16-byte chunks, chunk kk = {k = 4 kk .. +3 | 16 + 4 kk .. +3}:
// what one lane feeds one v_mfma_f32_16x16x32_bf16 (the K order inside an MFMA is free as long as A and B agree).
// ------------------------------------------------------------------------------------------------
struct NtBigArgs {
    Seg seg[3];            // up to three K segments ([x | h] W = x W_x + h W_h: the LSTM gates); K_s % 4 == 0
    int nseg;
    int M, N;
    float* y; int ldy;
    const float* bias; const float* bias2;
    const float* addend; int ld_addend;
    int epi, accumulate;
    int ksplit;            // > 1: grid.y K-splits, split z writes its partial tile to slab y + z * M * N (ldy = N, plain)
};
constexpr int NB_T = 128, NB_K = 32, NB_PLANE = NB_T * 64;          // bytes per plane and stage (128 rows x 64 B)

__device__ __forceinline__ void nt_big_body(const NtBigArgs& a, const int tile, const int split_of_block) {
    extern __shared__ __attribute__((aligned(16))) unsigned char nb_smem[];   // [2 buffers][A | W][3 planes][128 rows][64 B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kk = lane >> 4;
    const int wm = wave >> 2, wn = wave & 3;                          // 64-row half, 32-column quarter of the tile
    // tile map (XCD-aware): workgroup b runs on XCD b % 8, each XCD with its own L2.  XCD x takes a CONTIGUOUS run of the
    // tiles in column-fastest order, so the column blocks that share a 128-row slice of A are co-resident on one XCD and
    // stream it through one L2 (before: rows-fastest over all XCDs, every slice of A fetched once per column block from
    // beyond the L2s).  Measured with the bank swizzle below: +2 .. +6 % (profiles/r05_zz_many_row_kernel_lab.txt).
    const int mtiles = (a.M + NB_T - 1) / NB_T, ntiles = (a.N + NB_T - 1) / NB_T;
    const int tiles_all = mtiles * ntiles, xcd = tile & 7;
    const int order = xcd * (tiles_all >> 3) + min(xcd, tiles_all & 7) + (tile >> 3);
    const int m0 = (order / ntiles) * NB_T, n0 = (order % ntiles) * NB_T;
    // staging: thread -> (row = tid >> 3 (+64), float4 c4 = tid & 7 of the 32-deep stage)
    const int srow = tid >> 3, c4 = tid & 7;
    int arow[2], wrow[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        arow[p] = min(m0 + srow + 64 * p, a.M - 1);
        wrow[p] = min(n0 + srow + 64 * p, a.N - 1);
    }
    const int st0 = (a.seg[0].K + NB_K - 1) / NB_K, st1 = a.nseg > 1 ? (a.seg[1].K + NB_K - 1) / NB_K : 0,
              st2 = a.nseg > 2 ? (a.seg[2].K + NB_K - 1) / NB_K : 0;
    // stage s -> its segment and depth offset (wave-uniform selects); a segment's last stage may be partial (K_s % 4 == 0):
    // loads beyond K_s are clamped and replaced by zeros
    auto gload = [&](int s, float4 (&ra)[2], float4 (&rw)[2]) {
        const int g = s < st0 ? 0 : (s < st0 + st1 ? 1 : 2);
        const int k0 = (s - (g > 0 ? st0 : 0) - (g > 1 ? st1 : 0)) * NB_K + 4 * c4;
        const Seg& sg = a.seg[g];
        const bool ok = k0 < sg.K;
        const int kc = ok ? k0 : 0;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const float4 x = ld4(sg.A + (size_t)arow[p] * sg.lda + kc), y = ld4(sg.W + (size_t)wrow[p] * sg.ldw + kc);
            ra[p] = ok ? x : z;
            rw[p] = ok ? y : z;
        }
    };
    // Bank swizzle: a ds_read_b128 is served in groups of 16 lanes over 64 banks, and rows r, r + 4, r + 8, r + 12 of a
    // 64-byte-row plane start on the same bank: chunk kk of row r lives at position kk ^ ((-(r >> 2)) & 3), which gives the
    // four rows of every lane group four different chunk positions (unswizzled: 2-way conflicts, 8 instead of 4 LDS cycles
    // per fragment read).  srow, srow + 64 and the fragment rows 16 i + li all have (row >> 2) & 3 from their low bits.
    const int soff = srow * 64 + ((((c4 & 3) ^ (-(srow >> 2))) & 3) << 4) + ((c4 >> 2) << 3);   // this thread's 8 bytes inside a plane
    auto stage_store = [&](int buf, const float4 (&ra)[2], const float4 (&rw)[2]) {
        unsigned char* base = nb_smem + buf * (6 * NB_PLANE);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            uint2 p1, p2, p3;
            split3_f4(ra[p], p1, p2, p3);
            unsigned char* d = base + soff + p * 64 * 64;
            *reinterpret_cast<uint2*>(d) = p1;
            *reinterpret_cast<uint2*>(d + NB_PLANE) = p2;
            *reinterpret_cast<uint2*>(d + 2 * NB_PLANE) = p3;
            split3_f4(rw[p], p1, p2, p3);
            d += 3 * NB_PLANE;
            *reinterpret_cast<uint2*>(d) = p1;
            *reinterpret_cast<uint2*>(d + NB_PLANE) = p2;
            *reinterpret_cast<uint2*>(d + 2 * NB_PLANE) = p3;
        }
    };
    f32x4 hi[4][2], lo[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) hi[i][j] = lo[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int stages_all = st0 + st1 + st2;
    const int ks = a.ksplit > 1 ? a.ksplit : 1, split = split_of_block;
    const int s_lo = (int)(((long)split * stages_all) / ks), s_hi = (int)(((long)(split + 1) * stages_all) / ks);
    const int stages = s_hi - s_lo;
    // Two register sets: the loads of stage s + 2 are issued at the top of stage s and stored to LDS at the end of stage
    // s + 1 -- two stages (~1.2 us) of latency hiding; one stage is less than an HBM round trip under load.
    float4 ra[2][2], rw[2][2];
    auto compute = [&](int buf) {
        const int kpos = ((kk ^ (-(li >> 2))) & 3) * 16;
        const unsigned char* Ab = nb_smem + buf * (6 * NB_PLANE) + (wm * 64 + li) * 64 + kpos;
        const unsigned char* Wb = nb_smem + buf * (6 * NB_PLANE) + 3 * NB_PLANE + (wn * 32 + li) * 64 + kpos;
        Split8 wf[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) wf[j].p[pl] = *reinterpret_cast<const bf16x8*>(Wb + pl * NB_PLANE + j * 16 * 64);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            Split8 af;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) af.p[pl] = *reinterpret_cast<const bf16x8*>(Ab + pl * NB_PLANE + i * 16 * 64);
            mfma_split6_across<2>(af, [&](int j) -> const Split8& { return wf[j]; }, hi[i], lo[i]);
        }
    };
    const int last = s_hi - 1;
    gload(s_lo, ra[0], rw[0]);
    stage_store(0, ra[0], rw[0]);
    gload(min(s_lo + 1, last), ra[1], rw[1]);                        // set 1 <- stage 1 (in flight during stage 0)
    __syncthreads();
    for (int s = 0; s < stages; s += 2) {
        gload(min(s_lo + s + 2, last), ra[0], rw[0]);                 // set 0 <- stage s + 2
        compute(0);
        if (s + 1 < stages) stage_store(1, ra[1], rw[1]);             // stage s + 1 (issued one stage ago) -> buffer 1
        __syncthreads();
        if (s + 1 >= stages) break;
        gload(min(s_lo + s + 3, last), ra[1], rw[1]);                 // set 1 <- stage s + 3
        compute(1);
        if (s + 2 < stages) stage_store(0, ra[0], rw[0]);             // stage s + 2 -> buffer 0
        __syncthreads();
    }
    // epilogue: lane (li, kk) holds rows 4 kk + r, column li of every 16 x 16 tile
    if (ks > 1) {                                                     // K-split: the plain partial tile into this split's slab
        float* slab = a.y + (size_t)split * a.M * a.N;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 32 + j * 16 + li;
            if (col >= a.N) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = m0 + wm * 64 + i * 16 + kk * 4 + r;
                    if (row < a.M) slab[(size_t)row * a.N + col] = hi[i][j][r] + lo[i][j][r];
                }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 32 + j * 16 + li;
        if (col >= a.N) continue;
        float bsum = 0.f;
        if (a.bias) bsum += a.bias[col];
        if (a.bias2) bsum += a.bias2[col];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * 64 + i * 16 + kk * 4 + r;
                if (row >= a.M) continue;
                float v = (hi[i][j][r] + lo[i][j][r]) + bsum;
                if (a.addend) v += a.addend[(size_t)row * a.ld_addend + col];
                if (a.epi == EPI_TANH) v = tanhf(v);
                float* o = a.y + (size_t)row * a.ldy + col;
                *o = a.accumulate ? *o + v : v;
            }
    }
}

__global__ __launch_bounds__(512) void gemm_nt_big_kernel(NtBigArgs a) { nt_big_body(a, blockIdx.x, blockIdx.y); }

// SEVERAL many-row products in one launch (gemm_tn_group: the decoder's small weight gradients -- each a few dozen tiles,
// 35 dependent launches of 5-28 us between them when issued one by one): block b belongs to job j with
// first[j] <= b < first[j + 1]; inside the job blocks walk its tiles, then its K splits.
constexpr int NB_GROUP = 8;
struct NtBigGroup {
    NtBigArgs job[NB_GROUP];
    int first[NB_GROUP + 1];
    int n;
};
__global__ __launch_bounds__(512) void gemm_nt_big_group_kernel(NtBigGroup g) {
    int j = 0;
#pragma unroll
    for (int k = 1; k < NB_GROUP; ++k) j += (k < g.n && (int)blockIdx.x >= g.first[k]) ? 1 : 0;
    const NtBigArgs& a = g.job[j];
    const int local = (int)blockIdx.x - g.first[j];
    const int tiles = ((a.M + NB_T - 1) / NB_T) * ((a.N + NB_T - 1) / NB_T);
    nt_big_body(a, local % tiles, local / tiles);
}

// Several transposes in one launch: problem p's 32 x 32 tiles are blocks first[p] .. first[p + 1] - 1
constexpr int TR_GROUP = 16;
struct TrGroup {
    struct P { const float* src; float* dst; int lds, R, C; } p[TR_GROUP];
    int first[TR_GROUP + 1];
    int n;
};
__global__ __launch_bounds__(256) void transpose_group_kernel(TrGroup g) {
    __shared__ float tile[32][33];
    int j = 0;
#pragma unroll
    for (int k = 1; k < TR_GROUP; ++k) j += (k < g.n && (int)blockIdx.x >= g.first[k]) ? 1 : 0;
    const TrGroup::P q = g.p[j];
    const int local = (int)blockIdx.x - g.first[j], ct = (q.C + 31) / 32;
    const int c0 = (local % ct) * 32, r0 = (local / ct) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8)
        if (r0 + i < q.R && c0 + tx < q.C) tile[i][tx] = q.src[(size_t)(r0 + i) * q.lds + c0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < q.C && r0 + tx < q.R) q.dst[(size_t)(c0 + i) * q.R + r0 + tx] = tile[tx][i];
}

// out_j (+)= sum of job j's K-split slabs, all jobs in one launch (grid.y = job; fixed slab order: deterministic)
struct RedGroup {
    struct J { const float* slabs; float* y; int ks, M, N, ldy, accumulate; } j[NB_GROUP];
};
__global__ __launch_bounds__(256) void reduce_slabs_group_kernel(RedGroup g) {
    const RedGroup::J a = g.j[blockIdx.y];
    const size_t total = (size_t)a.M * a.N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / a.N), col = (int)(i % a.N);
        float v = 0.f;
        for (int s = 0; s < a.ks; ++s) v += a.slabs[(size_t)s * total + i];
        float* o = a.y + (size_t)row * a.ldy + col;
        *o = a.accumulate ? *o + v : v;
    }
}

// ------------------------------------------------------------------------------------------------
// Short reductions (K <= 1024: every Linear of the decode step except the LSTM gates, and the
// recurrent h*W_hh^T of the encoder / speaker decoder).  These are LATENCY bound: a wave that walks
// its K range chunk by chunk pays one HBM round trip per chunk (measured 13 us for K = 512).  So the
// K range of one output tile is split over the waves of the block, every wave issues ALL its loads
// before the first MFMA (<= CPW chunks of (1 + MT) float4 per lane), and the partial tiles meet in
// LDS where the epilogue runs: one HBM round trip per launch, no split-K slabs, no second kernel.
// ------------------------------------------------------------------------------------------------
// (Seg2, Frags, upfront_load / upfront_mma, SmallArgs and the kernel body live in sf_gemm_small.h:
// the paired launches of sf_attention.hip run the same body next to an attention body.)
template <int MT, int CPW>
__global__ __launch_bounds__(SMALL_WAVES * 64) void gemm_nt_small_kernel(SmallArgs a) {
    small_gemm_body<MT, CPW>(a, blockIdx.x, blockIdx.y);
}
template <int MT, int CPW>     // the product that completes dh1 + the LSTM cell's pointwise backward (see small_gemm_body)
__global__ __launch_bounds__(SMALL_WAVES * 64) void gemm_nt_small_pw_kernel(SmallArgs a, LstmPwBwd pw) {
    small_gemm_body<MT, CPW, true, true>(a, blockIdx.x, blockIdx.y, &pw);
}
template <int MT, int CPW>     // with the fused backward epilogues (see small_gemm_body)
__global__ __launch_bounds__(SMALL_WAVES * 64) void gemm_nt_small_x_kernel(SmallArgs a) {
    small_gemm_body<MT, CPW, true>(a, blockIdx.x, blockIdx.y);
}

// Fused recurrent LSTM step (nn.LSTMCell / one nn.LSTM time step): block = 16 waves = the 4 gate
// tiles (i,f,g,o) of one 16-row x 16-hidden-unit patch x 4 K-slices; gates meet in LDS and the cell
// update happens in the same kernel: ONE launch per time step.  grid (H/16, ceil(B/16)).
constexpr int LSTM_KS = 4;

template <int CPW>
__global__ __launch_bounds__(4 * LSTM_KS * 64) void lstm_step_fused_kernel(LstmStepArgs p) {
    __shared__ float s_g[4][LSTM_KS][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gate = wave & 3, ksl = wave >> 2;
    const int slice = blockIdx.x, m0 = blockIdx.y * 16;
    const int li = lane & 15, kk = lane >> 4;
    const int H = p.H;

    // tail operands (hoisted input product, biases) are fetched FIRST: xg[t] is a cold HBM read and
    // would otherwise sit, fully exposed, between the LDS reduction and the cell update
    const int prow = threadIdx.x >> 4, pcol = threadIdx.x & 15;
    const int pb = m0 + prow, pj = slice * 16 + pcol;
    const bool ptail = threadIdx.x < 256 && pb < p.B;
    float pre[4] = {0.f, 0.f, 0.f, 0.f};
    float c0v = 0.f;
    LstmLive lv{true, 0.f};
    {   // straight-line: rows / threads outside the tail read a clamped (valid) address
        const int qb = min(pb, p.B - 1);
        c0v = p.pw.c0[qb * H + pj];
        lv = lstm_live_load(p.pw, qb, pj);
        float bi[4], bh[4], xv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bi[g] = p.b_ih[g * H + pj];
            bh[g] = p.b_hh[g * H + pj];
        }
        if (p.xg) {                                              // block-uniform
            const size_t xr = p.xg_index ? (size_t)p.xg_index[(size_t)qb * max(p.xg_index_ld, 1)] : (size_t)qb;
#pragma unroll
            for (int g = 0; g < 4; ++g) xv[g] = p.xg[xr * 4 * H + g * H + pj];
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) pre[g] = bi[g] + bh[g] + xv[g];
    }

    Seg2 sg;
    sg.s0 = Seg{p.h0, H, p.w_hh, H, H};
    sg.n0 = (H + 15) >> 4;
    sg.s1 = p.x ? Seg{p.x, p.ldx, p.w_ih, p.I, p.I} : sg.s0;
    sg.total = sg.n0 + (p.x ? ((p.I + 15) >> 4) : 0);
    const int n = gate * H + slice * 16 + li;
    int mrow[1] = {min(m0 + li, p.B - 1)};
    const int c_lo = (ksl * sg.total) / LSTM_KS, c_hi = ((ksl + 1) * sg.total) / LSTM_KS;
    f32x4 acc[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
    if (c_hi > c_lo) {
        Frags<1, CPW> f;
        upfront_load<1, CPW>(f, sg, c_lo, c_hi, n, mrow, kk);
        upfront_mma<1, CPW>(f, c_hi - c_lo, acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) s_g[gate][ksl][(kk * 4 + r) * 16 + li] = acc[0][r];
    __syncthreads();

    if (!ptail) return;
    float g4[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float v = pre[g];
#pragma unroll
        for (int k = 0; k < LSTM_KS; ++k) v += s_g[g][k][threadIdx.x];
        g4[g] = v;
    }
    lstm_cell_update(p.pw, pb, pj, g4, c0v, lv);
}

// The same step with a 32-row x 8-hidden-unit patch per block (B > 16).  A recurrent step is bound by
// the bytes ONE block pulls through its CU (timestamped: the loads of the 16 x 16 patch -- 128 KB of
// W_hh, the 32 KB of h fetched once per gate wave = 256 KB issued -- keep landing for ~6 us of an
// 11.5 us step; the MFMAs take 1.7 us).  Per block the bytes are (4 hu + rows) x 2 KB, smallest at
// rows = 4 hu: 8 units x 32 rows = 128 KB, and with the 16 waves as 16 K-slices that each hold both
// m-tiles and both n-tiles nothing is fetched twice.  grid (H/8, ceil(B/32)).
//   n-tile q = gates (2q, 2q+1) x 8 units: column li -> W_hh row (2q + li/8) H + 8 slice + li%8.
constexpr int LSTMW_WAVES = 16;

template <int CPW>
__global__ __launch_bounds__(LSTMW_WAVES * 64) void lstm_step_wide_kernel(LstmStepArgs p) {
    __shared__ float s_red[LSTMW_WAVES / 2][4][256];     // 32 KB: partial tiles (t, q) of 8 waves
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int slice = blockIdx.x, m0 = blockIdx.y * 32;
    const int li = lane & 15, kk = lane >> 4;
    const int H = p.H;

    // Tail operands.  The block updates 256 (row, unit) elements (threads 0..255 do it); their gate
    // biases and hoisted input rows are 12 scattered dword loads per element -- issued by all 1024
    // threads for their own element they were more vector-memory instructions than the fragments
    // (and ahead of them in the queue).  Instead thread (g, e) fetches gate g of element e: 4 loads,
    // the sums meet in LDS with the partial tiles.
    __shared__ float s_pre[4][256];
    const int pe = threadIdx.x & 255, pg = threadIdx.x >> 8;
    const int prow = pe >> 3, pcol = pe & 7;
    const int pb = m0 + prow, pj = slice * 8 + pcol;
    const bool ptail = threadIdx.x < 256 && pb < p.B;
    const int qb = min(pb, p.B - 1);
    float pre_g;
    {   // straight-line: rows outside the batch read a clamped (valid) address
        const float bi = p.b_ih[pg * H + pj], bh = p.b_hh[pg * H + pj];
        float xv = 0.f;
        if (p.xg) {                                              // block-uniform
            const size_t xr = p.xg_index ? (size_t)p.xg_index[(size_t)qb * max(p.xg_index_ld, 1)] : (size_t)qb;
            xv = p.xg[xr * 4 * H + pg * H + pj];
        }
        pre_g = bi + bh + xv;
    }

    // this wave's K-slice of all four tiles: every fragment is loaded before the first MFMA
    const int n0c = H >> 4;
    const int total = n0c + (p.x ? ((p.I + 15) >> 4) : 0);
    const int c_lo = (wave * total) / LSTMW_WAVES, c_hi = ((wave + 1) * total) / LSTMW_WAVES;
    const int nrow[2] = {(li >> 3) * H + slice * 8 + (li & 7), (2 + (li >> 3)) * H + slice * 8 + (li & 7)};
    const int mrow[2] = {min(m0 + li, p.B - 1), min(m0 + 16 + li, p.B - 1)};
    float4 fa[CPW][2], fb[CPW][2];
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
        const int c = min(min(c_lo + i, max(c_hi - 1, c_lo)), total - 1);   // clamped: surplus slots re-load a valid chunk
        const bool second = c >= n0c;                            // wave-uniform
        const int lc = second ? c - n0c : c;
        const float* W = second ? p.w_ih : p.w_hh;
        const float* A = second ? p.x : p.h0;
        const int ldw = second ? p.I : H;
        const int lda = second ? p.ldx : H;
        const int K = second ? p.I : H;
        const int k = lc * 16 + 4 * kk;
        const bool ok = k < K && c_lo + i < c_hi;                // partial last chunk / surplus slot
        const int kc = k < K ? k : 0;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float4 bv = ld4(W + (size_t)nrow[q] * ldw + kc);
            fb[i][q] = ok ? bv : z;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float4 av = ld4(A + (size_t)mrow[t] * lda + kc);
            fa[i][t] = ok ? av : z;
        }
    }
    // state operands of the tail threads (waves 0-3), behind the fragment loads in the queue
    float c0v = 0.f;
    LstmLive lv{true, 0.f};
    if (threadIdx.x < 256) {                                     // wave-uniform
        c0v = p.pw.c0[qb * H + pj];
        lv = lstm_live_load(p.pw, qb, pj);
    }
    f32x4 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 2; ++q) acc[t][q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < CPW; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    acc[t][q] = mfma16(comp(fa[i][t], c), comp(fb[i][q], c), acc[t][q]);

    s_pre[pg][pe] = pre_g;
    // 16 partial tile sets -> 8 (through LDS) -> the cell update sums the 8
    if (wave >= LSTMW_WAVES / 2) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    s_red[wave - LSTMW_WAVES / 2][t * 2 + q][(kk * 4 + r) * 16 + li] = acc[t][q][r];
    }
    __syncthreads();
    if (wave < LSTMW_WAVES / 2) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float* sp = &s_red[wave][t * 2 + q][(kk * 4 + r) * 16 + li];
                    *sp += acc[t][q][r];                          // (same thread wrote / reads this slot)
                }
    }
    __syncthreads();

    if (!ptail) return;
    float g4[4];
    const int tt = prow >> 4, rr = prow & 15;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float v = s_pre[g][pe];
        const int e = rr * 16 + (g & 1) * 8 + pcol;
#pragma unroll
        for (int k = 0; k < LSTMW_WAVES / 2; ++k) v += s_red[k][tt * 2 + (g >> 1)][e];
        g4[g] = v;
    }
    lstm_cell_update(p.pw, pb, pj, g4, c0v, lv);
}

// Fused BACKWARD time step of the encoder LSTM: block = 16 rows x 16 hidden units, 8 waves = 8
// K-slices of dh_{t+1}[tile] = dgates_{t+1}[16 rows, 4H] . W_hh^T[16 units, 4H] (all loads up front,
// partial tiles meet in LDS), then the cell backward of step t for the tile's 256 elements in the
// same launch: one dependent launch per time step instead of two.  grid (H/16, ceil(B/16)).
constexpr int BWS_WAVES = 8;

template <int CPW>
__global__ __launch_bounds__(BWS_WAVES * 64) void lstm_bwd_step_fused_kernel(LstmBwdStepArgs p) {
    __shared__ float s_part[BWS_WAVES][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, kk = lane >> 4;
    const int H = p.H, B = p.B;
    const int j0 = blockIdx.x * 16, m0 = blockIdx.y * 16;

    // operands of the cell backward, fetched with the fragments (threads 0..255 own one element)
    const int prow = threadIdx.x >> 4, pcol = threadIdx.x & 15;
    const int pb = m0 + prow, pj = j0 + pcol;
    const bool ptail = threadIdx.x < 256 && pb < B;
    const int qb = min(pb, B - 1);
    const size_t idx = (size_t)qb * H + pj;
    const float* gp = p.gates + (size_t)qb * 4 * H + pj;
    const float ig = gp[0], fg = gp[H], gg = gp[2 * H], og = gp[3 * H];
    const float c1 = p.c1[idx], c0 = p.c0[idx];
    const float dc = p.dc_in[idx];
    float dh = p.dh_in[idx];
    const bool dead = p.t >= p.lengths[qb];
    if (p.dctx) {                                                // block-uniform
        float v = p.dctx[((size_t)qb * p.T + p.t) * H + pj];
        if (p.ctx_drop.on()) {
            const uint32_t rk = drop_key(p.ctx_drop, (uint32_t)(p.ctx_drop.row0 + qb));
            v = dropout_keep(rk, (uint32_t)(p.t * H + pj), p.ctx_drop.thresh) ? v * p.ctx_drop.scale : 0.f;
        }
        dh += v;
    }

    f32x4 acc[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
    if (p.dgates_next) {                                         // block-uniform (absent at t = T-1)
        Seg2 sg;
        sg.s0 = Seg{p.dgates_next, 4 * H, p.w_hh_t, 4 * H, 4 * H};
        sg.n0 = (4 * H) >> 4;
        sg.s1 = sg.s0;
        sg.total = sg.n0;
        const int n = j0 + li;
        int mrow[1] = {min(m0 + li, B - 1)};
        const int c_lo = (wave * sg.total) / BWS_WAVES, c_hi = ((wave + 1) * sg.total) / BWS_WAVES;
        if (c_hi > c_lo) {
            Frags<1, CPW> f;
            upfront_load<1, CPW>(f, sg, c_lo, c_hi, n, mrow, kk);
            upfront_mma<1, CPW>(f, c_hi - c_lo, acc);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) s_part[wave][(kk * 4 + r) * 16 + li] = acc[0][r];
    __syncthreads();
    if (!ptail) return;
#pragma unroll
    for (int w = 0; w < BWS_WAVES; ++w) dh += s_part[w][threadIdx.x];

    float* dg = p.dgates + (size_t)pb * 4 * H + pj;
    const float tc = tanhf(c1);
    const float dout = dh * tc;
    const float dcl = dc + dh * og * (1.f - tc * tc);
    // packed sequence: a step that did not happen passes dh / dc through, dgates = 0
    dg[0] = dead ? 0.f : dcl * gg * ig * (1.f - ig);
    dg[H] = dead ? 0.f : dcl * c0 * fg * (1.f - fg);
    dg[2 * H] = dead ? 0.f : dcl * ig * (1.f - gg * gg);
    dg[3 * H] = dead ? 0.f : dout * og * (1.f - og);
    p.dc_out[(size_t)pb * H + pj] = dead ? dc : dcl * fg;
    p.dh_out[(size_t)pb * H + pj] = dead ? dh : 0.f;
}

// ------------------------------------------------------------------------------------------------
// NN: C[M,N] = A[M,K] * W[K,N]     (dX = dY * W; A K-contiguous, W N-contiguous)
// grid (ceil(N/256), ksplit, mblocks); wave = MT m-tiles x 64 columns (4 virtual tiles).
// ------------------------------------------------------------------------------------------------
struct NnArgs {
    const float* A;
    int lda;
    const float* W;
    int ldw;
    int M, N, K;
    int ksplit;
    float* out;      // slabs [ksplit][M][N] or final
    int ldo;
    int accumulate;  // direct mode only
};

template <int MT>
__global__ __launch_bounds__(256) void gemm_nn_kernel(NnArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int n0 = (blockIdx.x * 4 + wave) * 64;
    if (n0 >= a.N) return;
    const int split = blockIdx.y;
    const int m0 = blockIdx.z * (16 * MT);
    const int li = lane & 15, kk = lane >> 4;
    const int ncol = n0 + 4 * li;                 // this lane's 4 consecutive columns
    const bool colok = ncol < a.N;                // N % 4 == 0
    const int ncl = colok ? ncol : 0;

    int mrow[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) mrow[t] = min(m0 + 16 * t + li, a.M - 1);

    f32x4 acc[MT][4];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[t][v] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int chunks = (a.K + 15) >> 4;
    const int c0 = (int)(((long)split * chunks) / a.ksplit);
    const int c1 = (int)(((long)(split + 1) * chunks) / a.ksplit);
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);

    for (int ch = c0; ch < c1; ++ch) {
        const int k = ch * 16 + 4 * kk;           // this lane's 4 consecutive k
        float4 bw[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
            bw[c] = (k + c < a.K) ? ld4(a.W + (size_t)(k + c) * a.ldw + ncl) : z;
        float4 av[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t) av[t] = (k < a.K) ? ld4(a.A + (size_t)mrow[t] * a.lda + k) : z;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int v = 0; v < 4; ++v)
#pragma unroll
                for (int t = 0; t < MT; ++t)
                    acc[t][v] = mfma16(comp(av[t], c), comp(bw[c], v), acc[t][v]);
    }

    if (!colok) return;
    float* out = a.out + (a.ksplit > 1 ? (size_t)split * a.M * a.N : 0);
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + 16 * t + kk * 4 + r;
            if (row >= a.M) continue;
            float4 v = make_float4(acc[t][0][r], acc[t][1][r], acc[t][2][r], acc[t][3][r]);
            float4* p = reinterpret_cast<float4*>(out + (size_t)row * a.ldo + ncol);
            if (a.ksplit == 1 && a.accumulate) {
                const float4 o = *p;
                v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
            }
            *p = v;
        }
}

// ------------------------------------------------------------------------------------------------
// TN: out[P,Q] (+)= Y[M,P]^T * X[M,Q]   (weight gradients; reduction over the batch rows)
// grid (ceil(Q/256), ceil(P/64)); wave = 64 P-rows x 64 Q-cols (4 x 4 virtual tiles).
// ------------------------------------------------------------------------------------------------
struct TnArgs {
    const float* Y;
    int ldy;
    const float* X;
    int ldx;
    int M, P, Q;
    float* out;
    int ldo;
    int accumulate;
    int msplit;            // > 1: block z reduces rows [z*M/msplit, (z+1)*M/msplit) into slab z of `out`
};

__global__ __launch_bounds__(256) void gemm_tn_kernel(TnArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int q0 = (blockIdx.x * 4 + wave) * 64;
    if (q0 >= a.Q) return;
    const int p0 = blockIdx.y * 64;
    if (a.msplit > 1) {    // row range of this split; its slab is a dense [P, Q] matrix
        const int z = blockIdx.z;
        const int m_lo = (int)(((long)z * a.M) / a.msplit), m_hi = (int)(((long)(z + 1) * a.M) / a.msplit);
        a.Y += (size_t)m_lo * a.ldy;
        a.X += (size_t)m_lo * a.ldx;
        a.M = m_hi - m_lo;
        a.out += (size_t)z * a.P * a.ldo;
    }
    const int li = lane & 15, kk = lane >> 4;
    const int pc = p0 + 4 * li;                 // Y columns pc..pc+3 (must be readable: ldy padded)
    const int qc = q0 + 4 * li;
    const bool pok = pc < a.ldy;                // stay inside the row (caller pads P to %4 via ldy)
    const bool qok = qc < a.Q;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // 8-deep clamped prefetch ring over the batch rows (4 rows per step): loads of steps s+1..s+7
    // are in flight while step s feeds 16 MFMAs; clamped (not predicated) addresses keep the loop
    // body branch-free, rows >= M contribute through a zero multiplier.
    const int steps = (a.M + 3) >> 2;
    const int pcl = pok ? pc : 0, qcl = qok ? qc : 0;
    auto ld = [&](int s, float4& yv, float4& xv) {
        const int m = min(4 * min(s, steps - 1) + kk, a.M - 1);
        yv = ld4(a.Y + (size_t)m * a.ldy + pcl);
        xv = ld4(a.X + (size_t)m * a.ldx + qcl);
    };
    auto mma = [&](int s, const float4& yv, const float4& xv) {
        const float z = (4 * s + kk < a.M && pok) ? 1.f : 0.f;     // tail rows / columns beyond P
        const float4 y = make_float4(yv.x * z, yv.y * z, yv.z * z, yv.w * z);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(comp(y, i), comp(xv, j), acc[i][j]);
    };
    constexpr int RING = 8;
    float4 yr[RING], xr[RING];
#pragma unroll
    for (int i = 0; i < RING; ++i) ld(i, yr[i], xr[i]);
    int s4 = 0;
    for (; s4 + RING <= steps; s4 += RING) {
#pragma unroll
        for (int i = 0; i < RING; ++i) {
            mma(s4 + i, yr[i], xr[i]);
            ld(s4 + RING + i, yr[i], xr[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < RING - 1; ++i)
        if (s4 + i < steps) mma(s4 + i, yr[i], xr[i]);

    if (!qok) return;
    // read-modify-write of the gradient tile: all 16 reads first (straight-line, clamped rows), then
    // the adds and stores -- one memory round trip instead of sixteen
    float4 w[4][4];
    if (a.accumulate) {                                  // block-uniform
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int p = min(p0 + 4 * (kk * 4 + r) + i, a.P - 1);
                w[i][r] = ld4(a.out + (size_t)p * a.ldo + qc);
            }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int p = p0 + 4 * (kk * 4 + r) + i;     // tile row (kk*4+r) is virtual: stride-4 rows
            float4 v = make_float4(acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]);
            if (a.accumulate) {
                v.x += w[i][r].x; v.y += w[i][r].y; v.z += w[i][r].z; v.w += w[i][r].w;
            }
            if (p < a.P) *reinterpret_cast<float4*>(a.out + (size_t)p * a.ldo + qc) = v;
        }
}

// ------------------------------------------------------------------------------------------------
// split-K reduce + epilogue
// ------------------------------------------------------------------------------------------------
struct RedArgs {
    const float* slabs;
    int ks;
    int M, N;
    float* y;
    int ldy;
    const float* bias;
    const float* bias2;
    const float* mul;
    float* y_pre;
    int ldy_pre;
    int epi;
    int accumulate;
};

__global__ __launch_bounds__(256) void reduce_slabs_kernel(RedArgs a) {
    const size_t total = (size_t)a.M * a.N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / a.N), col = (int)(i % a.N);
        float v = 0.f;
        for (int s = 0; s < a.ks; ++s) v += a.slabs[(size_t)s * total + i];
        if (a.bias) v += a.bias[col];
        if (a.bias2) v += a.bias2[col];
        if (a.epi == EPI_TANH) v = tanhf(v);
        if (a.epi == EPI_MUL) {
            if (a.y_pre) a.y_pre[(size_t)row * a.ldy_pre + col] = v;
            v *= a.mul[col];
        }
        float* o = a.y + (size_t)row * a.ldy + col;
        *o = a.accumulate ? *o + v : v;
    }
}

// out[n] (+)= sum_m Y[m,n]: block = 64 columns x 16 row groups, coalesced 256-B row segments.
// grid (ceil(N/64), msplit): with msplit > 1 block (x, z) sums rows [z*M/msplit, ...) into
// part[z][n] and a second tiny launch adds the parts (deterministic).  out2 (optional) receives
// the same sums (LSTM: b_ih and b_hh have the same gradient).
__global__ __launch_bounds__(1024) void colsum_kernel(const float* Y, int ldy, int M, int N,
                                                      float* out, float* out2, int accumulate,
                                                      int msplit, float* part) {
    __shared__ float s_p[16][64];
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + c;
    const int z = blockIdx.y;
    const int m_lo = (int)(((long)z * M) / msplit), m_hi = (int)(((long)(z + 1) * M) / msplit);
    const int nc = min(n, N - 1);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;      // 4 independent loads in flight per thread
    int m = m_lo + g;
    for (; m + 48 < m_hi; m += 64) {
        s0 += Y[(size_t)m * ldy + nc];
        s1 += Y[(size_t)(m + 16) * ldy + nc];
        s2 += Y[(size_t)(m + 32) * ldy + nc];
        s3 += Y[(size_t)(m + 48) * ldy + nc];
    }
    for (; m < m_hi; m += 16) s0 += Y[(size_t)m * ldy + nc];
    s_p[g][c] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && n < N) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += s_p[k][c];
        if (msplit > 1) {
            part[(size_t)z * N + n] = t;
        } else {
            out[n] = accumulate ? out[n] + t : t;
            if (out2) out2[n] = accumulate ? out2[n] + t : t;
        }
    }
}

__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* part, int msplit, int N,
                                                            float* out, float* out2, int accumulate) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float t = 0.f;
    for (int z = 0; z < msplit; ++z) t += part[(size_t)z * N + n];
    out[n] = accumulate ? out[n] + t : t;
    if (out2) out2[n] = accumulate ? out2[n] + t : t;
}

// The same product with the operands shared through LDS: block = 4 waves = 2 x 2 tiles of 64 x 64,
// i.e. 128 P-rows x 128 Q-cols; every 16 batch rows are staged once ([16][128] of Y and of X, 16 KB,
// double buffered) and read back as fragments by the two waves that need them.  The register-
// streaming kernel above pulls 2 KB per wave and 16 MFMAs -- at the matrix-pipe rate that is the
// ~40 GB/s one CU sustains; this one pulls half of that.  A fragment row is 256 contiguous bytes
// per 16 lanes: conflict-free without padding.
constexpr int TNT_BM = 16, TNT_B = 128;

__global__ __launch_bounds__(256) void gemm_tn_tiled_kernel(TnArgs a) {
    __shared__ float4 Ys[2][TNT_BM][TNT_B / 4];
    __shared__ float4 Xs[2][TNT_BM][TNT_B / 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wp = wave >> 1, wq = wave & 1;
    const int p0b = blockIdx.y * TNT_B, q0b = blockIdx.x * TNT_B;
    if (a.msplit > 1) {    // row range of this split; its slab is a dense [P, Q] matrix
        const int z = blockIdx.z;
        const int m_lo = (int)(((long)z * a.M) / a.msplit), m_hi = (int)(((long)(z + 1) * a.M) / a.msplit);
        a.Y += (size_t)m_lo * a.ldy;
        a.X += (size_t)m_lo * a.ldx;
        a.M = m_hi - m_lo;
        a.out += (size_t)z * a.P * a.ldo;
    }
    const int li = lane & 15, kk = lane >> 4;
    const int srow = tid >> 5, sc4 = tid & 31;          // staging: 8 rows x 32 float4 per pass, 2 passes
    const int pcl = min(p0b + 4 * sc4, a.ldy - 4);      // clamped: columns beyond P / Q are never stored
    const int qcl = min(q0b + 4 * sc4, a.Q - 4);
    const int stages = (a.M + TNT_BM - 1) / TNT_BM;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    float4 ry[2], rx[2];
    auto gload = [&](int st) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int m = min(st * TNT_BM + srow + 8 * h, a.M - 1);
            ry[h] = ld4(a.Y + (size_t)m * a.ldy + pcl);
            rx[h] = ld4(a.X + (size_t)m * a.ldx + qcl);
        }
    };
    auto lstore = [&](int st, int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const bool ok = st * TNT_BM + srow + 8 * h < a.M;    // rows beyond M contribute zeros
            Ys[buf][srow + 8 * h][sc4] = ok ? ry[h] : make_float4(0.f, 0.f, 0.f, 0.f);
            Xs[buf][srow + 8 * h][sc4] = rx[h];
        }
    };
    gload(0);
    lstore(0, 0);
    __syncthreads();
    for (int st = 0; st < stages; ++st) {
        const int buf = st & 1;
        gload(min(st + 1, stages - 1));                          // clamped, not predicated
#pragma unroll
        for (int s = 0; s < TNT_BM / 4; ++s) {
            const float4 y = Ys[buf][4 * s + kk][wp * 16 + li];
            const float4 x = Xs[buf][4 * s + kk][wq * 16 + li];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(comp(y, i), comp(x, j), acc[i][j]);
        }
        lstore(st + 1, buf ^ 1);                                 // (st + 1 == stages: all rows >= M -> zeros)
        __syncthreads();
    }

    const int p0 = p0b + wp * 64, qc = q0b + wq * 64 + 4 * li;
    if (qc >= a.Q) return;
    float4 w[4][4];
    if (a.accumulate) {                                  // block-uniform
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int p = min(p0 + 4 * (kk * 4 + r) + i, a.P - 1);
                w[i][r] = ld4(a.out + (size_t)p * a.ldo + qc);
            }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int p = p0 + 4 * (kk * 4 + r) + i;     // tile row (kk*4+r) is virtual: stride-4 rows
            float4 v = make_float4(acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]);
            if (a.accumulate) {
                v.x += w[i][r].x; v.y += w[i][r].y; v.z += w[i][r].z; v.w += w[i][r].w;
            }
            if (p < a.P) *reinterpret_cast<float4*>(a.out + (size_t)p * a.ldo + qc) = v;
        }
}

// ------------------------------------------------------------------------------------------------
// TN on the bf16 matrix cores (round 4): the weight gradients out[P,Q] (+)= Y[M,P]^T X[M,Q] as bf16x6 split products
// (sf_split.h: fp32-class accuracy, 6/16 of the fp32-MFMA time).  v_mfma_f32_16x16x32_bf16 wants, per lane, EIGHT
// consecutive values of the reduction index for one output row / column -- here eight different batch rows m of one
// column of Y (or X): the operands are transposed on their way into LDS.
//   block = 4 waves = 2 x 2 wave tiles of 64 x 64 -> 128 P-rows x 128 Q-cols; a stage = 64 batch rows.
//   staging: a thread loads an 8 (m) x 4 (columns) patch of Y and of X (8 + 8 float4), splits each COLUMN's eight values
//   into three bf16x8 pieces and stores each as one ds_write_b128 into plane[piece][column][m]: LDS rows = tile columns,
//   64 m = 128 bytes, 16-byte chunks XOR-swizzled with (row >> 1) -- the lanes of every write and of every fragment
//   read spread evenly over the eight chunk positions.  One stage buffer (96 KB: 2 operands x 3 planes x 16 KB, one
//   workgroup per CU); the next stage's global loads are in flight in registers while the 192 MFMAs of a stage run.
//   MEASURED (round 4): 3.8 us per stage where the MFMAs alone take 1.3 -- the split costs ~8 VALU operations per value
//   and every element of Y is split again by each of the Q / 128 column blocks; with one wave per SIMD the split, the
//   LDS traffic and the matrix pipe take turns.  It wins where the reduction is deep and the output small (the
//   encoder's dW_hh over T*B = 8 000 rows: 196 -> 131 us) and loses on the decoder's dW_ih (320 -> 351 us), so the
//   dispatch uses it for M >= 4096 only.  A variant that splits stage s+1 column by column between the MFMA groups
//   of stage s (32-row stages, two LDS buffers, one barrier per stage) was built and is slower still (210 us on the
//   encoder shape: 256 AGPRs, twice the barriers).  The next step is operands split ONCE (a pre-pass writing bf16
//   planes [column][m]) so that the product kernel only copies.
//   split-M over grid.z into slabs for small outputs (same reduce_slabs as the fp32 kernels: deterministic order).
// ------------------------------------------------------------------------------------------------
constexpr int TNS_B = 128, TNS_K = 64, TNS_ROWB = 128;
constexpr int TNS_PLANE = TNS_B * TNS_ROWB;           // one piece plane of one operand: 16 KB
constexpr int TNS_LDS = 2 * 3 * TNS_PLANE;

__global__ __launch_bounds__(256, 1) void gemm_tn_split_kernel(TnArgs a) {
    extern __shared__ __align__(16) unsigned char tns_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wp = wave >> 1, wq = wave & 1;
    const int p0b = blockIdx.y * TNS_B, q0b = blockIdx.x * TNS_B;
    if (a.msplit > 1) {    // row range of this split; its slab is a dense [P, Q] matrix
        const int z = blockIdx.z;
        const int m_lo = (int)(((long)z * a.M) / a.msplit), m_hi = (int)(((long)(z + 1) * a.M) / a.msplit);
        a.Y += (size_t)m_lo * a.ldy;
        a.X += (size_t)m_lo * a.ldx;
        a.M = m_hi - m_lo;
        a.out += (size_t)z * a.P * a.ldo;
    }
    const int mg = lane >> 3, cg = wave * 8 + (lane & 7);       // staging patch: rows 8 mg .. +7 of the stage, columns 4 cg .. +3
    const int li = lane & 15, kk = lane >> 4;
    const int stages = (a.M + TNS_K - 1) / TNS_K;
    const float* yp = a.Y + p0b + 4 * cg;
    const float* xp = a.X + q0b + 4 * cg;

    f32x4 hi[4][4], lo[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) hi[i][j] = lo[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    float4 ry[8], rx[8];
    auto gload = [&](int st) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int m = min(st * TNS_K + 8 * mg + r, a.M - 1);  // clamped, not predicated
            ry[r] = ld4(yp + (size_t)m * a.ldy);
            rx[r] = ld4(xp + (size_t)m * a.ldx);
        }
    };
    auto lstore = [&](int st) {
        const int live = a.M - (st * TNS_K + 8 * mg);             // rows of this patch inside the batch
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float z = r < live ? 1.f : 0.f;                 // rows beyond M contribute zeros
            ry[r] = make_float4(ry[r].x * z, ry[r].y * z, ry[r].z * z, ry[r].w * z);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int row = 4 * cg + c;
            const int off = row * TNS_ROWB + ((mg ^ (row >> 1)) & 7) * 16;
            const Split8 sy = split3_f8(make_float4(comp(ry[0], c), comp(ry[1], c), comp(ry[2], c), comp(ry[3], c)),
                                        make_float4(comp(ry[4], c), comp(ry[5], c), comp(ry[6], c), comp(ry[7], c)));
            const Split8 sx = split3_f8(make_float4(comp(rx[0], c), comp(rx[1], c), comp(rx[2], c), comp(rx[3], c)),
                                        make_float4(comp(rx[4], c), comp(rx[5], c), comp(rx[6], c), comp(rx[7], c)));
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                *reinterpret_cast<bf16x8*>(tns_smem + pl * TNS_PLANE + off) = sy.p[pl];
                *reinterpret_cast<bf16x8*>(tns_smem + (3 + pl) * TNS_PLANE + off) = sx.p[pl];
            }
        }
    };

    gload(0);
    for (int st = 0; st < stages; ++st) {
        lstore(st);
        __syncthreads();
        gload(min(st + 1, stages - 1));                           // in flight while this stage's MFMAs run
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            Split8 fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rowa = wp * 64 + 16 * i + li, rowb = wq * 64 + 16 * i + li;
                const int offa = rowa * TNS_ROWB + (((4 * ks + kk) ^ (rowa >> 1)) & 7) * 16;
                const int offb = rowb * TNS_ROWB + (((4 * ks + kk) ^ (rowb >> 1)) & 7) * 16;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    fa[i].p[pl] = *reinterpret_cast<const bf16x8*>(tns_smem + pl * TNS_PLANE + offa);
                    fb[i].p[pl] = *reinterpret_cast<const bf16x8*>(tns_smem + (3 + pl) * TNS_PLANE + offb);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) mfma_split6(fa[i], fb[j], hi[i][j], lo[i][j]);
        }
        __syncthreads();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // (the clamped prefetch of the last stage)

    // D[row 4 kk + r][col li] of every 16 x 16 tile; a gradient tile is read-modified-written 16 values at a time
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float* orow = a.out + (size_t)(p0b + wp * 64 + 16 * i + 4 * kk) * a.ldo + q0b + wq * 64 + li;
        float old[4][4];
        if (a.accumulate) {                                       // block-uniform
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) old[j][r] = orow[(size_t)r * a.ldo + 16 * j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = hi[i][j][r] + lo[i][j][r];
                if (a.accumulate) v += old[j][r];
                orow[(size_t)r * a.ldo + 16 * j] = v;
            }
    }
}

int pick_ksplit(int waves_per_split, int chunks) {
    // aim for ~2 waves per SIMD over the chip (1024 SIMDs), at least 4 16-deep chunks per split,
    // and at most 8 partial slabs (each split costs one extra write + read of the output)
    int ks = (2048 + waves_per_split - 1) / waves_per_split;
    ks = std::min(ks, std::max(1, chunks / 4));
    return std::max(1, std::min(ks, 8));
}

template <int MT>
void launch_nt(const NtArgs& a, dim3 grid, hipStream_t st) {
    static const std::string nm = "gemm_nt_kernel<" + std::to_string(MT) + ">";
    SF_LAUNCH_AS(nm.c_str(), gemm_nt_kernel<MT>, grid, dim3(256), 0, st, a);
}

template <int MT>
void launch_nn(const NnArgs& a, dim3 grid, hipStream_t st) {
    static const std::string nm = "gemm_nn_kernel<" + std::to_string(MT) + ">";
    SF_LAUNCH_AS(nm.c_str(), gemm_nn_kernel<MT>, grid, dim3(256), 0, st, a);
}

inline int red_grid(size_t total) { return (int)std::min<size_t>((total + 255) / 256, 2048); }

}  // namespace

// Short-reduction path: K <= 1024 and a small output (decode-step Linears).  Returns false when the
// streaming kernel should be used instead.
static bool small_shape(int M, int N, int chunks, int* mt, int* cpw) {
    const int mtiles = ceil_div(M, 16), ntiles = ceil_div(N, 16);
    if (chunks > 144 || (long)mtiles * ntiles > (chunks > 64 ? 512 : 2048)) return false;
    if (chunks > 128 && (long)mtiles * ntiles > 128) return false;   // (K = 2176 -> 256: the dq / dr products)
    const int c = ceil_div(chunks, SMALL_WAVES);          // chunks per wave: 1..18
    *cpw = c <= 2 ? 2 : (c <= 4 ? 4 : (c <= 8 ? 8 : (c <= 16 ? 16 : 18)));
    int m = std::max(1, std::min(mtiles, 32 / *cpw - 1)); // <= 32 float4 of loads per lane
    m = m >= 4 ? 4 : (m >= 2 ? 2 : 1);
    // a CU sustains only ~20-35 GB/s of loads, so what matters is the bytes ONE block pulls
    // (16 W rows + 16*m A rows, K deep): prefer more, lighter blocks until the chip is covered
    while (m > 1 && ntiles * ceil_div(mtiles, m) < 224) m >>= 1;
    *mt = m;
    return true;
}

template <int MT, int CPW>
static void launch_small(const SmallArgs& a, hipStream_t st) {
    dim3 grid(ceil_div(a.N, 16), ceil_div(ceil_div(a.M, 16), MT));
    static const std::string tp = "<" + std::to_string(MT) + ", " + std::to_string(CPW) + ">";
    static const std::string nx = "gemm_nt_small_x_kernel" + tp, nn = "gemm_nt_small_kernel" + tp;
    if (a.addend || a.r1_s || a.epi == EPI_TANHBWD)
        SF_LAUNCH_AS(nx.c_str(), (gemm_nt_small_x_kernel<MT, CPW>), grid, dim3(SMALL_WAVES * 64), 0, st, a);
    else
        SF_LAUNCH_AS(nn.c_str(), (gemm_nt_small_kernel<MT, CPW>), grid, dim3(SMALL_WAVES * 64), 0, st, a);
}

static void nt_shape(int M, int N, int Ktot_chunks, int* mt, int* mblocks, int* ks) {
    const int mtiles = ceil_div(M, 16), ntiles = ceil_div(N, 16);
    if (Ktot_chunks <= 64) {
        // short reduction with a large output (the hoisted encoder input product): never split K
        int m = 1;
        while (m < 8 && (long)ntiles * ceil_div(mtiles, m) > 4096) m *= 2;
        *mt = std::min(m, mtiles);
        *mblocks = ceil_div(mtiles, *mt);
        *ks = 1;
        return;
    }
    *mt = mtiles <= 8 ? mtiles : 8;
    *mblocks = ceil_div(mtiles, *mt);
    *ks = pick_ksplit(ntiles * *mblocks, Ktot_chunks);
    // the LDS-tiled kernel runs one workgroup per CU: a grid of more than 256 of them (34 n-tiles x 8
    // splits for the [100,2048] x [2048,2176] input gradient) pays a second, nearly empty round
    if (*mblocks == 1 && Ktot_chunks >= 128)
        while (*ks > 1 && ceil_div(N, 64) * *ks > 256) --*ks;
}

int linear_ksplit(int M, int N, int Ktot) {
    int mt, mb, ks;
    nt_shape(M, N, ceil_div(Ktot, 16), &mt, &mb, &ks);
    return ks;
}

size_t linear_ws_floats(int M, int N, int Ktot) {
    const int ks = linear_ksplit(M, N, Ktot);
    return ks > 1 ? (size_t)ks * M * N : 0;
}

bool linear_small_plan(const Seg* segs, int nseg, int M, int N, const LinearOut& out,
                       SmallPlan* plan) {
    if (nseg < 1 || nseg > 2) return false;
    int chunks = 0;
    for (int s = 0; s < nseg; ++s) {
        if (segs[s].K <= 0 || segs[s].K % 4 || segs[s].lda % 4 || segs[s].ldw % 4) return false;
        chunks += ceil_div(segs[s].K, 16);
    }
    int mt, cpw;
    if (!small_shape(M, N, chunks, &mt, &cpw)) return false;
    SmallArgs& sa = plan->args;
    sa = SmallArgs{};
    sa.sg.s0 = segs[0];
    sa.sg.n0 = ceil_div(segs[0].K, 16);
    sa.sg.s1 = nseg == 2 ? segs[1] : segs[0];
    sa.sg.total = chunks;
    sa.M = M; sa.N = N; sa.y = out.y; sa.ldy = out.ldy; sa.bias = out.bias; sa.bias2 = out.bias2;
    sa.epi = out.epi; sa.mul = out.mul; sa.y_pre = out.y_pre; sa.ldy_pre = out.ldy_pre;
    sa.accumulate = out.accumulate;
    sa.aux = out.aux; sa.ld_aux = out.ld_aux; sa.addend = out.addend; sa.ld_addend = out.ld_addend;
    sa.r1_s = out.r1_s; sa.r1_v = out.r1_v;
    plan->mt = mt;
    plan->cpw = cpw;
    plan->gx = ceil_div(N, 16);
    plan->gy = ceil_div(ceil_div(M, 16), mt);
    return true;
}

static int launch_small_plan(const SmallPlan& p, hipStream_t st) {
    const SmallArgs& sa = p.args;
    switch (p.mt * 32 + p.cpw) {
        case 1 * 32 + 2: launch_small<1, 2>(sa, st); break;
        case 1 * 32 + 4: launch_small<1, 4>(sa, st); break;
        case 1 * 32 + 8: launch_small<1, 8>(sa, st); break;
        case 1 * 32 + 16: launch_small<1, 16>(sa, st); break;
        case 1 * 32 + 18: launch_small<1, 18>(sa, st); break;
        case 2 * 32 + 2: launch_small<2, 2>(sa, st); break;
        case 2 * 32 + 4: launch_small<2, 4>(sa, st); break;
        case 2 * 32 + 8: launch_small<2, 8>(sa, st); break;
        case 4 * 32 + 2: launch_small<4, 2>(sa, st); break;
        case 4 * 32 + 4: launch_small<4, 4>(sa, st); break;
        default: return SF_ERR_UNSUPPORTED;
    }
    return launch_status();
}

int launch_small_plan_x(const SmallPlan& p, hipStream_t st) { return launch_small_plan(p, st); }

int launch_small_plan_pw(const SmallPlan& p, const LstmPwBwd& pw, hipStream_t st) {
    const SmallArgs& sa = p.args;
    if (sa.N != pw.H || sa.M != pw.B || sa.epi != EPI_NONE) return SF_ERR_UNSUPPORTED;
    const dim3 grid(ceil_div(sa.N, 16), ceil_div(ceil_div(sa.M, 16), p.mt));
    switch (p.mt * 32 + p.cpw) {
        case 1 * 32 + 2: SF_LAUNCH((gemm_nt_small_pw_kernel<1, 2>), grid, dim3(SMALL_WAVES * 64), 0, st, sa, pw); break;
        case 2 * 32 + 2: SF_LAUNCH((gemm_nt_small_pw_kernel<2, 2>), grid, dim3(SMALL_WAVES * 64), 0, st, sa, pw); break;
        case 4 * 32 + 2: SF_LAUNCH((gemm_nt_small_pw_kernel<4, 2>), grid, dim3(SMALL_WAVES * 64), 0, st, sa, pw); break;
        default: return SF_ERR_UNSUPPORTED;
    }
    return launch_status();
}

int transpose_ld(const float* src, int lds, int R, int C, float* dst, hipStream_t st);     // sf_pointwise.hip

static int nt_big_launch(const NtBigArgs& b, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_big_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const dim3 grid(ceil_div(b.M, NB_T) * ceil_div(b.N, NB_T), b.ksplit > 1 ? b.ksplit : 1);
    SF_LAUNCH(gemm_nt_big_kernel, grid, dim3(512),
              (size_t)2 * 6 * NB_PLANE, st, b);
    return launch_status();
}

int linear_nt(const Seg* segs, int nseg, int M, int N, const LinearOut& out, float* ws,
              size_t ws_floats, hipStream_t st, float** raw_slabs, int* ksplit_out) {
    SF_CHECK_ARG(nseg >= 1 && nseg <= 3 && M > 0 && N > 0);
    NtArgs a{};
    a.nseg = nseg;
    int chunks = 0;
    for (int s = 0; s < nseg; ++s) {
        SF_CHECK_ARG(segs[s].K > 0 && segs[s].K % 4 == 0 && segs[s].lda % 4 == 0 &&
                     segs[s].ldw % 4 == 0);
        a.seg[s] = segs[s];
        chunks += ceil_div(segs[s].K, 16);
    }
    int mt, mblocks, ks;
    SmallPlan sp;
    if (!raw_slabs && linear_small_plan(segs, nseg, M, N, out, &sp)) {
        const int rc = launch_small_plan(sp, st);
        if (ksplit_out) *ksplit_out = 1;
        return rc;
    }
    // (raw_slabs: the caller sums split-K slabs itself -- the LSTM cell kernel; here ONE slab without bias)
    bool big = M >= 512 && N >= 64 && chunks >= 4 && !out.r1_s && g_nt_big &&
               (out.epi == EPI_NONE || out.epi == EPI_TANH) && !(out.accumulate && out.epi != EPI_NONE) &&
               (!raw_slabs || (ws && ws_floats >= (size_t)M * N));
    for (int s = 0; s < nseg; ++s) big = big && segs[s].K >= NB_K;        // (a partial last stage per segment is fine)
    if (big) {
        // many rows: 128 x 128 tiles through LDS on the bf16 matrix cores (gemm_nt_big_kernel)
        NtBigArgs b{};
        for (int s = 0; s < nseg; ++s) b.seg[s] = segs[s];
        b.nseg = nseg; b.M = M; b.N = N; b.y = out.y; b.ldy = out.ldy; b.bias = out.bias; b.bias2 = out.bias2;
        b.addend = out.addend; b.ld_addend = out.ld_addend; b.epi = (int)out.epi; b.accumulate = out.accumulate;
        int kb = 1;
        if (raw_slabs) {
            b.y = ws; b.ldy = N; b.bias = b.bias2 = b.addend = nullptr; b.epi = EPI_NONE; b.accumulate = 0;
            *raw_slabs = ws;
            // The consumer adds the slabs up anyway, so K splits are free of a reduce launch: take them when the tile
            // count wastes a round of the 256 CUs (the beam step's 2 560 x 2 048: 320 tiles = 2 rounds; 3 splits =
            // 960 units = 4 rounds of a third each).  Cost of a split: one more slab written and read (M N floats).
            if (ksplit_out && g_nt_big_ksplit) {
                const int tiles = ceil_div(M, NB_T) * ceil_div(N, NB_T);
                int stages = 0;
                for (int s = 0; s < nseg; ++s) stages += ceil_div(segs[s].K, NB_K);
                double best = (double)ceil_div(tiles, 256);
                for (int c = 2; c <= 3; ++c) {
                    if (ws_floats < (size_t)c * M * N || stages < 16 * c) break;
                    const double t = (double)ceil_div(tiles * c, 256) / c + 0.04 * (c - 1);
                    if (t < best - 1e-9) { best = t; kb = c; }
                }
            }
        }
        b.ksplit = kb;
        if (ksplit_out) *ksplit_out = kb;
        return nt_big_launch(b, st);
    }
    if (out.addend || out.r1_s || out.epi == EPI_TANHBWD) return SF_ERR_UNSUPPORTED;   // small kernel only
    nt_shape(M, N, chunks, &mt, &mblocks, &ks);
    const bool slabs = ks > 1 || raw_slabs;
    if (slabs) {
        if (!ws || ws_floats < (size_t)ks * M * N) return SF_ERR_WORKSPACE;
    }
    a.M = M;
    a.N = N;
    a.chunks_total = chunks;
    a.ksplit = ks;
    a.epi = EPI_NONE;
    if (slabs) {
        a.out = ws;
        a.ldo = N;
        a.bias = nullptr;
        a.bias2 = nullptr;
    } else {
        a.out = out.y;
        a.ldo = out.ldy;
        a.bias = out.bias;
        a.bias2 = out.bias2;
        a.epi = out.epi;            // single split: epilogue fused into the GEMM
        a.accumulate = out.accumulate;
        a.mul = out.mul;
        a.y_pre = out.y_pre;
        a.ldy_pre = out.ldy_pre;
    }
    bool launched = false;
    const NtArgs& k = a;   // raw slabs with ks == 1: slab 0 is written without bias
    bool tiled = !launched && mblocks == 1 && chunks >= 128 && N % 64 == 0 && a.epi == EPI_NONE;
    for (int s = 0; s < nseg; ++s) tiled = tiled && segs[s].K % TBK == 0;
    if (tiled && M <= 128 && !g_nt_force_f32) {
        // fp32 accuracy on the bf16 matrix cores (gemm_nt_split_kernel): 6/16 of the fp32-MFMA time
        dim3 tgrid(N / 64, ks);
        const size_t lds = std::max<size_t>((size_t)2 * 3 * (((mt + 1) / 2) * 32) * SPL_ROWB, (size_t)4 * mt * 64 * 16);
#define SF_SPLIT(MTV)                                                                               \
    case MTV: {                                                                                    \
        static bool attr_set = false;                                                              \
        if (!attr_set) {                                                                           \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_split_kernel<MTV>),    \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);     \
            attr_set = true;                                                                       \
        }                                                                                          \
        SF_LAUNCH(gemm_nt_split_kernel<MTV>, tgrid, dim3(512), lds, st, k);                        \
    } break;
        switch (mt) {
            SF_SPLIT(1) SF_SPLIT(2) SF_SPLIT(3) SF_SPLIT(4) SF_SPLIT(5) SF_SPLIT(6) SF_SPLIT(7) SF_SPLIT(8)
        }
#undef SF_SPLIT
    } else if (tiled) {
        dim3 tgrid(N / 64, ks);
        const size_t lds = (size_t)2 * (mt * 16 + 64) * TLD * sizeof(float);
#define SF_TILED(MTV)                                                                              \
    case MTV: {                                                                                    \
        static bool attr_set = false;   /* > 64 KB of dynamic LDS must be opted into, once */      \
        if (!attr_set) {                                                                           \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_tiled_kernel<MTV>),    \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);     \
            attr_set = true;                                                                       \
        }                                                                                          \
        SF_LAUNCH(gemm_nt_tiled_kernel<MTV>, tgrid, dim3(512), lds, st, k);               \
    } break;
        switch (mt) {
            SF_TILED(1) SF_TILED(2) SF_TILED(3) SF_TILED(4) SF_TILED(5) SF_TILED(6) SF_TILED(7)
            default: {
                static bool attr8 = false;
                if (!attr8) {
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_tiled_kernel<8>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                    attr8 = true;
                }
                SF_LAUNCH(gemm_nt_tiled_kernel<8>, tgrid, dim3(512), lds, st, k);
            }
        }
#undef SF_TILED
    } else if (!launched) {
    dim3 grid(ceil_div(N, 64), ks, mblocks);
    switch (mt) {
        case 1: launch_nt<1>(k, grid, st); break;
        case 2: launch_nt<2>(k, grid, st); break;
        case 3: launch_nt<3>(k, grid, st); break;
        case 4: launch_nt<4>(k, grid, st); break;
        case 5: launch_nt<5>(k, grid, st); break;
        case 6: launch_nt<6>(k, grid, st); break;
        case 7: launch_nt<7>(k, grid, st); break;
        default: launch_nt<8>(k, grid, st); break;
    }
    }
    if (ksplit_out) *ksplit_out = ks;
    if (raw_slabs) {
        *raw_slabs = ws;
        return launch_status();
    }
    RedArgs r{};
    r.M = M;
    r.N = N;
    r.y = out.y;
    r.ldy = out.ldy;
    r.mul = out.mul;
    r.y_pre = out.y_pre;
    r.ldy_pre = out.ldy_pre;
    r.epi = out.epi;
    r.accumulate = out.accumulate;
    if (ks > 1) {
        r.slabs = ws;
        r.ks = ks;
        r.bias = out.bias;
        r.bias2 = out.bias2;
        SF_LAUNCH(reduce_slabs_kernel, dim3(red_grid((size_t)M * N)), dim3(256), 0, st, r);
    }
    return launch_status();
}

int lstm_step_fused(const LstmStepArgs& p, hipStream_t st) {
    if (p.H % 16 || (p.x && (p.I % 4 || p.ldx % 4))) return SF_ERR_UNSUPPORTED;
    const int total = ceil_div(p.H, 16) + (p.x ? ceil_div(p.I, 16) : 0);
    if (p.B > 16 && p.H % 8 == 0) {          // 32 x 8 patches: fewer bytes per block (see the kernel)
        const int cw = ceil_div(total, LSTMW_WAVES);
        dim3 wgrid(p.H / 8, ceil_div(p.B, 32)), wblock(LSTMW_WAVES * 64);
        if (cw <= 2) {
            SF_LAUNCH(lstm_step_wide_kernel<2>, wgrid, wblock, 0, st, p);
            return launch_status();
        }
        if (cw <= 4) {
            SF_LAUNCH(lstm_step_wide_kernel<4>, wgrid, wblock, 0, st, p);
            return launch_status();
        }
    }
    const int c = ceil_div(total, LSTM_KS);
    dim3 grid(p.H / 16, ceil_div(p.B, 16)), block(4 * LSTM_KS * 64);
    if (c <= 8)
        SF_LAUNCH(lstm_step_fused_kernel<8>, grid, block, 0, st, p);
    else if (c <= 16)
        SF_LAUNCH(lstm_step_fused_kernel<16>, grid, block, 0, st, p);
    else
        return SF_ERR_UNSUPPORTED;
    return launch_status();
}

int lstm_bwd_step_fused(const LstmBwdStepArgs& p, hipStream_t st) {
    if (p.H % 16 || (4 * p.H) % 16) return SF_ERR_UNSUPPORTED;
    const int c = ceil_div((4 * p.H) >> 4, BWS_WAVES);
    dim3 grid(p.H / 16, ceil_div(p.B, 16)), block(BWS_WAVES * 64);
    if (c <= 8)
        SF_LAUNCH(lstm_bwd_step_fused_kernel<8>, grid, block, 0, st, p);
    else if (c <= 16)
        SF_LAUNCH(lstm_bwd_step_fused_kernel<16>, grid, block, 0, st, p);
    else
        return SF_ERR_UNSUPPORTED;
    return launch_status();
}

// NN split-K workspace is provided by a per-stream scratch owned by the API layer.
int gemm_nn_ws(const float* A, int lda, const float* W, int ldw, int M, int N, int K, float* y,
               int ldy, int accumulate, float* ws, size_t ws_floats, hipStream_t st) {
    SF_CHECK_ARG(M > 0 && N > 0 && K > 0 && N % 4 == 0 && lda % 4 == 0 && ldw % 4 == 0 &&
                 ldy % 4 == 0);
    const int mtiles = ceil_div(M, 16);
    int mt = mtiles <= 1 ? 1 : (mtiles <= 2 ? 2 : (mtiles <= 4 ? 4 : 7));
    if (mtiles > 7) mt = 7;
    const int mblocks = ceil_div(mtiles, mt);
    const int chunks = ceil_div(K, 16);
    int ks = pick_ksplit(ceil_div(N, 64) * mblocks, chunks);
    if (ks > 1 && (!ws || ws_floats < (size_t)ks * M * N)) ks = 1;   // degrade, still correct
    NnArgs a{A, lda, W, ldw, M, N, K, ks, ks > 1 ? ws : y, ks > 1 ? N : ldy, accumulate};
    dim3 grid(ceil_div(N, 256), ks, mblocks);
    switch (mt) {
        case 1: launch_nn<1>(a, grid, st); break;
        case 2: launch_nn<2>(a, grid, st); break;
        case 4: launch_nn<4>(a, grid, st); break;
        default: launch_nn<7>(a, grid, st); break;
    }
    if (ks > 1) {
        RedArgs r{};
        r.slabs = ws;
        r.ks = ks;
        r.M = M;
        r.N = N;
        r.y = y;
        r.ldy = ldy;
        r.epi = EPI_NONE;
        r.accumulate = accumulate;
        SF_LAUNCH(reduce_slabs_kernel, dim3(red_grid((size_t)M * N)), dim3(256), 0, st, r);
    }
    return launch_status();
}

size_t gemm_nn_ws_floats(int M, int N, int K) {
    const int mtiles = ceil_div(M, 16);
    int mt = mtiles <= 1 ? 1 : (mtiles <= 2 ? 2 : (mtiles <= 4 ? 4 : 7));
    const int mblocks = ceil_div(mtiles, mt);
    const int ks = pick_ksplit(ceil_div(N, 64) * mblocks, ceil_div(K, 16));
    return ks > 1 ? (size_t)ks * M * N : 0;
}

int gemm_nn(const float* A, int lda, const float* W, int ldw, int M, int N, int K, float* y,
            int ldy, int accumulate, hipStream_t st) {
    return gemm_nn_ws(A, lda, W, ldw, M, N, K, y, ldy, accumulate, nullptr, 0, st);
}

// The leading columns of a LARGE gemm_tn output [P,Q] that fill whole rounds of the chip's 1024 SIMDs with 64 x 64 wave
// tiles (dW_ih of the decoder LSTM: 2176 tiles = 2.125 per SIMD, so some SIMDs would run 3); the narrow remainder is a
// product of its own (a row split or the many-row kernel spreads it over the chip).  Q when the output is not cut.
int gemm_tn_main_columns(int M, int P, int Q) {
    const int waves = ceil_div(Q, 64) * ceil_div(P, 64);
    const int pb = ceil_div(P, 64), qb = ceil_div(Q, 256);
    const int rem = waves % 1024;
    if (!(waves > 1024 && rem > 0 && rem <= 384 && rem % (4 * pb) == 0)) return Q;
    const int r = rem / (4 * pb);                       // column blocks of 256 in the remainder
    const int q_main = (qb - r) * 256, q_tail = Q - q_main;
    if (!(r < qb && q_tail > 0)) return Q;
    const int tail_waves = ceil_div(q_tail, 64) * pb;
    const int tail_ms = std::min(16, std::max(1, 1024 / tail_waves));
    if (!(tail_ms > 1 && M / 64 >= tail_ms)) return Q;
    return q_main;
}

// Whether gemm_tn runs this product as an NT product of the transposed operands on the many-row kernel (below)
static bool tn_as_many_row(int M, int P, int Q) {
    return g_nt_big && !g_nt_force_f32 && M >= 1024 && M % 4 == 0 && P >= 64 && Q >= 64 &&
           (size_t)P * Q <= (size_t)1536 * 1024 && !(P % TNS_B == 0 && Q % TNS_B == 0 && M >= g_tn_split_min_rows);
}

int g_tn_group = 1;          // sf_debug_grouped_weight_gradients (0: one gemm_tn per product, the round-5 first form)

// Several weight gradients dW_j[P_j, Q_j] (+)= Y_j^T X_j over the same kind of stacked rows in THREE launches -- all the
// transposes (an operand two products share is transposed once), all the many-row products (gemm_nt_big_group_kernel),
// all the slab sums -- instead of four dependent launches per product.  Products outside the many-row form (or when
// the workspace is short) go through gemm_tn one by one.
int gemm_tn_group(const TnJob* jobs, int n, hipStream_t st, float* ws, size_t ws_floats) {
    int pick[NB_GROUP], np = 0;
    for (int i = 0; i < n; ++i) {
        const TnJob& t = jobs[i];
        SF_CHECK_ARG(t.M > 0 && t.P > 0 && t.Q > 0 && t.Q % 4 == 0 && t.ldy % 4 == 0 && t.ldx % 4 == 0 && t.ldo % 4 == 0);
        if (g_tn_group && ws && np < NB_GROUP && tn_as_many_row(t.M, t.P, t.Q)) pick[np++] = i;
    }
    // the plan: distinct transposes, K splits (every job alike: ~3 rounds of the chip between them), workspace layout
    TrGroup tr{};
    NtBigGroup gg{};
    RedGroup rg{};
    size_t off = 0;
    auto transposed = [&](const float* src, int ld, int M, int C) -> const float* {
        for (int k = 0; k < tr.n; ++k)
            if (tr.p[k].src == src && tr.p[k].lds == ld && tr.p[k].R == M && tr.p[k].C == C) return tr.p[k].dst;
        if (tr.n == TR_GROUP) return nullptr;
        float* dst = ws + off;
        off += ((size_t)C * M + 63) & ~(size_t)63;
        tr.p[tr.n] = TrGroup::P{src, dst, ld, M, C};
        tr.first[tr.n + 1] = tr.first[tr.n] + ceil_div(C, 32) * ceil_div(M, 32);
        ++tr.n;
        return dst;
    };
    bool grouped = np >= 2;
    int tiles_all = 0;
    for (int k = 0; k < np; ++k) tiles_all += ceil_div(jobs[pick[k]].P, NB_T) * ceil_div(jobs[pick[k]].Q, NB_T);
    for (int k = 0; grouped && k < np; ++k) {
        const TnJob& t = jobs[pick[k]];
        const float* yt = transposed(t.Y, t.ldy, t.M, t.P);
        const float* xt = yt ? transposed(t.X, t.ldx, t.M, t.Q) : nullptr;
        if (!xt) { grouped = false; break; }
        const int stages = ceil_div(t.M, NB_K);
        const int ks = std::max(1, std::min(std::min(16, 768 / std::max(1, tiles_all)), stages / 4));
        NtBigArgs& b = gg.job[k];
        b.seg[0] = Seg{yt, t.M, xt, t.M, t.M};
        b.nseg = 1; b.M = t.P; b.N = t.Q; b.epi = EPI_NONE; b.ksplit = ks;
        gg.first[k + 1] = gg.first[k] + ceil_div(t.P, NB_T) * ceil_div(t.Q, NB_T) * ks;
        if (ks > 1) {
            b.y = nullptr; b.ldy = t.Q;                  // (slab base: below, behind every transposed operand)
            rg.j[k] = RedGroup::J{nullptr, t.out, ks, t.P, t.Q, t.ldo, t.accumulate};
        } else {
            b.y = t.out; b.ldy = t.ldo; b.accumulate = t.accumulate;
            rg.j[k] = RedGroup::J{nullptr, nullptr, 0, 0, 0, 0, 0};
        }
    }
    if (grouped) {
        size_t most = 1;
        for (int k = 0; k < np; ++k) {
            if (gg.job[k].ksplit > 1) {
                gg.job[k].y = ws + off;
                rg.j[k].slabs = ws + off;
                off += (((size_t)gg.job[k].ksplit * gg.job[k].M * gg.job[k].N) + 63) & ~(size_t)63;
                most = std::max(most, (size_t)gg.job[k].M * gg.job[k].N);
            }
        }
        if (off > ws_floats) grouped = false;
        if (grouped) {
            gg.n = np;
            static bool attr_set = false;
            if (!attr_set) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_big_group_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                attr_set = true;
            }
            SF_LAUNCH(transpose_group_kernel, dim3(tr.first[tr.n]), dim3(256), 0, st, tr);
            SF_LAUNCH(gemm_nt_big_group_kernel, dim3(gg.first[np]), dim3(512), (size_t)2 * 6 * NB_PLANE, st, gg);
            bool any = false;
            for (int k = 0; k < np; ++k) any = any || gg.job[k].ksplit > 1;
            if (any) SF_LAUNCH(reduce_slabs_group_kernel, dim3(red_grid(most), np), dim3(256), 0, st, rg);
            const int rc = launch_status();
            if (rc != SF_OK) return rc;
        }
    }
    for (int i = 0; i < n; ++i) {
        bool done = false;
        for (int k = 0; grouped && k < np; ++k) done = done || pick[k] == i;
        if (done) continue;
        const TnJob& t = jobs[i];
        const int rc = gemm_tn(t.Y, t.ldy, t.X, t.ldx, t.M, t.P, t.Q, t.out, t.ldo, t.accumulate, st, ws, ws_floats);
        if (rc != SF_OK) return rc;
    }
    return SF_OK;
}

int gemm_tn(const float* Y, int ldy, const float* X, int ldx, int M, int P, int Q, float* out,
            int ldo, int accumulate, hipStream_t st, float* ws, size_t ws_floats) {
    SF_CHECK_ARG(M > 0 && P > 0 && Q > 0 && Q % 4 == 0 && ldy % 4 == 0 && ldx % 4 == 0 &&
                 ldo % 4 == 0);
    // Round 5: a SMALL weight matrix with a deep reduction (the eight [256..512] x [512..2176] weight gradients of the decoder
    // over 2 000 stacked rows: 25 TFLOP/s in gemm_tn_kernel, 90 us each) as an NT product of the two TRANSPOSED operands on
    // the LDS-tiled many-row kernel: dW[P,Q] += Y^T[P,M] (X^T[Q,M])^T.  Two 32 x 32-tiled transposes into the workspace
    // (~6 us each) + gemm_nt_big_kernel (bf16x6, 3x closer to float64 than the fp32 kernels).
    if (tn_as_many_row(M, P, Q) && ws) {
        const int tiles = ceil_div(P, NB_T) * ceil_div(Q, NB_T), stages = ceil_div(M, NB_K);
        int ks = std::max(1, std::min(std::min(16, 256 / tiles), stages / 4));
        const size_t off_x = ((size_t)P * M + 63) & ~(size_t)63, off_s = off_x + (((size_t)Q * M + 63) & ~(size_t)63);
        if (ws_floats >= off_s + (ks > 1 ? (size_t)ks * P * Q : 0)) {
            float* yt = ws;
            float* xt = ws + off_x;
            int rc = transpose_ld(Y, ldy, M, P, yt, st);
            if (rc != SF_OK) return rc;
            rc = transpose_ld(X, ldx, M, Q, xt, st);
            if (rc != SF_OK) return rc;
            NtBigArgs b{};
            b.seg[0] = Seg{yt, M, xt, M, M};
            b.nseg = 1; b.M = P; b.N = Q; b.epi = EPI_NONE; b.ksplit = ks;
            if (ks > 1) {
                b.y = ws + off_s; b.ldy = Q;
                rc = nt_big_launch(b, st);
                if (rc != SF_OK) return rc;
                RedArgs r{};
                r.slabs = ws + off_s; r.ks = ks; r.M = P; r.N = Q; r.y = out; r.ldy = ldo; r.epi = EPI_NONE;
                r.accumulate = accumulate;
                SF_LAUNCH(reduce_slabs_kernel, dim3(red_grid((size_t)P * Q)), dim3(256), 0, st, r);
                return launch_status();
            }
            b.y = out; b.ldy = ldo; b.accumulate = accumulate;
            return nt_big_launch(b, st);
        }
    }
    // A small weight matrix with a deep reduction (e.g. [256, 2176] over 2000 stacked rows) is a
    // handful of waves each walking all M rows: split the rows over grid.z into slabs and add them
    // up (deterministic order) until the chip is covered.
    if (P % TNS_B == 0 && Q % TNS_B == 0 && M >= g_tn_split_min_rows && !g_nt_force_f32) {
        // bf16x6 split products (gemm_tn_split_kernel) where they are faster: deep reductions (see the kernel's note)
        const int blocks = (P / TNS_B) * (Q / TNS_B);
        int ms = blocks >= 512 ? 1 : std::min(16, std::max(1, std::min(512 / blocks, M / 256)));
        if (ms > 1 && (!ws || ws_floats < (size_t)ms * P * Q)) ms = 1;
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_split_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr_set = true;
        }
        dim3 grid(Q / TNS_B, P / TNS_B, ms);
        if (ms > 1) {
            TnArgs a{Y, ldy, X, ldx, M, P, Q, ws, Q, 0, ms};
            SF_LAUNCH(gemm_tn_split_kernel, grid, dim3(256), TNS_LDS, st, a);
            RedArgs r{};
            r.slabs = ws; r.ks = ms; r.M = P; r.N = Q; r.y = out; r.ldy = ldo; r.epi = EPI_NONE;
            r.accumulate = accumulate;
            SF_LAUNCH(reduce_slabs_kernel, dim3(red_grid((size_t)P * Q)), dim3(256), 0, st, r);
            return launch_status();
        }
        TnArgs a{Y, ldy, X, ldx, M, P, Q, out, ldo, accumulate, 1};
        SF_LAUNCH(gemm_tn_split_kernel, grid, dim3(256), TNS_LDS, st, a);
        return launch_status();
    }
    if (P >= 128 && Q >= 128 && ldy >= 128 && M >= 4096) {   // (measured: pays for the deep encoder reductions)
        // LDS-tiled kernel: 128 x 128 per block; small outputs split the batch rows into slabs
        const int blocks = ceil_div(Q, TNT_B) * ceil_div(P, TNT_B);
        int ms = std::min(16, std::max(1, 768 / blocks));
        ms = std::min(ms, std::max(1, M / 64));
        if (ms > 1 && (!ws || ws_floats < (size_t)ms * P * Q)) ms = 1;
        dim3 grid(ceil_div(Q, TNT_B), ceil_div(P, TNT_B), ms);
        if (ms > 1) {
            TnArgs a{Y, ldy, X, ldx, M, P, Q, ws, Q, 0, ms};
            SF_LAUNCH(gemm_tn_tiled_kernel, grid, dim3(256), 0, st, a);
            RedArgs r{};
            r.slabs = ws; r.ks = ms; r.M = P; r.N = Q; r.y = out; r.ldy = ldo; r.epi = EPI_NONE;
            r.accumulate = accumulate;
            SF_LAUNCH(reduce_slabs_kernel, dim3(red_grid((size_t)P * Q)), dim3(256), 0, st, r);
            return launch_status();
        }
        TnArgs a{Y, ldy, X, ldx, M, P, Q, out, ldo, accumulate, 1};
        SF_LAUNCH(gemm_tn_tiled_kernel, grid, dim3(256), 0, st, a);
        return launch_status();
    }
    const int waves = ceil_div(Q, 64) * ceil_div(P, 64);
    {
        // Wave quantisation: a large output whose 64 x 64 wave tiles are a little more than a whole
        // number of rounds over the chip's 1024 SIMDs (dW_ih of the decoder LSTM: 2176 tiles = 2.125
        // per SIMD, so some SIMDs run 3) is cut into the columns that fill whole rounds and a
        // narrow remainder, which the row split below spreads over the chip on its own.
        const int q_main = gemm_tn_main_columns(M, P, Q), q_tail = Q - q_main;
        if (q_tail > 0) {
            const int tail_ms = std::min(16, std::max(1, 1024 / (ceil_div(q_tail, 64) * ceil_div(P, 64))));
            if (ws && ws_floats >= (size_t)tail_ms * P * q_tail) {
                const int rc = gemm_tn(Y, ldy, X, ldx, M, P, q_main, out, ldo, accumulate, st, ws, ws_floats);
                if (rc != SF_OK) return rc;
                return gemm_tn(Y, ldy, X + q_main, ldx, M, P, q_tail, out + q_main, ldo, accumulate, st, ws,
                               ws_floats);
            }
        }
    }
    int ms = std::min(16, std::max(1, 1024 / waves));
    ms = std::min(ms, std::max(1, M / 64));
    if (ms > 1 && (!ws || ws_floats < (size_t)ms * P * Q)) ms = 1;
    if (ms > 1) {
        TnArgs a{Y, ldy, X, ldx, M, P, Q, ws, Q, 0, ms};
        dim3 grid(ceil_div(Q, 256), ceil_div(P, 64), ms);
        SF_LAUNCH(gemm_tn_kernel, grid, dim3(256), 0, st, a);
        RedArgs r{};
        r.slabs = ws; r.ks = ms; r.M = P; r.N = Q; r.y = out; r.ldy = ldo; r.epi = EPI_NONE;
        r.accumulate = accumulate;
        SF_LAUNCH(reduce_slabs_kernel, dim3(red_grid((size_t)P * Q)), dim3(256), 0, st, r);
        return launch_status();
    }
    TnArgs a{Y, ldy, X, ldx, M, P, Q, out, ldo, accumulate, 1};
    dim3 grid(ceil_div(Q, 256), ceil_div(P, 64));
    SF_LAUNCH(gemm_tn_kernel, grid, dim3(256), 0, st, a);
    return launch_status();
}

int colsum(const float* Y, int ldy, int M, int N, float* out, int accumulate, hipStream_t st,
           float* out2, float* ws, size_t ws_floats) {
    SF_CHECK_ARG(M > 0 && N > 0);
    const int nb = ceil_div(N, 64);
    int ms = std::min(32, std::max(1, 256 / nb));
    ms = std::min(ms, std::max(1, M / 64));
    if (ms > 1 && (!ws || ws_floats < (size_t)ms * N)) ms = 1;
    SF_LAUNCH(colsum_kernel, dim3(nb, ms), dim3(1024), 0, st, Y, ldy, M, N, out, out2,
                       accumulate, ms, ws);
    if (ms > 1)
        SF_LAUNCH(colsum_finish_kernel, dim3(ceil_div(N, 256)), dim3(256), 0, st, ws, ms, N,
                           out, out2, accumulate);
    return launch_status();
}

}  // namespace sf
