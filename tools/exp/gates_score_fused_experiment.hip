// EXPERIMENT (not built into libsf_hip.so): scoring + glue of step t fused with the LSTMCell gate
// product of step t+1 in one launch (score blocks publish u_t write-through, the GEMM blocks walk
// the [attended feature | h] stages first, wait for the row count, then the u_t stages).
// Correct (tests passed), NOT faster on MI355X: 36.1 us per fused launch vs 26.8 + 8.1 us as two
// launches (best of handicap / split sweeps 734K vs 745K agent-steps/s).  The 100 blocks that score
// first lose ~4 GEMM stages each, a stage costs 1.9 us (not the 1.5 us MFMA bound), and an
// agent-scope acquire after the wait (buffer_inv sc1 from 2048 waves) wiped the XCD L2s: 48.8 us.
// Timeline per K-split (tools/fuse_trace.py, us after first block start, no fence):
//   score done 7.5 (max 13.4) | seg0/1 reached 12.9 | wait done 15.9 | loop done 27.8 | end 30.7-35.4
// ---- sf_gemm_tiled.h ----
// The LDS-tiled skinny NT GEMM (the decoder LSTMCell gate product, model.py:393) as a device-side
// body, shared by the stand-alone kernel (sf_gemm.hip) and the fused "scoring of step t + gate
// product of step t+1" launch (sf_attention.hip).
#pragma once
#include "sf_gemm.h"
#include "sf_gemm_small.h"

#include <type_traits>

namespace sf {

// ------------------------------------------------------------------------------------------------
// NT: C[M,N] = sum_s A_s[M,K_s] * W_s[N,K_s]^T        (forward Linear; both operands K-contiguous)
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

// Row stride 72 dwords: conflict-free for the ds_read_b128 lane groups of gfx950 ({0-3,12-15,20-27},
// ...: MI355X_MICROARCH.md, LDS table).  68 gave a 2-way conflict in every group
// (SQ_LDS_BANK_CONFLICT = 35 % of SQ_LDS_IDX_ACTIVE).
constexpr int TBK = 64, TLD = TBK + 8;

// XCD-aware tile map: consecutive workgroup ids go round-robin to the 8 XCDs (each with its own
// L2).  Give every XCD a contiguous range of (split, n-tile) pairs, so that its L2 only ever
// holds ITS k-slices of the activation operand (M x K/8) instead of all of A.
// `ntiles` n-tiles x `ksplit` splits = the GEMM blocks; b = linear id of the block among them.
__device__ __forceinline__ void tiled_tile_map(int b, int ntiles, int ksplit, int* n0, int* split) {
    const int total = ntiles * ksplit;
    int g = b;
    if ((total & 7) == 0) g = (b & 7) * (total >> 3) + (b >> 3);
    *split = g / ntiles;
    *n0 = (g % ntiles) * 64;
}

// Fused mode (FUSED = true): segment 2 of the A operand (the chosen action's feature row u_t,
// follower.py:502) is WRITTEN DURING THIS LAUNCH by the scoring blocks of the same grid.  Every
// split first walks its share of the stages of segments 0/1, then waits until all B rows have been
// published (`arrive` >= `target`; write-through stores on the producer side, sc1 loads here:
// MI355X_MICROARCH.md, inter-workgroup visibility), then walks its share of segment 2.
struct TiledFused {
    unsigned* arrive;      // rows published so far (zero at launch; reset by the last block out)
    unsigned* done;        // GEMM blocks finished (zero at launch; reset by the last block out)
    unsigned* error;       // set to 1 when a bounded wait expired (results invalid, no hang)
    unsigned target;       // rows to wait for
    unsigned nblocks;      // GEMM blocks of the grid
    unsigned long long a_bnd;   // byte i = first stage (of segments 0/1) of split i+1; byte 7 = their count
    unsigned long long u_bnd;   // same for the stages of segment 2
    int fence;                  // experiment switch
    unsigned long long* trace;  // development only: [nblocks][8] wall-clock stamps, or null
};
__device__ __forceinline__ void tiled_stamp(const TiledFused& fz, int slot) {
    if (fz.trace && threadIdx.x == 0) fz.trace[(size_t)blockIdx.x * 8 + slot] = wall_clock64();
}
constexpr unsigned TILED_SPIN_LIMIT = 1u << 20;

// 8 waves per block: waves 0-3 take the first half of every 64-deep stage, waves 4-7 the second
// half (two waves per SIMD hide each other's barrier and LDS latencies); the halves meet in LDS.
// A block stages BK = 64 deep tiles of A (MT*16 rows) and W (64 rows) into LDS with full 256-B row
// segments (16 lanes per row), double buffered through registers, and the waves read their fragments
// with conflict-free ds_read_b128.  Block 512; a block's stages lie inside one K segment each
// (K_s % 64 == 0).  smem: 2 * (MT*16 + 64) * TLD floats.
template <int MT, bool FUSED>
__device__ __forceinline__ void tiled_gemm_body(const NtArgs& a, float* smem, int n0, int split,
                                                const TiledFused& fz) {
    constexpr int AROWS = MT * 16, WROWS = 64;
    constexpr int BUF = (AROWS + WROWS) * TLD;           // floats per stage buffer
    constexpr int APASS = (MT + 1) / 2;                  // 32 rows x 16 float4 per staging pass
    const int tid = threadIdx.x, lane = tid & 63, wave8 = tid >> 6;
    const int wave = wave8 & 3, khalf = wave8 >> 2;
    const int li = lane & 15, kk = lane >> 4;
    const int ldrow = tid >> 4, ldc4 = tid & 15;         // staging: 32 rows x 16 float4 per pass

    // stage range of this split (stages of 64 k over the concatenated segments)
    const int st0n = a.seg[0].K / TBK;
    const int st1n = a.nseg > 1 ? a.seg[1].K / TBK : 0;
    const int st2n = a.nseg > 2 ? a.seg[2].K / TBK : 0;
    const int stages = st0n + st1n + st2n;
    // local stage index i in [0, n_loc) -> global stage: a contiguous range, or (fused) a range of
    // the segment-0/1 stages followed by a range of the segment-2 stages
    int s_lo, n_loc, n_a = 0, u_off = 0;
    if (FUSED) {
        auto bnd = [](unsigned long long v, int i) { return i <= 0 ? 0 : (int)((v >> (8 * (i - 1))) & 0xff); };
        const int a_lo = bnd(fz.a_bnd, split), a_hi = bnd(fz.a_bnd, split + 1);
        const int u_lo = bnd(fz.u_bnd, split), u_hi = bnd(fz.u_bnd, split + 1);
        s_lo = a_lo;
        n_a = a_hi - a_lo;
        n_loc = n_a + (u_hi - u_lo);
        u_off = (st0n + st1n + u_lo) - (a_lo + n_a);     // added to s_lo + i for i >= n_a
    } else {
        s_lo = (int)(((long)split * stages) / a.ksplit);
        n_loc = (int)(((long)(split + 1) * stages) / a.ksplit) - s_lo;
    }
    auto gstage = [&](int i) { return s_lo + i + ((FUSED && i >= n_a) ? u_off : 0); };

    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    struct Regs {
        float4 a[APASS];
        float4 w[2];
    };
    // one stage's operand sources (block-uniform selects, no indexing of the argument struct)
    struct StageSrc {
        const float* A;      // per-thread: row 0 of the stage's A tile at this thread's float4 column
        const float* W;
        int lda, ldw;
    };
    auto stage_src = [&](int i) {
        const int s = gstage(i);
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
    // A rows.  (Fused: the rows of segment 2 were published write-through by other workgroups of
    // this launch; the reader's side of the hand-off is the agent-scope acquire in need() below,
    // after which plain loads are fresh -- MI355X_MICROARCH.md, "Consumer, always: ONE relaxed poll
    // -> ONE agent acquire -> plain loads".  sc1 loads for the whole A operand instead made every
    // stage slower: the activations are served from this XCD's L2 otherwise.)
    auto lda4 = [&](const StageSrc& ss, int row) -> float4 { return ld4(ss.A + (size_t)row * ss.lda); };
    auto gload = [&](Regs& r, int i) {
        const StageSrc ss = stage_src(i);
#pragma unroll
        for (int p = 0; p < APASS; ++p) r.a[p] = lda4(ss, min(p * 32 + ldrow, a.M - 1));
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int row = min(n0 + p * 32 + ldrow, a.N - 1);
            r.w[p] = ld4(ss.W + (size_t)row * ss.ldw);
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
    auto gpiece = [&](Regs& r, const StageSrc& ss, int piece) {
        if (piece < APASS) {
            r.a[piece < APASS ? piece : 0] = lda4(ss, min(piece * 32 + ldrow, a.M - 1));
        } else if (piece < APASS + 2) {
            const int p = piece - APASS;
            const int row = min(n0 + p * 32 + ldrow, a.N - 1);
            r.w[p & 1] = ld4(ss.W + (size_t)row * ss.ldw);
        }
    };
    // PF: refill `nx` with the staging set of local stage i_next while computing (one load between
    // groups of MFMAs).  A compile-time switch and a reference, not a nullable pointer: a pointer
    // select keeps the register sets in scratch memory.
    auto compute = [&](int buf, Regs& nx, int i_next, auto pf) {
        constexpr bool PF = decltype(pf)::value;
        const float* As = smem + buf * BUF;
        const float* Ws = As + AROWS * TLD + (wave * 16 + li) * TLD;
        StageSrc ss{};
        if (PF) ss = stage_src(i_next);
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
                if (PF) gpiece(nx, ss, cc * 4 + j);              // 8 slots >= APASS + 2 pieces
            }
        }
    };
    // fused: the first time a stage of segment 2 is about to be fetched, wait for its rows
    bool waited = !FUSED;
    auto need = [&](int i) {
        if (FUSED && !waited && i >= n_a) {
            tiled_stamp(fz, 2);
            unsigned spins = 0;
#pragma nounroll
            while (__hip_atomic_load(fz.arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < fz.target) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > TILED_SPIN_LIMIT) {
                    __hip_atomic_store(fz.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
            if (fz.fence) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // this CU's L1 (buffer_inv sc1)
            tiled_stamp(fz, 3);
            waited = true;
        }
    };

    // Two register sets alternate over two LDS buffers; the refill of a set (its six loads) is
    // interleaved with the MFMAs of the stage that follows its store.  Look-ahead: rb is loaded
    // during stage s and stored after stage s+1, ra during stage s+1 and stored after stage s+2.
    // Prefetch indices are clamped, not predicated.
    if (n_loc > 0) {
        const int last = n_loc - 1;
        Regs ra, rb;
        need(0);
        gload(ra, 0);
        lstore(ra, 0);
        need(min(1, last));
        gload(ra, min(1, last));
        need(min(2, last));
        gload(rb, min(2, last));
        __syncthreads();
        // stage 0 (its look-ahead set rb is already in flight)
        compute(0, rb, 0, std::false_type{});
        lstore(ra, 1);
        __syncthreads();
        for (int s = 1; s < n_loc; s += 2) {
            need(min(s + 2, last));
            compute(1, ra, min(s + 2, last), std::true_type{});    // stage s; ra <- s+2
            lstore(rb, 0);
            __syncthreads();
            if (s + 1 >= n_loc) break;
            need(min(s + 3, last));
            compute(0, rb, min(s + 3, last), std::true_type{});    // stage s+1; rb <- s+3
            lstore(ra, 1);
            __syncthreads();
        }
    }
    __syncthreads();
    if (FUSED) tiled_stamp(fz, 4);

    // the two K halves meet in LDS (the stage buffers are free after the loop's last barrier)
    f32x4* red = reinterpret_cast<f32x4*>(smem);
    if (khalf == 1) {
#pragma unroll
        for (int t = 0; t < MT; ++t) red[(wave * MT + t) * 64 + lane] = acc[t];
    }
    __syncthreads();
    if (FUSED && tid == 0) {
        // last GEMM block out re-arms the two counters for the next launch
        const unsigned d = __hip_atomic_fetch_add(fz.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (d == fz.nblocks - 1) {
            __hip_atomic_store(fz.arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(fz.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
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

}  // namespace sf
// ---- kernel + host side (from sf_attention.hip) ----
// Scoring of step t fused with the LSTMCell gate product of step t+1 (model.py:393-396 across the
// step boundary of follower.py:472-505).  The gate product [B, 2F+H] x [2F+H, 4H] is the longest
// stage of a decode step, and 55 % of its reduction (the attended feature and h_t) is known before
// the action of step t has been chosen; only the last 2176 columns (u_t, the chosen candidate's
// row) wait for the scoring.  One grid of 8-wave blocks: every block owns one (n-tile, K-split) of
// the gate product; the blocks of the first `score_splits` splits first score one sample each
// (2 candidates per wave, rows in registers), run the glue, publish u_t WRITE-THROUGH and count the
// row in; then all blocks walk their segment-0/1 stages, wait for the row count (normally long
// reached) and finish with their segment-2 stages.  One dependent stage less per step and the
// scoring hides behind the product.
// =================================================================================================
__device__ __forceinline__ void score_glue_body8(const ScoreArgs& a, const FGlue& g, int b, float* sm) {
    float* s_logit = sm;                                   // [64]
    int* s_at = reinterpret_cast<int*>(sm + 64);
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;      // 8 waves
    const int A = a.src.A;
    const int n4 = (a.src.IMG + a.src.LOC) >> 2;
    const CandRow row0 = cand_row(a.src, b, wave), row1 = cand_row(a.src, b, wave + 8);
    const FGlueIn gin = follower_glue_load(g, b);
    const float4* rv = reinterpret_cast<const float4*>(a.r + (size_t)b * a.ldr);
    float4 x0[SC_CPL], x1[SC_CPL], q[SC_CPL];
#pragma unroll
    for (int i = 0; i < SC_CPL; ++i) x0[i] = x1[i] = q[i] = f4zero();
    const bool have0 = wave < A && !row0.zero, have1 = wave + 8 < A && !row1.zero;   // wave-uniform
    if (have1) {                                           // (rare: more than 8 candidates)
#pragma unroll
        for (int i = 0; i < SC_CPL; ++i) {
            const int c = lane + 64 * i;
            x1[i] = cand_load(row1, c, c < n4, n4);
        }
    }
    if (have0 || have1) {
#pragma unroll
        for (int i = 0; i < SC_CPL; ++i) {
            const int c = lane + 64 * i;
            x0[i] = cand_load(row0, c, c < n4 && have0, n4);
            q[i] = rv[min(c, n4 - 1)];
        }
    }
    float d0 = 0.f, d1 = 0.f;
#pragma unroll
    for (int i = 0; i < SC_CPL; ++i) {
        d0 += dot4(x0[i], q[i]);                           // x is zero beyond n4 / for absent rows
        d1 += dot4(x1[i], q[i]);
    }
    const float cst = score_const(a, b, lane);
    d0 = wave_sum(d0);
    d1 = wave_sum(d1);
    if (lane == 0) {
        if (wave < A) s_logit[wave] = d0 + cst;
        if (wave + 8 < A) s_logit[wave + 8] = d1 + cst;
    }
    __syncthreads();
    if (wave == 0) {
        const int at = follower_glue_row(g, b, lane < A ? s_logit[lane] : 0.f, gin);
        if (lane == 0) *s_at = at;
    }
    __syncthreads();
    const int at = *s_at;
    if (g.u_next && (at & 7) == wave) {
        // write-through (sc1) stores: the gate-product blocks of this launch read the row back
        float* dst = g.u_next + (size_t)b * g.ld_u;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(dst, 0, n4 * 16, 0x00020000);
        const Dropout& ud = g.u_drop;
        const uint32_t rk = dropout_row_key(ud.seed, ud.stream, (uint32_t)(ud.row0 + b));
#pragma unroll
        for (int i = 0; i < SC_CPL; ++i) {
            const int c = lane + 64 * i;
            float4 v = at < 8 ? x0[i] : x1[i];
            if (ud.on()) {
                const uint32_t col = (uint32_t)(4 * c);
                v.x = dropout_keep(rk, col + 0, ud.thresh) ? v.x * ud.scale : 0.f;
                v.y = dropout_keep(rk, col + 1, ud.thresh) ? v.y * ud.scale : 0.f;
                v.z = dropout_keep(rk, col + 2, ud.thresh) ? v.z * ud.scale : 0.f;
                v.w = dropout_keep(rk, col + 3, ud.thresh) ? v.w * ud.scale : 0.f;
            }
            const v4u pk{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
            if (c < n4) __builtin_amdgcn_raw_buffer_store_b128(pk, rs, c * 16, 0, 16);
        }
    }
}

template <int MT>
__global__ __launch_bounds__(512) void gates_score_kernel(NtArgs a, TiledFused fz, ScoreArgs sa,
                                                          FGlue g, int score_splits) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int ntiles = a.N / 64;
    int n0, split;
    tiled_tile_map(blockIdx.x, ntiles, a.ksplit, &n0, &split);
    tiled_stamp(fz, 0);
    if (fz.trace && threadIdx.x == 0) fz.trace[(size_t)blockIdx.x * 8 + 6] = (unsigned long long)split;
    if (split < score_splits) {                            // block-uniform
        unsigned mine = 0;
        for (int b = (n0 >> 6) * score_splits + split; b < g.B; b += ntiles * score_splits) {
            score_glue_body8(sa, g, b, smem);
            ++mine;
        }
        if (mine) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // EVERY storing wave drains
            __syncthreads();
            if (threadIdx.x == 0)
                __hip_atomic_fetch_add(fz.arrive, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    tiled_stamp(fz, 1);
    tiled_gemm_body<MT, true>(a, smem, n0, split, fz);
    tiled_stamp(fz, 5);
}

}  // namespace
// development aid: per-block wall-clock stamps of the fused launch (see tools/fuse_trace.py)
static unsigned long long* g_fuse_trace = nullptr;
extern "C" void sf_debug_fuse_trace(unsigned long long* buf) { g_fuse_trace = buf; }

// Scoring + glue of one step and the gate product of the next in ONE launch (see gates_score_kernel).
// segs: 0/1 = the K segments known at launch, 2 = the rows the glue writes (A = g.u_next, lda = g.ld_u).
// sync: 3 dwords, zero before the first launch (re-armed by every launch).  Slabs [ks][M][N] -> ws.
// SF_ERR_UNSUPPORTED = shape not covered: the caller launches the two stages one after the other.
int gates_score_fused(const Seg* segs, int M, int N, float* ws, size_t ws_floats, int* ks_out,
                      const CandSrc& src, int B, int D, const float* r, const float* wt,
                      const float* b_a, const float* b_out, const FGlue& g, unsigned* sync,
                      hipStream_t st) {
    const int F = src.IMG + src.LOC;
    if (src.A > SC_NW || src.A < 1 || F > SC_CPL * 256 || (F & 3) ||
        (!src.dense && ((src.IMG & 3) || (src.LOC & 15))))
        return SF_ERR_UNSUPPORTED;
    constexpr int KS = 8;
    const int mt = (M + 15) / 16;
    if (mt > 8 || N % 64 || !sync || !g.u_next || segs[2].A != g.u_next || segs[2].lda != g.ld_u ||
        g.B != B || B > M)
        return SF_ERR_UNSUPPORTED;
    int chunks = 0;
    for (int s = 0; s < 3; ++s) {
        if (segs[s].K <= 0 || segs[s].K % TBK || segs[s].lda % 4 || segs[s].ldw % 4) return SF_ERR_UNSUPPORTED;
        chunks += segs[s].K / 16;
    }
    const int na = (segs[0].K + segs[1].K) / TBK, nu = segs[2].K / TBK;
    if (na > 255 || nu > 255 || na < KS || nu < KS) return SF_ERR_UNSUPPORTED;
    if (!ws || ws_floats < (size_t)KS * M * N) return SF_ERR_WORKSPACE;
    // the scoring splits start their GEMM stages late by about `handicap` stages: give them fewer
    // of the segment-0/1 stages (the segment-2 stages are shared equally: they start together)
    static const int P = [] { const char* e = getenv("SF_FUSE_SCORE_SPLITS"); return e ? atoi(e) : 4; }();
    static const float handicap = [] { const char* e = getenv("SF_FUSE_HANDICAP"); return e ? (float)atof(e) : 4.f; }();
    const int p = std::min(std::max(P, 1), KS);
    const float x = std::max(0.f, ((float)na - (float)(KS - p) * handicap) / (float)KS);
    const float y = p < KS ? ((float)na - (float)p * x) / (float)(KS - p) : 0.f;
    TiledFused fz{};
    float cum = 0.f;
    for (int i = 0; i < KS; ++i) {
        cum += i < p ? x : y;
        int ab = i == KS - 1 ? na : std::min(na, (int)std::lround(cum));
        int ub = (int)(((long)(i + 1) * nu) / KS);
        fz.a_bnd |= (unsigned long long)ab << (8 * i);
        fz.u_bnd |= (unsigned long long)ub << (8 * i);
    }
    fz.arrive = sync; fz.done = sync + 1; fz.error = sync + 2;
    fz.trace = g_fuse_trace;
    fz.fence = getenv("SF_FUSE_FENCE") ? atoi(getenv("SF_FUSE_FENCE")) : 0;
    fz.target = (unsigned)B;
    fz.nblocks = (unsigned)(N / 64 * KS);
    NtArgs a{};
    a.nseg = 3;
    for (int s = 0; s < 3; ++s) a.seg[s] = segs[s];
    a.M = M; a.N = N; a.chunks_total = chunks; a.ksplit = KS; a.out = ws; a.ldo = N; a.epi = EPI_NONE;
    if (getenv("SF_FUSE_DBG_A")) a.seg[2].A = a.seg[0].A;      // timing experiment: wrong results
    if (getenv("SF_FUSE_DBG_W")) a.seg[2].W = a.seg[0].W;
    ScoreArgs sa{src, F, nullptr, r, wt, b_a, b_out, D, g.logit, nullptr, nullptr};
    const dim3 grid(N / 64 * KS), block(512);
    const size_t lds = (size_t)2 * (mt * 16 + 64) * TLD * sizeof(float);
#define SF_GS(MTV)                                                                                 \
    case MTV: {                                                                                    \
        static bool attr_set = false;   /* > 64 KB of dynamic LDS must be opted into, once */      \
        if (!attr_set) {                                                                           \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gates_score_kernel<MTV>),      \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);     \
            attr_set = true;                                                                       \
        }                                                                                          \
        hipLaunchKernelGGL(gates_score_kernel<MTV>, grid, block, lds, st, a, fz, sa, g, p);        \
    } break;
    switch (mt) {
        SF_GS(1) SF_GS(2) SF_GS(3) SF_GS(4) SF_GS(5) SF_GS(6) SF_GS(7) SF_GS(8)
        default: return SF_ERR_UNSUPPORTED;
    }
#undef SF_GS
    if (ks_out) *ks_out = KS;
    return launch_status();
}

// ---- paired launches (host side).  SF_ERR_UNSUPPORTED = "not pairable": the caller launches the
