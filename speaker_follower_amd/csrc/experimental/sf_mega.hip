// The follower's decode loop (follower.py:446-531 over model.py:377-399) as ONE persistent launch for
// inference rollouts: S decode steps without a kernel boundary, 256 workgroups x 256 threads (one per CU,
// one wave per SIMD: the phases below keep whole row sets in registers and need the 512-register budget).
//
// Workgroup b: XCD x = b % 8, slot c = b / 8.  Three kinds of work share the launch:
//
//   gate product  [B, 2F+H] x [4H, 2F+H]^T of the LSTMCell (model.py:393): n-tile = hidden units [16c, +16) x 4
//                 gates per slot; the K range is cut into 76 stages of 64.  The u stages (the tail of the
//                 critical path) are split s % 8 over ALL XCDs and keep their weight blocks in registers for
//                 the whole episode (only operand rows stream); the feature / h stages are split over the
//                 visual XCDs 4-7, which are idle once their attention is done, and stream their weights.
//                 The operand row [u | attended feature | h] of step t lives in the exchange buffer
//                 XIN[t % 3]; partial [16 x 64] tiles go to the SLAB region of the workgroup (XCD = m-tile,
//                 slot = n-tile) that owns the cell of those 16 rows x 16 units, which sums its 8 partials,
//                 updates the cell and publishes h.  The three all-to-all edges (h, u, feature) carry an
//                 arrival counter: consumers request data only once all of it has been published.
//   chain groups  rows [32 g, +32), g = x & 3.  XCDs 0-3 run the text / scoring chain of group g, XCDs 4-7
//                 the visual chain of the same rows; workgroup c owns sample 32 g + c and tile lane c:
//        text:    t_text = W_in h1 (tile c)  ->  text attention of the sample  ->  h~ = tanh(W_out [wc ; h1])
//                 (tile c)  ->  [r | const] = M_a h~ + c_a (tiles c, c+32, .., sf_decoder_fold)  ->
//                 candidate scores, masking, CE term, action, u of the next step (the sample)
//        visual:  q' = M_v h1 + c_v (tiles c, c+32, ..)  ->  visual attention of the sample over the NEXT
//                 step's panorama  ->  attended feature of the next step
//   Every hand-off is data tagged with a sentinel (0xFFFFFFFF, never a finite float): producers publish
//   write-through (sc1), consumers re-read until no sentinel is left, and a producer resets the slot
//   two versions ahead (three slots per buffer; every step is an all-to-all through the gate product, so
//   all readers of a slot are done before it is reset).  Waits are bounded (0.25 s); a timed-out
//   workgroup poisons its outputs with NaN instead of hanging.
//
// Addressing discipline: every bulk load is a buffer load whose address is (one per-lane VGPR offset) +
// (a wave-uniform SGPR offset) + (an immediate).  Per-load 64-bit pointers would be hoisted out of the step
// loop by the compiler (hundreds of them), spilled to scratch and re-read in front of every load.
// No load sits inside a branch (the compiler ends such a block with a full vmcnt(0) wait).
//
// Only the inference forward exists in this form (no tapes for a backward, no dropout).  Status: correct
// (tests/test_gpu_mega.py) and ~20 % SLOWER than the per-stage path -- opt-in; DESIGN.md section 8 has the
// measured timeline and the reasons (an in-kernel hop costs 3.5-4 us, a CU pulls ~35 GB/s of L2 misses).
#include "sf_mega.h"
#include "sf_gemm_small.h"
#include "sf_rows.h"

namespace sf {
namespace {

typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

constexpr int MG_SLOTS = 32, MG_XCD = 8;
constexpr int MG_H = 512, MG_F = 2176, MG_IMG = 2048, MG_LOC = 128, MG_V = 36, MG_AMAX = 16, MG_LMAX = 80;
constexpr int MG_K = 2 * MG_F + MG_H;                 // 4864 = [u | feat | h]
constexpr int MG_OFF_F = MG_F, MG_OFF_H = 2 * MG_F;
constexpr int MG_F4 = MG_F / 4;                       // 544 float4 per feature row
constexpr int MG_RLD = 2192;                          // row stride of [r | const | pad]
constexpr int MG_BK = 64, MG_LD = MG_BK + 8;          // stage depth, LDS row stride (conflict-free b128 reads)
constexpr int MG_NU = MG_F / MG_BK, MG_NH = MG_H / MG_BK;      // 34 stages per input half, 8 of h
constexpr unsigned MG_SENT = 0xFFFFFFFFu;
constexpr long long MG_TIMEOUT = 25000000LL;          // 0.25 s of the 100 MHz wall clock
constexpr int MG_SC1 = 16;
constexpr int MG_TILES_R = (MG_F + 4 + 15) / 16;      // 137 tiles of [r | const]
constexpr int MG_TILES_Q = MG_F / 16;                 // 136
constexpr int MG_GM = 2;                              // MFMA m-tiles of a chain group (32 rows)

// exchange workspace (dword offsets)
constexpr unsigned MG_XIN = 0, MG_XIN_N = 3u * 128 * MG_K;
constexpr unsigned MG_TT = MG_XIN + MG_XIN_N, MG_HN = 3u * 128 * MG_H;
constexpr unsigned MG_WC = MG_TT + MG_HN;
constexpr unsigned MG_HT = MG_WC + MG_HN;
constexpr unsigned MG_Q = MG_HT + MG_HN, MG_Q_N = 3u * 128 * MG_F;
constexpr unsigned MG_R = MG_Q + MG_Q_N, MG_R_N = 3u * 128 * MG_RLD;
constexpr unsigned MG_SLAB = MG_R + MG_R_N, MG_SLAB_N = 2u * MG_XCD * MG_SLOTS * 8 * 1024;
constexpr unsigned MG_CNT = MG_SLAB + MG_SLAB_N, MG_CNT_N = 64;   // arrival counters (monotonic within a launch)
constexpr unsigned MG_CNT_H = 0, MG_CNT_U = 16, MG_CNT_F = 32;    // (one 64-byte line each)
constexpr unsigned MG_TOTAL = MG_CNT + MG_CNT_N;

struct MegaArgs {
    MegaHost h;
    unsigned* lock;                                     // persist_lock_addr()
    unsigned long long* trace;                          // sf_debug_trace: [blocks][32] (tick sums | stamps), or null
};

__global__ __launch_bounds__(256) void mega_prologue_kernel(unsigned* x, size_t n, unsigned* lock) {
    const size_t stride = (size_t)gridDim.x * blockDim.x, i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    v4u* x4 = reinterpret_cast<v4u*>(x);
    for (size_t i = i0; i < n / 4; i += stride) x4[i] = v4u{MG_SENT, MG_SENT, MG_SENT, MG_SENT};
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const long long t0 = wall_clock64();
        while (atomicCAS(lock, 0u, 1u) != 0u) {
            __builtin_amdgcn_s_sleep(32);
            if (wall_clock64() - t0 > 8 * MG_TIMEOUT) break;
        }
    }
}
// operand row of step 0 into XIN[0]: u = 0 (model.py:368), feat of step 0 (formed by the per-stage head),
// h = h_init; rows >= B are zero
__global__ __launch_bounds__(256) void mega_seed_kernel(unsigned* xin, const float* h_init, const float* feat0,
                                                        int ld_feat, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (int)MG_CNT_N) xin[MG_CNT - MG_XIN + i] = 0u;          // arrival counters (xin = base of the workspace)
    if (i >= 128 * (MG_K / 4)) return;
    const int row = i / (MG_K / 4), c4 = i - row * (MG_K / 4);
    const int col = 4 * c4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < B) {
        if (col >= MG_OFF_H) v = ld4(h_init + (size_t)row * MG_H + (col - MG_OFF_H));
        else if (col >= MG_OFF_F) v = ld4(feat0 + (size_t)row * ld_feat + (col - MG_OFF_F));
    }
    reinterpret_cast<float4*>(xin)[(size_t)row * (MG_K / 4) + c4] = v;
}

__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ rsrc_t make_rs(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ bool has_sent(const v4u& v) {
    return v.x == MG_SENT || v.y == MG_SENT || v.z == MG_SENT || v.w == MG_SENT;
}
__device__ __forceinline__ float4 as_f4(const v4u& v) {
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ v4u as_u4(const float4& v) {
    return v4u{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
}
__device__ __forceinline__ float qnan() { return __uint_as_float(0x7FC00000u); }
// all offsets below are BYTES: voff per lane (VGPR), soff wave-uniform (SGPR)
__device__ __forceinline__ v4u xld(rsrc_t rs, unsigned voff, unsigned soff) {               // exchange data (sc1)
    return __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, MG_SC1);
}
__device__ __forceinline__ void xst(rsrc_t rs, unsigned voff, unsigned soff, const v4u& v) {
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff, soff, MG_SC1);
}
__device__ __forceinline__ void xrst(rsrc_t rs, unsigned voff, unsigned soff) {
    xst(rs, voff, soff, v4u{MG_SENT, MG_SENT, MG_SENT, MG_SENT});
}
__device__ __forceinline__ float4 bld(rsrc_t rs, unsigned voff, unsigned soff) {             // weights / rows
    return as_f4(__builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
}
// re-read a published piece until it is complete (bounded)
__device__ __forceinline__ void settle(rsrc_t rs, unsigned voff, unsigned soff, v4u& v, bool& dead) {
    if (!has_sent(v) || dead) return;
    const long long t0 = wall_clock64();
    while (has_sent(v)) {
        asm volatile("" ::: "memory");
        v = xld(rs, voff, soff);
        if (wall_clock64() - t0 > MG_TIMEOUT) { dead = true; break; }
    }
}

// Before a workgroup reads exchanged data it parks: ONE lane polls one representative piece (with a short
// sleep between reads) and the rest wait at the barrier.  256 lanes per CU spinning on write-through lines
// load the fabric enough to slow the producers they are waiting for.
__device__ __forceinline__ void park(rsrc_t rs, unsigned voff, unsigned soff, bool& dead) {
    if (threadIdx.x == 0 && !dead) {
        const long long t0 = wall_clock64();
        v4u v = xld(rs, voff, soff);
        while (has_sent(v)) {
            __builtin_amdgcn_s_sleep(4);
            asm volatile("" ::: "memory");
            v = xld(rs, voff, soff);
            if (wall_clock64() - t0 > MG_TIMEOUT) break;       // the readers' own bounded waits report it
        }
    }
    __syncthreads();
}

// Arrival counters for the all-to-all hand-offs (h, u, attended feature of a step): every producer workgroup
// drains its write-through stores, then ONE lane adds 1; a consumer parks one lane on the counter and only
// then requests the data, so its loads find everything in place (one round trip) instead of sentinels.
__device__ __forceinline__ void arrive(unsigned* cnt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's publishes are acknowledged
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void wait_count(const unsigned* cnt, unsigned target, bool& dead) {
    if (threadIdx.x == 0 && !dead) {
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(4);
            if (wall_clock64() - t0 > MG_TIMEOUT) break;       // the readers' own bounded waits report it
        }
    }
    __syncthreads();
}

// One chain product of a chain group: Y[32 rows, tiles] = A[32, K] W[tile rows, K]^T (+ bias, tanh), the
// operand A read from exchange buffers (one or two K segments of 512), the result published tile by tile.
// The 4 waves split K; wave w keeps its K chunk of A (both m-tiles) in registers and walks the tiles, the
// partials meet in LDS.  The product is formed TRANSPOSED (W as the MFMA row operand), so a lane ends up
// with 4 consecutive output columns of one batch row: one b128 store.  Weight rows beyond w_rows read as
// zero (buffer bounds), an m-tile beyond `m_live` re-reads m-tile 0 and is not stored.
struct TileJob {
    unsigned a0, a1;            // BYTE offset of row 0 of the group in segment 0 / 1 (exchange workspace)
    int lda0, lda1;             // row strides in floats
    const float* w; int ldw, w_rows;
    const float* bias;
    int tile0, tile_stride, ntiles;
    unsigned out, rst; int ldo;  // publish / reset slot: BYTE offset of row 0 of the group, row stride in floats
    float* dbg; int ld_dbg;      // optional plain copy: pointer to row 0 of the group
    bool tanh_epi;
    int m_live;                  // m-tiles of the group that exist (rows < 16 MT)
};
template <int NI, int TC, int MAXT>
__device__ __forceinline__ void tile_gemm(rsrc_t rs, const TileJob& j, float* smem, int wv, int rows_valid, bool& dead) {
    const int lane = threadIdx.x & 63;
    const int li = lane & 15, kk = lane >> 4;
    const bool second = NI == 16 && wv >= 2;             // NI = 8: K = 512, NI = 16: two segments of 512
    const int lda = second ? j.lda1 : j.lda0;
    const unsigned va = (unsigned)(li * lda + 4 * kk) * 4u;                         // per lane
    const unsigned sa = (second ? j.a1 : j.a0) + (unsigned)(NI == 16 ? (wv & 1) * 256 : wv * 128) * 4u;   // uniform
    const unsigned vw = (unsigned)(li * j.ldw + 4 * kk) * 4u;
    const unsigned sw0 = (unsigned)(NI == 16 ? wv * 256 : wv * 128) * 4u;
    const rsrc_t rw = make_rs(j.w, (unsigned)j.w_rows * (unsigned)j.ldw * 4u);
    f32x4* red = reinterpret_cast<f32x4*>(smem);
    constexpr int NCH = (MAXT + TC - 1) / TC;
    float4 wf[2][TC][NI];
    auto wload = [&](int ch, float4 (&dst)[TC][NI]) {
#pragma unroll
        for (int tc = 0; tc < TC; ++tc) {
            const int tile = j.tile0 + min(ch * TC + tc, j.ntiles - 1) * j.tile_stride;
            const unsigned sw = sw0 + (unsigned)(16 * tile * j.ldw) * 4u;
#pragma unroll
            for (int i = 0; i < NI; ++i) dst[tc][i] = bld(rw, vw + 64u * i, sw);
        }
    };
    wload(0, wf[0]);
    v4u a[MG_GM][NI];
#pragma unroll
    for (int mm = 0; mm < MG_GM; ++mm) {
        const unsigned sam = sa + (unsigned)((mm < j.m_live ? mm : 0) * 16 * lda) * 4u;
#pragma unroll
        for (int i = 0; i < NI; ++i) a[mm][i] = xld(rs, va + 64u * i, sam);
    }
    park(rs, 0u, j.a0, dead);                            // (the loads above are in flight: a ready operand costs one trip)
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        if (ch * TC < j.ntiles || ch == 0) {
            if (ch + 1 < NCH) wload(ch + 1, wf[(ch + 1) & 1]);
            if (ch == 0) {
#pragma unroll
                for (int mm = 0; mm < MG_GM; ++mm) {
                    const unsigned sam = sa + (unsigned)((mm < j.m_live ? mm : 0) * 16 * lda) * 4u;
#pragma unroll
                    for (int i = 0; i < NI; ++i) settle(rs, va + 64u * i, sam, a[mm][i], dead);
                }
            }
#pragma unroll
            for (int tc = 0; tc < TC; ++tc) {
                if (ch * TC + tc < j.ntiles) {
                    f32x4 acc[MG_GM];
#pragma unroll
                    for (int mm = 0; mm < MG_GM; ++mm) acc[mm] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int i = 0; i < NI; ++i) {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
#pragma unroll
                            for (int mm = 0; mm < MG_GM; ++mm)
                                acc[mm] = mfma16(comp(wf[ch & 1][tc][i], q), comp(as_f4(a[mm][i]), q), acc[mm]);
                    }
#pragma unroll
                    for (int mm = 0; mm < MG_GM; ++mm)
                        red[((wv * MAXT + ch * TC + tc) * MG_GM + mm) * 64 + lane] = acc[mm];
                }
            }
        }
    }
    __syncthreads();
    for (int tj = wv; tj < j.ntiles * MG_GM; tj += 4) {
        const int ti = tj / MG_GM, mm = tj - ti * MG_GM;
        f32x4 s = red[(ti * MG_GM + mm) * 64 + lane];
#pragma unroll
        for (int ww = 1; ww < 4; ++ww) s += red[((ww * MAXT + ti) * MG_GM + mm) * 64 + lane];
        const int tile = j.tile0 + ti * j.tile_stride;
        const int col = 16 * tile + 4 * kk;
        const int row = 16 * mm + li;
        if (j.bias) {
#pragma unroll
            for (int q = 0; q < 4; ++q) s[q] += j.bias[min(col + q, j.w_rows - 1)];
        }
        if (j.tanh_epi) {
#pragma unroll
            for (int q = 0; q < 4; ++q) s[q] = tanhf(s[q]);
        }
        if (dead) s = f32x4{qnan(), qnan(), qnan(), qnan()};
        if (mm < j.m_live) {
            const unsigned vo = (unsigned)(li * j.ldo + 4 * kk) * 4u;
            const unsigned so = (unsigned)(16 * mm * j.ldo + 16 * tile) * 4u;
            xst(rs, vo, j.out + so, v4u{__float_as_uint(s[0]), __float_as_uint(s[1]), __float_as_uint(s[2]), __float_as_uint(s[3])});
            xrst(rs, vo, j.rst + so);
            if (j.dbg && row < rows_valid && col + 3 < j.ld_dbg)
                *reinterpret_cast<float4*>(j.dbg + (size_t)row * j.ld_dbg + col) = make_float4(s[0], s[1], s[2], s[3]);
        }
    }
    __syncthreads();
}

template <int MT>
__global__ __launch_bounds__(256, 1) void mega_kernel(MegaArgs pa) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const MegaHost& p = pa.h;
    constexpr int AROWS = MT * 16, WROWS = 64;
    constexpr int BUF = (AROWS + WROWS) * MG_LD;
    constexpr int APASS = MT;                            // 16 rows x 16 float4 per staging pass
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = uni(tid >> 6);                        // wave index as a wave-uniform (SGPR) value
    const int gate = wv;
    const int li = lane & 15, kk = lane >> 4;
    const int xcd = blockIdx.x & (MG_XCD - 1), slot = blockIdx.x >> 3;
    const int B = p.B, S = p.S, L = p.L, A = p.U.A;
    const int ldrow = tid >> 4, ldc4 = tid & 15;
    const size_t BH = (size_t)B * MG_H;
    const rsrc_t rs = make_rs(p.xchg, MG_TOTAL * 4u);
    const unsigned vl16 = (unsigned)lane * 16u;          // this lane's float4 of a 1 KB row chunk
    bool dead = false;

    constexpr int NGRP = (MT + 1) / 2;                   // chain groups
    // ---- cell ownership: (m-tile = xcd, units [16 slot, +16)); every thread owns one element
    const bool active = xcd < MT;
    const int er = (tid >> 4) & 15, eu = tid & 15;
    const int eb = xcd * 16 + er;
    const bool evalid = active && eb < B;
    const int ebc = min(eb, B - 1);
    const int ej = 16 * slot + eu;
    float bias[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bias[g] = p.b_ih[g * MG_H + ej] + p.b_hh[g * MG_H + ej];
    float c_state = p.c_init[(size_t)ebc * MG_H + ej];

    // ---- chain role: XCDs 0-3 run the text / scoring chain of rows [32 (x & 3), +32), XCDs 4-7 the visual
    // chain of the same rows; workgroup c of the XCD owns sample c of the group and tile lane c of its products
    const int cg = xcd & 3;
    const bool text_role = xcd < 4;
    const bool chain_active = cg < NGRP;
    const int e = slot;
    const int bs = cg * 32 + e;
    const bool sample_ok = chain_active && bs < B;
    const int bsc = min(bs, B - 1);
    const int row0 = cg * 32;                             // first row of the chain group
    const int rows_valid = max(0, min(32, B - row0));
    const int m_live = min(MG_GM, MT - 2 * cg);           // m-tiles of the group inside the gate product's rows
    bool ended_reg = p.ended[bsc] != 0;
    // instruction mask of the sample (step-invariant): lane l holds positions l and l + 64
    const bool mask0 = p.mask[(size_t)bsc * L + min(lane, L - 1)] != 0;
    const bool mask1 = p.mask[(size_t)bsc * L + min(lane + 64, L - 1)] != 0;

    // ---- this workgroup's stages of the gate product.  u stages (s < 34): split s % 8, on every XCD -- they
    // are the tail of the critical path (u of step t is the last thing the text chain produces).  Feature and
    // h stages (s = 34 + j): split j % 4 on the VISUAL XCDs only, which are idle once their attention is done;
    // the text XCDs go from u straight to the next step's chain.
    const int nu = (MG_NU - xcd + 7) >> 3;                // u stages xcd + 8k
    const int jx = xcd - 4;                               // visual XCDs: stages 34 + jx + 4k
    const int nfh = xcd >= 4 ? (MG_NU + MG_NH - jx + 3) >> 2 : 0;
    const int nfeat = xcd >= 4 ? (MG_NU - jx + 3) >> 2 : 0;   // of which the first nfeat are feature stages, the rest h
    // The u-stage weights never change: this workgroup's [64 gate rows x 64 k] block of each stays in registers
    // for the whole episode, in the staging layout (4 float4 per thread and stage).  (Indexed by unrolled loop
    // counters only.)  The feature / h stages stream theirs: those workgroups have slack.
    constexpr int NKMAX = 5;
    float4 wres_u[NKMAX][4];
    {
#pragma unroll
        for (int k = 0; k < NKMAX; ++k) {
            const int k0 = (xcd + 8 * min(k, nu - 1)) * MG_BK;
#pragma unroll
            for (int pp = 0; pp < 4; ++pp)
                wres_u[k][pp] = ld4(p.w_ih + (size_t)(pp * MG_H + 16 * slot + ldrow) * 2 * MG_F + k0 + 4 * ldc4);
        }
    }
    unsigned* const cnt_h = p.xchg + MG_CNT + MG_CNT_H;
    unsigned* const cnt_u = p.xchg + MG_CNT + MG_CNT_U;
    unsigned* const cnt_f = p.xchg + MG_CNT + MG_CNT_F;
    const unsigned n_cells = 32u * MT, n_chain = 32u * NGRP;   // publishers of h / of u and of the feature per step

    // development aid (pa.trace): per phase the ticks summed over the steps, and the stamps of the middle step
    unsigned tk[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tabs[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned tprev = (unsigned)wall_clock64();           // 32-bit ticks (10 ns): 42 s before they wrap
    int t_now = 0;
#define MG_STAMP(k)                                 \
    if (pa.trace) {                                 \
        const unsigned now_ = (unsigned)wall_clock64(); \
        tk[k] += now_ - tprev;                      \
        if (t_now == (S >> 1)) tabs[k] = now_;      \
        tprev = now_;                               \
    }

    // per-lane parts of the addresses used below
    const unsigned v_stage = (unsigned)(ldrow * MG_K + 4 * ldc4) * 4u;               // gate operand staging
    const unsigned v_slab = (unsigned)((((er >> 2)) * 16 + eu) * 4 + (er & 3)) * 4u;  // cell: element of a partial tile

    // indices of the sample for its NEXT per-sample phase, requested one step ahead (a dependent round trip each
    // otherwise): scoring of decode step st -> a_num, vp, target, per lane a the view and sin/cos of candidate a;
    // visual attention over step t's panorama -> vp, view
    int pf_anum = 0, pf_vp = -1, pf_view = 0;
    int64_t pf_tgt = -1;
    float4 pf_sc = make_float4(0.f, 0.f, 0.f, 0.f);
    int pv_vp = -1, pv_view = 0;
    auto fetch_score_idx = [&](int st) {
        const size_t sb = (size_t)min(st, S - 1) * B + bsc;
        pf_anum = p.U.a_num[sb];
        pf_vp = p.U.vp[sb];
        pf_tgt = p.target[sb];
        pf_view = p.U.cand_view[sb * A + min(lane, A - 1)];
        pf_sc = reinterpret_cast<const float4*>(p.U.cand_sincos)[sb * A + min(lane, A - 1)];
    };
    auto fetch_pano_idx = [&](int tt) {
        const size_t sb = (size_t)min(tt, S - 1) * B + bsc;
        pv_vp = p.X.vp[sb];
        pv_view = p.X.view[sb];
    };
    if (text_role) fetch_score_idx(0); else fetch_pano_idx(1);

    f32x4 acc[MT];
    for (int t = 0; t <= S; ++t) {
        t_now = t;
        const unsigned xb = (MG_XIN + (unsigned)((t % 3) * 128 * MG_K)) * 4u;            // operand rows of step t (bytes)
        const unsigned xn = (MG_XIN + (unsigned)(((t + 1) % 3) * 128 * MG_K)) * 4u;      // ... of step t + 1
        const int cs = (t + 2) % 3, cn = t % 3;           // chain step t - 1: its slot, and the one it resets
        const bool chain = t > 0 && chain_active;
        const bool gates = t < S;

        // ============================ gate product ==================================================
        auto issue = [&](v4u (&r)[APASS], int s) {          // operand rows of stage s (XIN columns are in stage order)
#pragma unroll
            for (int pp = 0; pp < APASS; ++pp) r[pp] = xld(rs, v_stage, xb + (unsigned)(pp * 16 * MG_K + s * MG_BK) * 4u);
        };
        auto settle_stage = [&](v4u (&r)[APASS], int s) {
#pragma unroll
            for (int pp = 0; pp < APASS; ++pp) settle(rs, v_stage, xb + (unsigned)(pp * 16 * MG_K + s * MG_BK) * 4u, r[pp], dead);
        };
        auto lstore = [&](const v4u (&r)[APASS], const float4 (&wr)[4], int buf) {
            float* As = smem + buf * BUF;
            float* Ws = As + AROWS * MG_LD;
#pragma unroll
            for (int pp = 0; pp < APASS; ++pp)
                *reinterpret_cast<v4u*>(As + (pp * 16 + ldrow) * MG_LD + 4 * ldc4) = r[pp];
#pragma unroll
            for (int pp = 0; pp < 4; ++pp)
                *reinterpret_cast<float4*>(Ws + (pp * 16 + ldrow) * MG_LD + 4 * ldc4) = wr[pp];
        };
        auto compute = [&](int buf) {
            const float* As = smem + buf * BUF;
            const float* Ws = As + AROWS * MG_LD + (gate * 16 + li) * MG_LD;
#pragma unroll
            for (int c = 0; c < MG_BK / 16; ++c) {
                const float4 bq = *reinterpret_cast<const float4*>(Ws + 16 * c + 4 * kk);
                float4 av[MT];
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    av[m] = *reinterpret_cast<const float4*>(As + (m * 16 + li) * MG_LD + 16 * c + 4 * kk);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[m] = mfma16(comp(av[m], j), comp(bq, j), acc[m]);
            }
        };
        // u stages xcd, xcd + 8, ... (nu of them): the caller has waited for the arrival counter, so the operand
        // rows of ALL stages are requested at once (a round trip is ~2.5 us, a stage's MFMAs 1.5 us)
        auto gate_group_u = [&]() {
            v4u ar[NKMAX][APASS];
#pragma unroll
            for (int k = 0; k < NKMAX; ++k) issue(ar[k], xcd + 8 * min(k, nu - 1));
            settle_stage(ar[0], xcd);
            lstore(ar[0], wres_u[0], 0);
            __syncthreads();
#pragma unroll
            for (int k = 0; k < NKMAX; ++k) {
                if (k < nu) {
                    const bool more = k + 1 < NKMAX && k + 1 < nu;
                    compute(k & 1);
                    if (more) {
                        settle_stage(ar[k + 1 < NKMAX ? k + 1 : NKMAX - 1], xcd + 8 * (k + 1));
                        lstore(ar[k + 1 < NKMAX ? k + 1 : NKMAX - 1], wres_u[k + 1 < NKMAX ? k + 1 : NKMAX - 1], (k + 1) & 1);
                    }
                    __syncthreads();
                }
            }
        };
        // feature / h stages 34 + jx + 4k, k in [k0, k1): weights streamed with the operand rows, one stage ahead
        auto gate_group_stream = [&](int k0, int k1) {
            if (k0 >= k1) return;
            const rsrc_t rwi = make_rs(p.w_ih, (unsigned)(4 * MG_H) * (unsigned)(2 * MG_F) * 4u);
            const rsrc_t rwh = make_rs(p.w_hh, (unsigned)(4 * MG_H) * (unsigned)MG_H * 4u);
            const unsigned vwi = (unsigned)(ldrow * 2 * MG_F + 4 * ldc4) * 4u, vwh = (unsigned)(ldrow * MG_H + 4 * ldc4) * 4u;
            v4u ar[APASS];
            float4 wr[4];
            auto issue_w = [&](int s) {
                const bool is_h = s >= 2 * MG_NU;
#pragma unroll
                for (int pp = 0; pp < 4; ++pp) {
                    const float4 a = bld(rwi, vwi, (unsigned)((pp * MG_H + 16 * slot) * 2 * MG_F + min(s, 2 * MG_NU - 1) * MG_BK) * 4u);
                    const float4 b = bld(rwh, vwh, (unsigned)((pp * MG_H + 16 * slot) * MG_H + max(s - 2 * MG_NU, 0) * MG_BK) * 4u);
                    wr[pp] = is_h ? b : a;
                }
            };
            int s = MG_NU + jx + 4 * k0;
            issue(ar, s);
            issue_w(s);
            settle_stage(ar, s);
            lstore(ar, wr, 0);
            __syncthreads();
            for (int k = k0; k < k1; ++k) {
                const bool more = k + 1 < k1;
                const int sn = MG_NU + jx + 4 * min(k + 1, k1 - 1);
                issue(ar, sn);                              // (re-reads the last stage when there is no next one)
                issue_w(sn);
                compute((k - k0) & 1);
                if (more) {
                    settle_stage(ar, sn);
                    lstore(ar, wr, (k + 1 - k0) & 1);
                }
                __syncthreads();
            }
        };

        MG_STAMP(11)                                        // loop back
        // ============================ chain of step t - 1 ============================================
        if (gates) {
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        constexpr int RPW = MG_LMAX / 4;                    // 20 context rows per wave
        float4 xctx[RPW][2];                                // text role: the sample's context rows
        if (chain && !text_role) {
            wait_count(cnt_h, (unsigned)t * n_cells, dead);
            if (gates) {                                    // q' = M_v h1 + c_v: the query of step t's panorama
                TileJob j{};
                j.a0 = xb + (unsigned)(row0 * MG_K + MG_OFF_H) * 4u; j.lda0 = MG_K;     // h1 of step t - 1
                j.tanh_epi = false; j.m_live = m_live;
                j.w = p.m_v; j.ldw = MG_H; j.w_rows = MG_F; j.bias = p.c_v;
                j.tile0 = e; j.tile_stride = 32; j.ntiles = (MG_TILES_Q - e + 31) >> 5;
                j.out = (MG_Q + (unsigned)((cs * 128 + row0) * MG_F)) * 4u; j.rst = (MG_Q + (unsigned)((cn * 128 + row0) * MG_F)) * 4u;
                j.ldo = MG_F;
                j.dbg = p.dbg_q ? p.dbg_q + ((size_t)t * B + row0) * MG_F : nullptr; j.ld_dbg = MG_F;
                tile_gemm<8, 2, 5>(rs, j, smem, wv, rows_valid, dead);
            }
        }
        if (!(chain && text_role)) {
            MG_STAMP(0)
            if (gates && xcd >= 4) {                        // the h stages (h of step t - 1 is complete: counted)
                if (!chain) wait_count(cnt_h, (unsigned)t * n_cells, dead);
                gate_group_stream(nfeat, nfh);
            }
            MG_STAMP(1)
        }
        if (chain && text_role) {
            // the context rows of the sample depend on nothing: requested before the wait for h so that their
            // ~160 KB land during the first hop (rows beyond L: a clamped, finite row; an invalid sample reads
            // the last valid sample's rows and discards them)
            {
                const rsrc_t rc = make_rs(p.ctx + (size_t)bsc * L * MG_H, (unsigned)(L * MG_H) * 4u);
#pragma unroll
                for (int r = 0; r < RPW; ++r) {
                    const unsigned so = (unsigned)(min(wv * RPW + r, L - 1) * MG_H) * 4u;
#pragma unroll
                    for (int i = 0; i < 2; ++i) xctx[r][i] = bld(rc, vl16 + 1024u * i, so);
                }
            }
            wait_count(cnt_h, (unsigned)t * n_cells, dead);
            {                                               // t_text = W_in h1 (model.py:129)
                TileJob j{};
                j.a0 = xb + (unsigned)(row0 * MG_K + MG_OFF_H) * 4u; j.lda0 = MG_K;     // h1 of step t - 1
                j.tile_stride = 1; j.tanh_epi = false; j.m_live = m_live;
                j.w = p.w_in; j.ldw = MG_H; j.w_rows = MG_H; j.bias = nullptr;
                j.tile0 = e; j.ntiles = 1;
                j.out = (MG_TT + (unsigned)((cs * 128 + row0) * MG_H)) * 4u; j.rst = (MG_TT + (unsigned)((cn * 128 + row0) * MG_H)) * 4u;
                j.ldo = MG_H;
                j.dbg = p.dbg_t_text ? p.dbg_t_text + ((size_t)(t - 1) * B + row0) * MG_H : nullptr; j.ld_dbg = MG_H;
                tile_gemm<8, 1, 1>(rs, j, smem, wv, rows_valid, dead);
            }
            MG_STAMP(0)
            // ---------------- text attention of sample bs (model.py:129-139) --------------------
            {
                float4(*slots)[2 * 64] = reinterpret_cast<float4(*)[2 * 64]>(smem);
                float* s_score = smem + 4 * 2 * 64 * 4;
                const unsigned ob = (MG_WC + (unsigned)((cs * 128 + bs) * MG_H)) * 4u, rb = (MG_WC + (unsigned)((cn * 128 + bs) * MG_H)) * 4u;
                float4 pw[2] = {f4zero(), f4zero()};
                if (sample_ok) {                            // (workgroup-uniform; the loads inside are unconditional)
                    float4 (&x)[RPW][2] = xctx;
                    const unsigned tb = (MG_TT + (unsigned)((cs * 128 + bs) * MG_H)) * 4u;
                    v4u tv[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) tv[i] = xld(rs, vl16 + 1024u * i, tb);
                    park(rs, 0u, tb, dead);
#pragma unroll
                    for (int i = 0; i < 2; ++i) settle(rs, vl16 + 1024u * i, tb, tv[i], dead);
                    const float4 v1[2] = {as_f4(tv[0]), as_f4(tv[1])};
#pragma unroll
                    for (int r = 0; r < RPW; ++r) {
                        float d = dot4(x[r][0], v1[0]) + dot4(x[r][1], v1[1]);
                        d = wave_sum(d);
                        const int l = wv * RPW + r;
                        if (lane == 0 && l < L) s_score[l] = d;
                    }
                    __syncthreads();
                    const int l0 = lane, l1 = lane + 64;
                    const float s0 = (l0 < L && !mask0) ? s_score[l0] : -INFINITY;
                    const float s1 = (l1 < L && !mask1) ? s_score[l1] : -INFINITY;
                    const float m = wave_max(fmaxf(s0, s1));
                    const float e0 = expf(s0 - m);
                    const float e1 = expf(s1 - m);
                    const float inv = 1.0f / wave_sum(e0 + e1);
                    const float w0 = e0 * inv, w1 = e1 * inv;
#pragma unroll
                    for (int r = 0; r < RPW; ++r) {
                        const int l = wv * RPW + r;
                        const int src = (l < L ? l : 0) & 63;
                        const float lo = __shfl(w0, src, WAVE), hi = __shfl(w1, src, WAVE);
                        const float wl = l < L ? (l < 64 ? lo : hi) : 0.f;
                        f4fma(pw[0], wl, x[r][0]);
                        f4fma(pw[1], wl, x[r][1]);
                    }
                }
                float* dbg = (p.dbg_cat2 && sample_ok) ? p.dbg_cat2 + ((size_t)(t - 1) * B + bs) * 2 * MG_H : nullptr;
                const bool poison = dead;
                block_row_sum<2, 4, 4>(pw, slots, MG_H / 4, [&](int c, float4 v) {
                    if (poison) v.x = qnan();
                    xst(rs, 16u * c, ob, as_u4(v));
                    xrst(rs, 16u * c, rb);
                    if (dbg) reinterpret_cast<float4*>(dbg)[c] = v;
                });
                __syncthreads();
            }
            MG_STAMP(2)
            // ---------------- h~ = tanh(W_out [wc ; h1]) (model.py:141-142) ----------------------
            {
                TileJob j{};
                j.a0 = (MG_WC + (unsigned)((cs * 128 + row0) * MG_H)) * 4u; j.lda0 = MG_H;
                j.a1 = xb + (unsigned)(row0 * MG_K + MG_OFF_H) * 4u; j.lda1 = MG_K;
                j.w = p.w_out; j.ldw = 2 * MG_H; j.w_rows = MG_H; j.bias = nullptr;
                j.tile0 = e; j.tile_stride = 1; j.ntiles = 1; j.tanh_epi = true; j.m_live = m_live;
                j.out = (MG_HT + (unsigned)((cs * 128 + row0) * MG_H)) * 4u; j.rst = (MG_HT + (unsigned)((cn * 128 + row0) * MG_H)) * 4u;
                j.ldo = MG_H;
                j.dbg = p.dbg_h_tilde ? p.dbg_h_tilde + ((size_t)(t - 1) * B + row0) * MG_H : nullptr; j.ld_dbg = MG_H;
                tile_gemm<16, 1, 1>(rs, j, smem, wv, rows_valid, dead);
            }
            MG_STAMP(3)
            // ---------------- [r | const] = M_a h~ + c_a (sf_decoder_fold) -----------------------
            {
                TileJob j{};
                j.a0 = (MG_HT + (unsigned)((cs * 128 + row0) * MG_H)) * 4u; j.lda0 = MG_H;
                j.w = p.m_a; j.ldw = MG_H; j.w_rows = MG_F + 4; j.bias = p.c_a;
                j.tile0 = e; j.tile_stride = 32; j.ntiles = (MG_TILES_R - e + 31) >> 5; j.tanh_epi = false; j.m_live = m_live;
                j.out = (MG_R + (unsigned)((cs * 128 + row0) * MG_RLD)) * 4u; j.rst = (MG_R + (unsigned)((cn * 128 + row0) * MG_RLD)) * 4u;
                j.ldo = MG_RLD;
                j.dbg = nullptr; j.ld_dbg = 0;
                tile_gemm<8, 2, 5>(rs, j, smem, wv, rows_valid, dead);
            }
            MG_STAMP(4)
            // ---------------- candidate scores + glue of sample bs (model.py:342-352, follower.py:476-505)
            {
                float* s_logit = smem;
                int* s_at = reinterpret_cast<int*>(smem + 64);
                const int st = t - 1;                       // decode step of this chain
                float4 x[4][9];
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int i = 0; i < 9; ++i) x[k][i] = f4zero();
                int anum = 0;
                int64_t tgt_in = -1;
                float d[4] = {0.f, 0.f, 0.f, 0.f};
                float cst = 0.f;
                if (sample_ok) {                            // (workgroup-uniform; the loads inside are unconditional)
                    // the sample's indices (requested a step ago) are wave-uniform: scalar registers
                    anum = uni(pf_anum);
                    const int vp = uni(pf_vp);
                    tgt_in = pf_tgt;
                    const rsrc_t rt = make_rs(p.U.table + (size_t)max(vp, 0) * MG_V * MG_IMG, (unsigned)(MG_V * MG_IMG) * 4u);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int ca = wv + 4 * k;
                        const bool real = ca > 0 && ca < anum && vp >= 0;   // stop / padding candidates are zero rows:
                        const int cl = real ? ca : 1;                        // they re-read candidate 1 (cache hits) x 0
                        const int view = __builtin_amdgcn_readlane(pf_view, cl);
                        const float4 sc = make_float4(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(pf_sc.x), cl)),
                                                      __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pf_sc.y), cl)),
                                                      __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pf_sc.z), cl)),
                                                      __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pf_sc.w), cl)));
                        const unsigned so = (unsigned)(min(max(view, 0), MG_V - 1) * MG_IMG) * 4u;
                        const float on = real ? 1.f : 0.f;
                        const unsigned vrow = real ? vl16 : 0u;    // (a padding candidate re-reads one 16-byte piece)
#pragma unroll
                        for (int i = 0; i < 8; ++i) {       // image features: 8 chunks of 1 KB
                            const float4 v = bld(rt, vrow + (real ? 1024u * (i & 3) : 0u), so + (real ? 4096u * (i >> 2) : 0u));
                            x[k][i] = make_float4(v.x * on, v.y * on, v.z * on, v.w * on);
                        }
                        // location features (env.py:60-75): sin h, cos h, sin e, cos e, each repeated LOC/4 times
                        const int grp = (lane >> 3) & 3;
                        const float lv = (grp == 0 ? sc.x : (grp == 1 ? sc.y : (grp == 2 ? sc.z : sc.w))) * (lane < 32 ? on : 0.f);
                        x[k][8] = make_float4(lv, lv, lv, lv);
                    }
                    const unsigned rb = (MG_R + (unsigned)((cs * 128 + bs) * MG_RLD)) * 4u;
                    v4u rv[9];
#pragma unroll
                    for (int i = 0; i < 8; ++i) rv[i] = xld(rs, vl16 + 1024u * (i & 3), rb + 4096u * (i >> 2));
                    const unsigned v8 = (unsigned)min(lane, 32) * 16u;      // lanes >= 32: chunk 544 = [const | ..]
                    rv[8] = xld(rs, v8, rb + 8192u);
                    MG_STAMP(12)
                    park(rs, 0u, rb + 8192u, dead);         // (the last tile of r)
                    MG_STAMP(13)
#pragma unroll
                    for (int i = 0; i < 8; ++i) settle(rs, vl16 + 1024u * (i & 3), rb + 4096u * (i >> 2), rv[i], dead);
                    settle(rs, v8, rb + 8192u, rv[8], dead);
#pragma unroll
                    for (int i = 0; i < 9; ++i) {
                        const float4 r4 = as_f4(rv[i]);
#pragma unroll
                        for (int k = 0; k < 4; ++k) d[k] += dot4(x[k][i], r4);   // x is zero beyond the row: the constant chunk adds 0
                    }
                    cst = __shfl(__uint_as_float(rv[8].x), 63, WAVE);           // the constant sits at column F
                }
                MG_STAMP(14)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float dk = wave_sum(d[k]) + cst;
                    if (lane == 0) s_logit[wv + 4 * k] = dk;
                }
                __syncthreads();
                if (wv == 0) {
                    const bool valid = lane < A && lane < anum;
                    const float raw = s_logit[min(lane, MG_AMAX - 1)];
                    float l = valid ? raw : -INFINITY;
                    if (dead && valid) l = qnan();
                    if (sample_ok && lane < A) p.logit[((size_t)st * B + bs) * A + lane] = l;
                    const float m = wave_max(l);
                    const float ex = valid ? expf(l - m) : 0.f;
                    const float se = wave_sum(ex);
                    const float lse = m + logf(se);
                    const bool was_ended = ended_reg;
                    const int64_t tgt = was_ended ? -1 : tgt_in;
                    const float lt = __shfl(l, tgt >= 0 ? (int)tgt : 0, WAVE);
                    const float ce = tgt >= 0 ? (lse - lt) : 0.f;
                    int at;
                    if (p.feedback == 0) {
                        at = tgt > 0 ? (int)tgt : 0;
                    } else if (p.feedback == 1) {
                        const unsigned long long hit = __ballot(lane < A && l == m);
                        at = hit ? (int)__ffsll((long long)hit) - 1 : 0;
                    } else {
                        const uint32_t key = dropout_row_key(p.sample_seed, p.sample_stream0 + (uint32_t)st, (uint32_t)(p.row0 + bs));
                        const float u = (float)(fmix32(key) >> 8) * (1.0f / 16777216.0f) * se;
                        float cdf = ex;
#pragma unroll
                        for (int off = 1; off < 64; off <<= 1) {
                            const float v = __shfl_up(cdf, off, WAVE);
                            if (lane >= off) cdf += v;
                        }
                        const unsigned long long hit = __ballot(valid && cdf > u);
                        const unsigned long long any = __ballot(valid);
                        at = hit ? (int)__ffsll((long long)hit) - 1 : (any ? 63 - __clzll((long long)any) : 0);
                    }
                    at = min(max(at, 0), MG_AMAX - 1);
                    const float la = __shfl(l, at, WAVE);
                    if (lane == 0 && sample_ok) {
                        const size_t o = (size_t)st * B + bs;
                        p.a_t[o] = at;
                        p.target_used[o] = tgt;
                        p.score[o] = la - lse;
                        p.ce_term[o] = ce;
                        p.live[o] = tgt >= 0 ? 1.f : 0.f;
                    }
                    ended_reg = was_ended || at == 0;
                    if (lane == 0) {
                        *s_at = sample_ok ? at : 0;
                        if (sample_ok && t == S) p.ended[bs] = ended_reg ? 1 : 0;
                    }
                }
                __syncthreads();
                const int at = uni(*s_at);
                MG_STAMP(15)
                if (gates && (at & 3) == wv) {              // the wave that holds the chosen row publishes u of step t
                    const int ak = at >> 2;
                    const unsigned ub = xb + (unsigned)(bs * MG_K) * 4u, un = xn + (unsigned)(bs * MG_K) * 4u;
                    float* dbg = (p.dbg_xin && sample_ok) ? p.dbg_xin + ((size_t)t * B + bs) * 2 * MG_F : nullptr;
#pragma unroll
                    for (int i = 0; i < 9; ++i) {
                        const int c = lane + 64 * i;
                        if (c < MG_F4) {
                            float4 v = ak == 0 ? x[0][i] : (ak == 1 ? x[1][i] : (ak == 2 ? x[2][i] : x[3][i]));
                            if (dead) v.x = qnan();
                            xst(rs, 16u * c, ub, as_u4(v));
                            xrst(rs, 16u * c, un);
                            if (dbg) reinterpret_cast<float4*>(dbg)[c] = v;
                        }
                    }
                }
                fetch_score_idx(st + 1);
                if (gates) arrive(cnt_u); else __syncthreads();
            }
            MG_STAMP(5)
        } else if (chain && gates) {
            // ---------------- visual attention of sample bs over the panorama of step t (model.py:310-326)
            float4(*slots)[9 * 64] = reinterpret_cast<float4(*)[9 * 64]>(smem);
            float* s_ml = smem + 4 * 9 * 64 * 4;            // [4] running maxima, [4] sums
            float4 P[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) P[i] = f4zero();
            float m_run = -INFINITY, l_run = 0.f;
            if (sample_ok) {                                // (workgroup-uniform; the loads inside are unconditional)
                const int vp = uni(pv_vp), view = uni(pv_view);
                const float on = vp >= 0 ? 1.f : 0.f;       // vp < 0: an all-zero panorama (padded speaker step)
                const rsrc_t rt = make_rs(p.X.table + (size_t)max(vp, 0) * MG_V * MG_IMG, (unsigned)(MG_V * MG_IMG) * 4u);
                const rsrc_t rl = make_rs(p.X.loc_table + (size_t)view * MG_V * MG_LOC, (unsigned)(MG_V * MG_LOC) * 4u);
                const unsigned v8 = (unsigned)min(lane, 31) * 16u;
                const float on8 = lane < 32 ? on : 0.f;
                float4 q[9];
                const unsigned qb = (MG_Q + (unsigned)((cs * 128 + bs) * MG_F)) * 4u;
                for (int pass = 0; pass < 3; ++pass) {
                    float4 x[3][9];
#pragma unroll
                    for (int jj = 0; jj < 3; ++jj) {
                        const int v = wv + 4 * (3 * pass + jj);                  // < 36 always
#pragma unroll
                        for (int i = 0; i < 8; ++i) x[jj][i] = bld(rt, vl16 + 1024u * (i & 3), (unsigned)(v * MG_IMG) * 4u + 4096u * (i >> 2));
                        x[jj][8] = bld(rl, v8, (unsigned)(v * MG_LOC) * 4u);
                    }
                    if (pass == 0) {
                        v4u qv[9];
#pragma unroll
                        for (int i = 0; i < 8; ++i) qv[i] = xld(rs, vl16 + 1024u * (i & 3), qb + 4096u * (i >> 2));
                        qv[8] = xld(rs, v8, qb + 8192u);
                        park(rs, 0u, qb + 8192u, dead);
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            settle(rs, vl16 + 1024u * (i & 3), qb + 4096u * (i >> 2), qv[i], dead);
                            q[i] = as_f4(qv[i]);
                        }
                        settle(rs, v8, qb + 8192u, qv[8], dead);
                        q[8] = as_f4(qv[8]);
                    }
                    float d[3] = {0.f, 0.f, 0.f};
#pragma unroll
                    for (int jj = 0; jj < 3; ++jj) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            x[jj][i].x *= on; x[jj][i].y *= on; x[jj][i].z *= on; x[jj][i].w *= on;
                            d[jj] += dot4(x[jj][i], q[i]);
                        }
                        x[jj][8].x *= on8; x[jj][8].y *= on8; x[jj][8].z *= on8; x[jj][8].w *= on8;
                        d[jj] += dot4(x[jj][8], q[8]);
                    }
#pragma unroll
                    for (int jj = 0; jj < 3; ++jj) d[jj] = wave_sum(d[jj]);
                    const float mn = fmaxf(fmaxf(m_run, d[0]), fmaxf(d[1], d[2]));
                    const float sc = expf(m_run - mn), e0 = expf(d[0] - mn), e1 = expf(d[1] - mn), e2 = expf(d[2] - mn);
                    l_run = l_run * sc + e0 + e1 + e2;
#pragma unroll
                    for (int i = 0; i < 9; ++i) {
                        P[i].x = P[i].x * sc + e0 * x[0][i].x + e1 * x[1][i].x + e2 * x[2][i].x;
                        P[i].y = P[i].y * sc + e0 * x[0][i].y + e1 * x[1][i].y + e2 * x[2][i].y;
                        P[i].z = P[i].z * sc + e0 * x[0][i].z + e1 * x[1][i].z + e2 * x[2][i].z;
                        P[i].w = P[i].w * sc + e0 * x[0][i].w + e1 * x[1][i].w + e2 * x[2][i].w;
                    }
                    m_run = mn;
                }
            }
            if (lane == 0) { s_ml[wv] = m_run; s_ml[4 + wv] = l_run; }
            __syncthreads();
            float sc = 0.f;
            if (sample_ok) {
                float mm = s_ml[0];
#pragma unroll
                for (int w = 1; w < 4; ++w) mm = fmaxf(mm, s_ml[w]);
                float lt = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) lt += s_ml[4 + w] * expf(s_ml[w] - mm);
                sc = expf(m_run - mm) / lt;
            }
#pragma unroll
            for (int i = 0; i < 9; ++i) { P[i].x *= sc; P[i].y *= sc; P[i].z *= sc; P[i].w *= sc; }
            __syncthreads();
            const unsigned fb = xb + (unsigned)(bs * MG_K + MG_OFF_F) * 4u, fn = xn + (unsigned)(bs * MG_K + MG_OFF_F) * 4u;
            float* dbg = (p.dbg_xin && sample_ok) ? p.dbg_xin + ((size_t)t * B + bs) * 2 * MG_F + MG_F : nullptr;
            const bool poison = dead;
            block_row_sum<9, 4, 4>(P, slots, MG_F4, [&](int c, float4 v) {
                if (poison) v.x = qnan();
                xst(rs, 16u * c, fb, as_u4(v));
                xrst(rs, 16u * c, fn);
                if (dbg) reinterpret_cast<float4*>(dbg)[c] = v;
            });
            fetch_pano_idx(t + 1);
            arrive(cnt_f);
            MG_STAMP(2)
        }
        if (!gates) break;

        // ============================ the rest of the gate product ==================================
        if (xcd >= 4) {                                     // feature stages (visual XCDs), as soon as all of it is there
            wait_count(cnt_f, (unsigned)t * n_chain, dead);
            gate_group_stream(0, nfeat);
        }
        MG_STAMP(7)
        wait_count(cnt_u, (unsigned)t * n_chain, dead);     // u of step t: the end of the text chain
        gate_group_u();
        MG_STAMP(6)
        // the result goes out as [16 x 64] tiles in MFMA layout, one per m-tile, into the region of the
        // workgroup that owns that m-tile's cell for these 16 units
        {
            const unsigned sbuf = (MG_SLAB + (unsigned)((t & 1) * MG_XCD * MG_SLOTS * 8 * 1024)) * 4u;
            const unsigned vs = (unsigned)((kk * 16 + li) * 4) * 4u;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const f32x4 v = acc[m];
                const unsigned so = sbuf + (unsigned)(((m * MG_SLOTS + slot) * 8 + xcd) * 1024 + gate * 256) * 4u;
                xst(rs, vs, so, v4u{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])});
            }
        }
        MG_STAMP(8)
        // ============================ cell update (owner of m-tile xcd, units 16 slot..) ============
        if (active) {
            float pre[4] = {bias[0], bias[1], bias[2], bias[3]};
            const unsigned base = (MG_SLAB + (unsigned)((t & 1) * MG_XCD * MG_SLOTS * 8 * 1024) +
                                   (unsigned)((xcd * MG_SLOTS + slot) * 8 * 1024)) * 4u;
            park(rs, 0u, base + 7u * 4096u, dead);
            {
                unsigned v[4][8];
                const long long t0 = wall_clock64();
                for (;;) {
                    asm volatile("" ::: "memory");
                    bool ok = true;
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int sp = 0; sp < 8; ++sp)
                            v[g][sp] = __builtin_amdgcn_raw_buffer_load_b32(rs, v_slab, base + (unsigned)(sp * 1024 + g * 256) * 4u, MG_SC1);
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int sp = 0; sp < 8; ++sp) ok = ok && v[g][sp] != MG_SENT;
                    if (__all(ok) || dead) break;
                    if (wall_clock64() - t0 > MG_TIMEOUT) { dead = true; break; }
                }
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int sp = 0; sp < 8; ++sp) pre[g] += __uint_as_float(v[g][sp]);
            }
            __syncthreads();                             // every wave has read the region: the owner resets it
            MG_STAMP(9)
#pragma unroll
            for (int i = 0; i < 8; ++i) xrst(rs, (unsigned)tid * 16u, base + 4096u * i);
            {
                const float ig = sigmoidf_(pre[0]), fg = sigmoidf_(pre[1]), gg = tanhf(pre[2]), og = sigmoidf_(pre[3]);
                c_state = fg * c_state + ig * gg;
                float h1 = og * tanhf(c_state);
                if (dead) h1 = qnan();
                // h for step t + 1: own [16 x 16] patch of XIN[(t+1) % 3] (and the reset of XIN[(t+2) % 3])
                const float hp = eb < B ? h1 : 0.f;
                const float h_1 = __shfl_down(hp, 1), h_2 = __shfl_down(hp, 2), h_3 = __shfl_down(hp, 3);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if ((tid & 3) == 0) {
                    const unsigned vo = (unsigned)(er * MG_K + eu) * 4u;
                    const unsigned so = (unsigned)(xcd * 16 * MG_K + MG_OFF_H + 16 * slot) * 4u;
                    xst(rs, vo, xn + so, v4u{__float_as_uint(hp), __float_as_uint(h_1), __float_as_uint(h_2), __float_as_uint(h_3)});
                    xrst(rs, vo, (MG_XIN + (unsigned)(((t + 2) % 3) * 128 * MG_K)) * 4u + so);
                }
                if (evalid) {
                    p.h1_tape[(size_t)t * BH + (size_t)eb * MG_H + ej] = h1;
                    p.c1_tape[(size_t)t * BH + (size_t)eb * MG_H + ej] = c_state;
                }
            }
        }
        if (active) arrive(cnt_h); else __syncthreads();
        MG_STAMP(10)
    }
    if (pa.trace && tid == 0) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            pa.trace[blockIdx.x * 32 + k] = (unsigned long long)tk[k];
            pa.trace[blockIdx.x * 32 + 16 + k] = (unsigned long long)tabs[k];
        }
    }
    if (tid == 0) {
        const unsigned n = atomicAdd(p.done, 1u);
        if (n == gridDim.x - 1) {
            atomicExch(p.done, 0u);
            atomicExch(pa.lock, 0u);
        }
    }
}

}  // namespace

size_t mega_xchg_dwords() { return MG_TOTAL; }

bool mega_supported(int B, int H, int L, int A, const PanoSrc& X, const CandSrc& U) {
    return B >= 1 && B <= 128 && H == MG_H && L >= 1 && L <= MG_LMAX && A >= 2 && A <= MG_AMAX && !X.dense && !U.dense &&
           X.IMG == MG_IMG && X.LOC == MG_LOC && X.V == MG_V && U.IMG == MG_IMG && U.LOC == MG_LOC && U.V == MG_V;
}

int mega_decode(const MegaHost& h, hipStream_t st) {
    if (!mega_supported(h.B, MG_H, h.L, h.U.A, h.X, h.U) || h.S < 1) return SF_ERR_UNSUPPORTED;
    MegaArgs a{};
    a.h = h;
    a.lock = persist_lock_addr();
    a.trace = g_trace;
    if (!a.lock) return SF_ERR_LAUNCH;
    SF_LAUNCH(mega_prologue_kernel, dim3(2048), dim3(256), 0, st, h.xchg, (size_t)MG_TOTAL, a.lock);
    SF_LAUNCH(mega_seed_kernel, dim3(128 * (MG_K / 4) / 256), dim3(256), 0, st, h.xchg + MG_XIN, h.h_init, h.feat0,
              h.ld_feat0, h.B);
    const dim3 grid(MG_XCD * MG_SLOTS), block(256);
    const int need = ceil_div(h.B, 16);
#define SF_MEGA(MTV)                                                                                          \
    {                                                                                                         \
        const size_t gate_lds = (size_t)2 * (MTV * 16 + 64) * MG_LD * sizeof(float);                          \
        const size_t lds = gate_lds > 80 * 1024 ? gate_lds : 80 * 1024;                                       \
        static bool attr_set = false;                                                                         \
        if (!attr_set) {                                                                                      \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mega_kernel<MTV>),                        \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                \
            attr_set = true;                                                                                  \
        }                                                                                                     \
        SF_LAUNCH(mega_kernel<MTV>, grid, block, lds, st, a);                                                 \
    }
#ifdef SF_MEGA_ONLY7                                     /* development builds: one instantiation */
    if (need <= 7) SF_MEGA(7) else return SF_ERR_UNSUPPORTED;
#else
    if (need <= 2) SF_MEGA(2)
    else if (need <= 4) SF_MEGA(4)
    else if (need <= 7) SF_MEGA(7)
    else SF_MEGA(8)
#endif
    return launch_status();
}

}  // namespace sf
