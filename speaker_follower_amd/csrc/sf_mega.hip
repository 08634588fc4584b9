// The follower's decode loop (follower.py:446-531 over model.py:377-399) as ONE persistent launch for
// inference rollouts: S decode steps without a kernel boundary, 256 workgroups x 256 threads (one per CU,
// one wave per SIMD: the phases below keep whole row sets in registers and need the 512-register budget).
//
// Workgroup b: XCD x = b % 8, slot c = b / 8.  Three kinds of work share the launch:
//
//   gate product  (all 256 CUs)  [B, 2F+H] x [4H, 2F+H]^T of the LSTMCell (model.py:393): n-tile = hidden
//                 units [16c, +16) x 4 gates, K split 8 ways by XCD (stage s of 64 k belongs to split
//                 s % 8).  The operand row [u | attended feature | h] of step t lives in the exchange
//                 buffer XIN[t % 3]; partial [16 x 64] tiles go to the SLAB region of the workgroup that
//                 owns the cell of those rows and units, which sums its 8 partials, updates the cell and
//                 publishes h.
//   row group x   (XCD x < ceil(B/16) owns batch rows [16x, +16): the MFMA m-tile).  The chain between two
//                 cells runs inside the group's XCD, sample r = c / 2 of the group:
//        even c:  t_text = W_in h1 (tiles)  ->  text attention of sample r  ->  h~ = tanh(W_out [wc ; h1])
//                 (tiles)  ->  [r | const] = M_a h~ + c_a (tiles, sf_decoder_fold)  ->  candidate scores,
//                 masking, CE term, action, u of the next step (sample r)
//        odd c:   q' = M_v h1 + c_v (tiles)  ->  visual attention of sample r over the NEXT step's panorama
//                 -> attended feature of the next step
//   Every hand-off is data tagged with a sentinel (0xFFFFFFFF, never a finite float): producers publish
//   write-through (sc1), consumers re-read until no sentinel is left, and a producer resets the slot
//   two versions ahead (three slots per buffer; every step is an all-to-all through the gate product, so
//   all readers of a slot are done before it is reset).  Waits are bounded (0.25 s); a timed-out
//   workgroup poisons its outputs with NaN instead of hanging.
//
// Only the inference forward exists in this form (no tapes for a backward, no dropout).
#include "sf_kernels.h"
#include "sf_gemm_small.h"
#include "sf_rows.h"

namespace sf {
namespace {

typedef unsigned v4u __attribute__((ext_vector_type(4)));

constexpr int MG_SLOTS = 32, MG_XCD = 8;
constexpr int MG_H = 512, MG_F = 2176, MG_V = 36, MG_AMAX = 16, MG_LMAX = 80;
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
constexpr int MG_MAXT = 9;                            // tiles per workgroup and phase

// exchange workspace (dword offsets)
constexpr unsigned MG_XIN = 0, MG_XIN_N = 3u * 128 * MG_K;
constexpr unsigned MG_TT = MG_XIN + MG_XIN_N, MG_HN = 3u * 128 * MG_H;
constexpr unsigned MG_WC = MG_TT + MG_HN;
constexpr unsigned MG_HT = MG_WC + MG_HN;
constexpr unsigned MG_Q = MG_HT + MG_HN, MG_Q_N = 3u * 128 * MG_F;
constexpr unsigned MG_R = MG_Q + MG_Q_N, MG_R_N = 3u * 128 * MG_RLD;
constexpr unsigned MG_SLAB = MG_R + MG_R_N, MG_SLAB_N = 2u * MG_XCD * MG_SLOTS * 8 * 1024;
constexpr unsigned MG_TOTAL = MG_SLAB + MG_SLAB_N;

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

__device__ __forceinline__ bool has_sent(const v4u& v) {
    return v.x == MG_SENT || v.y == MG_SENT || v.z == MG_SENT || v.w == MG_SENT;
}
__device__ __forceinline__ v4u xload(__amdgpu_buffer_rsrc_t rs, unsigned dw) {
    return __builtin_amdgcn_raw_buffer_load_b128(rs, dw * 4u, 0, MG_SC1);
}
__device__ __forceinline__ void xstore(__amdgpu_buffer_rsrc_t rs, unsigned dw, const v4u& v) {
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, dw * 4u, 0, MG_SC1);
}
__device__ __forceinline__ void xstore_f(__amdgpu_buffer_rsrc_t rs, unsigned dw, const float4& v) {
    xstore(rs, dw, v4u{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)});
}
__device__ __forceinline__ void xreset(__amdgpu_buffer_rsrc_t rs, unsigned dw) {
    xstore(rs, dw, v4u{MG_SENT, MG_SENT, MG_SENT, MG_SENT});
}
// re-read a published piece until it is complete (bounded)
__device__ __forceinline__ void settle(__amdgpu_buffer_rsrc_t rs, unsigned dw, v4u& v, bool& dead) {
    if (!has_sent(v) || dead) return;
    const long long t0 = wall_clock64();
    while (has_sent(v)) {
        asm volatile("" ::: "memory");
        v = xload(rs, dw);
        if (wall_clock64() - t0 > MG_TIMEOUT) { dead = true; break; }
    }
}
__device__ __forceinline__ float4 as_f4(const v4u& v) {
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ float qnan() { return __uint_as_float(0x7FC00000u); }

// One chain product of a row group: Y[16 rows, tiles] = A[16, K] W[tile rows, K]^T (+ bias, tanh), the
// operand A read from exchange buffers (one or two K segments of 512), the result published tile by tile.
// The 4 waves split K; wave w keeps its K chunk of A in registers and walks the tiles, the partials meet
// in LDS.  The product is formed TRANSPOSED (W as the MFMA row operand), so a lane ends up with 4
// consecutive output columns of one batch row: one b128 store.
struct TileJob {
    unsigned a0, a1;            // dword offset of row 0 of the group in segment 0 / 1
    int lda0, lda1;
    const float* w; int ldw, w_rows;
    const float* bias;
    int tile0, tile_stride, ntiles;
    unsigned out, rst; int ldo;  // publish / reset slot: dword offset of row 0 of the group
    float* dbg; int ld_dbg;      // optional plain copy: pointer to row 0 of the group
    bool tanh_epi;
};
template <int NI, int TC, int MAXT>
__device__ __forceinline__ void tile_gemm(__amdgpu_buffer_rsrc_t rs, const TileJob& j, float* smem, int rows_valid,
                                          bool& dead) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int li = lane & 15, kk = lane >> 4;
    const bool second = NI == 16 && w >= 2;              // NI = 8: K = 512, NI = 16: two segments of 512
    const unsigned abase = (second ? j.a1 : j.a0) + (unsigned)(li * (second ? j.lda1 : j.lda0)) +
                           (unsigned)((NI == 16 ? (w & 1) * 256 : w * 128) + 4 * kk);
    const int kw = (NI == 16 ? w * 256 : w * 128) + 4 * kk;
    f32x4* red = reinterpret_cast<f32x4*>(smem);
    // every load below is unconditional on a clamped index (a load inside a branch would be followed by a
    // full vmcnt(0) wait): chunk c + 1 of the weights is in flight while chunk c is multiplied
    constexpr int NCH = (MAXT + TC - 1) / TC;
    float4 wf[2][TC][NI];
    auto wload = [&](int ch, float4 (&dst)[TC][NI]) {
#pragma unroll
        for (int tc = 0; tc < TC; ++tc) {
            const int tile = j.tile0 + min(ch * TC + tc, j.ntiles - 1) * j.tile_stride;
            const float* wp = j.w + (size_t)min(16 * tile + li, j.w_rows - 1) * j.ldw + kw;
#pragma unroll
            for (int i = 0; i < NI; ++i) dst[tc][i] = ld4(wp + 16 * i);
        }
    };
    wload(0, wf[0]);
    v4u a[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) a[i] = xload(rs, abase + 16 * i);
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        if (ch * TC < j.ntiles || ch == 0) {
            if (ch + 1 < NCH) wload(ch + 1, wf[(ch + 1) & 1]);
            if (ch == 0) {
#pragma unroll
                for (int i = 0; i < NI; ++i) settle(rs, abase + 16 * i, a[i], dead);
            }
#pragma unroll
            for (int tc = 0; tc < TC; ++tc) {
                if (ch * TC + tc < j.ntiles) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int i = 0; i < NI; ++i) {
                        const float4 af = as_f4(a[i]);
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc = mfma16(comp(wf[ch & 1][tc][i], q), comp(af, q), acc);
                    }
                    red[(w * MG_MAXT + ch * TC + tc) * 64 + lane] = acc;
                }
            }
        }
    }
    __syncthreads();
    for (int ti = w; ti < j.ntiles; ti += 4) {
        f32x4 s = red[ti * 64 + lane];
#pragma unroll
        for (int ww = 1; ww < 4; ++ww) s += red[(ww * MG_MAXT + ti) * 64 + lane];
        const int tile = j.tile0 + ti * j.tile_stride;
        const int col = 16 * tile + 4 * kk;
        if (j.bias) {
#pragma unroll
            for (int q = 0; q < 4; ++q) s[q] += j.bias[min(col + q, j.w_rows - 1)];
        }
        if (j.tanh_epi) {
#pragma unroll
            for (int q = 0; q < 4; ++q) s[q] = tanhf(s[q]);
        }
        if (dead) s = f32x4{qnan(), qnan(), qnan(), qnan()};
        const unsigned o = (unsigned)(li * j.ldo + col);
        xstore(rs, j.out + o, v4u{__float_as_uint(s[0]), __float_as_uint(s[1]), __float_as_uint(s[2]), __float_as_uint(s[3])});
        xreset(rs, j.rst + o);
        if (j.dbg && li < rows_valid && col + 3 < j.ld_dbg)
            *reinterpret_cast<float4*>(j.dbg + (size_t)li * j.ld_dbg + col) = make_float4(s[0], s[1], s[2], s[3]);
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
    const int tid = threadIdx.x, lane = tid & 63, wave4 = tid >> 6;
    const int gate = wave4;
    const int li = lane & 15, kk = lane >> 4;
    const int xcd = blockIdx.x & (MG_XCD - 1), slot = blockIdx.x >> 3;
    const int B = p.B, S = p.S, L = p.L, A = p.U.A;
    const int ldrow = tid >> 4, ldc4 = tid & 15;
    const size_t BH = (size_t)B * MG_H;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.xchg, 0, MG_TOTAL * 4, 0x00020000);
    bool dead = false;

    // ---- cell ownership: (row group = xcd, units [16 slot, +16)); every thread owns one element
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

    // ---- per-sample role inside the row group
    const bool even = (slot & 1) == 0;
    const int e = slot >> 1;                              // sample of the group, and tile lane of the group
    const int bs = xcd * 16 + e;
    const bool sample_ok = active && bs < B;
    const int bsc = min(bs, B - 1);
    const int row0 = xcd * 16;                            // first row of the group
    const int rows_valid = max(0, min(16, B - row0));
    bool ended_reg = p.ended[bsc] != 0;
    // instruction mask of the sample (step-invariant): lane l holds positions l and l + 64
    const bool mask0 = p.mask[(size_t)bsc * L + min(lane, L - 1)] != 0;
    const bool mask1 = p.mask[(size_t)bsc * L + min(lane + 64, L - 1)] != 0;

    // ---- this workgroup's stages of the gate product: its h stage, its feature stages, its u stages
    // (computed, not tabulated: a register array filled through a running index compiles to movrel writes
    //  that the compiler also issues speculatively one past the end)
    const int sh = 2 * MG_NU + ((xcd - 2 * MG_NU) & 7);
    const int f0 = MG_NU + ((xcd - MG_NU) & 7), nf = (2 * MG_NU - f0 + 7) >> 3;
    const int nu = (MG_NU - xcd + 7) >> 3;
    auto stage_of = [&](int i) { return i == 0 ? sh : (i <= nf ? f0 + 8 * (i - 1) : xcd + 8 * (i - 1 - nf)); };

    // development aid (pa.trace): per phase the ticks summed over the steps, and the absolute stamps of the
    // middle step
    long long tk[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tabs[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long tprev = wall_clock64();
    int t_now = 0;
#define MG_STAMP(k)                                 \
    if (pa.trace) {                                 \
        const long long now_ = wall_clock64();      \
        tk[k] += now_ - tprev;                      \
        if (t_now == (S >> 1)) tabs[k] = now_;      \
        tprev = now_;                               \
    }

    f32x4 acc[MT];
    for (int t = 0; t <= S; ++t) {
        t_now = t;
        const unsigned xb = MG_XIN + (unsigned)((t % 3) * 128 * MG_K);           // operand rows of step t
        const unsigned xn = MG_XIN + (unsigned)(((t + 1) % 3) * 128 * MG_K);     // ... of step t + 1
        const int cs = (t + 2) % 3, cn = t % 3;           // chain step t - 1: its slot, and the one it resets
        const bool chain = t > 0 && active;
        const bool gates = t < S;

        // ============================ gate product: stages [i0, i1) of this workgroup ================
        struct Regs { v4u a[APASS]; float4 w[4]; };
        auto issue = [&](Regs& r, int s) {
            const int k0 = s * MG_BK;                       // XIN columns are in stage order: u | feat | h
            const bool is_h = s >= 2 * MG_NU;
#pragma unroll
            for (int pp = 0; pp < APASS; ++pp)
                r.a[pp] = xload(rs, xb + (unsigned)((pp * 16 + ldrow) * MG_K + k0 + 4 * ldc4));
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) {
                const int nl = pp * 16 + ldrow;
                const int wrow = (nl >> 4) * MG_H + 16 * slot + (nl & 15);
                r.w[pp] = is_h ? ld4(p.w_hh + (size_t)wrow * MG_H + (k0 - MG_OFF_H) + 4 * ldc4)
                               : ld4(p.w_ih + (size_t)wrow * 2 * MG_F + k0 + 4 * ldc4);
            }
        };
        auto settle_stage = [&](Regs& r, int s) {
            const int k0 = s * MG_BK;
#pragma unroll
            for (int pp = 0; pp < APASS; ++pp)
                settle(rs, xb + (unsigned)((pp * 16 + ldrow) * MG_K + k0 + 4 * ldc4), r.a[pp], dead);
        };
        auto lstore = [&](const Regs& r, int buf) {
            float* As = smem + buf * BUF;
            float* Ws = As + AROWS * MG_LD;
#pragma unroll
            for (int pp = 0; pp < APASS; ++pp)
                *reinterpret_cast<v4u*>(As + (pp * 16 + ldrow) * MG_LD + 4 * ldc4) = r.a[pp];
#pragma unroll
            for (int pp = 0; pp < 4; ++pp)
                *reinterpret_cast<float4*>(Ws + (pp * 16 + ldrow) * MG_LD + 4 * ldc4) = r.w[pp];
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
        auto gate_group = [&](int i0, int i1) {
            if (i0 >= i1) return;
            Regs r0, r1;
            issue(r0, stage_of(i0));
            settle_stage(r0, stage_of(i0));
            lstore(r0, 0);
            __syncthreads();
            for (int i = i0; i < i1; ++i) {
                const bool more = i + 1 < i1;
                if (more) issue(r1, stage_of(i + 1));
                compute((i - i0) & 1);
                if (more) {
                    settle_stage(r1, stage_of(i + 1));
                    lstore(r1, (i + 1 - i0) & 1);
                }
                __syncthreads();
            }
        };

        MG_STAMP(11)                                        // loop back
        // ============================ chain of step t - 1, first product ============================
        if (chain) {
            TileJob j{};
            j.a0 = xb + (unsigned)(row0 * MG_K + MG_OFF_H); j.lda0 = MG_K;          // h1 of step t - 1
            j.tile_stride = 1; j.tanh_epi = false;
            if (even) {                                     // t_text = W_in h1 (model.py:129)
                j.w = p.w_in; j.ldw = MG_H; j.w_rows = MG_H; j.bias = nullptr;
                j.tile0 = 2 * e; j.ntiles = 2;
                j.out = MG_TT + (unsigned)((cs * 128 + row0) * MG_H); j.rst = MG_TT + (unsigned)((cn * 128 + row0) * MG_H);
                j.ldo = MG_H;
                j.dbg = p.dbg_t_text ? p.dbg_t_text + ((size_t)(t - 1) * B + row0) * MG_H : nullptr; j.ld_dbg = MG_H;
                tile_gemm<8, 2, 2>(rs, j, smem, rows_valid, dead);
            } else if (gates) {                             // q' = M_v h1 + c_v: the query of step t's panorama
                j.w = p.m_v; j.ldw = MG_H; j.w_rows = MG_F; j.bias = p.c_v;
                j.tile0 = e; j.tile_stride = 16; j.ntiles = (MG_TILES_Q - e + 15) >> 4;
                j.out = MG_Q + (unsigned)((cs * 128 + row0) * MG_F); j.rst = MG_Q + (unsigned)((cn * 128 + row0) * MG_F);
                j.ldo = MG_F;
                j.dbg = p.dbg_q ? p.dbg_q + ((size_t)t * B + row0) * MG_F : nullptr; j.ld_dbg = MG_F;
                tile_gemm<8, 3, 9>(rs, j, smem, rows_valid, dead);
            }
        }
        MG_STAMP(0)
        if (gates) {
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
            gate_group(0, 1);                               // the h stage
        }
        MG_STAMP(1)
        if (chain && even) {
            // ---------------- text attention of sample bs (model.py:129-139) --------------------
            {
                float4(*slots)[2 * 64] = reinterpret_cast<float4(*)[2 * 64]>(smem);
                float* s_score = smem + 4 * 2 * 64 * 4;
                constexpr int RPW = MG_LMAX / 4;            // 20 context rows per wave
                const float4* ctx = reinterpret_cast<const float4*>(p.ctx) + (size_t)bsc * L * (MG_H / 4);
                float4 x[RPW][2];                           // rows beyond L: a clamped (finite) row, weight 0 below
#pragma unroll
                for (int r = 0; r < RPW; ++r) {
                    const int lc = min(wave4 * RPW + r, L - 1);
#pragma unroll
                    for (int i = 0; i < 2; ++i) x[r][i] = ctx[(size_t)lc * (MG_H / 4) + lane + 64 * i];
                }
                const unsigned tb = MG_TT + (unsigned)((cs * 128 + bs) * MG_H);
                v4u tv[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) tv[i] = xload(rs, tb + 4 * (lane + 64 * i));
#pragma unroll
                for (int i = 0; i < 2; ++i) settle(rs, tb + 4 * (lane + 64 * i), tv[i], dead);
                const float4 v1[2] = {as_f4(tv[0]), as_f4(tv[1])};
#pragma unroll
                for (int r = 0; r < RPW; ++r) {
                    float d = dot4(x[r][0], v1[0]) + dot4(x[r][1], v1[1]);
                    d = wave_sum(d);
                    const int l = wave4 * RPW + r;
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
                float4 pw[2] = {f4zero(), f4zero()};
#pragma unroll
                for (int r = 0; r < RPW; ++r) {
                    const int l = wave4 * RPW + r;
                    const int src = (l < L ? l : 0) & 63;
                    const float lo = __shfl(w0, src, WAVE), hi = __shfl(w1, src, WAVE);
                    const float wl = l < L ? (l < 64 ? lo : hi) : 0.f;
                    f4fma(pw[0], wl, x[r][0]);
                    f4fma(pw[1], wl, x[r][1]);
                }
                const unsigned ob = MG_WC + (unsigned)((cs * 128 + bs) * MG_H), rb = MG_WC + (unsigned)((cn * 128 + bs) * MG_H);
                float* dbg = (p.dbg_cat2 && sample_ok) ? p.dbg_cat2 + ((size_t)(t - 1) * B + bs) * 2 * MG_H : nullptr;
                const bool poison = dead;
                block_row_sum<2, 4, 4>(pw, slots, MG_H / 4, [&](int c, float4 v) {
                    if (!sample_ok) v = f4zero();
                    if (poison) v.x = qnan();
                    xstore_f(rs, ob + 4 * c, v);
                    xreset(rs, rb + 4 * c);
                    if (dbg) reinterpret_cast<float4*>(dbg)[c] = v;
                });
                __syncthreads();
            }
            MG_STAMP(2)
            // ---------------- h~ = tanh(W_out [wc ; h1]) (model.py:141-142) ----------------------
            {
                TileJob j{};
                j.a0 = MG_WC + (unsigned)((cs * 128 + row0) * MG_H); j.lda0 = MG_H;
                j.a1 = xb + (unsigned)(row0 * MG_K + MG_OFF_H); j.lda1 = MG_K;
                j.w = p.w_out; j.ldw = 2 * MG_H; j.w_rows = MG_H; j.bias = nullptr;
                j.tile0 = 2 * e; j.tile_stride = 1; j.ntiles = 2; j.tanh_epi = true;
                j.out = MG_HT + (unsigned)((cs * 128 + row0) * MG_H); j.rst = MG_HT + (unsigned)((cn * 128 + row0) * MG_H);
                j.ldo = MG_H;
                j.dbg = p.dbg_h_tilde ? p.dbg_h_tilde + ((size_t)(t - 1) * B + row0) * MG_H : nullptr; j.ld_dbg = MG_H;
                tile_gemm<16, 1, 2>(rs, j, smem, rows_valid, dead);
            }
            MG_STAMP(3)
            // ---------------- [r | const] = M_a h~ + c_a (sf_decoder_fold) -----------------------
            {
                TileJob j{};
                j.a0 = MG_HT + (unsigned)((cs * 128 + row0) * MG_H); j.lda0 = MG_H;
                j.w = p.m_a; j.ldw = MG_H; j.w_rows = MG_F + 4; j.bias = p.c_a;
                j.tile0 = e; j.tile_stride = 16; j.ntiles = (MG_TILES_R - e + 15) >> 4; j.tanh_epi = false;
                j.out = MG_R + (unsigned)((cs * 128 + row0) * MG_RLD); j.rst = MG_R + (unsigned)((cn * 128 + row0) * MG_RLD);
                j.ldo = MG_RLD;
                j.dbg = nullptr; j.ld_dbg = 0;
                tile_gemm<8, 3, 9>(rs, j, smem, rows_valid, dead);
            }
            MG_STAMP(4)
            // ---------------- candidate scores + glue of sample bs (model.py:342-352, follower.py:476-505)
            {
                float* s_logit = smem;
                int* s_at = reinterpret_cast<int*>(smem + 64);
                const int st = t - 1;                       // decode step of this chain
                CandSrc us = p.U;
                us.vp += (size_t)st * B; us.cand_view += (size_t)st * B * A; us.cand_sincos += (size_t)st * B * A * 4;
                us.a_num += (size_t)st * B;
                const int anum = us.a_num[bsc];
                float4 x[4][9];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int ca = wave4 + 4 * k;
                    const bool real = sample_ok && ca > 0 && ca < anum;     // stop / padding candidates are zero rows:
                    const CandRow cr = cand_row(us, bsc, real ? ca : 1);    // they re-load candidate 1 (cache hits) x 0
#pragma unroll
                    for (int i = 0; i < 9; ++i)
                        x[k][i] = cand_load(cr, lane + 64 * i, real && !cr.zero && lane + 64 * i < MG_F4, MG_F4);
                }
                const int64_t tgt_in = p.target[(size_t)st * B + bsc];
                const unsigned rb = MG_R + (unsigned)((cs * 128 + bs) * MG_RLD);
                v4u rv[9];
#pragma unroll
                for (int i = 0; i < 9; ++i) rv[i] = xload(rs, rb + 4 * min(lane + 64 * i, MG_F4));   // chunk 544 = [const | ..]
#pragma unroll
                for (int i = 0; i < 9; ++i) settle(rs, rb + 4 * min(lane + 64 * i, MG_F4), rv[i], dead);
                float d[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 9; ++i) {
                    const float4 r4 = as_f4(rv[i]);
#pragma unroll
                    for (int k = 0; k < 4; ++k) d[k] += dot4(x[k][i], r4);   // x is zero beyond the row: the constant chunk adds 0
                }
                // the constant sits at column F: chunk 544, held by lanes >= 32 of i = 8
                const float cst = __shfl(__uint_as_float(rv[8].x), 63, WAVE);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float dk = wave_sum(d[k]) + cst;
                    if (lane == 0) s_logit[wave4 + 4 * k] = dk;
                }
                __syncthreads();
                if (wave4 == 0) {
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
                const int at = *s_at;
                if (gates && (at & 3) == wave4) {           // the wave that holds the chosen row publishes u of step t
                    const int ak = at >> 2;
                    const unsigned ub = xb + (unsigned)(bs * MG_K), un = xn + (unsigned)(bs * MG_K);
                    float* dbg = (p.dbg_xin && sample_ok) ? p.dbg_xin + ((size_t)t * B + bs) * 2 * MG_F : nullptr;
#pragma unroll
                    for (int i = 0; i < 9; ++i) {
                        const int c = lane + 64 * i;
                        if (c < MG_F4) {
                            const float4 v = ak == 0 ? x[0][i] : (ak == 1 ? x[1][i] : (ak == 2 ? x[2][i] : x[3][i]));
                            xstore_f(rs, ub + 4 * c, v);
                            xreset(rs, un + 4 * c);
                            if (dbg) reinterpret_cast<float4*>(dbg)[c] = v;
                        }
                    }
                }
                __syncthreads();
            }
            MG_STAMP(5)
        } else if (chain && gates) {
            // ---------------- visual attention of sample bs over the panorama of step t (model.py:310-326)
            float4(*slots)[9 * 64] = reinterpret_cast<float4(*)[9 * 64]>(smem);
            float* s_ml = smem + 4 * 9 * 64 * 4;            // [4] running maxima, [4] sums
            PanoSrc xs = p.X;
            xs.vp += (size_t)t * B; xs.view += (size_t)t * B;
            const PanoRow prow = pano_row(xs, bsc);
            float4 P[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) P[i] = f4zero();
            float m_run = -INFINITY, l_run = 0.f;
            float4 q[9];
            const unsigned qb = MG_Q + (unsigned)((cs * 128 + bs) * MG_F);
            for (int pass = 0; pass < 3; ++pass) {
                float4 x[3][9];
#pragma unroll
                for (int jj = 0; jj < 3; ++jj) {
                    const int v = wave4 + 4 * (3 * pass + jj);               // < 36 always
#pragma unroll
                    for (int i = 0; i < 9; ++i) {
                        const int c = lane + 64 * i;
                        x[jj][i] = pano_load(prow, v, c, c < MG_F4, MG_V, MG_F4);
                    }
                }
                if (pass == 0) {
                    v4u qv[9];
#pragma unroll
                    for (int i = 0; i < 9; ++i) qv[i] = xload(rs, qb + 4 * min(lane + 64 * i, MG_F4 - 1));
#pragma unroll
                    for (int i = 0; i < 9; ++i) {
                        settle(rs, qb + 4 * min(lane + 64 * i, MG_F4 - 1), qv[i], dead);
                        q[i] = lane + 64 * i < MG_F4 ? as_f4(qv[i]) : f4zero();
                    }
                }
                float d[3] = {0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 9; ++i)
#pragma unroll
                    for (int jj = 0; jj < 3; ++jj) d[jj] += dot4(x[jj][i], q[i]);
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
            if (lane == 0) { s_ml[wave4] = m_run; s_ml[4 + wave4] = l_run; }
            __syncthreads();
            float mm = s_ml[0];
#pragma unroll
            for (int w = 1; w < 4; ++w) mm = fmaxf(mm, s_ml[w]);
            float lt = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) lt += s_ml[4 + w] * expf(s_ml[w] - mm);
            const float sc = expf(m_run - mm) / lt;
#pragma unroll
            for (int i = 0; i < 9; ++i) { P[i].x *= sc; P[i].y *= sc; P[i].z *= sc; P[i].w *= sc; }
            __syncthreads();
            const unsigned fb = xb + (unsigned)(bs * MG_K + MG_OFF_F), fn = xn + (unsigned)(bs * MG_K + MG_OFF_F);
            float* dbg = (p.dbg_xin && sample_ok) ? p.dbg_xin + ((size_t)t * B + bs) * 2 * MG_F + MG_F : nullptr;
            const bool poison = dead;
            block_row_sum<9, 4, 4>(P, slots, MG_F4, [&](int c, float4 v) {
                if (!sample_ok) v = f4zero();
                if (poison) v.x = qnan();
                xstore_f(rs, fb + 4 * c, v);
                xreset(rs, fn + 4 * c);
                if (dbg) reinterpret_cast<float4*>(dbg)[c] = v;
            });
            __syncthreads();
            MG_STAMP(2)
        }
        if (!gates) break;

        // ============================ the rest of the gate product ==================================
        gate_group(1 + nf, 1 + nf + nu);                    // u stages (u of step t arrives before the feature)
        MG_STAMP(6)
        gate_group(1, 1 + nf);                              // feature stages
        MG_STAMP(7)
        // the result goes out as [16 x 64] tiles in MFMA layout, one per row group, into the region of the
        // workgroup that owns that group's cell for these 16 units
        {
            const unsigned sbuf = MG_SLAB + (unsigned)((t & 1) * MG_XCD * MG_SLOTS * 8 * 1024);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const f32x4 v = acc[m];
                const unsigned off = sbuf + (unsigned)(((m * MG_SLOTS + slot) * 8 + xcd) * 1024 + ((gate * 4 + kk) * 16 + li) * 4);
                xstore(rs, off, v4u{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])});
            }
        }
        MG_STAMP(8)
        // ============================ cell update (owner of row group xcd, units 16 slot..) =========
        if (active) {
            float pre[4] = {bias[0], bias[1], bias[2], bias[3]};
            const unsigned base = MG_SLAB + (unsigned)((t & 1) * MG_XCD * MG_SLOTS * 8 * 1024) +
                                  (unsigned)((xcd * MG_SLOTS + slot) * 8 * 1024);
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
                            v[g][sp] = __builtin_amdgcn_raw_buffer_load_b32(
                                rs, (base + (unsigned)(sp * 1024 + ((g * 4 + (er >> 2)) * 16 + eu) * 4 + (er & 3))) * 4u, 0, MG_SC1);
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
            for (int i = 0; i < 8; ++i) xreset(rs, base + (unsigned)((tid + 256 * i) * 4));
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
                    const unsigned po = (unsigned)(eb * MG_K + MG_OFF_H + ej);
                    xstore(rs, xn + po, v4u{__float_as_uint(hp), __float_as_uint(h_1), __float_as_uint(h_2), __float_as_uint(h_3)});
                    xreset(rs, MG_XIN + (unsigned)(((t + 2) % 3) * 128 * MG_K) + po);
                }
                if (evalid) {
                    p.h1_tape[(size_t)t * BH + (size_t)eb * MG_H + ej] = h1;
                    p.c1_tape[(size_t)t * BH + (size_t)eb * MG_H + ej] = c_state;
                }
            }
        }
        __syncthreads();
        MG_STAMP(10)
    }
    if (pa.trace && tid == 0) {
#pragma unroll
        for (int k = 0; k < 12; ++k) {
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
    return B >= 1 && B <= 128 && H == MG_H && L >= 1 && L <= MG_LMAX && A >= 1 && A <= MG_AMAX && !X.dense && !U.dense &&
           X.IMG + X.LOC == MG_F && X.V == MG_V && U.IMG + U.LOC == MG_F && U.V == MG_V && (X.LOC % 16) == 0;
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
    if (need <= 2) SF_MEGA(2)
    else if (need <= 4) SF_MEGA(4)
    else if (need <= 7) SF_MEGA(7)
    else SF_MEGA(8)
#undef SF_MEGA
    return launch_status();
}

}  // namespace sf
