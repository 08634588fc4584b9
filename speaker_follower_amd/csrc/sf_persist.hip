// Persistent recurrent kernels: the T dependent time steps of an LSTM whose input is a table lookup
// (EncoderLSTM, model.py:81-104) in ONE launch, with the batch partitioned across the XCDs.
//
// A per-step launch (lstm_step_wide_kernel) is bound by the bytes one workgroup pulls: its 128 KB
// slice of W_hh is re-fetched through the fabric by every launch (an XCD's L2 does not survive a
// kernel boundary) and lands over ~6 us of an 8.4 us step whose MFMAs take 1.7 us.  Here:
//   * grid = 256 workgroups of 4 waves, one per CU; workgroup b belongs to row group b % 8 (the
//     XCD it is dispatched to -- a placement that is observed, used for speed only and never relied
//     on for correctness) and owns hidden units [16 (b / 8), +16) of that group's <= 16 batch rows;
//   * its [64 gate rows x 512] slice of W_hh lives in REGISTERS for all T steps (wave w holds K-slices
//     {w, w+8, w+4, w+12} of 32 for all four gates), the cell state too.  Round 4: the slice is held as
//     three bf16 planes (192 VGPRs) and the recurrent product runs on the BF16 matrix cores with error-free
//     operand splitting (sf_split.h: fp32 accuracy, 6/16 of the fp32-MFMA time -- the MFMAs were 1.9 of a
//     step's 4.5 us); h_t is split by the wave that has just gathered it;
//   * only h_t travels, and only inside a row group: 32 KB per workgroup per step, through the
//     group's own L2 when the placement holds.  The exchange needs no flag, counter or fence: every
//     dword of h IS its own flag.  Three buffers rotate; a producer resets its 1 KB patch of the
//     buffer two steps ahead to a sentinel (0xFFFFFFFF, a NaN no cell can produce) and consumers
//     re-read with L1-bypassing (sc1) loads until no sentinel is left.  Stores are write-through
//     (sc1), so the protocol is correct for ANY workgroup placement.
//   * the K-slice partials are combined as in lstm_step_wide_kernel (16 slices of 32, paired (k, k+8), then
//     added in order to bias + table row); since round 4 the products inside a slice are formed on the bf16
//     matrix cores, so the two paths agree to fp32 roundoff (1e-6), no longer bit for bit.
// Every wait is bounded (wall clock): on a timeout the workgroup stops waiting and poisons its
// outputs with NaN, so a starved launch (co-residency lost to another process) ends instead of
// hanging, and the loss shows it.  Launches of one process are serialised by a device-wide lock
// taken in the prologue kernel and released by the last workgroup.
#include "sf_kernels.h"
#include "sf_gemm_small.h"
#include "sf_sampling.h"
#include "sf_split.h"

namespace sf {
unsigned long long* g_trace = nullptr;     // sf_debug_trace buffer (set in sf_attention.hip)
int g_force_sc1 = 0;                       // sf_debug_force_write_through: never take the one-XCD fast path
long long g_persist_timeout = -1;          // sf_debug_persist_timeout: wait bound in 10 ns ticks, < 0 = the default
namespace {

typedef unsigned v4u __attribute__((ext_vector_type(4)));

constexpr int EP_GROUPS = 8;               // row groups (= XCDs)
constexpr int EP_SLOTS = 32;               // workgroups per group: 16 hidden units each (H = 512)
constexpr int EP_ROWS = 16;                // rows per group (one MFMA m-tile)
constexpr int EP_TMAX = 128;
constexpr unsigned EP_SENTINEL = 0xFFFFFFFFu;
constexpr long long EP_TIMEOUT_TICKS = 25000000LL;   // 0.25 s of the 100 MHz wall clock (default)
constexpr int AUX_SC1 = 16;                // cache-policy bit 4: agent scope (bypass L1 / write through)

__device__ unsigned g_persist_lock = 0;

// The workspace's fault word (`done` + EP_FAULT_WORD; sf_workspace_fault_offset): every wait that gives up ORs
// its launch's code into it.  The NaN poisoning below makes a starved launch visible in its outputs; the fault
// word makes it visible to the HOST, which reads it at its next sync and re-issues the pass on the per-step
// kernels (runtime.take_fault, FollowerEngine.run / SpeakerEngine.run).
constexpr int EP_FAULT_WORD = 40;
constexpr unsigned FAULT_ENC_FWD = 1u, FAULT_ENC_BWD = 2u, FAULT_SPK = 4u, FAULT_LOCK = 8u;
__device__ __forceinline__ void raise_fault(unsigned* fault, unsigned code) {
    if ((threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1) atomicOr(fault, code);   // one lane per wave
}

// Per-launch placement record of one row group: [0] arrivals, [1] min XCC id, [2] max XCC id
// (initialised by the prologue kernel).  EP_PLACE_WORDS dwords per group.
constexpr int EP_PLACE_WORDS = 4;
__device__ __forceinline__ void place_init(unsigned* place, size_t i, int force_sc1) {
    if (i < (size_t)EP_GROUPS * EP_PLACE_WORDS)
        place[i] = (i % EP_PLACE_WORDS) == 1 ? 0xFFFFFFFFu : ((i % EP_PLACE_WORDS) == 3 && force_sc1 ? 1u : 0u);
}
// True iff all EP_SLOTS workgroups of `grp` run on ONE XCD (read from the hardware id register, agreed
// on through agent-scope atomics once per launch).  Then their exchange can stay inside that XCD's
// L2: plain stores (L1 is write-through, the line stays in L2) + L1-bypassing loads -- no write-through
// to the fabric.  Any doubt (timeout, mixed ids) => false => the placement-independent sc1 protocol.
__device__ __forceinline__ bool group_on_one_xcd(unsigned* place, int grp) {
    __shared__ int s_fast;
    if (threadIdx.x == 0) {
        unsigned* p = place + grp * EP_PLACE_WORDS;
        const unsigned xcc = __builtin_amdgcn_s_getreg(63508) & 0xFu;      // hwreg(HW_REG_XCC_ID)
        atomicMin(p + 1, xcc);
        atomicMax(p + 2, xcc);
        __threadfence();
        atomicAdd(p, 1u);
        const long long t0 = wall_clock64();
        bool all = false;
        while (!(all = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)EP_SLOTS)) {
            __builtin_amdgcn_s_sleep(2);
            if (wall_clock64() - t0 > 2000) break;                         // 20 us: decide without the others
        }
        __threadfence();
        s_fast = all && p[3] == 0u &&                                      // [3]: write-through forced (debug)
                 __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ==
                     __hip_atomic_load(p + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    return s_fast != 0;
}

__device__ __forceinline__ void take_persist_lock(unsigned* fault, long long timeout) {
    const long long t0 = wall_clock64();
    while (atomicCAS(&g_persist_lock, 0u, 1u) != 0u) {
        __builtin_amdgcn_s_sleep(32);
        if (wall_clock64() - t0 > 8 * timeout) {          // a lost unlock must not hang the stream
            atomicOr(fault, FAULT_LOCK);
            break;
        }
    }
}

__global__ __launch_bounds__(256) void enc_persist_prologue_kernel(unsigned* xchg, size_t n_xchg, float* h0,
                                                                   float* c0, size_t n_state, unsigned* place,
                                                                   int force_sc1, unsigned* fault, long long timeout) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    place_init(place, (size_t)blockIdx.x * blockDim.x + threadIdx.x, force_sc1);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_xchg; i += stride) xchg[i] = EP_SENTINEL;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_state; i += stride) {
        h0[i] = 0.f;                       // model.py:67-79 init_state
        c0[i] = 0.f;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) take_persist_lock(fault, timeout < 0 ? EP_TIMEOUT_TICKS : timeout);
}

// 16-byte store into an exchange buffer: plain (stays in the XCD's L2) when the group shares an XCD,
// write-through (sc1) otherwise.
template <class RS>
__device__ __forceinline__ void xstore(bool local, v4u v, RS rs, unsigned off) {
    if (local)
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 0);
    else
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, AUX_SC1);
}

struct EncPersistArgs {
    const float* w_hh; const float* b_ih; const float* b_hh;   // [4H,H], [4H], [4H]
    const float* xw_table;                                     // [vocab,4H] = embedding W_ih^T
    const int64_t* seq; long seq_sb, seq_st;                   // token of (row b, step t) = seq[b * seq_sb + t * seq_st]
    const int* lengths;                                        // [B], or null: every row runs all T steps
    const float* h_init; const float* c_init;                  // [B,H] initial state, or null: zero (model.py:67-79)
    int B, H, T, rpg;                                          // rpg = rows per group
    float* gates; float* hs; float* cs;                        // tapes [T,B,4H], [T+1,B,H] x2
    float* ctx; int ld_ctx; Dropout ctx_drop;                  // ctx[b, t, :], row stride T*H
    float* c_out;                                              // [B,H] final cell state (c_T), or null
    unsigned* xchg;                                            // [8][3][16][H] dwords, sentinel-filled
    unsigned* done;                                            // arrival counter (0 before and after)
    unsigned* place;                                           // placement record (group_on_one_xcd)
    unsigned* fault; long long timeout;                        // fault word, wait bound (ticks)
    unsigned long long* trace;                                 // sf_debug_trace: [blocks][8] tick sums, or null
};

__global__ __launch_bounds__(256, 1) void enc_persist_kernel(EncPersistArgs p) {
    __shared__ float s_red[2][8][4][256];               // paired K-slice partials R_k of the 4 gates, double
                                                        // buffered by step parity: ONE barrier per step
    __shared__ int s_tok[EP_ROWS][EP_TMAX];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int li = lane & 15, kk = lane >> 4;
    const int grp = blockIdx.x & (EP_GROUPS - 1), slot = blockIdx.x >> 3;
    const int H = p.H, T = p.T, B = p.B;
    const int row0 = grp * p.rpg;
    const int nrows = max(0, min(p.rpg, B - row0));

    // ---- resident operands --------------------------------------------------------------------
    // wave w: K-slices sj = {w, w+8 | w+4, w+12} (32 k each = chunks 2s, 2s+1), all four gates
    const int sj[4] = {w, w + 8, w + 4, w + 12};
    // (lane (li, kk) holds, of K-slice sj, k = 32 sj + 4 kk + {0..3} and 32 sj + 16 + 4 kk + {0..3}: the K order inside
    // an MFMA is free as long as A and B agree, and this one is what the 16-byte exchange loads deliver)
    Split8 wq[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float* wp = p.w_hh + (size_t)(g * H + 16 * slot + li) * H + 32 * sj[j] + 4 * kk;
            wq[g][j] = split3_f8(ld4(wp), ld4(wp + 16));
        }
    for (int i = tid; i < EP_ROWS * T; i += 256) {
        const int r = i / T, t = i - r * T;
        s_tok[r][t] = r < nrows ? (int)p.seq[(size_t)(row0 + r) * p.seq_sb + (size_t)t * p.seq_st] : 0;
    }
    // the (row, unit) this thread updates; rows beyond the group's share compute on row B-1's
    // operands (clamped, as the per-step kernel does) and store nothing
    const int er = tid >> 4, eu = tid & 15;
    const bool evalid = er < nrows;
    const int eb = evalid ? row0 + er : B - 1;
    const int ej = 16 * slot + eu;
    float bias[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bias[g] = p.b_ih[g * H + ej] + p.b_hh[g * H + ej];
    const int len_b = p.lengths ? p.lengths[eb] : T;
    const uint32_t rk = drop_key(p.ctx_drop, (uint32_t)(p.ctx_drop.row0 + eb));
    float c_state = p.c_init ? p.c_init[(size_t)eb * H + ej] : 0.f;
    float h_state = p.h_init ? p.h_init[(size_t)eb * H + ej] : 0.f;
    const size_t BH = (size_t)B * H;

    unsigned* xg = p.xchg + (size_t)grp * 3 * EP_ROWS * H;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(xg, 0, 3 * EP_ROWS * H * 4, 0x00020000);
    // packed 16-byte access of this workgroup's own [16 x 16] patch: thread with (tid & 3) == 0
    // covers units eu .. eu+3 of row er
    const unsigned patch_off = (unsigned)((er * H + ej) * 4);
    bool dead = false;
    const bool local = group_on_one_xcd(p.place, grp);

    __syncthreads();
    float xv[4];
    {
        const int tok = s_tok[er][0];
#pragma unroll
        for (int g = 0; g < 4; ++g) xv[g] = p.xw_table[(size_t)tok * 4 * H + g * H + ej];
    }

    long long tk[5] = {0, 0, 0, 0, 0}, tprev = wall_clock64();    // development aid (p.trace)
#define EP_STAMP(k)                                 \
    if (p.trace) {                                  \
        const long long now_ = wall_clock64();      \
        tk[k] += now_ - tprev;                      \
        tprev = now_;                               \
    }
    for (int t = 0; t < T; ++t) {
        EP_STAMP(4)                                      // tapes + loop back
        float xn[4] = {0.f, 0.f, 0.f, 0.f};
        if (t > 0 || p.h_init) {                         // h_0 = 0 (no initial state given): the first step has no product
            v4u a[4][2];
            const unsigned base = (unsigned)((((t % 3) * EP_ROWS + li) * H) * 4);
            const long long t0 = wall_clock64();
            if (t == 0) {                                // the given initial state: an input, read where it lies
                const float* hrow = p.h_init + (size_t)min(row0 + li, B - 1) * H;
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int cc = 0; cc < 2; ++cc) {
                        const float4 x = ld4(hrow + 16 * (2 * sj[j] + cc) + 4 * kk);
                        a[j][cc] = v4u{__float_as_uint(x.x), __float_as_uint(x.y), __float_as_uint(x.z), __float_as_uint(x.w)};
                    }
            } else
            for (;;) {
                asm volatile("" ::: "memory");           // the loads below are re-issued every pass
                bool ok = true;
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int cc = 0; cc < 2; ++cc)
                        a[j][cc] = __builtin_amdgcn_raw_buffer_load_b128(
                            rs, base + (unsigned)((16 * (2 * sj[j] + cc) + 4 * kk) * 4), 0, AUX_SC1);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int cc = 0; cc < 2; ++cc)
                        ok = ok && a[j][cc].x != EP_SENTINEL && a[j][cc].y != EP_SENTINEL &&
                             a[j][cc].z != EP_SENTINEL && a[j][cc].w != EP_SENTINEL;
                if (__all(ok) || dead) break;
                if (wall_clock64() - t0 >= p.timeout) {
                    dead = true;
                    raise_fault(p.fault, FAULT_ENC_FWD);
                    break;
                }
            }
            EP_STAMP(0)                                  // waiting for h_t
            // input row of the NEXT step (table lookup by token): lands behind the MFMAs
            if (t + 1 < T) {
                const int tok = s_tok[er][t + 1];
#pragma unroll
                for (int g = 0; g < 4; ++g) xn[g] = p.xw_table[(size_t)tok * 4 * H + g * H + ej];
            }
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                f32x4 hi[2][4], lo[2][4];
#pragma unroll
                for (int g = 0; g < 4; ++g) hi[0][g] = hi[1][g] = lo[0][g] = lo[1][g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int j = 2 * h2 + jj;
                    const Split8 as = split3_f8(
                        make_float4(__uint_as_float(a[j][0].x), __uint_as_float(a[j][0].y), __uint_as_float(a[j][0].z),
                                    __uint_as_float(a[j][0].w)),
                        make_float4(__uint_as_float(a[j][1].x), __uint_as_float(a[j][1].y), __uint_as_float(a[j][1].z),
                                    __uint_as_float(a[j][1].w)));
                    // (per tile, not product by product across the four tiles: measured 325 vs 336 us per launch)
#pragma unroll
                    for (int g = 0; g < 4; ++g) mfma_split6(as, wq[g][j], hi[jj][g], lo[jj][g]);
                }
                // R_k = P_{k+8} + P_k goes to LDS at once (the stores of the first half ride under
                // the MFMAs of the second)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 Rk = (hi[1][g] + lo[1][g]) + (hi[0][g] + lo[0][g]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) s_red[t & 1][w + 4 * h2][g][(kk * 4 + r) * 16 + li] = Rk[r];
                }
            }
        } else {
            if (T > 1) {
                const int tok = s_tok[er][1];
#pragma unroll
                for (int g = 0; g < 4; ++g) xn[g] = p.xw_table[(size_t)tok * 4 * H + g * H + ej];
            }
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int r = 0; r < 4; ++r) s_red[t & 1][w + 4 * h2][g][(kk * 4 + r) * 16 + li] = 0.f;
        }
        __syncthreads();
        // A wave polls only its own 8 of the 32 source slots; behind this barrier the four waves
        // TOGETHER have seen all 32 workgroups publish h_t, i.e. every workgroup of the group has
        // finished reading h_{t-1}.  Its buffer becomes the buffer of h_{t+2}: reset the own patch there
        // (same lanes, same addresses as the later publish; ordered before it by the vmcnt(0) below).
        if (t > 0 && (tid & 3) == 0)
            xstore(local, v4u{EP_SENTINEL, EP_SENTINEL, EP_SENTINEL, EP_SENTINEL}, rs,
                   (unsigned)((((t + 2) % 3) * EP_ROWS) * H * 4) + patch_off);
        EP_STAMP(1)                                      // MFMAs + partials to LDS + barrier
        float g4[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v = bias[g] + xv[g];
#pragma unroll
            for (int k = 0; k < 8; ++k) v += s_red[t & 1][k][g][tid];
            g4[g] = v;
        }
        const float ig = sigmoidf_(g4[0]), fg = sigmoidf_(g4[1]), gg = tanhf(g4[2]), og = sigmoidf_(g4[3]);
        float c1 = fg * c_state + ig * gg;
        float h1 = og * tanhf(c1);
        const bool live = t < len_b;                     // packed sequence (model.py:88-95)
        if (!live) { c1 = c_state; h1 = h_state; }
        if (dead) h1 = __uint_as_float(0x7FC00000u);     // starved launch: poison, do not hang
        // publish h_{t+1} first (it is the critical path), the tapes at leisure
        if (t + 1 < T) {
            const float hp = evalid ? h1 : 0.f;
            const float h_1 = __shfl_down(hp, 1), h_2 = __shfl_down(hp, 2), h_3 = __shfl_down(hp, 3);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the sentinel reset is behind us
            if ((tid & 3) == 0)
                xstore(local, v4u{__float_as_uint(hp), __float_as_uint(h_1), __float_as_uint(h_2), __float_as_uint(h_3)},
                       rs, (unsigned)((((t + 1) % 3) * EP_ROWS) * H * 4) + patch_off);
        }
        EP_STAMP(2)                                      // reduce + cell + publish
        if (evalid) {
            if (p.gates) {
                float* gp = p.gates + ((size_t)t * B + eb) * 4 * H + ej;
                gp[0] = ig; gp[H] = fg; gp[2 * H] = gg; gp[3 * H] = og;
            }
            float cv = live ? h1 : 0.f;
            if (live && p.ctx_drop.on())
                cv = dropout_keep(rk, (uint32_t)(t * H + ej), p.ctx_drop.thresh) ? cv * p.ctx_drop.scale : 0.f;
            if (p.ctx) p.ctx[(size_t)eb * p.ld_ctx + (size_t)t * H + ej] = cv;
            if (p.gates || t == T - 1) {                 // inference (no gates tape): only the final state
                p.cs[(size_t)(t + 1) * BH + (size_t)eb * H + ej] = c1;
                p.hs[(size_t)(t + 1) * BH + (size_t)eb * H + ej] = h1;
            }
            if (t == T - 1 && p.c_out) p.c_out[(size_t)eb * H + ej] = c1;
        }
        c_state = c1;
        h_state = h1;
#pragma unroll
        for (int g = 0; g < 4; ++g) xv[g] = xn[g];
    }
    if (p.trace && lane == 0 && w == 0) {
#pragma unroll
        for (int k = 0; k < 5; ++k) p.trace[blockIdx.x * 8 + k] = (unsigned long long)tk[k];
        p.trace[blockIdx.x * 8 + 5] = local ? 1u : 0u;
    }
    // last workgroup out releases the device-wide lock and re-arms the counter
    __syncthreads();
    if (tid == 0) {
        const unsigned n = atomicAdd(p.done, 1u);
        if (n == gridDim.x - 1) {
            atomicExch(p.done, 0u);
            atomicExch(&g_persist_lock, 0u);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The BACKWARD recurrence of the same LSTM (sf_encoder_lstm_bwd; per-step form:
// lstm_bwd_step_fused_kernel) as one persistent launch.  Same partition: workgroup (group, slot)
// owns hidden units [16 slot, +16) of the group's <= 16 rows, keeps dc and the pass-through part of
// dh in registers and the SAME 64 rows of W_hh as the forward kernel -- now used as the K = 64 slice
// (its own four gates x 16 units) of dh_{t+1} = dgates_{t+1} W_hh.  Per step it
//   1. gathers, for its 16 x 16 patch, the 32 partial sums the group's workgroups published (32 KB,
//      its OWN contiguous region: after reading it resets it to the sentinel, so two buffers suffice),
//   2. runs the cell backward of step t  -> dgates_t (tape, and [16 x 64] into LDS),
//   3. multiplies: partial[16 rows x 512 units] = dgates_t[16 x 64] . W_hh[its 64 rows, :]
//      (128 MFMAs per wave, no K split inside the workgroup),
//   4. publishes the partial as 32 blocks of 1 KB, one per destination workgroup, in the MFMA output
//      layout (one coalesced 16-byte store per lane and block).
// ------------------------------------------------------------------------------------------------
struct EncBwdPersistArgs {
    const float* w_hh;                            // [4H,H]
    const int* lengths;
    int B, H, T, rpg;
    const float* gates; const float* cs;          // tapes [T,B,4H] (activated gates), [T+1,B,H]
    const float* dctx; Dropout ctx_drop;          // gradient wrt the (dropped) h_t handed out at step t, or null:
    long dctx_sb, dctx_st;                        //   element (b, t, j) at dctx[b * dctx_sb + t * dctx_st + j]
    const float* dh_in; const float* dc_in;       // [B,H] incoming dh_T, dc_T
    float* dgates;                                // [T,B,4H] out (pre-activation gate gradients)
    float* dc0_out;                               // [B,H] gradient wrt the initial cell state, or null (zero initial state)
    unsigned* xchg;                               // [8][2][32 dest][32 src][256] dwords, sentinel-filled
    unsigned* done;
    unsigned* place;
    unsigned* fault; long long timeout;
    unsigned long long* trace;
};
constexpr int EB_LDA = 68;                        // LDS row stride of the [16 x 64] dgates tile

__global__ __launch_bounds__(256, 1) void enc_bwd_persist_kernel(EncBwdPersistArgs p) {
    __shared__ float sA[2][EP_ROWS][EB_LDA];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int li = lane & 15, kk = lane >> 4;
    const int grp = blockIdx.x & (EP_GROUPS - 1), slot = blockIdx.x >> 3;
    const int H = p.H, T = p.T, B = p.B;
    const int row0 = grp * p.rpg;
    const int nrows = max(0, min(p.rpg, B - row0));
    // resident: wave w covers units [128 w, +128) = 8 n-tiles; k = 16 kk + c  <->  gate kk, unit c.  Round 4: held as
    // three bf16 planes (sf_split.h: the product runs on the bf16 matrix cores at fp32 accuracy); lane (li, kk) holds
    // units c = 8 s .. 8 s + 7 of gate kk in the operand of MFMA s (K = 64 = 2 x 32)
    Split8 wq[8][2];
#pragma unroll
    for (int nt = 0; nt < 8; ++nt)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            float v[8];
#pragma unroll
            for (int c = 0; c < 8; ++c)
                v[c] = p.w_hh[(size_t)(kk * H + 16 * slot + 8 * s2 + c) * H + 16 * (8 * w + nt) + li];
            wq[nt][s2] = split3_f8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]));
        }
    // this thread's element in the MFMA output layout: index e = (kk*16 + col)*4 + r
    const int er = 4 * (tid >> 6) + (tid & 3), eu = (tid >> 2) & 15;
    const bool evalid = er < nrows;
    const int eb = evalid ? row0 + er : B - 1;
    const int ej = 16 * slot + eu;
    const int len_b = p.lengths ? p.lengths[eb] : T;
    const uint32_t rk = drop_key(p.ctx_drop, (uint32_t)(p.ctx_drop.row0 + eb));
    const size_t BH = (size_t)B * H;
    float dc = p.dc_in ? p.dc_in[(size_t)eb * H + ej] : 0.f;
    float dh_pass = p.dh_in ? p.dh_in[(size_t)eb * H + ej] : 0.f;

    unsigned* xg = p.xchg + (size_t)grp * 2 * EP_SLOTS * EP_SLOTS * 256;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(xg, 0, 2 * EP_SLOTS * EP_SLOTS * 256 * 4, 0x00020000);
    bool dead_wg = false;
    const bool local = group_on_one_xcd(p.place, grp);
    long long tk[5] = {0, 0, 0, 0, 0}, tprev = wall_clock64();

    // operands of step t (activated gates, cell states, dctx), fetched one step ahead
    float g_i, g_f, g_g, g_o, c0v, c1v, dcx;
    auto fetch = [&](int t) {
        const float* gp = p.gates + ((size_t)t * B + eb) * 4 * H + ej;
        g_i = gp[0]; g_f = gp[H]; g_g = gp[2 * H]; g_o = gp[3 * H];
        c0v = p.cs[(size_t)t * BH + (size_t)eb * H + ej];
        c1v = p.cs[(size_t)(t + 1) * BH + (size_t)eb * H + ej];
        dcx = p.dctx ? p.dctx[(size_t)eb * p.dctx_sb + (size_t)t * p.dctx_st + ej] : 0.f;
    };
    fetch(T - 1);

    for (int t = T - 1; t >= 0; --t) {
        EP_STAMP(4)
        float dh = dh_pass;
        if (t < T - 1) {
            // ---- 1. gather the 32 partials of dgates_{t+1} W_hh for this patch
            unsigned v[EP_SLOTS];
            const unsigned base = (unsigned)(((((t + 1) & 1) * EP_SLOTS + slot) * EP_SLOTS) * 256 + tid) * 4u;
            const long long t0 = wall_clock64();
            for (;;) {
                asm volatile("" ::: "memory");
                bool ok = true;
#pragma unroll
                for (int c = 0; c < EP_SLOTS; ++c)
                    v[c] = __builtin_amdgcn_raw_buffer_load_b32(rs, base + (unsigned)(c * 256 * 4), 0, AUX_SC1);
#pragma unroll
                for (int c = 0; c < EP_SLOTS; ++c) ok = ok && v[c] != EP_SENTINEL;
                if (__all(ok) || dead_wg) break;
                if (wall_clock64() - t0 >= p.timeout) {
                    dead_wg = true;
                    raise_fault(p.fault, FAULT_ENC_BWD);
                    break;
                }
            }
            EP_STAMP(0)                                  // waiting for the partials
#pragma unroll
            for (int c = 0; c < EP_SLOTS; ++c) dh += __uint_as_float(v[c]);
        }
        // ---- 2. cell backward of step t (lstm_bwd_step_fused_kernel's arithmetic)
        {
            float vctx = dcx;
            if (p.dctx && p.ctx_drop.on())
                vctx = dropout_keep(rk, (uint32_t)(t * H + ej), p.ctx_drop.thresh) ? vctx * p.ctx_drop.scale : 0.f;
            dh += vctx;
        }
        if (dead_wg) dh = __uint_as_float(0x7FC00000u);
        const bool dead = t >= len_b;                    // packed sequence: the step did not happen
        const float tc = tanhf(c1v);
        const float dout = dh * tc;
        const float dcl = dc + dh * g_o * (1.f - tc * tc);
        const float dgi = dead ? 0.f : dcl * g_g * g_i * (1.f - g_i);
        const float dgf = dead ? 0.f : dcl * c0v * g_f * (1.f - g_f);
        const float dgg = dead ? 0.f : dcl * g_i * (1.f - g_g * g_g);
        const float dgo = dead ? 0.f : dout * g_o * (1.f - g_o);
        dc = dead ? dc : dcl * g_f;
        dh_pass = dead ? dh : 0.f;
        float* sa = &sA[t & 1][er][0];
        const float z = evalid ? 1.f : 0.f;              // rows beyond the group's share contribute nothing
        sa[eu] = dgi * z; sa[16 + eu] = dgf * z; sa[32 + eu] = dgg * z; sa[48 + eu] = dgo * z;
        if (evalid) {
            float* dg = p.dgates + ((size_t)t * B + eb) * 4 * H + ej;
            dg[0] = dgi; dg[H] = dgf; dg[2 * H] = dgg; dg[3 * H] = dgo;
        }
        __syncthreads();                                 // tile complete; every wave has finished its gather
        if (t < T - 1) {                                 // the region just read becomes the region of step t-1
            const unsigned rb = (unsigned)(((((t + 1) & 1) * EP_SLOTS + slot) * EP_SLOTS) * 256) * 4u;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                xstore(local, v4u{EP_SENTINEL, EP_SENTINEL, EP_SENTINEL, EP_SENTINEL}, rs,
                       rb + (unsigned)((tid + 256 * i) * 16));
        }
        EP_STAMP(1)                                      // cell backward + tile + barrier + reset issue
        if (t == 0) {                                    // nothing to publish (the caller forms d h_init = dgates_0 W_hh)
            if (p.dc0_out && evalid) p.dc0_out[(size_t)eb * H + ej] = dc;
            break;
        }
        fetch(t - 1);
        // ---- 3. partial[16 x 512] = dgates_t[16 x 64] . W_hh[own 64 rows, :]
        float a[16];
        {
            const float4* ap = reinterpret_cast<const float4*>(&sA[t & 1][li][16 * kk]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 x = ap[q];
                a[4 * q] = x.x; a[4 * q + 1] = x.y; a[4 * q + 2] = x.z; a[4 * q + 3] = x.w;
            }
        }
        f32x4 acc[8];
        {
            const Split8 a0 = split3_f8(make_float4(a[0], a[1], a[2], a[3]), make_float4(a[4], a[5], a[6], a[7]));
            const Split8 a1 = split3_f8(make_float4(a[8], a[9], a[10], a[11]), make_float4(a[12], a[13], a[14], a[15]));
            f32x4 hi[8], lo[8];
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) hi[nt] = lo[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            mfma_split6_across<8>(a0, [&](int nt) -> const Split8& { return wq[nt][0]; }, hi, lo);
            mfma_split6_across<8>(a1, [&](int nt) -> const Split8& { return wq[nt][1]; }, hi, lo);
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) acc[nt] = hi[nt] + lo[nt];
        }
        // ---- 4. publish: block (dest = 8 w + nt, src = slot), lane's four rows contiguous
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the resets above are behind us
        const unsigned pb = (unsigned)((t & 1) * EP_SLOTS * EP_SLOTS * 256) * 4u;
        EP_STAMP(2)                                      // MFMAs + drain
#pragma unroll
        for (int nt = 0; nt < 8; ++nt)
            xstore(local, v4u{__float_as_uint(acc[nt][0]), __float_as_uint(acc[nt][1]), __float_as_uint(acc[nt][2]),
                              __float_as_uint(acc[nt][3])},
                   rs, pb + (unsigned)((((8 * w + nt) * EP_SLOTS + slot) * 256 + (kk * 16 + li) * 4) * 4));
        EP_STAMP(3)                                      // publish issue
    }
    if (p.trace && lane == 0 && w == 0) {
#pragma unroll
        for (int k = 0; k < 5; ++k) p.trace[blockIdx.x * 8 + k] = (unsigned long long)tk[k];
        p.trace[blockIdx.x * 8 + 5] = local ? 1u : 0u;
    }
    __syncthreads();
    if (tid == 0) {
        const unsigned n = atomicAdd(p.done, 1u);
        if (n == gridDim.x - 1) {
            atomicExch(p.done, 0u);
            atomicExch(&g_persist_lock, 0u);
        }
    }
}

// sentinel fill + device-wide lock of the backward launch (no state to zero)
__global__ __launch_bounds__(256) void enc_bwd_persist_prologue_kernel(unsigned* xchg, size_t n_xchg, unsigned* place,
                                                                       int force_sc1, unsigned* fault, long long timeout) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    place_init(place, (size_t)blockIdx.x * blockDim.x + threadIdx.x, force_sc1);
    v4u* x4 = reinterpret_cast<v4u*>(xchg);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_xchg / 4; i += stride)
        x4[i] = v4u{EP_SENTINEL, EP_SENTINEL, EP_SENTINEL, EP_SENTINEL};
    if (blockIdx.x == 0 && threadIdx.x == 0) take_persist_lock(fault, timeout < 0 ? EP_TIMEOUT_TICKS : timeout);
}

// ------------------------------------------------------------------------------------------------
// The speaker's word loop (Seq2SeqSpeaker._score_obs_actions_and_instructions, speaker.py:158-197 over
// SpeakerDecoderLSTM.forward, model.py:487-519) for INFERENCE as one persistent launch: S word steps of
//   LSTMCell -> attention over the path context -> tanh(linear_out) -> vocabulary projection ->
//   log-softmax / next word (teacher or argmax) / score / NLL term / EOS flag
// with the same partition as the encoder kernels (row group = XCD, 16 hidden units per workgroup)
// and every weight slice register-resident.  Two identities keep the attention inside a workgroup:
//   s_l = ctx_l . (W_in h1) = (ctx_l W_in) . h1            -> cq = ctx W_in     [B,Tp,H], once per call
//   W_out [wc ; h1] = sum_l alpha_l (W_c ctx_l) + W_h h1   -> cw = ctx W_c^T    [B,Tp,H], once per call
// so a workgroup needs only ITS 16 columns of cq / cw (registers) and the step has three exchanges
// inside the row group: (1) h1 + the 16-column partial scores, (2) h~, (3) per-workgroup softmax
// statistics of its 31-32 vocabulary columns (max, arg max, sum exp, target logit).
// ------------------------------------------------------------------------------------------------
constexpr int SP_TPMAX = 12;
constexpr int SPX_H1 = 0;                                     // [16][512]
constexpr int SPX_PS = EP_ROWS * 512;                         // [32 src][16][SP_TPMAX]
constexpr int SPX_HT = SPX_PS + EP_SLOTS * EP_ROWS * SP_TPMAX;   // [16][512]
constexpr int SPX_ST = SPX_HT + EP_ROWS * 512;                // [32 src][16][4]
constexpr int SPX_SS = SPX_ST + EP_SLOTS * EP_ROWS * 4;       // [32 src][16][2]: `sample` feedback only
constexpr int SPX_BUF = SPX_SS + EP_SLOTS * EP_ROWS * 2;      // dwords per (group, buffer)

struct SpkPersistArgs {
    const float* w_hh; const float* b_ih; const float* b_hh; const float* xw_table;
    const float* w_out; int ld_wout;                          // attention linear_out [H,2H]; W_h = columns H..2H
    const float* w_d2a; const float* b_d2a; int vocab, ldv;
    const float* cq; const float* cw;                         // [B,Tp,H]
    const uint8_t* mask;                                      // [B,Tp], 1 = padded path step
    const float* h_init; const float* c_init;
    const int64_t* targets;                                   // [S,B]
    int feedback, pad, eos;
    uint32_t sample_seed, sample_stream; int sample_row0;     // feedback 2 (speaker.py:170-174): counter-based draws
    const uint32_t* sample_site;                              // device-side stream offset (never null), see Dropout.site
    int B, H, Tp, S, rpg;
    int64_t* words;                                           // [S+1,B], row 0 given
    float* step_scores; float* nll_term; float* live;         // [S,B]
    float* logits; float* alpha; float* h1_tape; float* c1_tape;   // optional tapes
    uint8_t* ended;
    unsigned* xchg; unsigned* done; unsigned* place;
    unsigned* fault; long long timeout;
    unsigned long long* trace;
};

template <class RS>
__device__ __forceinline__ void xstore32(bool local, unsigned v, RS rs, unsigned off) {
    if (local)
        __builtin_amdgcn_raw_buffer_store_b32(v, rs, off, 0, 0);
    else
        __builtin_amdgcn_raw_buffer_store_b32(v, rs, off, 0, AUX_SC1);
}
__device__ __forceinline__ float row16_sum(float v) {          // over the 16 lanes of a row
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) v += __shfl_xor(v, off, 16);
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 16));
    return v;
}
__device__ __forceinline__ int row16_min(int v) {
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off, 16));
    return v;
}
// Inverse-CDF pick over 32 weights held two per lane by the 16 lanes of a row (lane eu: items eu and eu + 16, in
// item order): the first item whose inclusive prefix sum exceeds `thr` and whose weight is positive, INT_MAX if none
// (thr rounded up to the total).  This is sf_sampling.h's two-level draw with the slots spread over workgroups.
__device__ __forceinline__ int row16_pick32(float w0, float w1, float thr, int eu) {
    float c0 = w0, c1 = w1;
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {
        const float v0 = __shfl_up(c0, off, 16), v1 = __shfl_up(c1, off, 16);
        if (eu >= off) { c0 += v0; c1 += v1; }
    }
    c1 += __shfl(c0, 15, 16);
    const int cand = (c0 > thr && w0 > 0.f) ? eu : ((c1 > thr && w1 > 0.f) ? eu + 16 : 0x7FFFFFFF);
    return row16_min(cand);
}

template <bool SAMPLING>      // `sample` feedback (speaker.py:170-174): one more pair of dwords in the statistics exchange
__global__ __launch_bounds__(256, 1) void spk_persist_kernel(SpkPersistArgs p) {
    __shared__ float s_part[4][5][256];
    __shared__ float s_voc[4][2][256];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int li = lane & 15, kk = lane >> 4;
    const int grp = blockIdx.x & (EP_GROUPS - 1), slot = blockIdx.x >> 3;
    const int H = p.H, B = p.B, Tp = p.Tp, S = p.S, vocab = p.vocab;
    const int row0 = grp * p.rpg;
    const int nrows = max(0, min(p.rpg, B - row0));
    // ---- resident weights.  Round 4: the four gate tiles and the W_h tile are held as three bf16 planes and their
    // products run on the bf16 matrix cores with error-free operand splitting (sf_split.h: fp32 accuracy, 6/16 of the
    // fp32-MFMA time; these five tiles were 160 of a step's 224 fp32 MFMAs per wave).  The two vocabulary tiles stay
    // fp32 (the register file is full: 512 per lane).
    Split8 wq[5][4];                                    // K-quarter w of: 4 gate tiles, W_h tile; [.][j] = chunks 2j, 2j+1
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = 16 * (8 * w + 2 * j) + 4 * kk;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float* wp = p.w_hh + (size_t)(g * H + 16 * slot + li) * H + k;
            wq[g][j] = split3_f8(ld4(wp), ld4(wp + 16));
        }
        const float* wp = p.w_out + (size_t)(16 * slot + li) * p.ld_wout + H + k;
        wq[4][j] = split3_f8(ld4(wp), ld4(wp + 16));
    }
    float4 wv[2][8];                                    // both vocabulary tiles, this wave's K-quarter
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int vrow = min(32 * slot + 16 * nt + li, vocab - 1);
#pragma unroll
        for (int i = 0; i < 8; ++i) wv[nt][i] = ld4(p.w_d2a + (size_t)vrow * H + 16 * (8 * w + i) + 4 * kk);
    }
    const int er = tid >> 4, eu = tid & 15;
    const bool evalid = er < nrows;
    const int eb = evalid ? row0 + er : B - 1;
    const int ej = 16 * slot + eu;
    float bias[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bias[g] = p.b_ih[g * H + ej] + p.b_hh[g * H + ej];
    float cqr[SP_TPMAX], cwr[SP_TPMAX];
#pragma unroll
    for (int l = 0; l < SP_TPMAX; ++l) {
        const int lc = min(l, Tp - 1);
        const float q = p.cq[((size_t)eb * Tp + lc) * H + ej], c = p.cw[((size_t)eb * Tp + lc) * H + ej];
        cqr[l] = l < Tp ? q : 0.f;
        cwr[l] = l < Tp ? c : 0.f;
    }
    const bool masked = eu < Tp ? (p.mask && p.mask[(size_t)eb * Tp + eu] != 0) : true;   // lane eu <-> path step eu
    const int col0 = 32 * slot + eu, col1 = col0 + 16;
    const float bv0 = col0 < vocab ? p.b_d2a[col0] : 0.f, bv1 = col1 < vocab ? p.b_d2a[col1] : 0.f;
    float c_state = p.c_init[(size_t)eb * H + ej];
    long long wprev = p.words[eb];
    const size_t BH = (size_t)B * H;

    unsigned* xg = p.xchg + (size_t)grp * 3 * SPX_BUF;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(xg, 0, 3 * SPX_BUF * 4, 0x00020000);
    const bool local = group_on_one_xcd(p.place, grp);
    bool dead = false;
    const unsigned patch = (unsigned)(er * H + ej);      // own [16 x 16] patch inside a [16][512] block
    long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = wall_clock64();

    // recurrent part of step 0: h_init W_hh^T (plain loads: h_init is complete before the launch)
    float R[4];
    {
        const int arow = li < nrows ? row0 + li : B - 1;
        f32x4 acc[4], accl[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = accl[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float* hp = p.h_init + (size_t)arow * H + 16 * (8 * w + 2 * j) + 4 * kk;
            const Split8 as = split3_f8(ld4(hp), ld4(hp + 16));
#pragma unroll
            for (int g = 0; g < 4; ++g) mfma_split6(as, wq[g][j], acc[g], accl[g]);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] += accl[g];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) s_part[w][g][(kk * 4 + r) * 16 + li] = acc[g][r];
        __syncthreads();
#pragma unroll
        for (int g = 0; g < 4; ++g) R[g] = (s_part[0][g][tid] + s_part[1][g][tid]) + (s_part[2][g][tid] + s_part[3][g][tid]);
    }
    float xv[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) xv[g] = p.xw_table[(size_t)wprev * 4 * H + g * H + ej];

    for (int t = 0; t < S; ++t) {
        const unsigned bo = (unsigned)((t % 3) * SPX_BUF);               // this step's buffer (dwords)
        const unsigned bn = (unsigned)(((t + 2) % 3) * SPX_BUF);         // the one two steps ahead
        EP_STAMP(7)
        // ---- A. LSTM cell (model.py:515)
        const float ig = sigmoidf_(bias[0] + xv[0] + R[0]), fg = sigmoidf_(bias[1] + xv[1] + R[1]);
        const float gg = tanhf(bias[2] + xv[2] + R[2]), og = sigmoidf_(bias[3] + xv[3] + R[3]);
        c_state = fg * c_state + ig * gg;
        float h1 = og * tanhf(c_state);
        if (dead) h1 = __uint_as_float(0x7FC00000u);
        // ---- B. publish h1 and this workgroup's share of the attention scores
        {
            const float hp = evalid ? h1 : 0.f;
            float ps = 0.f;
#pragma unroll
            for (int l = 0; l < SP_TPMAX; ++l) {
                const float v = row16_sum(cqr[l] * hp);
                ps = eu == l ? v : ps;
            }
            const float h_1 = __shfl_down(hp, 1), h_2 = __shfl_down(hp, 2), h_3 = __shfl_down(hp, 3);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if ((tid & 3) == 0)
                xstore(local, v4u{__float_as_uint(hp), __float_as_uint(h_1), __float_as_uint(h_2), __float_as_uint(h_3)},
                       rs, (bo + SPX_H1 + patch) * 4u);
            if (eu < SP_TPMAX)
                xstore32(local, __float_as_uint(ps), rs, (bo + SPX_PS + (unsigned)((slot * EP_ROWS + er) * SP_TPMAX + eu)) * 4u);
        }
        if (evalid && p.h1_tape) {                       // tapes AFTER the publish: nothing waits on them for a step
            p.h1_tape[(size_t)t * BH + (size_t)eb * H + ej] = h1;
            p.c1_tape[(size_t)t * BH + (size_t)eb * H + ej] = c_state;
        }
        EP_STAMP(0)                                      // cell + partial scores + publish
        // ---- C. gather h1 (this wave's K-quarter) and the 32 partial scores of (row er, path step eu)
        v4u a[8];
        unsigned sc[EP_SLOTS];
        {
            const unsigned ab = (bo + SPX_H1 + (unsigned)(li * H)) * 4u;
            const unsigned sb = (bo + SPX_PS + (unsigned)(er * SP_TPMAX + min(eu, SP_TPMAX - 1))) * 4u;
            const long long t0 = wall_clock64();
            for (;;) {
                asm volatile("" ::: "memory");
                bool ok = true;
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    a[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, ab + (unsigned)((16 * (8 * w + i) + 4 * kk) * 4), 0, AUX_SC1);
#pragma unroll
                for (int c = 0; c < EP_SLOTS; ++c)
                    sc[c] = __builtin_amdgcn_raw_buffer_load_b32(rs, sb + (unsigned)(c * EP_ROWS * SP_TPMAX * 4), 0, AUX_SC1);
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    ok = ok && a[i].x != EP_SENTINEL && a[i].y != EP_SENTINEL && a[i].z != EP_SENTINEL && a[i].w != EP_SENTINEL;
#pragma unroll
                for (int c = 0; c < EP_SLOTS; ++c) ok = ok && sc[c] != EP_SENTINEL;
                if (__all(ok) || dead) break;
                if (wall_clock64() - t0 >= p.timeout) { dead = true; raise_fault(p.fault, FAULT_SPK); break; }
            }
        }
        EP_STAMP(1)                                      // wait for h1 + scores
        // h1 / h~ / partial scores of step t-1 were all consumed in front of a workgroup barrier of step t-1
        // (phases D and F), so any wave that published for step t proves its whole workgroup is done with
        // them: reset the own regions of the buffer of step t+2
        if ((tid & 3) == 0) {
            xstore(local, v4u{EP_SENTINEL, EP_SENTINEL, EP_SENTINEL, EP_SENTINEL}, rs, (bn + SPX_H1 + patch) * 4u);
            xstore(local, v4u{EP_SENTINEL, EP_SENTINEL, EP_SENTINEL, EP_SENTINEL}, rs, (bn + SPX_HT + patch) * 4u);
        }
        if (eu < SP_TPMAX)
            xstore32(local, EP_SENTINEL, rs, (bn + SPX_PS + (unsigned)((slot * EP_ROWS + er) * SP_TPMAX + eu)) * 4u);
        // statistics rows: what this wave has seen (the partial scores of rows [4w, 4w+4) from all 32 slots) proves
        // only that WAVE w of every workgroup is past phase H of step t-1 -- those waves are the only readers of rows
        // [4w, 4w+4) of this block, so each wave resets exactly the rows its own evidence covers (the end-of-step
        // __syncthreads orders the waves of ONE workgroup, not the other workgroups' readers)
        if (lane < 4) {
            xstore(local, v4u{EP_SENTINEL, EP_SENTINEL, EP_SENTINEL, EP_SENTINEL}, rs,
                   (bn + SPX_ST + (unsigned)((slot * EP_ROWS + 4 * w + lane) * 4)) * 4u);
            if (SAMPLING) {
                xstore32(local, EP_SENTINEL, rs, (bn + SPX_SS + (unsigned)((slot * EP_ROWS + 4 * w + lane) * 2)) * 4u);
                xstore32(local, EP_SENTINEL, rs, (bn + SPX_SS + (unsigned)((slot * EP_ROWS + 4 * w + lane) * 2 + 1)) * 4u);
            }
        }
        // attention weights of row er: lane eu holds path step eu (model.py:131-137)
        float alpha_l;
        {
            float sl = 0.f;
#pragma unroll
            for (int c = 0; c < EP_SLOTS; ++c) sl += __uint_as_float(sc[c]);
            sl = masked ? -INFINITY : sl;
            const float m = row16_max(sl);
            const float e = masked ? 0.f : expf(sl - m);
            alpha_l = e / row16_sum(e);
            if (slot == 0 && evalid && eu < Tp && p.alpha) p.alpha[((size_t)t * B + eb) * Tp + eu] = alpha_l;
        }
        // h1 (this wave's K-quarter) as bf16 planes: the operand of the W_h tile and of the four gate tiles
        Split8 as[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            as[j] = split3_f8(make_float4(__uint_as_float(a[2 * j].x), __uint_as_float(a[2 * j].y), __uint_as_float(a[2 * j].z),
                                          __uint_as_float(a[2 * j].w)),
                              make_float4(__uint_as_float(a[2 * j + 1].x), __uint_as_float(a[2 * j + 1].y),
                                          __uint_as_float(a[2 * j + 1].z), __uint_as_float(a[2 * j + 1].w)));
        // ---- D1. W_h h1 over this wave's K-quarter: the one tile h~ waits for.  (The four gate tiles of the NEXT
        //      step's cell are not needed before this step's word is known: they are formed in D2, behind the h~
        //      publish, while the other workgroups' h~ tiles are still on their way.)
        {
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f}, accl = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) mfma_split6(as[j], wq[4][j], acc, accl);
            acc += accl;
            // (s_part[.][4] was last read in phase E of the previous step, in front of that step's phase-F barrier)
#pragma unroll
            for (int r = 0; r < 4; ++r) s_part[w][4][(kk * 4 + r) * 16 + li] = acc[r];
            __syncthreads();
        }
        EP_STAMP(2)                                      // softmax + W_h tile + LDS reduce
        // ---- E. h~ = tanh(sum_l alpha_l cw_l + W_h h1)   (model.py:139-141, folded)
        float ht = (s_part[0][4][tid] + s_part[1][4][tid]) + (s_part[2][4][tid] + s_part[3][4][tid]);
#pragma unroll
        for (int l = 0; l < SP_TPMAX; ++l) ht += __shfl(alpha_l, (lane & 48) + l, 64) * cwr[l];
        ht = tanhf(ht);
        {
            const float hp = evalid ? ht : 0.f;
            const float h_1 = __shfl_down(hp, 1), h_2 = __shfl_down(hp, 2), h_3 = __shfl_down(hp, 3);
            if ((tid & 3) == 0)      // (its region was reset two steps ago, behind the drain of step t-1's publish)
                xstore(local, v4u{__float_as_uint(hp), __float_as_uint(h_1), __float_as_uint(h_2), __float_as_uint(h_3)},
                       rs, (bo + SPX_HT + patch) * 4u);
        }
        // ---- D2a. the next step's recurrent gates h1 . W_hh^T (this wave's K-quarter), input and forget gate, under
        //      the h~ exchange; cell and output gate follow in D2b under the exchange of the softmax statistics.
        //      The partials meet in LDS behind phase F's barrier (a) and behind the barrier that ends the step (b);
        //      the last readers of s_part[.][0..3] -- the R sums of the previous step -- are behind those.
        auto gate_pair = [&](int g0) {
            f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            f32x4 accl[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int g = 0; g < 2; ++g) mfma_split6(as[j], wq[g0 + g][j], acc[g], accl[g]);   // (the across-tiles order spills here)
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r) s_part[w][g0 + g][(kk * 4 + r) * 16 + li] = acc[g][r] + accl[g][r];
        };
        gate_pair(0);
        EP_STAMP(3)                                      // h~ + publish + gate tiles
        // ---- F. vocabulary projection of this workgroup's 32 columns (model.py:518)
        {
            v4u av[8];
            const unsigned hb = (bo + SPX_HT + (unsigned)(li * H)) * 4u;
            const long long t0 = wall_clock64();
            for (;;) {
                asm volatile("" ::: "memory");
                bool ok = true;
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    av[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, hb + (unsigned)((16 * (8 * w + i) + 4 * kk) * 4), 0, AUX_SC1);
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    ok = ok && av[i].x != EP_SENTINEL && av[i].y != EP_SENTINEL && av[i].z != EP_SENTINEL && av[i].w != EP_SENTINEL;
                if (__all(ok) || dead) break;
                if (wall_clock64() - t0 >= p.timeout) { dead = true; raise_fault(p.fault, FAULT_SPK); break; }
            }
            EP_STAMP(4)                                  // wait for h~
            f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float4 af = make_float4(__uint_as_float(av[i].x), __uint_as_float(av[i].y),
                                              __uint_as_float(av[i].z), __uint_as_float(av[i].w));
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) acc[nt] = mfma16(comp(af, c), comp(wv[nt][i], c), acc[nt]);
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) s_voc[w][nt][(kk * 4 + r) * 16 + li] = acc[nt][r];
            __syncthreads();
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) R[g] = (s_part[0][g][tid] + s_part[1][g][tid]) + (s_part[2][g][tid] + s_part[3][g][tid]);
        const float l0 = (s_voc[0][0][tid] + s_voc[1][0][tid]) + (s_voc[2][0][tid] + s_voc[3][0][tid]) + bv0;
        const float l1 = (s_voc[0][1][tid] + s_voc[1][1][tid]) + (s_voc[2][1][tid] + s_voc[3][1][tid]) + bv1;
        if (evalid && p.logits) {
            float* lp = p.logits + ((size_t)t * B + eb) * p.ldv;
            if (col0 < vocab) lp[col0] = l0;
            if (col1 < vocab) lp[col1] = l1;
        }
        // ---- G. softmax statistics of these 32 columns for row er (speaker.py:163-182)
        const long long tgt = p.targets[(size_t)t * B + eb];
        {
            const float x0 = col0 < vocab ? l0 : -INFINITY, x1 = col1 < vocab ? l1 : -INFINITY;
            float m = x0;
            int am = col0;
            if (x1 > m) { m = x1; am = col1; }
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) {
                const float om = __shfl_xor(m, off, 16);
                const int oa = __shfl_xor(am, off, 16);
                if (om > m || (om == m && oa < am)) { m = om; am = oa; }
            }
            const float e0 = col0 < vocab ? expf(l0 - m) : 0.f, e1 = col1 < vocab ? expf(l1 - m) : 0.f;
            const float se = row16_sum(e0 + e1);
            const float tl = row16_sum((col0 == tgt ? l0 : 0.f) + (col1 == tgt ? l1 : 0.f));
            if (eu == 0)
                xstore(local, v4u{__float_as_uint(m), (unsigned)am, __float_as_uint(se), __float_as_uint(tl)}, rs,
                       (bo + SPX_ST + (unsigned)((slot * EP_ROWS + er) * 4)) * 4u);
            if (SAMPLING) {
                // speaker.py:170-174 (Categorical(probs).sample()), two-level: THIS workgroup's draw given that the
                // word falls into its 32 columns (second uniform of the row); which workgroup's draw counts is
                // decided in phase H from the first uniform and the published masses (sf_sampling.h)
                float u1, u2;
                sample_uniforms(p.sample_seed + 0x9E3779B9u * site_value(p.sample_site), p.sample_stream + (uint32_t)t, (uint32_t)(p.sample_row0 + eb), &u1, &u2);
                const int it = row16_pick32(e0, e1, u2 * se, eu);
                const int sc_ = it != 0x7FFFFFFF ? 32 * slot + it : min(32 * slot + 31, vocab - 1);
                const float sl_ = row16_sum((sc_ == col0 ? l0 : 0.f) + (sc_ == col1 ? l1 : 0.f));
                if (eu == 0) {
                    xstore32(local, (unsigned)sc_, rs, (bo + SPX_SS + (unsigned)((slot * EP_ROWS + er) * 2)) * 4u);
                    xstore32(local, __float_as_uint(sl_), rs, (bo + SPX_SS + (unsigned)((slot * EP_ROWS + er) * 2 + 1)) * 4u);
                }
            }
        }
        gate_pair(2);                                    // D2b: under the exchange of the statistics
        EP_STAMP(5)                                      // vocabulary MFMA + statistics + publish + gate tiles
        // ---- H. combine the 32 workgroups' statistics: every lane of row er learns the word
        float M, Z, tlog, slog = 0.f;
        int arg, sarg = 0;
        {
            v4u s0, s1;
            unsigned q0c = 0, q0l = 0, q1c = 0, q1l = 0;            // `sample`: (column, logit) drawn by slots eu, eu + 16
            const unsigned sb0 = (bo + SPX_ST + (unsigned)((eu * EP_ROWS + er) * 4)) * 4u;
            const unsigned sb1 = (bo + SPX_ST + (unsigned)(((eu + 16) * EP_ROWS + er) * 4)) * 4u;
            const unsigned qb0 = (bo + SPX_SS + (unsigned)((eu * EP_ROWS + er) * 2)) * 4u;
            const unsigned qb1 = (bo + SPX_SS + (unsigned)(((eu + 16) * EP_ROWS + er) * 2)) * 4u;
            constexpr bool sampling = SAMPLING;
            const long long t0 = wall_clock64();
            for (;;) {
                asm volatile("" ::: "memory");
                s0 = __builtin_amdgcn_raw_buffer_load_b128(rs, sb0, 0, AUX_SC1);
                s1 = __builtin_amdgcn_raw_buffer_load_b128(rs, sb1, 0, AUX_SC1);
                if (sampling) {                              // (compile-time: no load under a run-time branch)
                    q0c = __builtin_amdgcn_raw_buffer_load_b32(rs, qb0, 0, AUX_SC1);
                    q0l = __builtin_amdgcn_raw_buffer_load_b32(rs, qb0 + 4u, 0, AUX_SC1);
                    q1c = __builtin_amdgcn_raw_buffer_load_b32(rs, qb1, 0, AUX_SC1);
                    q1l = __builtin_amdgcn_raw_buffer_load_b32(rs, qb1 + 4u, 0, AUX_SC1);
                }
                const bool qok = !sampling || (q0c != EP_SENTINEL && q0l != EP_SENTINEL && q1c != EP_SENTINEL && q1l != EP_SENTINEL);
                const bool ok = qok &&
                                s0.x != EP_SENTINEL && s0.z != EP_SENTINEL && s1.x != EP_SENTINEL && s1.z != EP_SENTINEL &&
                                s0.w != EP_SENTINEL && s1.w != EP_SENTINEL && s0.y != EP_SENTINEL && s1.y != EP_SENTINEL;
                if (__all(ok) || dead) break;
                if (wall_clock64() - t0 >= p.timeout) { dead = true; raise_fault(p.fault, FAULT_SPK); break; }
            }
            float m0 = __uint_as_float(s0.x), m1 = __uint_as_float(s1.x);
            int a0 = (int)s0.y, a1 = (int)s1.y;
            float z0 = __uint_as_float(s0.z), z1 = __uint_as_float(s1.z);
            M = m0; arg = a0; Z = z0;
            if (m1 > M || (m1 == M && a1 < arg)) { arg = a1; }
            {
                const float mm = fmaxf(m0, m1);
                Z = z0 * wexp(m0, mm) + z1 * wexp(m1, mm);
                M = mm;
            }
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) {
                const float om = __shfl_xor(M, off, 16), oz = __shfl_xor(Z, off, 16);
                const int oa = __shfl_xor(arg, off, 16);
                const float mm = fmaxf(M, om);
                Z = Z * wexp(M, mm) + oz * wexp(om, mm);
                if (om > M || (om == M && oa < arg)) arg = oa;
                M = mm;
            }
            const int ts = (int)min(max(tgt, 0LL), (long long)vocab - 1) >> 5;        // workgroup that owns the target column
            const float mine = (ts & 16) ? __uint_as_float(s1.w) : __uint_as_float(s0.w);
            tlog = __shfl(mine, (lane & 48) + (ts & 15), 64);
            if (sampling) {
                // which workgroup's draw counts: inverse CDF over the 32 masses z_s exp(m_s - M) with the row's first uniform
                float u1, u2;
                sample_uniforms(p.sample_seed + 0x9E3779B9u * site_value(p.sample_site), p.sample_stream + (uint32_t)t, (uint32_t)(p.sample_row0 + eb), &u1, &u2);
                int cs = row16_pick32(z0 * wexp(m0, M), z1 * wexp(m1, M), u1 * Z, eu);
                if (cs == 0x7FFFFFFF) cs = arg >> 5;                                 // u1 Z rounded up to the total
                const unsigned qc = (cs & 16) ? q1c : q0c, ql = (cs & 16) ? q1l : q0l;
                sarg = (int)__shfl(qc, (lane & 48) + (cs & 15), 64);
                slog = __uint_as_float(__shfl(ql, (lane & 48) + (cs & 15), 64));
            }
        }
        EP_STAMP(6)                                      // wait for the statistics + combine
        const float lse = M + logf(Z);
        // (a starved launch may hold the sentinel in `arg`: the word fed back must stay a table row)
        const long long wt = dead ? (long long)p.pad
                                  : (SAMPLING ? (long long)sarg : (p.feedback == 0 ? tgt : (long long)arg));
        const float lw = SAMPLING ? slog : (p.feedback == 0 ? tlog : M);
        if (slot == 0 && eu == 0 && evalid) {
            const size_t o = (size_t)t * B + eb;
            p.words[(size_t)(t + 1) * B + eb] = wt;
            float scv = wt != p.pad ? lw - lse : 0.f;
            float nl = tgt != p.pad ? lse - tlog : 0.f;
            if (dead) scv = nl = __uint_as_float(0x7FC00000u);
            p.step_scores[o] = scv;
            p.nll_term[o] = nl;
            p.live[o] = tgt != p.pad ? 1.f : 0.f;
            if (wt == p.eos) p.ended[eb] = 1;
        }
        wprev = wt;
        // ---- I. input row of the next step (model.py:497 + the hoisted W_ih product)
#pragma unroll
        for (int g = 0; g < 4; ++g) xv[g] = p.xw_table[(size_t)wprev * 4 * H + g * H + ej];
        __syncthreads();                                 // D2b's partials (cell and output gate)
#pragma unroll
        for (int g = 2; g < 4; ++g) R[g] = (s_part[0][g][tid] + s_part[1][g][tid]) + (s_part[2][g][tid] + s_part[3][g][tid]);
    }
    if (p.trace && lane == 0 && w == 0) {
#pragma unroll
        for (int kx = 0; kx < 8; ++kx) p.trace[blockIdx.x * 8 + kx] = (unsigned long long)tk[kx];
    }
    __syncthreads();
    if (tid == 0) {
        const unsigned n = atomicAdd(p.done, 1u);
        if (n == gridDim.x - 1) {
            atomicExch(p.done, 0u);
            atomicExch(&g_persist_lock, 0u);
        }
    }
}

int device_cus() {
    static thread_local int cached_dev = -1, cached = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (dev != cached_dev) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
        cached_dev = dev;
        cached = n;
    }
    return cached;
}

}  // namespace

// Device address of the ONE device-wide lock all persistent launches of this library take (a __device__
// variable of this translation unit; other translation units pass the pointer to their kernels).
unsigned* persist_lock_addr() {
    static thread_local int cached_dev = -1;
    static thread_local unsigned* cached = nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    if (dev != cached_dev) {
        void* ptr = nullptr;
        if (hipGetSymbolAddress(&ptr, HIP_SYMBOL(g_persist_lock)) != hipSuccess) return nullptr;
        cached = static_cast<unsigned*>(ptr);
        cached_dev = dev;
    }
    return cached;
}

size_t persistent_fault_word() { return EP_FAULT_WORD; }
size_t encoder_persistent_xchg_floats(int H) { return (size_t)EP_GROUPS * 3 * EP_ROWS * H; }

bool encoder_persistent_supported(int B, int H, int T) {
    return H == 16 * EP_SLOTS && B >= 1 && B <= EP_GROUPS * EP_ROWS && T >= 1 && T <= EP_TMAX &&
           device_cus() >= EP_GROUPS * EP_SLOTS;
}

int encoder_persistent(const float* w_hh, const float* b_ih, const float* b_hh, const float* xw_table,
                       const int64_t* seq, int Lpad, const int* lengths, int B, int H, int T, float* gates,
                       float* hs, float* cs, float* ctx, const Dropout& ctx_drop, float* xchg, unsigned* done,
                       hipStream_t st, float* c_out, const float* h_init, const float* c_init, long seq_st) {
    if (!encoder_persistent_supported(B, H, T) || !xw_table || !xchg || !done || (!h_init != !c_init)) return SF_ERR_UNSUPPORTED;
    EncPersistArgs a{};
    a.w_hh = w_hh; a.b_ih = b_ih; a.b_hh = b_hh; a.xw_table = xw_table; a.seq = seq;
    a.seq_sb = Lpad; a.seq_st = seq_st; a.h_init = h_init; a.c_init = c_init;
    a.lengths = lengths; a.B = B; a.H = H; a.T = T; a.rpg = ceil_div(B, EP_GROUPS);
    a.gates = gates; a.hs = hs; a.cs = cs; a.ctx = ctx; a.ld_ctx = T * H; a.ctx_drop = ctx_drop; a.c_out = c_out;
    a.xchg = reinterpret_cast<unsigned*>(xchg); a.done = done; a.place = done + 4; a.trace = g_trace;
    a.fault = done + EP_FAULT_WORD; a.timeout = g_persist_timeout < 0 ? EP_TIMEOUT_TICKS : g_persist_timeout;
    // (slot 0 of the tapes: zeroed here, or -- given an initial state -- the caller's)
    SF_LAUNCH(enc_persist_prologue_kernel, dim3(96), dim3(256), 0, st, a.xchg, encoder_persistent_xchg_floats(H),
              hs, cs, h_init ? (size_t)0 : (size_t)B * H, a.place, g_force_sc1, a.fault, g_persist_timeout);
    SF_LAUNCH(enc_persist_kernel, dim3(EP_GROUPS * EP_SLOTS), dim3(256), 0, st, a);
    return launch_status();
}

size_t speaker_persistent_xchg_floats() { return (size_t)EP_GROUPS * 3 * SPX_BUF; }
bool speaker_persistent_supported(int B, int H, int Tp, int vocab) {
    return H == 16 * EP_SLOTS && B >= 1 && B <= EP_GROUPS * EP_ROWS && Tp >= 1 && Tp <= SP_TPMAX && vocab >= 32 &&
           vocab <= 32 * EP_SLOTS && device_cus() >= EP_GROUPS * EP_SLOTS;
}

int speaker_persistent(const float* w_hh, const float* b_ih, const float* b_hh, const float* xw_table,
                       const float* w_out, int ld_wout, const float* w_d2a, const float* b_d2a, int vocab, int ldv,
                       const float* cq, const float* cw, const uint8_t* mask, const float* h_init,
                       const float* c_init, const int64_t* targets, int feedback, int pad, int eos, int B, int H,
                       int Tp, int S, int64_t* words, float* step_scores, float* nll_term, float* live,
                       float* logits, float* alpha, float* h1_tape, float* c1_tape, uint8_t* ended, float* xchg,
                       unsigned* done, hipStream_t st, const sf_sample* sample) {
    if (!speaker_persistent_supported(B, H, Tp, vocab) || !xchg || !done || !xw_table) return SF_ERR_UNSUPPORTED;
    if (feedback == 2 && !sample) return SF_ERR_ARG;
    SpkPersistArgs a{};
    a.w_hh = w_hh; a.b_ih = b_ih; a.b_hh = b_hh; a.xw_table = xw_table; a.w_out = w_out; a.ld_wout = ld_wout;
    a.w_d2a = w_d2a; a.b_d2a = b_d2a; a.vocab = vocab; a.ldv = ldv; a.cq = cq; a.cw = cw; a.mask = mask;
    a.h_init = h_init; a.c_init = c_init; a.targets = targets; a.feedback = feedback; a.pad = pad; a.eos = eos;
    a.B = B; a.H = H; a.Tp = Tp; a.S = S; a.rpg = ceil_div(B, EP_GROUPS); a.words = words;
    a.step_scores = step_scores; a.nll_term = nll_term; a.live = live; a.logits = logits; a.alpha = alpha;
    a.h1_tape = h1_tape; a.c1_tape = c1_tape; a.ended = ended; a.xchg = reinterpret_cast<unsigned*>(xchg);
    a.done = done; a.place = done + 4; a.trace = g_trace;
    a.fault = done + EP_FAULT_WORD; a.timeout = g_persist_timeout < 0 ? EP_TIMEOUT_TICKS : g_persist_timeout;
    a.sample_site = sample && sample->stream_dev ? sample->stream_dev : site_zero();
    if (sample) { a.sample_seed = sample->seed; a.sample_stream = sample->stream; a.sample_row0 = sample->row0; }
    SF_LAUNCH(enc_bwd_persist_prologue_kernel, dim3(256), dim3(256), 0, st, a.xchg, speaker_persistent_xchg_floats(),
              a.place, g_force_sc1, a.fault, g_persist_timeout);
    if (feedback == 2)
        SF_LAUNCH_AS("spk_persist_kernel<sample>", spk_persist_kernel<true>, dim3(EP_GROUPS * EP_SLOTS), dim3(256), 0, st, a);
    else
        SF_LAUNCH_AS("spk_persist_kernel", spk_persist_kernel<false>, dim3(EP_GROUPS * EP_SLOTS), dim3(256), 0, st, a);
    return launch_status();
}

size_t encoder_bwd_persistent_xchg_floats() { return (size_t)EP_GROUPS * 2 * EP_SLOTS * EP_SLOTS * 256; }

int encoder_bwd_persistent(const float* w_hh, const int* lengths, int B, int H, int T, const float* gates,
                           const float* cs, const float* dctx, const Dropout& ctx_drop, const float* dh_in,
                           const float* dc_in, float* dgates, float* xchg, unsigned* done, hipStream_t st,
                           long dctx_sb, long dctx_st, float* dc0_out) {
    if (!encoder_persistent_supported(B, H, T) || !xchg || !done) return SF_ERR_UNSUPPORTED;
    EncBwdPersistArgs a{};
    a.w_hh = w_hh; a.lengths = lengths; a.B = B; a.H = H; a.T = T; a.rpg = ceil_div(B, EP_GROUPS);
    a.gates = gates; a.cs = cs; a.dctx = dctx; a.ctx_drop = ctx_drop; a.dh_in = dh_in; a.dc_in = dc_in;
    a.dctx_sb = dctx_sb ? dctx_sb : (long)T * H; a.dctx_st = dctx_st ? dctx_st : (long)H; a.dc0_out = dc0_out;
    a.dgates = dgates; a.xchg = reinterpret_cast<unsigned*>(xchg); a.done = done; a.place = done + 4; a.trace = g_trace;
    a.fault = done + EP_FAULT_WORD; a.timeout = g_persist_timeout < 0 ? EP_TIMEOUT_TICKS : g_persist_timeout;
    SF_LAUNCH(enc_bwd_persist_prologue_kernel, dim3(512), dim3(256), 0, st, a.xchg, encoder_bwd_persistent_xchg_floats(),
              a.place, g_force_sc1, a.fault, g_persist_timeout);
    SF_LAUNCH(enc_bwd_persist_kernel, dim3(EP_GROUPS * EP_SLOTS), dim3(256), 0, st, a);
    return launch_status();
}

}  // namespace sf
