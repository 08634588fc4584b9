// The follower's decode loop as ONE persistent launch (work in progress, built in milestones; see
// DESIGN.md).  Milestone 1 (this file today): the LSTMCell of every decode step -- the gate product
// [B, 2F+H] x [4H, 2F+H]^T (model.py:393), split-K partial tiles handed to the workgroup that owns the
// cell, the cell update, and h fed back as the next step's operand -- with the other two thirds of the
// operand (u_prev | attended feature) still read from a reference tape.
//
// Decomposition (256 workgroups x 512 threads, one per CU; workgroup b: XCD x = b % 8, slot c = b / 8):
//   gate product  all 256 CUs: n-tile = hidden units [16c, +16) x 4 gates (64 columns), K split 8 ways by
//                 XCD (stage s of 64 k belongs to split s % 8: every split owns one h stage and 4-5 stages
//                 of each of the other two segments), all <= 128 rows.  W streams from L2 / MALL through
//                 LDS; the A operand comes from the exchange buffer XIN (sentinel-tagged: a stage is
//                 re-read until complete).
//   partial tiles [16 rows x 64] per (row group, n-tile, split) go to the SLAB region of the workgroup
//                 (XCD = row group, slot = n-tile) that owns the cell update of those 16 rows x 16 units;
//                 that workgroup sums its 8 partials, updates the cell, resets the region (two buffers
//                 suffice: the consumer owns it) and publishes h into XIN for the next step.
// Row groups are the MFMA m-tiles: 16 rows each, ceil(B / 16) <= 8 groups.
#include "sf_kernels.h"
#include "sf_gemm_small.h"

namespace sf {
namespace {

typedef unsigned v4u __attribute__((ext_vector_type(4)));

constexpr int MG_SLOTS = 32, MG_XCD = 8, MG_ROWS = 16;
constexpr int MG_H = 512, MG_F = 2176;
constexpr int MG_K = 2 * MG_F + MG_H;                 // 4864
constexpr int MG_BK = 64, MG_LD = MG_BK + 8;          // stage depth, LDS row stride (conflict-free b128 reads)
constexpr int MG_NU = MG_F / MG_BK, MG_NH = MG_H / MG_BK;      // 34 stages per input half, 8 of h
constexpr int MG_STAGES = 2 * MG_NU + MG_NH;          // 76
constexpr unsigned MG_SENT = 0xFFFFFFFFu;
constexpr long long MG_TIMEOUT = 25000000LL;          // 0.25 s of the 100 MHz wall clock
constexpr int MG_SC1 = 16;

struct MegaArgs {
    const float* w_ih; const float* w_hh; const float* b_ih; const float* b_hh;   // [4H,2F], [4H,H], [4H] x2
    const float* h_init; const float* c_init;           // [B,H]
    const float* xin_ref;                               // [S+1,B,2F] reference operand tape (u | feat), milestone 1
    int B, S, MT;                                       // MT = row groups
    float* h1_tape; float* c1_tape; float* gates_tape;  // [S,B,H], [S,B,H], [S,B,4H] or null
    unsigned* xin;                                      // [3][128][MG_K] dwords (u | feat | h)
    unsigned* slab;                                     // [2][8 groups][32 nt][8 sp][1024] dwords
    unsigned* done;
    unsigned* lock;                                     // persist_lock_addr()
};

__global__ __launch_bounds__(256) void mega_prologue_kernel(unsigned* xin, size_t n_xin, unsigned* slab, size_t n_slab,
                                                            unsigned* lock) {
    const size_t stride = (size_t)gridDim.x * blockDim.x, i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (size_t i = i0; i < n_xin; i += stride) xin[i] = MG_SENT;
    for (size_t i = i0; i < n_slab; i += stride) slab[i] = MG_SENT;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const long long t0 = wall_clock64();
        while (atomicCAS(lock, 0u, 1u) != 0u) {
            __builtin_amdgcn_s_sleep(32);
            if (wall_clock64() - t0 > 8 * MG_TIMEOUT) break;
        }
    }
}
// h of step 0 into XIN[0] (after the sentinel fill: a second tiny launch keeps the order trivial)
__global__ __launch_bounds__(256) void mega_seed_kernel(unsigned* xin, const float* h_init, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 128 * MG_H) return;
    const int row = i / MG_H, j = i - row * MG_H;
    xin[(size_t)row * MG_K + 2 * MG_F + j] = row < B ? __float_as_uint(h_init[(size_t)row * MG_H + j]) : 0u;
}

template <int MT>
__global__ __launch_bounds__(512) void mega_kernel(MegaArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int AROWS = MT * 16, WROWS = 64;
    constexpr int BUF = (AROWS + WROWS) * MG_LD;
    constexpr int APASS = (MT + 1) / 2;                  // 32 rows x 16 float4 per staging pass
    const int tid = threadIdx.x, lane = tid & 63, wave8 = tid >> 6;
    const int gate = wave8 & 3, khalf = wave8 >> 2;
    const int li = lane & 15, kk = lane >> 4;
    const int xcd = blockIdx.x & (MG_XCD - 1), slot = blockIdx.x >> 3;
    const int B = p.B, S = p.S;
    const int ldrow = tid >> 4, ldc4 = tid & 15;
    const size_t BH = (size_t)B * MG_H;
    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(p.xin, 0, 3 * 128 * MG_K * 4, 0x00020000);
    const auto rs_s = __builtin_amdgcn_make_buffer_rsrc(p.slab, 0, 2 * MG_XCD * MG_SLOTS * 8 * 1024 * 4, 0x00020000);
    bool dead = false;

    // ---- cell ownership: (row group = xcd, units [16 slot, +16)); threads 0..255 own one element
    const bool cell_wg = xcd < MT;
    const int er = (tid >> 4) & 15, eu = tid & 15;
    const int eb = xcd * 16 + er;
    const bool evalid = cell_wg && tid < 256 && eb < B;
    const int ebc = min(eb, B - 1);
    const int ej = 16 * slot + eu;
    float bias[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bias[g] = p.b_ih[g * MG_H + ej] + p.b_hh[g * MG_H + ej];
    float c_state = p.c_init[(size_t)ebc * MG_H + ej];

    // ---- this workgroup's stages of the gate product: the h stage, then the feature half, then u
    // (computed, not tabulated: a register array filled through a running index compiles to movrel writes
    //  that the compiler also issues speculatively one past the end)
    const int sh = 2 * MG_NU + ((xcd - 2 * MG_NU) & 7);
    const int f0 = MG_NU + ((xcd - MG_NU) & 7), nf = (2 * MG_NU - f0 + 7) >> 3;
    const int nu = (MG_NU - xcd + 7) >> 3;
    const int nst = 1 + nf + nu;
    auto stage_of = [&](int i) { return i == 0 ? sh : (i <= nf ? f0 + 8 * (i - 1) : xcd + 8 * (i - 1 - nf)); };

    for (int t = 0; t < S; ++t) {
        const unsigned xb = (unsigned)((t % 3) * 128 * MG_K);            // XIN buffer of this step (dwords)
        // ============================ gate product =================================================
        f32x4 acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        struct Regs { v4u a[APASS]; float4 w[2]; };
        auto issue = [&](Regs& r, int s) {
            // A: XIN (h) or the reference tape (u | feat, milestone 1); W: the 64 gate-interleaved rows
            const bool is_h = s >= 2 * MG_NU;
            const int k0 = is_h ? (s - 2 * MG_NU) * MG_BK : s * MG_BK;
#pragma unroll
            for (int pp = 0; pp < APASS; ++pp) {
                const int row = pp * 32 + ldrow;
                if (is_h) {
                    r.a[pp] = __builtin_amdgcn_raw_buffer_load_b128(
                        rs_x, (xb + (unsigned)(min(row, 127) * MG_K + 2 * MG_F + k0 + 4 * ldc4)) * 4u, 0, MG_SC1);
                } else {
                    const float4 v = ld4(p.xin_ref + ((size_t)t * B + min(row, B - 1)) * 2 * MG_F + k0 + 4 * ldc4);
                    r.a[pp] = v4u{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
                }
            }
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                const int nl = pp * 32 + ldrow;
                const int wrow = (nl >> 4) * MG_H + 16 * slot + (nl & 15);
                r.w[pp] = is_h ? ld4(p.w_hh + (size_t)wrow * MG_H + k0 + 4 * ldc4)
                               : ld4(p.w_ih + (size_t)wrow * 2 * MG_F + k0 + 4 * ldc4);
            }
        };
        auto settle = [&](Regs& r, int s) {                 // re-read operand pieces that are not published yet
            if (s < 2 * MG_NU) return;                     // (milestone 1: only h comes through XIN)
            const int k0 = (s - 2 * MG_NU) * MG_BK;
            const long long t0 = wall_clock64();
#pragma unroll
            for (int pp = 0; pp < APASS; ++pp) {
                const int row = pp * 32 + ldrow;
                if (row >= AROWS) continue;                // rows beyond the last group are never published
                const unsigned off = (xb + (unsigned)(min(row, 127) * MG_K + 2 * MG_F + k0 + 4 * ldc4)) * 4u;
                while (!dead && (r.a[pp].x == MG_SENT || r.a[pp].y == MG_SENT || r.a[pp].z == MG_SENT || r.a[pp].w == MG_SENT)) {
                    asm volatile("" ::: "memory");
                    r.a[pp] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, MG_SC1);
                    if (wall_clock64() - t0 > MG_TIMEOUT) dead = true;
                }
            }
        };
        auto lstore = [&](const Regs& r, int buf) {
            float* As = smem + buf * BUF;
            float* Ws = As + AROWS * MG_LD;
#pragma unroll
            for (int pp = 0; pp < APASS; ++pp)
                if (pp * 32 + ldrow < AROWS)
                    *reinterpret_cast<v4u*>(As + (pp * 32 + ldrow) * MG_LD + 4 * ldc4) = r.a[pp];
#pragma unroll
            for (int pp = 0; pp < 2; ++pp)
                *reinterpret_cast<float4*>(Ws + (pp * 32 + ldrow) * MG_LD + 4 * ldc4) = r.w[pp];
        };
        auto compute = [&](int buf) {
            const float* As = smem + buf * BUF;
            const float* Ws = As + AROWS * MG_LD + (gate * 16 + li) * MG_LD;
#pragma unroll
            for (int cc = 0; cc < MG_BK / 32; ++cc) {
                const int c = khalf * (MG_BK / 32) + cc;
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
        {
            Regs r0, r1;
            issue(r0, stage_of(0));
            settle(r0, stage_of(0));
            lstore(r0, 0);
            __syncthreads();
            for (int i = 0; i < nst; ++i) {
                const bool more = i + 1 < nst;
                if (more) issue(r1, stage_of(i + 1));
                compute(i & 1);
                if (more) {
                    settle(r1, stage_of(i + 1));
                    lstore(r1, (i + 1) & 1);
                }
                __syncthreads();
            }
        }
        // the two K halves meet in LDS; the result goes out as [16 x 64] tiles in MFMA layout, one per row
        // group, into the region of the workgroup that owns that group's cell for these 16 units
        {
            f32x4* red = reinterpret_cast<f32x4*>(smem);
            if (khalf == 1) {
#pragma unroll
                for (int m = 0; m < MT; ++m) red[(gate * MT + m) * 64 + lane] = acc[m];
            }
            __syncthreads();
            if (khalf == 0) {
                const unsigned sbuf = (unsigned)((t & 1) * MG_XCD * MG_SLOTS * 8 * 1024);
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const f32x4 v = acc[m] + red[(gate * MT + m) * 64 + lane];
                    const unsigned off = sbuf + (unsigned)(((m * MG_SLOTS + slot) * 8 + xcd) * 1024 + ((gate * 4 + kk) * 16 + li) * 4);
                    __builtin_amdgcn_raw_buffer_store_b128(
                        v4u{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])},
                        rs_s, off * 4u, 0, MG_SC1);
                }
            }
            __syncthreads();
        }
        // ============================ cell update (owner of row group xcd, units 16 slot..) =========
        if (cell_wg) {
            float pre[4] = {bias[0], bias[1], bias[2], bias[3]};
            if (tid < 256) {
                unsigned v[4][8];
                const unsigned sbuf = (unsigned)((t & 1) * MG_XCD * MG_SLOTS * 8 * 1024);
                const unsigned base = sbuf + (unsigned)((xcd * MG_SLOTS + slot) * 8 * 1024);
                const long long t0 = wall_clock64();
                for (;;) {
                    asm volatile("" ::: "memory");
                    bool ok = true;
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int sp = 0; sp < 8; ++sp)
                            v[g][sp] = __builtin_amdgcn_raw_buffer_load_b32(
                                rs_s, (base + (unsigned)(sp * 1024 + ((g * 4 + (er >> 2)) * 16 + eu) * 4 + (er & 3))) * 4u, 0, MG_SC1);
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
            {
                const unsigned sbuf = (unsigned)((t & 1) * MG_XCD * MG_SLOTS * 8 * 1024);
                const unsigned base = sbuf + (unsigned)((xcd * MG_SLOTS + slot) * 8 * 1024);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    __builtin_amdgcn_raw_buffer_store_b128(v4u{MG_SENT, MG_SENT, MG_SENT, MG_SENT}, rs_s,
                                                           (base + (unsigned)((tid + 512 * i) * 4)) * 4u, 0, MG_SC1);
            }
            if (tid < 256) {
                const float ig = sigmoidf_(pre[0]), fg = sigmoidf_(pre[1]), gg = tanhf(pre[2]), og = sigmoidf_(pre[3]);
                c_state = fg * c_state + ig * gg;
                float h1 = og * tanhf(c_state);
                if (dead) h1 = __uint_as_float(0x7FC00000u);
                // h of the next step: own [16 x 16] patch of XIN[(t+1) % 3] (and the reset of XIN[(t+2) % 3])
                const float hp = eb < B ? h1 : 0.f;
                const float h_1 = __shfl_down(hp, 1), h_2 = __shfl_down(hp, 2), h_3 = __shfl_down(hp, 3);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if ((tid & 3) == 0) {
                    const unsigned po = (unsigned)(eb * MG_K + 2 * MG_F + ej);
                    __builtin_amdgcn_raw_buffer_store_b128(
                        v4u{__float_as_uint(hp), __float_as_uint(h_1), __float_as_uint(h_2), __float_as_uint(h_3)}, rs_x,
                        ((unsigned)(((t + 1) % 3) * 128 * MG_K) + po) * 4u, 0, MG_SC1);
                    __builtin_amdgcn_raw_buffer_store_b128(v4u{MG_SENT, MG_SENT, MG_SENT, MG_SENT}, rs_x,
                                                           ((unsigned)(((t + 2) % 3) * 128 * MG_K) + po) * 4u, 0, MG_SC1);
                }
                if (evalid) {
                    p.h1_tape[(size_t)t * BH + (size_t)eb * MG_H + ej] = h1;
                    p.c1_tape[(size_t)t * BH + (size_t)eb * MG_H + ej] = c_state;
                    if (p.gates_tape) {
                        float* gp = p.gates_tape + ((size_t)t * B + eb) * 4 * MG_H + ej;
                        gp[0] = ig; gp[MG_H] = fg; gp[2 * MG_H] = gg; gp[3 * MG_H] = og;
                    }
                }
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        const unsigned n = atomicAdd(p.done, 1u);
        if (n == gridDim.x - 1) {
            atomicExch(p.done, 0u);
            atomicExch(p.lock, 0u);
        }
    }
}

}  // namespace

size_t mega_xin_dwords() { return (size_t)3 * 128 * MG_K; }
size_t mega_slab_dwords() { return (size_t)2 * MG_XCD * MG_SLOTS * 8 * 1024; }

int mega_lstm_loop(const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, const float* h_init,
                   const float* c_init, const float* xin_ref, int B, int S, float* h1_tape, float* c1_tape,
                   float* gates_tape, float* ws_xin, float* ws_slab, unsigned* done, hipStream_t st) {
    if (B < 1 || B > 128 || S < 1) return SF_ERR_UNSUPPORTED;
    MegaArgs a{};
    a.w_ih = w_ih; a.w_hh = w_hh; a.b_ih = b_ih; a.b_hh = b_hh; a.h_init = h_init; a.c_init = c_init;
    a.xin_ref = xin_ref; a.B = B; a.S = S; a.MT = ceil_div(B, 16); a.h1_tape = h1_tape; a.c1_tape = c1_tape;
    a.gates_tape = gates_tape; a.xin = reinterpret_cast<unsigned*>(ws_xin); a.slab = reinterpret_cast<unsigned*>(ws_slab);
    a.done = done; a.lock = persist_lock_addr();
    if (!a.lock) return SF_ERR_LAUNCH;
    SF_LAUNCH(mega_prologue_kernel, dim3(1024), dim3(256), 0, st, a.xin, mega_xin_dwords(), a.slab, mega_slab_dwords(),
              a.lock);
    SF_LAUNCH(mega_seed_kernel, dim3(128 * MG_H / 256), dim3(256), 0, st, a.xin, h_init, B);
    const dim3 grid(MG_XCD * MG_SLOTS), block(512);
#define SF_MEGA(MTV)                                                                                          \
    case MTV: {                                                                                               \
        const size_t lds = (size_t)2 * (MTV * 16 + 64) * MG_LD * sizeof(float);                               \
        static bool attr_set = false;                                                                         \
        if (!attr_set) {                                                                                      \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mega_kernel<MTV>),                        \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                \
            attr_set = true;                                                                                  \
        }                                                                                                     \
        SF_LAUNCH(mega_kernel<MTV>, grid, block, lds, st, a);                                                 \
    } break;
    switch (a.MT) {
        SF_MEGA(1) SF_MEGA(2) SF_MEGA(3) SF_MEGA(4) SF_MEGA(5) SF_MEGA(6) SF_MEGA(7) SF_MEGA(8)
        default: return SF_ERR_UNSUPPORTED;
    }
#undef SF_MEGA
    return launch_status();
}

}  // namespace sf
