// The three soft attentions of the speaker/follower path, forward and backward.
// One workgroup per sample, row set resident in registers (see sf_rows.h).
#include "sf_kernels.h"
#include "sf_rows.h"
#include "sf_glue.h"
#include "sf_gemm_small.h"

namespace sf {

namespace {

// =================================================================================================
// Visual attention (model.py:310-326), V rows of F floats; CPL = 9 covers F <= 2304.
// MODE 0 (forward):  w_v = softmax_v(x_v . vec)             out = sum_v w_v x_v   (vec = q)
// MODE 1 (backward): d_v = x_v . vec, w_v = alpha_v (d_v - sum_u alpha_u d_u)      (vec = dout)
//                    out = dq = sum_v w_v x_v
// =================================================================================================
constexpr int VIS_CPL = 9, VIS_RPW = 3, VIS_NW = 12, VIS_SLOTS = 6;

struct VisArgs {
    PanoSrc src;
    const float* vec;      // [B, ldvec]
    int ldvec;
    float* alpha;          // fwd: out [B,V]; bwd: in
    float* out;            // [B, ldo]
    int ldo;
    Dropout drop;          // fwd: applied to out; bwd: applied to vec (same mask)
    int drop_col0;
    const double* vec64;   // the query in float64 (visual_split_body<0, true>: scores accumulated in float64), or null
    int vec_slabs;         // MODE 1: `vec` is the sum of this many K-split slabs (0 / 1: a plain vector) ...
    long vec_slab_stride;  // ... `vec_slab_stride` floats apart: the workgroup adds them up itself (no reduce launch)
};

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
    return v;
}
__device__ __forceinline__ double wave_max_f64(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, WAVE));
    return v;
}

template <int MODE>
__device__ __forceinline__ void visual_attn_body(const VisArgs& a, int b) {
    __shared__ float4 slots[VIS_SLOTS][VIS_CPL * 64];
    __shared__ float s_score[64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int V = a.src.V;
    const int n4 = (a.src.IMG + a.src.LOC) >> 2;

    const PanoRow prow = pano_row(a.src, b);
    float4 x[VIS_RPW][VIS_CPL];
#pragma unroll
    for (int r = 0; r < VIS_RPW; ++r) {
        const int v = wave * VIS_RPW + r;
#pragma unroll
        for (int i = 0; i < VIS_CPL; ++i) {
            const int c = lane + 64 * i;
            x[r][i] = pano_load(prow, v, c, v < V && c < n4, V, n4);
        }
    }

    const uint32_t rkey = drop_key(a.drop, (uint32_t)(a.drop.row0 + b));
    float dot[VIS_RPW];
#pragma unroll
    for (int r = 0; r < VIS_RPW; ++r) dot[r] = 0.f;
    // (all chunks of the vector requested before the first use: behind the dropout branch below each load
    //  would otherwise be its own memory round trip)
    float4 qv[VIS_CPL];
    if (MODE == 1 && a.vec_slabs > 1) {
        // the vector arrives as K-split slabs of the product that formed it (the LSTM's data gradient): thread c adds
        // up float4 c of this row in slab order (the order reduce_slabs_kernel uses: same bits) into LDS, once per
        // workgroup, while the panorama rows are still on their way
        __shared__ float4 s_vec[VIS_CPL * 64];
        const int c = threadIdx.x;
        if (c < n4) {
            const float4* p0 = reinterpret_cast<const float4*>(a.vec + (size_t)b * a.ldvec) + c;
            float4 t = *p0;
            for (int sl = 1; sl < a.vec_slabs; ++sl) {
                const float4 u = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p0) + (size_t)sl * a.vec_slab_stride);
                t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
            }
            s_vec[c] = t;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < VIS_CPL; ++i) qv[i] = s_vec[min(lane + 64 * i, n4 - 1)];
    } else {
#pragma unroll
    for (int i = 0; i < VIS_CPL; ++i)
        qv[i] = reinterpret_cast<const float4*>(a.vec + (size_t)b * a.ldvec)[min(lane + 64 * i, n4 - 1)];
    }
#pragma unroll
    for (int i = 0; i < VIS_CPL; ++i) {
        const int c = lane + 64 * i;
        float4 q = qv[i];
        if (c >= n4) q = f4zero();
        {
            if (MODE == 1 && a.drop.on()) {
                const uint32_t col = (uint32_t)(a.drop_col0 + 4 * c);
                q.x = dropout_keep(rkey, col + 0, a.drop.thresh) ? q.x * a.drop.scale : 0.f;
                q.y = dropout_keep(rkey, col + 1, a.drop.thresh) ? q.y * a.drop.scale : 0.f;
                q.z = dropout_keep(rkey, col + 2, a.drop.thresh) ? q.z * a.drop.scale : 0.f;
                q.w = dropout_keep(rkey, col + 3, a.drop.thresh) ? q.w * a.drop.scale : 0.f;
            }
        }
#pragma unroll
        for (int r = 0; r < VIS_RPW; ++r) dot[r] += dot4(x[r][i], q);
    }
#pragma unroll
    for (int r = 0; r < VIS_RPW; ++r) {
        const float s = wave_sum(dot[r]);
        const int v = wave * VIS_RPW + r;
        if (lane == 0 && v < V) s_score[v] = s;
    }
    __syncthreads();

    // every wave redoes the V-wide softmax with lane v holding score v (V <= 64)
    const float s = lane < V ? s_score[lane] : -INFINITY;
    float w;
    if (MODE == 0) {
        const float m = wave_max(s);
        const float e = lane < V ? expf(s - m) : 0.f;
        w = e / wave_sum(e);
        if (wave == 0 && lane < V) a.alpha[(size_t)b * V + lane] = w;
    } else {
        const float alv = a.alpha[(size_t)b * V + min(lane, V - 1)];
        const float al = lane < V ? alv : 0.f;
        const float d = lane < V ? s : 0.f;
        w = al * (d - wave_sum(al * d));
    }

    float4 p[VIS_CPL];
#pragma unroll
    for (int i = 0; i < VIS_CPL; ++i) p[i] = f4zero();
#pragma unroll
    for (int r = 0; r < VIS_RPW; ++r) {
        const int v = wave * VIS_RPW + r;
        const float wr = __shfl(w, v < V ? v : 0, WAVE);
        if (v < V) {
#pragma unroll
            for (int i = 0; i < VIS_CPL; ++i) f4fma(p[i], wr, x[r][i]);
        }
    }

    float* orow = a.out + (size_t)b * a.ldo;
    const Dropout dr = a.drop;
    const int col0 = a.drop_col0;
    block_row_sum<VIS_CPL, VIS_NW, VIS_SLOTS>(p, slots, n4, [&](int c, float4 t) {
        if (MODE == 0 && dr.on()) {
            const uint32_t col = (uint32_t)(col0 + 4 * c);
            t.x = dropout_keep(rkey, col + 0, dr.thresh) ? t.x * dr.scale : 0.f;
            t.y = dropout_keep(rkey, col + 1, dr.thresh) ? t.y * dr.scale : 0.f;
            t.z = dropout_keep(rkey, col + 2, dr.thresh) ? t.z * dr.scale : 0.f;
            t.w = dropout_keep(rkey, col + 3, dr.thresh) ? t.w * dr.scale : 0.f;
        }
        reinterpret_cast<float4*>(orow)[c] = t;
    });
}

template <int MODE>
__global__ __launch_bounds__(VIS_NW * 64) void visual_attn_kernel(VisArgs a) {
    visual_attn_body<MODE>(a, blockIdx.x);
}

// -------------------------------------------------------------------------------------------------
// Forward visual attention split over TWO workgroups per sample.  One workgroup per sample leaves
// 156 of 256 CUs idle at batch 100 and a CU sustains only ~20-35 GB/s of loads, so the 313 KB
// panorama of a sample took ~17 us to arrive; two half-panoramas on two CUs take half of that.
// Each block keeps flash-style partials of ITS 18 views (max m, sum l = sum e^(s-m), unnormalised
// weighted sum P, raw scores) in the workspace; the block that finishes second (ticket from a
// monotonic per-sample counter; partials are published write-through and read back with sc1
// loads, so no cache-wide fence is needed) merges: no spinning, no extra launch.
// -------------------------------------------------------------------------------------------------
#ifndef SF_VIS_GROUPS
#define SF_VIS_GROUPS 2
#endif
constexpr int VSP_G = SF_VIS_GROUPS;                   // workgroups per sample (2 or 4)
constexpr int VSP_NW = VIS_NW / VSP_G, VSP_RPG = VSP_NW * VIS_RPW, VSP_SLOTS = VSP_NW < 3 ? VSP_NW : 3;
static_assert(VSP_G == 2 || VSP_G == 4, "ticket arithmetic assumes a power of two");

struct VisSplit {
    float* part;          // [B][G][F + 64]: P | scores[32] | m, l
    unsigned* counter;    // [B] monotonic tickets (zero before the first launch)
    unsigned long long* trace;   // development aid (sf_debug_trace): [blocks][8] wall-clock stamps
};
__device__ __forceinline__ void vis_stamp(const VisSplit& sp, int block, int slot) {
    if (sp.trace && threadIdx.x == 0) sp.trace[(size_t)block * 8 + slot] = wall_clock64();
}

// PHASE 0: partials + ticket + merge by the last arriver, in one launch (hand-off inside the launch:
//          write-through stores, sc1 loads).
// PHASE 1: partials only, PHASE 2: merge only -- the same two halves as two consecutive launches
//          (plain stores / loads: the kernel boundary is the hand-off).  The pipelined decode step
//          runs them beside two different small products, so the visual attention of step t+1 never
//          holds up the text / scoring chain of step t.
// F64 (PHASE 0 only; the speaker's path encoder, csrc/sf_precise.hip): the query arrives in float64 and every score is
//          accumulated in float64; a group's record holds its scores RELATIVE to its own maximum (small numbers:
//          exact in fp32 to 1e-7 of the softmax weight, where a raw score of +-80 would carry 4e-6) and that maximum as a
//          float64 in two dwords.
template <int PHASE, bool F64 = false>
__device__ __forceinline__ void visual_split_body(const VisArgs& a, const VisSplit& sp, int g, int b) {
    __shared__ float4 slots[VSP_SLOTS][VIS_CPL * 64];
    __shared__ float s_score[64];
    __shared__ double s_score64[F64 ? 32 : 1];
    __shared__ int s_last;
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int V = a.src.V;
    const int F = a.src.IMG + a.src.LOC, n4 = F >> 2;
    const int pstride = F + 64;
    float* rec = sp.part + ((size_t)b * VSP_G + g) * pstride;

    if (PHASE != 2) {
    vis_stamp(sp, b * VSP_G + g, 0);
    const PanoRow prow = pano_row(a.src, b);
    float4 x[VIS_RPW][VIS_CPL];
#pragma unroll
    for (int r = 0; r < VIS_RPW; ++r) {
        const int v = g * VSP_RPG + wave * VIS_RPW + r;
#pragma unroll
        for (int i = 0; i < VIS_CPL; ++i) {
            const int c = lane + 64 * i;
            x[r][i] = pano_load(prow, v, c, v < V && c < n4, V, n4);
        }
    }
    float s, m, e;          // F64: s = score - m (relative), m unused (md holds the maximum)
    double md = 0.0;
    if (F64) {
        double dot[VIS_RPW];
#pragma unroll
        for (int r = 0; r < VIS_RPW; ++r) dot[r] = 0.0;
        const double* qrow = a.vec64 + (size_t)b * a.ldvec;
#pragma unroll
        for (int i = 0; i < VIS_CPL; ++i) {
            const int c = lane + 64 * i, cc = min(c, n4 - 1);
            const double2 q0 = reinterpret_cast<const double2*>(qrow)[2 * cc], q1 = reinterpret_cast<const double2*>(qrow)[2 * cc + 1];
            const double k = c < n4 ? 1.0 : 0.0;
#pragma unroll
            for (int r = 0; r < VIS_RPW; ++r) {
                double t = (double)x[r][i].x * q0.x;
                t = fma((double)x[r][i].y, q0.y, t);
                t = fma((double)x[r][i].z, q1.x, t);
                t = fma((double)x[r][i].w, q1.y, t);
                dot[r] = fma(t, k, dot[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < VIS_RPW; ++r) {
            const double sd = wave_sum_f64(dot[r]);
            const int vl = wave * VIS_RPW + r;
            if (lane == 0) s_score64[vl] = (g * VSP_RPG + vl < V) ? sd : -INFINITY;
        }
        __syncthreads();
        vis_stamp(sp, b * VSP_G + g, 1);
        const double sd = lane < VSP_RPG ? s_score64[lane] : -INFINITY;
        md = wave_max_f64(sd);
        s = sd > -INFINITY ? (float)(sd - md) : -INFINITY;
        m = 0.f;
        e = sd > -INFINITY ? expf(s) : 0.f;
    } else {
    float dot[VIS_RPW];
#pragma unroll
    for (int r = 0; r < VIS_RPW; ++r) dot[r] = 0.f;
#pragma unroll
    for (int i = 0; i < VIS_CPL; ++i) {
        const int c = lane + 64 * i;
        const float4 q = c < n4 ? reinterpret_cast<const float4*>(a.vec + (size_t)b * a.ldvec)[c]
                                : f4zero();
#pragma unroll
        for (int r = 0; r < VIS_RPW; ++r) dot[r] += dot4(x[r][i], q);
    }
#pragma unroll
    for (int r = 0; r < VIS_RPW; ++r) {
        const float sw = wave_sum(dot[r]);
        const int vl = wave * VIS_RPW + r;
        if (lane == 0) s_score[vl] = (g * VSP_RPG + vl < V) ? sw : -INFINITY;
    }
    __syncthreads();
    vis_stamp(sp, b * VSP_G + g, 1);
    s = lane < VSP_RPG ? s_score[lane] : -INFINITY;
    m = wave_max(s);
    e = s > -INFINITY ? expf(s - m) : 0.f;
    }
    const float l = wave_sum(e);

    float4 p[VIS_CPL];
#pragma unroll
    for (int i = 0; i < VIS_CPL; ++i) p[i] = f4zero();
#pragma unroll
    for (int r = 0; r < VIS_RPW; ++r) {
        const float er = __shfl(e, wave * VIS_RPW + r, WAVE);
#pragma unroll
        for (int i = 0; i < VIS_CPL; ++i) f4fma(p[i], er, x[r][i]);
    }
    // publish the partials WRITE-THROUGH (sc1: straight to the memory side, visible to every XCD),
    // drain, then draw a ticket: no release fence (a buffer_wbl2 of the whole L2 costs far more
    // than this kernel), and the merging block reads them back with sc1 loads: no acquire fence.
    if (PHASE == 1) {                                            // next launch reads them
        block_row_sum<VIS_CPL, VSP_NW, VSP_SLOTS>(p, slots, n4, [&](int c, float4 t) {
            reinterpret_cast<float4*>(rec)[c] = t;
        });
        if (wave == 0) {
            if (lane < 32) rec[F + lane] = s;
            if (lane == 0) {
                rec[F + 32] = m;
                rec[F + 33] = l;
            }
        }
        vis_stamp(sp, b * VSP_G + g, 2);
        return;
    }
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(rec, 0, pstride * 4, 0x00020000);
    block_row_sum<VIS_CPL, VSP_NW, VSP_SLOTS>(p, slots, n4, [&](int c, float4 t) {
        const v4u v{__float_as_uint(t.x), __float_as_uint(t.y), __float_as_uint(t.z), __float_as_uint(t.w)};
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, c * 16, 0, 16);
    });
    if (wave == 0) {
        if (lane < 32) __hip_atomic_store(rec + F + lane, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane == 0) {
            __hip_atomic_store(rec + F + 32, F64 ? __int_as_float(__double2hiint(md)) : m, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(rec + F + 33, l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (F64)
                __hip_atomic_store(rec + F + 34, __int_as_float(__double2loint(md)), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // EVERY storing wave drains
    __syncthreads();
    if (tid == 0)
        s_last = (__hip_atomic_fetch_add(sp.counter + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) &
                  (unsigned)(VSP_G - 1)) == (unsigned)(VSP_G - 1);
    __syncthreads();
    if (!s_last) return;
    }   // PHASE != 2

    float* r0 = sp.part + (size_t)b * VSP_G * pstride;
    auto ldf = [&](float* q) {
        return PHASE == 0 ? __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *q;
    };
    float mg[VSP_G], lg[VSP_G];
    float dm[VSP_G];                                             // F64: m_k - M formed in float64
#pragma unroll
    for (int k = 0; k < VSP_G; ++k) {
        mg[k] = ldf(r0 + (size_t)k * pstride + F + 32);
        lg[k] = ldf(r0 + (size_t)k * pstride + F + 33);
    }
    float M = mg[0];
    if (F64) {
        double m64[VSP_G];
#pragma unroll
        for (int k = 0; k < VSP_G; ++k)
            m64[k] = __hiloint2double(__float_as_int(mg[k]), __float_as_int(ldf(r0 + (size_t)k * pstride + F + 34)));
        double M64 = m64[0];
#pragma unroll
        for (int k = 1; k < VSP_G; ++k) M64 = fmax(M64, m64[k]);
#pragma unroll
        for (int k = 0; k < VSP_G; ++k) dm[k] = (float)(m64[k] - M64);       // 0 for the maximal group, -inf for an empty one
    } else {
#pragma unroll
        for (int k = 1; k < VSP_G; ++k) M = fmaxf(M, mg[k]);
#pragma unroll
        for (int k = 0; k < VSP_G; ++k) dm[k] = mg[k] - M;
    }
    float kk[VSP_G], L = 0.f;                                    // m = -inf (empty group): c = 0, l = 0
#pragma unroll
    for (int k = 0; k < VSP_G; ++k) {
        kk[k] = expf(dm[k]);
        L += lg[k] * kk[k];
    }
    const float inv = 1.0f / L;
    if (tid < V) {
        const int gk = tid / VSP_RPG;
        const float sc = ldf(r0 + (size_t)gk * pstride + F + (tid - gk * VSP_RPG));
        // (F64: sc is relative to its group's maximum: sc + dm = score - M, both parts small and exact)
        a.alpha[(size_t)b * V + tid] = (F64 ? expf(sc + dm[gk]) : expf(sc - M)) * inv;
    }
#pragma unroll
    for (int k = 0; k < VSP_G; ++k) kk[k] *= inv;
    float* orow = a.out + (size_t)b * a.ldo;
    const Dropout dr = a.drop;
    const uint32_t rkey = drop_key(dr, (uint32_t)(dr.row0 + b));
    const auto rs0 = __builtin_amdgcn_make_buffer_rsrc(r0, 0, VSP_G * pstride * 4, 0x00020000);
    for (int c = tid; c < n4; c += VSP_NW * 64) {
        v4u pk[VSP_G];
#pragma unroll
        for (int k = 0; k < VSP_G; ++k)
            pk[k] = __builtin_amdgcn_raw_buffer_load_b128(rs0, k * pstride * 4 + c * 16, 0, PHASE == 0 ? 16 : 0);
        float4 t = f4zero();
#pragma unroll
        for (int k = 0; k < VSP_G; ++k) {
            t.x += kk[k] * __uint_as_float(pk[k].x);
            t.y += kk[k] * __uint_as_float(pk[k].y);
            t.z += kk[k] * __uint_as_float(pk[k].z);
            t.w += kk[k] * __uint_as_float(pk[k].w);
        }
        if (dr.on()) {
            const uint32_t col = (uint32_t)(a.drop_col0 + 4 * c);
            t.x = dropout_keep(rkey, col + 0, dr.thresh) ? t.x * dr.scale : 0.f;
            t.y = dropout_keep(rkey, col + 1, dr.thresh) ? t.y * dr.scale : 0.f;
            t.z = dropout_keep(rkey, col + 2, dr.thresh) ? t.z * dr.scale : 0.f;
            t.w = dropout_keep(rkey, col + 3, dr.thresh) ? t.w * dr.scale : 0.f;
        }
        reinterpret_cast<float4*>(orow)[c] = t;
    }
}

__global__ __launch_bounds__(VSP_NW * 64) void visual_attn_split_kernel(VisArgs a, VisSplit sp) {
    visual_split_body<0>(a, sp, blockIdx.x, blockIdx.y);
}
__global__ __launch_bounds__(VSP_NW * 64) void visual_attn_split_f64_kernel(VisArgs a, VisSplit sp) {
    visual_split_body<0, true>(a, sp, blockIdx.x, blockIdx.y);
}

// =================================================================================================
// Text / path-context attention core (model.py:129-139): L rows of H floats, CPL = 2 (H <= 512).
// forward : s_l = ctx_l . t (masked -> -inf), alpha = softmax, wc = sum alpha_l ctx_l
// backward: d_l = ctx_l . dwc, ds_l = alpha_l (d_l - sum alpha d), dt = sum ds_l ctx_l,
//           dctx_l += alpha_l dwc + ds_l t
// =================================================================================================
constexpr int TXT_CPL = 2, TXT_NW = 16, TXT_SLOTS = 8;

struct TxtArgs {
    const float* ctx;      // [B, L, H]
    const uint8_t* mask;   // [B, L] or null
    int L, H;
    const float* vec;      // fwd: t [B, ldvec]; bwd: dwc [B, ldvec]
    int ldvec;
    const float* vec2;     // bwd: t [B, ldvec2]
    int ldvec2;
    float* alpha;          // [B, L]
    float* out;            // fwd: wc [B, ldo]; bwd: dt [B, ldo]
    int ldo;
    float* dctx;           // bwd, accumulated; may be null
    const int32_t* ctx_row;  // fwd: sample b attends over ctx / mask row ctx_row[b] (null = b)
    float* ds_out;         // bwd, optional [B, L]: the score gradients (for a deferred dctx update)
};

template <int RPW, int MODE, int NW = TXT_NW>
__device__ __forceinline__ void text_attn_body(const TxtArgs& a, int b) {
    __shared__ float4 slots[TXT_SLOTS][TXT_CPL * 64];
    __shared__ float s_score[NW * RPW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int L = a.L, n4 = a.H >> 2;
    const int bc = a.ctx_row ? a.ctx_row[b] : b;
    const float4* ctx = reinterpret_cast<const float4*>(a.ctx) + (size_t)bc * L * n4;

    // straight-line loads (clamped indices, value selects): a load inside a branch costs a full
    // memory round trip each (see sf_rows.h)
    float4 x[RPW][TXT_CPL];
    uint8_t mk[RPW];
    // the mask bytes are loaded UNCONDITIONALLY (from the context itself when there is no mask): a load
    // behind even a block-uniform branch is followed by a full `s_waitcnt vmcnt(0)`, which made every
    // context row its own memory round trip
    const bool use_mask = MODE == 0 && a.mask != nullptr;
    const uint8_t* mrow = use_mask ? a.mask + (size_t)bc * L : reinterpret_cast<const uint8_t*>(ctx);
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int l = wave * RPW + r;
        const int lc = min(l, L - 1);
#pragma unroll
        for (int i = 0; i < TXT_CPL; ++i) {
            const int c = lane + 64 * i;
            const float4 t = ctx[(size_t)lc * n4 + min(c, n4 - 1)];
            x[r][i] = (l < L && c < n4) ? t : f4zero();
        }
        mk[r] = mrow[lc];
    }
    float4 v1[TXT_CPL], v2[TXT_CPL];
#pragma unroll
    for (int i = 0; i < TXT_CPL; ++i) {
        const int c = lane + 64 * i;
        const int cc = min(c, n4 - 1);
        const float4 t1 = reinterpret_cast<const float4*>(a.vec + (size_t)b * a.ldvec)[cc];
        v1[i] = c < n4 ? t1 : f4zero();
        v2[i] = f4zero();
        if (MODE == 1) {
            const float4 t2 = reinterpret_cast<const float4*>(a.vec2 + (size_t)b * a.ldvec2)[cc];
            v2[i] = c < n4 ? t2 : f4zero();
        }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        float d = 0.f;
#pragma unroll
        for (int i = 0; i < TXT_CPL; ++i) d += dot4(x[r][i], v1[i]);
        d = wave_sum(d);
        const int l = wave * RPW + r;
        if (lane == 0 && l < L) s_score[l] = (use_mask && mk[r]) ? -INFINITY : d;
    }
    __syncthreads();

    // L <= 128: lane holds entries lane and lane + 64
    const int l0 = lane, l1 = lane + 64;
    float w0, w1;
    if (MODE == 0) {
        const float s0 = l0 < L ? s_score[l0] : -INFINITY;
        const float s1 = l1 < L ? s_score[l1] : -INFINITY;
        const float m = wave_max(fmaxf(s0, s1));
        const float e0 = l0 < L ? expf(s0 - m) : 0.f;
        const float e1 = l1 < L ? expf(s1 - m) : 0.f;
        const float inv = 1.0f / wave_sum(e0 + e1);
        w0 = e0 * inv;
        w1 = e1 * inv;
        if (wave == 0) {
            if (l0 < L) a.alpha[(size_t)b * L + l0] = w0;
            if (l1 < L) a.alpha[(size_t)b * L + l1] = w1;
        }
    } else {
        const float t0 = a.alpha[(size_t)b * L + min(l0, L - 1)];
        const float t1 = a.alpha[(size_t)b * L + min(l1, L - 1)];
        const float a0 = l0 < L ? t0 : 0.f;
        const float a1 = l1 < L ? t1 : 0.f;
        const float d0 = l0 < L ? s_score[l0] : 0.f;
        const float d1 = l1 < L ? s_score[l1] : 0.f;
        const float tot = wave_sum(a0 * d0 + a1 * d1);
        w0 = a0 * (d0 - tot);
        w1 = a1 * (d1 - tot);
        if (a.ds_out && wave == 0) {                             // block-uniform
            if (l0 < L) a.ds_out[(size_t)b * L + l0] = w0;
            if (l1 < L) a.ds_out[(size_t)b * L + l1] = w1;
        }
    }

    float4 p[TXT_CPL];
#pragma unroll
    for (int i = 0; i < TXT_CPL; ++i) p[i] = f4zero();
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int l = wave * RPW + r;
        const int src = (l < L ? l : 0) & 63;
        const float lo = __shfl(w0, src, WAVE), hi = __shfl(w1, src, WAVE);
        const float wr = (l < 64) ? lo : hi;
        const float wl = l < L ? wr : 0.f;                       // x is zero beyond L anyway
#pragma unroll
        for (int i = 0; i < TXT_CPL; ++i) f4fma(p[i], wl, x[r][i]);
    }
    if (MODE == 1 && a.dctx) {                                   // dctx_l += alpha_l dwc + ds_l t
        float4 g[RPW][TXT_CPL];
        float al[RPW];
#pragma unroll
        for (int r = 0; r < RPW; ++r) {                          // all read-modify-write loads first
            const int lc = min(wave * RPW + r, L - 1);
            al[r] = a.alpha[(size_t)b * L + lc];
            const float4* drow = reinterpret_cast<const float4*>(a.dctx) + ((size_t)b * L + lc) * n4;
#pragma unroll
            for (int i = 0; i < TXT_CPL; ++i) g[r][i] = drow[min(lane + 64 * i, n4 - 1)];
        }
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int l = wave * RPW + r;
            const int src = (l < L ? l : 0) & 63;
            const float lo = __shfl(w0, src, WAVE), hi = __shfl(w1, src, WAVE);
            const float wr = (l < 64) ? lo : hi;
            float4* drow = reinterpret_cast<float4*>(a.dctx) + ((size_t)b * L + min(l, L - 1)) * n4;
#pragma unroll
            for (int i = 0; i < TXT_CPL; ++i) {
                const int c = lane + 64 * i;
                f4fma(g[r][i], al[r], v1[i]);
                f4fma(g[r][i], wr, v2[i]);
                if (l < L && c < n4) drow[c] = g[r][i];
            }
        }
    }
    float* orow = a.out + (size_t)b * a.ldo;
    block_row_sum<TXT_CPL, NW, TXT_SLOTS>(p, slots, n4, [&](int c, float4 t) {
        reinterpret_cast<float4*>(orow)[c] = t;
    });
}

template <int RPW, int MODE>
__global__ __launch_bounds__(TXT_NW * 64) void text_attn_kernel(TxtArgs a) {
    text_attn_body<RPW, MODE>(a, blockIdx.x);
}

// =================================================================================================
// Candidate scoring (model.py:342-352 after folding): one wave per candidate, A <= 16.
// forward : logit[b,a] = u_a . r[b] + (wt[b] . b_a + b_out)
// backward: dr[b] = sum_a dlogit[b,a] u_a ;  dc[b] = sum_a dlogit[b,a]
// =================================================================================================
constexpr int SC_CPL = 9, SC_NW = 16, SC_SLOTS = 4;

struct ScoreArgs {
    CandSrc src;
    int ldr;               // row stride of r
    const float* cst;      // optional per-row constant (stride ldr); null -> wt.b_a + b_out
    const float* r;        // fwd [B,ldr]
    const float* wt;       // fwd [B,D]
    const float* b_a;      // [D]
    const float* b_out;    // [1]
    int D;
    float* logit;          // fwd out [B,A]; bwd: dlogit in
    float* dr;             // bwd out [B,F]
    float* dc;             // bwd out [B]
    CeSrc ce;              // bwd: ce.logit != null -> d(logit) is formed here (and stored to `logit`)
};

// wt[b] . b_a + b_out (or the precomputed per-row constant): straight-line loads for D <= 256
__device__ __forceinline__ float score_const(const ScoreArgs& a, int b, int lane) {
    if (a.cst) return a.cst[(size_t)b * a.ldr];                  // block-uniform
    float c = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = lane + 64 * i;
        const float w = a.wt[(size_t)b * a.D + min(k, a.D - 1)] * a.b_a[min(k, a.D - 1)];
        c += k < a.D ? w : 0.f;
    }
    for (int k = lane + 256; k < a.D; k += 64) c += a.wt[(size_t)b * a.D + k] * a.b_a[k];
    return wave_sum(c) + a.b_out[0];
}

__global__ __launch_bounds__(SC_NW * 64) void score_fwd_kernel(ScoreArgs a) {
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int A = a.src.A;
    if (wave >= A) return;
    const int n4 = (a.src.IMG + a.src.LOC) >> 2;
    const CandRow row = cand_row(a.src, b, wave);
    const float4* rv = reinterpret_cast<const float4*>(a.r + (size_t)b * a.ldr);
    float4 x[SC_CPL], q[SC_CPL];
#pragma unroll
    for (int i = 0; i < SC_CPL; ++i) x[i] = f4zero();
    if (!row.zero) {        // wave-uniform: stop / padding rows cost no traffic; one block of loads
#pragma unroll
        for (int i = 0; i < SC_CPL; ++i) {
            const int c = lane + 64 * i;
            x[i] = cand_load(row, c, c < n4, n4);
            q[i] = rv[min(c, n4 - 1)];
        }
    }
    float d = 0.f;
    if (!row.zero) {
#pragma unroll
        for (int i = 0; i < SC_CPL; ++i) d += dot4(x[i], q[i]);  // x is zero beyond n4
    }
    const float cst = score_const(a, b, lane);
    d = wave_sum(d);
    if (lane == 0) a.logit[(size_t)b * A + wave] = d + cst;
}

// Scoring + per-step glue fused (one dependent stage instead of two): wave a keeps candidate a's row
// in registers, the 16 logits meet in LDS, wave 0 masks / soft-maxes / picks the action, and the wave
// that owns the chosen row writes dropout(u_next) straight into the next step's LSTM input.
__device__ __forceinline__ void score_glue_body(const ScoreArgs& a, const FGlue& g, int b) {
    __shared__ float s_logit[64];
    __shared__ int s_at;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int A = a.src.A;
    const int n4 = (a.src.IMG + a.src.LOC) >> 2;
    const CandRow row = cand_row(a.src, b, wave);
    const FGlueIn gin = follower_glue_load(g, b);               // (used by wave 0; cheap for the rest)
    const float4* rv = reinterpret_cast<const float4*>(a.r + (size_t)b * a.ldr);
    float4 x[SC_CPL], q[SC_CPL];
#pragma unroll
    for (int i = 0; i < SC_CPL; ++i) x[i] = f4zero();
    const bool have = wave < A && !row.zero;   // wave-uniform: stop / padding rows cost no traffic
    if (have) {
#pragma unroll
        for (int i = 0; i < SC_CPL; ++i) {
            const int c = lane + 64 * i;
            x[i] = cand_load(row, c, c < n4, n4);
            q[i] = rv[min(c, n4 - 1)];
        }
    }
    float d = 0.f;
    if (have) {
#pragma unroll
        for (int i = 0; i < SC_CPL; ++i) d += dot4(x[i], q[i]);
    }
    const float cst = score_const(a, b, lane);
    d = wave_sum(d);
    if (lane == 0 && wave < A) s_logit[wave] = d + cst;
    __syncthreads();
    if (wave == 0) {
        const int at = follower_glue_row(g, b, lane < A ? s_logit[lane] : 0.f, gin);
        if (lane == 0) s_at = at;
        if (g.nav.on && lane < A) nav_advance_slot(g.nav, b, lane, at, gin.was_ended || at == 0);
    }
    __syncthreads();
    if (g.u_next && wave == s_at) {
#pragma unroll
        for (int i = 0; i < SC_CPL; ++i) {
            const int c = lane + 64 * i;
            if (c < n4) store_u_next(g, b, c, x[i]);
        }
    }
}

__global__ __launch_bounds__(SC_NW * 64) void score_glue_kernel(ScoreArgs a, FGlue g) { score_glue_body(a, g, blockIdx.x); }

__global__ __launch_bounds__(SC_NW * 64) void score_bwd_kernel(ScoreArgs a) {
    __shared__ float4 slots[SC_SLOTS][SC_CPL * 64];
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int A = a.src.A;
    const int n4 = (a.src.IMG + a.src.LOC) >> 2;
    float w, dlv;
    if (a.ce.logit) {                                            // block-uniform: fused CE backward
        const int64_t tgt = a.ce.target[b];
        const float lg = a.ce.logit[(size_t)b * a.ce.ld + min(lane, A - 1)];
        const float l = lane < A ? lg : -INFINITY;
        const float m = wave_max(l);
        const float e = l > -INFINITY ? expf(l - m) : 0.f;
        const float ssum = wave_sum(e);
        const float gs = a.ce.gscale[0];
        dlv = (tgt == a.ce.ignore || lane >= A) ? 0.f : gs * (e / ssum - (lane == (int)tgt ? 1.f : 0.f));
        if (wave == 0 && lane < A) a.logit[(size_t)b * A + lane] = dlv;
        w = __shfl(dlv, min(wave, A - 1), WAVE);
    } else {
        w = a.logit[(size_t)b * A + min(wave, A - 1)];
        dlv = a.logit[(size_t)b * A + min(lane, A - 1)];
    }
    const CandRow row = cand_row(a.src, b, wave);
    float4 p[SC_CPL];
#pragma unroll
    for (int i = 0; i < SC_CPL; ++i) p[i] = f4zero();
    if (wave < A && !row.zero && w != 0.f) {                     // wave-uniform
#pragma unroll
        for (int i = 0; i < SC_CPL; ++i) {
            const int c = lane + 64 * i;
            f4fma(p[i], w, cand_load(row, c, c < n4, n4));
        }
    }
    if (wave == 0) {
        const float tot = wave_sum(lane < A ? dlv : 0.f);
        if (lane == 0) a.dc[b] = tot;
    }
    float* orow = a.dr + (size_t)b * (n4 << 2);
    block_row_sum<SC_CPL, SC_NW, SC_SLOTS>(p, slots, n4, [&](int c, float4 t) {
        reinterpret_cast<float4*>(orow)[c] = t;
    });
}

// =================================================================================================
// Paired launches.  A decode step is a chain of dependent, latency-bound kernels that each use a
// fraction of the chip; the visual half of step t+1 (t_v, q, visual attention: needs only h1 of
// step t) is independent of the text / scoring half of step t.  Two hipGraph branches (or streams)
// need a fork and a join per step, which cost more than the overlap gains (724K vs 815K agent-steps/s
// measured), so two independent kernels share ONE grid:
// blocks [0, nA) run body A, the rest body B.  Threads beyond a body's block size exit at once
// (whole waves: a workgroup barrier only counts live waves).
// =================================================================================================
template <int MTA, int CPWA, int MTB, int CPWB>
__global__ __launch_bounds__(SMALL_WAVES * 64) void pair_small_small_kernel(SmallArgs a, int gxa,
                                                                           int na, SmallArgs b,
                                                                           int gxb) {
    const int bid = blockIdx.x;
    if (bid < na)
        small_gemm_body<MTA, CPWA>(a, bid % gxa, bid / gxa);
    else
        small_gemm_body<MTB, CPWB>(b, (bid - na) % gxb, (bid - na) / gxb);
}

template <int MT, int CPW, int RPW>
__global__ __launch_bounds__(TXT_NW * 64) void pair_small_text_kernel(SmallArgs a, int gxa, int na,
                                                                      TxtArgs t) {
    const int bid = blockIdx.x;
    if (bid < na) {
        if (threadIdx.x >= SMALL_WAVES * 64) return;
        small_gemm_body<MT, CPW>(a, bid % gxa, bid / gxa);
    } else {
        text_attn_body<RPW, 0>(t, bid - na);
    }
}

// Folded inference step (sf_decoder_fold): the attention partials of step t+1 ride beside the TEXT
// attention of step t.  512-thread blocks: the attention body needs its 163 registers per lane, so the
// text body runs with 8 waves x RPW context rows (L <= 8 RPW) instead of 16 x RPW/2.
template <int RPW>
__global__ __launch_bounds__(SMALL_WAVES * 64) void pair_vis_text_kernel(VisArgs v, VisSplit sp, int nv,
                                                                        TxtArgs t) {
    const int bid = blockIdx.x;
    if (bid < nv) {
        if (threadIdx.x >= VSP_NW * 64) return;
        visual_split_body<1>(v, sp, bid % VSP_G, bid / VSP_G);
    } else {
        text_attn_body<RPW, 0, SMALL_WAVES>(t, bid - nv);
    }
}

template <int MT, int CPW, int PHASE, bool APRO = false>
__global__ __launch_bounds__(SMALL_WAVES * 64) void pair_vis_small_kernel(VisArgs v, VisSplit sp,
                                                                         int nv, SmallArgs b,
                                                                         int gxb) {
    const int bid = blockIdx.x;
    if (bid < nv) {
        if (threadIdx.x >= VSP_NW * 64) return;
        if (PHASE == 2)
            visual_split_body<2>(v, sp, 0, bid);
        else
            visual_split_body<PHASE>(v, sp, bid % VSP_G, bid / VSP_G);
    } else {
        small_gemm_body<MT, CPW, false, false, APRO>(b, (bid - nv) % gxb, (bid - nv) / gxb);
    }
}


// scoring + glue of step t beside the merge of the attention partials of step t+1 (folded inference chain: the two
// halves of the next step's LSTM input -- u_next from the glue, the attended feature from the merge -- land in one launch)
__global__ __launch_bounds__(SC_NW * 64) void pair_score_merge_kernel(ScoreArgs a, FGlue g, int nb, VisArgs v, VisSplit sp) {
    const int bid = blockIdx.x;
    if (bid < nb) {
        score_glue_body(a, g, bid);
    } else {
        if (threadIdx.x >= VSP_NW * 64) return;
        visual_split_body<2>(v, sp, 0, bid - nb);
    }
}


// =================================================================================================
// The text attention in FOLDED form (inference only: nothing here is taped for a backward).
// model.py:129-141 per decode step: t = W_in h1, s_l = ctx_l . t, alpha = softmax(s), wc = sum alpha_l ctx_l,
// h~ = tanh(W_out [wc ; h1]).  The context does not change during an episode, so two products leave the per-step
// chain for good (sf_text_fold_build, once per episode):
//     ctx_q = ctx W_in           s_l = ctx_q[l] . h1                     (no t = W_in h1 product per step)
//     ctx_o = ctx W_out[:, :H]^T W_out [wc ; h1] = sum alpha_l ctx_o[l] + W_out[:, H:] h1
// A step then needs, behind the cell: this body (scores + softmax + z = sum alpha_l ctx_o[l]) BESIDE the product
// y = W_out[:, H:] h1 in one launch, and h~ = tanh(z + y) is formed by the A-prologue of the next product
// (t_a = W_h h~ + b: sf_gemm_small.h, APRO) -- two dependent launches fewer per decode step.
// The body reads TWO context tensors, so a sample is split over TXF_G = 4 workgroups (positions [g Lg, (g + 1) Lg)): a
// CU sustains ~25-45 GB/s of loads and bytes per workgroup are what the stage costs.  Each keeps a flash-style piece
// (m, l, unnormalised z); the last arriver of a sample merges them in the same launch (measured, round 6: two groups
// merged by the consumer's prologue: stage 10.3 us + consumer 9.4 us; four groups merged here: 10.9 + 7.4).
// =================================================================================================
constexpr int TXF_G = 4;          // workgroups per sample (fixed: the per-sample ticket arithmetic counts in fours)

struct TxtFoldArgs {
    const float* ctx_q;    // [B, L, H]
    const float* ctx_o;    // [B, L, H]
    const uint8_t* mask;   // [B, L] (1 = padding) or null
    int L, H;
    const float* vec;      // h1 as the text attention sees it (eval: h1 itself) [B, ldvec]
    int ldvec;
    float* part;           // [B][TXF_G][H + 64]: z | e[l - g Lg] (<= 62) | m at H + 62 | l at H + 63
    unsigned* counter;     // [B] monotonic tickets (zero before the first launch; every launch adds TXF_G per sample)
    float* z;              // out [B, ldz]: sum_l alpha_l ctx_o[l] (merged, normalised)
    int ldz;
    float* alpha;          // out [B, L]: the attention weights (the tape's contract), or null
};

// One group's share -- positions [g LG, (g + 1) LG) of sample b: scores, local softmax piece (m, l, e), unnormalised
// z = sum e_l ctx_o[l] -- published WRITE-THROUGH like the split visual attention's partials (visual_split_body<0>:
// sc1 stores, drain, agent-scope ticket; no cache-wide fence), and the workgroup that draws the sample's LAST ticket
// merges the TXF_G pieces (sc1 loads) into z [H] and alpha [L]: no spinning, no extra launch.
template <int RPW>
__device__ __forceinline__ void text_fold_body(const TxtFoldArgs& a, int g, int b) {
    constexpr int NW = SMALL_WAVES, SL = 4, LG = NW * RPW;
    static_assert(LG <= 62, "a group's weights live in 62 record slots");
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    __shared__ float4 slots[SL][TXT_CPL * 64];
    __shared__ float s_score[LG];
    __shared__ int s_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int L = a.L, H = a.H, n4 = H >> 2;
    const int pstride = H + 64;
    const float4* cq = reinterpret_cast<const float4*>(a.ctx_q) + (size_t)b * L * n4;
    const float4* co = reinterpret_cast<const float4*>(a.ctx_o) + (size_t)b * L * n4;
    const bool use_mask = a.mask != nullptr;
    const uint8_t* mrow = use_mask ? a.mask + (size_t)b * L : reinterpret_cast<const uint8_t*>(cq);
    // a group whose positions are ALL padding (short instructions: the length-sorted minibatch ends far below L)
    // pulls nothing: its piece is (m = -inf, l = 0, z = 0) and it only draws its ticket
    const int lt = g * LG + tid;
    const bool padded = tid >= LG || lt >= L || (use_mask && mrow[min(lt, L - 1)] != 0);
    const bool empty = __syncthreads_and(padded) != 0;              // block-uniform
    float m = -INFINITY, e = 0.f, lsum = 0.f;
    float4 p[TXT_CPL];
#pragma unroll
    for (int i = 0; i < TXT_CPL; ++i) p[i] = f4zero();
    if (!empty) {
    // straight-line loads, clamped indices, value selects (sf_rows.h): every row of both tensors is in flight at once
    float4 xq[RPW][TXT_CPL], xo[RPW][TXT_CPL];
    uint8_t mk[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int l = g * LG + wave * RPW + r;
        const int lc = min(l, L - 1);
#pragma unroll
        for (int i = 0; i < TXT_CPL; ++i) {
            const int c = lane + 64 * i;
            const float4 tq = cq[(size_t)lc * n4 + min(c, n4 - 1)];
            const float4 to = co[(size_t)lc * n4 + min(c, n4 - 1)];
            const bool ok = l < L && c < n4;
            xq[r][i] = ok ? tq : f4zero();
            xo[r][i] = ok ? to : f4zero();
        }
        mk[r] = mrow[lc];
    }
    float4 v[TXT_CPL];
#pragma unroll
    for (int i = 0; i < TXT_CPL; ++i) {
        const int c = lane + 64 * i;
        const float4 t = reinterpret_cast<const float4*>(a.vec + (size_t)b * a.ldvec)[min(c, n4 - 1)];
        v[i] = c < n4 ? t : f4zero();
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        float d = 0.f;
#pragma unroll
        for (int i = 0; i < TXT_CPL; ++i) d += dot4(xq[r][i], v[i]);
        d = wave_sum(d);
        const int l = g * LG + wave * RPW + r;
        if (lane == 0) s_score[wave * RPW + r] = (l >= L || (use_mask && mk[r])) ? -INFINITY : d;
    }
    __syncthreads();
    const float sc = lane < LG ? s_score[lane] : -INFINITY;
    m = wave_max(sc);                                              // -inf: every position of this group is padding
    e = sc > -INFINITY ? expf(sc - m) : 0.f;
    lsum = wave_sum(e);
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const float er = __shfl(e, wave * RPW + r, WAVE);
#pragma unroll
        for (int i = 0; i < TXT_CPL; ++i) f4fma(p[i], er, xo[r][i]);
    }
    }   // !empty
    float* r0 = a.part + (size_t)b * TXF_G * pstride;
    float* rec = r0 + (size_t)g * pstride;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(rec, 0, pstride * 4, 0x00020000);
    block_row_sum<TXT_CPL, NW, SL>(p, slots, n4, [&](int c, float4 t) {
        const v4u w{__float_as_uint(t.x), __float_as_uint(t.y), __float_as_uint(t.z), __float_as_uint(t.w)};
        __builtin_amdgcn_raw_buffer_store_b128(w, rs, c * 16, 0, 16);
    });
    if (wave == 0) {
        if (lane < LG) __hip_atomic_store(rec + H + lane, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane == 0) {
            __hip_atomic_store(rec + H + 62, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(rec + H + 63, lsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // EVERY storing wave drains
    __syncthreads();
    if (tid == 0)
        s_last = (__hip_atomic_fetch_add(a.counter + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) &
                  (unsigned)(TXF_G - 1)) == (unsigned)(TXF_G - 1);
    __syncthreads();
    if (!s_last) return;

    // ---- the sample's last arriver merges the pieces
    auto ldf = [&](float* q) { return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    float mg[TXF_G], kk[TXF_G];
    float M = -INFINITY;
#pragma unroll
    for (int k = 0; k < TXF_G; ++k) {
        mg[k] = ldf(r0 + (size_t)k * pstride + H + 62);
        M = fmaxf(M, mg[k]);
    }
    float Lsum = 0.f;
#pragma unroll
    for (int k = 0; k < TXF_G; ++k) {
        kk[k] = mg[k] > -INFINITY ? expf(mg[k] - M) : 0.f;      // an empty group: weight 0 (its l is 0 too)
        Lsum = fmaf(ldf(r0 + (size_t)k * pstride + H + 63), kk[k], Lsum);
    }
    const float inv = 1.0f / Lsum;                              // (a sample always has an unmasked position)
#pragma unroll
    for (int k = 0; k < TXF_G; ++k) kk[k] *= inv;
    if (a.alpha) {
        for (int l = tid; l < L; l += NW * 64) {
            const int gk = l / LG;
            float w = kk[0];
#pragma unroll
            for (int k = 1; k < TXF_G; ++k) w = gk == k ? kk[k] : w;
            a.alpha[(size_t)b * L + l] = ldf(r0 + (size_t)gk * pstride + H + (l - gk * LG)) * w;
        }
    }
    const auto rs0 = __builtin_amdgcn_make_buffer_rsrc(r0, 0, TXF_G * pstride * 4, 0x00020000);
    float4* zrow = reinterpret_cast<float4*>(a.z + (size_t)b * a.ldz);
    for (int c = tid; c < n4; c += NW * 64) {
        float4 t = f4zero();
#pragma unroll
        for (int k = 0; k < TXF_G; ++k) {
            const v4u pk = __builtin_amdgcn_raw_buffer_load_b128(rs0, k * pstride * 4 + c * 16, 0, 16);
            t.x = fmaf(kk[k], __uint_as_float(pk.x), t.x);
            t.y = fmaf(kk[k], __uint_as_float(pk.y), t.y);
            t.z = fmaf(kk[k], __uint_as_float(pk.z), t.z);
            t.w = fmaf(kk[k], __uint_as_float(pk.w), t.w);
        }
        zrow[c] = t;
    }
}

// the folded text stage: [text_fold groups | y = W_out[:, H:] h1 | t_v' = W_h h1 + b] in ONE grid
template <int RPW>
__global__ __launch_bounds__(SMALL_WAVES * 64) void pair_textfold_small_small_kernel(TxtFoldArgs t, int nt, SmallArgs a,
                                                                                    int gxa, int na, SmallArgs b, int gxb) {
    const int bid = blockIdx.x;
    if (bid < nt)
        text_fold_body<RPW>(t, bid % TXF_G, bid / TXF_G);
    else if (bid < nt + na)
        small_gemm_body<1, 4>(a, (bid - nt) % gxa, (bid - nt) / gxa);
    else
        small_gemm_body<1, 4>(b, (bid - nt - na) % gxb, (bid - nt - na) / gxb);
}

// ... the same with the next step's visual query as ONE folded product q' = M_v h1 + c_v (sf_decoder_fold) for third body
template <int RPW, int MTB>
__global__ __launch_bounds__(SMALL_WAVES * 64) void pair_textfold_small_wide_kernel(TxtFoldArgs t, int nt, SmallArgs a,
                                                                                   int gxa, int na, SmallArgs b, int gxb) {
    const int bid = blockIdx.x;
    if (bid < nt)
        text_fold_body<RPW>(t, bid % TXF_G, bid / TXF_G);
    else if (bid < nt + na)
        small_gemm_body<1, 4>(a, (bid - nt) % gxa, (bid - nt) / gxa);
    else
        small_gemm_body<MTB, 4>(b, (bid - nt - na) % gxb, (bid - nt - na) / gxb);
}

// t_a = W_h tanh(z + y) + b (A-prologue: z = the merged attention sum of the text_fold launch) beside q' = W_v^T t_v'
template <int MTB>
__global__ __launch_bounds__(SMALL_WAVES * 64) void pair_apro_small_kernel(SmallArgs a, int gxa, int na, SmallArgs b, int gxb) {
    const int bid = blockIdx.x;
    if (bid < na)
        small_gemm_body<1, 4, false, false, true>(a, bid % gxa, bid / gxa);
    else
        small_gemm_body<MTB, 2>(b, (bid - na) % gxb, (bid - na) / gxb);
}

// Deferred gradient of the instruction context (model.py:129-139 backward, summed over an episode):
// dctx[b,l,:] += sum_t ( alpha[t,b,l] dwc[t,b,:] + ds[t,b,l] tt[t,b,:] ).  The per-step form reads and
// writes the whole [B,L,H] gradient S times (2 x 16 MB per step at the headline shape); this reads each
// step's two H-vectors once.  One block per sample; thread = one float4 column x every 8th row.
constexpr int CG_ROWS = 10;        // rows per thread: L <= 8 * CG_ROWS

__global__ __launch_bounds__(1024) void ctx_grad_kernel(const float* alpha, const float* ds,
                                                        const float* dcat2, int lddc, const float* tt,
                                                        int S, int B, int L, int H, float* dctx) {
    extern __shared__ float s_w[];                       // [S][2][L]: alpha, ds of this sample
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < S * L; i += 1024) {
        const int t = i / L, l = i - t * L;
        s_w[(t * 2 + 0) * L + l] = alpha[((size_t)t * B + b) * L + l];
        s_w[(t * 2 + 1) * L + l] = ds[((size_t)t * B + b) * L + l];
    }
    __syncthreads();
    const int n4 = H >> 2;
    const int c = tid % n4, lg = tid / n4;               // n4 = 128: 8 row groups
    const int ngroups = 1024 / n4;
    if (lg >= ngroups) return;
    float4 acc[CG_ROWS];
#pragma unroll
    for (int r = 0; r < CG_ROWS; ++r) acc[r] = f4zero();
    for (int t = 0; t < S; ++t) {
        const float4 d = reinterpret_cast<const float4*>(dcat2 + ((size_t)t * B + b) * lddc)[c];
        const float4 x = reinterpret_cast<const float4*>(tt + ((size_t)t * B + b) * H)[c];
#pragma unroll
        for (int r = 0; r < CG_ROWS; ++r) {
            const int l = lg + ngroups * r;
            if (l < L) {
                f4fma(acc[r], s_w[(t * 2 + 0) * L + l], d);
                f4fma(acc[r], s_w[(t * 2 + 1) * L + l], x);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < CG_ROWS; ++r) {
        const int l = lg + ngroups * r;
        if (l < L) {
            float4* o = reinterpret_cast<float4*>(dctx + ((size_t)b * L + l) * H) + c;
            float4 v = *o;
            f4add(v, acc[r]);
            *o = v;
        }
    }
}


// Backward: the visual-attention backward of a step (needs the input gradient of the LSTM) beside the
// recurrent data gradient dh0 = dgates W_hh (needs only dgates): independent, one launch.
// Blocks [0, nv): attention backward (12 waves); the rest: the small product (threads >= 512 leave).
template <int MT, int CPW>
__global__ __launch_bounds__(VIS_NW * 64) void pair_visbwd_small_kernel(VisArgs v, int nv, SmallArgs b,
                                                                       int gxb) {
    const int bid = blockIdx.x;
    if (bid < nv) {
        visual_attn_body<1>(v, bid);
    } else {
        if (threadIdx.x >= SMALL_WAVES * 64) return;
        small_gemm_body<MT, CPW>(b, (bid - nv) % gxb, (bid - nv) / gxb);
    }
}

}  // namespace

size_t visual_attn_split_floats(int B, int F) { return (size_t)B * VSP_G * (F + 64); }

bool visual_attn_f64_supported(const PanoSrc& src, int B) {
    const int F = src.IMG + src.LOC;
    return !(src.V > VIS_RPW * VIS_NW || src.V > 64 || F > VIS_CPL * 256 || (F & 3) ||
             (!src.dense && ((src.IMG & 3) || (src.LOC & 3)))) &&
           src.V > (VSP_G - 1) * VSP_RPG && src.V <= VSP_G * VSP_RPG && B <= 256;
}

int visual_attn(int mode, const PanoSrc& src, int B, const float* vec, int ldvec, float* alpha,
                float* out, int ldo, const Dropout& drop, int drop_col0, hipStream_t st,
                float* split_part, unsigned* split_counter, const double* vec64, int vec_slabs, long vec_slab_stride) {
    const int F = src.IMG + src.LOC;
    if (src.V > VIS_RPW * VIS_NW || src.V > 64 || F > VIS_CPL * 256 || (F & 3) ||
        (!src.dense && ((src.IMG & 3) || (src.LOC & 3))) || (ldvec & 3) || (ldo & 3))
        return SF_ERR_UNSUPPORTED;
    VisArgs a{src, vec, ldvec, alpha, out, ldo, drop, drop_col0, vec64, mode == 1 ? vec_slabs : 0, vec_slab_stride};
    if (mode != 1 && vec_slabs > 1) return SF_ERR_UNSUPPORTED;
    if (vec64) {        // float64 scores: the two-workgroup forward only (callers check visual_attn_f64_supported)
        if (mode != 0 || !split_part || !split_counter || !visual_attn_f64_supported(src, B)) return SF_ERR_UNSUPPORTED;
        SF_LAUNCH(visual_attn_split_f64_kernel, dim3(VSP_G, B), dim3(VSP_NW * 64), 0, st, a,
                  VisSplit{split_part, split_counter, nullptr});
        return launch_status();
    }
    // small batches: two workgroups per sample (see visual_attn_split_kernel)
    if (mode == 0 && split_part && split_counter && src.V > (VSP_G - 1) * VSP_RPG &&
        src.V <= VSP_G * VSP_RPG && B <= 256) {
        SF_LAUNCH(visual_attn_split_kernel, dim3(VSP_G, B), dim3(VSP_NW * 64), 0, st, a,
                           VisSplit{split_part, split_counter, nullptr});
        return launch_status();
    }
    if (mode == 0)
        SF_LAUNCH(visual_attn_kernel<0>, dim3(B), dim3(VIS_NW * 64), 0, st, a);
    else
        SF_LAUNCH(visual_attn_kernel<1>, dim3(B), dim3(VIS_NW * 64), 0, st, a);
    return launch_status();
}

template <int MODE>
static int text_attn_launch(const TxtArgs& a, int B, hipStream_t st) {
    if (a.L <= TXT_NW)
        SF_LAUNCH((text_attn_kernel<1, MODE>), dim3(B), dim3(TXT_NW * 64), 0, st, a);
    else if (a.L <= TXT_NW * 5)
        SF_LAUNCH((text_attn_kernel<5, MODE>), dim3(B), dim3(TXT_NW * 64), 0, st, a);
    else if (a.L <= TXT_NW * 8)
        SF_LAUNCH((text_attn_kernel<8, MODE>), dim3(B), dim3(TXT_NW * 64), 0, st, a);
    else
        return SF_ERR_UNSUPPORTED;
    return launch_status();
}

int text_attn_fwd(const float* ctx, const uint8_t* mask, int B, int L, int H, const float* t,
                  int ldt, float* alpha, float* wc, int ldwc, hipStream_t st,
                  const int32_t* ctx_row) {
    if (H > TXT_CPL * 256 || (H & 3) || (ldt & 3) || (ldwc & 3) || L < 1) return SF_ERR_UNSUPPORTED;
    TxtArgs a{ctx, mask, L, H, t, ldt, nullptr, 0, alpha, wc, ldwc, nullptr, ctx_row, nullptr};
    return text_attn_launch<0>(a, B, st);
}

int text_attn_bwd(const float* ctx, int B, int L, int H, const float* dwc, int lddwc,
                  const float* t, int ldt, const float* alpha, float* dt, int lddt, float* dctx,
                  hipStream_t st, float* ds_out, const int32_t* ctx_row) {
    if (H > TXT_CPL * 256 || (H & 3) || (ldt & 3) || (lddwc & 3) || (lddt & 3) || L < 1 || (ctx_row && dctx))
        return SF_ERR_UNSUPPORTED;
    TxtArgs a{ctx, nullptr, L, H, dwc, lddwc, t, ldt, const_cast<float*>(alpha), dt, lddt, dctx, ctx_row, ds_out};
    return text_attn_launch<1>(a, B, st);
}

bool ctx_grad_supported(int S, int L, int H) {
    const int n4 = H >> 2;
    return !(H & 3) && n4 >= 1 && n4 <= 1024 && 1024 % n4 == 0 && L <= (1024 / n4) * CG_ROWS &&
           (size_t)S * 2 * L * sizeof(float) <= 64 * 1024;
}

int ctx_grad_accum(const float* alpha, const float* ds, const float* dcat2, int lddc, const float* tt,
                   int S, int B, int L, int H, float* dctx, hipStream_t st) {
    if (!ctx_grad_supported(S, L, H) || (lddc & 3)) return SF_ERR_UNSUPPORTED;
    SF_LAUNCH(ctx_grad_kernel, dim3(B), dim3(1024), (size_t)S * 2 * L * sizeof(float), st, alpha,
                       ds, dcat2, lddc, tt, S, B, L, H, dctx);
    return launch_status();
}

int score_fwd(const CandSrc& src, int B, int D, const float* r, const float* wt, const float* b_a,
              const float* b_out, float* logit, hipStream_t st, int ldr, const float* cst) {
    const int F = src.IMG + src.LOC;
    if (ldr <= 0) ldr = F;
    if (src.A > SC_NW || src.A < 1 || F > SC_CPL * 256 || (F & 3) ||
        (!src.dense && ((src.IMG & 3) || (src.LOC & 15))))
        return SF_ERR_UNSUPPORTED;
    ScoreArgs a{src, ldr, cst, r, wt, b_a, b_out, D, logit, nullptr, nullptr, CeSrc{}};
    SF_LAUNCH(score_fwd_kernel, dim3(B), dim3(SC_NW * 64), 0, st, a);
    return launch_status();
}

int score_glue_fwd(const CandSrc& src, int B, int D, const float* r, const float* wt,
                   const float* b_a, const float* b_out, const FGlue& g, hipStream_t st, int ldr,
                   const float* cst) {
    const int F = src.IMG + src.LOC;
    if (ldr <= 0) ldr = F;
    if (src.A > SC_NW || src.A < 1 || F > SC_CPL * 256 || (F & 3) ||
        (!src.dense && ((src.IMG & 3) || (src.LOC & 15))))
        return SF_ERR_UNSUPPORTED;
    ScoreArgs a{src, ldr, cst, r, wt, b_a, b_out, D, g.logit, nullptr, nullptr, CeSrc{}};
    SF_LAUNCH(score_glue_kernel, dim3(B), dim3(SC_NW * 64), 0, st, a, g);
    return launch_status();
}

// score_glue_fwd beside the phase-2 merge of visual-attention partials written by an earlier phase-1 launch
int pair_score_merge(const CandSrc& src, int B, int D, const float* r, const float* wt, const float* b_a,
                     const float* b_out, const FGlue& g, const PanoSrc& psrc, float* alpha, float* out, int ldo,
                     const Dropout& drop, int drop_col0, float* split_part, hipStream_t st, int ldr, const float* cst) {
    const int F = src.IMG + src.LOC;
    if (ldr <= 0) ldr = F;
    if (src.A > SC_NW || src.A < 1 || F > SC_CPL * 256 || (F & 3) || (!src.dense && ((src.IMG & 3) || (src.LOC & 15))))
        return SF_ERR_UNSUPPORTED;
    if (!split_part || psrc.V <= (VSP_G - 1) * VSP_RPG || psrc.V > VSP_G * VSP_RPG || B > 256 || (ldo & 3))
        return SF_ERR_UNSUPPORTED;
    ScoreArgs a{src, ldr, cst, r, wt, b_a, b_out, D, g.logit, nullptr, nullptr, CeSrc{}};
    VisArgs va{psrc, nullptr, 0, alpha, out, ldo, drop, drop_col0};
    const VisSplit sp{split_part, nullptr, g_trace};
    SF_LAUNCH(pair_score_merge_kernel, dim3(2 * B), dim3(SC_NW * 64), 0, st, a, g, B, va, sp);
    return launch_status();
}

int score_bwd(const CandSrc& src, int B, const float* dlogit, float* dr, float* dc,
              hipStream_t st, const CeSrc* ce) {
    const int F = src.IMG + src.LOC;
    if (src.A > SC_NW || src.A < 1 || F > SC_CPL * 256 || (F & 3) ||
        (!src.dense && ((src.IMG & 3) || (src.LOC & 15))))
        return SF_ERR_UNSUPPORTED;
    ScoreArgs a{src, F, nullptr, nullptr, nullptr, nullptr, nullptr, 0, const_cast<float*>(dlogit), dr, dc,
                ce ? *ce : CeSrc{}};
    if (ce && ce->ld < src.A) return SF_ERR_ARG;
    SF_LAUNCH(score_bwd_kernel, dim3(B), dim3(SC_NW * 64), 0, st, a);
    return launch_status();
}

// ---- paired launches (host side).  SF_ERR_UNSUPPORTED = "not pairable": the caller launches the
// two kernels one after the other instead.
int pair_small_small(const SmallPlan& a, const SmallPlan& b, hipStream_t st) {
    if (!(a.mt == 1 && a.cpw == 4 && b.cpw == 4 && (b.mt == 1 || b.mt == 2 || b.mt == 4))) return SF_ERR_UNSUPPORTED;
    const int na = a.gx * a.gy, nb = b.gx * b.gy;
    const dim3 grid(na + nb), block(SMALL_WAVES * 64);
    if (b.mt == 1)
        SF_LAUNCH((pair_small_small_kernel<1, 4, 1, 4>), grid, block, 0, st, a.args, a.gx, na, b.args, b.gx);
    else if (b.mt == 2)
        SF_LAUNCH((pair_small_small_kernel<1, 4, 2, 4>), grid, block, 0, st, a.args, a.gx, na, b.args, b.gx);
    else
        SF_LAUNCH((pair_small_small_kernel<1, 4, 4, 4>), grid, block, 0, st, a.args, a.gx, na, b.args, b.gx);
    return launch_status();
}


// The folded text stage of an inference decode step (see text_fold_body): SF_ERR_UNSUPPORTED = shapes outside the
// instantiations (the caller runs the unfolded stages).
int pair_textfold_small_small(const float* ctx_q, const float* ctx_o, const uint8_t* mask, int B, int L, int H,
                              const float* vec, int ldvec, float* part, unsigned* counter, float* z, int ldz, float* alpha,
                              const SmallPlan& a, const SmallPlan& b, hipStream_t st) {
    if (!(a.mt == 1 && a.cpw == 4 && b.cpw == 4 && (b.mt == 1 || b.mt == 2 || b.mt == 4))) return SF_ERR_UNSUPPORTED;
    if (H > TXT_CPL * 256 || (H & 3) || (ldvec & 3) || L < 1 || L > TXF_G * SMALL_WAVES * 5 || B > 512 || !counter || !z ||
        (ldz & 3) || ldz < H)
        return SF_ERR_UNSUPPORTED;
    const TxtFoldArgs ta{ctx_q, ctx_o, mask, L, H, vec, ldvec, part, counter, z, ldz, alpha};
    const int nt = TXF_G * B, na = a.gx * a.gy, nb = b.gx * b.gy;
    const dim3 grid(nt + na + nb), block(SMALL_WAVES * 64);
    const int rpw = (L + TXF_G * SMALL_WAVES - 1) / (TXF_G * SMALL_WAVES);
    if (b.mt > 1) {                                   // (the folded visual query: N = F columns)
#define SF_TFW(R, M) SF_LAUNCH((pair_textfold_small_wide_kernel<R, M>), grid, block, 0, st, ta, nt, a.args, a.gx, na, b.args, b.gx)
        if (b.mt == 2) {
            if (rpw <= 1) SF_TFW(1, 2); else if (rpw <= 2) SF_TFW(2, 2); else if (rpw <= 3) SF_TFW(3, 2); else SF_TFW(5, 2);
        } else {
            if (rpw <= 1) SF_TFW(1, 4); else if (rpw <= 2) SF_TFW(2, 4); else if (rpw <= 3) SF_TFW(3, 4); else SF_TFW(5, 4);
        }
#undef SF_TFW
        return launch_status();
    }
    if (rpw <= 1)
        SF_LAUNCH((pair_textfold_small_small_kernel<1>), grid, block, 0, st, ta, nt, a.args, a.gx, na, b.args, b.gx);
    else if (rpw <= 2)
        SF_LAUNCH((pair_textfold_small_small_kernel<2>), grid, block, 0, st, ta, nt, a.args, a.gx, na, b.args, b.gx);
    else if (rpw <= 3)
        SF_LAUNCH((pair_textfold_small_small_kernel<3>), grid, block, 0, st, ta, nt, a.args, a.gx, na, b.args, b.gx);
    else
        SF_LAUNCH((pair_textfold_small_small_kernel<5>), grid, block, 0, st, ta, nt, a.args, a.gx, na, b.args, b.gx);
    return launch_status();
}
size_t text_fold_part_floats(int B, int H) { return (size_t)B * TXF_G * (H + 64); }


// a: the product whose A operand the prologue forms (a.args.apro_part set by the caller), b: a plain small product
int pair_apro_small(const SmallPlan& a, const SmallPlan& b, hipStream_t st) {
    if (!(a.mt == 1 && a.cpw == 4 && b.cpw == 2 && (b.mt == 1 || b.mt == 2 || b.mt == 4)) || !a.args.apro_part ||
        a.args.sg.total != a.args.sg.n0)
        return SF_ERR_UNSUPPORTED;
    const int na = a.gx * a.gy, nb = b.gx * b.gy;
    const dim3 grid(na + nb), block(SMALL_WAVES * 64);
    if (b.mt == 4)
        SF_LAUNCH((pair_apro_small_kernel<4>), grid, block, 0, st, a.args, a.gx, na, b.args, b.gx);
    else if (b.mt == 2)
        SF_LAUNCH((pair_apro_small_kernel<2>), grid, block, 0, st, a.args, a.gx, na, b.args, b.gx);
    else
        SF_LAUNCH((pair_apro_small_kernel<1>), grid, block, 0, st, a.args, a.gx, na, b.args, b.gx);
    return launch_status();
}

// visual-attention partials (phase 1 of the split attention) beside the text attention
int pair_vis_text(const PanoSrc& src, int B, const float* vec, int ldvec, float* alpha, float* out, int ldo,
                  const Dropout& drop, int drop_col0, float* split_part, const float* ctx, const uint8_t* mask,
                  int L, int H, const float* t, int ldt, float* talpha, float* wc, int ldwc,
                  const int32_t* ctx_row, hipStream_t st) {
    const int F = src.IMG + src.LOC;
    if (!split_part || src.V <= (VSP_G - 1) * VSP_RPG || src.V > VSP_G * VSP_RPG || B > 256 || F > VIS_CPL * 256 ||
        (F & 3) || (!src.dense && ((src.IMG & 3) || (src.LOC & 3))) || (ldvec & 3) || (ldo & 3))
        return SF_ERR_UNSUPPORTED;
    if (H > TXT_CPL * 256 || (H & 3) || (ldt & 3) || (ldwc & 3) || L < 1 || L > SMALL_WAVES * 10)
        return SF_ERR_UNSUPPORTED;
    VisArgs va{src, vec, ldvec, alpha, out, ldo, drop, drop_col0};
    const VisSplit sp{split_part, nullptr, g_trace};
    TxtArgs ta{ctx, mask, L, H, t, ldt, nullptr, 0, talpha, wc, ldwc, nullptr, ctx_row, nullptr};
    const int nv = VSP_G * B;
    const dim3 grid(nv + B), block(SMALL_WAVES * 64);
    if (L <= SMALL_WAVES * 2)
        SF_LAUNCH((pair_vis_text_kernel<2>), grid, block, 0, st, va, sp, nv, ta);
    else if (L <= SMALL_WAVES * 5)
        SF_LAUNCH((pair_vis_text_kernel<5>), grid, block, 0, st, va, sp, nv, ta);
    else
        SF_LAUNCH((pair_vis_text_kernel<10>), grid, block, 0, st, va, sp, nv, ta);
    return launch_status();
}

int pair_small_text(const SmallPlan& a, const float* ctx, const uint8_t* mask, int B, int L, int H,
                    const float* t, int ldt, float* alpha, float* wc, int ldwc,
                    const int32_t* ctx_row, hipStream_t st) {
    if (!(a.mt == 4 && a.cpw == 2)) return SF_ERR_UNSUPPORTED;
    if (H > TXT_CPL * 256 || (H & 3) || (ldt & 3) || (ldwc & 3) || L < 1 || L > TXT_NW * 8)
        return SF_ERR_UNSUPPORTED;
    TxtArgs ta{ctx, mask, L, H, t, ldt, nullptr, 0, alpha, wc, ldwc, nullptr, ctx_row, nullptr};
    const int na = a.gx * a.gy;
    const dim3 grid(na + B), block(TXT_NW * 64);
    if (L <= TXT_NW)
        SF_LAUNCH((pair_small_text_kernel<4, 2, 1>), grid, block, 0, st, a.args, a.gx, na, ta);
    else if (L <= TXT_NW * 5)
        SF_LAUNCH((pair_small_text_kernel<4, 2, 5>), grid, block, 0, st, a.args, a.gx, na, ta);
    else
        SF_LAUNCH((pair_small_text_kernel<4, 2, 8>), grid, block, 0, st, a.args, a.gx, na, ta);
    return launch_status();
}

// visual-attention backward (mode 1 of visual_attn) paired with a small product (mt 1, cpw 16: the
// K = 4H recurrent data gradient); SF_ERR_UNSUPPORTED = not pairable, launch separately
int pair_visbwd_small(const PanoSrc& src, int B, const float* vec, int ldvec, float* alpha, float* out,
                      int ldo, const Dropout& drop, int drop_col0, const SmallPlan& b, hipStream_t st, int vec_slabs,
                      long vec_slab_stride) {
    const int F = src.IMG + src.LOC;
    if (!(b.mt == 1 && b.cpw == 16)) return SF_ERR_UNSUPPORTED;
    if (src.V > VIS_RPW * VIS_NW || src.V > 64 || F > VIS_CPL * 256 || (F & 3) ||
        (!src.dense && ((src.IMG & 3) || (src.LOC & 3))) || (ldvec & 3) || (ldo & 3))
        return SF_ERR_UNSUPPORTED;
    VisArgs va{src, vec, ldvec, alpha, out, ldo, drop, drop_col0, nullptr, vec_slabs, vec_slab_stride};
    const int nb = b.gx * b.gy;
    SF_LAUNCH((pair_visbwd_small_kernel<1, 16>), dim3(B + nb), dim3(VIS_NW * 64), 0, st, va, B,
                       b.args, b.gx);
    return launch_status();
}

// the folded scoring product [r | c] = M_a tanh(z + y) + c_a (A-prologue; b.args.apro_part set) beside the phase-1
// partials of the visual attention -- or alone (`src` null: a device-resident environment, an episode's last step)
int pair_vis_apro(const PanoSrc* src, int B, const float* vec, int ldvec, float* split_part, const SmallPlan& b,
                  hipStream_t st) {
    if (!(b.cpw == 4 && (b.mt == 1 || b.mt == 2 || b.mt == 4)) || !b.args.apro_part || b.args.sg.total != b.args.sg.n0)
        return SF_ERR_UNSUPPORTED;
    int nv = 0;
    VisArgs va{};
    if (src) {
        const int F = src->IMG + src->LOC;
        if (!split_part || src->V <= (VSP_G - 1) * VSP_RPG || src->V > VSP_G * VSP_RPG || B > 256 || F > VIS_CPL * 256 ||
            (F & 3) || (!src->dense && ((src->IMG & 3) || (src->LOC & 3))) || (ldvec & 3))
            return SF_ERR_UNSUPPORTED;
        va = VisArgs{*src, vec, ldvec, nullptr, nullptr, 0, Dropout{}, 0};
        nv = VSP_G * B;
    }
    const VisSplit sp{split_part, nullptr, g_trace};
    const int nb = b.gx * b.gy;
    const dim3 grid(nv + nb), block(SMALL_WAVES * 64);
    if (b.mt == 4)
        SF_LAUNCH((pair_vis_small_kernel<4, 4, 1, true>), grid, block, 0, st, va, sp, nv, b.args, b.gx);
    else if (b.mt == 2)
        SF_LAUNCH((pair_vis_small_kernel<2, 4, 1, true>), grid, block, 0, st, va, sp, nv, b.args, b.gx);
    else
        SF_LAUNCH((pair_vis_small_kernel<1, 4, 1, true>), grid, block, 0, st, va, sp, nv, b.args, b.gx);
    return launch_status();
}

extern "C" void sf_debug_trace(unsigned long long* buf) { g_trace = buf; }
extern "C" void sf_debug_force_write_through(int on) { g_force_sc1 = on; }

int pair_vis_small(const PanoSrc& src, int B, const float* vec, int ldvec, float* alpha, float* out,
                   int ldo, const Dropout& drop, int drop_col0, float* split_part,
                   unsigned* split_counter, const SmallPlan& b, hipStream_t st, int phase) {
    const int F = src.IMG + src.LOC;
    // (the r = W_a^T wt product of the folded text chain: K = D = 256 -> two chunks per wave)
    const bool wide = b.cpw == 2 && (b.mt == 1 || b.mt == 2 || b.mt == 4) && phase != 2;
    if (!wide && !(b.mt == 1 && (b.cpw == 8 || (phase == 2 && b.cpw == 4)))) return SF_ERR_UNSUPPORTED;
    if (!split_part || (phase == 0 && !split_counter) || src.V <= (VSP_G - 1) * VSP_RPG || src.V > VSP_G * VSP_RPG ||
        B > 256 ||
        F > VIS_CPL * 256 || (F & 3) || (!src.dense && ((src.IMG & 3) || (src.LOC & 3))) ||
        (ldvec & 3) || (ldo & 3))
        return SF_ERR_UNSUPPORTED;
    VisArgs va{src, vec, ldvec, alpha, out, ldo, drop, drop_col0};
    const int nv = (phase == 2 ? 1 : VSP_G) * B, nb = b.gx * b.gy;
    const dim3 grid(nv + nb), block(SMALL_WAVES * 64);
    const VisSplit sp{split_part, split_counter, g_trace};
    if (wide && b.mt == 4 && phase == 0)
        SF_LAUNCH((pair_vis_small_kernel<4, 2, 0>), grid, block, 0, st, va, sp, nv, b.args, b.gx);
    else if (wide && b.mt == 4)
        SF_LAUNCH((pair_vis_small_kernel<4, 2, 1>), grid, block, 0, st, va, sp, nv, b.args, b.gx);
    else if (wide && b.mt == 2 && phase == 0)
        SF_LAUNCH((pair_vis_small_kernel<2, 2, 0>), grid, block, 0, st, va, sp, nv, b.args, b.gx);
    else if (wide && b.mt == 2)
        SF_LAUNCH((pair_vis_small_kernel<2, 2, 1>), grid, block, 0, st, va, sp, nv, b.args, b.gx);
    else if (wide && phase == 0)
        SF_LAUNCH((pair_vis_small_kernel<1, 2, 0>), grid, block, 0, st, va, sp, nv, b.args, b.gx);
    else if (wide)
        SF_LAUNCH((pair_vis_small_kernel<1, 2, 1>), grid, block, 0, st, va, sp, nv, b.args, b.gx);
    else if (phase == 0)
        SF_LAUNCH((pair_vis_small_kernel<1, 8, 0>), grid, block, 0, st, va, sp, nv, b.args, b.gx);
    else if (phase == 1)
        SF_LAUNCH((pair_vis_small_kernel<1, 8, 1>), grid, block, 0, st, va, sp, nv, b.args, b.gx);
    else if (b.cpw == 8)
        SF_LAUNCH((pair_vis_small_kernel<1, 8, 2>), grid, block, 0, st, va, sp, nv, b.args, b.gx);
    else
        SF_LAUNCH((pair_vis_small_kernel<1, 4, 2>), grid, block, 0, st, va, sp, nv, b.args, b.gx);
    return launch_status();
}

}  // namespace sf
