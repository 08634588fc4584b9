// Row-set helpers shared by the three soft attentions on the path.
//
// All three (visual attention over 36 x 2176 panorama rows, text attention over <=80 x 512
// instruction-context rows, candidate scoring over <=16 x 2176 action rows) have the same
// shape: one workgroup per sample keeps the sample's whole row set in REGISTERS (each wave
// owns RPW rows, each lane CPL float4 chunks of every row, loaded with fully coalesced 1 KiB
// wave reads issued up front), does  dot(row, vec) -> wave-shuffle reduction -> softmax-like
// weights -> weighted row sum,  and only the per-wave partial sums cross waves through LDS.
// The row set is read from HBM exactly once per pass.
#pragma once
#include "sf_common.h"

namespace sf {

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void f4fma(float4& acc, float w, const float4& x) {
    acc.x += w * x.x; acc.y += w * x.y; acc.z += w * x.z; acc.w += w * x.w;
}
__device__ __forceinline__ void f4add(float4& acc, const float4& x) {
    acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
}

// Where a sample's panorama rows come from: a dense [B,V,F] tensor (what the reference's
// _feature_variables builds, follower.py:291-298) or the HBM-resident feature table plus the
// 36x36x128 location-embedding table, addressed by (viewpoint row, agent view index)
// (env.py:380-383, 771-773, 78-101).
struct PanoSrc {
    const float* dense;      // [B, V, F] or null
    const float* table;      // [n_vp, V, IMG]
    const float* loc_table;  // [V, V, LOC]   (agent view index, absolute view, :)
    const int* vp;           // [B] row into table, < 0 => all-zero panorama (padded speaker step)
    const int* view;         // [B] agent view index
    int V, IMG, LOC;
};

// Per-sample view of the panorama rows.  The row loaders below are STRAIGHT-LINE code (pointer
// selects, one unconditional load, value select): a load inside a branch makes the compiler end
// the block with `s_waitcnt vmcnt(0)`, which serialised the 27 loads per lane of the attention
// kernels into 27 memory round trips (~10 us of a 21 us kernel).
struct PanoRow {
    const float4* img;   // rows of I4 float4 (dense mode: the whole F4-wide row, I4 == F4)
    const float4* loc;   // rows of L4 float4
    int I4, L4;
    bool zero;           // padded speaker step: all-zero panorama
};

__device__ __forceinline__ PanoRow pano_row(const PanoSrc& s, int b) {
    PanoRow r;
    const int F4 = (s.IMG + s.LOC) >> 2;
    if (s.dense) {       // block-uniform
        r.img = reinterpret_cast<const float4*>(s.dense) + (size_t)b * s.V * F4;
        r.loc = r.img;
        r.I4 = F4;
        r.L4 = 0;
        r.zero = false;
    } else {
        const int vp = s.vp[b];
        r.zero = vp < 0;
        r.I4 = s.IMG >> 2;
        r.L4 = s.LOC >> 2;
        r.img = reinterpret_cast<const float4*>(s.table) + (size_t)max(vp, 0) * s.V * r.I4;
        r.loc = reinterpret_cast<const float4*>(s.loc_table) + (size_t)s.view[b] * s.V * r.L4;
    }
    return r;
}

// chunk `chunk` of row `v`; `ok` false (row / chunk out of range) or a zero panorama gives zeros.
// Indices are clamped, never branched on.
__device__ __forceinline__ float4 pano_load(const PanoRow& r, int v, int chunk, bool ok, int V,
                                            int n4) {
    v = min(v, V - 1);
    chunk = min(chunk, n4 - 1);
    const float4* p = chunk < r.I4 ? r.img + (size_t)v * r.I4 + chunk
                                   : r.loc + (size_t)v * r.L4 + (chunk - r.I4);
    const float4 x = *p;
    return (ok && !r.zero) ? x : f4zero();
}

__device__ __forceinline__ float4 pano_chunk(const PanoSrc& s, int b, int v, int chunk) {
    const PanoRow r = pano_row(s, b);
    return pano_load(r, v, chunk, true, s.V, (s.IMG + s.LOC) >> 2);
}

// Candidate-action rows: dense [B,A,F] (follower.py:300-320) or, by index, row `cand_view` of
// the agent's current panorama followed by sin/cos of the candidate's relative heading and
// elevation, each repeated LOC/4 times (env.py:60-75).  Candidate 0 is "stop": all zeros.
struct CandSrc {
    const float* dense;       // [B, A, F] or null
    const float* table;       // [n_vp, V, IMG]
    const int* vp;            // [B]
    const int* cand_view;     // [B, A]
    const float* cand_sincos; // [B, A, 4] = sin h, cos h, sin e, cos e (host float64 -> fp32)
    const int* a_num;         // [B] number of real candidates (incl. stop); rows >= a_num are zero
    int A, V, IMG, LOC;
};

// Per-(sample, candidate) view of a candidate row; same straight-line discipline as PanoRow.
struct CandRow {
    const float4* img;   // I4 float4 of image features (dense mode: the whole row, I4 == F4)
    float s0, s1, s2, s3;  // sin h, cos h, sin e, cos e (scalars: a float4 indexed per lane would
                           // be spilled to LDS by the compiler)
    int I4, g4;          // g4 = float4 per sin/cos group (LOC/16)
    bool zero;           // stop action, padding candidate or padded sample
};

__device__ __forceinline__ CandRow cand_row(const CandSrc& s, int b, int a) {
    CandRow r;
    a = min(max(a, 0), s.A - 1);
    if (s.dense) {       // block-uniform
        const int F4 = (s.IMG + s.LOC) >> 2;
        r.img = reinterpret_cast<const float4*>(s.dense) + ((size_t)b * s.A + a) * F4;
        r.s0 = r.s1 = r.s2 = r.s3 = 0.f;
        r.I4 = F4;
        r.g4 = 1;
        r.zero = false;
    } else {
        const int vp = s.vp[b];
        const int view = s.cand_view[(size_t)b * s.A + a];
        const float4 sc = reinterpret_cast<const float4*>(s.cand_sincos)[(size_t)b * s.A + a];
        r.zero = a == 0 || a >= s.a_num[b] || vp < 0;
        r.I4 = s.IMG >> 2;
        r.g4 = max(s.LOC >> 4, 1);
        r.img = reinterpret_cast<const float4*>(s.table) +
                ((size_t)max(vp, 0) * s.V + min(max(view, 0), s.V - 1)) * r.I4;
        r.s0 = sc.x; r.s1 = sc.y; r.s2 = sc.z; r.s3 = sc.w;
    }
    return r;
}

// chunk `chunk` of the row.  The load is unconditional on a clamped index and the result is
// BLENDED arithmetically (x * 1 or x * 0): behind a select the compiler sinks the load into a
// branch and every chunk costs its own memory round trip.  Feature values are finite, so x * 0 = 0.
// A zero row (r.zero) is the caller's business (skip the loads with one wave-uniform branch).
__device__ __forceinline__ float4 cand_load(const CandRow& r, int chunk, bool ok, int n4) {
    chunk = min(chunk, n4 - 1);
    const float4 x = r.img[min(chunk, r.I4 - 1)];
    const int dl = chunk - r.I4;                                 // location chunk -> sin/cos group
    // (bit masks, not a select chain: the compiler turns a chain over s0..s3 into an indexed
    // load from a struct it first spills to LDS)
    const int grp = (dl >= r.g4) + (dl >= 2 * r.g4) + (dl >= 3 * r.g4);
    const uint32_t vb = (__float_as_uint(r.s0) & (grp == 0 ? ~0u : 0u)) |
                        (__float_as_uint(r.s1) & (grp == 1 ? ~0u : 0u)) |
                        (__float_as_uint(r.s2) & (grp == 2 ? ~0u : 0u)) |
                        (__float_as_uint(r.s3) & (grp == 3 ? ~0u : 0u));
    const float v = __uint_as_float(vb);
    const float mi = (ok && chunk < r.I4) ? 1.f : 0.f;
    const float ml = (ok && chunk >= r.I4) ? v : 0.f;
    return make_float4(x.x * mi + ml, x.y * mi + ml, x.z * mi + ml, x.w * mi + ml);
}

__device__ __forceinline__ float4 cand_chunk(const CandSrc& s, int b, int a, int chunk) {
    const CandRow r = cand_row(s, b, a);
    return cand_load(r, chunk, a >= 0 && a < s.A && !r.zero, (s.IMG + s.LOC) >> 2);
}

// Where the scoring backward takes d(logit) from when it forms it itself: softmax(logit) - onehot
// (CrossEntropyLoss(ignore_index) backward, follower.py:278, 481), scaled by gscale[0].
struct CeSrc {
    const float* logit;      // [B, ld] masked logits of the step (-inf on padding candidates)
    const int64_t* target;   // [B]
    const float* gscale;     // [1] 1 / (live rows of the step)
    int ignore;              // target value of rows that do not count
    int ld;
};

// Sum per-wave partial rows (CPL float4 per lane) over the NW waves of the block.  Waves fold
// into waves [0, SLOTS) through SLOTS LDS slots, then thread `c` gets chunk c's total via
// `emit(c, total)`.  Every thread of the block must call this (it synchronises).
template <int CPL, int NW, int SLOTS, typename Emit>
__device__ __forceinline__ void block_row_sum(float4 (&p)[CPL], float4 (*slots)[CPL * 64],
                                              int n4, Emit emit) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = SLOTS; base < NW; base += SLOTS) {
        if (wave >= base && wave < base + SLOTS) {
#pragma unroll
            for (int i = 0; i < CPL; ++i) slots[wave - base][lane + 64 * i] = p[i];
        }
        __syncthreads();
        if (wave < SLOTS && wave + base < NW) {
#pragma unroll
            for (int i = 0; i < CPL; ++i) f4add(p[i], slots[wave][lane + 64 * i]);
        }
        __syncthreads();
    }
    constexpr int LIVE = NW < SLOTS ? NW : SLOTS;
    if (wave < LIVE) {
#pragma unroll
        for (int i = 0; i < CPL; ++i) slots[wave][lane + 64 * i] = p[i];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < n4; c += NW * 64) {
        float4 t = slots[0][c];
#pragma unroll
        for (int w = 1; w < LIVE; ++w) f4add(t, slots[w][c]);
        emit(c, t);
    }
}

}  // namespace sf
