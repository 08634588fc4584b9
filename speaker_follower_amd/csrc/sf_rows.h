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

__device__ __forceinline__ float4 pano_chunk(const PanoSrc& s, int b, int v, int chunk) {
    const int F4 = (s.IMG + s.LOC) >> 2;
    if (s.dense) return reinterpret_cast<const float4*>(s.dense)[((size_t)b * s.V + v) * F4 + chunk];
    const int vp = s.vp[b];
    if (vp < 0) return f4zero();
    const int I4 = s.IMG >> 2;
    if (chunk < I4)
        return reinterpret_cast<const float4*>(s.table)[((size_t)vp * s.V + v) * I4 + chunk];
    return reinterpret_cast<const float4*>(
        s.loc_table)[((size_t)s.view[b] * s.V + v) * (s.LOC >> 2) + (chunk - I4)];
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

__device__ __forceinline__ float4 cand_chunk(const CandSrc& s, int b, int a, int chunk) {
    const int F4 = (s.IMG + s.LOC) >> 2;
    if (s.dense) return reinterpret_cast<const float4*>(s.dense)[((size_t)b * s.A + a) * F4 + chunk];
    if (a == 0 || a >= s.a_num[b]) return f4zero();
    const int I4 = s.IMG >> 2;
    if (chunk < I4) {
        const int v = s.cand_view[(size_t)b * s.A + a];
        return reinterpret_cast<const float4*>(s.table)[((size_t)s.vp[b] * s.V + v) * I4 + chunk];
    }
    const int g = ((chunk - I4) << 2) / (s.LOC >> 2);       // LOC/4 is a multiple of 4
    const float val = s.cand_sincos[((size_t)b * s.A + a) * 4 + g];
    return make_float4(val, val, val, val);
}

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
