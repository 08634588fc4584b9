// fp32 products on the BF16 matrix cores at fp32 accuracy (round 4): error-free three-way operand splitting.
//
// v_mfma_f32_16x16x4_f32 runs at the f32 VECTOR rate (157 TFLOP/s dense, 1/16 of the bf16 matrix rate).  Every fp32
// value is the EXACT sum of three bf16 pieces (round-to-nearest splits: a1 = bf16(a), a2 = bf16(a - a1),
// a3 = a - a1 - a2; |a2| <= 2^-9 |a|, |a3| <= 2^-17 |a|), so
//     a b = a1 b1 + (a1 b2 + a2 b1) + (a2 b2 + a1 b3 + a3 b1) + O(2^-25 |a b|)
// is six v_mfma_f32_16x16x32_bf16 (exact bf16 x bf16 products, fp32 accumulate): 6/16 of the fp32-MFMA time.  The
// leading product keeps its own accumulator, the five small ones share a second one (one accumulator for all six
// costs the leading sum a rounding per small term: measured 3x the error and a 1e-8 bias).  Measured against float64
// on the gate-product shape [100 x 4864] x [2048 x 4864]^T (tools/exp/bf16x6_accuracy.hip): rms error 6.0e-7 / max
// 6.6e-6, against 1.6e-6 / 2.0e-5 for the fp32 MFMA's k-ordered fma chain (one rounding per 16 products instead of
// one per product): NOT bit-identical to the fp32 MFMA, closer to the exact sum.  Three pieces with only three products
// (the usual "bf16x3") are 5x WORSE than fp32 and are not used anywhere.
#pragma once
#include "sf_common.h"

namespace sf {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pk_bf16_rn(float lo, float hi) {        // v_cvt_pk_bf16_f32 (round to nearest even)
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
// float4 -> three planes of 4 bf16 (uint2 each): exact three-way split
__device__ __forceinline__ void split3_f4(const float4& v, uint2& p1, uint2& p2, uint2& p3) {
    p1.x = pk_bf16_rn(v.x, v.y);
    p1.y = pk_bf16_rn(v.z, v.w);
    const float r0 = v.x - __uint_as_float(p1.x << 16), r1 = v.y - __uint_as_float(p1.x & 0xFFFF0000u);
    const float r2 = v.z - __uint_as_float(p1.y << 16), r3 = v.w - __uint_as_float(p1.y & 0xFFFF0000u);
    p2.x = pk_bf16_rn(r0, r1);
    p2.y = pk_bf16_rn(r2, r3);
    const float q0 = r0 - __uint_as_float(p2.x << 16), q1 = r1 - __uint_as_float(p2.x & 0xFFFF0000u);
    const float q2 = r2 - __uint_as_float(p2.y << 16), q3 = r3 - __uint_as_float(p2.y & 0xFFFF0000u);
    p3.x = pk_bf16_rn(q0, q1);            // (exact: q has <= 8 significant bits)
    p3.y = pk_bf16_rn(q2, q3);
}
// Eight fp32 values of one lane (two float4: the K order inside an MFMA is free as long as A and B agree) -> the
// three bf16x8 operands of v_mfma_f32_16x16x32_bf16
struct Split8 {
    bf16x8 p[3];
};
__device__ __forceinline__ Split8 split3_f8(const float4& x, const float4& y) {
    uint2 x1, x2, x3, y1, y2, y3;
    split3_f4(x, x1, x2, x3);
    split3_f4(y, y1, y2, y3);
    Split8 s;
    s.p[0] = __builtin_bit_cast(bf16x8, uint4{x1.x, x1.y, y1.x, y1.y});
    s.p[1] = __builtin_bit_cast(bf16x8, uint4{x2.x, x2.y, y2.x, y2.y});
    s.p[2] = __builtin_bit_cast(bf16x8, uint4{x3.x, x3.y, y3.x, y3.y});
    return s;
}
// hi += a1 b1;  lo += a1 b2 + a2 b1 + a2 b2 + a1 b3 + a3 b1   (small terms first inside lo)
__device__ __forceinline__ void mfma_split6(const Split8& a, const Split8& b, f32x4& hi, f32x4& lo) {
    lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[2], b.p[0], lo, 0, 0, 0);
    lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[0], b.p[2], lo, 0, 0, 0);
    lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[1], b.p[1], lo, 0, 0, 0);
    lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[1], b.p[0], lo, 0, 0, 0);
    lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[0], b.p[1], lo, 0, 0, 0);
    hi = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[0], b.p[0], hi, 0, 0, 0);
}

// The same for N independent accumulator pairs that share the A operand (N weight tiles against one activation
// fragment), product by product ACROSS the tiles: consecutive MFMAs never wait for each other's accumulator.
template <int N, typename BOf>
__device__ __forceinline__ void mfma_split6_across(const Split8& a, BOf b_of, f32x4 (&hi)[N], f32x4 (&lo)[N]) {
#pragma unroll
    for (int n = 0; n < N; ++n) lo[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[2], b_of(n).p[0], lo[n], 0, 0, 0);
#pragma unroll
    for (int n = 0; n < N; ++n) lo[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[0], b_of(n).p[2], lo[n], 0, 0, 0);
#pragma unroll
    for (int n = 0; n < N; ++n) lo[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[1], b_of(n).p[1], lo[n], 0, 0, 0);
#pragma unroll
    for (int n = 0; n < N; ++n) lo[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[1], b_of(n).p[0], lo[n], 0, 0, 0);
#pragma unroll
    for (int n = 0; n < N; ++n) lo[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[0], b_of(n).p[1], lo[n], 0, 0, 0);
#pragma unroll
    for (int n = 0; n < N; ++n) hi[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[0], b_of(n).p[0], hi[n], 0, 0, 0);
}

}  // namespace sf
