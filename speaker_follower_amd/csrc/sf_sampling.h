// `sample` feedback of the speaker (speaker.py:170-174: probs = softmax(logit); D.Categorical(probs).sample()).
// The reference draws from torch's stateful generator, whose stream cannot be reproduced; the draw here is
// counter-based -- a pure function of (seed, stream = site + word step, GLOBAL row id) like the dropout masks and
// the follower's sampler (sf_glue.h) -- so it does not depend on how a batch is sharded, and it is the SAME
// two-level inverse-CDF draw in the per-step glue kernel (sf_pointwise.hip) and in the persistent word loop
// (sf_persist.hip), where the vocabulary is spread over 32 workgroups:
//   slot s = columns [32 s, 32 s + 32) of the vocabulary;  m_s = max, z_s = sum exp(l - m_s) over its columns
//   level 1 (uniform u1): the first slot whose inclusive prefix of  z_s exp(m_s - M)  exceeds  u1 * Z
//   level 2 (uniform u2): inside that slot, the first column whose inclusive prefix of exp(l - m_s) exceeds u2 * z_s
// P(column c of slot s) = P(s) P(c | s) = softmax(l)_c.  Fallbacks when a threshold rounds up to the total: the slot
// of the arg max; the last column of the slot.  oracle/rng.py mirrors the draw in float64.
#pragma once
#include "sf_common.h"

namespace sf {

__device__ __forceinline__ float wexp(float m, float mm) { return m == -INFINITY ? 0.f : expf(m - mm); }

__device__ __forceinline__ void sample_uniforms(uint32_t seed, uint32_t stream, uint32_t row, float* u1, float* u2) {
    const uint32_t key = dropout_row_key(seed, stream, row);
    *u1 = (float)(fmix32(key) >> 8) * (1.0f / 16777216.0f);                    // = the follower's uniform (sf_glue.h)
    *u2 = (float)(fmix32(key + 0x9E3779B9u) >> 8) * (1.0f / 16777216.0f);     // = dropout hash of column 1
}

}  // namespace sf
