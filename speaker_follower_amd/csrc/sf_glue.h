// Follower per-step glue (follower.py:476-505) as a wave-level device function, shared by the
// stand-alone glue kernel and the fused scoring+glue kernel.
#pragma once
#include "sf_rows.h"

namespace sf {

// env.step + env.observe + shortest-path teacher of one (sample, candidate slot) of a device-resident
// environment (env.py:126-146, 628-641, 742-761, 763-804 over the tabulated panorama sweep): the body of
// sf_nav_step, also run by the scoring + glue kernel right behind the action choice.
struct NavIO {
    sf_nav_table nav;
    const int32_t* row; const int32_t* view;      // [B] current state
    const int32_t* goal_hop; int ld_hop;          // [B, ld_hop] next nav row towards the goal, by LOCAL row
    const int32_t* hop_base;                      // [B] first nav row of the sample's scan
    int32_t* row_next; int32_t* vp_next; int32_t* view_next; int32_t* a_num_next;
    int32_t* cand_view_next; float* sincos_next;  // [B,A], [B,A,4]
    int64_t* target_next;                         // [B] or null
    bool on;
};
// act < 0: no move (the initial observation); ended: the row's flag AFTER this step
__device__ __forceinline__ void nav_advance_slot(const NavIO& p, int b, int a, int act, bool ended) {
    const int A = p.nav.A, V = p.nav.V;
    int row = p.row[b], view = p.view[b];
    if (act >= 0) {                               // env.py:126-146: stop, or a candidate that is the
        const int s0 = row * V + view;            // current viewpoint itself, leaves the state alone
        act = act >= p.nav.a_num[s0] ? 0 : act;
        const int nr = p.nav.next_row[(size_t)s0 * A + act];
        if (act != 0 && nr != row) {
            view = p.nav.cand_view[(size_t)s0 * A + act];
            row = nr;
        }
    }
    const size_t s = (size_t)row * V + view;
    const int n = p.nav.a_num[s];
    p.cand_view_next[(size_t)b * A + a] = a < n ? p.nav.cand_view[s * A + a] : 0;
    const float4 sc = a < n ? reinterpret_cast<const float4*>(p.nav.cand_sincos)[s * A + a]
                            : make_float4(0.f, 1.f, 0.f, 1.f);
    reinterpret_cast<float4*>(p.sincos_next)[(size_t)b * A + a] = sc;
    if (a != 0) return;
    p.row_next[b] = row;
    p.vp_next[b] = p.nav.feat_row[row];
    p.view_next[b] = view;
    p.a_num_next[b] = n;
    if (p.target_next) {
        long tgt = -1;                            // follower.py:322-328: -1 once ended
        if (!ended) {
            const int hop = p.goal_hop[(size_t)b * p.ld_hop + (row - p.hop_base[b])];
            tgt = 0;                              // at the goal: stop (env.py:744-745)
            if (hop != row)
                for (int c = 1; c < n; ++c)
                    if (p.nav.next_row[s * A + c] == hop) { tgt = c; break; }
        }
        p.target_next[b] = tgt;
    }
}

struct FGlue {
    CandSrc src;
    int B;
    float* logit;            // [B,A] masked in place
    const float* is_valid;   // [B,A] or null
    const int64_t* target;
    int feedback;            // 0 teacher, 1 argmax, 2 sample
    uint8_t* ended;
    int64_t* a_t;
    int64_t* target_used;
    float* score;
    float* u_next;           // [B, ld_u] or null
    int ld_u;                // row stride of u_next in floats
    Dropout u_drop;          // dropout applied to u_next (the next step's LSTM input), cols 0..F-1
    float* ce_term;          // [B]
    float* live;             // [B]
    uint32_t sample_seed;    // feedback 2: counter-based uniform per (seed, stream, row)
    uint32_t sample_stream;
    const uint32_t* sample_site;   // device-side stream offset (never null), see Dropout.site
    int row0;
    NavIO nav;               // nav.on: env step of a device-resident environment behind the action choice
};

// The glue's inputs, loaded up front with straight-line code (callers issue this BEFORE they wait
// for the logits, so the loads overlap with the scoring work).
struct FGlueIn {
    bool valid;       // lane's candidate is a real one
    bool was_ended;
    int64_t target;
};
__device__ __forceinline__ FGlueIn follower_glue_load(const FGlue& g, int b) {
    const int lane = threadIdx.x & 63;
    const int A = g.src.A;
    const int la = min(lane, A - 1);
    FGlueIn in;
    if (g.is_valid)                                              // block-uniform
        in.valid = lane < A && g.is_valid[(size_t)b * A + la] != 0.f;
    else
        in.valid = lane < A && lane < g.src.a_num[b];
    in.was_ended = g.ended[b] != 0;
    in.target = g.target[b];
    return in;
}

// One wave handles sample b; lane a holds the raw logit of candidate a (lanes >= A ignored).
// Returns the chosen action (wave-uniform).
__device__ __forceinline__ int follower_glue_row(const FGlue& g, int b, float raw, const FGlueIn& in) {
    const int lane = threadIdx.x & 63;
    const int A = g.src.A;
    const bool valid = in.valid;
    const float l = (lane < A && valid) ? raw : -INFINITY;
    if (lane < A) g.logit[(size_t)b * A + lane] = l;             // follower.py:477
    const float m = wave_max(l);
    const float e = (lane < A && valid) ? expf(l - m) : 0.f;
    const float se = wave_sum(e);
    const float lse = m + logf(se);
    const bool was_ended = in.was_ended;
    const int64_t tgt = was_ended ? -1 : in.target;              // follower.py:322-328
    const float lt = __shfl(l, tgt >= 0 ? (int)tgt : 0, WAVE);
    const float ce = tgt >= 0 ? (lse - lt) : 0.f;                // CrossEntropyLoss(ignore_index=-1)
    int at;
    if (g.feedback == 0) {
        at = tgt > 0 ? (int)tgt : 0;                             // follower.py:486
    } else if (g.feedback == 1) {
        const unsigned long long hit = __ballot(lane < A && l == m);
        at = hit ? (int)__ffsll((long long)hit) - 1 : 0;         // first maximum, follower.py:488
    } else {
        // follower.py:491-497: sample from softmax(logit) (invalid candidates have probability 0).
        // Inverse CDF over <= 64 lanes with a counter-based uniform (the reference's torch RNG
        // stream cannot be reproduced; parity is defined on teacher / argmax).
        const uint32_t key = dropout_row_key(g.sample_seed + 0x9E3779B9u * site_value(g.sample_site), g.sample_stream, (uint32_t)(g.row0 + b));
        const float u = (float)(fmix32(key) >> 8) * (1.0f / 16777216.0f) * se;
        float cdf = e;                                           // inclusive prefix sum over lanes
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const float v = __shfl_up(cdf, off, WAVE);
            if (lane >= off) cdf += v;
        }
        const unsigned long long hit = __ballot(lane < A && valid && cdf > u);
        const unsigned long long any = __ballot(lane < A && valid);
        at = hit ? (int)__ffsll((long long)hit) - 1 : (63 - __clzll((long long)any));
    }
    const float la = __shfl(l, at, WAVE);
    if (lane == 0) {
        g.a_t[b] = at;
        g.target_used[b] = tgt;
        g.score[b] = la - lse;                                   // follower.py:504 (per-step term)
        g.ce_term[b] = ce;
        g.live[b] = tgt >= 0 ? 1.f : 0.f;
        g.ended[b] = (was_ended || at == 0) ? 1 : 0;             // follower.py:527-530
    }
    return at;
}

// u_next[b, 4c..4c+3] = dropout(chunk) for the chosen action's row (follower.py:502 + model.py:392)
__device__ __forceinline__ void store_u_next(const FGlue& g, int b, int c, float4 v) {
    if (g.u_drop.on()) {
        const uint32_t rk = drop_key(g.u_drop, (uint32_t)(g.u_drop.row0 + b));
        const uint32_t col = (uint32_t)(4 * c);
        v.x = dropout_keep(rk, col + 0, g.u_drop.thresh) ? v.x * g.u_drop.scale : 0.f;
        v.y = dropout_keep(rk, col + 1, g.u_drop.thresh) ? v.y * g.u_drop.scale : 0.f;
        v.z = dropout_keep(rk, col + 2, g.u_drop.thresh) ? v.z * g.u_drop.scale : 0.f;
        v.w = dropout_keep(rk, col + 3, g.u_drop.thresh) ? v.w * g.u_drop.scale : 0.f;
    }
    *reinterpret_cast<float4*>(g.u_next + (size_t)b * g.ld_u + 4 * c) = v;
}

}  // namespace sf
