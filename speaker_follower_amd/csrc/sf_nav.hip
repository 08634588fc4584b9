// Device-resident navigation: env.step + env.observe + the shortest-path teacher of a whole batch as
// ONE small launch per decode step (reference: R2RBatch.step / observe / _shortest_path_action,
// tasks/R2R/env.py:628-641, 742-761, 763-804; the panorama sweep behind the candidate lists,
// env.py:149-224, is a pure function of (viewpoint, view index) and is tabulated once on the host).
// With it a student-forced rollout on REAL connectivity graphs needs no host round trip per step:
// the glue kernel leaves a_t on the device, this kernel turns (state, a_t) into the next state and
// writes the next step's index-form observation where the decoder kernels read it.
#include "sf_kernels.h"

namespace sf {
namespace {

struct NavArgs {
    sf_nav_table nav;
    int B;
    const int32_t* row; const int32_t* view;      // [B] current state
    const int64_t* a_t;                           // [B] chosen candidate, or null (initial observation)
    const uint8_t* ended;                         // [B] after this step (null = nobody ended)
    const int32_t* goal_hop; int ld_hop;          // [B, ld_hop] next nav row towards the goal, by LOCAL row
    const int32_t* hop_base;                      // [B] first nav row of the sample's scan
    int32_t* row_next; int32_t* vp_next; int32_t* view_next; int32_t* a_num_next;
    int32_t* cand_view_next; float* sincos_next;  // [B,A], [B,A,4]
    int64_t* target_next;                         // [B] or null
};

// one thread per (sample, candidate slot)
__global__ __launch_bounds__(256) void nav_step_kernel(NavArgs p) {
    const int A = p.nav.A, V = p.nav.V;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.B * A) return;
    const int b = i / A, a = i - b * A;
    int row = p.row[b], view = p.view[b];
    if (p.a_t) {                                  // env.py:126-146: stop, or a candidate that is the
        const int s0 = row * V + view;            // current viewpoint itself, leaves the state alone
        int act = (int)p.a_t[b];
        act = act < 0 ? 0 : (act >= p.nav.a_num[s0] ? 0 : act);
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
        if (!(p.ended && p.ended[b])) {
            const int hop = p.goal_hop[(size_t)b * p.ld_hop + (row - p.hop_base[b])];
            tgt = 0;                              // at the goal: stop (env.py:744-745)
            if (hop != row)
                for (int c = 1; c < n; ++c)
                    if (p.nav.next_row[s * A + c] == hop) { tgt = c; break; }
        }
        p.target_next[b] = tgt;
    }
}

}  // namespace

int nav_step(const sf_nav_table* nav, int B, const int32_t* row, const int32_t* view, const int64_t* a_t,
             const uint8_t* ended, const int32_t* goal_hop, int ld_hop, const int32_t* hop_base,
             int32_t* row_next, int32_t* vp_next, int32_t* view_next, int32_t* a_num_next,
             int32_t* cand_view_next, float* sincos_next, int64_t* target_next, hipStream_t st) {
    NavArgs p{*nav, B, row, view, a_t, ended, goal_hop, ld_hop, hop_base, row_next, vp_next, view_next,
              a_num_next, cand_view_next, sincos_next, target_next};
    SF_LAUNCH(nav_step_kernel, dim3(ceil_div(B * nav->A, 256)), dim3(256), 0, st, p);
    return launch_status();
}

}  // namespace sf

extern "C" int sf_nav_step(const sf_nav_table* nav, int B, const int32_t* row, const int32_t* view,
                           const int64_t* a_t, const uint8_t* ended, const int32_t* goal_hop, int ld_hop,
                           const int32_t* hop_base, int32_t* row_next, int32_t* vp_next, int32_t* view_next,
                           int32_t* a_num_next, int32_t* cand_view_next, float* sincos_next,
                           int64_t* target_next, sf_stream stream) {
    SF_ENTER();
    SF_CHECK_ARG(nav && nav->a_num && nav->next_row && nav->cand_view && nav->cand_sincos && nav->feat_row &&
                 nav->A > 0 && nav->V > 0 && B > 0 && row && view && row_next && vp_next && view_next &&
                 a_num_next && cand_view_next && sincos_next && (!target_next || (goal_hop && hop_base && ld_hop > 0)));
    return sf::nav_step(nav, B, row, view, a_t, ended, goal_hop, ld_hop, hop_base, row_next, vp_next, view_next,
                        a_num_next, cand_view_next, sincos_next, target_next, (hipStream_t)stream);
}
