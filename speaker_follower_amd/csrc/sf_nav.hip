// Device-resident navigation: env.step + env.observe + the shortest-path teacher of a whole batch as
// ONE small launch per decode step (reference: R2RBatch.step / observe / _shortest_path_action,
// tasks/R2R/env.py:628-641, 742-761, 763-804; the panorama sweep behind the candidate lists,
// env.py:149-224, is a pure function of (viewpoint, view index) and is tabulated once on the host).
// With it a student-forced rollout on REAL connectivity graphs needs no host round trip per step:
// the glue kernel leaves a_t on the device, this kernel turns (state, a_t) into the next state and
// writes the next step's index-form observation where the decoder kernels read it.
#include "sf_kernels.h"
#include "sf_glue.h"

namespace sf {
namespace {

// one thread per (sample, candidate slot)
__global__ __launch_bounds__(256) void nav_step_kernel(NavIO p, int B, const int64_t* a_t, const uint8_t* ended) {
    const int A = p.nav.A;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * A) return;
    const int b = i / A, a = i - b * A;
    int act = -1;
    if (a_t) {
        act = (int)a_t[b];
        act = act < 0 ? 0 : act;
    }
    nav_advance_slot(p, b, a, act, ended && ended[b]);
}

}  // namespace

int nav_step(const sf_nav_table* nav, int B, const int32_t* row, const int32_t* view, const int64_t* a_t,
             const uint8_t* ended, const int32_t* goal_hop, int ld_hop, const int32_t* hop_base,
             int32_t* row_next, int32_t* vp_next, int32_t* view_next, int32_t* a_num_next,
             int32_t* cand_view_next, float* sincos_next, int64_t* target_next, hipStream_t st) {
    NavIO p{*nav, row, view, goal_hop, ld_hop, hop_base, row_next, vp_next, view_next, a_num_next, cand_view_next,
            sincos_next, target_next, true};
    SF_LAUNCH(nav_step_kernel, dim3(ceil_div(B * nav->A, 256)), dim3(256), 0, st, p, B, a_t, ended);
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
