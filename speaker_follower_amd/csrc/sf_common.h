// Shared device/host helpers for the speaker/follower HIP hot path (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include <algorithm>
#include <string>

#include "../../include/sf_hip.h"

namespace sf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WAVE = 64;

#define SF_CHECK_ARG(cond) \
    do {                   \
        if (!(cond)) return SF_ERR_ARG; \
    } while (0)

// Launch-error check that does not synchronise (safe under stream capture).  The HIP "last
// error" is per host thread and sticky: other runtime users in the process (PyTorch's caching
// allocator polls events and leaves hipErrorNotReady behind) would otherwise be blamed on us, so
// every extern "C" entry point clears it first (SF_ENTER) and the code of a real failure is kept
// for sf_last_error_string().
extern thread_local hipError_t g_last_hip_error;
static inline void clear_stale_error() { (void)hipGetLastError(); }
static inline int launch_status() {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return SF_OK;
    g_last_hip_error = e;
    return SF_ERR_LAUNCH;
}
#define SF_ENTER() sf::clear_stale_error()

// ---------------------------------------------------------------------------------------------
// In-process kernel timing (sf_profile_begin / sf_profile_end, include/sf_hip.h).  While a host
// thread has it switched on, every launch it makes through SF_LAUNCH carries a start and a stop
// event on the dispatch itself (hipExtLaunchKernelGGL), so the elapsed time of a pair is that
// kernel's execution time on the stream it was launched on -- what rocprofv3 --kernel-trace
// reports, without a second process.  Off (the default) SF_LAUNCH is a plain launch.
// ---------------------------------------------------------------------------------------------
bool prof_active();
void prof_events(const char* name, hipEvent_t* e0, hipEvent_t* e1);
// SF_LAUNCH_AS: the same with an explicit name (a launch inside a template only sees "<MT, CPW>").
#define SF_LAUNCH(kernel, grid, block, shmem, st, ...) \
    SF_LAUNCH_AS(#kernel, kernel, grid, block, shmem, st, __VA_ARGS__)
#define SF_LAUNCH_AS(name, kernel, grid, block, shmem, st, ...)                                 \
    do {                                                                                         \
        hipEvent_t _e0 = nullptr, _e1 = nullptr;                                                 \
        if (sf::prof_active()) sf::prof_events(name, &_e0, &_e1); /* null: session just ended */ \
        if (_e0 && _e1) {                                                                        \
            hipExtLaunchKernelGGL(kernel, grid, block, shmem, st, _e0, _e1, 0, __VA_ARGS__);     \
        } else {                                                                                 \
            hipLaunchKernelGGL(kernel, grid, block, shmem, st, __VA_ARGS__);                     \
        }                                                                                        \
    } while (0)

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, WAVE));
    return v;
}

__device__ __forceinline__ float dot4(const float4 a, const float4 b) {
    return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---------------------------------------------------------------------------------------------
// Counter-based dropout (mirrored bit-for-bit by oracle/rng.py).  keep iff hash >= p * 2^32.
// `row` is the GLOBAL row id of the sample (so results do not depend on how a batch is sharded
// over ranks), `stream` identifies the dropout site and time step.
// ---------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint32_t fmix32(uint32_t h) {
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}

__host__ __device__ __forceinline__ uint32_t dropout_row_key(uint32_t seed, uint32_t stream,
                                                             uint32_t row) {
    uint32_t key = fmix32(seed + 0x9E3779B9u * stream);
    return fmix32(key ^ (row * 0x85EBCA6Bu));
}

__host__ __device__ __forceinline__ bool dropout_keep(uint32_t row_key, uint32_t col,
                                                      uint32_t thresh) {
    return fmix32(row_key + col * 0x9E3779B9u) >= thresh;
}

struct Dropout {            // p == 0 (thresh == 0) means "eval mode": everything kept, scale 1
    uint32_t seed;
    uint32_t stream;
    uint32_t thresh;        // p * 2^32
    float scale;            // 1 / (1 - p)
    int row0;               // global row id of local row 0
    // Device-side site offset (sf_dropout.site_dev, round 5): the site of this mask is stream + site_mul * *site.
    // `site` is never null (a device zero word when the caller gave none), so kernels read it unconditionally.  A
    // captured training iteration advances the word at its end: every replay draws fresh masks from the same graph.
    const uint32_t* site;
    uint32_t site_mul;
    __host__ __device__ bool on() const { return thresh != 0; }
};

const uint32_t* site_zero();        // a device word that holds 0 (sf_pointwise.hip)

// key of (seed, stream + mul * *site, row): seed + G * (stream + off) == (seed + G * off) + G * stream
// (a null site pointer reads as site 0: site_zero() returns null when the symbol lookup failed on this device)
__device__ __forceinline__ uint32_t site_value(const uint32_t* p) { return p ? *p : 0u; }
__device__ __forceinline__ uint32_t drop_seed(const Dropout& d) { return d.seed + 0x9E3779B9u * (d.site_mul * site_value(d.site)); }
__device__ __forceinline__ uint32_t drop_key(const Dropout& d, uint32_t row) {
    return dropout_row_key(drop_seed(d), d.stream, row);
}

// `mul`: how many stream ids one unit of the device-side site counter stands for at this call site -- 2 where the
// streams are numbered 2 * step + k (the decoder steps), the caller's sf_dropout.site_mul (default 1) where the entry
// point takes a raw stream id
static inline Dropout make_dropout(const sf_dropout* d, uint32_t stream, uint32_t mul = 0) {
    Dropout r;
    r.seed = d ? d->seed : 0;
    r.stream = stream;
    r.site = d && d->site_dev ? d->site_dev : site_zero();
    r.site_mul = mul ? mul : (d && d->site_mul ? d->site_mul : 1u);
    double p = d ? (double)d->p : 0.0;
    if (p <= 0.0) {
        r.thresh = 0;
        r.scale = 1.0f;
    } else {
        double t = p * 4294967296.0;
        r.thresh = t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
        r.scale = (float)(1.0 / (1.0 - p));
    }
    r.row0 = d ? d->row0 : 0;
    return r;
}

}  // namespace sf
