// Lab harness for the visual-attention kernels (not part of the product).  The timestamp hooks
// (LAB_VSTAMP) were removed from the product source; re-insert them locally to use SF_LAB_STAMP.
#ifdef SF_LAB_STAMP
#define LAB_VSTAMP(i) do { if (threadIdx.x == 0) g_vstamps[blockIdx.y * 4 + blockIdx.x][i] = wall_clock64(); } while (0)
__device__ long long g_vstamps[1024][8];
#else
#define LAB_VSTAMP(i) do {} while (0)
#endif
#include "../../speaker_follower_amd/csrc/sf_attention.hip"
#include <cstdio>
#include <vector>
#include <algorithm>
namespace sf { thread_local hipError_t g_last_hip_error = hipSuccess; }
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 100, NVP = argc > 3 ? atoi(argv[3]) : 512, V = 36, IMG = 2048, LOC = 128, F = IMG + LOC;
    const int split = argc > 2 ? atoi(argv[2]) : 1, reps = 300;
    float *table, *loc, *q, *alpha, *out, *part; int *vp, *view; unsigned* cnt;
    hipMalloc(&table, (size_t)NVP * V * IMG * 4); hipMalloc(&loc, V * V * LOC * 4); hipMalloc(&q, B * F * 4);
    hipMalloc(&alpha, B * V * 4); hipMalloc(&out, B * F * 4); hipMalloc(&part, sf::visual_attn_split_floats(B, F) * 4);
    hipMalloc(&vp, B * 4); hipMalloc(&view, B * 4); hipMalloc(&cnt, 4096); hipMemset(cnt, 0, 4096);
    std::vector<float> h((size_t)512 * V * IMG); unsigned s = 1;
    for (auto& x : h) { s = s * 1664525u + 1013904223u; x = (s >> 8) * (1.f / 16777216.f); }
    for (int r0 = 0; r0 < NVP; r0 += 512)      // replicate 512 random viewpoints over the table
        hipMemcpy(table + (size_t)r0 * V * IMG, h.data(), (size_t)std::min(512, NVP - r0) * V * IMG * 4, hipMemcpyHostToDevice);
    hipMemcpy(loc, h.data(), V * V * LOC * 4, hipMemcpyHostToDevice);
    for (int i = 0; i < B * F; ++i) h[i] = 0.01f * ((i * 7919) % 101 - 50);
    hipMemcpy(q, h.data(), B * F * 4, hipMemcpyHostToDevice);
    std::vector<int> iv(B), vw(B);
    for (int b = 0; b < B; ++b) { iv[b] = (int)(((long)b * 7919 * 131) % NVP); vw[b] = b % 36; }
    hipMemcpy(vp, iv.data(), B * 4, hipMemcpyHostToDevice); hipMemcpy(view, vw.data(), B * 4, hipMemcpyHostToDevice);
    sf::PanoSrc src{nullptr, table, loc, vp, view, V, IMG, LOC};
    sf::Dropout dr{}; 
    hipStream_t st; hipStreamCreate(&st);
    auto run = [&]() { return sf::visual_attn(0, src, B, q, F, alpha, out, F, dr, 0, st, split ? part : nullptr, split ? cnt : nullptr); };
    for (int i = 0; i < 10; ++i) run();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipStreamSynchronize(st); hipEventRecord(e0, st);
    for (int i = 0; i < reps; ++i) run();
    hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<float> o(8); hipMemcpy(o.data(), out, 32, hipMemcpyDeviceToHost);
    printf("B=%d split=%d  %.2f us/launch  out0=%g %g  %s\n", B, split, ms * 1e3 / reps, o[0], o[5], hipGetErrorString(hipGetLastError()));
#ifdef SF_LAB_STAMP
    static long long stp[512][8];
    hipMemcpyFromSymbol(stp, HIP_SYMBOL(g_vstamps), sizeof(stp));
    long long t0 = stp[0][0]; for (int i = 0; i < 2 * B; ++i) t0 = stp[i][0] < t0 ? stp[i][0] : t0;
    for (int i = 0; i < 2 * B; i += 23) { printf("blk %3d:", i); for (int k = 0; k < 8; ++k) printf(" %5lld", stp[i][k] ? stp[i][k] - t0 : -1); printf("\n"); }
#endif
    return 0;
}
