// Lab harness: times sf::linear_nt on the decoder-LSTM-gate shape with hipEvents, for kernel
// variants selected at compile time (-DSF_LAB_...).  Not part of the product.
#ifdef SF_LAB_SNAPSHOT
#include "sf_gemm_lab_snapshot.hip"   // instrumented copy with the SF_LAB_* switches and the "wide" kernel
#else
#include "../../speaker_follower_amd/csrc/sf_gemm.hip"
#endif
#include <cstdio>
namespace sf { thread_local hipError_t g_last_hip_error = hipSuccess; }
#include <vector>
int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 100, N = 2048, K1 = 4352, K2 = 512;
    const int reps = argc > 2 ? atoi(argv[2]) : 200;
    float *x, *h, *wi, *wh, *y, *ws;
    hipMalloc(&x, (size_t)M * K1 * 4); hipMalloc(&h, (size_t)M * K2 * 4);
    hipMalloc(&wi, (size_t)N * K1 * 4); hipMalloc(&wh, (size_t)N * K2 * 4);
    hipMalloc(&y, (size_t)M * N * 4);
    const size_t wsf = sf::linear_ws_floats(M, N, K1 + K2) + 1024;
    hipMalloc(&ws, wsf * 4);
    std::vector<float> buf((size_t)N * K1);
    unsigned s = 12345;
    auto fill = [&](float* d, size_t n, float sc) {
        for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; buf[i] = sc * ((int)(s >> 9) % 2001 - 1000) / 1000.f; }
        hipMemcpy(d, buf.data(), n * 4, hipMemcpyHostToDevice);
    };
    fill(x, (size_t)M * K1, 1.f); fill(h, (size_t)M * K2, 1.f); fill(wi, (size_t)N * K1, .02f); fill(wh, (size_t)N * K2, .02f);
    sf::Seg segs[2] = {{x, K1, wi, K1, K1}, {h, K2, wh, K2, K2}};
    sf::LinearOut out{}; out.y = y; out.ldy = N; out.epi = sf::EPI_NONE;
    hipStream_t st; hipStreamCreate(&st);
    float* slabs = nullptr; int ks = 0;
    const bool raw = argc > 3 && atoi(argv[3]);
    for (int i = 0; i < 20; ++i) sf::linear_nt(segs, 2, M, N, out, ws, wsf, st, raw ? &slabs : nullptr, &ks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    for (int i = 0; i < reps; ++i) sf::linear_nt(segs, 2, M, N, out, ws, wsf, st, raw ? &slabs : nullptr, &ks);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<float> yy(16);
    hipMemcpy(yy.data(), raw ? slabs : y, 64, hipMemcpyDeviceToHost);
    const double us = ms * 1000.0 / reps;
    printf("M=%d ksplit=%d raw=%d  %.2f us/launch  %.1f TFLOP/s  y0=%g %s\n", M, ks, (int)raw, us,
           2.0 * M * N * (K1 + K2) / us * 1e-6, yy[0], hipGetErrorString(hipGetLastError()));
#ifdef SF_LAB_STAMP
    {
        static long long st[256][32];
        hipMemcpyFromSymbol(st, HIP_SYMBOL(sf::g_lab_stamps), sizeof(st));
        long long t0 = st[0][0];
        for (int b = 0; b < 256; ++b) t0 = st[b][0] < t0 ? st[b][0] : t0;
        printf("wall_clock64 ticks (100 MHz => 10 ns) relative to the earliest block start\n");
        for (int b = 0; b < 256; b += 37) {
            printf("blk %3d: start %4lld  loop %4lld |", b, st[b][0] - t0, st[b][1] - t0);
            for (int i = 2; i < 12; i += 2) printf(" %4lld", st[b][i] - t0);
            printf(" | loopend %4lld end %4lld\n", st[b][28] - t0, st[b][29] - t0);
        }
        long long e = 0, smax = 0;
        for (int b = 0; b < 256; ++b) { e = st[b][29] - t0 > e ? st[b][29] - t0 : e; smax = st[b][0] - t0 > smax ? st[b][0] - t0 : smax; }
        printf("latest start %lld, latest end %lld ticks\n", smax, e);
    }
#endif
    return 0;
}
