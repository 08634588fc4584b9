// Probe: operand / result layout of v_mfma_f64_16x16x4_f64 on gfx950.  A one-hot in lane p, B = lane id + 1:
// the non-zero results say which row lane p's A value belongs to and which B lanes share its k.
//   hipcc --offload-arch=gfx950 -O2 -o tools/exp/lab_mfma_f64_layout tools/exp/mfma_f64_layout.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
__global__ void probe(double* out) {
    const int lane = threadIdx.x;
    for (int p = 0; p < 64; ++p) {
        f64x4 c = {0, 0, 0, 0};
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(lane == p ? 1.0 : 0.0, (double)(lane + 1), c, 0, 0, 0);
        for (int v = 0; v < 4; ++v) out[(p * 64 + lane) * 4 + v] = c[v];
    }
}
int main() {
    double* d;
    (void)hipMalloc(&d, 64 * 64 * 4 * sizeof(double));
    probe<<<1, 64>>>(d);
    static double h[64 * 64 * 4];
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int p = 0; p < 64; p += 1) {
        if (!(p < 3 || p == 15 || p == 16 || p == 17 || p == 32 || p == 63)) continue;
        printf("A one-hot in lane %d: non-zero results (lane.v = B lane):", p);
        for (int l = 0; l < 64; ++l)
            for (int v = 0; v < 4; ++v)
                if (h[(p * 64 + l) * 4 + v] != 0.0) printf(" %d.%d=%d", l, v, (int)h[(p * 64 + l) * 4 + v] - 1);
        printf("\n");
    }
    return 0;
}
