// EXPERIMENT (not built into libsf_hip.so): the gate-product GEMM with its stage tiles copied by the
// LDS-DMA path (global_load_lds_dwordx4), three unpadded stage buffers with the bank spread made on the
// SOURCE side (position p of row r holds k-chunk p ^ (r & 15)), two stages in flight across one raw
// s_barrier per stage with a counted vmcnt(6), and the fragment reads in inline asm (a compiler-visible
// ds_read behind a pending LDS-DMA makes hipcc wait vmcnt(0): it cannot tell the stage buffers apart).
// Correct (max abs err 2.4e-6 against fp64) but 1.0-1.4 us SLOWER than the register-staged kernel:
//   M=100: 26.80 vs 25.78 us, M=112: 26.65 vs 25.49, M=128: 29.17 vs 27.69 (tools/gemm_sweep.py).
// 72 -> 130 VGPRs; 48 KB per stage through the DMA path is the ~25-30 GB/s one CU's DMA sustains, and
// the 16 fragment reads per wave now cluster right behind the barrier.
// The same product with the stage tiles copied global -> LDS by the LDS-DMA path
// (global_load_lds_dwordx4): no staging registers, no ds_write pass, THREE stage buffers, two stages
// in flight across ONE raw barrier per stage (counted vmcnt: MI355X guide, "glds span barrier").
// The DMA writes lane-linear (wave base + lane * 16 B), so the LDS image is unpadded [rows][64] and
// the bank spread comes from the SOURCE side: position p of row r holds k-chunk p ^ (r & 15); a
// fragment read of chunk q for rows r = 16 t + li hits positions q ^ li: 16 distinct 16-B slots.
// A tile padded to 128 rows (clamped row index): 4 + 2 DMA instructions per wave and stage.
// ------------------------------------------------------------------------------------------------
constexpr int GL_AROWS = 128, GL_WROWS = 64, GL_BUF = (GL_AROWS + GL_WROWS) * TBK;   // floats per stage

template <int MT>
__global__ __launch_bounds__(512) void gemm_nt_glds_kernel(NtArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int tid = threadIdx.x, lane = tid & 63, wave8 = tid >> 6;
    const int wave = wave8 & 3, khalf = wave8 >> 2;
    const int li = lane & 15, kk = lane >> 4;
    int n0, split;
    {
        const int total = gridDim.x * gridDim.y;
        const int b = blockIdx.x + gridDim.x * blockIdx.y;
        int g = b;
        if ((total & 7) == 0) g = (b & 7) * (total >> 3) + (b >> 3);
        split = g / (int)gridDim.x;
        n0 = (g % (int)gridDim.x) * 64;
    }
    const int st0n = a.seg[0].K / TBK;
    const int st1n = a.nseg > 1 ? a.seg[1].K / TBK : 0;
    const int st2n = a.nseg > 2 ? a.seg[2].K / TBK : 0;
    const int stages = st0n + st1n + st2n;
    const int s_lo = (int)(((long)split * stages) / a.ksplit);
    const int s_hi = (int)(((long)(split + 1) * stages) / a.ksplit);
    const int nst = s_hi - s_lo;

    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // this lane's share of a stage copy: rows (4 j + lane/16) of the wave's 16 A rows / 8 W rows,
    // source chunk = position ^ (row & 15)
    const int lrow = lane >> 4, lpos = lane & 15;
    auto issue = [&](int s, int buf) {
        const float* A;
        const float* W;
        int lda, ldw, k0;
        if (s < st0n) {
            A = a.seg[0].A; W = a.seg[0].W; lda = a.seg[0].lda; ldw = a.seg[0].ldw; k0 = s * TBK;
        } else if (s < st0n + st1n) {
            A = a.seg[1].A; W = a.seg[1].W; lda = a.seg[1].lda; ldw = a.seg[1].ldw; k0 = (s - st0n) * TBK;
        } else {
            A = a.seg[2].A; W = a.seg[2].W; lda = a.seg[2].lda; ldw = a.seg[2].ldw;
            k0 = (s - st0n - st1n) * TBK;
        }
        float* As = smem + buf * GL_BUF;
        float* Ws = As + GL_AROWS * TBK;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = wave8 * 8 + j * 4 + lrow;                              // 0..63
            const float* src = W + (size_t)min(n0 + r, a.N - 1) * ldw + k0 + 4 * (lpos ^ (r & 15));
            float* dst = Ws + (wave8 * 8 + j * 4) * TBK;                         // wave-uniform base
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = wave8 * 16 + j * 4 + lrow;                             // 0..127
            const float* src = A + (size_t)min(r, a.M - 1) * lda + k0 + 4 * (lpos ^ (r & 15));
            float* dst = As + (wave8 * 16 + j * 4) * TBK;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
        }
    };
    // Fragment reads in inline asm: a compiler-visible LDS load behind a pending LDS-DMA makes hipcc
    // wait vmcnt(0) (it cannot tell the three stage buffers apart), which drains the copy pipeline
    // every stage.  The asm reads are invisible to that pass; their own counter (lgkmcnt) is waited
    // for by hand, in an asm that names the destination registers so nothing can move across it.
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
    auto lds_read = [](f32x4& d, unsigned addr) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(addr));
    };
    auto compute = [&](int buf) {
        const unsigned abase = lds0 + (unsigned)(buf * GL_BUF + li * TBK) * 4u;
        const unsigned wbase = lds0 + (unsigned)(buf * GL_BUF + GL_AROWS * TBK + (wave * 16 + li) * TBK) * 4u;
        f32x4 b0, b1, a0[MT], a1[MT];
        {
            const unsigned pos = 16u * (unsigned)((4 * (khalf * 2 + 0) + kk) ^ li);
            lds_read(b0, wbase + pos);
#pragma unroll
            for (int t = 0; t < MT; ++t) lds_read(a0[t], abase + pos + (unsigned)(t * 16 * TBK * 4));
        }
        {
            const unsigned pos = 16u * (unsigned)((4 * (khalf * 2 + 1) + kk) ^ li);
            lds_read(b1, wbase + pos);
#pragma unroll
            for (int t = 0; t < MT; ++t) lds_read(a1[t], abase + pos + (unsigned)(t * 16 * TBK * 4));
        }
        if constexpr (MT == 7)
            asm volatile("s_waitcnt lgkmcnt(8)"
                         : "+v"(b0), "+v"(a0[0]), "+v"(a0[1]), "+v"(a0[2]), "+v"(a0[3]), "+v"(a0[4]), "+v"(a0[5]), "+v"(a0[6]));
        else
            asm volatile("s_waitcnt lgkmcnt(9)"
                         : "+v"(b0), "+v"(a0[0]), "+v"(a0[1]), "+v"(a0[2]), "+v"(a0[3]), "+v"(a0[4]), "+v"(a0[5]), "+v"(a0[6]), "+v"(a0[MT - 1]));
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < MT; ++t) acc[t] = mfma16(a0[t][j], b0[j], acc[t]);
        // (the accumulators are named too: the wait must not be hoisted above the MFMAs of the first
        // half, which are what hides the latency of the second half's reads)
        if constexpr (MT == 7)
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(b1), "+v"(a1[0]), "+v"(a1[1]), "+v"(a1[2]), "+v"(a1[3]), "+v"(a1[4]), "+v"(a1[5]), "+v"(a1[6]),
                           "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]));
        else
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(b1), "+v"(a1[0]), "+v"(a1[1]), "+v"(a1[2]), "+v"(a1[3]), "+v"(a1[4]), "+v"(a1[5]), "+v"(a1[6]), "+v"(a1[MT - 1]),
                           "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[MT - 1]));
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < MT; ++t) acc[t] = mfma16(a1[t][j], b1[j], acc[t]);
    };

    if (nst > 0) {
        const int last = s_hi - 1;
        issue(s_lo, 0);
        issue(min(s_lo + 1, last), 1);
        for (int i = 0; i < nst; ++i) {
            // stage i has landed for this wave once at most the 6 copies of stage i+1 are pending;
            // behind the barrier it has landed for every wave, and buffer (i+2)%3 (stage i-1) is free
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            issue(min(s_lo + i + 2, last), (i + 2) % 3);
            compute(i % 3);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the clamped surplus copies
    }
    __builtin_amdgcn_s_barrier();

    // the two K halves meet in LDS (the stage buffers are free now)
    f32x4* red = reinterpret_cast<f32x4*>(smem);
    if (khalf == 1) {
#pragma unroll
        for (int t = 0; t < MT; ++t) red[(wave * MT + t) * 64 + lane] = acc[t];
    }
    __syncthreads();
    if (khalf == 1) return;
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] += red[(wave * MT + t) * 64 + lane];

    const int col = n0 + wave * 16 + li;
    if (col >= a.N) return;
    float* out = a.out + (a.ksplit > 1 ? (size_t)split * a.M * a.N : 0);
    float bsum = 0.f;
    if (a.bias) bsum += a.bias[col];
    if (a.bias2) bsum += a.bias2[col];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * t + kk * 4 + r;
            if (row < a.M) {
                float* o = out + (size_t)row * a.ldo + col;
                const float v = acc[t][r] + bsum;
                *o = (a.accumulate && a.ksplit == 1) ? *o + v : v;
            }
        }
}

