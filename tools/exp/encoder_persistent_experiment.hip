// EXPERIMENT (not built into libsf_hip.so): the encoder recurrence as ONE persistent launch.
// Correct (bit-identical to the per-step kernel, tests passed) but SLOWER on MI355X:
//   per-step launches (lstm_step_fused_kernel)            11.1 us / step
//   persistent, arrival counters + sc1 h loads            12.9 us / step
//   persistent, {epoch, value} granules (this version)    13.5 us / step
// An in-kernel hand-off between workgroups on different XCDs costs 2-3 memory-side round trips of
// ~2 us each (write-through store -> visible -> sc1 load, plus re-polls); a kernel boundary costs
// about the same and needs no co-residency, lock or bounded spins.  Kept for the record.
// ------------------------------------------------------------------------------------------------
// Persistent encoder recurrence: the T time steps of nn.LSTM(300 -> 512) (model.py:61-95) in ONE
// launch.  A per-step launch re-reads its 128 KB slice of W_hh every step (~5 us at the ~25 GB/s a
// CU sustains) and pays a kernel boundary; here block (slice, m-block) keeps the slice in REGISTERS
// (32 VGPRs per lane), the cell state of its 16 x 16 patch in registers too, and only h_t travels:
// each block publishes its h patch as 8-byte {epoch = t+1, value} granules with write-through
// (sc1) stores into a double-buffered [2][B][H] array, and the 32 blocks of the same m-block read
// the whole row set back with sc1 loads, re-reading until every tag carries the epoch (the data IS
// the flag: no counters, no drains, no cache-wide fences; MI355X_MICROARCH.md visibility rules).
// All blocks must be co-resident (<= 240 blocks of 1024 threads, one per CU).  Two such kernels at
// once (two streams) could starve each other, so launches are serialised by a device-wide lock
// taken by a one-thread kernel AHEAD of the launch in the same stream (a waiting stream holds one
// wave, never a partial grid).  Every wait is bounded: on a timeout `error` is set and the kernel
// runs to completion without waiting (results invalid, no hang).
// ------------------------------------------------------------------------------------------------
constexpr int ENC_CPW = 8;                              // chunks per (gate, k-slice) wave: H <= 512
constexpr unsigned ENC_SPIN_LIMIT = 1u << 21;

__device__ unsigned g_persistent_lock = 0;

__global__ void persistent_lock_kernel(unsigned* error) {
    if (threadIdx.x != 0) return;
    unsigned spins = 0;
    while (atomicCAS(&g_persistent_lock, 0u, 1u) != 0u) {
        __builtin_amdgcn_s_sleep(32);
        if (++spins > ENC_SPIN_LIMIT) {
            if (error) *error = 1u;
            break;
        }
    }
}
__global__ void persistent_unlock_kernel() {
    if (threadIdx.x == 0) atomicExch(&g_persistent_lock, 0u);
}

__global__ __launch_bounds__(4 * LSTM_KS * 64) void encoder_persistent_kernel(EncPersArgs p) {
    __shared__ float s_g[4][LSTM_KS][256];
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    typedef unsigned long long u64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int gate = wave & 3, ksl = wave >> 2;
    const int slice = blockIdx.x, mb = blockIdx.y, m0 = mb * 16;
    const int li = lane & 15, kk = lane >> 4;
    const int H = p.H, B = p.B;
    const size_t BH = (size_t)B * H;

    // this wave's 16 rows x (H/4)-deep slice of W_hh, resident for all T steps
    const int total = H >> 4;
    const int c_lo = (ksl * total) / LSTM_KS, c_hi = ((ksl + 1) * total) / LSTM_KS;
    const float* wrow = p.w_hh + (size_t)(gate * H + slice * 16 + li) * H;
    float4 wf[ENC_CPW];
#pragma unroll
    for (int i = 0; i < ENC_CPW; ++i) {
        const int c = min(c_lo + i, c_hi - 1);
        const float4 v = ld4(wrow + 16 * c + 4 * kk);
        wf[i] = c_lo + i < c_hi ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // the (row, unit) this thread updates (threads 0..255), its biases and its recurrent state
    const int prow = tid >> 4, pcol = tid & 15;
    const int pb = m0 + prow, pj = slice * 16 + pcol;
    const bool ptail = tid < 256 && pb < B;
    const int qb = min(pb, B - 1);
    float bias[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bias[g] = p.b_ih[g * H + pj] + p.b_hh[g * H + pj];
    const int len_b = p.lengths[qb];
    float c_state = 0.f, h_state = 0.f;                 // model.py:67-79 init_state
    const uint32_t rk = dropout_row_key(p.ctx_drop.seed, p.ctx_drop.stream, (uint32_t)(p.ctx_drop.row0 + qb));
    // h travels as 8-byte {epoch, value} granules (the data IS the flag): no counters, no drains
    const auto rs_g = __builtin_amdgcn_make_buffer_rsrc(p.gran, 0, (int)(2 * BH * 8), 0x00020000);
    const int arow = min(m0 + li, B - 1);
    bool timed_out = false;

    for (int t = 0; t < p.T; ++t) {
        float xv[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) xv[g] = p.xg[((size_t)t * B + qb) * 4 * H + g * H + pj];
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        if (t > 0) {                                      // h_0 = 0: the first step has no product
            // granules of h_t (epoch t) in buffer t & 1: 4 consecutive k = 32 B = two 16-B loads
            const unsigned base = (unsigned)((((size_t)(t & 1) * B + arow) * H) * 8);
            v4u lo[ENC_CPW], hi[ENC_CPW];
            for (unsigned spins = 0;; ++spins) {
                bool ok = true;
#pragma unroll
                for (int i = 0; i < ENC_CPW; ++i) {
                    const int c = min(c_lo + i, c_hi - 1);
                    const unsigned off = base + (unsigned)(16 * c + 4 * kk) * 8;
                    lo[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_g, off, 0, 16);
                    hi[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_g, off + 16, 0, 16);
                }
#pragma unroll
                for (int i = 0; i < ENC_CPW; ++i)
                    ok = ok && lo[i].y == (unsigned)t && lo[i].w == (unsigned)t && hi[i].y == (unsigned)t &&
                         hi[i].w == (unsigned)t;
                if (__all(ok) || timed_out) break;
                if (spins > ENC_SPIN_LIMIT) {
                    timed_out = true;
                    *p.error = 1u;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
#pragma unroll
            for (int i = 0; i < ENC_CPW; ++i) {
                const float4 a4 = make_float4(__uint_as_float(lo[i].x), __uint_as_float(lo[i].z),
                                              __uint_as_float(hi[i].x), __uint_as_float(hi[i].z));
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = mfma16(comp(a4, j), comp(wf[i], j), acc);
            }
        }
        __syncthreads();                                  // s_g of the previous step fully consumed
#pragma unroll
        for (int r = 0; r < 4; ++r) s_g[gate][ksl][(kk * 4 + r) * 16 + li] = acc[r];
        __syncthreads();
        if (ptail) {
            float g4[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v = bias[g] + xv[g];
#pragma unroll
                for (int k = 0; k < LSTM_KS; ++k) v += s_g[g][k][tid];
                g4[g] = v;
            }
            const float ig = sigmoidf_(g4[0]), fg = sigmoidf_(g4[1]), gg = tanhf(g4[2]), og = sigmoidf_(g4[3]);
            float c1 = fg * c_state + ig * gg;
            float h1 = og * tanhf(c1);
            const bool live = t < len_b;                  // packed sequence (model.py:88-95)
            if (!live) { c1 = c_state; h1 = h_state; }
            // publish first (write-through granule), then the tapes at leisure
            __hip_atomic_store(p.gran + ((size_t)((t + 1) & 1) * B + pb) * H + pj,
                               ((u64)(unsigned)(t + 1) << 32) | (u64)__float_as_uint(h1),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (p.gates) {
                float* gp = p.gates + ((size_t)t * B + pb) * 4 * H + pj;
                gp[0] = ig; gp[H] = fg; gp[2 * H] = gg; gp[3 * H] = og;
            }
            float cv = live ? h1 : 0.f;
            if (live && p.ctx_drop.on())
                cv = dropout_keep(rk, (uint32_t)(t * H + pj), p.ctx_drop.thresh) ? cv * p.ctx_drop.scale : 0.f;
            p.ctx[(size_t)pb * p.ld_ctx + (size_t)t * H + pj] = cv;
            p.cs[(size_t)(t + 1) * BH + (size_t)pb * H + pj] = c1;
            p.hs[(size_t)(t + 1) * BH + (size_t)pb * H + pj] = h1;
            c_state = c1;
            h_state = h1;
        }
    }
}

