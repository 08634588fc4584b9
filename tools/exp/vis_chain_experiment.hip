// EXPERIMENT (not built into libsf_hip.so): visual-attention partials of step t+1 beside the CHAIN
// h~ = tanh(W_out [wc ; h1]) -> t_a = W_h h~ + b_h of step t in ONE launch (producer tiles published
// write-through and counted in per m-tile; consumer blocks poll the counter, then fetch A with sc1 loads).
// Correct (tests passed), NOT faster: 776-788K agent-steps/s against 812K for the two paired launches.
//  * the attention body needs 163 VGPRs, so a 512-thread block of this kernel fits once per CU: the 536
//    blocks do not co-reside (chain first: the partials start 5-17 us late; partials first: the chain does);
//    forcing 128 VGPRs (launch_bounds(512,4)) spills 184 B/lane and the partials take 33 us;
//  * the hand-off (arrival atomics + polls of a device-scope counter) costs 6-8 us under the HBM load
//    of the partials (producers end at 5.3/8.8 us mean/max, consumers pass the wait at 14.8/15.9 us);
//    polling from every wave with s_sleep(1) made the launch 39 us, one poller per block 32 us.
// Timeline tool: tools/chain_trace.py (via sf_debug_trace).
// ---- chain body (from sf_gemm_small.h) ----
// ---- two dependent small products inside ONE launch ------------------------------------------------
// ROLE 1 (producer): the output tile is published WRITE-THROUGH (sc1 stores), drained, and counted in
// on the counter of its m-tile.  ROLE 2 (consumer): waits until every n-tile of its m-tile's rows
// has been counted in, then fetches its A fragments with sc1 loads (they bypass this CU's L1, which
// another CU's stores never refresh; the producer's write-through stores left no stale line in any
// L2) -- MI355X_MICROARCH.md, inter-workgroup visibility.  Waits are bounded: on expiry `error` is
// set and the block runs on (results invalid, no hang).  The last chain block out re-arms the
// counters.  Both roles: MT = 1 (one m-tile per block), a single K segment.
struct ChainSync {
    unsigned* cnt;         // [mtiles] tiles published per m-tile (zero at launch)
    unsigned* done;        // chain blocks finished (zero at launch)
    unsigned* error;
    unsigned target;       // producer n-tiles per m-tile
    unsigned nblocks;      // chain blocks in the grid (producers + consumers)
    unsigned ncnt;         // counters to re-arm
    unsigned long long* trace;   // development aid (sf_debug_trace): [grid blocks][8] stamps
};
__device__ __forceinline__ void chain_stamp(const ChainSync& cs, int slot) {
    if (cs.trace && threadIdx.x == 0) cs.trace[(size_t)blockIdx.x * 8 + slot] = wall_clock64();
}
constexpr unsigned CHAIN_SPIN_LIMIT = 1u << 20;

template <int CPW, int ROLE>
__device__ __forceinline__ void small_gemm_chain_body(const SmallArgs& a, int bx, int by,
                                                      const ChainSync& cs) {
    constexpr int MT = 1;
    __shared__ float s_part[SMALL_WAVES][256];
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = bx * 16, m0 = by * 16;
    const int li = lane & 15, kk = lane >> 4;
    const int n = min(n0 + li, a.N - 1);
    const int mrow = min(m0 + li, a.M - 1);
    const int ecol = min(n0 + (int)(threadIdx.x & 15), a.N - 1);
    float e_bias = 0.f, e_mul = 1.f;
    if (a.bias) e_bias = a.bias[ecol];
    if (a.bias2) e_bias += a.bias2[ecol];
    if (a.epi == EPI_MUL) e_mul = a.mul[ecol];

    chain_stamp(cs, 0);
    if (cs.trace && threadIdx.x == 0) cs.trace[(size_t)blockIdx.x * 8 + 7] = ROLE;
    const Seg& sg = a.sg.s0;
    const int total = a.sg.total;
    const int c_lo = (wave * total) / SMALL_WAVES, c_hi = ((wave + 1) * total) / SMALL_WAVES;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 fb[CPW], fa[CPW];
    // the weight fragments do not depend on the producer: in flight while the consumer waits
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
        const int c = min(c_lo + i, max(c_hi - 1, c_lo));
        const int k = c * 16 + 4 * kk;
        const bool ok = k < sg.K && c_lo + i < c_hi;
        const float4 bv = ld4(sg.W + (size_t)n * sg.ldw + (k < sg.K ? k : 0));
        fb[i] = ok ? bv : z;
    }
    if (ROLE == 2) {
        // ONE poller per block (every poll is a trip to the memory side; hundreds of pollers slow the
        // streaming kernels next to them), the rest of the block waits at the barrier
        if (threadIdx.x == 0) {
            unsigned spins = 0;
#pragma nounroll
            // (a returning atomic: performed at the memory side, never served from this XCD's L2)
            while (__hip_atomic_fetch_add(cs.cnt + by, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < cs.target) {
                __builtin_amdgcn_s_sleep(48);      // ~1 us: 16 blocks poll each word; a word serves ~88 atomics per us
                if (++spins > CHAIN_SPIN_LIMIT) {
                    __hip_atomic_store(cs.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        __syncthreads();
        chain_stamp(cs, 1);
    }
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sg.A), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
        const int c = min(c_lo + i, max(c_hi - 1, c_lo));
        const int k = c * 16 + 4 * kk;
        const bool ok = k < sg.K && c_lo + i < c_hi;
        const int kc = k < sg.K ? k : 0;
        float4 av;
        if (ROLE == 2) {
            const v4u v = __builtin_amdgcn_raw_buffer_load_b128(rs, (mrow * sg.lda + kc) * 4, 0, 16);
            av = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        } else {
            av = ld4(sg.A + (size_t)mrow * sg.lda + kc);
        }
        fa[i] = ok ? av : z;
    }
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < CPW; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc = mfma16(comp(fa[i], c), comp(fb[i], c), acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) s_part[wave][(kk * 4 + r) * 16 + li] = acc[r];
    __syncthreads();
    chain_stamp(cs, 2);

    if (threadIdx.x < 256) {
        const int rc = threadIdx.x;
        const int row = m0 + (rc >> 4), col = n0 + (rc & 15);
        if (row < a.M && col < a.N) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < SMALL_WAVES; ++w) v += s_part[w][rc];
            v += e_bias;
            if (a.epi == EPI_TANH) v = tanhf(v);
            if (a.epi == EPI_MUL) {
                if (a.y_pre) a.y_pre[(size_t)row * a.ldy_pre + col] = v;
                v *= e_mul;
            }
            float* o = a.y + (size_t)row * a.ldy + col;
            if (ROLE == 1) __hip_atomic_store(o, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sc1
            else *o = v;
        }
    }
    if (ROLE == 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // EVERY storing wave drains
        __syncthreads();
        if (threadIdx.x == 0)
            __hip_atomic_fetch_add(cs.cnt + by, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    chain_stamp(cs, 3);
    if (threadIdx.x == 0) {       // last chain block out re-arms the counters for the next launch
        const unsigned d = __hip_atomic_fetch_add(cs.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (d == cs.nblocks - 1) {
            for (unsigned i = 0; i < cs.ncnt; ++i)
                __hip_atomic_store(cs.cnt + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(cs.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// Host side: the launch plan of a small product (null plan => not a "small" shape).
// ---- kernel + host side (from sf_attention.hip) ----
// Visual-attention partials of step t+1 beside the CHAIN h~ = tanh(W_out [wc ; h1]) -> t_a = W_h h~ + b_h
// of step t in one launch: the partials are the long pole (HBM-bound, ~12 us inside the kernel) and
// feed nothing before the next gate product, while the two small products are the critical chain;
// as separate launches the second product had to wait for the partials of the first launch.
// Blocks [0, n1): producer product; [n1, n1 + n2): consumer product; the rest: partials.
template <int CPW1, int CPW2>
__global__ __launch_bounds__(SMALL_WAVES * 64, 4) void pair_vis_chain_kernel(VisArgs v, VisSplit sp, int nv,
                                                                         SmallArgs b1, int gx1, int n1,
                                                                         SmallArgs b2, int gx2,
                                                                         ChainSync cs) {
    // (the chain goes FIRST in dispatch order: it is the critical path and its consumers only wait
    // for blocks dispatched before them; the partials fill the remaining slots)
    const int bid = blockIdx.x;
    const int n2 = (int)cs.nblocks - n1;
    if (bid < n1) {
        small_gemm_chain_body<CPW1, 1>(b1, bid % gx1, bid / gx1, cs);
    } else if (bid < n1 + n2) {
        small_gemm_chain_body<CPW2, 2>(b2, (bid - n1) % gx2, (bid - n1) / gx2, cs);
    } else {
        if (threadIdx.x >= VSP_NW * 64) return;
        const int vb = bid - n1 - n2;
        visual_split_body<1>(v, sp, vb % VSP_G, vb / VSP_G);
    }
}
// Partials of the split visual attention + two chained small products (b2 consumes b1's output as
// its A operand) in one launch; sync: 16 dwords, zero before the first launch.  SF_ERR_UNSUPPORTED =
// shapes not covered (the caller launches the stages one after the other).
int pair_vis_chain(const PanoSrc& src, int B, const float* vec, int ldvec, float* alpha, float* out,
                   int ldo, const Dropout& drop, int drop_col0, float* split_part,
                   const SmallPlan& b1, const SmallPlan& b2, unsigned* sync, hipStream_t st) {
    const int F = src.IMG + src.LOC;
    if (!(b1.mt == 1 && b1.cpw == 8 && b2.mt == 1 && b2.cpw == 4)) return SF_ERR_UNSUPPORTED;
    if (!split_part || !sync || src.V <= (VSP_G - 1) * VSP_RPG || src.V > VSP_G * VSP_RPG || B > 256 ||
        F > VIS_CPL * 256 || (F & 3) || (!src.dense && ((src.IMG & 3) || (src.LOC & 3))) ||
        (ldvec & 3) || (ldo & 3))
        return SF_ERR_UNSUPPORTED;
    const SmallArgs &a1 = b1.args, &a2 = b2.args;
    // the chain: b2's A operand is b1's output, row for row; single K segments; no accumulation
    if (a2.sg.s0.A != a1.y || a2.sg.s0.lda != a1.ldy || a2.sg.s0.K != a1.N || a1.M != a2.M ||
        a1.sg.total != a1.sg.n0 || a2.sg.total != a2.sg.n0 || a1.accumulate || a2.accumulate ||
        b1.gy != b2.gy || b1.gy > 14 || (a1.sg.s0.lda & 3) || (a2.sg.s0.lda & 3))
        return SF_ERR_UNSUPPORTED;
    VisArgs va{src, vec, ldvec, alpha, out, ldo, drop, drop_col0};
    const int nv = VSP_G * B, n1 = b1.gx * b1.gy, n2 = b2.gx * b2.gy;
    ChainSync cs{sync, sync + 14, sync + 15, (unsigned)b1.gx, (unsigned)(n1 + n2), (unsigned)b1.gy, g_trace};
    hipLaunchKernelGGL((pair_vis_chain_kernel<8, 4>), dim3(nv + n1 + n2), dim3(SMALL_WAVES * 64), 0, st,
                       va, VisSplit{split_part, nullptr, g_trace}, nv, a1, b1.gx, n1, a2, b2.gx, cs);
    return launch_status();
}
