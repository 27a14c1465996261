// Fused decode "attention block" of one layer in ONE launch (one row pair per wave, every workgroup resident):
//     RMSNorm + QKV GEMV + M-RoPE + KV append  ->  flash-decoding slices  ->  slice merge  ->  O-proj + residual
// with the three all-to-all dependencies as in-launch grid barriers instead of kernel boundaries, and the O-proj
// weights of every wave requested at the start of the launch so that they arrive while the attention phases run.
//
// STATUS (round 1, measured on MI355X, 3B shape, ctx 800-1400): bit-identical to the four stand-alone kernels and
// break-even with them -- 590 +- 5 ms per question either way.  A barrier costs 2.5 us against 1.3 us for a kernel
// boundary inside a captured graph; what the launch wins back is the O-proj preload and three x-staging prologues.
// (An earlier reading of -30 ms came from a broken barrier: see mg_barrier.)  It is therefore OFF by default
// (ze_tune knob 3) and kept, with its parity tests, as the base for the step that can win: streaming the next
// gate/up weights into LDS while the attention phases leave HBM idle ("next").
//
// Barrier: two-level (the workgroups of a group label blockIdx % 8 -> last arriver of the group -> top counter ->
// the last group's last arriver publishes a generation word per group), relaxed agent-scope atomics and polls only.
// NO release / acquire fences: every byte that crosses workgroups inside the launch is stored write-through (sc1)
// and loaded with sc1 (L1-bypassing) loads, each storing wave drains (s_waitcnt vmcnt(0)) and the workgroup
// barriers before its lane 0 arrives.  Measured on MI355X (tools/probes/barrier_probe.hip): 2.46 us per barrier
// INCLUDING the payload hop, against 7.6 us with release/acquire fences and 12 us for a flat counter; 0 stale
// reads.  Correctness never depends on placement (blockIdx % 8 is only a label that happens to be the XCD under
// round-robin dispatch: it makes the polls L2-local).  Every spin is bounded: on a timeout the launch sets
// *timeout and exits; the host then reports an error and re-zeroes the counters.
//
// Arithmetic is bit-identical to the stand-alone kernels (same row-pair ownership, same per-lane accumulation
// order, same reductions); the test-suite compares the two paths token for token and logit for logit.
#include <algorithm>

#include "../ze_attn_decode.h"

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ unsigned mg_ld(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Self-resetting, so a launch needs neither a zeroing memset in front of it (a memset NODE in a captured decode step
// was observed to run unordered with the step's first kernel: stale counts, barriers that pass early) nor an epoch
// from the host: arrival counters return to 0 inside every barrier, the generation words only grow, and a launch
// counts its barriers from the generation it finds at its start (all launches of a stream are serialised, so that
// value is final).  returns false on timeout
__device__ __forceinline__ bool mg_barrier(ze_grid_barrier* b, unsigned target, unsigned groups, unsigned per_group) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's sc1 stores have left the CU
    __syncthreads();
    __shared__ unsigned s_ok;
    if (threadIdx.x == 0) {
        unsigned ok = 1;
        const unsigned g = blockIdx.x % groups;
        const unsigned old = __hip_atomic_fetch_add(&b->cnt[g * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == per_group - 1) {  // last of its group: nobody of the group adds again before gen[g] moves
            __hip_atomic_store(&b->cnt[g * 32], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned t = __hip_atomic_fetch_add(&b->top[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t == groups - 1) {  // last group: release everybody
                __hip_atomic_store(&b->top[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (unsigned k = 0; k < groups; ++k)
                    __hip_atomic_store(&b->gen[k * 32], target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        unsigned spins = 0;
        while ((int)(mg_ld(&b->gen[g * 32]) - target) < 0) {
            if (++spins > 2000000u) {
                ok = 0;
                break;
            }
        }
        if (!ok) __hip_atomic_store(&b->timeout[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_ok = ok;
    }
    __syncthreads();
    return s_ok != 0;
}

__device__ __forceinline__ uint4 mg_load_w16(const bf16_t* p) {
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}

// dot products of two weight rows with x (bf16 in LDS, K <= 4 * 512, zero padded): the accumulation order of
// k_gemv<.., PAIRS=1, KSPLIT=1, CH=4> -- chunk by chunk, lo then hi of each packed pair -- then the xor butterfly.
__device__ __forceinline__ void mg_fma_rows(const uint4 (&w)[4][2], const bf16_t* xs, int nch, int lane, float& a1,
                                            float& a2) {
    a1 = 0.f;
    a2 = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        if (u >= nch) break;
        const uint4 xq = *reinterpret_cast<const uint4*>(xs + (u << 9) + lane * 8);
        const uint32_t xu[4] = {xq.x, xq.y, xq.z, xq.w};
        const uint32_t w1[4] = {w[u][0].x, w[u][0].y, w[u][0].z, w[u][0].w};
        const uint32_t w2[4] = {w[u][1].x, w[u][1].y, w[u][1].z, w[u][1].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a1 = fmaf(bf16lo(w1[j]), bf16lo(xu[j]), a1);
            a1 = fmaf(bf16hi(w1[j]), bf16hi(xu[j]), a1);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a2 = fmaf(bf16lo(w2[j]), bf16lo(xu[j]), a2);
            a2 = fmaf(bf16hi(w2[j]), bf16hi(xu[j]), a2);
        }
    }
    a1 = wave_sum(a1);
    a2 = wave_sum(a2);
}

// NCH: 512-element chunks of the hidden size and of heads * 128 (equal: host-checked), compile-time so that every
// load of a trip is issued without a branch
template <int NCH>
__global__ void __launch_bounds__(256) k_layer_attn(const ze_layer_attn_args a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    bf16_t* xs = reinterpret_cast<bf16_t*>(smem);           // 4 KB: x of the current GEMV phase
    bf16_t* hs = reinterpret_cast<bf16_t*>(smem + 4096);    // 4 KB: the hidden stream before this block (residual)
    float* red = reinterpret_cast<float*>(smem + 8192);     // 16 floats
    ad_split_lds& AL = *reinterpret_cast<ad_split_lds*>(smem + 8192 + 64);
    float* sW = reinterpret_cast<float*>(smem + 8192 + 64); // the merge phase reuses the slice LDS
    if (mg_ld(&a.bar->timeout[0])) return;                  // an earlier launch gave up: do not spin again
    const unsigned gen0 = mg_ld(&a.bar->gen[(blockIdx.x % 8) * 32]);  // barriers of this launch: gen0 + 1, + 2, + 3
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = a.hidden;                                              // <= 2048 (host-checked)
    constexpr int nch = NCH, nchq = NCH;
    const int lane_off = lane * 8;
    const int ctx = a.st->ctx, pos = ctx + a.st->rope_delta;
    const int D = 128, halfD = 64;
    const int NQ = a.heads * D;  // O-proj reduction length (<= 2048)
    const int gw = blockIdx.x * 4 + wid, nwaves = gridDim.x * 4;
    const unsigned groups = 8, per_group = gridDim.x / 8;

    // ---------------- early issue: activation vector, norm weight, QKV rows of this wave, O-proj rows of this wave
    const bf16_t* xin = a.embed ? a.embed + (size_t)a.st->token * K : a.h;
    const bool v0_in = tid * 8 < K;
    uint4 xq0 = make_uint4(0, 0, 0, 0), g0 = make_uint4(0, 0, 0, 0);
    if (v0_in) {
        xq0 = *reinterpret_cast<const uint4*>(xin + tid * 8);
        g0 = *reinterpret_cast<const uint4*>(a.in_norm + tid * 8);
    }
    const int nqkv_pairs = (a.heads + 2 * a.kv_heads) * halfD;
    const int o_pairs = K >> 1;  // O-proj output rows / 2
    // The weight rows this wave owns are requested NOW: its QKV row pair (the grid is sized to one pair per wave)
    // and its O-proj pair, whose 8 KB then arrive while the attention phases run.  Loads of a trip are unconditional per lane; tail lanes re-read the last 16 B
    // of the row against the zero pad of x.
    auto qkv_rows = [&](int p, int& r1, int& r2) {
        r1 = (p / halfD) * D + (p % halfD);
        r2 = r1 + halfD;
    };
    auto load_rows = [&](const bf16_t* W, int ld, int r1, int r2, int kk, int n, uint4 (&w)[4][2]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int off = min((u << 9) + lane_off, kk - 8);
            if (u < n) {
                w[u][0] = mg_load_w16(W + (size_t)r1 * ld + off);
                w[u][1] = mg_load_w16(W + (size_t)r2 * ld + off);
            }
        }
    };
    // Unconditional (clamped) so that the waits below count exactly: a load behind a branch makes hipcc wait for
    // nearly everything outstanding at the first use of x.
    uint4 wq0[4][2], wo0[4][2];
    {
        int r1, r2;
        qkv_rows(min(gw, nqkv_pairs - 1), r1, r2);
        load_rows(a.wqkv, a.ldqkv, r1, r2, K, nch, wq0);
        const int po = min(gw, o_pairs - 1);
        load_rows(a.wo, a.ldo, 2 * po, 2 * po + 1, NQ, nchq, wo0);
    }

    // ---------------- phase A: x = RMSNorm(h) -> LDS
    {
        *reinterpret_cast<uint4*>(xs + tid * 8) = xq0;  // zero beyond K
        *reinterpret_cast<uint4*>(hs + tid * 8) = xq0;
        float ss = 0.f;
        const uint32_t u[4] = {xq0.x, xq0.y, xq0.z, xq0.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) ss += bf16lo(u[j]) * bf16lo(u[j]) + bf16hi(u[j]) * bf16hi(u[j]);
        ss = wave_sum(ss);
        if (lane == 0) red[wid] = ss;
        __syncthreads();
        const float inv = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)K + a.eps);
        __syncthreads();
        if (v0_in) {
            const uint32_t gwt[4] = {g0.x, g0.y, g0.z, g0.w};
            uint32_t o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                o[j] = pack_bf16x2(bf16_round(bf16lo(u[j]) * inv) * bf16lo(gwt[j]),
                                   bf16_round(bf16hi(u[j]) * inv) * bf16hi(gwt[j]));
            *reinterpret_cast<uint4*>(xs + tid * 8) = make_uint4(o[0], o[1], o[2], o[3]);
        }
        __syncthreads();
    }
    // QKV row pairs (i, i + 64) of a head, one pair per wave-iteration
    auto qkv_pair = [&](int p, const uint4 (&w)[4][2]) {
        int r1, r2;
        qkv_rows(p, r1, r2);
        const float b1 = a.bqkv ? bf16_to_f32(a.bqkv[r1]) : 0.f, b2 = a.bqkv ? bf16_to_f32(a.bqkv[r2]) : 0.f;
        const int j = r1 % D, hh = r1 / D;
        const float cs = bf16_to_f32(a.cosT[(size_t)pos * halfD + j]), sn = bf16_to_f32(a.sinT[(size_t)pos * halfD + j]);
        float a1, a2;
        mg_fma_rows(w, xs, nch, lane, a1, a2);
        if (lane == 0) {
            const float v1 = bf16_round(a1 + b1), v2 = bf16_round(a2 + b2);
            if (hh >= a.heads + a.kv_heads) {
                const uint32_t d = (uint32_t)((((hh - a.heads - a.kv_heads) * a.max_ctx + ctx) * D + j) * 2);
                ad_store2<true>(a.vcache, d, f32_to_bf16(v1));
                ad_store2<true>(a.vcache, d + halfD * 2, f32_to_bf16(v2));
            } else {
                const bf16_t o1 = f32_to_bf16(bf16_round(v1 * cs) + bf16_round(-v2 * sn));
                const bf16_t o2 = f32_to_bf16(bf16_round(v2 * cs) + bf16_round(v1 * sn));
                if (hh < a.heads) {
                    const uint32_t d = (uint32_t)((hh * D + j) * 2);
                    ad_store2<true>(a.q, d, o1);
                    ad_store2<true>(a.q, d + halfD * 2, o2);
                } else {
                    const uint32_t d = (uint32_t)((((hh - a.heads) * a.max_ctx + ctx) * D + j) * 2);
                    ad_store2<true>(a.kcache, d, o1);
                    ad_store2<true>(a.kcache, d + halfD * 2, o2);
                }
            }
        }
    };
    if (gw < nqkv_pairs) qkv_pair(gw, wq0);
    for (int p = gw + nwaves; p < nqkv_pairs; p += nwaves) {
        int r1, r2;
        qkv_rows(p, r1, r2);
        uint4 w[4][2];
        load_rows(a.wqkv, a.ldqkv, r1, r2, K, nch, w);
        qkv_pair(p, w);
    }
    if (!mg_barrier(a.bar, gen0 + 1, groups, per_group)) return;

    // ---------------- phase B: flash-decoding slices (workgroup -> (kv head, slice))
    {
        int chunk, nsplit;
        split_geometry(ctx + 1, a.max_splits, chunk, nsplit);
        const int nwork = nsplit * a.kv_heads;
        for (int wk = blockIdx.x; wk < nwork; wk += gridDim.x) {
            attn_split_body<true>(AL, a.q, a.kcache, a.vcache, ctx + 1, wk % a.kv_heads, wk / a.kv_heads, a.heads,
                                  a.kv_heads, a.max_ctx, a.scale_log2e, a.partial, a.max_splits);
            __syncthreads();
        }
        // (Measured, rejected: letting the workgroups without a slice touch the next gate/up weights -- one word per
        //  128-B line, 8-90 MB -- so that they sit in L2 / MALL when the MLP kernel starts: the question got 10-89 ms
        //  SLOWER, about 1 ms per prefetched MB; the later nt stream gains nothing from it.)
    }
    if (!mg_barrier(a.bar, gen0 + 2, groups, per_group)) return;

    // ---------------- phase C: merge the slices (workgroup -> two heads, 128 threads each)
    for (int hp = blockIdx.x; hp * 2 < a.heads; hp += gridDim.x) {
        const int h = hp * 2 + (tid >> 7), d = tid & 127;
        float* w = sW + (tid >> 7) * 80;  // the two halves of the block use disjoint LDS (heads is even: host-checked)
        attn_combine_body<true>(w, w + 64, a.partial, ctx + 1, h, d, a.heads, a.max_splits, a.attn);
        __syncthreads();
    }
    if (!mg_barrier(a.bar, gen0 + 3, groups, per_group)) return;

    // ---------------- phase D: O-proj + residual; x = attention output (fresh, sc1)
    {
        uint4 q = make_uint4(0, 0, 0, 0);
        if (tid * 8 < NQ) q = ad_load16<true>(a.attn, (uint32_t)(tid * 16));
        *reinterpret_cast<uint4*>(xs + tid * 8) = q;
        __syncthreads();
    }
    auto o_pair = [&](int p, const uint4 (&w)[4][2]) {
        const int r1 = 2 * p, r2 = 2 * p + 1;
        const float b1 = a.bo ? bf16_to_f32(a.bo[r1]) : 0.f, b2 = a.bo ? bf16_to_f32(a.bo[r2]) : 0.f;
        float a1, a2;
        mg_fma_rows(w, xs, nchq, lane, a1, a2);
        if (lane == 0) {
            const uint32_t hh = *reinterpret_cast<const uint32_t*>(hs + r1);  // rows 2p, 2p+1
            *reinterpret_cast<uint32_t*>(a.h + r1) =
                pack_bf16x2(bf16lo(hh) + bf16_round(a1 + b1), bf16hi(hh) + bf16_round(a2 + b2));
        }
    };
    if (gw < o_pairs) o_pair(gw, wo0);
    for (int p = gw + nwaves; p < o_pairs; p += nwaves) {
        uint4 w[4][2];
        load_rows(a.wo, a.ldo, 2 * p, 2 * p + 1, NQ, nchq, w);
        o_pair(p, w);
    }
}

static const size_t kLayerAttnLds = 8192 + 64 + sizeof(ad_split_lds) + 64;

template <int NCH>
static int layer_attn_resident(int cus) {
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_layer_attn<NCH>, 256, kLayerAttnLds) != hipSuccess) return 0;
    return occ * cus;
}

int ze_layer_attn_blocks(int hidden, int heads, int kv_heads, int head_dim) {
    if (head_dim != 128 || hidden > 2048 || hidden % 8 || (heads & 1) || kv_heads <= 0 || heads % kv_heads ||
        heads / kv_heads > AD_GMAX)
        return 0;
    const int nch = (hidden + 511) / 512;
    if ((heads * head_dim + 511) / 512 != nch) return 0;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    const int resident = nch == 1 ? layer_attn_resident<1>(cus) : nch == 2 ? layer_attn_resident<2>(cus)
                       : nch == 3 ? layer_attn_resident<3>(cus) : layer_attn_resident<4>(cus);
    // one row pair per wave in the larger of the two GEMV phases; every workgroup must be resident (grid barrier)
    const int pairs = std::max((heads + 2 * kv_heads) * 64, hidden / 2);
    int blocks = (ze_cdiv(pairs, 4) + 7) / 8 * 8;
    blocks = std::min(blocks, std::min(resident / 8 * 8, 8 * 64));
    return blocks >= 8 ? blocks : 0;
}

void ze_launch_layer_attn(const ze_layer_attn_args& a, int blocks, hipStream_t s) {
    const size_t lds = kLayerAttnLds;
    switch ((a.hidden + 511) / 512) {
        case 1: hipLaunchKernelGGL(k_layer_attn<1>, dim3(blocks), dim3(256), lds, s, a); break;
        case 2: hipLaunchKernelGGL(k_layer_attn<2>, dim3(blocks), dim3(256), lds, s, a); break;
        case 3: hipLaunchKernelGGL(k_layer_attn<3>, dim3(blocks), dim3(256), lds, s, a); break;
        default: hipLaunchKernelGGL(k_layer_attn<4>, dim3(blocks), dim3(256), lds, s, a); break;
    }
}

// =====================================================================================================================
// Second half of a decode layer in ONE launch:  O-proj + residual  ->  RMSNorm + gate/up + SiLU*up  ->  down + residual
//
// The two all-to-all dependencies here sit between phases that STREAM (8 / 90 / 45 MB), so every compute wave
// requests the first weight rows of the next phase BEFORE it arrives at the barrier (32 KB per wave for gate/up,
// 12 KB for down), waits only for its own stores (counted s_waitcnt vmcnt(N): loads, stores and atomics retire in
// issue order) and the barrier round trips run while HBM keeps streaming.  Workgroup = 4 compute waves + 1 SYNC
// wave (320 threads, one workgroup per CU): the sync wave owns the grid barrier -- it has no loads in flight, so
// its atomics and polls return at once, whereas a compute wave's own round trips would queue behind its 12-32
// outstanding HBM loads (in-order return); compute waves park at the workgroup barrier meanwhile.
//
// STATUS (round 1, MI355X, 3B shape): bit-identical to the three stand-alone GEMV launches and SLOWER: 36.5 us per
// layer against 4.7 + 15.8 + 10.4 = 30.9 us (651 vs 579 ms per question), so it is OFF (ze_tune knob 4).  In-kernel
// stamps of workgroup 0 (make EXTRA=-DZE_MLP_STAMPS), us: O-proj phase 5.5 | barrier 3.3 | h reload + norm 1.3 |
// gate/up stream 11 (13.8 stand-alone: the preload works) | store drain 1 | barrier 3.5 | act reload + staging 4 |
// down 6.1.  The streaming phases are FASTER than their launches; the hand-offs are not: write-through drain +
// barrier under a busy memory system + re-reading freshly written data costs 6-8 us per all-to-all hand-off,
// a kernel boundary plus ramp about 4.  Every persistent-layer variant tried this round ends at this arithmetic
// (see also k_layer_attn above); what remains open is point-to-point hand-off (data-tagged granules consumed as
// they land) instead of barriers.  Kept, with its parity test, as the measured reference for that work.
//
// Arithmetic per output element is that of k_gemv<RESIDUAL,1,1,4> (O-proj), <SWIGLU,1,1,4> (gate/up) and
// <RESIDUAL,1,4,6> (down: the 4 waves of a workgroup split K chunk-interleaved, partials added in wave order).
// Shapes: hidden, heads * 128 <= 2048 and equal in 512-chunks; 4096 < ipad <= 12288 (one trip of 6 chunks per wave).
template <int N>
__device__ __forceinline__ void mg_wait_stores() {  // all but the N youngest vector-memory operations are done
    static_assert(N == 0 || N == 12 || N == 33, "add the literal");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(33)" ::: "memory");
}

// grid barrier run by the sync wave (wave 4); the compute waves have drained their stores (mg_wait_stores) before
__device__ __forceinline__ bool mg_barrier_sync_wave(ze_grid_barrier* b, unsigned target, unsigned groups,
                                                     unsigned per_group, unsigned* s_ok) {
    __syncthreads();  // every compute wave's stores have left the CU
    if (threadIdx.x == 256) {
        unsigned ok = 1;
        const unsigned g = blockIdx.x % groups;
        const unsigned old = __hip_atomic_fetch_add(&b->cnt[g * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == per_group - 1) {
            __hip_atomic_store(&b->cnt[g * 32], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned t = __hip_atomic_fetch_add(&b->top[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t == groups - 1) {
                __hip_atomic_store(&b->top[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (unsigned k = 0; k < groups; ++k)
                    __hip_atomic_store(&b->gen[k * 32], target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        unsigned spins = 0;
        while ((int)(mg_ld(&b->gen[g * 32]) - target) < 0) {
            if (++spins > 2000000u) {
                ok = 0;
                break;
            }
        }
        if (!ok) __hip_atomic_store(&b->timeout[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *s_ok = ok;
    }
    __syncthreads();
    return *s_ok != 0;
}

template <int NCH>
__global__ void __launch_bounds__(320) k_layer_mlp(const ze_layer_mlp_args a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    bf16_t* xs = reinterpret_cast<bf16_t*>(smem);                    // x of the current phase (<= 24 chunks + zero chunk)
    bf16_t* hs = reinterpret_cast<bf16_t*>(smem + 25 * 1024);        // 4 KB: the hidden stream entering the phase
    float* red = reinterpret_cast<float*>(smem + 29 * 1024);         // 16 floats
    unsigned* s_ok = reinterpret_cast<unsigned*>(smem + 29 * 1024 + 64);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);        // 0..3 compute, 4 sync
    const bool compute = wid < 4;
    const int K = a.hidden, NQ = a.nq, IP = a.ipad;
    constexpr int nch = NCH;
    const int nchd = (IP + 511) >> 9;                                // chunks of the down reduction (<= 24)
    const int lane_off = lane * 8;
    const int gw = blockIdx.x * 4 + wid, nwaves = gridDim.x * 4;
    const unsigned groups = 8, per_group = gridDim.x / 8;
    if (mg_ld(&a.bar->timeout[0])) return;
    const unsigned gen0 = mg_ld(&a.bar->gen[(blockIdx.x % 8) * 32]);
#ifdef ZE_MLP_STAMPS
#define MLP_STAMP(i)                                                                               \
    do {                                                                                           \
        if (blockIdx.x == 0 && tid == 0) {                                                         \
            const unsigned long long t = __builtin_amdgcn_s_memrealtime();                         \
            reinterpret_cast<unsigned long long*>(a.bar->timeout + 8)[i] = t;                      \
        }                                                                                          \
    } while (0)
#else
#define MLP_STAMP(i)
#endif
    MLP_STAMP(0);

    auto load_rows = [&](const bf16_t* W, int ld, int r1, int r2, int kk, uint4 (&w)[4][2]) {
#pragma unroll
        for (int u = 0; u < NCH; ++u) {
            const int off = min((u << 9) + lane_off, kk - 8);
            w[u][0] = mg_load_w16(W + (size_t)r1 * ld + off);
            w[u][1] = mg_load_w16(W + (size_t)r2 * ld + off);
        }
    };
    const int o_pairs = K >> 1, gu_pairs = IP, d_pairs = K >> 1;
    auto load_gu = [&](int p, uint4 (&w)[4][2]) {  // p clamped: a trip past the end re-reads the last pair
        p = min(p, gu_pairs - 1);
        const int r1 = (p >> 4) * 32 + (p & 15);
        load_rows(a.wgu, a.ldgu, r1, r1 + 16, K, w);
    };
    auto load_down = [&](int pd, uint4 (&w)[6][2]) {  // wave w takes chunks w, w + 4, ..., w + 20
        pd = min(pd, d_pairs - 1);
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int off = min(((wid + 4 * u) << 9) + lane_off, IP - 8);
            w[u][0] = mg_load_w16(a.wdown + (size_t)(2 * pd) * a.lddown + off);
            w[u][1] = mg_load_w16(a.wdown + (size_t)(2 * pd + 1) * a.lddown + off);
        }
    };

    // ================================================================ phase O: h += Wo . attn
    uint4 wo0[4][2], xq0 = make_uint4(0, 0, 0, 0), hq0 = make_uint4(0, 0, 0, 0);
    if (compute) {
        const int po = min(gw, o_pairs - 1);
        if (tid * 8 < NQ) xq0 = *reinterpret_cast<const uint4*>(a.attn + tid * 8);
        if (tid * 8 < K) hq0 = *reinterpret_cast<const uint4*>(a.h + tid * 8);
        load_rows(a.wo, a.ldo, 2 * po, 2 * po + 1, NQ, wo0);
        *reinterpret_cast<uint4*>(xs + tid * 8) = xq0;  // zero beyond NQ
        *reinterpret_cast<uint4*>(hs + tid * 8) = hq0;
    }
    __syncthreads();
    uint4 wg[4][4][2], g0 = make_uint4(0, 0, 0, 0);  // gate/up ring: four row pairs in flight per wave
    if (compute) {
        auto o_pair = [&](int p, const uint4 (&w)[4][2]) {
            const int r1 = 2 * p;
            const float b1 = a.bo ? bf16_to_f32(a.bo[r1]) : 0.f, b2 = a.bo ? bf16_to_f32(a.bo[r1 + 1]) : 0.f;
            float a1, a2;
            mg_fma_rows(w, xs, nch, lane, a1, a2);
            if (lane == 0) {
                const uint32_t hh = *reinterpret_cast<const uint32_t*>(hs + r1);
                __builtin_amdgcn_raw_buffer_store_b32(
                    pack_bf16x2(bf16lo(hh) + bf16_round(a1 + b1), bf16hi(hh) + bf16_round(a2 + b2)), ad_rsrc(a.h),
                    (uint32_t)(r1 * 2), 0, 16);
            }
        };
        if (gw < o_pairs) o_pair(gw, wo0);
        for (int p = gw + nwaves; p < o_pairs; p += nwaves) {
            uint4 w[4][2];
            load_rows(a.wo, a.ldo, 2 * p, 2 * p + 1, NQ, w);
            o_pair(p, w);
        }
        asm volatile("" ::: "memory");  // the preload below must be issued AFTER the stores above
#pragma unroll
        for (int k = 0; k < 4; ++k) load_gu(gw + k * nwaves, wg[k]);
        if (tid * 8 < K) g0 = *reinterpret_cast<const uint4*>(a.post_norm + tid * 8);
        MLP_STAMP(1);
        mg_wait_stores<33>();
        MLP_STAMP(2);
    }
    if (!mg_barrier_sync_wave(a.bar, gen0 + 1, groups, per_group, s_ok)) return;
    MLP_STAMP(3);

    // ================================================================ phase G: act = SiLU(gate) * up of RMSNorm(h)
    if (compute) {
        uint4 q = make_uint4(0, 0, 0, 0);
        if (tid * 8 < K) q = ad_load16<true>(a.h, (uint32_t)(tid * 16));
        *reinterpret_cast<uint4*>(xs + tid * 8) = q;
        *reinterpret_cast<uint4*>(hs + tid * 8) = q;  // residual of the down projection
        float ss = 0.f;
        const uint32_t u[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) ss += bf16lo(u[j]) * bf16lo(u[j]) + bf16hi(u[j]) * bf16hi(u[j]);
        ss = wave_sum(ss);
        if (lane == 0) red[wid] = ss;
    }
    __syncthreads();
    if (compute) {
        const float inv = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)K + a.eps);
        if (tid * 8 < K) {
            const uint4 q = *reinterpret_cast<const uint4*>(xs + tid * 8);
            const uint32_t u[4] = {q.x, q.y, q.z, q.w}, gwt[4] = {g0.x, g0.y, g0.z, g0.w};
            uint32_t o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                o[j] = pack_bf16x2(bf16_round(bf16lo(u[j]) * inv) * bf16lo(gwt[j]),
                                   bf16_round(bf16hi(u[j]) * inv) * bf16hi(gwt[j]));
            *reinterpret_cast<uint4*>(xs + tid * 8) = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
    __syncthreads();
    MLP_STAMP(4);
    uint4 wd[3][6][2];  // down ring: three row pairs of this workgroup in flight
    if (compute) {
        auto gu_pair = [&](int p, const uint4 (&w)[4][2]) {
            float a1, a2;
            mg_fma_rows(w, xs, nch, lane, a1, a2);
            if (lane == 0) {
                const float v1 = bf16_round(a1 + 0.f), v2 = bf16_round(a2 + 0.f);
                ad_store2<true>(a.act, (uint32_t)(p * 2), f32_to_bf16(bf16_round(silu_f(v1)) * v2));
            }
        };
        bool more = true;
        for (int p = gw; more && p < gu_pairs; p += 4 * nwaves) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int pk = p + k * nwaves;
                if (pk >= gu_pairs) {  // wave-uniform
                    more = false;
                    break;
                }
                gu_pair(pk, wg[k]);
                load_gu(pk + 4 * nwaves, wg[k]);
            }
        }
        MLP_STAMP(5);
        asm volatile("" ::: "memory");
        load_down(blockIdx.x, wd[0]);
        mg_wait_stores<12>();
        MLP_STAMP(6);
    }
    if (!mg_barrier_sync_wave(a.bar, gen0 + 2, groups, per_group, s_ok)) return;
    MLP_STAMP(7);

    // ================================================================ phase D: h += Wdown . act
    const int units = (d_pairs - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;  // pairs of this workgroup
    if (compute) {
        // act first (it returns ahead of the weight rows requested behind it), zero padded to one chunk past the end
        const int nvec = (nchd + 1) * 64;
        uint4 xv[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int v = tid + i * 256;
            xv[i] = ad_load16<true>(a.act, (uint32_t)(min(v, (IP >> 3) - 1) * 16));
        }
        load_down(blockIdx.x + gridDim.x, wd[1]);
        load_down(blockIdx.x + 2 * gridDim.x, wd[2]);
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int v = tid + i * 256;
            if (v < nvec) *reinterpret_cast<uint4*>(xs + v * 8) = (v * 8 < IP) ? xv[i] : make_uint4(0, 0, 0, 0);
        }
    }
    __syncthreads();
    MLP_STAMP(8);
    auto down_unit = [&](int u, const uint4 (&w)[6][2]) {  // called by every wave (workgroup barriers inside)
        const int pd = blockIdx.x + u * gridDim.x;
        if (compute) {
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int c6 = 0; c6 < 6; ++c6) {
                const int c = min(wid + 4 * c6, nchd);  // chunk nchd is all zero
                const uint4 xq = *reinterpret_cast<const uint4*>(xs + (c << 9) + lane_off);
                const uint32_t xu[4] = {xq.x, xq.y, xq.z, xq.w};
                const uint32_t w1[4] = {w[c6][0].x, w[c6][0].y, w[c6][0].z, w[c6][0].w};
                const uint32_t w2[4] = {w[c6][1].x, w[c6][1].y, w[c6][1].z, w[c6][1].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    a1 = fmaf(bf16lo(w1[j]), bf16lo(xu[j]), a1);
                    a1 = fmaf(bf16hi(w1[j]), bf16hi(xu[j]), a1);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    a2 = fmaf(bf16lo(w2[j]), bf16lo(xu[j]), a2);
                    a2 = fmaf(bf16hi(w2[j]), bf16hi(xu[j]), a2);
                }
            }
            a1 = wave_sum(a1);
            a2 = wave_sum(a2);
            if (lane == 0) {
                red[wid * 2] = a1;
                red[wid * 2 + 1] = a2;
            }
        }
        __syncthreads();
        if (tid == 0) {
            const float s1 = red[0] + red[2] + red[4] + red[6], s2 = red[1] + red[3] + red[5] + red[7];
            const uint32_t hh = *reinterpret_cast<const uint32_t*>(hs + 2 * pd);
            *reinterpret_cast<uint32_t*>(a.h + 2 * pd) =
                pack_bf16x2(bf16lo(hh) + bf16_round(s1 + 0.f), bf16hi(hh) + bf16_round(s2 + 0.f));
        }
        __syncthreads();
    };
    bool more = true;
    for (int u0 = 0; more && u0 < units; u0 += 3) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int u = u0 + k;
            if (u >= units) {  // workgroup-uniform
                more = false;
                break;
            }
            down_unit(u, wd[k]);
            if (compute && u + 3 < units) load_down(blockIdx.x + (u + 3) * gridDim.x, wd[k]);
        }
    }
    MLP_STAMP(9);
}

static const size_t kLayerMlpLds = 29 * 1024 + 128;

template <int NCH>
static int layer_mlp_resident(int cus) {
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_layer_mlp<NCH>, 320, kLayerMlpLds) != hipSuccess) return 0;
    return occ * cus;
}

int ze_layer_mlp_blocks(int hidden, int nq, int ipad) {
    if (hidden > 2048 || hidden % 8 || nq > 2048 || nq % 8 || ipad <= 4096 || ipad > 12288 || ipad % 16) return 0;
    const int nch = (hidden + 511) / 512;
    if ((nq + 511) / 512 != nch || nch != 4) return 0;  // the counted waits are written for four chunks
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    // ONE workgroup per CU: the occupancy query promises two, but five-wave workgroups put their fifth wave on the
    // same SIMD and a second workgroup then does not fit there -- 512 workgroups were not co-resident (barrier
    // timeouts), 256 are.
    const int resident = layer_mlp_resident<4>(cus);
    int blocks = std::min(cus, resident) / 8 * 8;
    blocks = std::min(blocks, 8 * 64);
    return blocks >= 8 ? blocks : 0;
}

void ze_launch_layer_mlp(const ze_layer_mlp_args& a, int blocks, hipStream_t s) {
    hipLaunchKernelGGL(k_layer_mlp<4>, dim3(blocks), dim3(320), kLayerMlpLds, s, a);
}

extern "C" int ze_mega_available() { return 1; }
