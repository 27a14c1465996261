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

#include "ze_attn_decode.h"

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
    const int lane_off = lane * 8, last_off = K - 8;
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
