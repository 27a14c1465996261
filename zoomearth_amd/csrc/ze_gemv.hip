// Decode-path weight-streaming kernels (SURVEY.md K16-K22 at one token per step): out[N] = W[N,K] . x[K].
// This is THE dominant kernel family of the batch-1 zoom chain: every decode step streams all 6.17 GB of bf16
// weights once, so these kernels are HBM-bound and everything else is fused around the stream:
//   prologue : token-embedding fetch (layer 0) and RMSNorm of the 4-KB activation vector (recomputed per block)
//   epilogue : +bias, bf16 rounding, M-RoPE + KV-cache append (QKV), SiLU(gate)*up (gate/up rows are interleaved
//              in blocks of 16 in the packed weight), residual add in place, fp32 logits.
// Weights go straight to VGPRs with non-temporal 16-B loads (no LDS round trip: each row is read by exactly one
// wave), 8-16 independent 1-KiB wave loads in flight per wave and NO per-lane branch around any load; x sits in
// LDS as bf16.  Issue order follows the in-order vmcnt retirement: activation vector + norm weight first, then the
// first trip of the weight stream (so the prologue only waits for two L2-resident vectors while the HBM latency
// of the first weight loads hides behind it), then the epilogue operands (bias / residual / cos-sin).
#include <algorithm>
#include <type_traits>

#include "ze_kernels.h"

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// weights are read exactly once per token by exactly one wave: non-temporal 16-B loads
__device__ __forceinline__ uint4 load_w16(const bf16_t* p) {
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}

template <int PAIRS>
struct epi_in {
    float b1[PAIRS], b2[PAIRS];  // bias of the two rows
    float x1[PAIRS], x2[PAIRS];  // RESIDUAL: old hidden values; QKV_ROPE: cos, sin
};

// EPI: epilogue; PAIRS: row pairs per wave-iteration; KSPLIT: waves of a block sharing one pair set along K;
// CH: 512-element chunks per load trip (CH * 2 * PAIRS loads of 16 B in flight per lane).
// (Measured alternative, rejected: keeping each lane's x slices in registers with a per-wave RMSNorm removes the
//  LDS staging and all barriers but makes every wave re-read x and the norm weight from L2 -- 8 KB per 8-16 KB of
//  weight rows -- and ran 10-30 % slower on every decode shape.)
// (Measured, rejected: one-shot forms of the down projection that request the whole 45-MB matrix at t = 0 -- one
//  pair set per workgroup, 12 / 24 / 48 loads in flight per lane, every workgroup resident -- ran 10.8-12.9 us
//  against 10.4: across this family a launch costs about 2.5 us + bytes / 7 TB/s inside the kernel plus 1.3 us at
//  the boundary whatever the issue structure, so the remaining lever is hiding the fixed part across launches.)
template <int EPI, int PAIRS, int KSPLIT, int CH>
__global__ void __launch_bounds__(256) k_gemv(const ze_gemv_args a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    bf16_t* xs = reinterpret_cast<bf16_t*>(smem);
    const int K = a.K;
    const int nch = (K + 511) >> 9;  // 512-element chunks (64 lanes x 8); the last one may be partial
    const int Kp = nch << 9;
    float* red = reinterpret_cast<float*>(smem + (size_t)(Kp + 512) * 2);  // [4][2*PAIRS] partials, behind the zero chunk
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform -> scalar branches below

    const int P = a.N >> 1;
    const int halfD = a.D >> 1;
    const int unit = (KSPLIT == 1) ? (blockIdx.x * 4 + wid) : blockIdx.x;
    const int nunits = (KSPLIT == 1) ? gridDim.x * 4 : gridDim.x;
    const int c_begin = (KSPLIT == 1) ? 0 : wid;
    constexpr int c_step = KSPLIT;
    // Lanes past the end of a partial last chunk re-read the last 16 B of the row (valid memory) and multiply it
    // by the zero padding of x in LDS: a divergent branch per load would make hipcc wait vmcnt(0) after each one
    // and serialise the stream.
    const int lane_off = lane * 8;
    const int last_off = K - 8;
    int ctx = 0, pos = 0;
    if (EPI == ZE_GV_QKV_ROPE) {
        ctx = a.st->ctx;
        pos = ctx + a.st->rope_delta;
    }

    auto rows_of = [&](int p0, const bf16_t* (&wrow)[2 * PAIRS], int (&r1)[PAIRS], int (&r2)[PAIRS]) {
#pragma unroll
        for (int i = 0; i < PAIRS; ++i) {
            const int p = min(p0 + i, P - 1);
            if (EPI == ZE_GV_QKV_ROPE) {
                r1[i] = (p / halfD) * a.D + (p % halfD);
                r2[i] = r1[i] + halfD;
            } else if (EPI == ZE_GV_SWIGLU) {
                r1[i] = (p >> 4) * 32 + (p & 15);
                r2[i] = r1[i] + 16;
            } else {
                r1[i] = 2 * p;
                r2[i] = 2 * p + 1;
            }
            wrow[2 * i] = a.W + (size_t)r1[i] * a.ldw;
            wrow[2 * i + 1] = a.W + (size_t)r2[i] * a.ldw;
        }
    };
    auto load_full = [&](const bf16_t* const (&wrow)[2 * PAIRS], int c0, uint4 (&w)[CH][2 * PAIRS]) {
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int off = min(((c0 + u * c_step) << 9) + lane_off, last_off);
#pragma unroll
            for (int i = 0; i < 2 * PAIRS; ++i) w[u][i] = load_w16(wrow[i] + off);
        }
    };
    auto load_epi = [&](const int (&r1)[PAIRS], const int (&r2)[PAIRS], epi_in<PAIRS>& e) {
#pragma unroll
        for (int i = 0; i < PAIRS; ++i) {
            e.b1[i] = a.bias ? bf16_to_f32(a.bias[r1[i]]) : 0.f;
            e.b2[i] = a.bias ? bf16_to_f32(a.bias[r2[i]]) : 0.f;
            e.x1[i] = e.x2[i] = 0.f;
            if (EPI == ZE_GV_RESIDUAL) {
                e.x1[i] = bf16_to_f32(a.out_bf16[r1[i]]);
                e.x2[i] = bf16_to_f32(a.out_bf16[r2[i]]);
            } else if (EPI == ZE_GV_QKV_ROPE) {
                const int j = r1[i] % a.D;
                e.x1[i] = bf16_to_f32(a.cosT[(size_t)pos * halfD + j]);
                e.x2[i] = bf16_to_f32(a.sinT[(size_t)pos * halfD + j]);
            }
        }
    };

    // ---------------- early issue: x, norm weight, first weight trip, first epilogue operands
    const bf16_t* xin = a.x;
    if (a.embed) xin = a.embed + (size_t)a.st->token * K;
    // every x vector of this thread is requested before the first weight trip: loads retire in issue order, so an
    // x load issued later (the staging loop used to fetch vectors 2..6 of the 22-KB down-projection input one per
    // iteration) returns only after the wave's 8-12 HBM weight loads AND then pays one L2 round trip per iteration
    // (measured: down-projection 11.1 -> 10.4 us).  Addresses are clamped, not branched on.
    constexpr int XV = (KSPLIT > 1) ? 6 : 1;
    const bool v0_in = tid * 8 < K;
    uint4 xq[XV];
    uint4 g0 = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < XV; ++i) {
        const int v = min(tid + i * 256, (K >> 3) - 1);
        xq[i] = *reinterpret_cast<const uint4*>(xin + v * 8);
    }
    if (a.norm_w && v0_in) g0 = *reinterpret_cast<const uint4*>(a.norm_w + tid * 8);
    const int p_first = unit * PAIRS;
    // (the first trip may run past the last chunk: those loads re-read the row's last 16 B and meet the all-zero
    //  chunk kept behind x in LDS, so every wave -- not only those with a full first trip -- streams from t = 0)
    const bool pre = p_first < P;  // wave-uniform
    uint4 wpre[CH][2 * PAIRS];
    const bf16_t* wrow0[2 * PAIRS];
    int r10[PAIRS], r20[PAIRS];
    epi_in<PAIRS> e0;
    if (pre) {
        rows_of(p_first, wrow0, r10, r20);
        load_full(wrow0, c_begin, wpre);
        load_epi(r10, r20, e0);
    }

    // ---------------- prologue: x -> LDS as bf16, zero padded to Kp (optionally embed fetch and/or RMSNorm)
    float ss = 0.f;
    auto stage_x = [&](int v, uint4 q) {
        if (v * 8 >= K) q = make_uint4(0, 0, 0, 0);
        if (a.embed && blockIdx.x == 0 && v * 8 < K) *reinterpret_cast<uint4*>(a.embed_out + v * 8) = q;
        *reinterpret_cast<uint4*>(xs + v * 8) = q;
        const uint32_t u[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) ss += bf16lo(u[j]) * bf16lo(u[j]) + bf16hi(u[j]) * bf16hi(u[j]);
    };
#pragma unroll
    for (int i = 0; i < XV; ++i)
        if (tid + i * 256 < ((Kp + 512) >> 3)) stage_x(tid + i * 256, xq[i]);
    for (int v = tid + XV * 256; v < ((Kp + 512) >> 3); v += 256) {
        uint4 q = make_uint4(0, 0, 0, 0);
        if (v * 8 < K) q = *reinterpret_cast<const uint4*>(xin + v * 8);
        stage_x(v, q);
    }
    if (a.norm_w) {
        ss = wave_sum(ss);
        if (lane == 0) red[wid] = ss;
        __syncthreads();
        const float inv = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)K + a.eps);
        __syncthreads();
        for (int v = tid; v < (K >> 3); v += 256) {
            const uint4 q = *reinterpret_cast<const uint4*>(xs + v * 8);
            const uint4 g = (v == tid) ? g0 : *reinterpret_cast<const uint4*>(a.norm_w + v * 8);
            const uint32_t u[4] = {q.x, q.y, q.z, q.w}, gw[4] = {g.x, g.y, g.z, g.w};
            uint32_t o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                o[j] = pack_bf16x2(bf16_round(bf16lo(u[j]) * inv) * bf16lo(gw[j]),
                                   bf16_round(bf16hi(u[j]) * inv) * bf16hi(gw[j]));
            *reinterpret_cast<uint4*>(xs + v * 8) = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
    __syncthreads();

    // one pair set (2*PAIRS rows); `w0` / `ein` optionally hold its already-issued first trip and epilogue operands
    auto pair_set = [&](int p0, const bf16_t* const (&wrow)[2 * PAIRS], const int (&r1)[PAIRS], const int (&r2)[PAIRS],
                        auto have_first, uint4 (&w0)[CH][2 * PAIRS], epi_in<PAIRS>& ein) {
        float acc[2 * PAIRS];
#pragma unroll
        for (int i = 0; i < 2 * PAIRS; ++i) acc[i] = 0.f;
        auto fma_chunk = [&](int c, const uint4 (&wc)[2 * PAIRS]) {
            c = min(c, nch);  // chunk nch is the zero chunk
            const uint4 xq = *reinterpret_cast<const uint4*>(xs + (c << 9) + lane_off);
            const uint32_t xu[4] = {xq.x, xq.y, xq.z, xq.w};
#pragma unroll
            for (int i = 0; i < 2 * PAIRS; ++i) {
                const uint32_t wu[4] = {wc[i].x, wc[i].y, wc[i].z, wc[i].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[i] = fmaf(bf16lo(wu[j]), bf16lo(xu[j]), acc[i]);
                    acc[i] = fmaf(bf16hi(wu[j]), bf16hi(xu[j]), acc[i]);
                }
            }
        };
        int c0 = c_begin;
        if (decltype(have_first)::value) {
#pragma unroll
            for (int u = 0; u < CH; ++u) fma_chunk(c0 + u * c_step, w0[u]);
            c0 += CH * c_step;
        } else {
            load_epi(r1, r2, ein);
        }
        // full trips: CH chunks x 2*PAIRS rows of independent 16-B loads per lane, no conditions at all
        for (; c0 + (CH - 1) * c_step < nch; c0 += CH * c_step) {
            uint4 w[CH][2 * PAIRS];
            load_full(wrow, c0, w);
#pragma unroll
            for (int u = 0; u < CH; ++u) fma_chunk(c0 + u * c_step, w[u]);
        }
        // tail: the remaining (< CH) chunks of this wave, guarded by wave-uniform (scalar) conditions only
        if (c0 < nch) {
            uint4 w[CH - 1][2 * PAIRS];
#pragma unroll
            for (int u = 0; u < CH - 1; ++u) {
                const int c = c0 + u * c_step;
                if (c < nch) {
                    const int off = min((c << 9) + lane_off, last_off);
#pragma unroll
                    for (int i = 0; i < 2 * PAIRS; ++i) w[u][i] = load_w16(wrow[i] + off);
                }
            }
#pragma unroll
            for (int u = 0; u < CH - 1; ++u) {
                const int c = c0 + u * c_step;
                if (c < nch) fma_chunk(c, w[u]);
            }
        }
#pragma unroll
        for (int i = 0; i < 2 * PAIRS; ++i) acc[i] = wave_sum(acc[i]);
        if (KSPLIT > 1) {
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < 2 * PAIRS; ++i) red[wid * 2 * PAIRS + i] = acc[i];
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 2 * PAIRS; ++i)
                acc[i] = red[i] + red[2 * PAIRS + i] + red[4 * PAIRS + i] + red[6 * PAIRS + i];
            __syncthreads();
            if (wid != 0) return;
        }
        if (lane != 0) return;

        // ---------------- epilogue (one lane per row pair)
#pragma unroll
        for (int i = 0; i < PAIRS; ++i) {
            if (p0 + i >= P) break;
            const float v1 = bf16_round(acc[2 * i] + ein.b1[i]);
            const float v2 = bf16_round(acc[2 * i + 1] + ein.b2[i]);
            if (EPI == ZE_GV_QKV_ROPE) {
                const int hh = r1[i] / a.D, j = r1[i] % a.D;
                if (hh >= a.heads + a.kv_heads) {
                    bf16_t* d = a.vcache + ((size_t)(hh - a.heads - a.kv_heads) * a.max_ctx + ctx) * a.D;
                    d[j] = f32_to_bf16(v1);
                    d[j + halfD] = f32_to_bf16(v2);
                } else {
                    const float c = ein.x1[i], s = ein.x2[i];
                    const bf16_t o1 = f32_to_bf16(bf16_round(v1 * c) + bf16_round(-v2 * s));
                    const bf16_t o2 = f32_to_bf16(bf16_round(v2 * c) + bf16_round(v1 * s));
                    bf16_t* d = hh < a.heads ? a.out_bf16 + (size_t)hh * a.D
                                             : a.kcache + ((size_t)(hh - a.heads) * a.max_ctx + ctx) * a.D;
                    d[j] = o1;
                    d[j + halfD] = o2;
                }
            } else if (EPI == ZE_GV_SWIGLU) {
                a.out_bf16[p0 + i] = f32_to_bf16(bf16_round(silu_f(v1)) * v2);
            } else if (EPI == ZE_GV_RESIDUAL) {
                a.out_bf16[r1[i]] = f32_to_bf16(ein.x1[i] + v1);
                a.out_bf16[r2[i]] = f32_to_bf16(ein.x2[i] + v2);
            } else if (EPI == ZE_GV_LOGITS) {
                a.out_f32[r1[i]] = v1;
                a.out_f32[r2[i]] = v2;
            } else {
                a.out_bf16[r1[i]] = f32_to_bf16(v1);
                a.out_bf16[r2[i]] = f32_to_bf16(v2);
            }
        }
    };

    if (pre) pair_set(p_first, wrow0, r10, r20, std::true_type{}, wpre, e0);
    for (int p0 = p_first + nunits * PAIRS; p0 < P; p0 += nunits * PAIRS) {
        const bf16_t* wrow[2 * PAIRS];
        int r1[PAIRS], r2[PAIRS];
        epi_in<PAIRS> e;
        rows_of(p0, wrow, r1, r2);
        pair_set(p0, wrow, r1, r2, std::false_type{}, wpre, e);
    }
}

extern int ze_gemv_knobs[8];

template <int EPI, int PAIRS, int KSPLIT, int CH>
static void launch_gemv_cfg(const ze_gemv_args& a, hipStream_t s) {
    const int P = a.N / 2;
    const int nch = (a.K + 511) / 512;
    const size_t lds = (size_t)(nch + 1) * 512 * 2 + 4 * 2 * PAIRS * sizeof(float) + 64;
    int grid = (KSPLIT == 1) ? ze_cdiv(P, 4 * PAIRS) : ze_cdiv(P, PAIRS);
    if (grid > 2048) grid = 2048;
    // One resident round: with more blocks than the chip holds at once the tail of the grid waits for slots and
    // pays the x-staging prologue a second time (measured: gate_up 18.5 us -> 15.8 us at 3 blocks/CU, down
    // 13.2 -> 11.1 at 2 blocks/CU; one block more per CU is a cliff).  Long streams (lm_head) keep 2048 blocks.
    int dev = 0, cus = 256, occ = 0;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_gemv<EPI, PAIRS, KSPLIT, CH>, 256, lds) == hipSuccess &&
        occ > 0) {
        const int resident = occ * cus;
        const int natural = (KSPLIT == 1) ? ze_cdiv(P, 4 * PAIRS) : ze_cdiv(P, PAIRS);
        if (natural <= 4 * resident && grid > resident) grid = resident;
        // K-split units (down projection): two workgroups per CU taking two pair sets each beat 1024 resident
        // one-set workgroups and 768 + 256 (10.4 vs 10.9 us)
        if (KSPLIT > 1 && natural > 2 * cus && natural <= 4 * cus) grid = ze_cdiv(natural, 2);
    }
    if (ze_gemv_knobs[2] > 0) grid = std::min(ze_cdiv(P, PAIRS * (KSPLIT == 1 ? 4 : 1)), ze_gemv_knobs[2]);
    hipLaunchKernelGGL((k_gemv<EPI, PAIRS, KSPLIT, CH>), dim3(grid), dim3(256), lds, s, a);
}

// overrides set through ze_tune(): [0] down variant, [1] gate_up variant, [2] grid cap, [3] fused attention block
int ze_gemv_knobs[8] = {0, 0, 0, 0, 0, 0, 0, 0};

bool ze_launch_gemv(int epi, const ze_gemv_args& a, hipStream_t s) {
    // shape policy: long-K / few-row matrices split K over the 4 waves of a block (each wave streams its K/4 share
    // in ONE trip of 6 chunks: 12 loads in flight per lane, no second latency-exposed phase); many-row matrices give
    // each wave two row pairs (16 loads in flight per lane).
    if ((size_t)((a.K + 511) / 512 + 1) * 1024 > 60000) return false;  // x must fit the LDS stage
    const bool long_k = a.K > 4096;
    const bool many_rows = a.N >= 8192;
    switch (epi) {
        case ZE_GV_QKV_ROPE: launch_gemv_cfg<ZE_GV_QKV_ROPE, 1, 1, 4>(a, s); break;
        case ZE_GV_SWIGLU:
            if (many_rows && ze_gemv_knobs[1] == 1) launch_gemv_cfg<ZE_GV_SWIGLU, 2, 1, 4>(a, s);
            else launch_gemv_cfg<ZE_GV_SWIGLU, 1, 1, 4>(a, s);
            break;
        case ZE_GV_RESIDUAL:
            if (long_k && ze_gemv_knobs[0] == 1) launch_gemv_cfg<ZE_GV_RESIDUAL, 2, 4, 6>(a, s);
            else if (long_k && ze_gemv_knobs[0] == 2) launch_gemv_cfg<ZE_GV_RESIDUAL, 1, 4, 4>(a, s);
            else if (long_k && ze_gemv_knobs[0] == 3) launch_gemv_cfg<ZE_GV_RESIDUAL, 1, 1, 4>(a, s);
            else if (long_k) launch_gemv_cfg<ZE_GV_RESIDUAL, 1, 4, 6>(a, s);
            else launch_gemv_cfg<ZE_GV_RESIDUAL, 1, 1, 4>(a, s);
            break;
        case ZE_GV_LOGITS:
            if (many_rows) launch_gemv_cfg<ZE_GV_LOGITS, 2, 1, 4>(a, s);
            else launch_gemv_cfg<ZE_GV_LOGITS, 1, 1, 4>(a, s);
            break;
        default:
            if (long_k) launch_gemv_cfg<ZE_GV_PLAIN, 1, 4, 6>(a, s);
            else launch_gemv_cfg<ZE_GV_PLAIN, 1, 1, 4>(a, s);
            break;
    }
    return true;
}

// Rejected experiment (measured, round 1): a side-branch kernel in the decode graph that pre-touches the next MLP
// weights so they sit in the 256-MiB Infinity Cache while the latency-bound QKV / attention / O-proj kernels leave
// HBM idle.  With 32-256 prefetch blocks per layer the whole question got 1.3-2.3x SLOWER (the prefetcher competes
// with the chain for CUs / memory queues and the join stalls the next layer), so it is not in the tree.
