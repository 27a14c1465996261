// The decode weight-streaming kernel template (k_gemv) and its launch helper, shared by the bf16 instantiations
// (ze_gemv.hip) and the fp8 ones (ze_gemv8.hip).  Two translation units on purpose: instantiations of one template
// compiled together share register-allocation context, and adding the fp8 variants to the bf16 unit moved the
// dominant bf16 kernel by +2 % (16.1 vs 15.8 us) with an unchanged instruction mix.
#pragma once
#include <algorithm>
#include <type_traits>

#include "ze_kernels.h"

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// weights are read exactly once per token by exactly one wave: non-temporal 16-B loads
// (-DZE_PLAIN_WEIGHT_LOADS: measurement build with default-policy loads, tools/probe_mall.py)
__device__ __forceinline__ uint4 load_w16(const bf16_t* p) {
#ifdef ZE_PLAIN_WEIGHT_LOADS
    return *reinterpret_cast<const uint4*>(p);
#else
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
#endif
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
// WB: bits per weight.  16: bf16 rows.  8: OCP E4M3 rows with a per-row power-of-two scale (ze_quant.hip): a 16-B load
// carries 16 weights, a chunk is 1024 elements, the conversion is v_cvt_pk_f32_fp8 in registers, and the row's dot
// product is scaled once (exactly) in the epilogue.
template <int EPI, int PAIRS, int KSPLIT, int CH, int WB = 16>
__global__ void __launch_bounds__(256) k_gemv(const ze_gemv_args a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    bf16_t* xs = reinterpret_cast<bf16_t*>(smem);
    const int K = a.K;
    constexpr int CSH = (WB == 8) ? 10 : 9;       // log2(elements per chunk): 64 lanes x 16 B of weights
    constexpr int CE = 1 << CSH, EPL = CE / 64;   // elements per chunk / per lane
    const int nch = (K + CE - 1) >> CSH;          // the last chunk may be partial
    const int Kp = nch << CSH;
    float* red = reinterpret_cast<float*>(smem + (size_t)(Kp + CE) * 2);  // [4][2*PAIRS] partials, behind the zero chunk
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
    const int lane_off = lane * EPL;
    const int last_off = K - EPL;
    int ctx = 0, pos = 0;
    if (EPI == ZE_GV_QKV_ROPE) {
        ctx = a.st->ctx;
        pos = ctx + a.st->rope_delta;
    }

    // weight rows are addressed in BYTES (bf16: 2 per element, fp8: 1)
    constexpr int WBYTES = WB / 8;
    const uint8_t* Wb = (WB == 8) ? a.W8 : reinterpret_cast<const uint8_t*>(a.W);
    const size_t ldb = (WB == 8) ? (size_t)a.ldw8 : (size_t)a.ldw * 2;
    auto rows_of = [&](int p0, const uint8_t* (&wrow)[2 * PAIRS], int (&r1)[PAIRS], int (&r2)[PAIRS]) {
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
            wrow[2 * i] = Wb + (size_t)r1[i] * ldb;
            wrow[2 * i + 1] = Wb + (size_t)r2[i] * ldb;
        }
    };
    auto load_full = [&](const uint8_t* const (&wrow)[2 * PAIRS], int c0, uint4 (&w)[CH][2 * PAIRS]) {
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int off = min(((c0 + u * c_step) << CSH) + lane_off, last_off) * WBYTES;
#pragma unroll
            for (int i = 0; i < 2 * PAIRS; ++i) w[u][i] = load_w16(reinterpret_cast<const bf16_t*>(wrow[i] + off));
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
            } else if (EPI == ZE_GV_LOGITS) {  // folded arg-max: the seen flags travel with the first weight trip
                if (a.amax_ws && a.penalty != 1.0f) {
                    e.x1[i] = (float)a.seen[r1[i]];
                    e.x2[i] = (float)a.seen[r2[i]];
                }
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
    const uint8_t* wrow0[2 * PAIRS];
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
        if (tid + i * 256 < ((Kp + CE) >> 3)) stage_x(tid + i * 256, xq[i]);
    for (int v = tid + XV * 256; v < ((Kp + CE) >> 3); v += 256) {
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
        if constexpr (WB == 8) {
            if (a.act8) {  // FP8 activations: the row's E4M3 quantisation (per-row power-of-two scale), kept as bf16 in LDS
                typedef float f32x2_q __attribute__((ext_vector_type(2)));
                __syncthreads();
                float amax = 0.f;
                for (int v = tid; v < (K >> 3); v += 256) {
                    const uint4 q = *reinterpret_cast<const uint4*>(xs + v * 8);
                    const uint32_t u[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) amax = fmaxf(amax, fmaxf(fabsf(bf16lo(u[j])), fabsf(bf16hi(u[j]))));
                }
                amax = wave_max(amax);
                if (lane == 0) red[wid] = amax;
                __syncthreads();
                amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
                int kx = 0;
                if (amax > 0.f) {
                    int ex;
                    const float m = frexpf(amax / 448.0f, &ex);
                    kx = (m == 0.5f) ? ex - 1 : ex;
                }
                const float s8 = ldexpf(1.0f, kx), inv8 = ldexpf(1.0f, -kx);
                __syncthreads();
                for (int v = tid; v < (K >> 3); v += 256) {
                    const uint4 q = *reinterpret_cast<const uint4*>(xs + v * 8);
                    const uint32_t u[4] = {q.x, q.y, q.z, q.w};
                    uint32_t o[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int pk = __builtin_amdgcn_cvt_pk_fp8_f32(bf16lo(u[j]) * inv8, bf16hi(u[j]) * inv8, 0, false);
                        const f32x2_q back = __builtin_amdgcn_cvt_pk_f32_fp8(pk, false);
                        o[j] = pack_bf16x2(back.x * s8, back.y * s8);
                    }
                    *reinterpret_cast<uint4*>(xs + v * 8) = make_uint4(o[0], o[1], o[2], o[3]);
                }
            }
        }
    }
    __syncthreads();

    float best_v = -INFINITY;  // LOGITS + amax_ws: lane 0 of a wave keeps the best of the rows it finished
    int best_i = 0x7fffffff;
    // one pair set (2*PAIRS rows); `w0` / `ein` optionally hold its already-issued first trip and epilogue operands
    auto pair_set = [&](int p0, const uint8_t* const (&wrow)[2 * PAIRS], const int (&r1)[PAIRS], const int (&r2)[PAIRS],
                        auto have_first, uint4 (&w0)[CH][2 * PAIRS], epi_in<PAIRS>& ein) {
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        float acc[2 * PAIRS];
        f32x2_t acc2[2 * PAIRS];  // fp8 path: even / odd elements accumulate separately (v_pk_fma_f32)
#pragma unroll
        for (int i = 0; i < 2 * PAIRS; ++i) {
            acc[i] = 0.f;
            acc2[i] = f32x2_t{0.f, 0.f};
        }
        auto fma_chunk = [&](int c, const uint4 (&wc)[2 * PAIRS]) {
            c = min(c, nch);  // chunk nch is the zero chunk
            if constexpr (WB == 8) {
                // 16 weights per lane: word j of the load holds elements 4j .. 4j+3.  x comes as two 16-B LDS reads and
                // is widened to f32 pairs once for all rows of the set; a weight word costs two v_cvt_pk_f32_fp8 and
                // two packed FMAs (the stream is otherwise VALU-co-limited: 2.5 scalar ops per weight measured 20 %
                // over the bf16 stream instead of 2x).
                const uint4 xa = *reinterpret_cast<const uint4*>(xs + (c << CSH) + lane_off);
                const uint4 xb = *reinterpret_cast<const uint4*>(xs + (c << CSH) + lane_off + 8);
                const uint32_t xu[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
                f32x2_t xf[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) xf[j] = f32x2_t{bf16lo(xu[j]), bf16hi(xu[j])};
#pragma unroll
                for (int i = 0; i < 2 * PAIRS; ++i) {
                    const uint32_t wu[4] = {wc[i].x, wc[i].y, wc[i].z, wc[i].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x2_t lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)wu[j], false);
                        const f32x2_t hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)wu[j], true);
                        acc2[i] = __builtin_elementwise_fma(lo, xf[2 * j], acc2[i]);
                        acc2[i] = __builtin_elementwise_fma(hi, xf[2 * j + 1], acc2[i]);
                    }
                }
            } else {
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
        if (CH > 1 && c0 < nch) {
            uint4 w[CH > 1 ? CH - 1 : 1][2 * PAIRS];
#pragma unroll
            for (int u = 0; u < CH - 1; ++u) {
                const int c = c0 + u * c_step;
                if (c < nch) {
                    const int off = min((c << CSH) + lane_off, last_off) * WBYTES;
#pragma unroll
                    for (int i = 0; i < 2 * PAIRS; ++i) w[u][i] = load_w16(reinterpret_cast<const bf16_t*>(wrow[i] + off));
                }
            }
#pragma unroll
            for (int u = 0; u < CH - 1; ++u) {
                const int c = c0 + u * c_step;
                if (c < nch) fma_chunk(c, w[u]);
            }
        }
        if constexpr (WB == 8) {
#pragma unroll
            for (int i = 0; i < 2 * PAIRS; ++i) acc[i] = acc2[i].x + acc2[i].y;
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
            float d1 = acc[2 * i], d2 = acc[2 * i + 1];
            if constexpr (WB == 8) {  // the rows' power-of-two scales: exact
                d1 *= a.scale8[r1[i]];
                d2 *= a.scale8[r2[i]];
            }
            const float v1 = bf16_round(d1 + ein.b1[i]);
            const float v2 = bf16_round(d2 + ein.b2[i]);
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
                if (a.amax_ws) {  // k_argmax_partial's arithmetic: penalty on seen ids, larger value, then lower index
                    float p1 = v1, p2 = v2;
                    if (ein.x1[i] != 0.f) p1 = p1 < 0.f ? p1 * a.penalty : p1 / a.penalty;
                    if (ein.x2[i] != 0.f) p2 = p2 < 0.f ? p2 * a.penalty : p2 / a.penalty;
                    if (p1 > best_v || (p1 == best_v && r1[i] < best_i)) { best_v = p1; best_i = r1[i]; }
                    if (p2 > best_v || (p2 == best_v && r2[i] < best_i)) { best_v = p2; best_i = r2[i]; }
                }
            } else {
                a.out_bf16[r1[i]] = f32_to_bf16(v1);
                a.out_bf16[r2[i]] = f32_to_bf16(v2);
            }
        }
    };

    if (pre) pair_set(p_first, wrow0, r10, r20, std::true_type{}, wpre, e0);
    for (int p0 = p_first + nunits * PAIRS; p0 < P; p0 += nunits * PAIRS) {
        const uint8_t* wrow[2 * PAIRS];
        int r1[PAIRS], r2[PAIRS];
        epi_in<PAIRS> e;
        rows_of(p0, wrow, r1, r2);
        pair_set(p0, wrow, r1, r2, std::false_type{}, wpre, e);
    }
    if constexpr (EPI == ZE_GV_LOGITS && KSPLIT == 1) {
        if (a.amax_ws) {  // workgroup-uniform
            __syncthreads();  // `red` is free again
            if (lane == 0) {
                red[2 * wid] = best_v;
                red[2 * wid + 1] = __int_as_float(best_i);
            }
            __syncthreads();
            if (tid == 0) {
                for (int w = 1; w < 4; ++w) {
                    const float v = red[2 * w];
                    const int i = __float_as_int(red[2 * w + 1]);
                    if (v > best_v || (v == best_v && i < best_i)) { best_v = v; best_i = i; }
                }
                a.amax_ws[2 * blockIdx.x] = best_v;
                reinterpret_cast<int*>(a.amax_ws)[2 * blockIdx.x + 1] = best_i;
                // slots beyond this launch's grid (a smaller grid than an earlier launch's) go back to "nothing"
                for (int sl = blockIdx.x + gridDim.x; sl < 2048; sl += gridDim.x) {
                    a.amax_ws[2 * sl] = -INFINITY;
                    reinterpret_cast<int*>(a.amax_ws)[2 * sl + 1] = 0x7fffffff;
                }
            }
        }
    }
}

extern int ze_gemv_knobs[24];

template <int EPI, int PAIRS, int KSPLIT, int CH, int WB = 16>
static void launch_gemv_cfg(const ze_gemv_args& a, hipStream_t s) {
    const int P = a.N / 2;
    constexpr int CE = (WB == 8) ? 1024 : 512;
    const int nch = (a.K + CE - 1) / CE;
    const size_t lds = (size_t)(nch + 1) * CE * 2 + 4 * 2 * PAIRS * sizeof(float) + 64;
    int grid = (KSPLIT == 1) ? ze_cdiv(P, 4 * PAIRS) : ze_cdiv(P, PAIRS);
    if (grid > 2048) grid = 2048;
    // One resident round: with more blocks than the chip holds at once the tail of the grid waits for slots and
    // pays the x-staging prologue a second time (measured: gate_up 18.5 us -> 15.8 us at 3 blocks/CU, down
    // 13.2 -> 11.1 at 2 blocks/CU; one block more per CU is a cliff).  Long streams (lm_head) keep 2048 blocks.
    int dev = 0, cus = 256, occ = 0;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_gemv<EPI, PAIRS, KSPLIT, CH, WB>, 256, lds) == hipSuccess &&
        occ > 0) {
        const int resident = occ * cus;
        const int natural = (KSPLIT == 1) ? ze_cdiv(P, 4 * PAIRS) : ze_cdiv(P, PAIRS);
        if (natural <= 4 * resident && grid > resident) grid = resident;
        // K-split units (down projection): two workgroups per CU taking two pair sets each beat 1024 resident
        // one-set workgroups and 768 + 256 (10.4 vs 10.9 us)
        if (KSPLIT > 1 && natural > 2 * cus && natural <= 4 * cus) grid = ze_cdiv(natural, 2);
    }
    if (ze_gemv_knobs[2] > 0) grid = std::min(ze_cdiv(P, PAIRS * (KSPLIT == 1 ? 4 : 1)), ze_gemv_knobs[2]);
    if (EPI == ZE_GV_LOGITS && a.amax_ws && grid > 2048) grid = 2048;  // one arg-max slot per workgroup (k_argmax_final_folded)
    hipLaunchKernelGGL((k_gemv<EPI, PAIRS, KSPLIT, CH, WB>), dim3(grid), dim3(256), lds, s, a);
}

