// Decode-path weight-streaming kernels (SURVEY.md K16-K22 at one token per step): out[N] = W[N,K] . x[K].
// This is THE dominant kernel family of the batch-1 zoom chain: every decode step streams all 6.17 GB of bf16
// weights once, so these kernels are HBM-bound and everything else is fused around the stream:
//   prologue : token-embedding fetch (layer 0) and RMSNorm of the 4-KB activation vector (recomputed per block)
//   epilogue : +bias, bf16 rounding, M-RoPE + KV-cache append (QKV), SiLU(gate)*up (gate/up rows are interleaved
//              in blocks of 16 in the packed weight), residual add in place, fp32 logits.
// Weights go straight to VGPRs with 16-B loads (no LDS round trip: each row is read by exactly one wave), >= 8
// independent 1-KiB wave-loads in flight per wave; x sits in LDS as bf16.
#include "ze_kernels.h"

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// weights are read exactly once per token by exactly one wave: non-temporal 16-B loads
__device__ __forceinline__ uint4 load_w16(const bf16_t* p) {
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}

template <int EPI, int PAIRS, int KSPLIT>
__global__ void __launch_bounds__(256) k_gemv(const ze_gemv_args a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    bf16_t* xs = reinterpret_cast<bf16_t*>(smem);
    const int K = a.K;
    const int nch = (K + 511) >> 9;  // 512-element chunks (64 lanes x 8)
    const int Kp = nch << 9;
    float* red = reinterpret_cast<float*>(smem + (size_t)Kp * 2);  // [4][2*PAIRS] partials + 1 scratch row
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;

    // ---------------- prologue: x -> LDS (optionally embed fetch and/or RMSNorm)
    const bf16_t* xin = a.x;
    if (a.embed) xin = a.embed + (size_t)a.st->token * K;
    float ss = 0.f;
    for (int v = tid; v < (Kp >> 3); v += 256) {
        uint4 q = make_uint4(0, 0, 0, 0);
        if (v * 8 < K) q = *reinterpret_cast<const uint4*>(xin + v * 8);
        if (a.embed && blockIdx.x == 0 && v * 8 < K) *reinterpret_cast<uint4*>(a.embed_out + v * 8) = q;
        *reinterpret_cast<uint4*>(xs + v * 8) = q;
        const uint32_t u[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) ss += bf16lo(u[j]) * bf16lo(u[j]) + bf16hi(u[j]) * bf16hi(u[j]);
    }
    if (a.norm_w) {
        ss = wave_sum(ss);
        if (lane == 0) red[wid] = ss;
        __syncthreads();
        const float inv = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)K + a.eps);
        __syncthreads();
        for (int v = tid; v < (K >> 3); v += 256) {
            const uint4 q = *reinterpret_cast<const uint4*>(xs + v * 8);
            const uint4 g = *reinterpret_cast<const uint4*>(a.norm_w + v * 8);
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

    const int P = a.N >> 1;
    const int halfD = a.D >> 1;
    const int unit = (KSPLIT == 1) ? (blockIdx.x * 4 + wid) : blockIdx.x;
    const int nunits = (KSPLIT == 1) ? gridDim.x * 4 : gridDim.x;

    for (int p0 = unit * PAIRS; p0 < P; p0 += nunits * PAIRS) {
        int r1[PAIRS], r2[PAIRS];
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
        }
        float acc[2 * PAIRS];
#pragma unroll
        for (int i = 0; i < 2 * PAIRS; ++i) acc[i] = 0.f;

        // chunk loop: wave `wid` of a KSPLIT group takes chunks wid, wid+KSPLIT, ...; 4 chunks per trip
        const int c_begin = (KSPLIT == 1) ? 0 : wid;
        const int c_step = KSPLIT;
        for (int c0 = c_begin; c0 < nch; c0 += 4 * c_step) {
            uint4 w[4][2 * PAIRS];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int off = ((c0 + u * c_step) << 9) + lane * 8;
                const bool in = (c0 + u * c_step) < nch && off < K;
#pragma unroll
                for (int i = 0; i < PAIRS; ++i) {
                    w[u][2 * i] = in ? load_w16(a.W + (size_t)r1[i] * a.ldw + off) : make_uint4(0, 0, 0, 0);
                    w[u][2 * i + 1] = in ? load_w16(a.W + (size_t)r2[i] * a.ldw + off) : make_uint4(0, 0, 0, 0);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = c0 + u * c_step;
                if (c >= nch) break;
                const uint4 xq = *reinterpret_cast<const uint4*>(xs + (c << 9) + lane * 8);
                const uint32_t xu[4] = {xq.x, xq.y, xq.z, xq.w};
#pragma unroll
                for (int i = 0; i < 2 * PAIRS; ++i) {
                    const uint32_t wu[4] = {w[u][i].x, w[u][i].y, w[u][i].z, w[u][i].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[i] = fmaf(bf16lo(wu[j]), bf16lo(xu[j]), acc[i]);
                        acc[i] = fmaf(bf16hi(wu[j]), bf16hi(xu[j]), acc[i]);
                    }
                }
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
            if (wid != 0) continue;
        }
        if (lane != 0) continue;

        // ---------------- epilogue (one lane per row pair)
#pragma unroll
        for (int i = 0; i < PAIRS; ++i) {
            if (p0 + i >= P) break;
            float v1 = acc[2 * i], v2 = acc[2 * i + 1];
            if (a.bias) {
                v1 += bf16_to_f32(a.bias[r1[i]]);
                v2 += bf16_to_f32(a.bias[r2[i]]);
            }
            v1 = bf16_round(v1);
            v2 = bf16_round(v2);
            if (EPI == ZE_GV_QKV_ROPE) {
                const int hh = r1[i] / a.D, j = r1[i] % a.D;
                const int ctx = a.st->ctx;
                if (hh >= a.heads + a.kv_heads) {
                    bf16_t* d = a.vcache + ((size_t)(hh - a.heads - a.kv_heads) * a.max_ctx + ctx) * a.D;
                    d[j] = f32_to_bf16(v1);
                    d[j + halfD] = f32_to_bf16(v2);
                } else {
                    const int pos = ctx + a.st->rope_delta;
                    const float c = bf16_to_f32(a.cosT[(size_t)pos * halfD + j]);
                    const float s = bf16_to_f32(a.sinT[(size_t)pos * halfD + j]);
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
                a.out_bf16[r1[i]] = f32_to_bf16(bf16_to_f32(a.out_bf16[r1[i]]) + v1);
                a.out_bf16[r2[i]] = f32_to_bf16(bf16_to_f32(a.out_bf16[r2[i]]) + v2);
            } else if (EPI == ZE_GV_LOGITS) {
                a.out_f32[r1[i]] = v1;
                a.out_f32[r2[i]] = v2;
            } else {
                a.out_bf16[r1[i]] = f32_to_bf16(v1);
                a.out_bf16[r2[i]] = f32_to_bf16(v2);
            }
        }
    }
}

template <int EPI, int PAIRS, int KSPLIT>
static void launch_gemv_cfg(const ze_gemv_args& a, hipStream_t s) {
    const int P = a.N / 2;
    const int nch = (a.K + 511) / 512;
    const size_t lds = (size_t)nch * 512 * 2 + 4 * 2 * PAIRS * sizeof(float) + 64;
    int grid;
    if (KSPLIT == 1)
        grid = ze_cdiv(P, 4 * PAIRS);
    else
        grid = ze_cdiv(P, PAIRS);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL((k_gemv<EPI, PAIRS, KSPLIT>), dim3(grid), dim3(256), lds, s, a);
}

void ze_launch_gemv(int epi, const ze_gemv_args& a, hipStream_t s) {
    // shape policy: long-K / few-row matrices split K over the 4 waves of a block so every CU keeps >= 32 KiB of
    // loads in flight; many-row matrices give each wave two row pairs.
    const bool long_k = a.K > 4096;
    const bool many_rows = a.N >= 8192;
    switch (epi) {
        case ZE_GV_QKV_ROPE: launch_gemv_cfg<ZE_GV_QKV_ROPE, 1, 1>(a, s); break;
        case ZE_GV_SWIGLU:
            if (many_rows) launch_gemv_cfg<ZE_GV_SWIGLU, 2, 1>(a, s);
            else launch_gemv_cfg<ZE_GV_SWIGLU, 1, 1>(a, s);
            break;
        case ZE_GV_RESIDUAL:
            if (long_k) launch_gemv_cfg<ZE_GV_RESIDUAL, 1, 4>(a, s);
            else launch_gemv_cfg<ZE_GV_RESIDUAL, 1, 1>(a, s);
            break;
        case ZE_GV_LOGITS:
            if (many_rows) launch_gemv_cfg<ZE_GV_LOGITS, 2, 1>(a, s);
            else launch_gemv_cfg<ZE_GV_LOGITS, 1, 1>(a, s);
            break;
        default:
            if (long_k) launch_gemv_cfg<ZE_GV_PLAIN, 1, 4>(a, s);
            else launch_gemv_cfg<ZE_GV_PLAIN, 1, 1>(a, s);
            break;
    }
}
