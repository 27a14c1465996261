// The lm_head of the PREFILL paths: fp32 logits of the last position of up to eight chains in ONE pass over the weight matrix.
//
// ze_prefill_batch used to launch the single-chain logits GEMV (k_gemv<LOGITS>) once per chain of the pass: 16-32 launches that
// each streamed all 622 MB of the tied lm_head for ONE activation row (93 us each: 2.4 % of a 16-chain pass; more at 32).  Here
// a workgroup stages the final-norm'd last rows of NX chains in LDS and every weight fragment it streams feeds NX dot products:
// one pass per eight chains, still memory-bound (eight FMAs per weight element are ~17 us of vector ALU against ~95 us of HBM).
//
// Arithmetic per (chain, vocabulary row) = the single-chain GEMV's, spelled out so that it cannot depend on the instantiation:
// the RMSNorm prologue (fp32 sum of squares per thread in staging order, wave sums, the four waves' sums added in wave order,
// HF's cast points: bf16(x * inv) * weight rounded to bf16), then per lane an fma chain over its 8 elements of every 512-element
// chunk in chunk order, a wave sum, one bf16 rounding, the fp32 copy (HF:generation/utils.py:2894).  ze_prefill (one chain) runs
// the NX = 1 instantiation of THIS kernel, so a chain's first token is the same bits whether it was prefilled alone or in a
// batch (tests/test_gpu_batch.py::test_prefill_batch_is_bit_identical_to_single_prefills).
// Replaces: lm_head on the kept position + .float() (HF:models/qwen2_5_vl/modeling_qwen2_5_vl.py:1386-1387).
#include "ze_kernels.h"

typedef __attribute__((ext_vector_type(4))) unsigned int lg_u32x4;

struct ze_logits_multi_args {
    const bf16_t* W;       // [N, ldw] lm_head (bf16)
    int ldw, N, K;
    const bf16_t* norm_w;  // final norm weight [K]
    float eps;
    const bf16_t* x[8];    // the chains' last hidden rows [K]
    float* out[8];         // fp32 logits [N] per chain
};

template <int NX>
__global__ void __launch_bounds__(256) k_logits_multi(const ze_logits_multi_args a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int K = a.K;
    const int nch = (K + 511) >> 9;  // 512-element chunks; chunk `nch` of every staged row is all zero
    const int Kp = nch << 9;
    const int rowe = Kp + 512;       // staged elements per chain
    bf16_t* xs = reinterpret_cast<bf16_t*>(smem);
    float* red = reinterpret_cast<float*>(smem + (size_t)NX * rowe * 2);  // [NX][4]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- first weight trip requested before the prologue (its HBM latency hides behind the staging)
    constexpr int CH = 4, ROWS = 4;
    const int P4 = (a.N + ROWS - 1) / ROWS;           // groups of four consecutive vocabulary rows
    const int nunits = gridDim.x * 4;
    const int lane_off = lane * 8, last_off = K - 8;
    auto load_trip = [&](int g, int c0, lg_u32x4 (&w)[CH][ROWS]) {
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int off = min(((c0 + u) << 9) + lane_off, last_off);
#pragma unroll
            for (int i = 0; i < ROWS; ++i) {
                const int r = min(g * ROWS + i, a.N - 1);
                w[u][i] = __builtin_nontemporal_load(reinterpret_cast<const lg_u32x4*>(a.W + (size_t)r * a.ldw + off));
            }
        }
    };
    const int g_first = blockIdx.x * 4 + wid;
    lg_u32x4 wpre[CH][ROWS];
    if (g_first < P4) load_trip(g_first, 0, wpre);

    // ---- prologue: every chain's row -> LDS, RMSNorm'd as HF rounds it
    for (int n = 0; n < NX; ++n) {
        float ss = 0.f;
        for (int v = tid; v < (rowe >> 3); v += 256) {
            uint4 q = make_uint4(0, 0, 0, 0);
            if (v * 8 < K) q = *reinterpret_cast<const uint4*>(a.x[n] + v * 8);
            *reinterpret_cast<uint4*>(xs + (size_t)n * rowe + v * 8) = q;
            const uint32_t u[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ss = __fmaf_rn(bf16lo(u[j]), bf16lo(u[j]), ss);
                ss = __fmaf_rn(bf16hi(u[j]), bf16hi(u[j]), ss);
            }
        }
        ss = wave_sum(ss);
        if (lane == 0) red[n * 4 + wid] = ss;
    }
    __syncthreads();
    for (int n = 0; n < NX; ++n) {
        const float tot = __fadd_rn(__fadd_rn(__fadd_rn(red[n * 4 + 0], red[n * 4 + 1]), red[n * 4 + 2]), red[n * 4 + 3]);
        const float inv = rsqrtf(__fadd_rn(tot / (float)K, a.eps));
        for (int v = tid; v < (K >> 3); v += 256) {
            const uint4 q = *reinterpret_cast<const uint4*>(xs + (size_t)n * rowe + v * 8);
            const uint4 g = *reinterpret_cast<const uint4*>(a.norm_w + v * 8);
            const uint32_t u[4] = {q.x, q.y, q.z, q.w}, gw[4] = {g.x, g.y, g.z, g.w};
            uint32_t o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                o[j] = pack_bf16x2(__fmul_rn(bf16_round(__fmul_rn(bf16lo(u[j]), inv)), bf16lo(gw[j])),
                                   __fmul_rn(bf16_round(__fmul_rn(bf16hi(u[j]), inv)), bf16hi(gw[j])));
            *reinterpret_cast<uint4*>(xs + (size_t)n * rowe + v * 8) = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
    __syncthreads();

    // ---- the stream: four vocabulary rows per wave-iteration, CH chunks per trip, NX dot products per row
    bool first = true;
    for (int g = g_first; g < P4; g += nunits) {
        float acc[NX][ROWS];
#pragma unroll
        for (int n = 0; n < NX; ++n)
#pragma unroll
            for (int i = 0; i < ROWS; ++i) acc[n][i] = 0.f;
        for (int c0 = 0; c0 < nch; c0 += CH) {
            lg_u32x4 w[CH][ROWS];
            if (first && c0 == 0) {
#pragma unroll
                for (int u = 0; u < CH; ++u)
#pragma unroll
                    for (int i = 0; i < ROWS; ++i) w[u][i] = wpre[u][i];
            } else {
                load_trip(g, c0, w);
            }
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int c = min(c0 + u, nch);  // (chunks past the last one meet the zero chunk)
#pragma unroll
                for (int n = 0; n < NX; ++n) {
                    const uint4 xq = *reinterpret_cast<const uint4*>(xs + (size_t)n * rowe + (c << 9) + lane_off);
                    const uint32_t xu[4] = {xq.x, xq.y, xq.z, xq.w};
#pragma unroll
                    for (int i = 0; i < ROWS; ++i) {
                        const uint32_t wu[4] = {w[u][i].x, w[u][i].y, w[u][i].z, w[u][i].w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            acc[n][i] = __fmaf_rn(bf16lo(wu[j]), bf16lo(xu[j]), acc[n][i]);
                            acc[n][i] = __fmaf_rn(bf16hi(wu[j]), bf16hi(xu[j]), acc[n][i]);
                        }
                    }
                }
            }
        }
        first = false;
#pragma unroll
        for (int n = 0; n < NX; ++n)
#pragma unroll
            for (int i = 0; i < ROWS; ++i) acc[n][i] = wave_sum(acc[n][i]);
        if (lane == 0) {
#pragma unroll
            for (int n = 0; n < NX; ++n)
#pragma unroll
                for (int i = 0; i < ROWS; ++i) {
                    const int r = g * ROWS + i;
                    if (r < a.N) a.out[n][r] = bf16_round(acc[n][i]);
                }
        }
    }
}

template <int NX>
static void launch_logits_multi(const ze_logits_multi_args& a, hipStream_t s) {
    const int nch = (a.K + 511) / 512;
    const size_t lds = (size_t)NX * (nch + 1) * 1024 + NX * 4 * sizeof(float) + 64;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_logits_multi<NX>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        attr_set = true;
    }
    const int groups = (a.N + 3) / 4;
    int grid = std::min(2048, (groups + 3) / 4);
    hipLaunchKernelGGL((k_logits_multi<NX>), dim3(grid), dim3(256), lds, s, a);
}

// logits of the last rows of n chains: x_rows[i] bf16 [K] -> out_rows[i] fp32 [N]; false when the shape does not qualify
// (K % 8, a row that would not fit the LDS stage): the caller falls back to the per-chain GEMV
bool ze_launch_logits_rows(const bf16_t* W, int ldw, int N, int K, const bf16_t* norm_w, float eps, const bf16_t* const* x_rows,
                           float* const* out_rows, int n, hipStream_t s) {
    if (K % 8 != 0 || K < 8 || !norm_w || (size_t)8 * ((K + 511) / 512 + 1) * 1024 > 60 * 1024) return false;
    for (int i0 = 0; i0 < n; i0 += 8) {
        const int m = std::min(8, n - i0);
        ze_logits_multi_args a;
        a.W = W;
        a.ldw = ldw;
        a.N = N;
        a.K = K;
        a.norm_w = norm_w;
        a.eps = eps;
        for (int i = 0; i < 8; ++i) {
            a.x[i] = x_rows[i0 + std::min(i, m - 1)];
            a.out[i] = out_rows[i0 + std::min(i, m - 1)];
        }
        switch (m) {
            case 1: launch_logits_multi<1>(a, s); break;
            case 2: launch_logits_multi<2>(a, s); break;
            case 3: case 4: launch_logits_multi<4>(a, s); break;   // (spare slots repeat the last chain: same values written twice)
            default: launch_logits_multi<8>(a, s); break;
        }
    }
    return true;
}
