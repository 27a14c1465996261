// One-shot skinny GEMM for the short projections of the batched decode step (qkv: N = 2560, o: N = 2048; K = 2048).
//
// These launches move 8-10 MB of weights -- 1.5 us of HBM time -- and were 8-9 us each: k_gemm_skinny walks its K
// quarter in four dependent load -> MFMA trips and fetches the residual behind the reduction.  Here a workgroup is
// SIXTEEN waves (1024 threads, one workgroup per CU) over the 16 weight rows of one fragment block, the K range cut
// into 16 shares: for K = 2048 a wave owns four 32-deep slices and requests everything it will ever need -- 4 weight
// fragments (contiguous 1-KiB reads of the fragment-major copy) and 4 x MT activation fragments -- in ONE batch, so the
// launch is a single exposed memory round trip; epilogue operands (bias, residual, chain state) are requested before it.
// The sixteen partial tiles meet in LDS and are added in a fixed two-level order ((w0+w1+w2+w3) + (w4+..+w7) + ...),
// a function of K alone: a chain's result does not depend on the batch.
//
// EPI_QKV: the M-RoPE + KV-append epilogue of the batched step (replaces k_rope_kv_batch: one launch and one pass over
// the qkv buffer less per layer).  The fragment copy of the qkv weight is packed with its rows permuted so that a
// block holds dims (8j .. 8j+7) and (64 + 8j .. 64 + 8j+7) of one head: the rotate_half partner of column fr is column
// fr ^ 8 of the same MFMA tile, one lane shuffle away.  Arithmetic = k_rope_kv_batch on the bf16-rounded projections
// (HF:models/qwen2_5_vl/modeling_qwen2_5_vl.py:557-599): q heads go to the q buffer, k / v rows straight into the KV
// cache at the chain's position.
#include <string.h>

#include "ze_kernels.h"

typedef __attribute__((ext_vector_type(8))) __bf16 os_bf16x8;
typedef __attribute__((ext_vector_type(4))) float os_f32x4;

enum { OS_EPI_BIAS = 0, OS_EPI_RESIDUAL = 1, OS_EPI_QKV = 2 };

struct ze_oneshot_args {
    const bf16_t* Xf;      // activations, fragment-major (k_rmsnorm(frag) / attention merge)
    const bf16_t* Wf;      // weights, fragment-major (row-permuted for OS_EPI_QKV)
    const bf16_t* bias;    // [N] in ORIGINAL row order, or null
    const float* wscale;   // W8: per-row power-of-two scales of the FP8 fragment copy, ORIGINAL row order
    const float* ascale;   // A8: Xf holds FP8 fragments (k_rmsnorm_row act8 = 2), one power-of-two scale per activation row
    const bf16_t* R;       // residual rows (OS_EPI_RESIDUAL)
    bf16_t* C;             // output rows (BIAS / RESIDUAL: [M, ldc]; QKV: the q buffer, row stride ldc)
    int ldr, ldc, M, N, K;
    // OS_EPI_QKV
    const ze_seq_dev* st;
    const int* seq_ids;
    const bf16_t *cosT, *sinT;
    bf16_t *kcache, *vcache;
    size_t cache_seq_stride;
    int heads, kv_heads, max_ctx;
};

typedef __attribute__((ext_vector_type(2))) __bf16 os_bf16x2;
typedef __attribute__((ext_vector_type(2))) unsigned int os_u32x2;
__device__ __forceinline__ os_bf16x8 os_deq_fp8x8(os_u32x2 w, float scale) {  // see deq_fp8x8 (ze_gemm.hip)
    union {
        os_bf16x2 h[4];
        os_bf16x8 v;
    } u;
    u.h[0] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w.x, scale, false);
    u.h[1] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w.x, scale, true);
    u.h[2] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w.y, scale, false);
    u.h[3] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w.y, scale, true);
    return u.v;
}

// W8: Wf is the FP8 fragment copy (8 B per lane per fragment), dequantised in registers with the row's scale.
// A8 (with W8): the activations are FP8 fragments too and the product runs on v_mfma_f32_16x16x32_fp8_fp8 -- both
// operands straight from memory into the matrix cores, half the bytes of each; the row and column scales (powers of two)
// multiply the fp32 sums in the epilogue, exactly.
template <int EPI, int MT, bool W8 = false, bool A8 = false>
__global__ void __launch_bounds__(1024) k_gemm_oneshot(const ze_oneshot_args a) {
    static_assert(!A8 || W8, "fp8 activations go with fp8 weights");
    constexpr int D = 128;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int nb = blockIdx.x;                   // fragment block: 16 weight rows
    const int ns_all = a.K >> 5;
    const int base = ns_all >> 4, rem = ns_all & 15;
    const int s0 = wid * base + min(wid, rem), ns = base + (wid < rem ? 1 : 0);

    // ---- epilogue operands first (waves 0 .. MT-1 own row tile `wid` at the end): they ride under the main loads
    // output column of this lane in ORIGINAL order
    int col;
    if (EPI == OS_EPI_QKV) {
        const int h = nb >> 3, j = nb & 7;       // head, 8-dim group
        col = h * D + (fr < 8 ? 8 * j + fr : 64 + 8 * j + (fr - 8));
    } else {
        col = nb * 16 + fr;
    }
    float bias_v = 0.f;
    float res[4] = {0.f, 0.f, 0.f, 0.f};
    int seq[4] = {0, 0, 0, 0};
    if (wid < MT) {
        if (a.bias) bias_v = bf16_to_f32(a.bias[col]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = min(wid * 16 + fq * 4 + r, a.M - 1);
            if (EPI == OS_EPI_RESIDUAL) res[r] = bf16_to_f32(a.R[(size_t)row * a.ldr + col]);
            if (EPI == OS_EPI_QKV) seq[r] = a.seq_ids[row];
        }
    }

    // ---- the wave's K share: every load of a chunk of up to 4 slices in one batch
    os_f32x4 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = os_f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16_t* wp = a.Wf + ((size_t)nb * ns_all * 64 + lane) * (W8 ? 4 : 8);
    const float wsc = W8 ? a.wscale[col] : 1.0f;  // fragment row fr of this block IS output column `col`
    const bf16_t* ap[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) ap[i] = a.Xf + ((size_t)i * ns_all * 64 + lane) * (A8 ? 4 : 8);
    for (int c0 = 0; c0 < ns; c0 += 4) {
        os_bf16x8 fb[4], fa[4][MT];
        os_u32x2 fb8[4], fa8[4][MT];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const size_t k = (size_t)(s0 + min(c0 + c, ns - 1)) * 512;  // slices past the share re-read its last one
            if constexpr (W8) fb8[c] = __builtin_nontemporal_load(reinterpret_cast<const os_u32x2*>(wp + (k >> 1)));
            else fb[c] = __builtin_nontemporal_load(reinterpret_cast<const os_bf16x8*>(wp + k));
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                if constexpr (A8) fa8[c][i] = *reinterpret_cast<const os_u32x2*>(ap[i] + (k >> 1));
                else fa[c][i] = *reinterpret_cast<const os_bf16x8*>(ap[i] + k);
            }
        }
        if constexpr (W8 && !A8) {
#pragma unroll
            for (int c = 0; c < 4; ++c) fb[c] = os_deq_fp8x8(fb8[c], wsc);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c0 + c < ns) {  // wave-uniform
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    if constexpr (A8)
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(*reinterpret_cast<const long*>(&fa8[c][i]),
                                                                             *reinterpret_cast<const long*>(&fb8[c]), acc[i], 0, 0, 0);
                    else
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[c][i], fb[c], acc[i], 0, 0, 0);
                }
            }
    }

    // chain state of this lane's rows (QKV): a dependent load behind seq_ids, requested while the partials travel
    int ctx[4] = {0, 0, 0, 0}, pos[4] = {0, 0, 0, 0};
    if (EPI == OS_EPI_QKV && wid < MT) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            ctx[r] = a.st[seq[r]].ctx;
            pos[r] = ctx[r] + a.st[seq[r]].rope_delta;
        }
    }

    // ---- the sixteen K shares meet in LDS: red[wave][row tile][lane]; level 1: wave w adds shares 4g .. 4g+3 of tile t
    // (w = 4g + t when MT == 4; with fewer row tiles the spare waves idle), level 2: wave t adds the four group sums
    os_f32x4* red = reinterpret_cast<os_f32x4*>(smem);
#pragma unroll
    for (int i = 0; i < MT; ++i) red[(wid * MT + i) * 64 + lane] = acc[i];
    __syncthreads();
    os_f32x4* red2 = red + 16 * MT * 64;
    {
        const int g = wid >> 2, t = wid & 3;
        if (t < MT) {
            os_f32x4 v = red[((4 * g) * MT + t) * 64 + lane];
#pragma unroll
            for (int q = 1; q < 4; ++q) {
                const os_f32x4 u = red[((4 * g + q) * MT + t) * 64 + lane];
                v[0] += u[0];
                v[1] += u[1];
                v[2] += u[2];
                v[3] += u[3];
            }
            red2[(g * MT + t) * 64 + lane] = v;
        }
    }
    __syncthreads();
    if (wid >= MT) return;
    os_f32x4 v = red2[(0 * MT + wid) * 64 + lane];
#pragma unroll
    for (int g = 1; g < 4; ++g) {
        const os_f32x4 u = red2[(g * MT + wid) * 64 + lane];
        v[0] += u[0];
        v[1] += u[1];
        v[2] += u[2];
        v[3] += u[3];
    }

    if constexpr (A8) {  // fp8 x fp8 sums -> values: the activation row's and the weight row's powers of two
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= a.ascale[min(wid * 16 + fq * 4 + r, a.M - 1)] * wsc;
    }
    // ---- epilogue: v[r] -> row = wid*16 + fq*4 + r, column `col`
    if (EPI != OS_EPI_QKV) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = wid * 16 + fq * 4 + r;
            if (row >= a.M) continue;
            float o = bf16_round(v[r] + bias_v);
            if (EPI == OS_EPI_RESIDUAL) o = res[r] + o;
            a.C[(size_t)row * a.ldc + col] = f32_to_bf16(o);
        }
        return;
    }
    const int h = nb >> 3;                       // head over q | k | v
    const int jdim = (nb & 7) * 8 + (fr & 7);    // rotary index in [0, 64)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = wid * 16 + fq * 4 + r;
        const float mine = bf16_round(v[r] + bias_v);
        const float other = __shfl_xor(mine, 8, 64);  // the rotate_half partner (dims d and d +- 64) sits 8 columns away
        if (row >= a.M) continue;
        if (h >= a.heads + a.kv_heads) {          // v head: no rotation, straight into the cache
            a.vcache[(size_t)seq[r] * a.cache_seq_stride + ((size_t)(h - a.heads - a.kv_heads) * a.max_ctx + ctx[r]) * D +
                     (col - h * D)] = f32_to_bf16(mine);
            continue;
        }
        const float c = bf16_to_f32(a.cosT[(size_t)pos[r] * 64 + jdim]);
        const float sn = bf16_to_f32(a.sinT[(size_t)pos[r] * 64 + jdim]);
        // first half (fr < 8): x1*c + (-x2)*s ; second half: x2*c + x1*s   (each product rounded to bf16, then the sum)
        const float o = fr < 8 ? bf16_round(mine * c) + bf16_round(-other * sn) : bf16_round(mine * c) + bf16_round(other * sn);
        const bf16_t ob = f32_to_bf16(o);
        if (h < a.heads)
            a.C[(size_t)row * a.ldc + col] = ob;
        else
            a.kcache[(size_t)seq[r] * a.cache_seq_stride + ((size_t)(h - a.heads) * a.max_ctx + ctx[r]) * D + (col - h * D)] = ob;
    }
}

template <int EPI, bool W8, bool A8>
static void launch_oneshot_w(const ze_oneshot_args& a, hipStream_t s) {
    const int grid = a.N / 16;
#define ZE_OS_LAUNCH(MT)                                                                                              \
    do {                                                                                                              \
        const size_t lds = (size_t)(16 + 4) * MT * 64 * 16;                                                           \
        static bool attr_set = false;                                                                                 \
        if (!attr_set) {                                                                                              \
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_oneshot<EPI, MT, W8, A8>),                      \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                \
            attr_set = true;                                                                                          \
        }                                                                                                             \
        hipLaunchKernelGGL((k_gemm_oneshot<EPI, MT, W8, A8>), dim3(grid), dim3(1024), lds, s, a);                     \
    } while (0)
    if (a.M <= 16) ZE_OS_LAUNCH(1);
    else if (a.M <= 32) ZE_OS_LAUNCH(2);
    else ZE_OS_LAUNCH(4);
#undef ZE_OS_LAUNCH
}
template <int EPI>
static void launch_oneshot(const ze_oneshot_args& a, hipStream_t s) {
    if (a.wscale && a.ascale) launch_oneshot_w<EPI, true, true>(a, s);
    else if (a.wscale) launch_oneshot_w<EPI, true, false>(a, s);
    else launch_oneshot_w<EPI, false, false>(a, s);
}

// C = X W^T (+ bias) (+ residual): M <= 64, N % 16 == 0, K % 32 == 0
void ze_launch_gemm_oneshot(int epi, const bf16_t* Xf, const bf16_t* Wf, const bf16_t* bias, const bf16_t* R, int ldr,
                            bf16_t* C, int ldc, int M, int N, int K, hipStream_t s, const float* wscale, const float* ascale) {
    if (M <= 0 || N <= 0) return;
    ze_oneshot_args a;
    memset(&a, 0, sizeof(a));
    a.wscale = wscale;
    a.ascale = ascale;
    a.Xf = Xf; a.Wf = Wf; a.bias = bias; a.R = R; a.ldr = ldr; a.C = C; a.ldc = ldc; a.M = M; a.N = N; a.K = K;
    if (epi == ZE_EPI_RESIDUAL) launch_oneshot<OS_EPI_RESIDUAL>(a, s);
    else launch_oneshot<OS_EPI_BIAS>(a, s);
}

// qkv projection + M-RoPE + KV append of the batched decode step; Wf packed by ze_launch_pack_fragments(..., head_dim)
void ze_launch_qkv_rope_oneshot(const bf16_t* Xf, const bf16_t* Wf_perm, const bf16_t* bias, bf16_t* q_out, int ldq, int M,
                                int K, int heads, int kv_heads, const bf16_t* cosT, const bf16_t* sinT,
                                const ze_seq_dev* st, const int* seq_ids, bf16_t* kcache, bf16_t* vcache,
                                size_t cache_seq_stride, int max_ctx, hipStream_t s, const float* wscale, const float* ascale) {
    if (M <= 0) return;
    ze_oneshot_args a;
    memset(&a, 0, sizeof(a));
    a.wscale = wscale;
    a.ascale = ascale;
    a.Xf = Xf; a.Wf = Wf_perm; a.bias = bias; a.C = q_out; a.ldc = ldq; a.M = M; a.N = (heads + 2 * kv_heads) * 128; a.K = K;
    a.st = st; a.seq_ids = seq_ids; a.cosT = cosT; a.sinT = sinT; a.kcache = kcache; a.vcache = vcache;
    a.cache_seq_stride = cache_seq_stride; a.heads = heads; a.kv_heads = kv_heads; a.max_ctx = max_ctx;
    launch_oneshot<OS_EPI_QKV>(a, s);
}
