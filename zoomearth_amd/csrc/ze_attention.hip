// Attention kernels.
//  * k_flash_attn<D, CAUSAL>: varlen flash attention on MFMA 16x16x32 bf16 for the ViT (D = 80, window / full
//    segments, non-causal) and for the LLM prefill (D = 128, causal GQA; K19).  Tiles come from a
//    host-built list so one launch covers ragged segments.
//  * k_attn_decode_split / k_attn_decode_combine: one-token GQA attention over the KV cache (K19 at q = 1):
//    flash-decoding split over the context so K/V are read once per kv head (shared by its q heads).
// Replaces: SDPA / eager_attention_forward as called at HF:models/qwen2_5_vl/modeling_qwen2_5_vl.py:225-291
// (vision) and :670-689 (text); softmax in fp32, P rounded to bf16 for the PV MFMA (as eager does, :202).
#include "ze_kernels.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define FA_BQ 64
#define FA_BK 64

template <int D, int CAUSAL>
__global__ void __launch_bounds__(256) k_flash_attn(const bf16_t* __restrict__ q, int q_rs, int q_hs,
                                                    const bf16_t* __restrict__ k, int k_rs, int k_hs,
                                                    const bf16_t* __restrict__ v, int v_rs, int v_hs,
                                                    bf16_t* __restrict__ o, int o_rs, int o_hs,
                                                    const int4* __restrict__ tiles, int group, float scale_log2e,
                                                    int q_pos_offset) {
    constexpr int DK = (D + 31) / 32 * 32;  // padded head dim for the QK^T k-steps
    constexpr int QCH = DK / 8;             // 16-B chunks per (padded) row
    constexpr int DCH = D / 8;              // real chunks per row
    constexpr int NV = D / 16;              // output n-tiles
    // LDS images (16-B units): [chunk][row ^ (chunk&7)]
    __shared__ uint4 sQ[QCH * FA_BQ];
    __shared__ uint4 sK[QCH * FA_BK];
    __shared__ uint4 sVt[(FA_BK / 8) * D];       // [key-chunk][d ^ (kc&7)] : 8 keys x bf16 per unit
    __shared__ uint4 sP[4][(FA_BK / 8) * 16];    // per wave [key-chunk][row ^ (kc&7)]

    const int4 tile = tiles[blockIdx.x];
    const int q0 = tile.x, q1 = tile.y, kv0 = tile.z, kv1 = tile.w;
    const int head = blockIdx.y, kvh = head / group;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;

    // ---- stage Q (zero padded rows / chunks)
    for (int i = tid; i < QCH * FA_BQ; i += 256) {
        const int row = i / QCH, ch = i % QCH;
        uint4 val = make_uint4(0, 0, 0, 0);
        if (q0 + row < q1 && ch < DCH)
            val = *reinterpret_cast<const uint4*>(q + (size_t)(q0 + row) * q_rs + (size_t)head * q_hs + ch * 8);
        sQ[ch * FA_BQ + (row ^ (ch & 7))] = val;
    }

    f32x4 oacc[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) oacc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run[4], l_run[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        m_run[r] = -INFINITY;
        l_run[r] = 0.f;
    }
    int kv_hi = kv1;
    if (CAUSAL) kv_hi = min(kv1, q1 + q_pos_offset);  // keys beyond the last query position are never visible

    for (int kt = kv0; kt < kv_hi; kt += FA_BK) {
        __syncthreads();  // previous tile's LDS reads done (also orders the Q staging before first use)
        // ---- stage K tile and V^T tile
        for (int i = tid; i < QCH * FA_BK; i += 256) {
            const int row = i / QCH, ch = i % QCH;
            uint4 val = make_uint4(0, 0, 0, 0);
            if (kt + row < kv_hi && ch < DCH)
                val = *reinterpret_cast<const uint4*>(k + (size_t)(kt + row) * k_rs + (size_t)kvh * k_hs + ch * 8);
            sK[ch * FA_BK + (row ^ (ch & 7))] = val;
        }
        for (int i = tid; i < DCH * FA_BK; i += 256) {
            const int key = i / DCH, ch = i % DCH;
            uint4 val = make_uint4(0, 0, 0, 0);
            if (kt + key < kv_hi)
                val = *reinterpret_cast<const uint4*>(v + (size_t)(kt + key) * v_rs + (size_t)kvh * v_hs + ch * 8);
            const uint32_t u[4] = {val.x, val.y, val.z, val.w};
            bf16_t* base = reinterpret_cast<bf16_t*>(sVt);
            const int kc = key >> 3, ki = key & 7;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int d = ch * 8 + e;
                const bf16_t x = (bf16_t)((e & 1) ? (u[e >> 1] >> 16) : (u[e >> 1] & 0xffff));
                base[((size_t)(kc * D + (d ^ (kc & 7)))) * 8 + ki] = x;
            }
        }
        __syncthreads();

        // ---- S = Q K^T for this wave's 16 rows x 64 keys
        f32x4 sacc[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) sacc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < DK / 32; ++ks) {
            const int ch = ks * 4 + fq;
            const int qrow = wid * 16 + fr;
            const uint4 qa = sQ[ch * FA_BQ + (qrow ^ (ch & 7))];
            const bf16x8 fa = *reinterpret_cast<const bf16x8*>(&qa);
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int krow = n * 16 + fr;
                const uint4 kb = sK[ch * FA_BK + (krow ^ (ch & 7))];
                sacc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, *reinterpret_cast<const bf16x8*>(&kb), sacc[n],
                                                                 0, 0, 0);
            }
        }
        // ---- mask + online softmax (rows = wid*16 + fq*4 + r, keys = n*16 + fr)
        float p[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qi = q0 + wid * 16 + fq * 4 + r;
            float mx = -INFINITY;
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int kj = kt + n * 16 + fr;
                bool ok = kj < kv_hi;
                if (CAUSAL) ok = ok && (kj <= qi + q_pos_offset);
                const float sv = ok ? sacc[n][r] * scale_log2e : -INFINITY;
                p[n][r] = sv;
                mx = fmaxf(mx, sv);
            }
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
            const float m_new = fmaxf(m_run[r], mx);
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = exp2f(m_run[r] - m_use);  // m_run = -inf -> 0
            float rs = 0.f;
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const float e = exp2f(p[n][r] - m_use);
                p[n][r] = e;
                rs += e;
            }
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) rs += __shfl_xor(rs, off, 64);
            l_run[r] = l_run[r] * alpha + rs;
            m_run[r] = m_new;
#pragma unroll
            for (int j = 0; j < NV; ++j) oacc[j][r] *= alpha;
        }
        // ---- P (bf16) -> this wave's LDS image as an A operand
        {
            bf16_t* pb = reinterpret_cast<bf16_t*>(sP[wid]);
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = fq * 4 + r, key = n * 16 + fr;
                    const int kc = key >> 3;
                    pb[((size_t)(kc * 16 + (row ^ (kc & 7)))) * 8 + (key & 7)] = f32_to_bf16(p[n][r]);
                }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's own LDS writes have landed
        __builtin_amdgcn_wave_barrier();
        // ---- O += P V
#pragma unroll
        for (int ks = 0; ks < FA_BK / 32; ++ks) {
            const int kc = ks * 4 + fq;
            const uint4 pa = sP[wid][kc * 16 + (fr ^ (kc & 7))];
            const bf16x8 fa = *reinterpret_cast<const bf16x8*>(&pa);
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int d = j * 16 + fr;
                const uint4 vb = sVt[kc * D + (d ^ (kc & 7))];
                oacc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, *reinterpret_cast<const bf16x8*>(&vb), oacc[j],
                                                                 0, 0, 0);
            }
        }
    }
    // ---- normalise and store
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int qi = q0 + wid * 16 + fq * 4 + r;
        if (qi >= q1) continue;
        const float inv = l_run[r] > 0.f ? 1.0f / l_run[r] : 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j)
            o[(size_t)qi * o_rs + (size_t)head * o_hs + j * 16 + fr] = f32_to_bf16(oacc[j][r] * inv);
    }
}

void ze_launch_flash_attn(int D, int causal, const bf16_t* q, int q_rs, int q_hs, const bf16_t* k, int k_rs,
                          int k_hs, const bf16_t* v, int v_rs, int v_hs, bf16_t* o, int o_rs, int o_hs,
                          const int4* tiles, int n_tiles, int heads, int group, float scale, int q_pos_offset,
                          hipStream_t s) {
    if (n_tiles == 0) return;
    const float sl = scale * 1.4426950408889634f;
    dim3 grid(n_tiles, heads);
#define FA_LAUNCH(DD, CC)                                                                                      \
    hipLaunchKernelGGL((k_flash_attn<DD, CC>), grid, dim3(256), 0, s, q, q_rs, q_hs, k, k_rs, k_hs, v, v_rs, v_hs, \
                       o, o_rs, o_hs, tiles, group, sl, q_pos_offset)
    if (D == 80) {
        if (causal) FA_LAUNCH(80, 1); else FA_LAUNCH(80, 0);
    } else {
        if (causal) FA_LAUNCH(128, 1); else FA_LAUNCH(128, 0);
    }
#undef FA_LAUNCH
}

// ------------------------------------------------------------------ decode attention (D = 128)
#define AD_GMAX 8
#define AD_STRIDE 132  // floats per (split, head) partial: m, l, pad, pad, o[128]

__device__ __forceinline__ void split_geometry(int ctx, int max_splits, int& chunk, int& nsplit) {
    chunk = (ctx + max_splits - 1) / max_splits;
    chunk = (chunk + 15) & ~15;
    if (chunk < 64) chunk = 64;
    nsplit = (ctx + chunk - 1) / chunk;
}

__global__ void __launch_bounds__(256) k_attn_decode_split(const bf16_t* __restrict__ q,
                                                           const bf16_t* __restrict__ kcache,
                                                           const bf16_t* __restrict__ vcache,
                                                           const ze_seq_dev* __restrict__ st, int heads, int kv_heads,
                                                           int max_ctx, float scale_log2e, float* __restrict__ ws,
                                                           int max_splits) {
    constexpr int D = 128;
    const int G = heads / kv_heads;
    const int ctx = st->ctx + 1;
    int chunk, nsplit;
    split_geometry(ctx, max_splits, chunk, nsplit);
    const int split = blockIdx.y, kvh = blockIdx.x;
    if (split >= nsplit) return;
    const int t0 = split * chunk, t1 = min(ctx, t0 + chunk);
    const int tid = threadIdx.x, gid = tid >> 4, li = tid & 15, lane = tid & 63, wid = tid >> 6;

    float qv[AD_GMAX][8];
#pragma unroll
    for (int g = 0; g < AD_GMAX; ++g) {
        if (g < G) {
            const uint4 u = *reinterpret_cast<const uint4*>(q + (size_t)(kvh * G + g) * D + li * 8);
            const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                qv[g][2 * j] = bf16lo(w[j]) * scale_log2e;
                qv[g][2 * j + 1] = bf16hi(w[j]) * scale_log2e;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) qv[g][j] = 0.f;
        }
    }
    float m[AD_GMAX], l[AD_GMAX], oa[AD_GMAX][8];
#pragma unroll
    for (int g = 0; g < AD_GMAX; ++g) {
        m[g] = -INFINITY;
        l[g] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) oa[g][j] = 0.f;
    }
    const bf16_t* kb = kcache + (size_t)kvh * max_ctx * D + li * 8;
    const bf16_t* vb = vcache + (size_t)kvh * max_ctx * D + li * 8;
    for (int t = t0 + gid; t < t1; t += 16) {
        const uint4 ku = *reinterpret_cast<const uint4*>(kb + (size_t)t * D);
        const uint4 vu = *reinterpret_cast<const uint4*>(vb + (size_t)t * D);
        const uint32_t kw[4] = {ku.x, ku.y, ku.z, ku.w}, vw[4] = {vu.x, vu.y, vu.z, vu.w};
        float kf[8], vf[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            kf[2 * j] = bf16lo(kw[j]);
            kf[2 * j + 1] = bf16hi(kw[j]);
            vf[2 * j] = bf16lo(vw[j]);
            vf[2 * j + 1] = bf16hi(vw[j]);
        }
#pragma unroll
        for (int g = 0; g < AD_GMAX; ++g) {
            if (g >= G) break;
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) s = fmaf(qv[g][j], kf[j], s);
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
            const float mn = fmaxf(m[g], s);
            const float alpha = exp2f(m[g] - mn);
            const float p = exp2f(s - mn);
            l[g] = l[g] * alpha + p;
            m[g] = mn;
#pragma unroll
            for (int j = 0; j < 8; ++j) oa[g][j] = oa[g][j] * alpha + p * vf[j];
        }
    }
    // ---- combine the 4 token-groups of a wave (lanes li, li+16, li+32, li+48) with shuffles
#pragma unroll
    for (int g = 0; g < AD_GMAX; ++g) {
        if (g >= G) break;
#pragma unroll
        for (int off = 16; off <= 32; off <<= 1) {
            const float m2 = __shfl_xor(m[g], off, 64), l2 = __shfl_xor(l[g], off, 64);
            const float mn = fmaxf(m[g], m2);
            const float mu = (mn == -INFINITY) ? 0.f : mn;
            const float a1 = exp2f(m[g] - mu), a2 = exp2f(m2 - mu);
            l[g] = l[g] * a1 + l2 * a2;
#pragma unroll
            for (int j = 0; j < 8; ++j) oa[g][j] = oa[g][j] * a1 + __shfl_xor(oa[g][j], off, 64) * a2;
            m[g] = mn;
        }
    }
    // ---- combine the 4 waves through LDS
    __shared__ float sm[4][AD_GMAX][AD_STRIDE];
    if (lane < 16) {
#pragma unroll
        for (int g = 0; g < AD_GMAX; ++g) {
            if (g >= G) break;
            if (li == 0) {
                sm[wid][g][0] = m[g];
                sm[wid][g][1] = l[g];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) sm[wid][g][4 + li * 8 + j] = oa[g][j];
        }
    }
    __syncthreads();
    for (int i = tid; i < G * D; i += 256) {
        const int g = i / D, d = i % D;
        float mn = -INFINITY;
#pragma unroll
        for (int w = 0; w < 4; ++w) mn = fmaxf(mn, sm[w][g][0]);
        const float mu = (mn == -INFINITY) ? 0.f : mn;
        float ls = 0.f, os = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float a = exp2f(sm[w][g][0] - mu);
            ls += sm[w][g][1] * a;
            os += sm[w][g][4 + d] * a;
        }
        float* dst = ws + ((size_t)(split * heads + kvh * G + g)) * AD_STRIDE;
        if (d == 0) {
            dst[0] = mn;
            dst[1] = ls;
        }
        dst[4 + d] = os;
    }
}

__global__ void __launch_bounds__(128) k_attn_decode_combine(const float* __restrict__ ws,
                                                             const ze_seq_dev* __restrict__ st, int heads,
                                                             int max_splits, bf16_t* __restrict__ out) {
    constexpr int D = 128;
    const int ctx = st->ctx + 1;
    int chunk, nsplit;
    split_geometry(ctx, max_splits, chunk, nsplit);
    const int h = blockIdx.x, d = threadIdx.x;
    float mn = -INFINITY;
    for (int s = 0; s < nsplit; ++s) mn = fmaxf(mn, ws[((size_t)(s * heads + h)) * AD_STRIDE]);
    float ls = 0.f, os = 0.f;
    for (int s = 0; s < nsplit; ++s) {
        const float* p = ws + ((size_t)(s * heads + h)) * AD_STRIDE;
        const float a = exp2f(p[0] - mn);
        ls += p[1] * a;
        os += p[4 + d] * a;
    }
    out[(size_t)h * D + d] = f32_to_bf16(os / ls);
}

void ze_launch_attn_decode(const bf16_t* q, const bf16_t* kcache, const bf16_t* vcache, bf16_t* out,
                           const ze_seq_dev* st, int heads, int kv_heads, int D, int max_ctx, float scale,
                           float* ws_partial, int max_splits, hipStream_t s) {
    (void)D;
    const float sl = scale * 1.4426950408889634f;
    k_attn_decode_split<<<dim3(kv_heads, max_splits), 256, 0, s>>>(q, kcache, vcache, st, heads, kv_heads, max_ctx,
                                                                    sl, ws_partial, max_splits);
    k_attn_decode_combine<<<heads, 128, 0, s>>>(ws_partial, st, heads, max_splits, out);
}
