// Device bodies of the one-token (decode) attention, shared by the stand-alone kernels (ze_attention.hip) and the
// fused per-layer kernel (ze_mega.hip).  Same arithmetic in both; the FRESH form moves cross-workgroup data with
// sc1 (write-through / L1-bypassing) accesses so it can be handed over inside a running launch.
#pragma once
#include "ze_kernels.h"

#define AD_GMAX 8
#define AD_STRIDE 132  // floats per (split, head) partial: m, l, pad, pad, o[128]
#define AD_TOK 64

__device__ __forceinline__ void split_geometry(int ctx, int max_splits, int& chunk, int& nsplit) {
    chunk = (ctx + max_splits - 1) / max_splits;
    chunk = (chunk + AD_TOK - 1) / AD_TOK * AD_TOK;
    nsplit = (ctx + chunk - 1) / chunk;
}

// LDS of one slice block (carved by the caller so fused kernels can share their LDS)
struct ad_split_lds {
    __attribute__((aligned(16))) bf16_t sV[AD_TOK][128];  // 16 KB
    float sS[AD_TOK][AD_GMAX];                              // scores, then probabilities
    float sM[AD_GMAX], sL[AD_GMAX];
};

// FRESH: q / K / V / partials cross workgroups INSIDE the running launch (fused layer kernel): every such access
// is a write-through (sc1) store or an L1-bypassing (sc1) load.  Arithmetic is identical in both forms.
// Addresses are (wave-uniform base, per-lane byte offset): the buffer descriptor of the sc1 forms must live in
// SGPRs -- a per-lane base pointer makes hipcc wrap every access in a 64-trip readfirstlane waterfall loop
// (measured: the fused kernel 4x slower than the four launches it replaces).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ad_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
template <bool FRESH>
__device__ __forceinline__ uint4 ad_load16(const void* base, uint32_t off) {
    if (FRESH) {
        typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
        const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(ad_rsrc(base), off, 0, 16);
        return make_uint4(v.x, v.y, v.z, v.w);
    }
    return *reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(base) + off);
}
template <bool FRESH>
__device__ __forceinline__ float ad_load4(const void* base, uint32_t off) {
    if (FRESH) return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ad_rsrc(base), off, 0, 16));
    return *reinterpret_cast<const float*>(reinterpret_cast<const uint8_t*>(base) + off);
}
template <bool FRESH>
__device__ __forceinline__ void ad_store2(void* base, uint32_t off, bf16_t v) {
    if (FRESH)
        __builtin_amdgcn_raw_buffer_store_b16(v, ad_rsrc(base), off, 0, 16);
    else
        *reinterpret_cast<bf16_t*>(reinterpret_cast<uint8_t*>(base) + off) = v;
}
template <bool FRESH>
__device__ __forceinline__ void ad_store16(void* base, uint32_t off, float a, float b, float c, float d) {
    if (FRESH) {
        typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
        u32x4_t v;
        v.x = __float_as_uint(a);
        v.y = __float_as_uint(b);
        v.z = __float_as_uint(c);
        v.w = __float_as_uint(d);
        __builtin_amdgcn_raw_buffer_store_b128(v, ad_rsrc(base), off, 0, 16);
    } else {
        *reinterpret_cast<float4*>(reinterpret_cast<uint8_t*>(base) + off) = make_float4(a, b, c, d);
    }
}

// FRESH: q and the newest K/V row were written inside this launch (fused layer kernel).  PUB: the partials are
// published write-through (sc1) because they are merged inside this launch (by the fused kernel's merge phase or by
// the last-arriving slice workgroup of the stand-alone kernel).
template <bool FRESH, bool PUB = FRESH>
__device__ __forceinline__ void attn_split_body(ad_split_lds& L, const bf16_t* __restrict__ q,
                                                const bf16_t* __restrict__ kcache, const bf16_t* __restrict__ vcache,
                                                int ctx, int kvh, int split, int heads, int kv_heads, int max_ctx,
                                                float scale_log2e, float* __restrict__ ws, int max_splits) {
    constexpr int D = 128;
    auto& sV = L.sV;
    auto& sS = L.sS;
    auto& sM = L.sM;
    auto& sL = L.sL;
    const int G = heads / kv_heads;
    int chunk, nsplit;
    split_geometry(ctx, max_splits, chunk, nsplit);
    if (split >= nsplit) return;
    const int t0 = split * chunk, t1 = min(ctx, t0 + chunk);
    const int tid = threadIdx.x, gid = tid >> 4, li = tid & 15, lane = tid & 63, wid = tid >> 6;

    // q of the (up to) 8 heads of this kv head, this lane's 8 dims, pre-scaled by scale*log2(e)
    float qv[AD_GMAX][8];
#pragma unroll
    for (int g = 0; g < AD_GMAX; ++g) {
        uint4 u = make_uint4(0, 0, 0, 0);
        if (g < G) u = ad_load16<FRESH>(q, (uint32_t)(((kvh * G + g) * D + li * 8) * 2));
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            qv[g][2 * j] = bf16lo(w[j]) * scale_log2e;
            qv[g][2 * j + 1] = bf16hi(w[j]) * scale_log2e;
        }
    }
    const bf16_t* kb = kcache + (size_t)kvh * max_ctx * D;  // wave-uniform bases, per-lane byte offsets
    const bf16_t* vb = vcache + (size_t)kvh * max_ctx * D;
    // phase-C ownership: head og, dims od..od+3
    const int og = tid >> 5, od = (tid & 31) * 4;
    float m_run = -INFINITY, l_run = 0.f, o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;

    for (int base = t0; base < t1; base += AD_TOK) {
        // ---------------- phase A: loads first
        // Only the newest row was written inside this launch (sc1 stores of the fused QKV phase) and is read with
        // sc1 loads; every older row comes from earlier launches and is read the way the stand-alone kernel reads
        // it.  The newest row is fetched once per lane up front (workgroup-uniform branch) and selected in, so the
        // eight row loads below stay free of per-lane control flow.
        uint4 kn = make_uint4(0, 0, 0, 0), vn = make_uint4(0, 0, 0, 0);
        const int tn = ctx - 1;
        if (FRESH && tn >= base && tn < base + AD_TOK) {
            const uint32_t off = (uint32_t)((tn * D + li * 8) * 2);
            kn = ad_load16<true>(kb, off);
            vn = ad_load16<true>(vb, off);
        }
        uint4 ku[4], vu[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int t = base + gid + 16 * i;
            const bool ok = t < t1;
            ku[i] = make_uint4(0, 0, 0, 0);
            vu[i] = make_uint4(0, 0, 0, 0);
            if (ok) {
                const uint32_t off = (uint32_t)((t * D + li * 8) * 2);
                ku[i] = ad_load16<false>(kb, off);
                vu[i] = ad_load16<false>(vb, off);
            }
            if (FRESH && t == tn) {
                ku[i] = kn;
                vu[i] = vn;
            }
        }
        float v[32];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t kw[4] = {ku[i].x, ku[i].y, ku[i].z, ku[i].w};
            float kf[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                kf[2 * j] = bf16lo(kw[j]);
                kf[2 * j + 1] = bf16hi(kw[j]);
            }
#pragma unroll
            for (int g = 0; g < AD_GMAX; ++g) {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) s = fmaf(qv[g][j], kf[j], s);
                v[i * 8 + g] = s;
            }
            *reinterpret_cast<uint4*>(&sV[gid + 16 * i][li * 8]) = vu[i];
        }
        // halving butterfly over the 16 lanes of the group: lane li ends with the totals of idx 2*li, 2*li+1
        {
            const bool b8 = li & 8, b4 = li & 4, b2 = li & 2, b1 = li & 1;
            float w16[16], w8[8], w4[4], w2[2];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const float keep = b8 ? v[16 + k] : v[k], send = b8 ? v[k] : v[16 + k];
                w16[k] = keep + __shfl_xor(send, 8, 64);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float keep = b4 ? w16[8 + k] : w16[k], send = b4 ? w16[k] : w16[8 + k];
                w8[k] = keep + __shfl_xor(send, 4, 64);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float keep = b2 ? w8[4 + k] : w8[k], send = b2 ? w8[k] : w8[4 + k];
                w4[k] = keep + __shfl_xor(send, 2, 64);
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float keep = b1 ? w4[2 + k] : w4[k], send = b1 ? w4[k] : w4[2 + k];
                w2[k] = keep + __shfl_xor(send, 1, 64);
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int idx = 2 * li + k, i = idx >> 3, g = idx & 7;
                const int tl = gid + 16 * i;
                sS[tl][g] = (base + tl < t1) ? w2[k] : -INFINITY;
            }
        }
        __syncthreads();
        // ---------------- phase B: per-head max / sum over the 64 scores; wave w handles heads 2w, 2w+1
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int g = wid * 2 + hh;
            const float s = sS[lane][g];
            const float mx = wave_max(s);
            const float p = (mx == -INFINITY) ? 0.f : exp2f(s - mx);
            const float sum = wave_sum(p);
            sS[lane][g] = p;
            if (lane == 0) {
                sM[g] = mx;
                sL[g] = sum;
            }
        }
        __syncthreads();
        // ---------------- phase C: o[og][od..od+3] += sum_t p[t][og] * V[t][od..]
        {
            const float mr = sM[og], lr = sL[og];
            const float mn = fmaxf(m_run, mr);
            const float mu = (mn == -INFINITY) ? 0.f : mn;
            const float a_old = exp2f(m_run - mu), a_new = exp2f(mr - mu);
            float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
#pragma unroll 16
            for (int t = 0; t < AD_TOK; ++t) {
                const float p = sS[t][og];
                const uint2 vv = *reinterpret_cast<const uint2*>(&sV[t][od]);
                c0 = fmaf(p, bf16lo(vv.x), c0);
                c1 = fmaf(p, bf16hi(vv.x), c1);
                c2 = fmaf(p, bf16lo(vv.y), c2);
                c3 = fmaf(p, bf16hi(vv.y), c3);
            }
            o0 = o0 * a_old + c0 * a_new;
            o1 = o1 * a_old + c1 * a_new;
            o2 = o2 * a_old + c2 * a_new;
            o3 = o3 * a_old + c3 * a_new;
            l_run = l_run * a_old + lr * a_new;
            m_run = mn;
        }
        __syncthreads();  // sS / sV are rewritten by the next round
    }
    if (og < G) {
        const uint32_t dst = (uint32_t)((split * heads + kvh * G + og) * AD_STRIDE * 4);
        if ((tid & 31) == 0) ad_store16<PUB>(ws, dst, m_run, l_run, 0.f, 0.f);
        ad_store16<PUB>(ws, dst + (4 + od) * 4, o0, o1, o2, o3);
    }
}

// grid = heads, 128 threads (one per output dim): split weights are computed lane-parallel by the first wave.
// merge of the slices of head h by threads 0..127 of a block (d = thread = output dim); sW: 64 floats, sInv: 1 float
template <bool FRESH>
__device__ __forceinline__ void attn_combine_body(float* sW, float* sInvp, const float* __restrict__ ws, int ctx, int h,
                                                  int d, int heads, int max_splits, bf16_t* __restrict__ out) {
    constexpr int D = 128;
    float& sInv = *sInvp;
    int chunk, nsplit;
    split_geometry(ctx, max_splits, chunk, nsplit);
    if (d < 64) {
        float m = -INFINITY, l = 0.f;
        if (d < nsplit) {
            const uint4 ml = ad_load16<FRESH>(ws, (uint32_t)((d * heads + h) * AD_STRIDE * 4));
            m = __uint_as_float(ml.x);
            l = __uint_as_float(ml.y);
        }
        const float mx = wave_max(m);
        const float w = (m == -INFINITY) ? 0.f : exp2f(m - mx);
        const float tot = wave_sum(w * l);
        sW[d] = w;
        if (d == 0) sInv = tot > 0.f ? 1.0f / tot : 0.f;
    }
    __syncthreads();
    float acc = 0.f;
    const uint32_t p = (uint32_t)((h * AD_STRIDE + 4 + d) * 4);
    const uint32_t step = (uint32_t)(heads * AD_STRIDE * 4);
    int s = 0;
    for (; s + 8 <= nsplit; s += 8) {
        float x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = ad_load4<FRESH>(ws, p + (s + u) * step);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = fmaf(sW[s + u], x[u], acc);
    }
    for (; s < nsplit; ++s) acc = fmaf(sW[s], ad_load4<FRESH>(ws, p + s * step), acc);
    ad_store2<FRESH>(out, (uint32_t)((h * D + d) * 2), f32_to_bf16(acc * sInv));
}

// Merge of the slices of ALL q heads of one kv head by one 256-thread workgroup (the last-arriving slice workgroup):
// wave w computes the slice weights of heads 2w, 2w+1 (lane = slice), then thread (head og, dims od..od+3)
// accumulates in slice order -- per output element the arithmetic of attn_combine_body.  Partials are read with
// sc1 loads (they were published sc1 inside this launch).  sW: [AD_GMAX][64] floats, sInv: [AD_GMAX].
__device__ __forceinline__ void attn_merge_group(float* sW, float* sInv, const float* __restrict__ ws, int ctx, int kvh,
                                                 int heads, int kv_heads, int max_splits, bf16_t* __restrict__ out) {
    constexpr int D = 128;
    const int G = heads / kv_heads;
    int chunk, nsplit;
    split_geometry(ctx, max_splits, chunk, nsplit);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int og = tid >> 5, od = (tid & 31) * 4;
    const int hc = kvh * G + min(og, G - 1);  // idle threads (og >= G) shadow the last head: no branch around loads
    const uint32_t p = (uint32_t)((hc * AD_STRIDE + 4 + od) * 4), step = (uint32_t)(heads * AD_STRIDE * 4);
    // Everything the merge needs is requested at once -- the (m, l) pairs and the first 24 slices of this thread's
    // four output dims (contexts up to 1536 tokens) -- so the last arriver pays one memory round trip, not one per
    // dependent step.
    uint4 ml[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int g = min(wid * 2 + hh, G - 1);
        ml[hh] = ad_load16<true>(ws, (uint32_t)((min(lane, nsplit - 1) * heads + kvh * G + g) * AD_STRIDE * 4));
    }
    constexpr int MB = 24;
    uint4 x0[MB];
#pragma unroll
    for (int u = 0; u < MB; ++u) x0[u] = ad_load16<true>(ws, p + min(u, nsplit - 1) * step);
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int g = wid * 2 + hh;
        if (g < G) {  // wave-uniform
            const float m = lane < nsplit ? __uint_as_float(ml[hh].x) : -INFINITY;
            const float l = lane < nsplit ? __uint_as_float(ml[hh].y) : 0.f;
            const float mx = wave_max(m);
            const float w = (m == -INFINITY) ? 0.f : exp2f(m - mx);
            const float tot = wave_sum(w * l);
            sW[g * 64 + lane] = w;
            if (lane == 0) sInv[g] = tot > 0.f ? 1.0f / tot : 0.f;
        }
    }
    __syncthreads();
    if (og >= G) return;
    const int h = kvh * G + og;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int u = 0; u < MB; ++u) {
        if (u < nsplit) {
            const float w = sW[og * 64 + u];
            a0 = fmaf(w, __uint_as_float(x0[u].x), a0);
            a1 = fmaf(w, __uint_as_float(x0[u].y), a1);
            a2 = fmaf(w, __uint_as_float(x0[u].z), a2);
            a3 = fmaf(w, __uint_as_float(x0[u].w), a3);
        }
    }
    int s = MB;
    for (; s + 8 <= nsplit; s += 8) {
        uint4 x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = ad_load16<true>(ws, p + (s + u) * step);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float w = sW[og * 64 + s + u];
            a0 = fmaf(w, __uint_as_float(x[u].x), a0);
            a1 = fmaf(w, __uint_as_float(x[u].y), a1);
            a2 = fmaf(w, __uint_as_float(x[u].z), a2);
            a3 = fmaf(w, __uint_as_float(x[u].w), a3);
        }
    }
    for (; s < nsplit; ++s) {
        const uint4 x = ad_load16<true>(ws, p + s * step);
        const float w = sW[og * 64 + s];
        a0 = fmaf(w, __uint_as_float(x.x), a0);
        a1 = fmaf(w, __uint_as_float(x.y), a1);
        a2 = fmaf(w, __uint_as_float(x.z), a2);
        a3 = fmaf(w, __uint_as_float(x.w), a3);
    }
    const float inv = sInv[og];
    uint2 o;
    o.x = pack_bf16x2(a0 * inv, a1 * inv);
    o.y = pack_bf16x2(a2 * inv, a3 * inv);
    *reinterpret_cast<uint2*>(out + (size_t)h * D + od) = o;
}
