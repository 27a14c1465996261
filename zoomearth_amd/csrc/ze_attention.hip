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
// Flash-decoding over the KV cache of one chain: grid = (kv_heads, max_splits); a block owns one 64-token slice
// of the context for ALL q heads of its kv head (K/V are read once per kv head).  Three phases per 64-token round,
// each thread issuing all its global loads up front (the kernel is latency-, not bandwidth-bound: 1-2 MB per layer):
//   A  16-lane groups take 4 tokens each: 4 K + 4 V 16-B loads in flight per lane; 32 (token, head) partial dots per
//      lane are reduced across the 16 lanes with a halving butterfly (30 shuffles instead of 128); V goes to LDS.
//   B  softmax statistics per head over the 64 scores (one wave handles two heads, wavefront max / sum).
//   C  thread (head, 4-dim slice) accumulates sum_t p[t] * V[t] from LDS: no cross-thread reduction at the end.
// Partials (m, l, o[128]) per (split, head) go to a workspace; k_attn_decode_combine merges the splits.
// (Measured alternative, rejected: merging in the last-arriving slice block -- sc1 partial stores + ticket + agent
//  acquire, as the split-K GEMM does -- removed the combine launch but cost 30 ms more per question: the acquire
//  and the re-read of 90 KB of partials through memory are slower than the 4.8-us combine kernel.)
#include "ze_attn_decode.h"

__global__ void __launch_bounds__(256) k_attn_decode_split(const bf16_t* __restrict__ q, int q_row_stride,
                                                           const bf16_t* __restrict__ kcache,
                                                           const bf16_t* __restrict__ vcache, size_t cache_seq_stride,
                                                           const ze_seq_dev* __restrict__ st_base,
                                                           const int* __restrict__ seq_ids, int heads, int kv_heads,
                                                           int max_ctx, float scale_log2e, float* __restrict__ ws,
                                                           int max_splits, unsigned* __restrict__ tickets,
                                                           bf16_t* __restrict__ out, int out_row_stride) {
    // chain of this block (grid.z): batched decode indexes the chain table, single-chain decode passes its state
    const int bz = blockIdx.z;
    const ze_seq_dev* st = seq_ids ? st_base + seq_ids[bz] : st_base;
    if (seq_ids) {
        kcache += (size_t)seq_ids[bz] * cache_seq_stride;
        vcache += (size_t)seq_ids[bz] * cache_seq_stride;
    }
    __shared__ ad_split_lds L;
    const int ctx = st->ctx + 1, kvh = blockIdx.x;
    int chunk, nsplit;
    split_geometry(ctx, max_splits, chunk, nsplit);
    if ((int)blockIdx.y >= nsplit) return;  // workgroup-uniform: no slice, no ticket
    float* wsb = ws + (size_t)bz * max_splits * heads * AD_STRIDE;
    attn_split_body<false, true>(L, q + (size_t)bz * q_row_stride, kcache, vcache, ctx, kvh, blockIdx.y, heads, kv_heads,
                                 max_ctx, scale_log2e, wsb, max_splits);
    // ---- merge by the last-arriving slice of this (chain, kv head).  Hand-off without fences: the partials above
    // went out write-through (sc1), every storing wave drains, the workgroup barriers, ONE lane takes a ticket
    // (relaxed agent-scope add on one unsharded counter); the workgroup whose add came last reads every partial
    // with sc1 loads, in slice order, so the result does not depend on who is last.  The ticket returns to 0 for
    // the next launch.  (With an agent-scope acquire fence + plain loads in place of the sc1 pair this merge cost
    // 30 ms per question MORE than a separate merge launch; in this form it replaces that 4.8-us launch.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned* flag = reinterpret_cast<unsigned*>(&L.sM[0]);  // the slice LDS is dead now
    if (threadIdx.x == 0) {
        unsigned* t = tickets + (size_t)bz * kv_heads + kvh;
        const unsigned old = __hip_atomic_fetch_add(t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned last = old == (unsigned)nsplit - 1u;
        if (last) __hip_atomic_store(t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = last;
    }
    __syncthreads();
    if (*flag == 0u) return;
    __syncthreads();  // everybody has read the flag before the merge reuses the LDS
    float* sW = reinterpret_cast<float*>(&L.sV[0][0]);
    attn_merge_group(sW, sW + AD_GMAX * 64, wsb, ctx, kvh, heads, kv_heads, max_splits,
                     out + (size_t)bz * out_row_stride);
}

void ze_launch_attn_decode(const bf16_t* q, int q_row_stride, const bf16_t* kcache, const bf16_t* vcache,
                           size_t cache_seq_stride, bf16_t* out, int out_row_stride, const ze_seq_dev* st,
                           const int* seq_ids, int n, int heads, int kv_heads, int D, int max_ctx, float scale,
                           float* ws_partial, int max_splits, unsigned* tickets, hipStream_t s) {
    (void)D;
    const float sl = scale * 1.4426950408889634f;
    k_attn_decode_split<<<dim3(kv_heads, max_splits, n), 256, 0, s>>>(q, q_row_stride, kcache, vcache, cache_seq_stride,
                                                                       st, seq_ids, heads, kv_heads, max_ctx, sl,
                                                                       ws_partial, max_splits, tickets, out,
                                                                       out_row_stride);
}
