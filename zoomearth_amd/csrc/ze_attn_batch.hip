// Decode attention of the BATCHED step (many chains per launch): K/V streamed through an LDS-DMA double buffer.
//
// The single-chain kernel (ze_attn_decode.hip) cuts a context into 64-token slices, one workgroup each: at 64 chains
// that is ~2300 short-lived workgroups per layer, each paying the chain-state loads, one exposed HBM round trip, a
// write-through publish + drain + ticket -- 32.8 us for 69 MB of K/V (2.1 TB/s, profiles/r01_batch64_kernel_stats.csv).
// Here a workgroup owns a PART of 192 tokens (more beyond 1536) of one (chain, kv head): six 32-token rounds whose K / V
// tiles arrive by `global_load_lds_dwordx4` into a three-stage ring (no VGPR staging; two rounds are in flight while the
// current one is on the matrix cores, one barrier per round), so the launch is a few hundred long-lived workgroups,
// three per CU (48 KB of LDS each: 64 chains at ~1100 tokens are 640 workgroups, all resident at once), every CU keeping
// up to 96 KB of K/V in flight.  (Measured at 64 chains, contexts 804..1436: 34.8 us for the slice kernel, 29.9 us for a
// first form of this one with 64-token rounds in a two-stage ring -- two workgroups per CU, 1.25 rounds of residency.)  The
// arithmetic of a round is that of attn_split_body (S^T = K Q^T with the q heads of the kv head as MFMA columns, online
// softmax lane-locally, O^T = V^T P^T through ds_read_b64_tr_b16); parts are merged in part order by the last-arriving
// workgroup of the (chain, kv head) with the fence-free sc1 hand-off of the single-chain kernel.
// The part geometry is a function of the chain's own context length alone, so a chain's result does not depend on
// which chains share the launch.
#include "ze_kernels.h"
#include "ze_attn_decode.h"

#define AB_TOK 32                        // keys per round
#define AB_STAGES 3
#define AB_STAGE (2 * AB_TOK * 256)      // K image + V image of one round, bytes (16 KB)

// This wave's 4 of the 16 1-KiB pieces (8 K + 8 V) of the round starting at token t0r: piece p covers rows 4p .. 4p+3,
// lane l lands at LDS position l & 15 of row 4p + (l >> 4), so the ad_off swizzle goes on the SOURCE chunk index.
// Rows past t1 re-read row t1 - 1 (finite values; their keys are masked out of the softmax).
__device__ __forceinline__ void ab_issue(const bf16_t* __restrict__ kb, const bf16_t* __restrict__ vb, int t0r, int t1,
                                         unsigned lds_stage, int wid, int lane) {
    const bf16_t* base = wid < 2 ? kb : vb;
    const unsigned img = lds_stage + (wid < 2 ? 0u : (unsigned)(AB_TOK * 256));
    const int r4 = lane >> 4, pos = lane & 15;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int p = (wid & 1) * 4 + g;
        const int row = 4 * p + r4;
        const int ch = pos ^ ((r4 << 2) | (p & 3));
        const bf16_t* src = base + (size_t)min(t0r + row, t1 - 1) * 128 + ch * 8;
        unsigned keep;
        asm volatile(
            "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(src), "s"(img + (unsigned)p * 1024u)
            : "memory");
    }
}

template <int MB>
__global__ void __launch_bounds__(256) k_attn_decode_stream(const bf16_t* __restrict__ q, int q_row_stride,
                                                            const bf16_t* __restrict__ kcache,
                                                            const bf16_t* __restrict__ vcache, size_t cache_seq_stride,
                                                            const ze_seq_dev* __restrict__ st_base,
                                                            const int* __restrict__ seq_ids, int heads, int kv_heads,
                                                            int max_ctx, float scale_log2e, float* __restrict__ ws,
                                                            int max_parts, unsigned* __restrict__ tickets,
                                                            bf16_t* __restrict__ out, int out_row_stride, int chunk_arg) {
    constexpr int D = 128;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];  // AB_STAGES stages of (K image | V image)
    const int bz = blockIdx.y;
    const int kvh = blockIdx.x % kv_heads, part = blockIdx.x / kv_heads;
    const int seq = seq_ids[bz];
    const int ctx = st_base[seq].ctx + 1;
    // Tokens per part: 192 (six 32-token rounds), or an eighth of the context once that is longer -- at most 8 parts per
    // (chain, kv head).  Measured at 64 chains of 804..1436 tokens: 24.7 us (fixed 128: 33.2 -- 1152 workgroups are 1.5
    // rounds of the 768 resident slots; 256 / 288: 29.6 / 31.0 -- 576-640 workgroups leave CUs with two or three of them;
    // 384: 22.3 but 16.1 instead of 12.0 us at 8 chains; parts proportional to each chain's length: 27.8 -- the longest
    // chain's parts set the time).  chunk_arg > 0 (measurements): a fixed size.  A function of the chain's own length
    // alone, so its result does not depend on the batch.
    const int chunk = chunk_arg > 0 ? chunk_arg : max(192, ((ctx + 7) / 8 + 31) / 32 * 32);
    const int nparts = (ctx + chunk - 1) / chunk;
    if (part >= nparts) return;  // workgroup-uniform: no part, no ticket
    const int G = heads / kv_heads;
    const int t0 = part * chunk, t1 = min(ctx, t0 + chunk);
    const int nr = (t1 - t0 + AB_TOK - 1) / AB_TOK;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const bf16_t* kb = kcache + (size_t)seq * cache_seq_stride + (size_t)kvh * max_ctx * D;
    const bf16_t* vb = vcache + (size_t)seq * cache_seq_stride + (size_t)kvh * max_ctx * D;
    const unsigned smem_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) uint8_t*)smem);

    // Q^T fragments first (lane: head column fr, d = ks*32 + fq*8 .. +7; columns >= G are zero queries), then the DMAs
    // of the first two rounds right behind them
    const bf16_t* qrow = q + (size_t)bz * q_row_stride;
    ad_bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        uint4 u = *reinterpret_cast<const uint4*>(qrow + (kvh * G + min(fr, G - 1)) * D + (ks * 4 + fq) * 8);
        if (fr >= G) u = make_uint4(0, 0, 0, 0);
        qf[ks] = *reinterpret_cast<const ad_bf16x8*>(&u);
    }
    ab_issue(kb, vb, t0, t1, smem_lds, wid, lane);
    if (nr > 1) ab_issue(kb, vb, t0 + AB_TOK, t1, smem_lds + AB_STAGE, wid, lane);

    ad_f32x4 oacc[2];
    oacc[0] = oacc[1] = ad_f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    for (int r = 0; r < nr; ++r) {
        // this wave's pieces of round r have landed; the round in flight behind it (4 DMAs) may stay outstanding
        // (vmcnt retires in issue order: Q, round 0, round 1, ...)
        if (r + 1 < nr) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // everybody's pieces of round r have, and everybody is done reading round r - 1 ...
        asm volatile("" ::: "memory");
        // ... whose stage therefore takes round r + 2 now: ONE barrier per round, two rounds in flight during the math
        if (r + 2 < nr) ab_issue(kb, vb, t0 + (r + 2) * AB_TOK, t1, smem_lds + ((r + 2) % AB_STAGES) * AB_STAGE, wid, lane);
        const uint8_t* sK = smem + (r % AB_STAGES) * AB_STAGE;
        const uint8_t* sV = sK + AB_TOK * 256;
        const int base = t0 + r * AB_TOK;
        // S^T = K Q^T: 2 key tiles x 4 steps of 32 d (every wave forms the whole S^T: softmax statistics stay lane-local)
        ad_f32x4 sacc[2];
        sacc[0] = sacc[1] = ad_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const uint4 ka = *reinterpret_cast<const uint4*>(sK + ad_off(n * 16 + fr, ks * 4 + fq));
                sacc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const ad_bf16x8*>(&ka), qf[ks], sacc[n], 0, 0, 0);
            }
        float p[2][4];
        float mx = -INFINITY;
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const bool ok = base + n * 16 + fq * 4 + rr < t1;
                const float sv = ok ? sacc[n][rr] * scale_log2e : -INFINITY;
                p[n][rr] = sv;
                mx = fmaxf(mx, sv);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = exp2f(m_run - m_use);  // m_run = -inf -> 0
        float rs = 0.f;
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const float ev = __builtin_amdgcn_exp2f(p[n][rr] - m_use);  // arguments <= 0
                p[n][rr] = ev;
                rs += ev;
            }
        rs += __shfl_xor(rs, 16, 64);
        rs += __shfl_xor(rs, 32, 64);
        l_run = l_run * alpha + rs;
        m_run = m_new;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            oacc[jj][0] *= alpha;
            oacc[jj][1] *= alpha;
            oacc[jj][2] *= alpha;
            oacc[jj][3] *= alpha;
        }
        // O^T += V^T P^T for this wave's d-tiles 2*wid, 2*wid + 1 (P rounded to bf16, as HF's eager attention rounds it):
        // the lane's 8 keys of the 32-key step are (fq*4 + 0..3) and (16 + fq*4 + 0..3), the same permutation on both operands
        {
            const uint4 pq = make_uint4(ad_pack_bf16(p[0][0], p[0][1]), ad_pack_bf16(p[0][2], p[0][3]),
                                        ad_pack_bf16(p[1][0], p[1][1]), ad_pack_bf16(p[1][2], p[1][3]));
            const ad_bf16x8 pb = *reinterpret_cast<const ad_bf16x8*>(&pq);
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int j = 2 * wid + jj;
                const int tq = fr >> 2, tp = fr & 3;
                const int r0a = fq * 4, r0b = r0a + 16;
                const int ch = j * 2 + (tp >> 1), half = 8 * (tp & 1);
                const ad_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) ad_v4s*)(sV + ad_off(r0a + tq, ch) + half));
                const ad_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) ad_v4s*)(sV + ad_off(r0b + tq, ch) + half));
                const ad_v8s va = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                oacc[jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const ad_bf16x8*>(&va), pb, oacc[jj], 0, 0, 0);
            }
        }
        asm volatile("" ::: "memory");
    }
    __syncthreads();  // the tail reuses the staging LDS

    // partial of head fr for this part: (m, l) and O[d = (2*wid + jj)*16 + fq*4 .. +3], published write-through
    float* wsb = ws + (size_t)bz * max_parts * heads * AD_STRIDE;
    if (fr < G) {
        const uint32_t dst = (uint32_t)((part * heads + kvh * G + fr) * AD_STRIDE * 4);
        if (wid == 0 && fq == 0) ad_store16<true>(wsb, dst, m_run, l_run, 0.f, 0.f);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
            ad_store16<true>(wsb, dst + (uint32_t)((4 + (2 * wid + jj) * 16 + fq * 4) * 4), oacc[jj][0], oacc[jj][1],
                             oacc[jj][2], oacc[jj][3]);
    }
    // merge by the last-arriving part of this (chain, kv head): every storing wave drains, ONE lane takes a ticket
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned* flag = reinterpret_cast<unsigned*>(smem + AB_STAGE);  // the stages are dead now
    if (threadIdx.x == 0) {
        unsigned* t = tickets + (size_t)bz * kv_heads + kvh;
        const unsigned old = __hip_atomic_fetch_add(t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned last = old == (unsigned)nparts - 1u;
        if (last) __hip_atomic_store(t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = last;
    }
    __syncthreads();
    if (*flag == 0u) return;
    __syncthreads();
    float* sW = reinterpret_cast<float*>(smem);
    if (out_row_stride < 0)
        attn_merge_group<MB>(sW, sW + AD_GMAX * 64, wsb, ctx, kvh, heads, kv_heads, max_parts, out, -out_row_stride, bz, nparts);
    else
        attn_merge_group<MB>(sW, sW + AD_GMAX * 64, wsb, ctx, kvh, heads, kv_heads, max_parts,
                             out + (size_t)bz * out_row_stride, 0, 0, nparts);
}

void ze_launch_attn_decode_stream(const bf16_t* q, int q_row_stride, const bf16_t* kcache, const bf16_t* vcache,
                                  size_t cache_seq_stride, bf16_t* out, int out_row_stride, const ze_seq_dev* st,
                                  const int* seq_ids, int n, int heads, int kv_heads, int max_ctx, float scale,
                                  float* ws_partial, int max_parts, unsigned* tickets, hipStream_t s, int chunk) {
    const float sl = scale * 1.4426950408889634f;
    const size_t lds = AB_STAGES * AB_STAGE;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attn_decode_stream<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    k_attn_decode_stream<8><<<dim3(kv_heads * max_parts, n), 256, lds, s>>>(q, q_row_stride, kcache, vcache, cache_seq_stride, st,
                                                                          seq_ids, heads, kv_heads, max_ctx, sl, ws_partial,
                                                                          max_parts, tickets, out, out_row_stride, chunk);
}
