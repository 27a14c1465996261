// One-token (decode) GQA attention over the KV cache: flash-decoding slices merged in the same launch.
// Own translation unit: it is compiled WITH hipcc's SLP vectoriser (11.5 vs 11.95 us per launch), which the rest of
// the library is built without (see Makefile); every result of this kernel is pinned bit for bit against the fused
// variant and within tolerance against the prefill attention and the oracle (tests/test_gpu_fused.py,
// test_gpu_longctx.py, test_gpu_model.py).
#include "ze_kernels.h"

// ------------------------------------------------------------------ decode attention (D = 128)
// Flash-decoding over the KV cache of one chain: grid = (kv_heads, max_splits, chains); a block owns one 64-token
// slice of the context for ALL q heads of its kv head (K/V are read once per kv head).  Per 64-token round every
// thread issues its 4 K + 4 V 16-B loads up front (the kernel is latency-, not bandwidth-bound at one chain: 1-2 MB
// per layer), stages them into two LDS images and the slice runs on the matrix cores (attn_split_body: S^T = K Q^T,
// online softmax per head lane-locally, O^T = V^T P^T).  Partials (m, l, o[128]) per (split, head) go to a
// workspace and are merged by the last-arriving slice block of the (chain, kv head) in the same launch.
// (History: the slice as fp32 FMAs on the vector ALUs with a halving-butterfly reduction cost 11.5 us per launch
//  against 8.1 now; a separate merge launch before that 7.4 + 4.8 us; the in-launch merge with an agent-scope
//  acquire fence + plain loads instead of the sc1 pair cost 30 ms per question MORE than the separate launch.)
#include "ze_attn_decode.h"

// (Measured, rejected: the merge as its own launch for the batched step -- slices retire without the drain + ticket tail --
//  4.32 -> 4.28 ms per 64-chain step, 3.16 -> 3.13 at 8: not worth a second kernel.)
// MB: slices whose partials the merging workgroup requests up front (24 covers 1536 tokens in one round trip: the
// single-chain step, where the launch is latency-bound; 8 keeps the kernel at 4 workgroups per CU for the batched
// step, where it is throughput-bound).  The merge adds the slices in the same order either way.
// The batched instantiation (MB = 8) is compiled for five waves per SIMD (87 VGPRs, no spills; the LDS images allow five
// workgroups per CU): occupancy is what this throughput-bound form lives on (three instead of four workgroups per CU
// cost 15 %).
// NS: slices per workgroup (attn_split_body): 1 for the single-chain step (most workgroups, shortest chain), 2 for
// the batched step (grid.y = max_splits / 2).
template <int MB, int NS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MB == 8 ? 5 : 1))) k_attn_decode_split(const bf16_t* __restrict__ q, int q_row_stride,
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
    if ((int)blockIdx.y * NS >= nsplit) return;  // workgroup-uniform: no slice, no ticket
    float* wsb = ws + (size_t)bz * max_splits * heads * AD_STRIDE;
    attn_split_body<false, true, NS>(L, q + (size_t)bz * q_row_stride, kcache, vcache, ctx, kvh, blockIdx.y, heads, kv_heads,
                                 max_ctx, scale_log2e, wsb, max_splits);
    // ---- merge by the last-arriving slice of this (chain, kv head).  Hand-off without fences: the partials above
    // went out write-through (sc1), every storing wave drains, the workgroup barriers, ONE lane takes a ticket
    // (relaxed agent-scope add on one unsharded counter); the workgroup whose add came last reads every partial
    // with sc1 loads, in slice order, so the result does not depend on who is last.  The ticket returns to 0 for
    // the next launch.  (With an agent-scope acquire fence + plain loads in place of the sc1 pair this merge cost
    // 30 ms per question MORE than a separate merge launch; in this form it replaces that 4.8-us launch.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned* flag = reinterpret_cast<unsigned*>(&L.sV[0]);  // the slice LDS is dead now
    if (threadIdx.x == 0) {
        unsigned* t = tickets + (size_t)bz * kv_heads + kvh;
        const unsigned old = __hip_atomic_fetch_add(t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned last = old == (unsigned)((nsplit + NS - 1) / NS) - 1u;  // one ticket per workgroup with slices
        if (last) __hip_atomic_store(t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = last;
    }
    __syncthreads();
    if (*flag == 0u) return;
    __syncthreads();  // everybody has read the flag before the merge reuses the LDS
    float* sW = reinterpret_cast<float*>(&L.sK[0]);
    // out_row_stride < 0: fragment-major output with -out_row_stride slices per row tile, chain bz = row bz
    if (out_row_stride < 0)
        attn_merge_group<MB>(sW, sW + AD_GMAX * 64, wsb, ctx, kvh, heads, kv_heads, max_splits, out, -out_row_stride, bz);
    else
        attn_merge_group<MB>(sW, sW + AD_GMAX * 64, wsb, ctx, kvh, heads, kv_heads, max_splits,
                             out + (size_t)bz * out_row_stride);
}

void ze_launch_attn_decode(const bf16_t* q, int q_row_stride, const bf16_t* kcache, const bf16_t* vcache,
                           size_t cache_seq_stride, bf16_t* out, int out_row_stride, const ze_seq_dev* st,
                           const int* seq_ids, int n, int heads, int kv_heads, int D, int max_ctx, float scale,
                           float* ws_partial, int max_splits, unsigned* tickets, hipStream_t s) {
    (void)D;
    const float sl = scale * 1.4426950408889634f;
    // (rope + KV append fused into this launch -- q fragments rotated in registers, the last slice's workgroup rotating and
    //  appending the new K / V row -- was bit-identical and SLOWER: 132 VGPRs instead of 120 drop the kernel from four to
    //  three workgroups per CU, 4.42 against 3.84 ms per 64-chain step, more than the removed 4.9-us launch)
    // (two slices per workgroup with both slices' K/V requested up front -- NS = 2 -- win at contexts of about 850
    //  tokens, 3.93 -> 3.86 ms per 64-chain step, and lose at the benchmark's 800-1400: 29.6 against 30.3 questions/s)
    if (seq_ids)
        k_attn_decode_split<8, 1><<<dim3(kv_heads, max_splits, n), 256, 0, s>>>(q, q_row_stride, kcache, vcache,
                                                                                 cache_seq_stride, st, seq_ids, heads,
                                                                                 kv_heads, max_ctx, sl, ws_partial, max_splits,
                                                                                 tickets, out, out_row_stride);
    else
        k_attn_decode_split<24, 1><<<dim3(kv_heads, max_splits, n), 256, 0, s>>>(q, q_row_stride, kcache, vcache,
                                                                               cache_seq_stride, st, seq_ids, heads,
                                                                               kv_heads, max_ctx, sl, ws_partial, max_splits,
                                                                               tickets, out, out_row_stride);
}
