// Greedy sampling: repetition penalty over the ids already in the sequence
// (HF:generation/logits_process.py:409-413: score<0 ? score*p : score/p), argmax with lowest-index tie-break
// (torch.argmax), EOS / pad bookkeeping (HF:generation/utils.py:2921-2936), and the chain-state update that
// lets the next decode step run without a host round trip.
#include "ze_kernels.h"

#define SAMPLE_BLOCKS 128

__device__ __forceinline__ void better(float& bv, int& bi, float v, int i) {
    if (v > bv || (v == bv && i < bi)) {
        bv = v;
        bi = i;
    }
}

__global__ void __launch_bounds__(256) k_argmax_partial(const float* __restrict__ logits, int vocab,
                                                        const uint8_t* __restrict__ seen, float penalty,
                                                        float* __restrict__ ws) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < vocab; i += SAMPLE_BLOCKS * 256) {
        float v = logits[i];
        if (penalty != 1.0f && seen[i]) v = v < 0.f ? v * penalty : v / penalty;
        better(bv, bi, v, i);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(bv, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        better(bv, bi, ov, oi);
    }
    __shared__ float sv[4];
    __shared__ int si[4];
    if ((threadIdx.x & 63) == 0) {
        sv[threadIdx.x >> 6] = bv;
        si[threadIdx.x >> 6] = bi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) better(bv, bi, sv[w], si[w]);
        ws[2 * blockIdx.x] = bv;
        reinterpret_cast<int*>(ws)[2 * blockIdx.x + 1] = bi;
    }
}

__global__ void __launch_bounds__(64) k_argmax_final(const float* __restrict__ ws, uint8_t* __restrict__ seen,
                                                     ze_seq_dev* __restrict__ st, const int* __restrict__ eos_ids,
                                                     int n_eos, int pad_id, int ignore_eos, int advance_ctx,
                                                     int32_t* __restrict__ out_tokens) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < SAMPLE_BLOCKS; i += 64) better(bv, bi, ws[2 * i], reinterpret_cast<const int*>(ws)[2 * i + 1]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(bv, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        better(bv, bi, ov, oi);
    }
    if (threadIdx.x == 0) {
        int tok = bi;
        if (st->finished) tok = pad_id;  // finished rows emit pad (HF:generation/utils.py:2927-2929)
        if (advance_ctx) st->ctx += 1;
        if (st->n_gen < st->max_gen) out_tokens[st->n_gen] = tok;
        st->n_gen += 1;
        st->token = tok;
        seen[tok] = 1;
        if (!ignore_eos && !st->finished) {
            for (int e = 0; e < n_eos; ++e)
                if (tok == eos_ids[e]) st->finished = 1;
        }
    }
}

void ze_launch_sample(const float* logits, int vocab, uint8_t* seen, float penalty, ze_seq_dev* st,
                      const int* eos_ids, int n_eos, int pad_id, int ignore_eos, int advance_ctx,
                      int32_t* out_tokens, float* ws, hipStream_t s) {
    k_argmax_partial<<<SAMPLE_BLOCKS, 256, 0, s>>>(logits, vocab, seen, penalty, ws);
    k_argmax_final<<<1, 64, 0, s>>>(ws, seen, st, eos_ids, n_eos, pad_id, ignore_eos, advance_ctx, out_tokens);
}

__global__ void k_advance_ctx(ze_seq_dev* st) { st->ctx += 1; }
void ze_launch_advance_ctx(ze_seq_dev* st, hipStream_t s) { k_advance_ctx<<<1, 1, 0, s>>>(st); }

__global__ void k_mark_seen(uint8_t* seen, const int* ids, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) seen[ids[i]] = 1;
}
void ze_launch_mark_seen(uint8_t* seen, const int* ids, int n, hipStream_t s) {
    if (n > 0) k_mark_seen<<<ze_cdiv(n, 256), 256, 0, s>>>(seen, ids, n);
}

// ------------------------------------------------------------------ batched sampling: grid.y = chain
__global__ void __launch_bounds__(256) k_argmax_partial_batch(const float* __restrict__ logits, int vocab,
                                                              const uint8_t* __restrict__ seen_base,
                                                              const int* __restrict__ seq_ids, float penalty,
                                                              float* __restrict__ ws) {
    const int b = blockIdx.y;
    const float* lg = logits + (size_t)b * vocab;
    const uint8_t* seen = seen_base + (size_t)seq_ids[b] * vocab;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < vocab; i += SAMPLE_BLOCKS * 256) {
        float v = lg[i];
        if (penalty != 1.0f && seen[i]) v = v < 0.f ? v * penalty : v / penalty;
        better(bv, bi, v, i);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(bv, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        better(bv, bi, ov, oi);
    }
    __shared__ float sv[4];
    __shared__ int si[4];
    if ((threadIdx.x & 63) == 0) {
        sv[threadIdx.x >> 6] = bv;
        si[threadIdx.x >> 6] = bi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) better(bv, bi, sv[w], si[w]);
        float* o = ws + (size_t)b * 2 * SAMPLE_BLOCKS;
        o[2 * blockIdx.x] = bv;
        reinterpret_cast<int*>(o)[2 * blockIdx.x + 1] = bi;
    }
}

__global__ void __launch_bounds__(64) k_argmax_final_batch(const float* __restrict__ ws, uint8_t* __restrict__ seen_base,
                                                           ze_seq_dev* __restrict__ st_base,
                                                           const int* __restrict__ seq_ids, int vocab,
                                                           const int* __restrict__ eos_ids, int n_eos, int pad_id,
                                                           int ignore_eos, int advance_ctx, int sample,
                                                           int32_t* __restrict__ out_base, int max_gen) {
    const int b = blockIdx.x, seq = seq_ids[b];
    ze_seq_dev* st = st_base + seq;
    if (!sample) {  // teacher forcing: only the cache grows
        if (threadIdx.x == 0 && advance_ctx) st->ctx += 1;
        return;
    }
    const float* w = ws + (size_t)b * 2 * SAMPLE_BLOCKS;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < SAMPLE_BLOCKS; i += 64) better(bv, bi, w[2 * i], reinterpret_cast<const int*>(w)[2 * i + 1]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(bv, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        better(bv, bi, ov, oi);
    }
    if (threadIdx.x == 0) {
        int tok = bi;
        if (st->finished) tok = pad_id;
        if (advance_ctx) st->ctx += 1;
        if (st->n_gen < st->max_gen) out_base[(size_t)seq * max_gen + st->n_gen] = tok;
        st->n_gen += 1;
        st->token = tok;
        seen_base[(size_t)seq * vocab + tok] = 1;
        if (!ignore_eos && !st->finished) {
            for (int e = 0; e < n_eos; ++e)
                if (tok == eos_ids[e]) st->finished = 1;
        }
    }
}

void ze_launch_sample_batch(const float* logits, int vocab, uint8_t* seen_base, float penalty, ze_seq_dev* st,
                            const int* seq_ids, int n, const int* eos_ids, int n_eos, int pad_id, int ignore_eos,
                            int advance_ctx, int sample, int32_t* out_tokens_base, int max_gen, float* ws, hipStream_t s) {
    if (n <= 0) return;
    if (sample)
        k_argmax_partial_batch<<<dim3(SAMPLE_BLOCKS, n), 256, 0, s>>>(logits, vocab, seen_base, seq_ids, penalty, ws);
    k_argmax_final_batch<<<n, 64, 0, s>>>(ws, seen_base, st, seq_ids, vocab, eos_ids, n_eos, pad_id, ignore_eos,
                                          advance_ctx, sample, out_tokens_base, max_gen);
}
