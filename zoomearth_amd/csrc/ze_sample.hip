// Greedy sampling: repetition penalty over the ids already in the sequence
// (HF:generation/logits_process.py:409-413: score<0 ? score*p : score/p), argmax with lowest-index tie-break
// (torch.argmax), EOS / pad bookkeeping (HF:generation/utils.py:2921-2936), and the chain-state update that
// lets the next decode step run without a host round trip.
#include "ze_kernels.h"
#include "ze_prng.h"

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

// the chosen token enters the chain state (one thread)
__device__ __forceinline__ void accept_token(int bi, uint8_t* __restrict__ seen, ze_seq_dev* __restrict__ st,
                                             const int* __restrict__ eos_ids, int n_eos, int pad_id, int ignore_eos,
                                             int advance_ctx, int32_t* __restrict__ out_tokens, int vocab);

__global__ void __launch_bounds__(64) k_argmax_final(const float* __restrict__ ws, uint8_t* __restrict__ seen,
                                                     ze_seq_dev* __restrict__ st, const int* __restrict__ eos_ids,
                                                     int n_eos, int pad_id, int ignore_eos, int advance_ctx,
                                                     int32_t* __restrict__ out_tokens, int vocab) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < SAMPLE_BLOCKS; i += 64) better(bv, bi, ws[2 * i], reinterpret_cast<const int*>(ws)[2 * i + 1]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(bv, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        better(bv, bi, ov, oi);
    }
    if (threadIdx.x == 0) accept_token(bi, seen, st, eos_ids, n_eos, pad_id, ignore_eos, advance_ctx, out_tokens, vocab);
}

// The partials of the lm_head GEMV's workgroups (ze_gemv_args::amax_ws: AMAX_SLOTS (value, index) pairs; slots no
// workgroup writes keep the (-inf, INT_MAX) they were created with): every thread requests its eight pairs at once --
// one memory round trip for the whole reduction.
#define AMAX_SLOTS 2048
__global__ void __launch_bounds__(256) k_argmax_final_folded(const float2* __restrict__ ws, uint8_t* __restrict__ seen,
                                                            ze_seq_dev* __restrict__ st, const int* __restrict__ eos_ids,
                                                            int n_eos, int pad_id, int ignore_eos, int advance_ctx,
                                                            int32_t* __restrict__ out_tokens, int vocab) {
    float2 p[AMAX_SLOTS / 256];
#pragma unroll
    for (int u = 0; u < AMAX_SLOTS / 256; ++u) p[u] = ws[threadIdx.x + u * 256];
    float bv = -INFINITY;
    int bi = 0x7fffffff;
#pragma unroll
    for (int u = 0; u < AMAX_SLOTS / 256; ++u) better(bv, bi, p[u].x, __float_as_int(p[u].y));
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
        accept_token(bi, seen, st, eos_ids, n_eos, pad_id, ignore_eos, advance_ctx, out_tokens, vocab);
    }
}

__device__ __forceinline__ void accept_token(int bi, uint8_t* __restrict__ seen, ze_seq_dev* __restrict__ st,
                                             const int* __restrict__ eos_ids, int n_eos, int pad_id, int ignore_eos,
                                             int advance_ctx, int32_t* __restrict__ out_tokens, int vocab) {
    {
        int tok = bi;
        if ((unsigned)tok >= (unsigned)vocab) tok = 0;  // no comparable logit at all (every one NaN): torch.argmax gives 0
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

void ze_launch_multinomial(const float* logits, int vocab, const uint8_t* seen_base, const ze_seq_dev* st,
                           const int* seq_ids, int slot0, int n, float penalty, float temperature,
                           unsigned long long seed, float* ws_part, float* ws_sum, hipStream_t s);

void ze_launch_sample(const float* logits, int vocab, uint8_t* seen, float penalty, ze_seq_dev* st,
                      const int* eos_ids, int n_eos, int pad_id, int ignore_eos, int advance_ctx,
                      int32_t* out_tokens, float* ws, const ze_sample_opts& so, hipStream_t s) {
    k_argmax_partial<<<SAMPLE_BLOCKS, 256, 0, s>>>(logits, vocab, seen, penalty, ws);
    if (so.temperature > 0.f)
        ze_launch_multinomial(logits, vocab, seen, st, nullptr, so.slot, 1, penalty, so.temperature, so.seed, ws,
                              ws + 2 * SAMPLE_BLOCKS + 64, s);
    k_argmax_final<<<1, 64, 0, s>>>(ws, seen, st, eos_ids, n_eos, pad_id, ignore_eos, advance_ctx, out_tokens, vocab);
}

// greedy token when the lm_head GEMV already left its workgroups' partial maxima in amax_ws
void ze_launch_sample_folded(const float* amax_ws, int vocab, uint8_t* seen, ze_seq_dev* st, const int* eos_ids, int n_eos,
                             int pad_id, int ignore_eos, int advance_ctx, int32_t* out_tokens, hipStream_t s) {
    k_argmax_final_folded<<<1, 256, 0, s>>>(reinterpret_cast<const float2*>(amax_ws), seen, st, eos_ids, n_eos, pad_id,
                                            ignore_eos, advance_ctx, out_tokens, vocab);
}

// amax_ws as the engine creates it: every slot (-inf, INT_MAX)
__global__ void k_amax_init(float2* ws) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < AMAX_SLOTS) ws[i] = make_float2(-INFINITY, __int_as_float(0x7fffffff));
}
void ze_launch_amax_init(float* amax_ws, hipStream_t s) { k_amax_init<<<AMAX_SLOTS / 256, 256, 0, s>>>(reinterpret_cast<float2*>(amax_ws)); }

__global__ void k_advance_ctx(ze_seq_dev* st) { st->ctx += 1; }
void ze_launch_advance_ctx(ze_seq_dev* st, hipStream_t s) { k_advance_ctx<<<1, 1, 0, s>>>(st); }

__global__ void k_mark_seen(uint8_t* seen, const int* ids, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) seen[ids[i]] = 1;
}
void ze_launch_mark_seen(uint8_t* seen, const int* ids, int n, hipStream_t s) {
    if (n > 0) k_mark_seen<<<ze_cdiv(n, 256), 256, 0, s>>>(seen, ids, n);
}

// the prompts of several chains in one launch: hdr = [offs (n + 1) | chain slots (n)], ids follow; grid.y = chain
__global__ void k_mark_seen_batch(uint8_t* __restrict__ seen, size_t vocab, const int* __restrict__ hdr, const int* __restrict__ ids, int n) {
    const int c = blockIdx.y, o0 = hdr[c], cnt = hdr[c + 1] - o0;
    uint8_t* sn = seen + (size_t)hdr[n + 1 + c] * vocab;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < cnt; i += gridDim.x * 256) sn[ids[o0 + i]] = 1;
}
void ze_launch_mark_seen_batch(uint8_t* seen, int vocab, const int* hdr, const int* ids, int n, int max_count, hipStream_t s) {
    if (n > 0 && max_count > 0) k_mark_seen_batch<<<dim3(std::min(ze_cdiv(max_count, 256), 16), n), 256, 0, s>>>(seen, (size_t)vocab, hdr, ids, n);
}

// the generated ids of several chains, gathered for ONE device -> host copy: out = [n_gen, finished per chain (2n) | n rows of cap ids]
__global__ void k_gather_chain_tokens(const ze_seq_dev* __restrict__ st, const int* __restrict__ out_tokens, int max_ctx,
                                      const int* __restrict__ slots, int n, int cap, int* __restrict__ out) {
    const int c = blockIdx.y, seq = slots[c];
    const int ng = min(min(st[seq].n_gen, cap), max_ctx);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        out[2 * c] = ng;
        out[2 * c + 1] = st[seq].finished;
    }
    const int* src = out_tokens + (size_t)seq * max_ctx;
    int* dst = out + 2 * n + (size_t)c * cap;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < ng; i += gridDim.x * 256) dst[i] = src[i];
}
void ze_launch_gather_chain_tokens(const ze_seq_dev* st, const int* out_tokens, int max_ctx, const int* slots, int n, int cap, int* out,
                                   hipStream_t s) {
    if (n > 0) k_gather_chain_tokens<<<dim3(std::max(1, std::min(ze_cdiv(cap, 256), 8)), n), 256, 0, s>>>(st, out_tokens, max_ctx, slots, n, cap, out);
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
        if ((unsigned)tok >= (unsigned)vocab) tok = 0;  // every logit NaN: torch.argmax gives 0
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
                            int advance_ctx, int sample, int32_t* out_tokens_base, int max_gen, float* ws,
                            float* ws_sum, const ze_sample_opts& so, hipStream_t s) {
    if (n <= 0) return;
    if (sample) {
        k_argmax_partial_batch<<<dim3(SAMPLE_BLOCKS, n), 256, 0, s>>>(logits, vocab, seen_base, seq_ids, penalty, ws);
        if (so.temperature > 0.f)
            ze_launch_multinomial(logits, vocab, seen_base, st, seq_ids, 0, n, penalty, so.temperature, so.seed, ws,
                                  ws_sum, s);
    }
    k_argmax_final_batch<<<n, 64, 0, s>>>(ws, seen_base, st, seq_ids, vocab, eos_ids, n_eos, pad_id, ignore_eos,
                                          advance_ctx, sample, out_tokens_base, max_gen);
}

// ------------------------------------------------------------------ temperature sampling (do_sample=True)
// replaces: TemperatureLogitsWarper + softmax + torch.multinomial(probs, 1) in GenerationMixin._sample
// (HF:generation/utils.py:2894-2916, HF:generation/logits_process.py:285-345) as src/eval/infer.py:109-115 calls it
// (temperature 0.01, top_k = top_p = None).  Same distribution; the random stream is this repo's counter-based
// generator, not torch's, so a draw is reproducible from (seed, row of the chain in the generate call, index of the
// generated token) whatever chain slot the request landed in:
//     u = (stream64(mix64(seed ^ mix64(row + 1)), n_gen) >> 40) * 2^-24                         in [0, 1)
//     e_i = expf(score_i / T - max_j score_j / T)        score = repetition-penalised fp32 logit
//     token = first i (ascending) whose running sum of e exceeds u * sum(e)
// Summation order (fp32), which the oracle (oracle/qwen25vl.py:sample_temperature) restates: the vocabulary is cut
// into SAMPLE_BLOCKS contiguous chunks, a chunk into 256 contiguous runs; run sums, then the 256 run sums in order,
// then the chunk sums in order.  Two extra launches per token after the arg-max partials; the pick is handed to
// the arg-max final kernels as an unbeatable partial (+inf, token), so EOS / pad / state bookkeeping is shared.
__device__ __forceinline__ float sample_score(const float* lg, const uint8_t* seen, float penalty, int i) {
    float v = lg[i];
    if (penalty != 1.0f && seen[i]) v = v < 0.f ? v * penalty : v / penalty;
    return v;
}

__device__ __forceinline__ float sample_zmax(const float* part, float temperature) {
    float bv = -INFINITY;
    for (int i = threadIdx.x & 63; i < SAMPLE_BLOCKS; i += 64) bv = fmaxf(bv, part[2 * i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) bv = fmaxf(bv, __shfl_xor(bv, off, 64));
    return bv / temperature;
}

// sum of e over this thread's contiguous run, then the 256 run sums in thread order (by thread 0) -> *total;
// sRun[t] keeps the run sums for the caller
__device__ __forceinline__ void sample_chunk_sums(const float* lg, const uint8_t* seen, float penalty, float temperature,
                                                  float zmax, int start, int end, int run, float* sRun, float* total) {
    const int t = threadIdx.x;
    float acc = 0.f;
    for (int j = 0; j < run; ++j) {
        const int i = start + t * run + j;
        if (i < end) acc += expf(sample_score(lg, seen, penalty, i) / temperature - zmax);
    }
    sRun[t] = acc;
    __syncthreads();
    if (t == 0) {
        float tot = 0.f;
        for (int k = 0; k < 256; ++k) tot += sRun[k];
        *total = tot;
    }
    __syncthreads();
}

__global__ void __launch_bounds__(256) k_softmax_partial(const float* __restrict__ logits, int vocab,
                                                         const uint8_t* __restrict__ seen_base,
                                                         const int* __restrict__ seq_ids, float penalty,
                                                         float temperature, const float* __restrict__ ws_part,
                                                         float* __restrict__ ws_sum) {
    const int b = blockIdx.y;
    const float* lg = logits + (size_t)b * vocab;
    const uint8_t* seen = seen_base + (seq_ids ? (size_t)seq_ids[b] * vocab : 0);
    const float zmax = sample_zmax(ws_part + (size_t)b * 2 * SAMPLE_BLOCKS, temperature);
    const int chunk = (vocab + SAMPLE_BLOCKS - 1) / SAMPLE_BLOCKS, run = (chunk + 255) / 256;
    const int start = blockIdx.x * chunk, end = min(vocab, start + chunk);
    __shared__ float sRun[256];
    __shared__ float sTot;
    sample_chunk_sums(lg, seen, penalty, temperature, zmax, start, end, run, sRun, &sTot);
    if (threadIdx.x == 0) ws_sum[(size_t)b * SAMPLE_BLOCKS + blockIdx.x] = sTot;
}

__global__ void __launch_bounds__(256) k_multinomial_pick(const float* __restrict__ logits, int vocab,
                                                          const uint8_t* __restrict__ seen_base,
                                                          const ze_seq_dev* __restrict__ st_base,
                                                          const int* __restrict__ seq_ids, int slot0, float penalty,
                                                          float temperature, unsigned long long seed,
                                                          float* __restrict__ ws_part,
                                                          const float* __restrict__ ws_sum) {
    const int b = blockIdx.x, slot = seq_ids ? seq_ids[b] : slot0;
    const ze_seq_dev* st = seq_ids ? st_base + slot : st_base;
    const float* lg = logits + (size_t)b * vocab;
    const uint8_t* seen = seen_base + (seq_ids ? (size_t)slot * vocab : 0);
    float* part = ws_part + (size_t)b * 2 * SAMPLE_BLOCKS;
    const float zmax = sample_zmax(part, temperature);
    const int chunk = (vocab + SAMPLE_BLOCKS - 1) / SAMPLE_BLOCKS, run = (chunk + 255) / 256;
    __shared__ float sRun[256];
    __shared__ float sTot;
    __shared__ int sBlk;
    __shared__ float sTarget;
    if (threadIdx.x == 0) {
        const float* sums = ws_sum + (size_t)b * SAMPLE_BLOCKS;
        float total = 0.f;
        for (int k = 0; k < SAMPLE_BLOCKS; ++k) total += sums[k];
        const unsigned long long key = ze_mix64(seed ^ ze_mix64((unsigned long long)st->stream + 1ull));
        const float u = (float)(ze_stream64(key, (unsigned long long)st->n_gen) >> 40) * 5.9604644775390625e-08f;
        const float target = u * total;
        float cum = 0.f;
        int blk = -1, last_nz = 0;
        float excl = 0.f;
        for (int k = 0; k < SAMPLE_BLOCKS; ++k) {
            if (sums[k] > 0.f) last_nz = k;
            const float nxt = cum + sums[k];
            if (blk < 0 && nxt > target) {
                blk = k;
                excl = cum;
            }
            cum = nxt;
        }
        if (blk < 0) {  // only by rounding: take the end of the last non-empty chunk
            blk = last_nz;
            excl = INFINITY;
        }
        sBlk = blk;
        sTarget = target - excl;  // -inf: pick the last element with mass
    }
    __syncthreads();
    const int blk = sBlk;
    const float target = sTarget;
    const int start = blk * chunk, end = min(vocab, start + chunk);
    sample_chunk_sums(lg, seen, penalty, temperature, zmax, start, end, run, sRun, &sTot);
    if (threadIdx.x == 0) {
        int tok = -1, last_nz = start;
        float cum = 0.f;
        for (int t = 0; t < 256 && tok < 0; ++t) {
            if (cum + sRun[t] > target) {  // the pick is inside run t: walk it
                for (int j = 0; j < run; ++j) {
                    const int i = start + t * run + j;
                    if (i >= end) break;
                    const float e = expf(sample_score(lg, seen, penalty, i) / temperature - zmax);
                    cum += e;
                    if (cum > target) {
                        tok = i;
                        break;
                    }
                }
                // (a run whose sum crossed the target but whose elements, re-added, did not: rounding -- go on)
            } else {
                cum += sRun[t];
            }
            if (sRun[t] > 0.f) last_nz = min(end - 1, start + t * run + run - 1);
        }
        if (tok < 0) {  // rounding at the very end (or target = -inf): last element of the chunk that carries mass
            tok = last_nz;
            while (tok > start && !(expf(sample_score(lg, seen, penalty, tok) / temperature - zmax) > 0.f)) --tok;
        }
        part[0] = INFINITY;  // unbeatable arg-max partial: the final kernel adopts the pick
        reinterpret_cast<int*>(part)[1] = tok;
    }
}

void ze_launch_multinomial(const float* logits, int vocab, const uint8_t* seen_base, const ze_seq_dev* st,
                           const int* seq_ids, int slot0, int n, float penalty, float temperature,
                           unsigned long long seed, float* ws_part, float* ws_sum, hipStream_t s) {
    if (n <= 0) return;
    k_softmax_partial<<<dim3(SAMPLE_BLOCKS, n), 256, 0, s>>>(logits, vocab, seen_base, seq_ids, penalty, temperature,
                                                             ws_part, ws_sum);
    k_multinomial_pick<<<n, 256, 0, s>>>(logits, vocab, seen_base, st, seq_ids, slot0, penalty, temperature, seed,
                                         ws_part, ws_sum);
}

// ----------------------------------------------------------------------------------------------------------------
// Per-token log-probabilities of given targets (rollout scoring: replaces logits.log_softmax(-1).gather(ids) of
// _get_per_token_logps, src/train/RL/.../open_r1/trainer/grpo_trainer.py:494-504).  One workgroup per row of bf16
// logits; two passes over the row (maximum, then sum of expf(l - max)), both with a fixed reduction order, so the
// result is a function of the row alone.  out[r] = l[target] - max - logf(sum).
__global__ void __launch_bounds__(256) k_token_logprob(const bf16_t* __restrict__ logits, int ld, int vocab,
                                                       const int* __restrict__ targets, float* __restrict__ out) {
    __shared__ float red[4];
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const bf16_t* row = logits + (size_t)r * ld;
    const int nv = vocab / 8;  // 16-byte groups (ld % 8 == 0)
    float m = -INFINITY;
    for (int g = tid; g < nv; g += 256) {
        const uint4 q = *(const uint4*)(row + (size_t)g * 8);
        const uint32_t u[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            m = fmaxf(m, __uint_as_float(u[j] << 16));
            m = fmaxf(m, __uint_as_float(u[j] & 0xffff0000u));
        }
    }
    for (int i = nv * 8 + tid; i < vocab; i += 256) m = fmaxf(m, bf16_to_f32(row[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (lane == 0) red[w] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int g = tid; g < nv; g += 256) {
        const uint4 q = *(const uint4*)(row + (size_t)g * 8);
        const uint32_t u[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sum += expf(__uint_as_float(u[j] << 16) - m);
            sum += expf(__uint_as_float(u[j] & 0xffff0000u) - m);
        }
    }
    for (int i = nv * 8 + tid; i < vocab; i += 256) sum += expf(bf16_to_f32(row[i]) - m);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if (lane == 0) red[w] = sum;
    __syncthreads();
    if (tid == 0) {
        const float tot = (red[0] + red[1]) + (red[2] + red[3]);
        const int t = targets[r];
        out[r] = (t >= 0 && t < vocab) ? bf16_to_f32(row[t]) - m - logf(tot) : 0.f;
    }
}

void ze_launch_token_logprob(const bf16_t* logits, int ld, int vocab, const int* targets, float* out, int rows,
                             hipStream_t s) {
    if (rows <= 0) return;
    hipLaunchKernelGGL(k_token_logprob, dim3(rows), dim3(256), 0, s, logits, ld, vocab, targets, out);
}
