// Decode-path weight-streaming kernels (SURVEY.md K16-K22 at one token per step): out[N] = W[N,K] . x[K].
// This is THE dominant kernel family of the batch-1 zoom chain: every decode step streams all 6.17 GB of bf16
// weights once, so these kernels are HBM-bound and everything else is fused around the stream:
//   prologue : token-embedding fetch (layer 0) and RMSNorm of the 4-KB activation vector (recomputed per block)
//   epilogue : +bias, bf16 rounding, M-RoPE + KV-cache append (QKV), SiLU(gate)*up (gate/up rows are interleaved
//              in blocks of 16 in the packed weight), residual add in place, fp32 logits.
// Weights go straight to VGPRs with non-temporal 16-B loads (no LDS round trip: each row is read by exactly one
// wave), 8-16 independent 1-KiB wave loads in flight per wave and NO per-lane branch around any load; x sits in
// LDS as bf16.  Issue order follows the in-order vmcnt retirement: activation vector + norm weight first, then the
// first trip of the weight stream (so the prologue only waits for two L2-resident vectors while the HBM latency
// of the first weight loads hides behind it), then the epilogue operands (bias / residual / cos-sin).
#include "ze_gemv_kernel.h"

bool ze_launch_gemv8(int epi, const ze_gemv_args& a, hipStream_t s);

// overrides set through ze_tune(): [0] down variant, [1] gate_up variant, [2] grid cap, [3] fused attention block
int ze_gemv_knobs[24] = {0};
int ze_live_engines = 0;  // engines alive in this process (one GPU per process): counted by ze_engine_create / ze_engine_destroy

bool ze_launch_gemv(int epi, const ze_gemv_args& a, hipStream_t s) {
    // shape policy: long-K / few-row matrices split K over the 4 waves of a block (each wave streams its K/4 share
    // in ONE trip of 6 chunks: 12 loads in flight per lane, no second latency-exposed phase); many-row matrices give
    // each wave two row pairs (16 loads in flight per lane).
    if ((size_t)((a.K + 511) / 512 + 2) * 1024 > 60000) return false;  // x must fit the LDS stage
    const bool long_k = a.K > 4096;
    const bool many_rows = a.N >= 8192;
    if (a.W8) return ze_launch_gemv8(epi, a, s);  // fp8 weight stream (ze_gemv8.hip)
    switch (epi) {
        // (K split over the waves -- 1280 one-pair workgroups, five per CU -- measured 8.2 us against 6.9: every workgroup
        //  pays the x-staging prologue)
        case ZE_GV_QKV_ROPE: launch_gemv_cfg<ZE_GV_QKV_ROPE, 1, 1, 4>(a, s); break;
        case ZE_GV_SWIGLU:
            if (many_rows && ze_gemv_knobs[1] == 1) launch_gemv_cfg<ZE_GV_SWIGLU, 2, 1, 4>(a, s);
            else launch_gemv_cfg<ZE_GV_SWIGLU, 1, 1, 4>(a, s);
            break;
        case ZE_GV_RESIDUAL:
            if (long_k && ze_gemv_knobs[0] == 1) launch_gemv_cfg<ZE_GV_RESIDUAL, 2, 4, 6>(a, s);
            else if (long_k && ze_gemv_knobs[0] == 2) launch_gemv_cfg<ZE_GV_RESIDUAL, 1, 4, 4>(a, s);
            else if (long_k && ze_gemv_knobs[0] == 3) launch_gemv_cfg<ZE_GV_RESIDUAL, 1, 1, 4>(a, s);
            else if (long_k) launch_gemv_cfg<ZE_GV_RESIDUAL, 1, 4, 6>(a, s);
            else launch_gemv_cfg<ZE_GV_RESIDUAL, 1, 1, 4>(a, s);
            break;
        case ZE_GV_LOGITS:
            if (many_rows) launch_gemv_cfg<ZE_GV_LOGITS, 2, 1, 4>(a, s);
            else launch_gemv_cfg<ZE_GV_LOGITS, 1, 1, 4>(a, s);
            break;
        default:
            if (long_k) launch_gemv_cfg<ZE_GV_PLAIN, 1, 4, 6>(a, s);
            else launch_gemv_cfg<ZE_GV_PLAIN, 1, 1, 4>(a, s);
            break;
    }
    return true;
}

// Rejected experiment (measured, round 1): a side-branch kernel in the decode graph that pre-touches the next MLP
// weights so they sit in the 256-MiB Infinity Cache while the latency-bound QKV / attention / O-proj kernels leave
// HBM idle.  With 32-256 prefetch blocks per layer the whole question got 1.3-2.3x SLOWER (the prefetcher competes
// with the chain for CUs / memory queues and the join stalls the next layer), so it is not in the tree.
