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
        const int ch = pos ^ ze_kv_swz(row);
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

template <int N>
__device__ __forceinline__ void ab_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ----------------------------------------------------------------------------------------------------------------
// Every WAVE a stream of its own.  What bounds k_attn_decode_stream is not its arithmetic (compiled out, the launch is 7 %
// shorter at 256 chains) but its per-workgroup fixed cost against a ring that holds two 16-KB rounds behind a workgroup
// barrier.  Here a part is 192 keys = three rounds of 64, and wave w owns keys 16 w .. 16 w + 15 of every round, start to
// finish:
//   * its K rows go straight from global memory into MFMA A-operand registers (lane (fr, fq): key fr, 16 B of d -- no LDS);
//     its V rows arrive by LDS-DMA in three 4-KB stages of the wave's own, read back transposed by ds_read_b64_tr_b16;
//     ALL three rounds are requested at t = 0 and ordered by the wave's own counted vmcnt -- no barrier, no loop, no refill;
//   * S^T = K Q^T is ONE 16-key tile per wave and round (4 MFMAs instead of every wave forming every tile), the softmax
//     statistics of the wave's keys stay lane-local plus two shuffles, and P^T feeds the B operand of O^T += V^T P^T
//     (v_mfma_f32_16x16x16_bf16: the accumulator lane that holds S^T[4 fq + r][fr] is the lane that needs those four keys);
//   * every wave keeps its own running (m, l, O^T) -- a flash-decoding split inside the workgroup -- and the four are
//     merged in wave order once (LDS, one barrier), then published and merged across parts as before.
// The loads of Q / K and the MFMAs that consume them are inline asm: a load the compiler knows of gets its own s_waitcnt,
// computed without the DMAs in the queue (too strict: it drains everything), and a register the compiler merely sees
// DEFINED by an asm load may be copied before the data has landed (seen: tied operands of a separate wait statement made
// hipcc copy the K registers ahead of the wait on two of three paths).  So the code is straight-line -- always three
// rounds; rounds past the end of a short last part re-read the last row and are masked -- and every load result is first
// touched by the asm statement that waits for it.
// Parts are 192 keys whatever the context (up to max_ctx / 192 of them): a function of the chain's own length alone, like
// every sum order here (batch invariance).  The sums are ordered differently from k_attn_decode_stream's, so the two
// kernels agree within rounding, not bit for bit.
#define AW_TOK 64
#define AW_ROUNDS 3
#define AW_PART (AW_TOK * AW_ROUNDS)
#define AW_VSTAGE 4096
#define AW_VSTAGES 3                     // V stages per wave (2: round 2's tile takes round 0's stage once that has been read --
                                         // four workgroups per CU instead of three: 24.5 against 23.2 us at 64 chains, equal at 256)

typedef __attribute__((ext_vector_type(4))) unsigned int aw_u32x4;

// Measurement hooks of tools/probes/fence_hunt.sh (the shipped build defines none of them): the part length of the pipelined
// kernel, the two sites where round 4 fenced the scheduler (A: behind the wait + K Q^T statement, B: behind the P V loop; empty
// since round 5 -- the cause was the asm conversion in ad_pack_bf16, see aw_wait_qk; -DAW_FENCED restores the fences for A/B
// runs) and text spliced into the statement.
#ifndef AW_LONG_ROUNDS
#define AW_LONG_ROUNDS 6
#endif
#ifdef AW_FENCED
#define AW_FENCE_A() __builtin_amdgcn_sched_barrier(0)
#define AW_FENCE_B() __builtin_amdgcn_sched_barrier(0)
#endif
#ifndef AW_FENCE_A
#define AW_FENCE_A() do {} while (0)
#endif
#ifndef AW_FENCE_B
#define AW_FENCE_B() do {} while (0)
#endif
#ifndef AW_ASM_PRE
#define AW_ASM_PRE ""
#endif
#ifndef AW_ASM_POST
#define AW_ASM_POST ""
#endif

// S^T tile of one round: wait until all but the N youngest memory operations of the wave have retired, then K Q^T.
// The statement pads its last MFMA with 24 wait states (an 8-pass MFMA's result may be read 11 after it): hipcc cannot see that
// the statement is four MFMAs, so it inserts none of its own.  ROUND 4 fenced every use with __builtin_amdgcn_sched_barrier(0)
// because two instantiations of k_attn_decode_wave_long (192- / 256-key parts) returned wrong rows without -- keys 2, 3 of every
// group of four missing from one P V product.  ROUND 5 found the cause, and it is not this statement: ad_pack_bf16 converted P
// with `asm("v_cvt_pk_bf16_f32 ...")`, a VALU write hipcc cannot see, and where the scheduler put the first P V MFMA one
// instruction behind it the MFMA read the register's old value (gfx950 wants two wait states between a VALU write and an MFMA
// read; the compiler pads only pairs it knows).  Either fence happened to keep the pair apart; two s_nops behind the cvt, or the
// conversion as a plain vector conversion (what ships), fix it with no fence at all -- tools/probes/fence_hunt.sh has the eleven
// variants, tools/check_mfma_hazards.py flags exactly the failing ones from the assembly alone and is a CPU test over the whole
// library.  The fences are gone (-DAW_FENCED brings them back for A/B runs).
template <int N>
__device__ __forceinline__ ad_f32x4 aw_wait_qk(const aw_u32x4 (&k4)[4], const aw_u32x4 (&q4)[4]) {
    ad_f32x4 sacc;
    asm volatile(
        "s_waitcnt vmcnt(%9)\n\t" AW_ASM_PRE
        "v_mfma_f32_16x16x32_bf16 %0, %1, %5, 0\n\t"
        "v_mfma_f32_16x16x32_bf16 %0, %2, %6, %0\n\t"
        "v_mfma_f32_16x16x32_bf16 %0, %3, %7, %0\n\t"
        "v_mfma_f32_16x16x32_bf16 %0, %4, %8, %0\n\t"
        "s_nop 15\n\ts_nop 7" AW_ASM_POST
        : "=&v"(sacc)
        : "v"(k4[0]), "v"(k4[1]), "v"(k4[2]), "v"(k4[3]), "v"(q4[0]), "v"(q4[1]), "v"(q4[2]), "v"(q4[3]), "n"(N)
        : "memory");
    return sacc;
}

template <int MB>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(AW_VSTAGES == 2 ? 4 : 3))) k_attn_decode_wave(
    const bf16_t* __restrict__ q, int q_row_stride, const bf16_t* __restrict__ kcache, const bf16_t* __restrict__ vcache,
    size_t cache_seq_stride, const ze_seq_dev* __restrict__ st_base, const int* __restrict__ seq_ids, int heads, int kv_heads,
    int max_ctx, float scale_log2e, float* __restrict__ ws, int max_parts, unsigned* __restrict__ tickets,
    bf16_t* __restrict__ out, int out_row_stride, int x_rot, const int* __restrict__ prefix) {
    constexpr int D = 128;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];  // 4 waves x AW_VSTAGES V stages of 4 KB
    __shared__ float s_ml[4][32];                                   // the waves' (m, l) per head column
    const int bz = blockIdx.y;
    // Workgroups go to the eight XCDs round robin by their linear id, blockIdx.y * gridDim.x + blockIdx.x.  A chain's work is
    // a PREFIX of its x range (the parts it has), so with gridDim.x a multiple of 8 every chain would put the same part on the
    // same XCD and the XCDs of the high part numbers would idle (measured: 16 or 32 workgroups per chain 3.97 / 2.49 TB/s
    // against 4.58 with 22).  The x range is therefore rotated, by 3 per group of FOUR consecutive chains (x_rot = 3 | 2 << 4: a
    // bijection per chain, whatever the grid; four chains of a tile in a row keep the same part on the same XCD, whose L2 then
    // serves the prefix rows they share -- rotating per chain cost the shared case 2 %, per 8 or 16 chains the balance).
    const int xr = (int)((blockIdx.x + (unsigned)(x_rot & 15) * ((unsigned)bz >> (x_rot >> 4))) % gridDim.x);
    const int kvh = xr % kv_heads, part = xr / kv_heads;
    const int seq = seq_ids[bz];
    const int ctx = st_base[seq].ctx + 1;
    const int nparts = (ctx + AW_PART - 1) / AW_PART;
    if (part >= nparts) return;  // workgroup-uniform: no part, no ticket
    const int G = heads / kv_heads;
    const int t0 = part * AW_PART, t1 = min(ctx, t0 + AW_PART);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const bf16_t* kb = kcache + (size_t)seq * cache_seq_stride + (size_t)kvh * max_ctx * D;
    const bf16_t* vb = vcache + (size_t)seq * cache_seq_stride + (size_t)kvh * max_ctx * D;
    // rows below P: the same bits live in the source chain's cache (prefix[seq], ze_engine::pfx_dev) -- read THAT copy, the one
    // every question of the tile reads, so that the image prefix crosses the HBM interface once per step and not once per chain
    const int hint = prefix ? prefix[seq] : 0;
    const int pfx_rows = hint & 0xffff;
    const long long pfx_delta = ((long long)(hint >> 16) - (long long)seq) * (long long)cache_seq_stride;  // in elements
    const unsigned ring_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) uint8_t*)smem) +
                              (unsigned)wid * (AW_VSTAGES * AW_VSTAGE);
    const uint8_t* ring = smem + wid * (AW_VSTAGES * AW_VSTAGE);

    // ---- every request of the wave, in the order it will be waited for: Q (4), then per round K (4 loads) + V (4 DMA pieces)
    // (Q: lane = head column fr, d = ks*32 + fq*8 .. +7; columns >= G repeat head G - 1: finite, never stored)
    aw_u32x4 qf4[4], kreg[AW_ROUNDS][4];
    {
        const bf16_t* qsrc = q + (size_t)bz * q_row_stride + (kvh * G + min(fr, G - 1)) * D + fq * 8;
        asm volatile(
            "global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:64\n\t"
            "global_load_dwordx4 %2, %4, off offset:128\n\tglobal_load_dwordx4 %3, %4, off offset:192"
            : "=&v"(qf4[0]), "=&v"(qf4[1]), "=&v"(qf4[2]), "=&v"(qf4[3])
            : "v"(qsrc)
            : "memory");
    }
    const int r4 = lane >> 4, pos = lane & 15;
    auto issue_v = [&](int u) {  // the V tile of round u into stage u % 2: four 1-KB pieces (no register results)
        const int tok0 = t0 + u * AW_TOK + wid * 16;
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {  // piece pp = rows 4pp .. 4pp + 3; the ad_off swizzle on the source chunk (ab_issue)
            const int row = 4 * pp + r4;
            const int ch = pos ^ ze_kv_swz(row);
            const int tok = min(tok0 + row, t1 - 1);
            const bf16_t* src = vb + (tok < pfx_rows ? pfx_delta : 0ll) + (size_t)tok * D + ch * 8;
            unsigned keep;
            asm volatile(
                "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                : "=&s"(keep)
                : "v"(src), "s"(ring_lds + (unsigned)(u % AW_VSTAGES) * AW_VSTAGE + (unsigned)pp * 1024u)
                : "memory");
        }
    };
#pragma unroll
    for (int u = 0; u < AW_ROUNDS; ++u) {
        const int tok0 = t0 + u * AW_TOK + wid * 16;  // rows past t1 re-read row t1 - 1 (finite; masked below)
        const int ktok = min(tok0 + fr, t1 - 1);
        const bf16_t* ksrc = kb + (ktok < pfx_rows ? pfx_delta : 0ll) + (size_t)ktok * D + fq * 8;
        asm volatile(
            "global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:64\n\t"
            "global_load_dwordx4 %2, %4, off offset:128\n\tglobal_load_dwordx4 %3, %4, off offset:192"
            : "=&v"(kreg[u][0]), "=&v"(kreg[u][1]), "=&v"(kreg[u][2]), "=&v"(kreg[u][3])
            : "v"(ksrc)
            : "memory");
        if (u < AW_VSTAGES) issue_v(u);
    }

    ad_f32x4 oacc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) oacc[j] = ad_f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    const int tq = fr >> 2, tp = fr & 3;
#pragma unroll
    for (int u = 0; u < AW_ROUNDS; ++u) {
        // operations retire in issue order -- Q, K0, V0, K1, V1, K2, V2 (with two stages V2 goes out behind round 0): round 0
        // has landed when the 16 (12) behind V0 are left, round 1 when K2, V2 (8) are
        ad_f32x4 sacc;
        if (u == 0) sacc = aw_wait_qk<(AW_VSTAGES == 2 ? 12 : 16)>(kreg[0], qf4);
        else if (u == 1) sacc = aw_wait_qk<8>(kreg[1], qf4);
        else sacc = aw_wait_qk<0>(kreg[2], qf4);
        AW_FENCE_A();
        const int kbase = t0 + u * AW_TOK + wid * 16 + fq * 4;
        float p[4];
        float mx = -INFINITY;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const float sv = (kbase + rr < t1) ? sacc[rr] * scale_log2e : -INFINITY;
            p[rr] = sv;
            mx = fmaxf(mx, sv);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = exp2f(m_run - m_use);  // m_run = -inf -> 0
        float rs = 0.f;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            p[rr] = __builtin_amdgcn_exp2f(p[rr] - m_use);  // arguments <= 0
            rs += p[rr];
        }
        rs += __shfl_xor(rs, 16, 64);
        rs += __shfl_xor(rs, 32, 64);
        l_run = l_run * alpha + rs;
        m_run = m_new;
        // O^T += V^T P^T over the wave's 16 keys (P rounded to bf16, as HF's eager attention rounds it)
        const uint2 pq = make_uint2(ad_pack_bf16(p[0], p[1]), ad_pack_bf16(p[2], p[3]));
        const ad_v4s pb = *reinterpret_cast<const ad_v4s*>(&pq);
        const uint8_t* sV = ring + (u % AW_VSTAGES) * AW_VSTAGE;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const ad_v4s va = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) ad_v4s*)(sV + ad_off(fq * 4 + tq, j * 2 + (tp >> 1)) + 8 * (tp & 1)));
            oacc[j][0] *= alpha;
            oacc[j][1] *= alpha;
            oacc[j][2] *= alpha;
            oacc[j][3] *= alpha;
            oacc[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(va, pb, oacc[j], 0, 0, 0);
        }
        AW_FENCE_B();
        if (AW_VSTAGES == 2 && u == 0) {  // stage 0 has been read (the transposed reads have returned): it takes round 2's tile
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            issue_v(2);
        }
    }
    // ---- the four waves' results meet in LDS (their stages are idle now): wave w, [8 tiles][64 lanes] float4 in its own
    // stages + its (m, l) per head column; then wave w merges d-tiles 2w, 2w + 1 of the four in wave order
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    {
        ad_f32x4* mine = reinterpret_cast<ad_f32x4*>(smem + wid * (AW_VSTAGES * AW_VSTAGE));
#pragma unroll
        for (int j = 0; j < 8; ++j) mine[j * 64 + lane] = oacc[j];
        if (fq == 0) {
            s_ml[wid][fr] = m_run;
            s_ml[wid][16 + fr] = l_run;
        }
    }
    __syncthreads();
    float mw[4], lw[4];
    float m_all = -INFINITY;
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        mw[x] = s_ml[x][fr];
        lw[x] = s_ml[x][16 + fr];
        m_all = fmaxf(m_all, mw[x]);
    }
    float l_all = 0.f;
    ad_f32x4 om[2];
    om[0] = om[1] = ad_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        const float w = (mw[x] == -INFINITY) ? 0.f : exp2f(mw[x] - m_all);
        l_all += w * lw[x];
        const ad_f32x4* theirs = reinterpret_cast<const ad_f32x4*>(smem + x * (AW_VSTAGES * AW_VSTAGE));
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const ad_f32x4 t = theirs[(2 * wid + jj) * 64 + lane];
            om[jj][0] += w * t[0];
            om[jj][1] += w * t[1];
            om[jj][2] += w * t[2];
            om[jj][3] += w * t[3];
        }
    }
    __syncthreads();  // the tail reuses the LDS

    // partial of head fr for this part, published write-through (the layout and the merge of k_attn_decode_stream)
    float* wsb = ws + (size_t)bz * max_parts * heads * AD_STRIDE;
    if (fr < G) {
        const uint32_t dst = (uint32_t)((part * heads + kvh * G + fr) * AD_STRIDE * 4);
        if (wid == 0 && fq == 0) ad_store16<true>(wsb, dst, m_all, l_all, 0.f, 0.f);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
            ad_store16<true>(wsb, dst + (uint32_t)((4 + (2 * wid + jj) * 16 + fq * 4) * 4), om[jj][0], om[jj][1], om[jj][2], om[jj][3]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned* flag = reinterpret_cast<unsigned*>(smem + 16 * 1024);
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

// ----------------------------------------------------------------------------------------------------------------
// Round 4: the same per-wave streams over LONGER parts (ROUNDS x 64 keys, ROUNDS = 6: 384 keys), three rounds in flight
// throughout.  A 192-key part requests everything at t = 0 and then only drains: chain-state look-up, ramp, cross-wave merge,
// publish, ticket -- about 6 of a workgroup's 22 us -- are paid per 192 keys, and the queue is empty while they run.  Here
// round u + 3 is requested as soon as round u has been consumed (its K registers and its V stage are free: the K loads re-use
// kreg[u % 3], the V pieces stage u % 3 behind an lgkmcnt(0) that retires the transposed reads), so a workgroup keeps two to
// three rounds = 32-48 KB in flight until its last round and pays the fixed costs once per 384 keys: half the partials,
// tickets and merges.  A part's LIVE rounds NR = ceil(keys / 64) select one of ROUNDS straight-line bodies (a short last part
// requests and computes nothing it does not have -- the 192-key kernel always ran three rounds, masked), each with literal
// wait counts: retirement is in issue order, and at the wait for round u exactly the min(2, NR - 1 - u) younger rounds (8
// operations each: 4 K loads + 4 V pieces) may stay outstanding.  The bodies only meet after their last load has been consumed
// (ordinary values from there on: no register holding an in-flight load crosses a control-flow merge -- see the note on the
// 192-key kernel).  Parts stay a function of the chain's own length alone (batch invariance); the per-key arithmetic is that
// of the 192-key kernel, the grouping of the partial sums differs (384-key parts), so the two agree within rounding.
template <int NR>
__device__ __forceinline__ void aw_run_part(const bf16_t* __restrict__ qsrc, const bf16_t* __restrict__ kb,
                                            const bf16_t* __restrict__ vb, long long pfx_delta, int pfx_rows, int t0, int t1,
                                            int wid, int lane, unsigned ring_lds, const uint8_t* ring, float scale_log2e,
                                            ad_f32x4 (&oacc)[8], float& m_run, float& l_run) {
    constexpr int D = 128;
    const int fr = lane & 15, fq = lane >> 4;
    const int r4 = lane >> 4, pos = lane & 15;
    aw_u32x4 qf4[4], kreg[3][4];
    asm volatile(
        "global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:64\n\t"
        "global_load_dwordx4 %2, %4, off offset:128\n\tglobal_load_dwordx4 %3, %4, off offset:192"
        : "=&v"(qf4[0]), "=&v"(qf4[1]), "=&v"(qf4[2]), "=&v"(qf4[3])
        : "v"(qsrc)
        : "memory");
    auto issue_v = [&](int u) {  // the V tile of round u into stage u % 3: four 1-KB pieces (no register results)
        const int tok0 = t0 + u * AW_TOK + wid * 16;
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
            const int row = 4 * pp + r4;
            const int ch = pos ^ ze_kv_swz(row);
            const int tok = min(tok0 + row, t1 - 1);
            const bf16_t* src = vb + (tok < pfx_rows ? pfx_delta : 0ll) + (size_t)tok * D + ch * 8;
            unsigned keep;
            asm volatile(
                "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                : "=&s"(keep)
                : "v"(src), "s"(ring_lds + (unsigned)(u % 3) * AW_VSTAGE + (unsigned)pp * 1024u)
                : "memory");
        }
    };
#define AW_ISSUE_K(U, KR)                                                                                                   \
    do {                                                                                                                    \
        const int ktok_ = min(t0 + (U) * AW_TOK + wid * 16 + fr, t1 - 1);                                                   \
        const bf16_t* ksrc_ = kb + (ktok_ < pfx_rows ? pfx_delta : 0ll) + (size_t)ktok_ * D + fq * 8;                       \
        asm volatile(                                                                                                       \
            "global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:64\n\t"                              \
            "global_load_dwordx4 %2, %4, off offset:128\n\tglobal_load_dwordx4 %3, %4, off offset:192"                      \
            : "=&v"((KR)[0]), "=&v"((KR)[1]), "=&v"((KR)[2]), "=&v"((KR)[3])                                                \
            : "v"(ksrc_)                                                                                                    \
            : "memory");                                                                                                    \
    } while (0)
#pragma unroll
    for (int u = 0; u < 3; ++u)
        if (u < NR) {
            AW_ISSUE_K(u, kreg[u]);
            issue_v(u);
        }
    const int tq = fr >> 2, tp = fr & 3;
#pragma unroll
    for (int u = 0; u < NR; ++u) {
        const int younger = (NR - 1 - u) < 2 ? (NR - 1 - u) : 2;  // rounds requested behind round u at this point
        ad_f32x4 sacc;
        if (younger == 2) sacc = aw_wait_qk<16>(kreg[u % 3], qf4);
        else if (younger == 1) sacc = aw_wait_qk<8>(kreg[u % 3], qf4);
        else sacc = aw_wait_qk<0>(kreg[u % 3], qf4);
        AW_FENCE_A();
        const int kbase = t0 + u * AW_TOK + wid * 16 + fq * 4;
        float p[4];
        float mx = -INFINITY;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const float sv = (kbase + rr < t1) ? sacc[rr] * scale_log2e : -INFINITY;
            p[rr] = sv;
            mx = fmaxf(mx, sv);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = exp2f(m_run - m_use);  // m_run = -inf -> 0
        float rs = 0.f;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            p[rr] = __builtin_amdgcn_exp2f(p[rr] - m_use);  // arguments <= 0
            rs += p[rr];
        }
        rs += __shfl_xor(rs, 16, 64);
        rs += __shfl_xor(rs, 32, 64);
        l_run = l_run * alpha + rs;
        m_run = m_new;
        const uint2 pq = make_uint2(ad_pack_bf16(p[0], p[1]), ad_pack_bf16(p[2], p[3]));
        const ad_v4s pb = *reinterpret_cast<const ad_v4s*>(&pq);
        const uint8_t* sV = ring + (u % 3) * AW_VSTAGE;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const ad_v4s va = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) ad_v4s*)(sV + ad_off(fq * 4 + tq, j * 2 + (tp >> 1)) + 8 * (tp & 1)));
            oacc[j][0] *= alpha;
            oacc[j][1] *= alpha;
            oacc[j][2] *= alpha;
            oacc[j][3] *= alpha;
            oacc[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(va, pb, oacc[j], 0, 0, 0);
        }
        AW_FENCE_B();
        if (u + 3 < NR) {  // round u is consumed: its K registers and its V stage take round u + 3
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            AW_ISSUE_K(u + 3, kreg[u % 3]);
            issue_v(u + 3);
        }
    }
#undef AW_ISSUE_K
}

// SPLIT (round 6, not the shipped form): the split-row partition and the paired prefix parts are an instantiation of their own, so that
// the shipped kernel is the round-5 code to the instruction (the disabled branches cost it 1-2 % in a same-box A/B: 83.1 -> 85.0 us).
template <int MB, int ROUNDS, bool SPLIT = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) k_attn_decode_wave_long(
    const bf16_t* __restrict__ q, int q_row_stride, const bf16_t* __restrict__ kcache, const bf16_t* __restrict__ vcache,
    size_t cache_seq_stride, const ze_seq_dev* __restrict__ st_base, const int* __restrict__ seq_ids, int heads, int kv_heads,
    int max_ctx, float scale_log2e, float* __restrict__ ws, int max_parts, unsigned* __restrict__ tickets,
    bf16_t* __restrict__ out, int out_row_stride, int x_rot, const int* __restrict__ prefix, const int* __restrict__ mate, int use_split) {
    constexpr int D = 128, PART = AW_TOK * ROUNDS;
    static_assert(ROUNDS >= 3 && ROUNDS <= 8, "part length");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];  // 4 waves x 3 V stages of 4 KB
    __shared__ float s_ml[4][32];                                   // the waves' (m, l) per head column
    const int bz = blockIdx.y;
    const int xr = (int)((blockIdx.x + (unsigned)(x_rot & 15) * ((unsigned)bz >> (x_rot >> 4))) % gridDim.x);  // (k_attn_decode_wave)
    const int kvh = xr % kv_heads, part = xr / kv_heads;
    const int seq = seq_ids[bz];
    const int ctx = st_base[seq].ctx + 1;
    // Round 6: the chain's SPLIT ROW (ze_seq_dev::split: the end of its first image block -- a property of the chain's own tokens,
    // set when they are prefilled or copied, never of the batch) cuts the parts: [0, split) in PART-key pieces, then [split, ctx) in
    // PART-key pieces.  The rows below the split are what the questions of a tile share, so a PREFIX part holds no row of the chain's
    // own -- and two chains that read those rows from one holder (mate[], built by the host per burst) share ONE workgroup for it:
    // the second chain's q heads take the eight MFMA columns that otherwise repeat head G - 1.  A column's arithmetic does not depend
    // on what the other columns hold: a chain's partial is the same bits paired or alone, leader or follower.
    const int sp_ = (SPLIT && use_split) ? st_base[seq].split : 0;
    const int split = (SPLIT && sp_ > 0 && sp_ < ctx) ? sp_ : 0;
    const int np0 = (split + PART - 1) / PART;
    const int nparts = np0 + (ctx - split + PART - 1) / PART;
    if (part >= nparts) return;  // workgroup-uniform: no part, no ticket
    const int G = heads / kv_heads;
    const int t0 = part < np0 ? part * PART : split + (part - np0) * PART;
    const int t1 = part < np0 ? min(split, t0 + PART) : min(ctx, t0 + PART);
    int bz2 = bz;                // the chain whose q heads ride in columns G .. 2G - 1 (bz: nobody's)
    if (SPLIT && part < np0 && mate != nullptr && 2 * G <= 16) {
        const int m = mate[bz];
        if (m >= 0) {
            if (m < bz) return;  // the pair's leader (the lower row of the batch) computes this part for both; it takes this chain's ticket too
            bz2 = m;
        }
    }
    const bool paired = SPLIT && bz2 != bz;   // workgroup-uniform
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const bf16_t* kb = kcache + (size_t)seq * cache_seq_stride + (size_t)kvh * max_ctx * D;
    const bf16_t* vb = vcache + (size_t)seq * cache_seq_stride + (size_t)kvh * max_ctx * D;
    const int hint = prefix ? prefix[seq] : 0;
    const int pfx_rows = hint & 0xffff;
    const long long pfx_delta = ((long long)(hint >> 16) - (long long)seq) * (long long)cache_seq_stride;  // in elements
    const unsigned ring_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) uint8_t*)smem) +
                              (unsigned)wid * (3 * AW_VSTAGE);
    const uint8_t* ring = smem + wid * (3 * AW_VSTAGE);
    const int qrow = (paired && fr >= G) ? bz2 : bz;
    const int qhead = (paired && fr >= G) ? min(fr - G, G - 1) : min(fr, G - 1);
    const bf16_t* qsrc = q + (size_t)qrow * q_row_stride + (kvh * G + qhead) * D + fq * 8;

    ad_f32x4 oacc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) oacc[j] = ad_f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    const int nr = __builtin_amdgcn_readfirstlane((t1 - t0 + AW_TOK - 1) / AW_TOK);  // live rounds of this part: 1 .. ROUNDS
#define AW_CASE(N)                                                                                                          \
    case N:                                                                                                                 \
        if constexpr (N <= ROUNDS)                                                                                          \
            aw_run_part<N>(qsrc, kb, vb, pfx_delta, pfx_rows, t0, t1, wid, lane, ring_lds, ring, scale_log2e, oacc, m_run, l_run); \
        break
    switch (nr) {
        AW_CASE(1);
        AW_CASE(2);
        AW_CASE(3);
        AW_CASE(4);
        AW_CASE(5);
        AW_CASE(6);
        AW_CASE(7);
        AW_CASE(8);
        default: break;
    }
#undef AW_CASE
    // ---- the four waves' results meet in LDS (their stages are idle now), merged in wave order (k_attn_decode_wave)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    {
        ad_f32x4* mine = reinterpret_cast<ad_f32x4*>(smem + wid * (3 * AW_VSTAGE));
#pragma unroll
        for (int j = 0; j < 8; ++j) mine[j * 64 + lane] = oacc[j];
        if (fq == 0) {
            s_ml[wid][fr] = m_run;
            s_ml[wid][16 + fr] = l_run;
        }
    }
    __syncthreads();
    float mw[4], lw[4];
    float m_all = -INFINITY;
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        mw[x] = s_ml[x][fr];
        lw[x] = s_ml[x][16 + fr];
        m_all = fmaxf(m_all, mw[x]);
    }
    float l_all = 0.f;
    ad_f32x4 om[2];
    om[0] = om[1] = ad_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        const float w = (mw[x] == -INFINITY) ? 0.f : exp2f(mw[x] - m_all);
        l_all += w * lw[x];
        const ad_f32x4* theirs = reinterpret_cast<const ad_f32x4*>(smem + x * (3 * AW_VSTAGE));
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const ad_f32x4 t = theirs[(2 * wid + jj) * 64 + lane];
            om[jj][0] += w * t[0];
            om[jj][1] += w * t[1];
            om[jj][2] += w * t[2];
            om[jj][3] += w * t[3];
        }
    }
    __syncthreads();  // the tail reuses the LDS

    float* wsb = ws + (size_t)bz * max_parts * heads * AD_STRIDE;
    float* wsb2 = ws + (size_t)bz2 * max_parts * heads * AD_STRIDE;   // (the mate's slots; a workgroup-uniform base each: no waterfall)
    if (fr < G) {
        const uint32_t dst = (uint32_t)((part * heads + kvh * G + fr) * AD_STRIDE * 4);
        if (wid == 0 && fq == 0) ad_store16<true>(wsb, dst, m_all, l_all, 0.f, 0.f);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
            ad_store16<true>(wsb, dst + (uint32_t)((4 + (2 * wid + jj) * 16 + fq * 4) * 4), om[jj][0], om[jj][1], om[jj][2], om[jj][3]);
    } else if (paired && fr < 2 * G) {
        const uint32_t dst = (uint32_t)((part * heads + kvh * G + (fr - G)) * AD_STRIDE * 4);
        if (wid == 0 && fq == 0) ad_store16<true>(wsb2, dst, m_all, l_all, 0.f, 0.f);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
            ad_store16<true>(wsb2, dst + (uint32_t)((4 + (2 * wid + jj) * 16 + fq * 4) * 4), om[jj][0], om[jj][1], om[jj][2], om[jj][3]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned* flag = reinterpret_cast<unsigned*>(smem + 16 * 1024);
    int ctx2 = ctx, nparts2 = nparts;
    if (paired) {   // the mate's own length: its parts beyond the split are its own business, its ticket counts all of them
        ctx2 = st_base[seq_ids[bz2]].ctx + 1;
        nparts2 = np0 + (ctx2 - split + PART - 1) / PART;
    }
    if (threadIdx.x == 0) {
        unsigned* t = tickets + (size_t)bz * kv_heads + kvh;
        const unsigned old = __hip_atomic_fetch_add(t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned last = old == (unsigned)nparts - 1u ? 1u : 0u;
        if (last) __hip_atomic_store(t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (paired) {
            unsigned* t2 = tickets + (size_t)bz2 * kv_heads + kvh;
            const unsigned old2 = __hip_atomic_fetch_add(t2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old2 == (unsigned)nparts2 - 1u) {
                __hip_atomic_store(t2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last |= 2u;
            }
        }
        *flag = last;
    }
    __syncthreads();
    const unsigned last = *flag;
    if (last == 0u) return;
    __syncthreads();
    float* sW = reinterpret_cast<float*>(smem);
    if (last & 1u) {
        if (out_row_stride < 0)
            attn_merge_group<MB>(sW, sW + AD_GMAX * 64, wsb, ctx, kvh, heads, kv_heads, max_parts, out, -out_row_stride, bz, nparts);
        else
            attn_merge_group<MB>(sW, sW + AD_GMAX * 64, wsb, ctx, kvh, heads, kv_heads, max_parts,
                                 out + (size_t)bz * out_row_stride, 0, 0, nparts);
    }
    if (last & 2u) {   // this workgroup also drew the MATE's last ticket: merge that chain too
        __syncthreads();
        if (out_row_stride < 0)
            attn_merge_group<MB>(sW, sW + AD_GMAX * 64, wsb2, ctx2, kvh, heads, kv_heads, max_parts, out, -out_row_stride, bz2, nparts2);
        else
            attn_merge_group<MB>(sW, sW + AD_GMAX * 64, wsb2, ctx2, kvh, heads, kv_heads, max_parts,
                                 out + (size_t)bz2 * out_row_stride, 0, 0, nparts2);
    }
}

extern int ze_gemv_knobs[24];

// (measurements: ze_tune knob 16 = step | shift << 4 overrides the grid rotation of the pipelined kernel; 0 = the shipped 3 | 2 << 4)
static int xrot_knob() { return ze_gemv_knobs[16] > 0 ? ze_gemv_knobs[16] : (3 | (2 << 4)); }

void ze_launch_attn_decode_stream(const bf16_t* q, int q_row_stride, const bf16_t* kcache, const bf16_t* vcache,
                                  size_t cache_seq_stride, bf16_t* out, int out_row_stride, const ze_seq_dev* st,
                                  const int* seq_ids, int n, int heads, int kv_heads, int max_ctx, float scale,
                                  float* ws_partial, int max_parts, unsigned* tickets, hipStream_t s, int chunk, int per_wave,
                                  const int* prefix, const int* mate, int long_parts, int use_split) {
    const float sl = scale * 1.4426950408889634f;
    const size_t lds = AB_STAGES * AB_STAGE;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attn_decode_stream<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    if (per_wave) {  // 192-key parts whatever the context: max_parts here = ceil(max_ctx / 192)
        const int wparts = (max_ctx + AW_PART - 1) / AW_PART;
        // per_wave = the parts the LONGEST chain of this batch can have (the caller's host-side context lengths): workgroups
        // for parts no chain has would only look their chain up and leave -- 22 parts instead of 11 per chain and kv head
        // (max_ctx 4096 against 2048, contexts of ~1100) cost the launch 9 %
        // (knob 8 = 3, for A/B runs: every part of max_ctx in the grid, no rotation)
        const bool plain = ze_gemv_knobs[8] == 3;
        const int gparts = plain ? wparts : std::min(wparts, std::max(1, per_wave));
        // Round 4: 384-key parts on the pipelined form (k_attn_decode_wave_long<8, 6>: 4.4 -> 4.7-4.9 TB/s on independent chains at
        // 410-768 of them, 5.5 -> 6.1-6.4 with the stream's shared prefixes; 320-key parts 4.70 / 5.92 at 410, 512-key 4.54 / 5.41,
        // 256-key 4.51 / 5.49).  knob 8 = 4: the 192-key kernel (A/B runs and the agreement test); 3: every part of max_ctx in the grid, no rotation
        if (ze_gemv_knobs[8] != 4) {
            constexpr int rounds = AW_LONG_ROUNDS;
            const int lparts = (max_ctx + rounds * AW_TOK - 1) / (rounds * AW_TOK);
            // (the split adds at most one part to a chain: with it on the grid covers lparts + 1, or the caller's exact count; the
            //  partial buffer holds wparts = ceil(max_ctx / 192) slots per chain: the split is only honoured when lparts + 1 fit)
            // Round 6, BUILT, MEASURED, NOT SHIPPED (knob 23 = 2: split rows, 3: split rows + paired prefix parts; 0 = parts of
            // the whole context as in round 5).  Same box, 490 chains of ten per tile on 800-1440 rows: 95.8 us as shipped, 97.2 with the
            // split (a chain of 1127 rows has four parts instead of three), 100.5 with the pairing on top -- the prefix rows a pair no longer
            // loads twice were L2 hits already (0.856 x algorithmic bytes at the HBM interface), while the leader's workgroup now publishes two
            // sets of partials, draws two tickets and may merge two chains; the stream does not move (85.9-87.0 in every form).
            const int split_on = (use_split && (ze_gemv_knobs[23] == 2 || ze_gemv_knobs[23] == 3) && lparts + 1 <= wparts) ? 1 : 0;
            int lg = plain ? lparts : std::min(lparts, std::max(1, (per_wave * AW_PART + rounds * AW_TOK - 1) / (rounds * AW_TOK)));
            if (split_on) lg = (plain || long_parts <= 0) ? std::min(lparts + 1, lg + 1) : std::min(lparts + 1, long_parts);
            if (split_on)
                k_attn_decode_wave_long<8, rounds, true><<<dim3(kv_heads * lg, n), 256, 4 * 3 * AW_VSTAGE, s>>>(
                    q, q_row_stride, kcache, vcache, cache_seq_stride, st, seq_ids, heads, kv_heads, max_ctx, sl, ws_partial, wparts, tickets, out,
                    out_row_stride, plain ? 0 : xrot_knob(), prefix, ze_gemv_knobs[23] == 3 ? mate : nullptr, 1);
            else
                k_attn_decode_wave_long<8, rounds, false><<<dim3(kv_heads * lg, n), 256, 4 * 3 * AW_VSTAGE, s>>>(
                    q, q_row_stride, kcache, vcache, cache_seq_stride, st, seq_ids, heads, kv_heads, max_ctx, sl, ws_partial, wparts, tickets, out,
                    out_row_stride, plain ? 0 : xrot_knob(), prefix, nullptr, 0);
            return;
        }
        k_attn_decode_wave<8><<<dim3(kv_heads * gparts, n), 256, 4 * AW_VSTAGES * AW_VSTAGE, s>>>(
            q, q_row_stride, kcache, vcache, cache_seq_stride, st, seq_ids, heads, kv_heads, max_ctx, sl, ws_partial, wparts, tickets,
            out, out_row_stride, plain ? 0 : (3 | (2 << 4)), prefix);
    }
    else
        k_attn_decode_stream<8><<<dim3(kv_heads * max_parts, n), 256, lds, s>>>(q, q_row_stride, kcache, vcache, cache_seq_stride, st,
                                                                              seq_ids, heads, kv_heads, max_ctx, sl, ws_partial,
                                                                              max_parts, tickets, out, out_row_stride, chunk);
}
