// Attention kernels.
//  * k_flash_attn<D, CAUSAL>: varlen flash attention on MFMA 16x16x32 bf16 for the ViT (D = 80, window / full
//    segments, non-causal) and for the LLM prefill (D = 128, causal GQA; K19).  Tiles come from a
//    host-built list so one launch covers ragged segments.
//  (The one-token decode attention lives in ze_attn_decode.hip / ze_attn_decode.h.)
// Replaces: SDPA / eager_attention_forward as called at HF:models/qwen2_5_vl/modeling_qwen2_5_vl.py:225-291
// (vision) and :670-689 (text); softmax in fp32, P rounded to bf16 for the PV MFMA (as eager does, :202).
#include "ze_kernels.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define FA_BQ 64
#ifndef ZE_FA_CAUSAL_BKV
// keys per iteration of the causal (prefill) kernel.  128 halves the barriers of a long single-chain prefill but its 128 KB
// of LDS leave ONE workgroup per CU; 64 (two per CU) wins wherever the grid is many rounds: batched prefill of the 64-slot
// question stream 1874 -> 1790 ms (33.4 against 32.8 questions/s), single chain 17.75 -> 18.06 ms per question.  One value
// for every caller: a chain's prefill must not depend on what shares the pass (prefix reuse is bit-exact).
#define ZE_FA_CAUSAL_BKV 64
#endif
#define FA_BK 64

typedef short v4s __attribute__((ext_vector_type(4)));
typedef unsigned int fa_u32x4 __attribute__((ext_vector_type(4)));

// two f32 -> packed bf16 (round to nearest even) in ONE instruction (v_cvt_pk_bf16_f32); the software conversion costs ~6 VALU
// ops per element and the probabilities of a tile are 32 elements per lane.  A vector conversion, NOT inline asm (round 5): an
// MFMA reading a VALU result needs two wait states on gfx950 and hipcc cannot count them from a VALU write hidden in an asm
// statement (ze_attn_decode.h: ad_pack_bf16 -- the root cause of round 4's mis-scheduled decode-attention instantiations, and
// the likely one of the D = 80 flash instantiation that came out wrong under the SLP vectoriser and under waves_per_eu(2)).
__device__ __forceinline__ uint32_t fa_pack_bf16(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) float fa_f32x2_;
    typedef __attribute__((ext_vector_type(2))) __bf16 fa_bf16x2_;
    const fa_bf16x2_ b = __builtin_convertvector(fa_f32x2_{lo, hi}, fa_bf16x2_);
    uint32_t r;
    __builtin_memcpy(&r, &b, 4);
    return r;
}

// LDS image of a [64 keys][128 d] bf16 tile with 256-byte rows: byte offset of 16-B chunk ch (0..15) of row `row`.
// The XOR serves both the ds_read_b128 row reads (K as an MFMA operand) and the ds_read_b64_tr_b16 transposed reads
// (V^T as an MFMA operand) -- cdna_hip_programming.md T10, image (b).
__device__ __forceinline__ int fa_off(int row, int ch) { return 256 * row + 16 * (ch ^ ze_kv_swz(row)); }

// Flash attention with the SWAPPED product: S^T = K Q^T, O^T = V^T P^T (T12 of the guide).
//   * a lane of the S^T accumulator holds one query (column fr) and 4 keys per 16-key tile, so the softmax statistics
//     of a query are lane-local plus two shuffles (xor 16, 32), the rescale factor is ONE scalar per lane, and the
//     probabilities go from the accumulator straight into the B operand of the PV product -- no LDS round trip for
//     P.  The 8 keys a lane contributes per 32-key step are (fq*4 .. +3) and (16 + fq*4 .. +3): a permutation of the
//     contraction index, applied identically to the V^T operand.
//   * V stays row-major in LDS (16-B staging writes); V^T fragments come from ds_read_b64_tr_b16 (per 16-lane group a
//     4-key x 16-d block, delivered column-wise: probe tools/probes/tr_read_probe.hip), two reads per fragment at
//     exactly the key rows of the permutation above.  The previous kernel transposed V with 2-byte LDS writes and
//     round-tripped P through LDS: 98.7 us per prefill layer at 802 tokens, 1.5 % MFMA utilisation.
//   * Q fragments live in registers for the whole kernel; K/V tiles are double-buffered in LDS with split staging
//     (global loads of tile t+1 are issued before the products of tile t, written to LDS after).
// D = 128 (LLM, causal GQA) and D = 80 (ViT; rows padded to 128 in LDS, the pad chunks are zero).
// BKV: keys per iteration (64, or 128: half the barriers and load-latency exposures of a long causal prefill)
// QT: 16-query tiles per wave (1: 64 queries per workgroup; 2: 128 -- every K and V^T fragment read from LDS feeds TWO MFMAs:
// the kernel is bound by its LDS fragment reads, one per MFMA with QT = 1).  A query's arithmetic does not depend on QT: the
// same key tiles in the same order, the same online softmax -- bit-identical rows.
// DMA (D = 128 only: a cache row is one whole 256-byte LDS row): K / V tiles go from global memory straight into the LDS image
// by `global_load_lds_dwordx4` -- the swizzle on the SOURCE chunk, as the decode kernels stage them -- instead of through 32
// staging registers and a ds_write phase; keys past the end re-read the last valid row (finite; masked out of the softmax).
template <int D, int CAUSAL, int BKV, int QT = 1, bool DMA = false>
// amdgpu_waves_per_eu(2): two workgroups per CU are what hides the S phase of one behind the products of the other; with the two
// wave-uniform branches below hipcc's own choice was 218 VGPRs + 96 AGPRs (one workgroup per CU) -- held to 256 it allocates 234 plain
// VGPRs, no AGPR copies around the rescale, no scratch.  (Round 3 saw wrong rows from this attribute on the D = 80 instantiation: that
// was the asm conversion's hazard, fa_pack_bf16 above; tools/check_mfma_hazards.py and test_attention_every_instantiation hold it now.)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) k_flash_attn(const bf16_t* __restrict__ q, int q_rs, int q_hs,
                                                    const bf16_t* __restrict__ k, int k_rs, int k_hs,
                                                    const bf16_t* __restrict__ v, int v_rs, int v_hs,
                                                    bf16_t* __restrict__ o, int o_rs, int o_hs,
                                                    const int4* __restrict__ tiles, int group, float scale_log2e,
                                                    int q_pos_offset, const int* __restrict__ tile_aux,
                                                    size_t kv_seq_stride, ze_fa_rope rope) {
    constexpr int KS = (D + 31) / 32;  // 32-deep steps of the QK^T contraction (D = 80: 3, the third half zero)
    constexpr int DCH = D / 8;         // real 16-B chunks per row
    constexpr int NV = D / 16;         // d-tiles of the output
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];  // 2 x (K tile + V tile), 256 B per key each
    constexpr int TILE_B = BKV * 256, NKT = BKV / 16, RP = BKV / 64;  // bytes per tile image, key tiles, staging passes

    const int4 tile = tiles[blockIdx.x];
    const int q0 = tile.x, q1 = tile.y, kv0 = tile.z, kv1 = tile.w;
    if (tile_aux) {  // batched prefill: this tile's chain has its own KV cache and position offset
        k += (size_t)tile_aux[2 * blockIdx.x] * kv_seq_stride;
        v += (size_t)tile_aux[2 * blockIdx.x] * kv_seq_stride;
        q_pos_offset = tile_aux[2 * blockIdx.x + 1];
    }
    const int head = blockIdx.y, kvh = head / group;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    int qi[QT];  // the queries of this lane (columns of S^T / O^T), one per query tile of the wave
#pragma unroll
    for (int t = 0; t < QT; ++t) qi[t] = q0 + (wid * QT + t) * 16 + fr;

    // Q^T fragments: lane (n = fr, k-group fq) holds d = ks*32 + fq*8 .. +7 of its query
    bf16x8 qf[QT][KS];
#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            uint4 val = make_uint4(0, 0, 0, 0);
            const int ch = ks * 4 + fq;
            if (qi[t] < q1 && ch < DCH) val = *reinterpret_cast<const uint4*>(q + (size_t)qi[t] * q_rs + (size_t)head * q_hs + ch * 8);
            qf[t][ks] = *reinterpret_cast<const bf16x8*>(&val);
        }

    f32x4 oacc[QT][NV];
    float m_run[QT], l_run[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
#pragma unroll
        for (int j = 0; j < NV; ++j) oacc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        m_run[t] = -INFINITY;
        l_run[t] = 0.f;
    }
    int kv_hi = kv1;
    if (CAUSAL) kv_hi = min(kv1, q1 + q_pos_offset);  // keys beyond the last query position are never visible
    const int ntile = (kv_hi - kv0 + BKV - 1) / BKV;

    // staging: thread -> (row = tid >> 2, chunks (tid & 3) * 4 .. +3) of the 64 x 16-chunk tile
    const int srow = tid >> 2, sch = (tid & 3) * 4;
    fa_u32x4 rk[RP][4], rv[RP][4];  // native vectors: a HIP uint4 struct behind the selects below went to scratch
    auto stage_load = [&](int kt) {
        // Branch-free: a per-lane "in range ? load : 0" makes hipcc branch around every load and wait vmcnt(0) inside
        // each branch -- 8-16 dependent L2 round trips per tile (the kernel ran 52 us per prefill layer that way).
        // Out-of-range keys / pad chunks re-read a valid address and are zeroed by a select at stage_write time --
        // NOT here: a select right behind the loads makes the first MFMA of the tile wait for all of them.
#pragma unroll
        for (int rp = 0; rp < RP; ++rp) {
            const int keyc = min(kt + rp * 64 + srow, kv_hi - 1);
            const bf16_t* kp = k + (size_t)keyc * k_rs + (size_t)kvh * k_hs;
            const bf16_t* vp = v + (size_t)keyc * v_rs + (size_t)kvh * v_hs;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ch = min(sch + i, DCH - 1);
                rk[rp][i] = *reinterpret_cast<const fa_u32x4*>(kp + ch * 8);
                rv[rp][i] = *reinterpret_cast<const fa_u32x4*>(vp + ch * 8);
            }
        }
    };
    auto stage_write = [&](int buf, int kt) {  // kt: first key of the tile that stage_load fetched
        uint8_t* kb = smem + buf * 2 * TILE_B;
#pragma unroll
        for (int rp = 0; rp < RP; ++rp) {
            const bool ok = kt + rp * 64 + srow < kv_hi;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned m = (ok && (sch + i < DCH)) ? 0xffffffffu : 0u;  // zero out-of-range keys / pad chunks
                *reinterpret_cast<fa_u32x4*>(kb + fa_off(rp * 64 + srow, sch + i)) = rk[rp][i] & m;
                *reinterpret_cast<fa_u32x4*>(kb + TILE_B + fa_off(rp * 64 + srow, sch + i)) = rv[rp][i] & m;
            }
        }
    };
    static_assert(!DMA || (D == 128 && BKV == 64), "the LDS-DMA staging moves whole 256-byte rows of 64-key tiles");
    const unsigned smem_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) uint8_t*)smem);
    const int wid_u = __builtin_amdgcn_readfirstlane(wid);
    // this wave's 4 of the 16 one-KiB pieces of the K image and of the V image (piece p = rows 4p .. 4p + 3)
    auto stage_dma = [&](int buf, int kt) {
        const unsigned kimg = smem_lds + (unsigned)buf * 2u * TILE_B, vimg = kimg + TILE_B;
        const int r4 = lane >> 4, pos = lane & 15;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int pc = wid_u * 4 + g;
            const int row = 4 * pc + r4;
            const int ch = pos ^ ze_kv_swz(row);
            const int keyc = min(kt + row, kv_hi - 1);
            const bf16_t* ksrc = k + (size_t)keyc * k_rs + (size_t)kvh * k_hs + ch * 8;
            const bf16_t* vsrc = v + (size_t)keyc * v_rs + (size_t)kvh * v_hs + ch * 8;
            unsigned keep;
            asm volatile(
                "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
                : "=&s"(keep)
                : "v"(ksrc), "v"(vsrc), "s"(kimg + (unsigned)pc * 1024u), "s"(vimg + (unsigned)pc * 1024u)
                : "memory");
        }
    };
    if (ntile > 0) {
        if constexpr (DMA) stage_dma(0, kv0);
        else stage_load(kv0);
    }
    if constexpr (D == 128) {
        // (behind the first tile's loads, so that its two dependent lookups -- position, then cos / sin -- ride on their latency)
        // M-RoPE of Q here instead of in k_mrope_kv_vec (which then writes K and V only): fragments ks and ks + 2 of a lane are the two
        // halves of eight rotate_half pairs j = ks * 32 + fq * 8 .. + 7; one position axis per such chunk (ze_mrope_vec_ok)
        if (rope.cosT) {
#pragma unroll
            for (int t = 0; t < QT; ++t)
                if (qi[t] < q1) {
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const int j = ks * 32 + fq * 8;
                        const int axis = j < rope.s0 ? 0 : (j < rope.s01 ? 1 : 2);
                        const int pos = rope.pos3[axis * rope.T + qi[t]];
                        const uint4 c4 = *reinterpret_cast<const uint4*>(rope.cosT + (size_t)pos * 64 + j);
                        const uint4 s4 = *reinterpret_cast<const uint4*>(rope.sinT + (size_t)pos * 64 + j);
                        uint4 o1, o2;
                        rope8(*reinterpret_cast<const uint4*>(&qf[t][ks]), *reinterpret_cast<const uint4*>(&qf[t][ks + 2]), c4, s4, o1, o2);
                        qf[t][ks] = *reinterpret_cast<const bf16x8*>(&o1);
                        qf[t][ks + 2] = *reinterpret_cast<const bf16x8*>(&o2);
                    }
                }
        }
    }

    if (ntile > 0) {
        if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else stage_write(0, kv0);
    }
    __syncthreads();

    for (int t = 0; t < ntile; ++t) {
        const int kt = kv0 + t * BKV;
        const uint8_t* kb = smem + (t & 1) * 2 * TILE_B;
        const uint8_t* vb = kb + TILE_B;
        if (t + 1 < ntile) {
            if constexpr (DMA) stage_dma((t + 1) & 1, kt + BKV);  // lands during the products of tile t
            else stage_load(kt + BKV);
        }
        // ---- S^T = K Q^T : NKT key tiles x KS steps; one K fragment read feeds the QT query tiles
        f32x4 sacc[QT][NKT];
#pragma unroll
        for (int u = 0; u < QT; ++u)
#pragma unroll
            for (int n = 0; n < NKT; ++n) sacc[u][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int n = 0; n < NKT; ++n) {
                const uint4 ka = *reinterpret_cast<const uint4*>(kb + fa_off(n * 16 + fr, ks * 4 + fq));
#pragma unroll
                for (int u = 0; u < QT; ++u)
                    sacc[u][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&ka), qf[u][ks], sacc[u][n],
                                                                        0, 0, 0);
            }
        }
        // ---- mask + online softmax per query: keys kt + n*16 + fq*4 + r
        // (The library is compiled with -fno-slp-vectorize: with hipcc's SLP vectoriser on, the D = 80 instantiations
        //  produced wrong rows -- first seen behind a fast path that skipped this compare + select pair on fully
        //  visible tiles, then in the plain D = 80 causal kernel; tools/check_attn.py and
        //  test_attention_every_instantiation pin every instantiation.  The fast path was worth 10 % of the
        //  kernel = 0.2 ms per question and is not worth re-validating.)
        uint32_t pw[QT][NKT][2];  // the probabilities, packed to bf16 pairs (keys fq*4 + 0,1 | 2,3 of key tile n)
#pragma unroll
        for (int u = 0; u < QT; ++u) {
            float p[NKT][4];
            float mx = -INFINITY;
            // The S phase is what bounds this kernel (11 VALU instructions per MFMA, SQ counters of round 5): a key tile every
            // query of the wave's tile sees whole -- all but the last one or two of a causal row of tiles, all but the last of a
            // ViT segment -- skips the two compares and selects per score.  Wave-uniform condition, same values either way
            // (a visible score is sacc * scale in both branches).  ze_tune-free: -DZE_FA_NO_FAST_PATH builds without it.
            bool whole = kt + BKV <= kv_hi;
            if (CAUSAL) whole = whole && (kt + BKV - 1 <= q0 + (wid * QT + u) * 16 + q_pos_offset);
#ifdef ZE_FA_NO_FAST_PATH
            whole = false;
#endif
            if (__builtin_amdgcn_readfirstlane((int)whole)) {
#pragma unroll
                for (int n = 0; n < NKT; ++n)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float sv = sacc[u][n][r] * scale_log2e;
                        p[n][r] = sv;
                        mx = fmaxf(mx, sv);
                    }
            } else {
#pragma unroll
                for (int n = 0; n < NKT; ++n)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int kj = kt + n * 16 + fq * 4 + r;
                        bool ok = kj < kv_hi;
                        if (CAUSAL) ok = ok && (kj <= qi[u] + q_pos_offset);
                        const float sv = ok ? sacc[u][n][r] * scale_log2e : -INFINITY;
                        p[n][r] = sv;
                        mx = fmaxf(mx, sv);
                    }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run[u], mx);
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = exp2f(m_run[u] - m_use);  // m_run = -inf -> 0
            float rs = 0.f;
#pragma unroll
            for (int n = 0; n < NKT; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __builtin_amdgcn_exp2f(p[n][r] - m_use);  // raw v_exp_f32: arguments <= 0, results in [0, 1]
                    p[n][r] = e;
                    rs += e;
                }
            rs += __shfl_xor(rs, 16, 64);
            rs += __shfl_xor(rs, 32, 64);
            l_run[u] = l_run[u] * alpha + rs;
            m_run[u] = m_new;
            // once a query's running maximum has settled, alpha is exactly 1: the 4 * NV multiplies are skipped when it is for every
            // lane of the wave (x * 1.0f == x: the same bits)
            if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    oacc[u][j][0] *= alpha;
                    oacc[u][j][1] *= alpha;
                    oacc[u][j][2] *= alpha;
                    oacc[u][j][3] *= alpha;
                }
            }
#pragma unroll
            for (int n = 0; n < NKT; ++n) {
                pw[u][n][0] = fa_pack_bf16(p[n][0], p[n][1]);
                pw[u][n][1] = fa_pack_bf16(p[n][2], p[n][3]);
            }
        }
        // ---- O^T += V^T P^T : per 32-key step the lane's 8 keys are (ks*32 + fq*4 + 0..3) and (ks*32 + 16 + fq*4 + 0..3);
        // one V^T fragment (two transposed reads) feeds the QT query tiles
#pragma unroll
        for (int ks = 0; ks < BKV / 32; ++ks) {
            bf16x8 pb[QT];
#pragma unroll
            for (int u = 0; u < QT; ++u) {
                const uint4 pq = make_uint4(pw[u][2 * ks][0], pw[u][2 * ks][1], pw[u][2 * ks + 1][0], pw[u][2 * ks + 1][1]);
                pb[u] = *reinterpret_cast<const bf16x8*>(&pq);
            }
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                // block of 4 keys x 16 d: lane 4q+p of the 16-lane group addresses key row (r0 + q), d = j*16 + 4p .. +3
                const int l16 = lane & 15, tq = l16 >> 2, tp = l16 & 3;
                const int r0a = ks * 32 + fq * 4, r0b = r0a + 16;
                const int ch = j * 2 + (tp >> 1), half = 8 * (tp & 1);
                const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) v4s*)(vb + fa_off(r0a + tq, ch) + half));
                const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) v4s*)(vb + fa_off(r0b + tq, ch) + half));
                typedef short v8s __attribute__((ext_vector_type(8)));
                const v8s va = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                for (int u = 0; u < QT; ++u)
                    oacc[u][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&va), pb[u], oacc[u][j], 0, 0, 0);
            }
        }
        if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of tile t + 1 have landed
        else if (t + 1 < ntile) stage_write((t + 1) & 1, kt + BKV);
        __syncthreads();
    }
    // ---- normalise and store: lane holds O[qi][j*16 + fq*4 .. +3]
#pragma unroll
    for (int u = 0; u < QT; ++u)
        if (qi[u] < q1) {
            const float inv = l_run[u] > 0.f ? 1.0f / l_run[u] : 0.f;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                uint2 w;
                w.x = pack_bf16x2(oacc[u][j][0] * inv, oacc[u][j][1] * inv);
                w.y = pack_bf16x2(oacc[u][j][2] * inv, oacc[u][j][3] * inv);
                *reinterpret_cast<uint2*>(o + (size_t)qi[u] * o_rs + (size_t)head * o_hs + j * 16 + fq * 4) = w;
            }
        }
}

void ze_launch_flash_attn(int D, int causal, const bf16_t* q, int q_rs, int q_hs, const bf16_t* k, int k_rs,
                          int k_hs, const bf16_t* v, int v_rs, int v_hs, bf16_t* o, int o_rs, int o_hs,
                          const int4* tiles, int n_tiles, int heads, int group, float scale, int q_pos_offset,
                          hipStream_t s, const int* tile_aux, size_t kv_seq_stride, int q_tile, int max_kv, ze_fa_rope rope) {
    if (n_tiles == 0) return;
    const float sl = scale * 1.4426950408889634f;
    dim3 grid(n_tiles, heads);
    // max_kv: the longest key range of any tile of the list, 0 = unknown.  A list whose tiles all fit ONE key tile (the ViT's
    // window blocks: segments of at most 64 tokens) never touches the second K / V buffer: half the LDS, so three workgroups
    // per CU instead of two -- these launches are thousands of short-lived workgroups bound by their own latency
#define FA_LAUNCH_(DD, CC, QQ, DM)                                                                                 \
    do {                                                                                                          \
        constexpr int BKV_ = (CC) ? ZE_FA_CAUSAL_BKV : 64;                                                        \
        constexpr int LDS_ = 4 * BKV_ * 256;                                                                      \
        static bool attr_set = false;                                                                             \
        if (!attr_set) {                                                                                          \
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_flash_attn<DD, CC, BKV_, QQ, DM>),               \
                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS_);                                \
            attr_set = true;                                                                                      \
        }                                                                                                         \
        const int lds_ = (max_kv > 0 && max_kv <= BKV_) ? LDS_ / 2 : LDS_;                                        \
        hipLaunchKernelGGL((k_flash_attn<DD, CC, BKV_, QQ, DM>), grid, dim3(256), lds_, s, q, q_rs, q_hs, k, k_rs, k_hs, v, v_rs, \
                           v_hs, o, o_rs, o_hs, tiles, group, sl, q_pos_offset, tile_aux, kv_seq_stride, rope);    \
    } while (0)
#define FA_LAUNCH(DD, CC, QQ) FA_LAUNCH_(DD, CC, QQ, false)
    // D = 128 causal (the prefill: K / V rows are whole 256-byte cache rows, 16-byte aligned): the LDS-DMA staging form;
    // ze_tune knob 1 = 7 keeps the register-staged form for A/B runs and the bit-equality test
    extern int ze_gemv_knobs[24];
    if (D == 128 && causal && ZE_FA_CAUSAL_BKV == 64 && ze_gemv_knobs[1] != 7 && k_rs % 8 == 0 && v_rs % 8 == 0 && k_hs % 8 == 0 &&
        v_hs % 8 == 0 && ((size_t)k % 16) == 0 && ((size_t)v % 16) == 0 && kv_seq_stride % 8 == 0) {
        if (q_tile > 64) FA_LAUNCH_(128, 1, 2, true);
        else FA_LAUNCH_(128, 1, 1, true);
        return;
    }
    // q_tile: the query rows a tile of the caller's list spans at most -- 64 (one 16-query tile per wave) or 128 (two)
    if (q_tile > 64) {
        if (D == 80) {
            if (causal) FA_LAUNCH(80, 1, 2); else FA_LAUNCH(80, 0, 2);
        } else {
            if (causal) FA_LAUNCH(128, 1, 2); else FA_LAUNCH(128, 0, 2);
        }
        return;
    }
    if (D == 80) {
        if (causal) FA_LAUNCH(80, 1, 1); else FA_LAUNCH(80, 0, 1);
    } else {
        if (causal) FA_LAUNCH(128, 1, 1); else FA_LAUNCH(128, 0, 1);
    }
#undef FA_LAUNCH
#undef FA_LAUNCH_
}
