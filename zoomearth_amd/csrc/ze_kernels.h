// Launcher declarations for the HIP kernels of libzoomearth_hip.so (internal C++ interface).
#pragma once
#include "ze_common.h"

// GEMM epilogues
enum { ZE_EPI_NONE = 0, ZE_EPI_GELU = 1, ZE_EPI_RESIDUAL = 2, ZE_EPI_SWIGLU = 3, ZE_EPI_F32 = 4, ZE_EPI_QKV_ROPE = 5 };
// decode GEMV epilogues
enum { ZE_GV_QKV_ROPE = 0, ZE_GV_RESIDUAL = 1, ZE_GV_SWIGLU = 2, ZE_GV_LOGITS = 3, ZE_GV_PLAIN = 4 };

// Device-resident state of one question chain; decode kernels read it so that a captured hipGraph of one
// decode step can be replayed without host-side argument changes.
// XOR swizzle of the 16-B slots of a 256-byte LDS row, by row & 15 -- the K / V tile images of every attention kernel.  gfx950 serves
// a wave's ds_read_b128 in four groups of 16 NON-contiguous lanes ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, + 32) and its
// ds_read_b64_tr_b16 in two halves of 32 lanes; banks = (byte / 4) mod 64 for both (cdna_hip_programming.md 2, MI355X_MICROARCH.md
// LDS).  The K fragment read (lane = row l & 15, slot ks * 4 + (l >> 4)) therefore puts rows {0-3, 12-15} at slot c and rows {4-11}
// at slot c ^ 1 into ONE group, and the transposed V read puts rows 8h .. 8h + 7, two slots each, into one half.  Conflict-free for
// both: rows 0-7 take the even slots 2r, rows 8-15 the odd slots of the pairs of rows (r ^ 4) -- {4-11} and {0-3, 12-15} each own four
// whole slot pairs, and eight consecutive rows eight different pairs.  (Rounds 2-5 used ((row & 3) << 2) | ((row >> 2) & 3), laid out
// for contiguous 16-lane groups: SQ_LDS_BANK_CONFLICT = 44 % of the LDS cycles of the prefill flash kernel, 20 % of the decode
// attention's -- every fragment read two-way.)
#if defined(__HIPCC__)
__device__ __forceinline__ int ze_kv_swz(int row) {
#ifdef ZE_KV_SWZ_OLD  // (A/B builds only: tools/probes/swz_ab.sh)
    return ((row & 3) << 2) | ((row >> 2) & 3);
#else
    const int hi = (row >> 3) & 1;
    return ((((row & 7) ^ (hi << 2)) << 1) | hi);
#endif
}
#endif

#if defined(__HIPCC__)
// eight rotate_half pairs at once: (x1[k], x2[k]) = elements (j + k, j + half + k), cos / sin of the same eight j.  The
// arithmetic per element is the scalar kernels': bf16(bf16(x1 c) + bf16(-x2 s)), bf16(bf16(x2 c) + bf16(x1 s)).
__device__ __forceinline__ void rope8(const uint4& a, const uint4& b, const uint4& c4, const uint4& s4, uint4& o1, uint4& o2) {
    const uint32_t* pa = reinterpret_cast<const uint32_t*>(&a);
    const uint32_t* pb = reinterpret_cast<const uint32_t*>(&b);
    const uint32_t* pc = reinterpret_cast<const uint32_t*>(&c4);
    const uint32_t* ps = reinterpret_cast<const uint32_t*>(&s4);
    uint32_t r1[4], r2[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint32_t lo1, hi1, lo2, hi2;
        {
            const float x1 = __uint_as_float(pa[q] << 16), x2 = __uint_as_float(pb[q] << 16);
            const float c = __uint_as_float(pc[q] << 16), sn = __uint_as_float(ps[q] << 16);
            lo1 = f32_to_bf16(bf16_round(x1 * c) + bf16_round(-x2 * sn));
            lo2 = f32_to_bf16(bf16_round(x2 * c) + bf16_round(x1 * sn));
        }
        {
            const float x1 = __uint_as_float(pa[q] & 0xffff0000u), x2 = __uint_as_float(pb[q] & 0xffff0000u);
            const float c = __uint_as_float(pc[q] & 0xffff0000u), sn = __uint_as_float(ps[q] & 0xffff0000u);
            hi1 = f32_to_bf16(bf16_round(x1 * c) + bf16_round(-x2 * sn));
            hi2 = f32_to_bf16(bf16_round(x2 * c) + bf16_round(x1 * sn));
        }
        r1[q] = lo1 | (hi1 << 16);
        r2[q] = lo2 | (hi2 << 16);
    }
    o1 = make_uint4(r1[0], r1[1], r1[2], r1[3]);
    o2 = make_uint4(r2[0], r2[1], r2[2], r2[3]);
}

#endif

struct ze_seq_dev {
    int32_t ctx;        // tokens in the KV cache
    int32_t rope_delta; // position of the next token = ctx + rope_delta
    int32_t token;      // last sampled token (input of the next decode step)
    int32_t finished;   // 1 after an EOS was emitted (subsequent tokens are pad)
    int32_t n_gen;      // tokens written to out_tokens so far
    int32_t max_gen;    // capacity of out_tokens
    int32_t stream;     // sampling stream of this chain: its row in the generate call (0 for single-chain calls)
    int32_t split;      // round 6: the chain's split row -- the end of its first image block (0: none) -- a property of the chain's own
                        // tokens, set when they are prefilled / copied (pushed with the rest of the state, constant while the chain
                        // decodes); the decode attention cuts its parts there (ze_attn_batch.hip).  (Until round 3 this word held
                        // the shared-prefix hint, which lives in ze_engine::pfx_dev, written by the decode stream alone.)
};

// ---- front-end
void ze_launch_resize_h(const uint8_t* src, int src_h, int src_w, int bx0, int by0, int box_h, uint8_t* dst,
                        int out_w, const int* xmin, const int* xcnt, const int* kk, int ksize, int max_span,
                        hipStream_t s, int inside = 0);  // inside: the box lies inside the image (16-byte staging path)
void ze_launch_resize_v(const uint8_t* src, int src_h, int src_w, int bx0, int by0, int row_bytes, uint8_t* dst,
                        int out_h, const int* ymin, const int* ycnt, const int* kk, int ksize, int boxed,
                        hipStream_t s);
void ze_launch_crop(const uint8_t* src, int src_h, int src_w, int bx0, int by0, uint8_t* dst, int out_h, int out_w,
                    hipStream_t s);
void ze_launch_patchify(const uint8_t* img, int h, int w, const float* lut, float* out, int P, int M, int T, int C,
                        hipStream_t s);

// ---- elementwise
void ze_launch_pack_rows(const void* src, int dtype, int row0, int nrows, int cols, bf16_t* dst, int ld, int mode,
                         int offset, hipStream_t s);
void ze_launch_fill_rows(uint64_t seed, float c_scale, float base, int rows, int cols, bf16_t* dst, int ld, int mode,
                         int offset, hipStream_t s);
// split-K workspace of the calling engine (fp32 slabs + per-tile tickets); empty = never split
struct ze_gemm_ws {
    float* slab = nullptr;
    size_t slab_floats = 0;
    unsigned* tickets = nullptr;
    int ticket_cap = 0;
};
// act8 (FP8 activations, ze_set_fp8_activations): 1 = the output row is replaced by its per-row E4M3 quantisation,
// written back as bf16 (q * 2^k, exact); 2 (frag only, rows <= 64) = FP8 fragment-major bytes into y8 + the row scales
// into yscale (the A operand of the fp8 MFMA kernels of the batched step)
void ze_launch_rmsnorm(const bf16_t* x, int ldx, const bf16_t* w, bf16_t* y, int ldy, int rows, int cols, float eps,
                       hipStream_t s, int frag = 0, int act8 = 0, uint8_t* y8 = nullptr, float* yscale = nullptr);
void ze_launch_pack_fragments(const bf16_t* W, int ldw, int n, int k, bf16_t* Wf, hipStream_t s, int rope_dim = 0);
// One-shot skinny GEMMs of the batched decode step (ze_gemm_oneshot.hip): sixteen waves per workgroup, one memory round
// trip per launch.  epi: ZE_EPI_NONE (+bias) or ZE_EPI_RESIDUAL; M <= 64, N % 16 == 0, K % 32 == 0.
void ze_launch_gemm_oneshot(int epi, const bf16_t* Xf, const bf16_t* Wf, const bf16_t* bias, const bf16_t* R, int ldr,
                            bf16_t* C, int ldc, int M, int N, int K, hipStream_t s, const float* wscale = nullptr,
                            const float* ascale = nullptr);  // ascale: Xf is FP8 fragments (fp8 x fp8 MFMA)
// qkv projection + M-RoPE + KV append in one launch; Wf_perm = ze_launch_pack_fragments(..., rope_dim = 128); q heads go
// to q_out rows (stride ldq, original column order), k / v rows into the caches at each chain's position.
void ze_launch_qkv_rope_oneshot(const bf16_t* Xf, const bf16_t* Wf_perm, const bf16_t* bias, bf16_t* q_out, int ldq, int M,
                                int K, int heads, int kv_heads, const bf16_t* cosT, const bf16_t* sinT,
                                const ze_seq_dev* st, const int* seq_ids, bf16_t* kcache, bf16_t* vcache,
                                size_t cache_seq_stride, int max_ctx, hipStream_t s, const float* wscale = nullptr,
                                const float* ascale = nullptr);
// block-scaled FP8 GEMM (k_gemm_ring_mx): A8 [M, K] / W8 [N, K] E4M3 bytes with one power-of-two scale per row each; epi
// ZE_EPI_NONE or ZE_EPI_SWIGLU; false when the shape does not qualify (K % 128, leading dimensions % 16)
bool ze_launch_gemm_mx(int epi, const uint8_t* A, int lda, const float* sa, const uint8_t* W, int ldw, const float* sw,
                       const bf16_t* bias, bf16_t* C, int ldc, int M, int N, int K, hipStream_t s);
// batched decode on fragment-major operands (k_gemm_skinny<..., FRAG>): Xf from ze_launch_rmsnorm(frag = 1), Wf from
// ze_launch_pack_fragments; M <= 64, N % 16 == 0, K % 32 == 0, K <= 4096 (no split-K)
// wscale != null: Wf is the FP8 fragment copy (ze_launch_pack_fragments8) and wscale the per-row power-of-two scales
void ze_launch_gemm_frag(int epi, const bf16_t* Xf, const bf16_t* Wf, const bf16_t* bias, const bf16_t* R, int ldr,
                         bf16_t* C, int ldc, int M, int N, int K, hipStream_t s, const float* wscale = nullptr,
                         const float* ascale = nullptr);  // ascale (gate/up at > 32 chains only): Xf is FP8 fragments
// FP8 weights [n, ld8] row-major (ze_launch_quantize_rows) -> fragment-major: fragment (nb, ks) = 64 lanes x 8 B,
// lane = (col % 32) / 8 * 16 + row % 16 holds columns 8 (lane / 16) .. +7 of its row; rope_dim as ze_launch_pack_fragments
void ze_launch_pack_fragments8(const uint8_t* W8, int ld8, int n, int k, uint8_t* Wf8, hipStream_t s, int rope_dim = 0);
void ze_launch_gather_cast_rows(const float* src, int k, const int* perm, bf16_t* dst, int kp, int rows,
                                hipStream_t s);
void ze_launch_vision_rope(bf16_t* qkv, const float* cosT, const float* sinT, int n, int heads, int D, hipStream_t s);
void ze_launch_mrope_kv(bf16_t* qkv, int T, int heads, int kv_heads, int D, const bf16_t* cosT, const bf16_t* sinT,
                        const int* pos3, const int* axis_of, bf16_t* kcache, bf16_t* vcache, int max_ctx, int past, const int* row_aux, size_t cache_seq_stride,
                        hipStream_t s, int q_skip = 0);
// q_skip: Q stays unrotated in qkv -- the prefill flash kernel ropes it as it loads it (ze_fa_rope below; D = 128 and the 16-byte
// form of the M-RoPE kernel, ze_mrope_vec_ok, only)
extern int ze_mrope_vec_ok;
void ze_launch_embed_rows(const int* src, const bf16_t* embed, const bf16_t* image_embeds, bf16_t* out, int T,
                          int hidden, hipStream_t s);
void ze_launch_scatter_rows(const bf16_t* src, int lds_, const int* dst_idx, bf16_t* dst, int ldd, int rows, int cols,
                            hipStream_t s);

// ---- qkv projection of the batched decode step in the row-streaming regime with M-RoPE + KV append as its epilogue
// (ZE_EPI_QKV_ROPE; removes the stand-alone k_rope_kv_batch launch).  The weight rows of every 128-wide head are PERMUTED in
// blocks of 32: [d0 .. d0+15 | 64+d0 .. 64+d0+15] for d0 = 0, 16, 32, 48 (ze_launch_permute_qkv; the bias likewise), so the two
// 16-column MFMA tiles a wave holds side by side are the rotate_half partners (d, d + 64) of the same rows and lanes -- the pairing
// the SwiGLU epilogue uses for gate / up.  Per element the arithmetic is k_rope_kv_batch's on the bf16-rounded projection:
// bf16(bf16(x1 c) + bf16(-x2 s)), bf16(bf16(x2 c) + bf16(x1 s)) with the chain's cos / sin at position ctx + rope_delta; q heads
// go to C (original column order), k / v heads into the caches at the chain's ctx.  Device-resident per layer (one pointer as a
// kernel argument, in the slot of the unused residual pointer).
struct ze_qkv_epi {
    const ze_seq_dev* st;
    const int* seq_ids;
    const bf16_t* cosT;
    const bf16_t* sinT;
    bf16_t* kcache;  // of this layer, chain 0
    bf16_t* vcache;
    size_t cache_seq_stride;
    int max_ctx, heads, kv_heads, pad_;
};
void ze_launch_permute_qkv(const bf16_t* W, int ldw, const bf16_t* bias, int n_heads_total, int K, bf16_t* Wp, bf16_t* bias_p, hipStream_t s);
void ze_launch_gemm_qkv_rope(const bf16_t* A, int lda, const bf16_t* Wp, int ldw, const bf16_t* bias_p, const ze_qkv_epi* dev_args,
                             bf16_t* C, int ldc, int M, int N, int K, hipStream_t s);

// ---- GEMM (prefill / ViT)
// `ws` (round 6): the engine's PREFILL split-K workspace -- with it a long-K projection (K > 4096: the down projection) is summed in
// three K slices on every tile and kernel (ze_gemm.hip: ze_prefill_ksplit); without it (ViT, bare unit ops) K runs in sequence.
void ze_launch_gemm(int epi, const bf16_t* A, int lda, const bf16_t* W, int ldw, const bf16_t* bias, const bf16_t* R,
                    int ldr, bf16_t* C, int ldc, const int* c_rows, int M, int N, int K, hipStream_t s, const ze_gemm_ws& ws = ze_gemm_ws());

// Weight-streaming form for batched decode (few rows): 64x64 tiles, deterministic split-K chosen from (N, K) only.
void ze_launch_gemm_stream(int epi, const bf16_t* A, int lda, const bf16_t* W, int ldw, const bf16_t* bias,
                           const bf16_t* R, int ldr, bf16_t* C, int ldc, int M, int N, int K, const ze_gemm_ws& ws,
                           hipStream_t s);
// split-K workspace: fp32 slabs (>= ksplit * tiles * BM * BN floats) and zero-initialised per-tile tickets
// The projections of the batched decode step in the row-streaming regime (ze_set_decode_regime: engines with more than 64
// chain slots), rows = chains, 1 <= M <= max_seqs.  Every output element is summed in an order fixed by (N, K) alone --
// K in sequence where K <= 4096, eight K slices added in slice order for the long down projection -- whatever tile the
// row count selects: a chain's result does not depend on the batch it shares.
void ze_launch_gemm_wide(int epi, const bf16_t* A, int lda, const bf16_t* W, int ldw, const bf16_t* bias, const bf16_t* R,
                         int ldr, bf16_t* C, int ldc, int M, int N, int K, const ze_gemm_ws& ws, hipStream_t s);

// the eight-phase 256 x 256 kernel directly (ze_launch_gemm picks it for many-round grids); K % 64 == 0, lda / ldw % 8 == 0
void ze_launch_gemm_p8(int epi, const bf16_t* A, int lda, const bf16_t* W, int ldw, const bf16_t* bias, const bf16_t* R, int ldr,
                       bf16_t* C, int ldc, const int* c_rows, int M, int N, int K, hipStream_t s, const ze_gemm_ws& ws = ze_gemm_ws());

// ---- decode GEMV family (batch-1 weight streaming)
struct ze_gemv_args {
    const bf16_t* W;      // [N, ldw] packed weight
    int ldw, N, K;
    const bf16_t* x;      // input vector [K] (bf16)
    const bf16_t* norm_w; // non-null: RMSNorm prologue over x with this weight
    float eps;
    const bf16_t* bias;   // [N] or null
    // epilogue targets
    bf16_t* out_bf16;     // RESIDUAL: hidden stream (in place) ; SWIGLU: act ; PLAIN: out ; QKV: q buffer
    float* out_f32;       // LOGITS
    // QKV_ROPE extras
    const ze_seq_dev* st;
    const bf16_t* cosT;
    const bf16_t* sinT;   // [max_pos, D/2]
    bf16_t* kcache;
    bf16_t* vcache;       // [kv_heads, max_ctx, D] of this layer / chain
    int heads, kv_heads, D, max_ctx;
    // token embedding prologue (layer 0): x = embed[st->token] copied to hidden_out first
    const bf16_t* embed;  // non-null: x := embed row of st->token ; also written to embed_out by block 0
    bf16_t* embed_out;
    // fp8 weight stream (non-null W8 selects it): E4M3 bytes [N, ldw8], row r dequantises as value * scale8[r]
    // (scale8 a power of two, so the scaling of a row's dot product is exact)
    const uint8_t* W8;
    const float* scale8;
    int ldw8;
    // FP8 activations (fp8 weight stream only, norm prologue only): the normalised x is replaced by its E4M3
    // quantisation (one power-of-two scale for the row) before the dot products -- the values k_rmsnorm(act8) writes
    int act8;
    // LOGITS with the greedy arg-max folded in (amax_ws non-null): every workgroup also leaves the best (penalised value,
    // index) of its rows in amax_ws[2 b], amax_ws[2 b + 1] (at most 2048 workgroups: the launch grid's cap);
    // k_argmax_final_folded reduces the 2048 slots -- no k_argmax_partial pass over the 600-KB logits row
    const uint8_t* seen;
    float penalty;
    float* amax_ws;
};
// returns false when x[K] does not fit the LDS stage
bool ze_launch_gemv(int epi, const ze_gemv_args& a, hipStream_t s);
// fp32 logits of the last hidden rows of n chains in one pass over the lm_head per eight chains (ze_gemv_logits.hip: the
// prefill paths; final RMSNorm fused); false = shape not covered, the caller launches the single-chain GEMV per chain
bool ze_launch_logits_rows(const bf16_t* W, int ldw, int N, int K, const bf16_t* norm_w, float eps, const bf16_t* const* x_rows,
                           float* const* out_rows, int n, hipStream_t s);
// per-row power-of-two-scale E4M3 quantisation of a bf16 matrix [rows, ld] (cols valid): writes the fp8 bytes
// [rows, ld8], the scales, and REPLACES the bf16 values by the dequantised ones (exactly representable)
void ze_launch_quantize_rows(bf16_t* w, int rows, int cols, int ld, uint8_t* q, int ld8, float* scale, hipStream_t s);

// ---- attention
// Varlen flash attention (prefill / ViT). Tiles: host-built list of (q_start, q_end, kv_start, kv_end) int4 rows.
// M-RoPE of the queries inside the flash kernel (round 5; D = 128 only): the kernel applies rope8 to the Q fragments it has just loaded --
// a lane holds d = ks * 32 + fq * 8 .. + 7 for ks = 0 .. 3, i.e. both halves of eight rotate_half pairs -- with the arithmetic, tables and
// position lookup of k_mrope_kv_vec: the same bf16 values that kernel would have written back.  cosT = nullptr: Q is used as it is.
struct ze_fa_rope {
    const bf16_t* cosT;
    const bf16_t* sinT;
    const int* pos3;     // [3][T] position ids of the pass
    int s0, s01;         // M-RoPE sections: pairs [0, s0) follow axis 0, [s0, s01) axis 1, the rest axis 2 (no table load in the kernel)
    int T;
};
void ze_launch_flash_attn(int D, int causal, const bf16_t* q, int q_row_stride, int q_head_stride, const bf16_t* k,
                          int k_row_stride, int k_head_stride, const bf16_t* v, int v_row_stride, int v_head_stride,
                          bf16_t* o, int o_row_stride, int o_head_stride, const int4* tiles, int n_tiles, int heads,
                          int group, float scale, int q_pos_offset, hipStream_t s, const int* tile_aux = nullptr,
                          size_t kv_seq_stride = 0, int q_tile = 64, int max_kv = 0, ze_fa_rope rope = ze_fa_rope{nullptr, nullptr, nullptr, 0, 0, 0});
// q_tile: the most query rows a tile of the list spans -- 64 (default) or ZE_FA_BQ_LONG = 128, where every wave holds two
// 16-query tiles and each K / V^T fragment it reads from LDS feeds two MFMAs (long segments: the prefill, the ViT's
// full-attention blocks); a row's bits do not depend on the choice
#define ZE_FA_BQ_LONG 128
// (tile_aux: per tile (chain slot, position offset): K/V move by slot * kv_seq_stride elements, the offset replaces
//  q_pos_offset -- batched prefill)
// Decode attention for one chain: q [heads, D]; caches [kv_heads, max_ctx, D]; context = st->ctx + 1 tokens.
// Batched form: n chains; chain b = seq_ids[b] (null: chain 0 is `st` itself), q/out rows b, caches offset by
// seq * cache_seq_stride elements, partials offset by b * max_splits * heads * 132 floats; tickets: n * kv_heads zeroed
// words (they return to zero inside every launch).
void ze_launch_attn_decode(const bf16_t* q, int q_row_stride, const bf16_t* kcache, const bf16_t* vcache,
                           size_t cache_seq_stride, bf16_t* out, int out_row_stride, const ze_seq_dev* st,
                           const int* seq_ids, int n, int heads, int kv_heads, int D, int max_ctx, float scale,
                           float* ws_partial, int max_splits, unsigned* tickets, hipStream_t s);

// Batched decode attention (ze_attn_batch.hip): parts of 256 tokens per (chain, kv head) streamed through an LDS-DMA
// double buffer; same argument meaning as the batched form above, max_parts = ceil(max_ctx / 256) (the partial
// workspace holds max_parts * heads * 132 floats per chain); seq_ids must not be null.
void ze_launch_attn_decode_stream(const bf16_t* q, int q_row_stride, const bf16_t* kcache, const bf16_t* vcache,
                                  size_t cache_seq_stride, bf16_t* out, int out_row_stride, const ze_seq_dev* st,
                                  const int* seq_ids, int n, int heads, int kv_heads, int max_ctx, float scale,
                                  float* ws_partial, int max_parts, unsigned* tickets, hipStream_t s, int chunk = 0,
                                  int per_wave = 0,   // per_wave: k_attn_decode_wave (every wave a stream of its own)
                                  const int* prefix = nullptr,
                                  // round 6 (the pipelined kernel): mate[row of the batch] = the row whose q heads share this row's PREFIX
                                  // parts (or -1; symmetric; null = nobody pairs), long_parts = the 384-key parts the batch's longest
                                  // chain has under its split (0: derive from per_wave), use_split = ze_seq_dev::split cuts the parts
                                  const int* mate = nullptr, int long_parts = 0, int use_split = 0);
// prefix (per_wave only; per chain SLOT, null = none): (source chain << 16) | P -- rows 0 .. P-1 of the source chain's KV cache
// hold the same bits as the chain's own (ze_seq_copy_prefix) and are read from the SOURCE, so the questions of one tile stream
// one copy of their image prefix (Infinity Cache hits)

// first n_tokens cached K/V rows of chain src -> chain dst (all layers / kv heads); strides in elements
void ze_launch_kv_copy_prefix(bf16_t* kcache, bf16_t* vcache, size_t layer_stride, size_t seq_stride, size_t head_stride,
                              int layers, int kv_heads, int D, int src, int dst, int n_tokens, hipStream_t s);

// ---- sampling
struct ze_sample_opts {
    float temperature = 0.f;      // 0: greedy arg-max; > 0: multinomial draw from softmax(score / temperature)
    unsigned long long seed = 0;  // draw = f(seed, ze_seq_dev::stream of the chain, index of the generated token)
    int slot = 0;                 // unused (kept for layout)
};
// ws: 2 * 128 arg-max partials + 64 spare + 128 chunk sums (floats)
void ze_launch_sample(const float* logits, int vocab, uint8_t* seen, float penalty, ze_seq_dev* st,
                      const int* eos_ids, int n_eos, int pad_id, int ignore_eos, int advance_ctx,
                      int32_t* out_tokens, float* ws, const ze_sample_opts& so, hipStream_t s);
void ze_launch_sample_folded(const float* amax_ws, int vocab, uint8_t* seen, ze_seq_dev* st, const int* eos_ids, int n_eos,
                             int pad_id, int ignore_eos, int advance_ctx, int32_t* out_tokens, hipStream_t s);
void ze_launch_amax_init(float* amax_ws, hipStream_t s);  // 2048 (value, index) slots
void ze_launch_advance_ctx(ze_seq_dev* st, hipStream_t s);
// batched decode helpers (one token for each of n chains)
// dst[0..n) = vals[0..n): the values travel as kernel arguments (no host staging buffer to keep alive, no stream
// synchronisation before it can be reused): chain-state pushes and the chain-id list of a batched step
void ze_launch_set_ints(int* dst, const int* host_vals, int n, hipStream_t s);
// dst[idx[i]] = vals[i], i < n: the pairs travel as kernel arguments too
void ze_launch_scatter_ints(int* dst, const int* host_idx, const int* host_vals, int n, hipStream_t s);
void ze_launch_embed_tokens_batch(const ze_seq_dev* st, const int* seq_ids, int n, const bf16_t* embed, bf16_t* out,
                                  int hidden, hipStream_t s);
void ze_launch_rope_kv_batch(bf16_t* qkv, int n, int heads, int kv_heads, int D, const bf16_t* cosT, const bf16_t* sinT,
                             const ze_seq_dev* st, const int* seq_ids, bf16_t* kcache, bf16_t* vcache,
                             size_t cache_seq_stride, int max_ctx, hipStream_t s);
void ze_launch_token_logprob(const bf16_t* logits, int ld, int vocab, const int* targets, float* out, int rows,
                             hipStream_t s);
void ze_launch_sample_batch(const float* logits, int vocab, uint8_t* seen_base, float penalty, ze_seq_dev* st,
                            const int* seq_ids, int n, const int* eos_ids, int n_eos, int pad_id, int ignore_eos,
                            int advance_ctx, int sample, int32_t* out_tokens_base, int max_gen, float* ws,
                            float* ws_sum, const ze_sample_opts& so, hipStream_t s);
void ze_launch_mark_seen(uint8_t* seen, const int* ids, int n, hipStream_t s);
void ze_launch_mark_seen_batch(uint8_t* seen, int vocab, const int* hdr, const int* ids, int n, int max_count, hipStream_t s);
void ze_launch_gather_chain_tokens(const ze_seq_dev* st, const int* out_tokens, int max_ctx, const int* slots, int n, int cap, int* out,
                                   hipStream_t s);
void ze_launch_numeric_helpers(const float* x, const float* y, uint32_t* out, uint32_t* out2, int n, hipStream_t s);
