/*
 * zoomearth.h -- C ABI of libzoomearth_hip.so, the MI355X (gfx950) engine for ZoomEarth's
 * zoom-crop-reason inference path.
 *
 * The reference (earth-insights/ZoomEarth) has no FFI / plugin interface: its hot path is reached
 * through HuggingFace Python calls.  Each entry point below names the
 * reference / transformers interface it replaces ("replaces:" lines; paths relative to
 * /root/reference, `HF:` = site-packages/transformers).  The Python host shim in zoomearth_amd/
 * binds exactly these symbols with ctypes.
 *
 * Conventions
 *   - every function returns 0 on success, a negative ze_status on error; the message is
 *     available from ze_last_error() (per engine, or the global one when the engine is NULL);
 *   - the caller owns every buffer it passes (device pointers normally come from torch tensors'
 *     data_ptr()); the library owns weights, KV cache and workspaces and frees them in
 *     ze_engine_destroy();
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *   - an engine is not thread-safe; distinct engines are independent (one per process / GPU);
 *   - plain pointers and sizes only, no torch types.
 */
#ifndef ZOOMEARTH_H
#define ZOOMEARTH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ze_engine ze_engine;

typedef enum {
    ZE_OK = 0,
    ZE_ERR_INVALID = -1,   /* bad argument / shape / state                         */
    ZE_ERR_HIP = -2,       /* a HIP runtime call failed                            */
    ZE_ERR_NOMEM = -3,     /* capacity (max_ctx / max_seqs / workspace) exceeded   */
    ZE_ERR_NOTFOUND = -4,  /* unknown weight name / sequence id                    */
    ZE_ERR_MISMATCH = -5   /* image features and image tokens do not match         */
} ze_status;

typedef enum { ZE_F32 = 0, ZE_F16 = 1, ZE_BF16 = 2 } ze_dtype;

#define ZE_MAX_FULLATT 16
#define ZE_MAX_EOS 4

/* Model + capacity description.  Field meaning follows the HF config.json of Qwen2.5-VL
 * (replaces: Qwen2_5_VLConfig, HF:models/qwen2_5_vl/configuration_qwen2_5_vl.py:32-207). */
typedef struct ze_config {
    /* vision tower */
    int32_t vit_depth, vit_hidden, vit_heads, vit_intermediate, vit_out_hidden;
    int32_t patch_size, temporal_patch_size, spatial_merge_size, window_size, in_channels;
    int32_t n_fullatt;
    int32_t fullatt_block_indexes[ZE_MAX_FULLATT];
    /* language model */
    int32_t hidden, layers, heads, kv_heads, intermediate, vocab;
    float rms_eps;
    float rope_theta;
    int32_t mrope_section[3];
    int32_t tie_word_embeddings;
    /* special tokens */
    int32_t image_token_id, vision_start_token_id, vision_end_token_id, pad_token_id;
    int32_t n_eos;
    int32_t eos_token_ids[ZE_MAX_EOS];
    /* capacities */
    int32_t max_seqs;        /* concurrent question chains (KV slots)             */
    int32_t max_ctx;         /* tokens per chain                                  */
    int32_t max_patches;     /* ViT patches per ze_vit_forward call               */
    int32_t max_tile_side;   /* largest tile edge handed to ze_op_resize_bicubic  */
    int32_t max_prefill_rows; /* rows of one (batched) prefill call; 0 = max_ctx    */
} ze_config;

/* ------------------------------------------------------------------ lifecycle */
/* replaces: Qwen2_5_VLForConditionalGeneration.from_pretrained(...).eval() + accelerator.prepare(model)
 * (src/eval/infer.py:147-151,171; src/demo.py:128): allocates weights / KV cache / workspaces on `device_id`. */
int ze_engine_create(const ze_config* cfg, int device_id, ze_engine** out);
int ze_engine_destroy(ze_engine* e);
const char* ze_last_error(const ze_engine* e);
int ze_version(void);
/* Blocks until all work queued on `stream` has finished (hipStreamSynchronize). */
int ze_sync(ze_engine* e, void* stream);

/* ------------------------------------------------------------------ weights */
/* replaces: the safetensors -> nn.Parameter copy done by from_pretrained (src/eval/infer.py:147-150).
 * `name` is an HF checkpoint key in either layout ("model.visual.*" / "model.language_model.*" (5.x) or
 * "visual.*" / "model.layers.*" (4.49), see src/train/RL/src/open-r1-multimodal/src/open_r1/model/
 * modeling_qwen2_vl.py:1290-1293).  `host_ptr` holds `shape` row-major elements of `dtype`; the engine converts to
 * bf16 (round-to-nearest-even; fp16 checkpoints pass through fp32) and packs into its own layouts. */
int ze_load_weight(ze_engine* e, const char* name, int dtype, int ndim, const int64_t* shape, const void* host_ptr);
/* Synthetic 3B-shape (or any config) checkpoint generated on the device from the repo PRNG
 * (oracle/prng.py is the CPU mirror): matrices N(0,(std*matrix_gain)^2), embed/lm_head N(0,std^2),
 * norm weights 1+N(0,norm_jitter^2), biases N(0,bias_std^2). */
int ze_weights_fill_synthetic(ze_engine* e, uint64_t seed, float std, float matrix_gain, float bias_std,
                              float norm_jitter);
/* Number of HF tensors still missing (0 = ready); names of missing tensors are in ze_last_error(). */
int ze_weights_missing(ze_engine* e);
/* The packed bf16 weight arena (one contiguous device allocation): lets the host broadcast it once over
 * RCCL/xGMI with torch.distributed and checksum it.  Does not transfer ownership. */
int ze_weights_arena(ze_engine* e, void** dev_ptr, size_t* bytes);
/* To be called after the host WROTE the whole arena through that pointer (the torch.distributed broadcast that replaces
 * every rank's own from_pretrained, src/eval/infer.py:147-151: rank 0 reads the checkpoint, the others receive the packed
 * arena; a weight refresh of the rollout engine): every tensor counts as loaded, the derived copies (fragment-major, FP8)
 * and the captured graphs are dropped and rebuilt from the new values on their next use.  ze_weights_arena itself has no
 * side effect. */
int ze_weights_invalidate(ze_engine* e);
/* The same broadcast for a host that owns an RCCL communicator (`ncclComm_t`, passed as void*): ncclBroadcast of the whole
 * arena from rank `root` on `stream`, in place (replaces: accelerator.prepare / every rank's own from_pretrained,
 * src/eval/infer.py:147-151,171 -- the checkpoint is read from disk once).  RCCL is not linked: the symbol must already be
 * loaded in the process (ZE_ERR_NOTFOUND otherwise).  Derived weight copies are dropped as after ze_load_weight. */
int ze_weights_broadcast(ze_engine* e, void* nccl_comm, int root, void* stream);

/* ------------------------------------------------------------------ image front-end (K0-K2) */
/* replaces: `Image.open(image_fp).convert("RGB")` arriving on the device (src/eval/infer.py:215,237 + the `.to(device)`
 * of :107): copies a decoded RGB u8 [h, w, 3] tile from host memory (pinned for full PCIe rate) into the caller's
 * device buffer on `stream`.  The copy is asynchronous when host_rgb is pinned; the tile is uploaded ONCE per tile and
 * every view / crop of every question about it reads it in place. */
int ze_tile_upload(ze_engine* e, const uint8_t* host_rgb, int h, int w, uint8_t* dev_rgb, void* stream);
/* replaces: PIL `image.crop(box).resize((dst_w, dst_h), Image.BICUBIC)` on RGB u8 as used by cut_image /
 * resize_image (src/eval/infer.py:41-85, src/demo.py:30-93) and by the HF image processor
 * (HF:models/qwen2_vl/image_processing_pil_qwen2_vl.py:126-150).  Bit-exact with Pillow.
 * src: u8 [src_h, src_w, 3] device, row stride src_w*3.  box = {x0,y0,x1,y1} may leave the image (zero fill).
 * dst: u8 [dst_h, dst_w, 3] device.  If the box size equals the dst size this is a pure crop. */
int ze_op_crop_resize(ze_engine* e, const uint8_t* src, int src_h, int src_w, const int32_t box[4], uint8_t* dst,
                      int dst_h, int dst_w, void* stream);
/* replaces: smart_resize (HF:...image_processing_pil_qwen2_vl.py:57-83). Host-only integer helper. */
int ze_smart_resize(int height, int width, int factor, int64_t min_pixels, int64_t max_pixels, int* out_h,
                    int* out_w);
/* replaces: rescale + normalize + patchify (HF:...image_processing_pil_qwen2_vl.py:152-187,197-246).
 * img: u8 [h, w, 3] device with h, w multiples of patch*merge.  out: f32 [gh*gw, C*T*P*P] device (HF row/column
 * order).  Bit-exact. */
int ze_op_patchify(ze_engine* e, const uint8_t* img, int h, int w, float* out_pixel_values, void* stream);
/* replaces: Qwen2VLImageProcessor._preprocess for one image: smart_resize -> bicubic -> patchify.
 * Writes grid_thw[3] (host) and pixel_values (device, capacity_rows rows). */
int ze_preprocess_image(ze_engine* e, const uint8_t* img, int h, int w, int64_t min_pixels, int64_t max_pixels,
                        float* out_pixel_values, int64_t capacity_rows, int32_t grid_thw[3], void* stream);

/* ------------------------------------------------------------------ integer index builders (host) */
/* replaces: get_vision_window_index / get_vision_cu_seqlens (HF:vision_utils.py:42-65,130-188).
 * window_index: [sum(t*h*w)/merge^2]; cu_window: capacity cap_cu, *n_cu written. */
int ze_vision_window_index(const ze_config* cfg, const int32_t* grid_thw, int n_images, int64_t* window_index,
                           int32_t* cu_window, int cap_cu, int* n_cu);
/* replaces: Qwen2_5_VLModel.get_rope_index (HF:models/qwen2_5_vl/modeling_qwen2_5_vl.py:944-1058) for ONE
 * unpadded sequence.  position_ids: [3, len] int32; *rope_delta = max+1-len. */
int ze_rope_index(const ze_config* cfg, const int32_t* input_ids, int len, const int32_t* grid_thw, int n_images,
                  int32_t* position_ids, int32_t* rope_delta);

/* ------------------------------------------------------------------ model */
/* replaces: Qwen2_5_VisionTransformerPretrainedModel.forward (HF:...modeling_qwen2_5_vl.py:408-471).
 * pixel_values: f32 [sum(t*h*w), C*T*P*P] device in HF order; grid_thw: host int32 [n_images,3];
 * out_embeds: bf16 [sum/merge^2, vit_out_hidden] device, HF order. */
int ze_vit_forward(ze_engine* e, const float* pixel_values, const int32_t* grid_thw, int n_images,
                   void* out_embeds_bf16, void* stream);

/* Sequence (question-chain) slots: 0 <= seq < max_seqs.  ze_seq_reset drops the KV cache of a slot and clears
 * its repetition-penalty set; ze_seq_truncate keeps the first `keep_len` cached tokens (stage-1 prefix reuse) and
 * also clears the repetition-penalty set (the caller re-marks the new prompt with ze_seq_mark_seen). */
int ze_seq_reset(ze_engine* e, int seq, void* stream);
int ze_seq_truncate(ze_engine* e, int seq, int keep_len, void* stream);
/* The chain in `seq` is over and the slot may go to another chain.  Chains that copied their prompt prefix from it
 * (ze_seq_copy_prefix) read those rows from ITS cache during decode (one copy per tile crosses the memory interface): this
 * call moves them to another holder of the same rows -- the reader with the longest copy THAT HAS LANDED (an event behind
 * every copy), or each to its own rows.  Hints are host state; the device words the decode attention reads are written by
 * the batched decode call itself, on ITS stream, for ITS chains, before it enqueues the step -- so a chain state pushed from a
 * prefill stream can never bring a stale hint back (round 4).  Contract: call this after the decode steps that may
 * still read the slot have been enqueued and before anything that overwrites the slot's rows is (a scheduler: between two
 * bursts); every step enqueued afterwards reads the new holders.  ze_seq_reset / ze_seq_truncate / a copy into the slot move
 * the readers likewise.  `stream` is unused (kept for the ABI).
 * (reference: no counterpart -- HF generate() holds every chain's cache for the whole call). */
int ze_seq_retire(ze_engine* e, int seq, void* stream);
/* Declares that the first `rows` cached tokens of `seq` equal the source chain's, bit for bit (ze_seq_copy_prefix records
 * this itself; a caller that wrote the same prefix into both may say so; rows = 0 clears).  Also how bench.py times the
 * decode attention with the sharing its question stream has. */
int ze_seq_set_prefix_hint(ze_engine* e, int seq, int src_seq, int rows, void* stream);
/* Round 6.  A chain's SPLIT ROW is the row behind the <|vision_end|> of its first image block (the rows the questions about one tile
 * share end there); ze_prefill / ze_prefill_batch find it in the ids they are given, ze_seq_copy_prefix hands it on, and -- under ze_tune
 * knob 23 = 2 / 3 only: the form was measured slower and is not the default -- the batched decode attention cuts a chain's parts there
 * ([0, split) and [split, ctx) in 384-key pieces) so that two chains reading those rows from one holder can share a workgroup per
 * prefix part.  This entry sets it by hand -- measurement / tests on text-only stand-ins. */
int ze_seq_set_split(ze_engine* e, int seq, int rows, void* stream);
/* (source chain << 16) | rows: whose cache the decode attention reads the first rows of `seq` from; 0 = its own. */
int ze_seq_prefix_hint(ze_engine* e, int seq);
int ze_seq_len(ze_engine* e, int seq);
/* Shared prompt prefixes (the questions about one tile start with the same system turn and the same view's image tokens --
 * 347 of the 802 tokens of a stage-1 prompt, src/eval/infer.py:180-214): chain `dst_seq` becomes the first `n_tokens`
 * cached tokens of chain `src_seq` (K/V rows of every layer copied, repetition-penalty set cleared); the caller then
 * ze_prefill's only the tokens after the prefix.  A prefix's K/V rows depend on the prefix alone and every kernel of the
 * prefill computes a row independently of what else shares the pass, so this is bit-identical to prefilling the whole
 * prompt into dst_seq. */
int ze_seq_copy_prefix(ze_engine* e, int dst_seq, int src_seq, int n_tokens, void* stream);

/* replaces: the prefill forward of Qwen2_5_VLForConditionalGeneration (HF:...:1185-1253,1308-1400): embed,
 * image scatter, M-RoPE, decoder layers, final norm, lm_head on the last position.
 * input_ids: host int32 [len] = the NEW tokens appended after the `ze_seq_len` cached ones (image placeholders
 * already expanded); image_embeds: bf16 device rows consumed in order by the image tokens of input_ids;
 * position_ids: host int32 [3,len] (from ze_rope_index over the whole sequence); rope_delta: stored for decode.
 * out_logits: f32 [vocab] device (values are bf16-rounded like HF's `.float()` of bf16 logits), may be NULL. */
int ze_prefill(ze_engine* e, int seq, const int32_t* input_ids, int len, const void* image_embeds_bf16,
               int n_image_rows, const int32_t* position_ids, int rope_delta, float* out_logits, void* stream);

/* replaces: one iteration of GenerationMixin._sample in greedy mode (HF:generation/utils.py:2876-2936) for one
 * chain WITHOUT the sampling: forward one token with the KV cache and return the raw fp32 logits (teacher forcing
 * / host-side sampling with ze_op_sample_greedy).  Feeds `token` (or, if token < 0, the chain's last sampled
 * token); out_logits may be NULL.  ze_generate runs the same step with the device-side sampler appended. */
int ze_decode_step(ze_engine* e, int seq, int token, float* out_logits, void* stream);

/* Generation options (replaces the kwargs of model.generate at src/eval/infer.py:109-115, src/demo.py:14-19).
 * do_sample = 0: greedy arg-max (src/demo.py:17).  do_sample = 1 with temperature > 0: multinomial draw from
 * softmax(penalised logits / temperature) -- TemperatureLogitsWarper + torch.multinomial as src/eval/infer.py:109-115
 * uses them (temperature 0.01, top_k = top_p = None); same distribution, but the random stream is this library's
 * counter-based generator: draw = f(seed, row of the chain in the call, index of the generated token): the same
 * request reproduces whatever chain slot it lands in (see ze_op_sample_temperature). */
typedef struct ze_gen_params {
    int32_t max_new_tokens;
    float repetition_penalty;  /* 1.0 = off */
    int32_t ignore_eos;        /* scripted benchmark mode: run exactly max_new_tokens steps */
    int32_t use_graph;         /* replay the decode step as a hipGraph */
    int32_t sync_every;        /* host EOS check interval in steps (>=1) */
    int32_t do_sample;         /* 0 greedy, 1 temperature sampling */
    float temperature;         /* used when do_sample != 0; must be > 0 then */
    uint64_t seed;             /* sampling stream */
} ze_gen_params;

/* replaces: GenerationMixin.generate greedy loop after prefill for one chain.  Samples the first token from the
 * logits left by ze_prefill, then runs decode steps until EOS / max_new_tokens.  out_tokens: host int32
 * [max_new_tokens]; *n_out = number of tokens produced (EOS included). */
int ze_generate(ze_engine* e, int seq, const ze_gen_params* p, int32_t* out_tokens, int* n_out, void* stream);
/* Prefill of several chains in ONE pass (BASELINE configs[2]): the rows of all chains go through every GEMM
 * together (M = sum of lens), attention runs per chain against its own KV cache.  Arguments are the per-chain
 * arguments of ze_prefill, concatenated in chain order: input_ids [sum lens]; position_ids [3, sum lens] (axis-major
 * over the concatenation); image_embeds rows in chain order (n_image_rows[i] per chain); rope_deltas[i].  A chain's
 * result is bit-identical to a ze_prefill of that chain alone.  Logits of each chain's last position stay in the
 * engine (first token of ze_generate / ze_generate_batch).  sum lens <= max_prefill_rows. */
int ze_prefill_batch(ze_engine* e, const int32_t* seqs, int n, const int32_t* lens, const int32_t* input_ids,
                     const void* image_embeds, const int32_t* n_image_rows, const int32_t* position_ids,
                     const int32_t* rope_deltas, void* stream);
/* Batched decode (BASELINE configs[2]: many question chains per GPU).  One token for each of the n distinct chains
 * `seqs[i]`: the weights are streamed once for the whole batch (MFMA path, rows = chains); each chain keeps its own
 * KV cache, position and repetition-penalty set, and its results do not depend on the batch composition.
 * ze_decode_batch: teacher forcing / raw logits (tokens[i] < 0 or tokens == NULL feeds the chain's last token);
 * out_logits: f32 [n, vocab] device or NULL.  ze_generate_batch: greedy loop for all chains after their prefills
 * (continuous batching: chains that hit EOS leave the batch); out_tokens host int32 [n, max_new_tokens], n_out [n]. */
int ze_decode_batch(ze_engine* e, const int32_t* seqs, int n, const int32_t* tokens, float* out_logits, void* stream);
/* Kernel family of the batched decode step: 0 = the fragment kernels (at most 64 chains per step; more is an error),
 * 1 = the row-streaming kernels (any count), -1 (default) = by the engine's capacity: max_seqs > 64 -> 1.  The family is a
 * property of the engine, never of how many chains are live, so a chain's tokens do not depend on the batch it shares
 * (within a family bit for bit; the two families agree within bf16 rounding).  Returns the family in force (0 / 1). */
int ze_set_decode_regime(ze_engine* e, int regime);
int ze_generate_batch(ze_engine* e, const int32_t* seqs, int n, const ze_gen_params* p, int32_t* out_tokens,
                      int32_t* n_out, void* stream);
/* Continuous batching (replaces: the request stream the reference keeps in flight against its serving back-end,
 * src/eval/infer_vllm.py:244-271, and -- run one sample at a time there -- the question loop of
 * src/eval/infer.py:173-249).  Chains join and leave a running batch BETWEEN bursts of decode steps:
 *   ze_prefill / ze_prefill_batch the newcomers (+ ze_seq_mark_seen), ze_chain_begin each of them (first token from
 *   the prefill logits; `sample_stream` selects the chain's random stream when p->do_sample, e.g. the request number,
 *   so a request reproduces whatever batch it lands in), then ze_decode_burst for ALL live chains, read which ones
 *   finished, ze_chain_tokens them, hand their slots to waiting requests, repeat.
 * ze_decode_burst runs `steps` sampled decode steps for the n distinct chains (fewer if a chain's max_ctx budget ends
 * sooner) and returns the number of steps run (>= 0) or a negative ze_status; n_generated[i] / finished[i] (host,
 * either may be NULL) report each chain's token count so far and its EOS flag after the burst.  p->max_new_tokens and
 * p->sync_every are not used here: the caller owns the step budget.  A chain's tokens do not depend on which chains
 * share its bursts.  ze_chain_tokens copies the chain's generated ids (trimmed after the first EOS) to the host. */
int ze_chain_begin(ze_engine* e, int seq, const ze_gen_params* p, int sample_stream, void* stream);
int ze_decode_burst(ze_engine* e, const int32_t* seqs, int n, int steps, const ze_gen_params* p, int32_t* n_generated,
                    int32_t* finished, void* stream);
/* The same burst in two halves, for a host that overlaps it with other work: _begin enqueues the steps on `stream` and
 * returns at once (the number of steps enqueued, or a negative ze_status); the host may now enqueue the ViT / prefill
 * round of the NEXT newcomers on ANOTHER stream (the batched step has activation buffers of its own; chains being
 * prefilled are not part of the burst, so no state is shared) -- the GPU overlaps the matrix-bound prefill with the
 * bandwidth-bound decode steps; _end waits for the burst and reports as ze_decode_burst does.  No other call may touch the
 * burst's chains or use the burst's stream between the two halves. */
int ze_decode_burst_begin(ze_engine* e, const int32_t* seqs, int n, int steps, const ze_gen_params* p, void* stream);
int ze_decode_burst_end(ze_engine* e, const int32_t* seqs, int n, int32_t* n_generated, int32_t* finished, void* stream);
int ze_chain_tokens(ze_engine* e, int seq, int32_t* out_tokens, int capacity, int* n_out, void* stream);
/* ze_chain_tokens for the n chains a burst retires, in one gather launch + one device -> host copy + one wait.
 * out_tokens: host int32 [n, capacity] (row i: the ids of seqs[i], trimmed after the first EOS), n_out [n]. */
int ze_chain_tokens_batch(ze_engine* e, const int32_t* seqs, int n, int32_t* out_tokens, int capacity, int32_t* n_out, void* stream);
/* Rollout scoring (replaces _get_per_token_logps, src/train/RL/src/open-r1-multimodal/src/open_r1/trainer/
 * grpo_trainer.py:494-504, as the trainer calls it under torch.no_grad for the old policy and the reference model,
 * :660-683): ze_prefill of the sequence, plus, for EVERY position t < len - 1,
 *   out_logps[t] = log_softmax(logits[t])[input_ids[t + 1]]
 * with logits[t] the lm_head output in bf16 (as HF's bf16 lm_head returns it) and the log-softmax in fp32.
 * out_logps: device f32 [len - 1].  The chain is left as after ze_prefill (KV cache filled, last-position logits
 * ready), so a generation can continue from it.  Arguments as ze_prefill. */
int ze_score(ze_engine* e, int seq, const int32_t* input_ids, int len, const void* image_embeds, int n_image_rows,
             const int32_t* position_ids, int rope_delta, float* out_logps, void* stream);
/* Marks every id in `ids` (host int32) as seen for the repetition penalty of `seq` (the prompt). */
int ze_seq_mark_seen(ze_engine* e, int seq, const int32_t* ids, int n, void* stream);
/* ze_seq_mark_seen for the n chains of a prefill pass at once: ids = the chains' prompts back to back, counts[i] ids for seqs[i]. */
int ze_seq_mark_seen_batch(ze_engine* e, const int32_t* seqs, const int32_t* counts, int n, const int32_t* ids, void* stream);
/* Applies penalty + argmax to f32 logits [vocab] (device) with the seen-set of `seq`; *out_token host. */
int ze_op_sample_greedy(ze_engine* e, int seq, const float* logits, float repetition_penalty, int32_t* out_token,
                        void* stream);
/* One temperature-sampling draw (replaces TemperatureLogitsWarper + softmax + torch.multinomial,
 * HF:generation/utils.py:2894-2916) on f32 logits [vocab] (device) with the seen-set of `seq`:
 *   u = (stream64(mix64(seed ^ mix64(row + 1)), index) >> 40) * 2^-24 (row = 0 here, the chain's row in a batched
 *   generate call) ; e_i = expf(score_i / T - max score / T) ;
 *   token = first i whose running sum of e exceeds u * sum(e)   (fp32 sums in the order given in ze_sample.hip).
 * `index` = position of the draw in the generated sequence (what ze_generate passes). *out_token host. */
int ze_op_sample_temperature(ze_engine* e, int seq, const float* logits, float repetition_penalty, float temperature,
                             uint64_t seed, int index, int32_t* out_token, void* stream);

/* FP8 decode weights (BASELINE.json configs[4], "fp8 weights"): quantises the decoder's linear layers (and an untied
 * lm_head) to OCP E4M3 with one power-of-two scale per output row, REPLACES the bf16 copies by the dequantised
 * values (exactly representable) so that prefill and decode compute with identical weights, and switches the batch-1
 * decode GEMVs to the fp8 stream (half the HBM bytes per token).  Call once, after the weights are loaded. */
int ze_weights_quantize_fp8(ze_engine* e, void* stream);
/* FP8 activations on top of the FP8 weights (opt-in; ZE_ERR_INVALID unless ze_weights_quantize_fp8 ran): the inputs of
 * the qkv and gate/up projections -- the two RMSNorm outputs of every decoder layer -- are quantised to E4M3 with one
 * dynamic power-of-two scale per token row (k = smallest integer with max|y| / 2^k <= 448, the rule of the weight
 * rows).  Batched decode (<= 64 chains) then multiplies FP8 x FP8 on v_mfma_f32_16x16x32_fp8_fp8; every other path
 * (prefill, scoring, single-chain GEMV decode, > 64 chains) computes with the same values, q * 2^k kept as bf16, so
 * the model is ONE model whichever kernel serves a token.  The o / down / lm_head inputs stay bf16.  A weight
 * change (load, fill, arena hand-out, broadcast) switches it off together with the FP8 weights.  `on` = 0 returns
 * to bf16 activations.  There is no counterpart in the reference (its checkpoints run bf16); oracle:
 * oracle/qwen25vl.py `act_fp8`. */
int ze_set_fp8_activations(ze_engine* e, int on);
/* The quantiser on one matrix: w bf16 [rows, cols] (device, overwritten with the dequantised values), q_out u8
 * [rows, cols], scale_out f32 [rows]; cols % 16 == 0. */
int ze_op_quantize_fp8(ze_engine* e, void* w_bf16, int rows, int cols, void* q_out, void* scale_out, void* stream);
/* The block-scaled FP8 GEMM of the prefill path (v_mfma_scale_f32_16x16x128_f8f6f4), one call: C[M, N] (bf16; [M, N / 2]
 * with swiglu != 0: interleaved gate / up rows as in the packed weights) = (A8 2^ka) (W8 2^kw)^T + bias, A8 u8 [M, K] and
 * W8 u8 [N, K] E4M3 bytes with one power-of-two scale per row each (sa f32 [M], sw f32 [N]: what ze_op_quantize_fp8
 * writes); K % 128 == 0.  Unit-test / measurement entry: the prefill uses the same kernel behind ze_set_fp8_activations. */
int ze_op_linear_mx(ze_engine* e, const void* a8, const void* sa, const void* w8, const void* sw, const void* bias, void* c,
                    int M, int N, int K, int swiglu, void* stream);

/* ------------------------------------------------------------------ unit ops for parity tests (K3-K22) */
/* C[M,N] = A[M,K] * W[N,K]^T (+bias[N]) ; bf16 in, fp32 accumulate, bf16 out (one rounding).  act: 0 none, 1 exact GELU,
 * 2 none through the weight-streaming launcher of the batched decode step (split-K; for measurements),
 * 3 none through the fragment-major kernels of that step (operands packed inside the call; M <= 64, N % 16 == 0,
 *   K % 32 == 0, K <= 4096); 5 the same operands through the sixteen-wave one-shot kernel of the qkv / o projections,
 * 4 the SwiGLU epilogue of the MLP (HF:...modeling_qwen2_5_vl.py:85-96,541-554): W holds gate and up rows interleaved
 *   in blocks of 16 ([gate 0..15 | up 0..15 | gate 16..31 | ...], N = 2 * width, width % 16 == 0), bias likewise;
 *   C is [M, N/2] = bf16(bf16(silu(bf16(gate))) * bf16(up));
 * 6 / 7 none / SwiGLU through the launcher of the row-streaming decode regime (ze_set_decode_regime = 1; rows = chains);
 * 10 the lm_head of that regime: no bias, C is FP32 [M, N] (HF casts the logits to fp32, HF:generation/utils.py:2894). */
int ze_op_linear(ze_engine* e, const void* a_bf16, const void* w_bf16, const void* bias_bf16, void* c_bf16, int M,
                 int N, int K, int act, void* stream);
/* y = weight * bf16(x * rsqrt(mean(x^2)+eps))  (HF:...modeling_qwen2_5_vl.py:64-79), rows x cols bf16. */
int ze_op_rmsnorm(ze_engine* e, const void* x_bf16, const void* weight_bf16, void* y_bf16, int rows, int cols,
                  float eps, void* stream);
/* out[r] = log_softmax(logits[r, :vocab])[targets[r]] in fp32 (the pick of ze_score): logits bf16 [rows, ld] device
 * (ld >= vocab, ld % 8 == 0), targets int32 [rows] device, out f32 [rows] device. */
int ze_op_token_logprob(ze_engine* e, const void* logits_bf16, int rows, int vocab, int ld, const int32_t* targets,
                        float* out, void* stream);
/* Varlen attention over segments: q,k,v,o bf16 [T, heads, D] (D = 80 or 128); cu_seqlens host int32 [n_seg+1];
 * causal applies inside each segment; kv_heads divides heads (GQA). */
int ze_op_attention(ze_engine* e, const void* q, const void* k, const void* v, void* o, int T, int heads,
                    int kv_heads, int D, const int32_t* cu_seqlens, int n_seg, int causal, void* stream);

/* K4 (HF:...modeling_qwen2_5_vl.py:434-439; HF:vision_utils.py:130-188): pixel_values f32 [n, 1176] (device, HF patch order) ->
 * bf16 rows in WINDOW order (groups of merge^2 rows follow window_index) -- what the patch embed reads.  ze_op_window_scatter
 * is the inverse on merged rows (HF:...:464-466): x bf16 [n / merge^2, cols] in window order -> HF order. */
int ze_op_window_gather(ze_engine* e, const float* pixel_values, const int32_t* grid_thw, int n_images, void* out_bf16, void* stream);
int ze_op_window_scatter(ze_engine* e, const void* x_bf16, int cols, const int32_t* grid_thw, int n_images, void* out_bf16,
                         void* stream);
/* K5 + K8 (HF:...:125-134,160-171,441-446): 2-D vision rotary on the q and k thirds of qkv bf16 [n, 3 * heads * 80] in place,
 * tables and arithmetic in fp32, one rounding to bf16; rows in window order (window_order != 0: the ViT's own layout) or HF order. */
int ze_op_vision_rope(ze_engine* e, void* qkv_bf16, const int32_t* grid_thw, int n_images, int window_order, void* stream);
/* K13 (HF:...:1206-1215, get_placeholder_mask + masked_scatter :1094-1133): out bf16 [len, hidden] = embed_tokens[input_ids],
 * image-token rows replaced by the rows of image_embeds (bf16 [n_image_rows, hidden]) in order; ZE_ERR_MISMATCH when the counts
 * differ (HF raises "Image features and image tokens do not match"). */
int ze_op_embed_scatter(ze_engine* e, const int32_t* input_ids, int len, const void* image_embeds, int n_image_rows, void* out_bf16,
                        void* stream);
/* K15 + K18 (apply_multimodal_rotary_pos_emb HF:...:557-599 with cos / sin cast to bf16 :538; DynamicCache.update :667-668):
 * qkv bf16 [T, (heads + 2 kv_heads) * 128] -- q roped in place, roped k and v appended to `layer`'s cache of chain `seq` at
 * positions past .. past + T - 1; position_ids host int32 [3, T].  The chain's length does not change (unit op).
 * ze_op_rope_kv_decode: the decode-step form, row b = chain seqs[b] at its own position (ctx + rope_delta), appended at ctx.
 * ze_op_kv_read: rows [start, start + n) of that cache, out_k / out_v bf16 [kv_heads, n, 128] device. */
int ze_op_mrope_kv(ze_engine* e, int seq, int layer, void* qkv_bf16, int T, const int32_t* position_ids, int past, void* stream);
int ze_op_rope_kv_decode(ze_engine* e, const int32_t* seqs, int n, int layer, void* qkv_bf16, void* stream);
int ze_op_kv_read(ze_engine* e, int seq, int layer, int start, int n, void* out_k, void* out_v, void* stream);
/* The attention of one batched decode step, alone (K16 at decode: HF:modeling_qwen2_5_vl.py:606-639 on the rows of the KV cache):
 * qkv_bf16 [n, (heads + 2 kv_heads) x 128] = the chains' projections after ze_op_rope_kv_decode (their k / v are row ctx of the
 * cache); out_bf16 [n, heads x 128] = softmax(q K^T / sqrt(128)) V over rows 0 .. ctx, per head, with the step's own kernel. */
int ze_op_attn_decode(ze_engine* e, const int32_t* seqs, int n, int layer, const void* qkv_bf16, void* out_bf16, void* stream);

/* The numeric helpers every epilogue shares, alone (device pointers, n elements): out[i] = bf16(x[i]) | bf16(bf16(SiLU(x[i])) * y[i]) << 16
 * -- HF Qwen2MLP `act_fn(gate_proj(x)) * up_proj(x)` on bf16 modules, transformers/models/qwen2_5_vl/modeling_qwen2_5_vl.py:541-554 --
 * and out2[i] = the packed pair (bf16(x[i]) | bf16(y[i]) << 16) of the library's f32 -> bf16 store.  Parity ledger: tests hold both
 * to float64 over every bf16 input. */
int ze_op_numeric_helpers(ze_engine* e, const float* x, const float* y, uint32_t* out, uint32_t* out2, int n, void* stream);

/* ------------------------------------------------------------------ measurement */
/* Runs the decode-path weight-streaming kernel `which` (0 qkv, 1 o_proj, 2 gate_up, 3 down, 4 lm_head) `iters`
 * times back to back on `stream`, cycling through the layers' real weights, bracketed by HIP events on that
 * stream; returns the average launch duration (us) and the algorithmic bytes of one launch. */
int ze_profile_decode_kernel(ze_engine* e, int which, int iters, float* avg_us, double* bytes_per_launch,
                             void* stream);
/* The same for the kernels of the BATCHED decode step at n chains (slots 0..n-1 with the contexts they hold):
 * which = 0 qkv, 1 o_proj, 2 gate_up, 3 down, 4 lm_head, 5 decode attention (bytes = the K/V rows of the n chains),
 * 6 RMSNorm, 7 rope + KV append. */
int ze_profile_batch_kernel(ze_engine* e, int which, int n, int iters, float* avg_us, double* bytes_per_launch,
                            void* stream);
/* The same for the projections of a PREFILL pass (HF Qwen2_5_VLDecoderLayer: q/k/v_proj, o_proj, gate/up_proj, down_proj,
 * transformers/models/qwen2_5_vl/modeling_qwen2_5_vl.py:541-554, 692-757) on the pass's own operands: the layers' real weights in
 * rotation and the activation rows the last ze_prefill_batch left in the workspace, through the launcher the pass uses.
 * which = 0 qkv, 1 o_proj, 2 gate_up (SwiGLU), 3 down; rows <= max_prefill_rows; outputs go to scratch.  flops_per_launch =
 * 2 x rows x N x K (unpadded). */
int ze_profile_prefill_kernel(ze_engine* e, int which, int rows, int iters, float* avg_us, double* flops_per_launch,
                              void* stream);
/* The four projections in PASS ORDER (qkv, o, gate/up, down; `layers_run` layers, the weights in rotation), every launch between its own
 * pair of HIP events: the per-projection averages a pass sees (clocks and cache state of the neighbouring launches), which the line's
 * `top_kernel_by_gpu_time` quotes next to the rocprofv3 figure of the replayed pass.  avg_us / flops: [0] qkv [1] o [2] gate/up [3] down. */
int ze_profile_prefill_layer(ze_engine* e, int rows, int layers_run, float avg_us[4], double flops[4], void* stream);
/* Measurement-only kernel-configuration override (A/B of kernels and launch shapes inside one process; value 0 is always
 * the shipped default, every alternative computes the same function -- bit for bit unless noted).  Knobs (0..23; round 4 re-used
 * knobs 3 and 4, which until round 3 switched the removed one-launch-per-layer kernels: an old `3:1` / `4:1` habit now changes GEMM
 * tiling / launch form -- same results, different speed):
 *   0, 1  variants of the single-chain down / gate-up GEMVs          2   grid cap of the GEMV family
 *   3     first row count of the 320 x 192 decode tiles (measurements) 5   1: no fragment / skinny kernels in the batched step
 *   4     launch form of the eight-phase GEMM: 0 = by the number of engines alive in this process (one: persistent workgroups;
 *         several -- lanes sharing the GPU --: one tile per workgroup, so that the other engine's kernels get CUs at tile boundaries;
 *         the library counts its engines itself, no caller sets this), 1 = always one tile per workgroup, 2 = always persistent
 *         (same bits in every form)
 *   6, 7  GEMM policy (register-staged vs LDS-DMA ring; forced tile) 8   batched decode attention: 2 = ring kernel,
 *                                                                        1 = 64-key slice kernel (agree within rounding)
 *   9     1: skinny instead of one-shot o projection                 10  1: bf16 fragments on a quantised engine
 *   12    1: bf16 GEMMs instead of the block-scaled FP8 MFMA in a prefill with FP8 activations (agree within rounding)
 *   11    fixed part size (keys) of the ring attention kernel        13  1: streaming launcher beyond 64 chains
 *   14    1: separate arg-max pass in single-chain greedy decode
 *   15    tile family of the row-streaming decode GEMMs (measurements)  16  grid rotation step | shift << 4 of the decode attention
 *   17    1: no shared-prefix hints (every chain reads its own K/V rows: same bits, more HBM traffic)
 *   18    1: round 4's K loop on the 513 .. 768-row decode tiles (K-steps of 32 in four stages; default: 64 in two; same bits)
 *   19    4: the tall qkv / o decode tiles on a plain four-stage ring, one barrier per K-step (default: six stages in groups of two)
 *   20    3: the long-K projection of a prefill pass (K > 4096: down) in three K slices on every prefill kernel (round 6: built, slower --
 *            down 469 -> 725 us at 12.8 K rows, the stream 86.4 -> 82.4 questions/s -- kept for its bit-equality test; other bits than 0)
 *   21    1: column walk of the eight-phase GEMM's tile grid (default: 8 x 4 blocks; > 1: R << 8 | C blocks)
 *   22    1: the prefill's queries rotated in place by the M-RoPE kernel (default at head_dim 128: inside the flash kernel, as it loads
 *            them; the M-RoPE kernel then writes K and V only; same bits)
 *   23    batched decode attention, round 6 (built, measured slower, off): 2 = a chain's parts cut at its split row, 3 = that + two chains
 *            of a tile per workgroup on the prefix parts (2 and 3: the same bits; against 0 -- parts of the whole context -- within rounding)
 * Changing a knob invalidates captured decode graphs (they are re-captured on the next step).
 * PROCESS-WIDE, by design: there is no engine argument, every engine of the process sees the value at its next launch, and nothing in
 * the product path (Engine, scheduler, entry points, clone_lane) calls it -- a knob is for A/B measurements and the bit-equality tests,
 * which reset it to 0 when they are done.  A library user who wants a per-engine policy has none to set: value 0 is the only supported
 * production setting. */
int ze_tune(int knob, int value);
/* Per-phase device time (ms) accumulated by HIP events since the last reset: [0] front-end, [1] ViT,
 * [2] prefill, [3] decode, [4] sampling.  Only recorded while enabled. */
int ze_phase_timers(ze_engine* e, int enable, int reset, float out_ms[5]);

#ifdef __cplusplus
}
#endif
#endif /* ZOOMEARTH_H */
