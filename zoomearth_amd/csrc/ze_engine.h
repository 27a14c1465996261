// ze_engine: owns the packed bf16 weight arena, the KV cache, chain state and workspaces of one GPU.
//
// HBM layout (3B config, sizes for max_seqs=4, max_ctx=4096):
//   weight arena  (one hipMalloc, ~7.5 GB bf16)   all matrices [N, K] row-major, K contiguous (the MFMA / GEMV feed)
//       ViT   : patch_embed [1280,1176]; per block qkv [3840,1280]+b, proj [1280,1280]+b,
//               gate_up [2*3424,1280]+b (gate/up rows interleaved in blocks of 16, zero padded 3420->3424),
//               down [1280,3424]+b (K zero padded); merger ln_q, mlp.0 [5120,5120]+b, mlp.2 [2048,5120]+b
//       LLM   : embed_tokens [151936,2048] (= lm_head when tied); per layer qkv [2560,2048]+b (q|k|v stacked),
//               o [2048,2048], gate_up [2*11008,2048] (interleaved), down [2048,11008]; norms
//   KV cache      [layers][max_seqs][kv_heads][max_ctx][128] bf16, K and V separate (36,864 B per cached token)
//   tables        text cos/sin bf16 [max_ctx + 512][64]; normalise LUT f32 [3][256]
//   chain state   ze_seq_dev[max_seqs], seen-set u8 [max_seqs][vocab], out tokens int32 [max_seqs][max_ctx]
//   workspaces    ViT activations for max_patches rows; prefill activations for max_ctx rows; decode vectors
#pragma once
#include <map>
#include <tuple>
#include <set>
#include <string>
#include <vector>

#include "ze_host.h"
#include "ze_kernels.h"

struct ze_linear {
    bf16_t* w = nullptr;
    bf16_t* bias = nullptr;
    int n = 0, k = 0, ld = 0;
    // fp8 decode copy (ze_weights_quantize_fp8): E4M3 bytes [n, ld8] + per-row scale exponent applied as 2^k
    uint8_t* w8 = nullptr;
    float* scale8 = nullptr;
    int ld8 = 0;
    // MFMA-fragment-major copy for the batched decode step (ze_launch_pack_fragments), null = none
    bf16_t* wf = nullptr;
    // qkv only, row-streaming regime: rows (and bias) permuted per head for the rope + KV-append epilogue (ze_launch_permute_qkv)
    bf16_t* wp = nullptr;
    bf16_t* bias_p = nullptr;
    // the same in FP8 (ze_launch_pack_fragments8 of w8), streamed with scale8 when the engine is quantised, null = none
    uint8_t* wf8 = nullptr;
};
struct ze_vit_block {
    bf16_t *norm1 = nullptr, *norm2 = nullptr;
    ze_linear qkv, proj, gate_up, down;
};
struct ze_text_layer {
    bf16_t *in_norm = nullptr, *post_norm = nullptr;
    ze_linear qkv, o, gate_up, down;
};
// where an HF tensor lands in the packed arena
struct ze_dest {
    bf16_t* dst = nullptr;
    int rows = 0, cols = 0, ld = 0, mode = 0, offset = 0;
    int kind = 0;  // 0 matrix, 1 norm weight, 2 bias, 3 embedding / lm_head
};

struct ze_engine {
    ze_config cfg{};
    int device = 0;
    std::string err;
    int head_dim = 128, vit_head_dim = 80, vit_ipad = 0, text_ipad = 0, max_pos = 0;

    // weights
    bf16_t* arena = nullptr;
    size_t arena_elems = 0, arena_used = 0;
    std::map<std::string, ze_dest> dests;
    std::set<std::string> loaded;
    void* staging = nullptr;
    size_t staging_bytes = 0;
    ze_linear patch_embed, merger0, merger2;
    bf16_t* ln_q = nullptr;
    std::vector<ze_vit_block> vb;
    bf16_t *embed = nullptr, *lm_head = nullptr, *final_norm = nullptr;
    ze_linear lm_head8;          // fp8 copy of an untied lm_head (w / ld unused)
    uint8_t* arena8 = nullptr;    // fp8 decode weights (0 until ze_weights_quantize_fp8)
    bool fp8_ready = false;
    bool fp8_act = false;         // ze_set_fp8_activations: qkv / gate-up inputs quantised to E4M3 per row (needs fp8_ready)
    uint8_t* ty8p = nullptr;      // prefill with FP8 activations: the normalised rows as E4M3 bytes, row-major [rows, hidden]
    float* ty8p_scale = nullptr;  // ... and their scales (block-scaled MFMA GEMM, ze_gemm.hip: k_gemm_ring_mx)
    float* damax = nullptr;       // single-chain decode: arg-max partials of the lm_head GEMV's workgroups (count, pairs)
    uint8_t* ty8 = nullptr;       // batched decode: the normalised rows as FP8 fragments (64 rows x hidden bytes)
    float* ty8_scale = nullptr;   // ... and their per-row scales (64)
    // second, fragment-major copy of the wide decode projections (qkv, gate/up, lm_head) for batched decode: built on
    // the first batched step, rebuilt after any weight change (the 288 GB of HBM make the extra 4.2 GB free)
    bf16_t* arena_f = nullptr;
    uint8_t* arena_f8 = nullptr;  // FP8 fragment copies (fp8_ready only)
    bf16_t* lm_head_f = nullptr;
    bool frag_ready = false;
    bf16_t* arena_p = nullptr;        // permuted qkv rows + biases of every layer (row-streaming regime, head_dim 128)
    ze_qkv_epi* qkv_epi_dev = nullptr;  // [layers] epilogue arguments of the fused qkv launch
    // Kernel family of the batched decode step (ze_set_decode_regime): 0 = fragment kernels (at most 64 chains per step),
    // 1 = row-streaming kernels (any count), -1 = by the engine's capacity (max_seqs > 64 -> 1).  Never a function of how
    // many chains are live: a chain's tokens do not depend on the batch it happens to share.
    int decode_regime = -1;
    bool wide_regime() const { return decode_regime == 1 || (decode_regime < 0 && cfg.max_seqs > 64); }
    std::vector<ze_text_layer> tl;

    // tables
    bf16_t *cosT = nullptr, *sinT = nullptr;
    int* axis_of = nullptr;
    float* lut = nullptr;
    int* eos_dev = nullptr;

    // KV cache + chain state
    bf16_t *kcache = nullptr, *vcache = nullptr;
    ze_seq_dev* st_dev = nullptr;
    uint8_t* seen = nullptr;
    int32_t* out_tokens = nullptr;
    std::vector<int> ctx_host, delta_host;
    std::vector<int> split_host;   // round 6: the chains' split rows (ze_seq_dev::split), host truth
    int* bmate = nullptr;          // [max_seqs] per row of the batched step: the row it shares its prefix parts with, or -1 (upload_batch)
    int live_parts_long = 0;       // 384-key parts of the batch's longest chain under its split (the pipelined attention's grid extent)
    // Shared-prefix hints, (source chain << 16) | P per chain slot.  pfx_host is the truth, kept by whatever call changes it
    // (ze_seq_copy_prefix on the admission stream, ze_seq_retire, ...); the device copy pfx_dev -- what the decode attention
    // reads -- is written by ONE stream only: the stream of the batched decode step, which pushes the words that differ from
    // pfx_pushed for ITS chains before it enqueues the step (sync_prefix, ze_forward.hip).  A chain state pushed from another
    // stream can therefore never bring a stale hint back.  pfx_copy_ev[seq]: recorded behind the chain's last
    // ze_seq_copy_prefix (null = none): only a chain whose copy has COMPLETED may become the holder other chains read from.
    std::vector<int> pfx_host, pfx_pushed;
    int* pfx_dev = nullptr;
    std::vector<hipEvent_t> pfx_copy_ev;
    bool prefix_hints = true;      // the hint fits its 16 + 16 bits (ze_tune knob 17 = 1: every chain reads its own rows)
    std::vector<hipGraphExec_t> graphs;
    std::vector<float> graph_penalty;
    std::vector<int> graph_ignore_eos;

    // front-end workspace
    uint8_t *fe_tmp = nullptr, *fe_img = nullptr;
    int *fe_coef = nullptr;  // device coefficient tables
    size_t fe_tmp_bytes = 0, fe_img_bytes = 0, fe_coef_ints = 0;
    int* fe_coef_host = nullptr;  // pinned
    hipEvent_t v_staged = nullptr, t_staged = nullptr;  // behind the last H2D copy out of v_host_* / t_host_ints (stage_acquire)
    hipEvent_t fe_done = nullptr;  // recorded behind the last kernel that reads the front-end workspace (any stream)
    bool fe_in_flight = false;

    // ViT workspace
    bf16_t *vx = nullptr, *vh = nullptr, *vy = nullptr, *vqkv = nullptr, *vo = nullptr, *va = nullptr, *vz = nullptr,
           *vz2 = nullptr;
    float *vcos = nullptr, *vsin = nullptr;
    int *vperm = nullptr, *vinv = nullptr;
    int4* vtiles_win = nullptr;
    int4* vtiles_full = nullptr;
    int* v_host_ints = nullptr;   // pinned staging for perm / tiles
    float* v_host_f32 = nullptr;  // pinned staging for cos/sin
    size_t v_host_ints_cap = 0, v_host_f32_cap = 0;

    // prefill workspace
    bf16_t *th = nullptr, *ty = nullptr, *tqkv = nullptr, *to = nullptr, *ta = nullptr;
    int *tsrc = nullptr, *tpos = nullptr;
    int4* ttiles = nullptr;
    int* ttile_aux = nullptr;  // batched prefill: (chain slot, position offset) per attention tile
    int* trow_aux = nullptr;   // batched prefill: (chain slot, cache position) per row
    int prefill_rows = 0;
    int* t_host_ints = nullptr;  // pinned
    size_t t_host_ints_cap = 0;
    // batched host <-> device id transfers of the scheduler (ze_seq_mark_seen_batch: host -> device on the admission stream;
    // ze_chain_tokens_batch: device -> host on the decode stream): one pinned + one device buffer per direction, max_seqs x
    // max_ctx ints, allocated at first use
    int *xs_host = nullptr, *xs_dev = nullptr, *xt_host = nullptr, *xt_dev = nullptr;
    size_t xs_cap = 0, xt_cap = 0;
    hipEvent_t xs_staged = nullptr;

    // decode workspace
    bf16_t *dh = nullptr, *dq = nullptr, *dattn = nullptr, *dact = nullptr;
    float *dlogits = nullptr, *dpartial = nullptr, *dsample = nullptr;
    int max_splits = 64;
    int* d_host_ints = nullptr;  // pinned, small
    unsigned* atickets = nullptr;  // decode attention: one arrival ticket per (chain, kv head)
    std::vector<int> graph_variant;  // ze_tune epoch the chain's graph was captured under
    unsigned bgraph_epoch = 0;
    std::vector<float> graph_temperature;
    std::vector<unsigned long long> graph_seed;
    // split-K GEMM workspace
    float* gslab = nullptr;       // split-K slabs of the weight-streaming GEMMs (ze_gemm_ws)
    size_t gslab_floats = 0;
    int gticket_cap = 0;
    ze_gemm_ws gemm_ws() const { return ze_gemm_ws{gslab, gslab_floats, gtickets, gticket_cap}; }
    unsigned* gtickets = nullptr;
    // split-K workspace of the PREFILL family (round 6: the down projection of a pass in three K slices, ze_gemm.hip ze_prefill_ksplit):
    // slabs and tickets of its own -- a prefill pass on the admission stream runs beside the decode step, which owns the ones above
    float* pslab = nullptr;
    size_t pslab_floats = 0;
    unsigned* ptickets = nullptr;
    int pticket_cap = 0;
    // (allocated on first use with the split switched on -- ze_tune knob 20 = 3: the form is not shipped, DESIGN 7i c, and 0.7 GB of
    //  slabs per engine are not reserved for it)
    ze_gemm_ws prefill_ws();
    // batched decode: activations of its own (rows = chains), so that a decode burst on one HIP stream and a prefill / ViT
    // round on another never share a buffer (the scheduler overlaps them: zoomearth_amd/scheduler.py)
    bf16_t *bh = nullptr, *by = nullptr, *bqkv = nullptr, *bo = nullptr, *ba = nullptr;
    int* bseq = nullptr;
    float *blogits = nullptr, *bpartial = nullptr, *bsample = nullptr;
    ze_seq_dev* bstate_host = nullptr;  // pinned
    std::map<std::tuple<int, float, int, float, unsigned long long, int>, hipGraphExec_t> bgraphs;  // captured batched decode step per batch size (and attention grid)
    int live_parts = 0;  // 192-key parts the longest chain of the current batch needs (the attention grid's extent); 0 = all

    // timers
    bool timers_on = false;
    bool counted = false;  // in ze_live_engines (a create that failed half-way is destroyed uncounted)
    struct ev_pair { int phase; hipEvent_t a, b; };
    std::vector<ev_pair> ev_used;
    std::vector<ev_pair> ev_free;
    float phase_ms[5] = {0, 0, 0, 0, 0};

    bf16_t* kc(int layer, int seq) const {
        return kcache + (((size_t)layer * cfg.max_seqs + seq) * cfg.kv_heads) * (size_t)cfg.max_ctx * head_dim;
    }
    bf16_t* vc(int layer, int seq) const {
        return vcache + (((size_t)layer * cfg.max_seqs + seq) * cfg.kv_heads) * (size_t)cfg.max_ctx * head_dim;
    }
};

// engine internals used across translation units
extern unsigned ze_tune_epoch;
void ze_weights_changed(ze_engine* e);
int ze_engine_build_layout(ze_engine* e);
int ze_timer_begin(ze_engine* e, int phase, hipStream_t s);
void ze_timer_end(ze_engine* e, int handle, hipStream_t s);
int ze_enqueue_decode_step(ze_engine* e, int seq, float penalty, int ignore_eos, bool sample, hipStream_t s);
