// Engine lifecycle, weight arena layout, checkpoint loading and synthetic fill.
#include <math.h>
#include <string.h>

#include "ze_engine.h"
#include "ze_prng.h"

thread_local std::string ze_global_error;
extern int ze_mrope_vec_ok;  // ze_elementwise.hip: cleared by an engine whose M-RoPE sections are not multiples of eight pairs
unsigned ze_tune_epoch = 0;  // bumped whenever captured decode graphs go stale (launch policy or weight streams changed)
extern int ze_live_engines;  // ze_gemv.hip (read by the launch policy of the eight-phase GEMM: persistent only while ONE engine owns the GPU)
static int ze_bound_device = -1;  // the launch-policy caches (hipFuncSetAttribute, CU count) are per process: one GPU per process

// Every weight mutation (load, synthetic fill, arena hand-out for a broadcast / RL refresh) invalidates the derived
// copies: the fragment-major decode copy is rebuilt on the next batched step, the fp8 stream is dropped (the caller
// quantises again) and captured decode graphs, which bake the weight pointers in, are re-captured.
void ze_weights_changed(ze_engine* e) {
    e->frag_ready = false;
    if (e->fp8_ready) {
        e->fp8_ready = false;
        e->fp8_act = false;  // goes with the fp8 weights: the caller quantises and switches it on again
        for (auto& L : e->tl)
            for (ze_linear* l : {&L.qkv, &L.o, &L.gate_up, &L.down}) {
                l->w8 = nullptr;
                l->scale8 = nullptr;
                l->wf8 = nullptr;
            }
        e->lm_head8.w8 = nullptr;
        e->lm_head8.scale8 = nullptr;
        e->lm_head8.wf8 = nullptr;
    }
    ++ze_tune_epoch;
}

int ze_fail(ze_engine* e, int code, const std::string& msg) {
    if (e) e->err = msg;
    ze_global_error = msg;
    return code;
}

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ------------------------------------------------------------------ layout
namespace {
struct layout_builder {
    ze_engine* e;
    size_t off = 0;
    bool dry;
    bf16_t* take(size_t elems) {
        off = align_up(off, 128);  // 256-B aligned rows of the arena
        bf16_t* p = dry ? nullptr : e->arena + off;
        off += elems;
        return p;
    }
    void lin(ze_linear& l, int n, int k, int ld, bool bias) {
        l.n = n;
        l.k = k;
        l.ld = ld;
        l.w = take((size_t)n * ld);
        l.bias = bias ? take(n) : nullptr;
    }
};

void add_dest(ze_engine* e, const std::string& name, bf16_t* dst, int rows, int cols, int ld, int mode, int offset,
              int kind) {
    ze_dest d;
    d.dst = dst;
    d.rows = rows;
    d.cols = cols;
    d.ld = ld;
    d.mode = mode;
    d.offset = offset;
    d.kind = kind;
    e->dests[name] = d;
}
}  // namespace

static size_t build(ze_engine* e, bool dry) {
    const ze_config& c = e->cfg;
    layout_builder b{e, 0, dry};
    e->dests.clear();
    const int vh = c.vit_hidden, ip = e->vit_ipad;
    const int pk = c.in_channels * c.temporal_patch_size * c.patch_size * c.patch_size;
    b.lin(e->patch_embed, vh, pk, pk, false);
    add_dest(e, "model.visual.patch_embed.proj.weight", e->patch_embed.w, vh, pk, pk, 0, 0, 0);
    e->vb.resize(c.vit_depth);
    for (int i = 0; i < c.vit_depth; ++i) {
        ze_vit_block& k = e->vb[i];
        const std::string p = "model.visual.blocks." + std::to_string(i) + ".";
        k.norm1 = b.take(vh);
        k.norm2 = b.take(vh);
        add_dest(e, p + "norm1.weight", k.norm1, vh, 1, 1, 0, 0, 1);
        add_dest(e, p + "norm2.weight", k.norm2, vh, 1, 1, 0, 0, 1);
        b.lin(k.qkv, 3 * vh, vh, vh, true);
        add_dest(e, p + "attn.qkv.weight", k.qkv.w, 3 * vh, vh, vh, 0, 0, 0);
        add_dest(e, p + "attn.qkv.bias", k.qkv.bias, 3 * vh, 1, 1, 0, 0, 2);
        b.lin(k.proj, vh, vh, vh, true);
        add_dest(e, p + "attn.proj.weight", k.proj.w, vh, vh, vh, 0, 0, 0);
        add_dest(e, p + "attn.proj.bias", k.proj.bias, vh, 1, 1, 0, 0, 2);
        b.lin(k.gate_up, 2 * ip, vh, vh, true);
        add_dest(e, p + "mlp.gate_proj.weight", k.gate_up.w, c.vit_intermediate, vh, vh, 1, 0, 0);
        add_dest(e, p + "mlp.up_proj.weight", k.gate_up.w, c.vit_intermediate, vh, vh, 1, 16, 0);
        add_dest(e, p + "mlp.gate_proj.bias", k.gate_up.bias, c.vit_intermediate, 1, 1, 1, 0, 2);
        add_dest(e, p + "mlp.up_proj.bias", k.gate_up.bias, c.vit_intermediate, 1, 1, 1, 16, 2);
        b.lin(k.down, vh, ip, ip, true);
        add_dest(e, p + "mlp.down_proj.weight", k.down.w, vh, c.vit_intermediate, ip, 0, 0, 0);
        add_dest(e, p + "mlp.down_proj.bias", k.down.bias, vh, 1, 1, 0, 0, 2);
    }
    const int mu = c.spatial_merge_size * c.spatial_merge_size;
    const int mh = vh * mu;
    e->ln_q = b.take(vh);
    add_dest(e, "model.visual.merger.ln_q.weight", e->ln_q, vh, 1, 1, 0, 0, 1);
    b.lin(e->merger0, mh, mh, mh, true);
    add_dest(e, "model.visual.merger.mlp.0.weight", e->merger0.w, mh, mh, mh, 0, 0, 0);
    add_dest(e, "model.visual.merger.mlp.0.bias", e->merger0.bias, mh, 1, 1, 0, 0, 2);
    b.lin(e->merger2, c.vit_out_hidden, mh, mh, true);
    add_dest(e, "model.visual.merger.mlp.2.weight", e->merger2.w, c.vit_out_hidden, mh, mh, 0, 0, 0);
    add_dest(e, "model.visual.merger.mlp.2.bias", e->merger2.bias, c.vit_out_hidden, 1, 1, 0, 0, 2);

    const int hd = e->head_dim, H = c.hidden, tip = e->text_ipad;
    const int nq = c.heads * hd, nkv = c.kv_heads * hd;
    e->embed = b.take((size_t)c.vocab * H);
    add_dest(e, "model.language_model.embed_tokens.weight", e->embed, c.vocab, H, H, 0, 0, 3);
    if (c.tie_word_embeddings) {
        e->lm_head = e->embed;
    } else {
        e->lm_head = b.take((size_t)c.vocab * H);
        add_dest(e, "lm_head.weight", e->lm_head, c.vocab, H, H, 0, 0, 3);
    }
    e->final_norm = b.take(H);
    add_dest(e, "model.language_model.norm.weight", e->final_norm, H, 1, 1, 0, 0, 1);
    e->tl.resize(c.layers);
    for (int i = 0; i < c.layers; ++i) {
        ze_text_layer& t = e->tl[i];
        const std::string p = "model.language_model.layers." + std::to_string(i) + ".";
        t.in_norm = b.take(H);
        t.post_norm = b.take(H);
        add_dest(e, p + "input_layernorm.weight", t.in_norm, H, 1, 1, 0, 0, 1);
        add_dest(e, p + "post_attention_layernorm.weight", t.post_norm, H, 1, 1, 0, 0, 1);
        b.lin(t.qkv, nq + 2 * nkv, H, H, true);
        add_dest(e, p + "self_attn.q_proj.weight", t.qkv.w, nq, H, H, 0, 0, 0);
        add_dest(e, p + "self_attn.k_proj.weight", dry ? nullptr : t.qkv.w + (size_t)nq * H, nkv, H, H, 0, 0, 0);
        add_dest(e, p + "self_attn.v_proj.weight", dry ? nullptr : t.qkv.w + (size_t)(nq + nkv) * H, nkv, H, H, 0, 0,
                 0);
        add_dest(e, p + "self_attn.q_proj.bias", t.qkv.bias, nq, 1, 1, 0, 0, 2);
        add_dest(e, p + "self_attn.k_proj.bias", dry ? nullptr : t.qkv.bias + nq, nkv, 1, 1, 0, 0, 2);
        add_dest(e, p + "self_attn.v_proj.bias", dry ? nullptr : t.qkv.bias + nq + nkv, nkv, 1, 1, 0, 0, 2);
        b.lin(t.o, H, nq, nq, false);
        add_dest(e, p + "self_attn.o_proj.weight", t.o.w, H, nq, nq, 0, 0, 0);
        b.lin(t.gate_up, 2 * tip, H, H, false);
        add_dest(e, p + "mlp.gate_proj.weight", t.gate_up.w, c.intermediate, H, H, 1, 0, 0);
        add_dest(e, p + "mlp.up_proj.weight", t.gate_up.w, c.intermediate, H, H, 1, 16, 0);
        b.lin(t.down, H, tip, tip, false);
        add_dest(e, p + "mlp.down_proj.weight", t.down.w, H, c.intermediate, tip, 0, 0, 0);
    }
    return align_up(b.off, 128);
}

int ze_engine_build_layout(ze_engine* e) {
    e->arena_elems = build(e, true);
    return 0;
}

template <typename T>
static int dev_alloc(ze_engine* e, T** p, size_t count, bool zero = true) {
    ZE_HIP(hipMalloc((void**)p, std::max<size_t>(count, 1) * sizeof(T)));
    if (zero) ZE_HIP(hipMemset(*p, 0, std::max<size_t>(count, 1) * sizeof(T)));
    return 0;
}
#define ZE_TRY(x)              \
    do {                       \
        int _r = (x);          \
        if (_r != 0) return _r; \
    } while (0)

static int init_tables(ze_engine* e) {
    const ze_config& c = e->cfg;
    const int half = e->head_dim / 2;
    // text rotary table: inv_freq fp32, freqs = pos * inv_freq (fp32), cos/sin fp32 -> bf16
    // (HF:models/qwen2_5_vl/modeling_qwen2_5_vl.py:486-538)
    std::vector<float> inv(half);
    for (int i = 0; i < half; ++i) inv[i] = 1.0f / powf(c.rope_theta, (float)(2 * i) / (float)e->head_dim);
    std::vector<bf16_t> ct((size_t)e->max_pos * half), st((size_t)e->max_pos * half);
    auto to_bf16 = [](float f) {
        uint32_t u;
        memcpy(&u, &f, 4);
        return (bf16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
    };
    for (int p = 0; p < e->max_pos; ++p)
        for (int i = 0; i < half; ++i) {
            const float f = (float)p * inv[i];
            ct[(size_t)p * half + i] = to_bf16(cosf(f));
            st[(size_t)p * half + i] = to_bf16(sinf(f));
        }
    ZE_TRY(dev_alloc(e, &e->cosT, ct.size(), false));
    ZE_TRY(dev_alloc(e, &e->sinT, st.size(), false));
    ZE_HIP(hipMemcpy(e->cosT, ct.data(), ct.size() * 2, hipMemcpyHostToDevice));
    ZE_HIP(hipMemcpy(e->sinT, st.data(), st.size() * 2, hipMemcpyHostToDevice));
    // mrope: rotary dim j (< half) takes its position from axis (section index % 3)
    std::vector<int> axis(half);
    {
        int o = 0;
        for (int s = 0; s < 3; ++s)
            for (int k = 0; k < c.mrope_section[s] && o < half; ++k) axis[o++] = s;
        for (; o < half; ++o) axis[o] = 0;
        // (the 16-byte form of k_mrope_kv gives a thread eight consecutive pairs: they have to share their axis)
        for (int g = 0; g + 8 <= half; g += 8)
            for (int k = 1; k < 8; ++k)
                if (axis[g + k] != axis[g]) ze_mrope_vec_ok = 0;
        if (half % 8) ze_mrope_vec_ok = 0;
    }
    ZE_TRY(dev_alloc(e, &e->axis_of, half, false));
    ZE_HIP(hipMemcpy(e->axis_of, axis.data(), half * sizeof(int), hipMemcpyHostToDevice));
    // normalise LUT: (f32(f64(v) * (1/255)) - f32(mean)) / f32(std)   (oracle/frontend.py normalize_lut)
    static const double mean[3] = {0.48145466, 0.4578275, 0.40821073};
    static const double sd[3] = {0.26862954, 0.26130258, 0.27577711};
    std::vector<float> lut(3 * 256);
    for (int ch = 0; ch < 3; ++ch)
        for (int v = 0; v < 256; ++v) {
            const float r = (float)((double)v * (1.0 / 255.0));
            lut[ch * 256 + v] = (r - (float)mean[ch]) / (float)sd[ch];
        }
    ZE_TRY(dev_alloc(e, &e->lut, lut.size(), false));
    ZE_HIP(hipMemcpy(e->lut, lut.data(), lut.size() * sizeof(float), hipMemcpyHostToDevice));
    ZE_TRY(dev_alloc(e, &e->eos_dev, ZE_MAX_EOS, true));
    ZE_HIP(hipMemcpy(e->eos_dev, c.eos_token_ids, sizeof(int) * c.n_eos, hipMemcpyHostToDevice));
    return 0;
}

extern "C" int ze_version(void) { return 100; }

extern "C" const char* ze_last_error(const ze_engine* e) { return e ? e->err.c_str() : ze_global_error.c_str(); }

extern "C" int ze_engine_create(const ze_config* cfg, int device_id, ze_engine** out) {
    if (!cfg || !out) return ze_fail(nullptr, ZE_ERR_INVALID, "null argument");
    ze_engine* e = new ze_engine();
    e->cfg = *cfg;
    e->device = device_id;
    const ze_config& c = e->cfg;
    auto bad = [&](const char* m) {
        const int r = ze_fail(nullptr, ZE_ERR_INVALID, m);
        delete e;
        return r;
    };
    if (c.heads <= 0 || c.hidden % c.heads) return bad("hidden must be divisible by heads");
    if (c.vit_heads <= 0 || c.vit_hidden % c.vit_heads) return bad("vit_hidden must be divisible by vit_heads (> 0)");
    if (c.kv_heads <= 0) return bad("kv_heads must be positive");
    if (ze_bound_device >= 0 && ze_bound_device != device_id)
        return bad("this process already drives another GPU: one process per GPU (launch attributes are cached per process)");
    e->head_dim = c.hidden / c.heads;
    e->vit_head_dim = c.vit_hidden / c.vit_heads;
    if (e->head_dim != 128) return bad("text head_dim must be 128");
    if (e->vit_head_dim != 80 && e->vit_head_dim != 128) return bad("vision head_dim must be 80 or 128");
    if (c.heads % c.kv_heads || c.heads / c.kv_heads > 8) return bad("GQA group must divide heads and be <= 8");
    if (c.vit_hidden % 8 || c.hidden % 8 || c.vocab % 2) return bad("hidden sizes must be multiples of 8");
    if (c.n_fullatt > ZE_MAX_FULLATT || c.n_eos > ZE_MAX_EOS) return bad("too many fullatt blocks / eos ids");
    if (c.mrope_section[0] + c.mrope_section[1] + c.mrope_section[2] != e->head_dim / 2)
        return bad("mrope_section must sum to head_dim/2");
    if (c.max_seqs <= 0 || c.max_ctx <= 0 || c.max_patches <= 0) return bad("capacities must be positive");
    if (c.max_seqs > c.max_ctx) return bad("max_seqs must not exceed max_ctx");
    // the decode GEMV stages x in LDS (<= 60 KB)
    if (ze_pad32(c.intermediate) > 29000) return bad("intermediate > 29000 unsupported by the decode GEMV");
    // MLP width padded to the GEMM K-step (64): zero rows / columns in the packed weights, so the down projection
    // takes the LDS-DMA ring kernel (K % 64 == 0) for the ViT too (3420 -> 3456)
    // flash-decoding slices are 64 tokens: no context needs more than max_ctx / 64 of them (the launch grid covers
    // max_splits slices per kv head, so a smaller bound also means fewer idle workgroups per launch)
    e->max_splits = std::max(1, std::min(64, (c.max_ctx + 63) / 64));
    e->vit_ipad = (c.vit_intermediate + 63) / 64 * 64;
    e->text_ipad = (c.intermediate + 63) / 64 * 64;
    e->max_pos = c.max_ctx + 512;

    if (hipSetDevice(device_id) != hipSuccess) return bad("hipSetDevice failed");
    ze_bound_device = device_id;
    ze_engine_build_layout(e);
    int r = 0;
    auto chk = [&](int rr) {
        if (rr != 0 && r == 0) r = rr;
    };
    chk(dev_alloc(e, &e->arena, e->arena_elems, true));
    if (r == 0) {
        e->arena_used = build(e, false);
        chk(init_tables(e));
    }
    const size_t kv_elems = (size_t)c.layers * c.max_seqs * c.kv_heads * c.max_ctx * e->head_dim;
    chk(dev_alloc(e, &e->kcache, kv_elems, false));
    chk(dev_alloc(e, &e->vcache, kv_elems, false));
    chk(dev_alloc(e, &e->st_dev, c.max_seqs));
    chk(dev_alloc(e, &e->seen, (size_t)c.max_seqs * c.vocab));
    chk(dev_alloc(e, &e->out_tokens, (size_t)c.max_seqs * c.max_ctx));
    e->ctx_host.assign(c.max_seqs, 0);
    e->pfx_host.assign(c.max_seqs, 0);
    e->pfx_pushed.assign(c.max_seqs, 0);
    e->pfx_copy_ev.assign(c.max_seqs, nullptr);
    chk(dev_alloc(e, &e->pfx_dev, c.max_seqs));
    e->prefix_hints = c.max_seqs < 32768 && c.max_ctx < 65536;  // (ze_tune knob 17 = 1: every chain reads its own rows, for A/B runs)
    e->delta_host.assign(c.max_seqs, 0);
    e->split_host.assign(c.max_seqs, 0);
    e->graphs.assign(c.max_seqs, nullptr);
    e->graph_penalty.assign(c.max_seqs, 0.f);
    e->graph_ignore_eos.assign(c.max_seqs, 0);
    e->graph_variant.assign(c.max_seqs, 0);
    e->graph_temperature.assign(c.max_seqs, 0.f);
    e->graph_seed.assign(c.max_seqs, 0ull);

    // front-end workspace: horizontal-pass image (box_h x out_w) and resized image
    const size_t side = (size_t)std::max(c.max_tile_side, 1024);
    e->fe_tmp_bytes = side * 4096 * 3;
    e->fe_img_bytes = (size_t)4096 * 4096 * 3;
    chk(dev_alloc(e, &e->fe_tmp, e->fe_tmp_bytes, false));
    chk(dev_alloc(e, &e->fe_img, e->fe_img_bytes, false));
    e->fe_coef_ints = (size_t)4096 * 96 * 2 + 4096 * 4;
    chk(dev_alloc(e, &e->fe_coef, e->fe_coef_ints, false));
    if (r == 0 && hipHostMalloc((void**)&e->fe_coef_host, e->fe_coef_ints * sizeof(int)) != hipSuccess)
        r = ze_fail(e, ZE_ERR_HIP, "hipHostMalloc failed");

    // ViT workspace
    const size_t np = c.max_patches;
    const int vh = c.vit_hidden, mu = c.spatial_merge_size * c.spatial_merge_size;
    const int pk = c.in_channels * c.temporal_patch_size * c.patch_size * c.patch_size;
    chk(dev_alloc(e, &e->vx, np * pk));
    chk(dev_alloc(e, &e->vh, np * vh));
    chk(dev_alloc(e, &e->vy, np * vh));
    chk(dev_alloc(e, &e->vqkv, np * 3 * vh));
    chk(dev_alloc(e, &e->vo, np * vh));
    chk(dev_alloc(e, &e->va, np * e->vit_ipad));
    chk(dev_alloc(e, &e->vz, np / mu * (size_t)vh * mu + 8));
    chk(dev_alloc(e, &e->vz2, np / mu * (size_t)c.vit_out_hidden + 8));
    chk(dev_alloc(e, &e->vcos, np * (e->vit_head_dim / 2)));
    chk(dev_alloc(e, &e->vsin, np * (e->vit_head_dim / 2)));
    chk(dev_alloc(e, &e->vperm, np));
    chk(dev_alloc(e, &e->vinv, np));
    chk(dev_alloc(e, &e->vtiles_win, np));
    chk(dev_alloc(e, &e->vtiles_full, np));
    e->v_host_ints_cap = np * 2 + np * 4 * 2 + 64;
    e->v_host_f32_cap = np * e->vit_head_dim;
    if (r == 0 && (hipHostMalloc((void**)&e->v_host_ints, e->v_host_ints_cap * sizeof(int)) != hipSuccess ||
                   hipHostMalloc((void**)&e->v_host_f32, e->v_host_f32_cap * sizeof(float)) != hipSuccess))
        r = ze_fail(e, ZE_ERR_HIP, "hipHostMalloc failed");

    // prefill workspace
    const size_t tm = std::max(c.max_ctx, c.max_prefill_rows);  // rows of one prefill pass
    e->prefill_rows = (int)tm;
    const int nqkv = (c.heads + 2 * c.kv_heads) * e->head_dim;
    chk(dev_alloc(e, &e->th, tm * c.hidden));
    chk(dev_alloc(e, &e->ty, tm * c.hidden));
    chk(dev_alloc(e, &e->ty8p, tm * c.hidden, false));
    chk(dev_alloc(e, &e->ty8p_scale, tm, false));
    chk(dev_alloc(e, &e->damax, 2 * 2048));
    if (r == 0 && e->damax) ze_launch_amax_init(e->damax, nullptr);
    chk(dev_alloc(e, &e->ty8, (size_t)64 * c.hidden));
    chk(dev_alloc(e, &e->ty8_scale, 64));
    chk(dev_alloc(e, &e->tqkv, tm * nqkv));
    chk(dev_alloc(e, &e->to, tm * c.heads * e->head_dim));
    chk(dev_alloc(e, &e->ta, tm * e->text_ipad));
    chk(dev_alloc(e, &e->tsrc, tm));
    chk(dev_alloc(e, &e->tpos, tm * 3));
    const size_t ntile_cap = tm / 64 + c.max_seqs + 2;
    chk(dev_alloc(e, &e->ttiles, ntile_cap));
    chk(dev_alloc(e, &e->ttile_aux, ntile_cap * 2));
    chk(dev_alloc(e, &e->trow_aux, tm * 2));
    e->t_host_ints_cap = tm * 6 + ntile_cap * 6 + 64;
    if (r == 0 && hipHostMalloc((void**)&e->t_host_ints, e->t_host_ints_cap * sizeof(int)) != hipSuccess)
        r = ze_fail(e, ZE_ERR_HIP, "hipHostMalloc failed");

    // decode workspace
    chk(dev_alloc(e, &e->dh, c.hidden));
    chk(dev_alloc(e, &e->dq, (size_t)c.heads * e->head_dim));
    chk(dev_alloc(e, &e->dattn, (size_t)c.heads * e->head_dim));
    chk(dev_alloc(e, &e->dact, e->text_ipad));
    chk(dev_alloc(e, &e->dlogits, (size_t)c.max_seqs * c.vocab));
    {
        // fp32 split-K slabs of the weight-streaming GEMMs, sized for EVERY row count this engine can see: the slice count is a
        // function of (N, K) alone (launch_cfg: ksplit * ceil(N / 64) < 400 whenever it splits), rows <= max_seqs padded to the
        // largest row tile -- so the split, and with it the order of a chain's sums, never follows the row count (ADVICE r3:
        // with a fixed 64 MB the 3B shape sat exactly at the limit at 1024 rows and larger shapes would have dropped to one
        // slice above some batch size).  64 MB at least (the unit-op entry points call with shapes of their own).
        const size_t rows_pad = ((size_t)std::max(c.max_seqs, 64) + 255) / 256 * 256;
        const size_t slab_floats = std::max((size_t)24 << 20, (size_t)400 * 64 * rows_pad);   // (floor: unit ops of up to ~1400 rows x 2048 columns in eight slices)
        const int ticket_cap = std::max(4096, (int)(200 * (rows_pad / 64)) + 64);
        chk(dev_alloc(e, &e->gslab, slab_floats, false));
        chk(dev_alloc(e, &e->gtickets, (size_t)ticket_cap, true));
        e->gslab_floats = slab_floats;
        e->gticket_cap = ticket_cap;
    }
    {
        const size_t br = (size_t)(std::max(c.max_seqs, 64) + 63) / 64 * 64;  // rows of the batched step (whole 64-row tiles)
        chk(dev_alloc(e, &e->bh, br * c.hidden));
        chk(dev_alloc(e, &e->by, br * c.hidden));
        chk(dev_alloc(e, &e->bqkv, br * nqkv));
        chk(dev_alloc(e, &e->bo, br * c.heads * e->head_dim));
        chk(dev_alloc(e, &e->ba, br * e->text_ipad));
    }
    chk(dev_alloc(e, &e->bseq, c.max_seqs));
    chk(dev_alloc(e, &e->bmate, c.max_seqs));
    chk(dev_alloc(e, &e->blogits, (size_t)c.max_seqs * c.vocab));
    chk(dev_alloc(e, &e->bpartial, (size_t)c.max_seqs * std::max(e->max_splits, 8) * c.heads * 132));  // >= 8 parts per chain (ze_attn_batch.hip)
    chk(dev_alloc(e, &e->bsample, (size_t)c.max_seqs * 3 * 128 + 8));  // arg-max partials, then chunk sums
    if (r == 0 && hipHostMalloc((void**)&e->bstate_host, sizeof(ze_seq_dev) * c.max_seqs) != hipSuccess)
        r = ze_fail(e, ZE_ERR_HIP, "hipHostMalloc failed");
    chk(dev_alloc(e, &e->dpartial, (size_t)e->max_splits * c.heads * 132));
    chk(dev_alloc(e, &e->dsample, 2 * 128 + 64 + 128 + 8));  // arg-max partials, spare, chunk sums
    chk(dev_alloc(e, &e->atickets, (size_t)c.max_seqs * c.kv_heads));
    if (r == 0 && hipHostMalloc((void**)&e->d_host_ints, (64 + c.max_seqs) * sizeof(int)) != hipSuccess)
        r = ze_fail(e, ZE_ERR_HIP, "hipHostMalloc failed");
    e->staging_bytes = (size_t)64 << 20;
    if (r == 0 && hipMalloc(&e->staging, e->staging_bytes) != hipSuccess) r = ze_fail(e, ZE_ERR_HIP, "hipMalloc staging");

    if (r != 0) {
        ze_global_error = e->err;
        ze_engine_destroy(e);
        return r;
    }
    e->counted = true;
    __atomic_add_fetch(&ze_live_engines, 1, __ATOMIC_RELAXED);
    *out = e;
    return ZE_OK;
}

extern int ze_gemv_knobs[24];
// prefill split-K slabs: 3 slices x (rows of the largest pass, padded to a 256-row tile) x (hidden, padded to a 256-column tile) floats --
// the same whatever tile a row count selects -- and a zeroed ticket per 64 x 64 tile (the smallest).  Only with knob 20 = 3.
ze_gemm_ws ze_engine::prefill_ws() {
    if (!pslab && ze_gemv_knobs[20] == 3) {
        const size_t rows_pad = ((size_t)prefill_rows + 255) / 256 * 256, cols_pad = ((size_t)cfg.hidden + 255) / 256 * 256;
        const size_t floats = 3 * rows_pad * cols_pad;
        const int cap = (int)((rows_pad / 64) * (cols_pad / 64)) + 64;
        float* sl = nullptr;
        unsigned* tk = nullptr;
        hipSetDevice(device);
        if (hipMalloc((void**)&sl, floats * sizeof(float)) == hipSuccess && hipMalloc((void**)&tk, (size_t)cap * sizeof(unsigned)) == hipSuccess &&
            hipMemset(tk, 0, (size_t)cap * sizeof(unsigned)) == hipSuccess) {
            pslab = sl, pslab_floats = floats, ptickets = tk, pticket_cap = cap;
        } else {
            if (sl) hipFree(sl);
            if (tk) hipFree(tk);
        }
    }
    return ze_gemm_ws{pslab, pslab_floats, ptickets, pticket_cap};
}

extern "C" int ze_engine_destroy(ze_engine* e) {
    if (!e) return ZE_OK;
    if (e->counted) __atomic_sub_fetch(&ze_live_engines, 1, __ATOMIC_RELAXED);
    hipSetDevice(e->device);
    hipDeviceSynchronize();
    for (auto g : e->graphs)
        if (g) hipGraphExecDestroy(g);
    for (auto& kv : e->bgraphs)
        if (kv.second) hipGraphExecDestroy(kv.second);
    for (auto& p : e->ev_used) {
        hipEventDestroy(p.a);
        hipEventDestroy(p.b);
    }
    for (auto& p : e->ev_free) {
        hipEventDestroy(p.a);
        hipEventDestroy(p.b);
    }
    void* dev[] = {e->arena, e->staging, e->cosT, e->sinT, e->axis_of, e->lut, e->eos_dev, e->kcache, e->vcache,
                   e->st_dev, e->seen, e->out_tokens, e->fe_tmp, e->fe_img, e->fe_coef, e->vx, e->vh, e->vy, e->vqkv,
                   e->vo, e->va, e->vz, e->vz2, e->vcos, e->vsin, e->vperm, e->vinv, e->vtiles_win, e->vtiles_full,
                   e->th, e->ty, e->tqkv, e->to, e->ta, e->tsrc, e->tpos, e->ttiles, e->ttile_aux, e->trow_aux, e->dh, e->dq, e->dattn, e->dact,
                   e->dlogits, e->dpartial, e->dsample, e->atickets, e->gslab, e->gtickets, e->pslab, e->ptickets, e->bh, e->by, e->bqkv, e->bo, e->ba, e->bseq, e->bmate, e->blogits, e->bpartial, e->bsample, e->arena8, e->arena_f, e->arena_f8,
                   e->ty8, e->ty8_scale, e->damax, e->ty8p, e->ty8p_scale};
    for (void* p : dev)
        if (p) hipFree(p);
    if (e->pfx_dev) hipFree(e->pfx_dev);
    if (e->arena_p) hipFree(e->arena_p);
    if (e->qkv_epi_dev) hipFree(e->qkv_epi_dev);
    for (hipEvent_t ev : e->pfx_copy_ev)
        if (ev) hipEventDestroy(ev);
    if (e->fe_done) hipEventDestroy(e->fe_done);
    if (e->v_staged) hipEventDestroy(e->v_staged);
    if (e->t_staged) hipEventDestroy(e->t_staged);
    if (e->xs_staged) hipEventDestroy(e->xs_staged);
    if (e->xs_dev) hipFree(e->xs_dev);
    if (e->xt_dev) hipFree(e->xt_dev);
    void* host[] = {e->fe_coef_host, e->v_host_ints, e->v_host_f32, e->t_host_ints, e->d_host_ints, e->bstate_host, e->xs_host, e->xt_host};
    for (void* p : host)
        if (p) hipHostFree(p);
    delete e;
    return ZE_OK;
}

extern "C" int ze_sync(ze_engine* e, void* stream) {
    ZE_HIP(hipStreamSynchronize((hipStream_t)stream));
    return ZE_OK;
}

// ------------------------------------------------------------------ weights
static std::string canonical_name(const char* name) {
    std::string n(name);
    if (n.rfind("visual.", 0) == 0) return "model." + n;  // 4.49-era layout
    if (n.rfind("model.layers.", 0) == 0 || n.rfind("model.embed_tokens.", 0) == 0 || n.rfind("model.norm.", 0) == 0)
        return "model.language_model." + n.substr(6);
    return n;
}

extern "C" int ze_load_weight(ze_engine* e, const char* name, int dtype, int ndim, const int64_t* shape,
                              const void* host_ptr) {
    if (e) ze_weights_changed(e);
    if (!e || !name || !shape || !host_ptr) return ze_fail(e, ZE_ERR_INVALID, "null argument");
    const std::string cn = canonical_name(name);
    if (cn == "lm_head.weight" && e->cfg.tie_word_embeddings) {
        e->loaded.insert(cn);
        return ZE_OK;  // tied: embed_tokens is the lm_head
    }
    auto it = e->dests.find(cn);
    if (it == e->dests.end()) return ze_fail(e, ZE_ERR_NOTFOUND, std::string("unknown weight: ") + name);
    const ze_dest& d = it->second;
    int64_t numel = 1;
    for (int i = 0; i < ndim; ++i) numel *= shape[i];
    if (numel != (int64_t)d.rows * d.cols || (ndim >= 1 && shape[0] != d.rows))
        return ze_fail(e, ZE_ERR_INVALID, std::string("shape mismatch for ") + name);
    const size_t esz = dtype == ZE_F32 ? 4 : 2;
    if (dtype != ZE_F32 && dtype != ZE_F16 && dtype != ZE_BF16) return ze_fail(e, ZE_ERR_INVALID, "bad dtype");
    hipSetDevice(e->device);
    const int rows_per = (int)std::max<size_t>(1, e->staging_bytes / ((size_t)d.cols * esz));
    for (int r0 = 0; r0 < d.rows; r0 += rows_per) {
        const int nr = std::min(rows_per, d.rows - r0);
        ZE_HIP(hipMemcpy(e->staging, (const uint8_t*)host_ptr + (size_t)r0 * d.cols * esz, (size_t)nr * d.cols * esz,
                         hipMemcpyHostToDevice));
        ze_launch_pack_rows(e->staging, dtype, r0, nr, d.cols, d.dst, d.ld, d.mode, d.offset, 0);
        ZE_HIP(hipStreamSynchronize(0));
    }
    e->loaded.insert(cn);
    return ZE_OK;
}

extern "C" int ze_weights_fill_synthetic(ze_engine* e, uint64_t seed, float std_, float matrix_gain, float bias_std,
                                         float norm_jitter) {
    if (e) ze_weights_changed(e);
    if (!e) return ze_fail(e, ZE_ERR_INVALID, "null engine");
    hipSetDevice(e->device);
    for (auto& kv : e->dests) {
        const ze_dest& d = kv.second;
        const uint64_t ts = ze_tensor_seed(seed, kv.first.c_str());
        float sd = 0.f, base = 0.f;
        switch (d.kind) {
            case 0: sd = std_ * matrix_gain; break;
            case 1: sd = norm_jitter; base = 1.0f; break;
            case 2: sd = bias_std; break;
            default: sd = std_; break;
        }
        const float cs = sd > 0.f ? (float)((double)sd / ZE_IH4_STD) : 0.f;
        ze_launch_fill_rows(ts, cs, base, d.rows, d.cols, d.dst, d.ld, d.mode, d.offset, 0);
        e->loaded.insert(kv.first);
    }
    ZE_HIP(hipStreamSynchronize(0));
    ZE_HIP(hipGetLastError());
    return ZE_OK;
}

extern "C" int ze_weights_missing(ze_engine* e) {
    if (!e) return ze_fail(e, ZE_ERR_INVALID, "null engine");
    int n = 0;
    std::string names;
    for (auto& kv : e->dests)
        if (!e->loaded.count(kv.first)) {
            if (n < 8) names += (n ? ", " : "") + kv.first;
            ++n;
        }
    if (n) e->err = "missing weights: " + names + (n > 8 ? ", ..." : "");
    return n;
}

extern "C" int ze_weights_arena(ze_engine* e, void** dev_ptr, size_t* bytes) {
    if (!e || !dev_ptr || !bytes) return ze_fail(e, ZE_ERR_INVALID, "null argument");
    *dev_ptr = e->arena;
    *bytes = e->arena_used * sizeof(bf16_t);
    return ZE_OK;
}

// A caller that WROTE the whole arena through the pointer above (the broadcast that replaces the other ranks' own
// from_pretrained, a weight refresh) says so here: every tensor counts as loaded (a receiving rank never called
// ze_load_weight), the derived copies (fragment-major, FP8) and the captured graphs are dropped and rebuilt from the new
// values on the next use.
extern "C" int ze_weights_invalidate(ze_engine* e) {
    if (!e) return ze_fail(e, ZE_ERR_INVALID, "null engine");
    for (auto& kv : e->dests) e->loaded.insert(kv.first);
    ze_weights_changed(e);
    return ZE_OK;
}

// RCCL broadcast of the arena for hosts that own an ncclComm_t (a C++ launcher; the Python shim goes through
// torch.distributed, whose communicator is not exposed).  The library does not link RCCL: the symbol is taken from
// whatever RCCL the process already loaded (dlsym), so there is never a second copy of it in the address space.
#include <dlfcn.h>
extern "C" int ze_weights_broadcast(ze_engine* e, void* nccl_comm, int root, void* stream) {
    if (!e || !nccl_comm || root < 0) return ze_fail(e, ZE_ERR_INVALID, "null engine / communicator or negative root");
    typedef int (*bcast_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
    static bcast_fn fn = nullptr;
    if (!fn) fn = reinterpret_cast<bcast_fn>(dlsym(RTLD_DEFAULT, "ncclBroadcast"));
    if (!fn) return ze_fail(e, ZE_ERR_NOTFOUND, "ncclBroadcast is not loaded in this process (load librccl before calling)");
    hipSetDevice(e->device);
    ze_weights_changed(e);
    const size_t bytes = e->arena_used * sizeof(bf16_t);
    const int rc = fn(e->arena, e->arena, bytes, /*ncclUint8*/ 1, root, nccl_comm, (hipStream_t)stream);
    if (rc != 0) return ze_fail(e, ZE_ERR_HIP, "ncclBroadcast failed with code " + std::to_string(rc));
    return ZE_OK;
}

// ------------------------------------------------------------------ phase timers
int ze_timer_begin(ze_engine* e, int phase, hipStream_t s) {
    if (!e->timers_on) return -1;
    ze_engine::ev_pair p;
    if (!e->ev_free.empty()) {
        p = e->ev_free.back();
        e->ev_free.pop_back();
    } else {
        hipEventCreate(&p.a);
        hipEventCreate(&p.b);
    }
    p.phase = phase;
    hipEventRecord(p.a, s);
    e->ev_used.push_back(p);
    return (int)e->ev_used.size() - 1;
}
void ze_timer_end(ze_engine* e, int handle, hipStream_t s) {
    if (handle < 0) return;
    hipEventRecord(e->ev_used[handle].b, s);
}
extern "C" int ze_phase_timers(ze_engine* e, int enable, int reset, float out_ms[5]) {
    if (!e) return ze_fail(e, ZE_ERR_INVALID, "null engine");
    for (auto& p : e->ev_used) {
        float ms = 0.f;
        hipEventSynchronize(p.b);
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) e->phase_ms[p.phase] += ms;
        e->ev_free.push_back(p);
    }
    e->ev_used.clear();
    if (out_ms)
        for (int i = 0; i < 5; ++i) out_ms[i] = e->phase_ms[i];
    if (reset)
        for (int i = 0; i < 5; ++i) e->phase_ms[i] = 0.f;
    e->timers_on = enable != 0;
    return ZE_OK;
}
