// Forward passes of the engine: image front-end, ViT, LLM prefill, decode step (eager or hipGraph), greedy
// generation, plus the unit ops and the measurement hooks of the C ABI.
#include <math.h>
#include <string.h>

#include <algorithm>

#include "ze_engine.h"

#define ZE_TRY(x)               \
    do {                        \
        int _r = (x);           \
        if (_r != 0) return _r; \
    } while (0)
#define ZE_KCHECK() ZE_HIP(hipGetLastError())
extern int ze_gemv_knobs[24];

// ================================================================== front-end
// dst = crop(src, box).resize((dst_w, dst_h), BICUBIC), Pillow-exact (two passes, u8 intermediate).
static int frontend_acquire(ze_engine* e) {
    if (e->fe_in_flight) {
        ZE_HIP(hipEventSynchronize(e->fe_done));
        e->fe_in_flight = false;
    }
    return ZE_OK;
}
// called behind the last kernel that reads the workspace
static int frontend_release(ze_engine* e, hipStream_t s) {
    if (!e->fe_done) ZE_HIP(hipEventCreateWithFlags(&e->fe_done, hipEventDisableTiming));
    ZE_HIP(hipEventRecord(e->fe_done, s));
    e->fe_in_flight = true;
    return ZE_OK;
}
// A pinned staging buffer is rewritten by the next call: wait until the copies that read it have run -- an event right behind
// them, not the whole stream (which holds the ViT / prefill kernels enqueued since: the scheduler's thread used to sit 20-40 ms
// in every vit_forward / prefill_batch call, and with bursts shorter than that -- 64 chain slots -- the decode stream idled)
static int stage_acquire(ze_engine* e, hipEvent_t& ev) {
    if (ev) ZE_HIP(hipEventSynchronize(ev));
    return ZE_OK;
}
static int stage_release(ze_engine* e, hipEvent_t& ev, hipStream_t s) {
    if (!ev) ZE_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    ZE_HIP(hipEventRecord(ev, s));
    return ZE_OK;
}
static int crop_resize(ze_engine* e, const uint8_t* src, int src_h, int src_w, const int32_t box[4], uint8_t* dst,
                       int dst_h, int dst_w, hipStream_t s) {
    const int bx0 = box[0], by0 = box[1];
    const int bw = box[2] - box[0], bh = box[3] - box[1];
    if (bw <= 0 || bh <= 0 || dst_w <= 0 || dst_h <= 0) return ze_fail(e, ZE_ERR_INVALID, "empty crop box or output");
    const bool need_h = bw != dst_w, need_v = bh != dst_h;
    if (!need_h && !need_v) {
        ze_launch_crop(src, src_h, src_w, bx0, by0, dst, dst_h, dst_w, s);
        ZE_KCHECK();
        return ZE_OK;
    }
    ze_coeffs ch, cv;
    size_t ints = 0;
    if (need_h) {
        ze_bicubic_coeffs(bw, dst_w, &ch);
        ints += 2 * (size_t)dst_w + ch.kk.size();
    }
    if (need_v) {
        ze_bicubic_coeffs(bh, dst_h, &cv);
        ints += 2 * (size_t)dst_h + cv.kk.size();
    }
    if (ints > e->fe_coef_ints) return ze_fail(e, ZE_ERR_NOMEM, "bicubic coefficient table exceeds workspace");
    if (need_h && need_v && (size_t)bh * dst_w * 3 > e->fe_tmp_bytes)
        return ze_fail(e, ZE_ERR_NOMEM, "resize intermediate exceeds workspace (max_tile_side)");
    // ONE workspace per engine (coefficient tables, their pinned staging buffer, the horizontal-pass image, the resized
    // image) and callers on several streams (the scheduler crops on its side stream while the loop that feeds it resizes
    // the next tile's view on its own): wait for the previous front-end op, whatever stream it ran on
    ZE_TRY(frontend_acquire(e));
    int* hp = e->fe_coef_host;
    int* dp = e->fe_coef;
    size_t o = 0;
    const int *d_xmin = nullptr, *d_xcnt = nullptr, *d_xk = nullptr, *d_ymin = nullptr, *d_ycnt = nullptr,
              *d_yk = nullptr;
    auto put = [&](const std::vector<int>& v) {
        memcpy(hp + o, v.data(), v.size() * sizeof(int));
        const int* d = dp + o;
        o += v.size();
        return d;
    };
    if (need_h) {
        d_xmin = put(ch.xmin);
        d_xcnt = put(ch.xcnt);
        d_xk = put(ch.kk);
    }
    if (need_v) {
        d_ymin = put(cv.xmin);
        d_ycnt = put(cv.xcnt);
        d_yk = put(cv.kk);
    }
    ZE_HIP(hipMemcpyAsync(dp, hp, o * sizeof(int), hipMemcpyHostToDevice, s));
    if (need_h) {
        uint8_t* hout = need_v ? e->fe_tmp : dst;
        // widest source span of a 64-column block: 64*scale + 2*support  (+ slack)
        int max_span = 0;
        for (int c0 = 0; c0 < dst_w; c0 += 64) {
            const int last = std::min(dst_w, c0 + 64) - 1;
            max_span = std::max(max_span, ch.xmin[last] + ch.xcnt[last] - ch.xmin[c0]);
        }
        const int inside = bx0 >= 0 && by0 >= 0 && bx0 + bw <= src_w && by0 + bh <= src_h;
        ze_launch_resize_h(src, src_h, src_w, bx0, by0, bh, hout, dst_w, d_xmin, d_xcnt, d_xk, ch.ksize, max_span, s, inside);
        ZE_KCHECK();
        if (need_v) {
            ze_launch_resize_v(e->fe_tmp, bh, dst_w, 0, 0, dst_w * 3, dst, dst_h, d_ymin, d_ycnt, d_yk, cv.ksize, 0, s);
            ZE_KCHECK();
        }
    } else {
        ze_launch_resize_v(src, src_h, src_w, bx0, by0, dst_w * 3, dst, dst_h, d_ymin, d_ycnt, d_yk, cv.ksize, 1, s);
        ZE_KCHECK();
    }
    return ZE_OK;
}

extern "C" int ze_tile_upload(ze_engine* e, const uint8_t* host_rgb, int h, int w, uint8_t* dev_rgb, void* stream) {
    if (!e || !host_rgb || !dev_rgb || h <= 0 || w <= 0) return ze_fail(e, ZE_ERR_INVALID, "bad tile_upload arguments");
    hipSetDevice(e->device);
    ZE_HIP(hipMemcpyAsync(dev_rgb, host_rgb, (size_t)h * w * 3, hipMemcpyHostToDevice, (hipStream_t)stream));
    return ZE_OK;
}

extern "C" int ze_op_crop_resize(ze_engine* e, const uint8_t* src, int src_h, int src_w, const int32_t box[4],
                                 uint8_t* dst, int dst_h, int dst_w, void* stream) {
    if (!e || !src || !dst || !box) return ze_fail(e, ZE_ERR_INVALID, "null argument");
    hipSetDevice(e->device);
    hipStream_t s = (hipStream_t)stream;
    const int h = ze_timer_begin(e, 0, s);
    int r = crop_resize(e, src, src_h, src_w, box, dst, dst_h, dst_w, s);
    if (r == ZE_OK) r = frontend_release(e, s);
    ze_timer_end(e, h, s);
    return r;
}

extern "C" int ze_smart_resize(int height, int width, int factor, int64_t min_pixels, int64_t max_pixels, int* out_h,
                               int* out_w) {
    if (ze_smart_resize_impl(height, width, factor, min_pixels, max_pixels, out_h, out_w) != 0)
        return ze_fail(nullptr, ZE_ERR_INVALID, "absolute aspect ratio must be smaller than 200");
    return ZE_OK;
}

extern "C" int ze_op_patchify(ze_engine* e, const uint8_t* img, int h, int w, float* out, void* stream) {
    if (!e || !img || !out) return ze_fail(e, ZE_ERR_INVALID, "null argument");
    const ze_config& c = e->cfg;
    const int f = c.patch_size * c.spatial_merge_size;
    if (h % f || w % f) return ze_fail(e, ZE_ERR_INVALID, "image size must be a multiple of patch*merge");
    hipSetDevice(e->device);
    ze_launch_patchify(img, h, w, e->lut, out, c.patch_size, c.spatial_merge_size, c.temporal_patch_size,
                       c.in_channels, (hipStream_t)stream);
    ZE_KCHECK();
    return ZE_OK;
}

extern "C" int ze_preprocess_image(ze_engine* e, const uint8_t* img, int h, int w, int64_t min_pixels,
                                   int64_t max_pixels, float* out, int64_t capacity_rows, int32_t grid_thw[3],
                                   void* stream) {
    if (!e || !img || !out || !grid_thw) return ze_fail(e, ZE_ERR_INVALID, "null argument");
    const ze_config& c = e->cfg;
    hipStream_t s = (hipStream_t)stream;
    hipSetDevice(e->device);
    const int f = c.patch_size * c.spatial_merge_size;
    int rh, rw;
    if (ze_smart_resize_impl(h, w, f, min_pixels, max_pixels, &rh, &rw) != 0)
        return ze_fail(e, ZE_ERR_INVALID, "absolute aspect ratio must be smaller than 200");
    const int64_t rows = (int64_t)(rh / c.patch_size) * (rw / c.patch_size);
    if (rows > capacity_rows) return ze_fail(e, ZE_ERR_NOMEM, "pixel_values capacity too small");
    if ((size_t)rh * rw * 3 > e->fe_img_bytes) return ze_fail(e, ZE_ERR_NOMEM, "resized image exceeds workspace");
    const int th = ze_timer_begin(e, 0, s);
    const uint8_t* cur = img;
    if (rh != h || rw != w) {
        const int32_t box[4] = {0, 0, w, h};
        ZE_TRY(crop_resize(e, img, h, w, box, e->fe_img, rh, rw, s));
        cur = e->fe_img;
    }
    ze_launch_patchify(cur, rh, rw, e->lut, out, c.patch_size, c.spatial_merge_size, c.temporal_patch_size,
                       c.in_channels, s);
    if (cur == e->fe_img) ZE_TRY(frontend_release(e, s));
    ze_timer_end(e, th, s);
    ZE_KCHECK();
    grid_thw[0] = 1;
    grid_thw[1] = rh / c.patch_size;
    grid_thw[2] = rw / c.patch_size;
    return ZE_OK;
}

// ================================================================== index helpers (C ABI wrappers)
extern "C" int ze_vision_window_index(const ze_config* cfg, const int32_t* grid_thw, int n_images,
                                      int64_t* window_index, int32_t* cu_window, int cap_cu, int* n_cu) {
    if (!cfg || !grid_thw || !window_index || !cu_window || !n_cu)
        return ze_fail(nullptr, ZE_ERR_INVALID, "null argument");
    std::vector<int64_t> wi;
    std::vector<int32_t> cu;
    ze_window_index_impl(grid_thw, n_images, cfg->spatial_merge_size, cfg->window_size, cfg->patch_size, wi, cu);
    if ((int)cu.size() > cap_cu) return ze_fail(nullptr, ZE_ERR_NOMEM, "cu_window capacity too small");
    memcpy(window_index, wi.data(), wi.size() * sizeof(int64_t));
    memcpy(cu_window, cu.data(), cu.size() * sizeof(int32_t));
    *n_cu = (int)cu.size();
    return ZE_OK;
}

extern "C" int ze_rope_index(const ze_config* cfg, const int32_t* input_ids, int len, const int32_t* grid_thw,
                             int n_images, int32_t* position_ids, int32_t* rope_delta) {
    if (!cfg || !input_ids || !position_ids || !rope_delta) return ze_fail(nullptr, ZE_ERR_INVALID, "null argument");
    const int r = ze_rope_index_impl(input_ids, len, grid_thw, n_images, cfg->image_token_id, cfg->spatial_merge_size,
                                     position_ids, rope_delta);
    if (r != 0)
        return ze_fail(nullptr, ZE_ERR_MISMATCH, "Image features and image tokens do not match (rope index)");
    return ZE_OK;
}

// ================================================================== attention tile lists
// one tile = up to 64 query rows of one segment against that segment's keys
static int build_tiles(const int32_t* cu, int n_seg, int* out /* int4 rows */, int cap_tiles, int bq = 64) {
    int n = 0;
    for (int sgi = 0; sgi < n_seg; ++sgi)
        for (int q0 = cu[sgi]; q0 < cu[sgi + 1]; q0 += bq) {
            if (n >= cap_tiles) return -1;
            out[4 * n + 0] = q0;
            out[4 * n + 1] = std::min(q0 + bq, cu[sgi + 1]);
            out[4 * n + 2] = cu[sgi];
            out[4 * n + 3] = cu[sgi + 1];
            ++n;
        }
    return n;
}

// ================================================================== ViT
extern "C" int ze_vit_forward(ze_engine* e, const float* pixel_values, const int32_t* grid_thw, int n_images,
                              void* out_embeds, void* stream) {
    if (!e || !pixel_values || !grid_thw || !out_embeds || n_images <= 0)
        return ze_fail(e, ZE_ERR_INVALID, "null argument");
    const ze_config& c = e->cfg;
    hipStream_t s = (hipStream_t)stream;
    hipSetDevice(e->device);
    const int mu = c.spatial_merge_size * c.spatial_merge_size;
    int n = 0;
    for (int i = 0; i < n_images; ++i) {
        const int t = grid_thw[3 * i], h = grid_thw[3 * i + 1], w = grid_thw[3 * i + 2];
        if (t <= 0 || h <= 0 || w <= 0 || h % c.spatial_merge_size || w % c.spatial_merge_size)
            return ze_fail(e, ZE_ERR_INVALID, "bad grid_thw");
        n += t * h * w;
    }
    if (n > c.max_patches) return ze_fail(e, ZE_ERR_NOMEM, "too many patches for max_patches");
    const int vh = c.vit_hidden, hd = e->vit_head_dim, half = hd / 2, nh = c.vit_heads;
    const int pk = c.in_channels * c.temporal_patch_size * c.patch_size * c.patch_size;

    // ---- host index prep (integer): window permutation, segment tiles, rotary tables
    std::vector<int64_t> widx;
    std::vector<int32_t> cu_win, cu_full(1, 0), hw;
    ze_window_index_impl(grid_thw, n_images, c.spatial_merge_size, c.window_size, c.patch_size, widx, cu_win);
    for (int i = 0; i < n_images; ++i)
        for (int t = 0; t < grid_thw[3 * i]; ++t)
            cu_full.push_back(cu_full.back() + grid_thw[3 * i + 1] * grid_thw[3 * i + 2]);
    ze_vision_pos_ids_impl(grid_thw, n_images, c.spatial_merge_size, hw);
    ZE_TRY(stage_acquire(e, e->v_staged));  // pinned staging reuse
    int* hi = e->v_host_ints;
    int* perm = hi;              // [n]   new row r <- old row perm[r]
    int* inv = hi + n;           // [n/mu] merged row j (window order) -> HF order row
    int* tw = hi + 2 * n;        // window tiles
    for (int j = 0; j < n / mu; ++j) {
        for (int u = 0; u < mu; ++u) perm[j * mu + u] = (int)widx[j] * mu + u;
        inv[j] = (int)widx[j];
    }
    const int ntw = build_tiles(cu_win.data(), (int)cu_win.size() - 1, tw, n);
    int* tf = tw + 4 * ntw;
    int max_win = 0;  // longest window segment (64 tokens for 112-px windows of 14-px patches)
    for (size_t i = 1; i < cu_win.size(); ++i) max_win = std::max(max_win, cu_win[i] - cu_win[i - 1]);
    // (the full-attention blocks: segments of a whole image -- 128-query tiles, two per wave; ze_tune knob 1 = 9: 64)
    const int bq_full = ze_gemv_knobs[1] == 9 ? 64 : ZE_FA_BQ_LONG;
    const int ntf = build_tiles(cu_full.data(), (int)cu_full.size() - 1, tf, n, bq_full);
    if (ntw < 0 || ntf < 0) return ze_fail(e, ZE_ERR_NOMEM, "attention tile list overflow");
    // rotary: inv_freq over dim = head_dim/2 -> half/2 frequencies per axis (HF:...:125-134, 441-446)
    const int nf = half / 2;
    std::vector<float> invf(nf);
    for (int i = 0; i < nf; ++i) invf[i] = 1.0f / powf(10000.0f, (float)(2 * i) / (float)half);
    float* hc = e->v_host_f32;
    float* hs = hc + (size_t)n * half;
    for (int r = 0; r < n; ++r) {
        const int old = perm[r];
        for (int i = 0; i < nf; ++i) {
            const float fh = (float)hw[2 * old] * invf[i], fw = (float)hw[2 * old + 1] * invf[i];
            hc[(size_t)r * half + i] = cosf(fh);
            hs[(size_t)r * half + i] = sinf(fh);
            hc[(size_t)r * half + nf + i] = cosf(fw);
            hs[(size_t)r * half + nf + i] = sinf(fw);
        }
    }
    ZE_HIP(hipMemcpyAsync(e->vperm, perm, (size_t)n * sizeof(int), hipMemcpyHostToDevice, s));
    ZE_HIP(hipMemcpyAsync(e->vinv, inv, (size_t)(n / mu) * sizeof(int), hipMemcpyHostToDevice, s));
    ZE_HIP(hipMemcpyAsync(e->vtiles_win, tw, (size_t)ntw * 16, hipMemcpyHostToDevice, s));
    ZE_HIP(hipMemcpyAsync(e->vtiles_full, tf, (size_t)ntf * 16, hipMemcpyHostToDevice, s));
    ZE_HIP(hipMemcpyAsync(e->vcos, hc, (size_t)n * half * sizeof(float), hipMemcpyHostToDevice, s));
    ZE_HIP(hipMemcpyAsync(e->vsin, hs, (size_t)n * half * sizeof(float), hipMemcpyHostToDevice, s));
    ZE_TRY(stage_release(e, e->v_staged, s));

    const int th = ze_timer_begin(e, 1, s);
    // ---- patch embed on window-ordered rows (pixel_values.type(bf16), conv3d == GEMM, no bias)
    ze_launch_gather_cast_rows(pixel_values, pk, e->vperm, e->vx, pk, n, s);
    ze_launch_gemm(ZE_EPI_NONE, e->vx, pk, e->patch_embed.w, e->patch_embed.ld, nullptr, nullptr, 0, e->vh, vh,
                   nullptr, n, vh, pk, s);
    const float scale = 1.0f / sqrtf((float)hd);
    for (int li = 0; li < c.vit_depth; ++li) {
        const ze_vit_block& b = e->vb[li];
        bool full = false;
        for (int k = 0; k < c.n_fullatt; ++k) full |= c.fullatt_block_indexes[k] == li;
        ze_launch_rmsnorm(e->vh, vh, b.norm1, e->vy, vh, n, vh, 1e-6f, s);
        ze_launch_gemm(ZE_EPI_NONE, e->vy, vh, b.qkv.w, b.qkv.ld, b.qkv.bias, nullptr, 0, e->vqkv, 3 * vh, nullptr, n,
                       3 * vh, vh, s);
        ze_launch_vision_rope(e->vqkv, e->vcos, e->vsin, n, nh, hd, s);
        ze_launch_flash_attn(hd, 0, e->vqkv, 3 * vh, hd, e->vqkv + vh, 3 * vh, hd, e->vqkv + 2 * vh, 3 * vh, hd, e->vo,
                             vh, hd, full ? e->vtiles_full : e->vtiles_win, full ? ntf : ntw, nh, 1, scale, 0, s, nullptr, 0,
                             full ? bq_full : 64, full ? 0 : max_win);
        ze_launch_gemm(ZE_EPI_RESIDUAL, e->vo, vh, b.proj.w, b.proj.ld, b.proj.bias, e->vh, vh, e->vh, vh, nullptr, n,
                       vh, vh, s);
        ze_launch_rmsnorm(e->vh, vh, b.norm2, e->vy, vh, n, vh, 1e-6f, s);
        ze_launch_gemm(ZE_EPI_SWIGLU, e->vy, vh, b.gate_up.w, b.gate_up.ld, b.gate_up.bias, nullptr, 0, e->va,
                       e->vit_ipad, nullptr, n, 2 * e->vit_ipad, vh, s);
        ze_launch_gemm(ZE_EPI_RESIDUAL, e->va, e->vit_ipad, b.down.w, b.down.ld, b.down.bias, e->vh, vh, e->vh, vh,
                       nullptr, n, vh, e->vit_ipad, s);
    }
    // ---- merger: RMSNorm -> [n/mu, vh*mu] -> Linear+GELU -> Linear, rows scattered back to HF order
    ze_launch_rmsnorm(e->vh, vh, e->ln_q, e->vy, vh, n, vh, 1e-6f, s);
    const int mh = vh * mu;
    ze_launch_gemm(ZE_EPI_GELU, e->vy, mh, e->merger0.w, e->merger0.ld, e->merger0.bias, nullptr, 0, e->vz, mh, nullptr,
                   n / mu, mh, mh, s);
    ze_launch_gemm(ZE_EPI_NONE, e->vz, mh, e->merger2.w, e->merger2.ld, e->merger2.bias, nullptr, 0,
                   (bf16_t*)out_embeds, c.vit_out_hidden, e->vinv, n / mu, c.vit_out_hidden, mh, s);
    ze_timer_end(e, th, s);
    ZE_KCHECK();
    return ZE_OK;
}

// ================================================================== chains
static int check_seq(ze_engine* e, int seq) {
    if (!e) return ze_fail(e, ZE_ERR_INVALID, "null engine");
    if (seq < 0 || seq >= e->cfg.max_seqs) return ze_fail(e, ZE_ERR_NOTFOUND, "sequence id out of range");
    return ZE_OK;
}

// ---- shared-prefix hints (ze_engine::pfx_host / pfx_dev; the ownership rule is stated in ze_engine.h)
static void set_prefix_hint(ze_engine* e, int seq, int value) { e->pfx_host[seq] = value; }
// the K/V rows chain `d` copied (ze_seq_copy_prefix) have landed: nothing recorded, or the event behind the copy is over
static bool prefix_copy_done(ze_engine* e, int d) {
    hipEvent_t ev = e->pfx_copy_ev[d];
    return !ev || hipEventQuery(ev) == hipSuccess;
}
// rows below `keep` of chain `seq` are about to change (or the chain is over): chains that read their prefix from it move to
// the one among them with the longest prefix WHOSE COPY HAS LANDED (its own rows hold the same bits), which goes back to reading
// its own rows; readers it does not cover -- and all of them when no copy has landed yet -- go back to their own rows as well
// (a chain joins a decode step only behind its own prefill pass, so its own rows are always good by then)
static void prefix_source_gone(ze_engine* e, int seq, int keep) {
    int leader = -1, lead_p = 0;
    const int n = (int)e->pfx_host.size();
    for (int d = 0; d < n; ++d) {
        const int h = e->pfx_host[d];
        if (h != 0 && (h >> 16) == seq && (h & 0xffff) > keep && (h & 0xffff) > lead_p && prefix_copy_done(e, d))
            leader = d, lead_p = h & 0xffff;
    }
    for (int d = 0; d < n; ++d) {
        const int h = e->pfx_host[d];
        if (h == 0 || (h >> 16) != seq || (h & 0xffff) <= keep) continue;
        const int p = h & 0xffff;
        set_prefix_hint(e, d, (leader < 0 || d == leader || p > lead_p) ? 0 : ((leader << 16) | p));
    }
}
// Before a batched decode step is enqueued on `s`: the device words of ITS chains follow the host's (the one place pfx_dev is
// written, on the one stream that reads it)
static void sync_prefix(ze_engine* e, const int32_t* seqs, int n, hipStream_t s) {
    int idx[64], val[64], m = 0;
    for (int i = 0; i < n; ++i) {
        const int q = seqs[i];
        const int want = (e->prefix_hints && ze_gemv_knobs[17] != 1) ? e->pfx_host[q] : 0;
        if (e->pfx_pushed[q] == want) continue;
        e->pfx_pushed[q] = want;
        idx[m] = q, val[m] = want;
        if (++m == 64) ze_launch_scatter_ints(e->pfx_dev, idx, val, m, s), m = 0;
    }
    if (m) ze_launch_scatter_ints(e->pfx_dev, idx, val, m, s);
}

static int push_state(ze_engine* e, int seq, hipStream_t s, int token, int n_gen, int finished) {
    ze_seq_dev st;
    memset(&st, 0, sizeof(st));
    st.ctx = e->ctx_host[seq];
    st.rope_delta = e->delta_host[seq];
    st.token = token;
    st.finished = finished;
    st.n_gen = n_gen;
    st.max_gen = e->cfg.max_ctx;
    st.split = e->split_host[seq];
    static_assert(sizeof(ze_seq_dev) == 8 * sizeof(int), "chain state is eight ints");
    ze_launch_set_ints(reinterpret_cast<int*>(e->st_dev + seq), reinterpret_cast<const int*>(&st), 8, s);
    return ZE_OK;
}

extern "C" int ze_seq_reset(ze_engine* e, int seq, void* stream) {
    ZE_TRY(check_seq(e, seq));
    hipSetDevice(e->device);
    hipStream_t s = (hipStream_t)stream;
    e->ctx_host[seq] = 0;
    e->delta_host[seq] = 0;
    e->split_host[seq] = 0;
    prefix_source_gone(e, seq, 0);
    e->pfx_host[seq] = 0;
    ZE_HIP(hipMemsetAsync(e->seen + (size_t)seq * e->cfg.vocab, 0, e->cfg.vocab, s));
    return push_state(e, seq, s, 0, 0, 0);
}

// The chain in `seq` is over and its slot may be given to another chain: no decode step ENQUEUED after this call reads a prefix
// from it (the step pushes the changed hints on its own stream first: sync_prefix).  The caller orders the call behind the
// steps already in flight that may still read the slot's rows -- a scheduler calls it between bursts -- before it lets anything
// overwrite them.  `stream` is unused since round 4 (kept for the ABI): hints are host state until a decode step picks them up.
extern "C" int ze_seq_retire(ze_engine* e, int seq, void* stream) {
    (void)stream;
    ZE_TRY(check_seq(e, seq));
    prefix_source_gone(e, seq, 0);
    set_prefix_hint(e, seq, 0);
    return ZE_OK;
}

// Declares that the first `rows` cached tokens of `seq` hold the same bits as the source chain's (the caller's knowledge: a
// prefix it wrote to both; ze_seq_copy_prefix records this by itself).  rows = 0 clears.  Also the measurement hook of
// bench.py / tools/pmc_kernel.py: the decode attention timed with the sharing the question stream has.
extern "C" int ze_seq_set_prefix_hint(ze_engine* e, int seq, int src_seq, int rows, void* stream) {
    (void)stream;
    ZE_TRY(check_seq(e, seq));
    if (rows <= 0 || src_seq == seq) {
        set_prefix_hint(e, seq, 0);
        return ZE_OK;
    }
    ZE_TRY(check_seq(e, src_seq));
    if (rows > e->ctx_host[seq] || rows > e->ctx_host[src_seq] || rows >= 65536 || !e->prefix_hints)
        return ze_fail(e, ZE_ERR_INVALID, "prefix hint: rows exceed a chain's context (or hints are switched off)");
    set_prefix_hint(e, seq, (src_seq << 16) | rows);
    return ZE_OK;
}

// (source chain << 16) | rows: where the decode attention reads the first rows of `seq` from; 0 = its own cache
// Measurement / test entry: declares row `rows` the chain's split row (what prefilling tokens with an image block does by itself,
// note_split) -- for chains built from text ids that stand in for image prompts (tools/pmc_kernel.py, bench.py's annex, tests).
extern "C" int ze_seq_set_split(ze_engine* e, int seq, int rows, void* stream) {
    ZE_TRY(check_seq(e, seq));
    if (rows < 0 || rows > e->ctx_host[seq] || rows >= 65536) return ze_fail(e, ZE_ERR_INVALID, "split row out of range");
    hipSetDevice(e->device);
    e->split_host[seq] = rows;
    ze_launch_set_ints(&(e->st_dev + seq)->split, &rows, 1, (hipStream_t)stream);
    return ZE_OK;
}

extern "C" int ze_seq_prefix_hint(ze_engine* e, int seq) {
    if (check_seq(e, seq) != 0) return ZE_ERR_NOTFOUND;
    return e->pfx_host[seq];
}

extern "C" int ze_seq_truncate(ze_engine* e, int seq, int keep_len, void* stream) {
    ZE_TRY(check_seq(e, seq));
    if (keep_len < 0 || keep_len > e->ctx_host[seq]) return ze_fail(e, ZE_ERR_INVALID, "keep_len out of range");
    hipSetDevice(e->device);
    e->ctx_host[seq] = keep_len;
    if (keep_len < e->split_host[seq]) e->split_host[seq] = 0;   // (the image block itself is cut: the rest is a chain without one)
    prefix_source_gone(e, seq, keep_len);   // rows from keep_len on will be rewritten
    if ((e->pfx_host[seq] & 0xffff) > keep_len) e->pfx_host[seq] = 0;
    // the seen-set belongs to the dropped continuation: the caller re-marks the (new) prompt
    ZE_HIP(hipMemsetAsync(e->seen + (size_t)seq * e->cfg.vocab, 0, e->cfg.vocab, (hipStream_t)stream));
    return push_state(e, seq, (hipStream_t)stream, 0, 0, 0);
}

extern "C" int ze_seq_copy_prefix(ze_engine* e, int dst_seq, int src_seq, int n_tokens, void* stream) {
    ZE_TRY(check_seq(e, dst_seq));
    ZE_TRY(check_seq(e, src_seq));
    if (dst_seq == src_seq) return ze_fail(e, ZE_ERR_INVALID, "source and destination chain are the same");
    if (n_tokens <= 0 || n_tokens > e->ctx_host[src_seq]) return ze_fail(e, ZE_ERR_INVALID, "n_tokens exceeds the source chain's context");
    const ze_config& c = e->cfg;
    hipSetDevice(e->device);
    hipStream_t s = (hipStream_t)stream;
    const size_t head_stride = (size_t)c.max_ctx * e->head_dim, seq_stride = (size_t)c.kv_heads * head_stride;
    ze_launch_kv_copy_prefix(e->kcache, e->vcache, (size_t)c.max_seqs * seq_stride, seq_stride, head_stride, c.layers, c.kv_heads,
                             e->head_dim, src_seq, dst_seq, n_tokens, s);
    ZE_KCHECK();
    e->ctx_host[dst_seq] = n_tokens;
    e->delta_host[dst_seq] = 0;
    // the split row travels with the rows: the copy holds the source's first image block iff it reaches past its end (the same
    // tokens, the same split -- however a chain came by its rows)
    e->split_host[dst_seq] = (e->split_host[src_seq] > 0 && e->split_host[src_seq] <= n_tokens) ? e->split_host[src_seq] : 0;
    prefix_source_gone(e, dst_seq, 0);
    if (!e->pfx_copy_ev[dst_seq]) ZE_HIP(hipEventCreateWithFlags(&e->pfx_copy_ev[dst_seq], hipEventDisableTiming));
    ZE_HIP(hipEventRecord(e->pfx_copy_ev[dst_seq], s));  // (a holder other chains may be pointed at only once this is over)
    // the copy stays (the prefill attention of a later pass reads the chain's own rows); the decode attention reads the
    // source's -- or the source's own source, when that one covers these rows
    {
        int src = src_seq;
        const int hs = e->pfx_host[src_seq];
        if (hs != 0 && (hs & 0xffff) >= n_tokens) src = hs >> 16;
        e->pfx_host[dst_seq] = (e->prefix_hints && n_tokens < 65536 && src != dst_seq) ? ((src << 16) | n_tokens) : 0;
    }
    ZE_HIP(hipMemsetAsync(e->seen + (size_t)dst_seq * c.vocab, 0, c.vocab, s));
    return push_state(e, dst_seq, s, 0, 0, 0);
}

extern "C" int ze_seq_len(ze_engine* e, int seq) {
    if (check_seq(e, seq) != 0) return ZE_ERR_NOTFOUND;
    return e->ctx_host[seq];
}

extern "C" int ze_seq_mark_seen(ze_engine* e, int seq, const int32_t* ids, int n, void* stream) {
    ZE_TRY(check_seq(e, seq));
    if (n < 0 || (n > 0 && !ids)) return ze_fail(e, ZE_ERR_INVALID, "bad ids");
    if ((size_t)n > e->t_host_ints_cap) return ze_fail(e, ZE_ERR_NOMEM, "too many ids");
    for (int i = 0; i < n; ++i)
        if (ids[i] < 0 || ids[i] >= e->cfg.vocab) return ze_fail(e, ZE_ERR_INVALID, "token id out of range");
    hipSetDevice(e->device);
    hipStream_t s = (hipStream_t)stream;
    // the ids travel as kernel arguments (128 per launch) into the prefill's id buffer, in stream order behind whatever still
    // reads it: no pinned staging buffer to wait for -- the stream synchronisation this call used to begin with made the
    // scheduler wait for its whole prefill pass once per request (2.1 ms per call on the real entry point)
    ze_launch_set_ints(e->tsrc, ids, n, s);
    ze_launch_mark_seen(e->seen + (size_t)seq * e->cfg.vocab, e->tsrc, n, s);
    ZE_KCHECK();
    return ZE_OK;
}

// pinned + device scratch of the batched id transfers, sized by the REQUEST and grown on demand (max_seqs x max_ctx ints per
// direction were ~100 MB four times over at 768 slots x 32 K context, for calls that move a few hundred KB): `need` ints, rounded up
// to the next power of two from 64 Ki ints on.  Growing waits for the work that may still read the old block (the caller's stream and
// the block's staging event), frees it, and allocates anew; a failed device allocation frees the pinned half it had just obtained.
static int xfer_reserve(ze_engine* e, int*& host, int*& dev, size_t& cap, size_t need, hipEvent_t staged, hipStream_t s) {
    if (need <= cap) return ZE_OK;
    size_t want = (size_t)64 << 10;
    while (want < need) want <<= 1;
    if (cap) {
        if (staged) ZE_HIP(hipEventSynchronize(staged));
        ZE_HIP(hipStreamSynchronize(s));
        hipHostFree(host);
        hipFree(dev);
        host = dev = nullptr;
        cap = 0;
    }
    int *h = nullptr, *d = nullptr;
    ZE_HIP(hipHostMalloc((void**)&h, want * sizeof(int)));
    if (hipMalloc((void**)&d, want * sizeof(int)) != hipSuccess) {
        hipHostFree(h);
        return ze_fail(e, ZE_ERR_NOMEM, "hipMalloc of the id transfer scratch failed");
    }
    host = h;
    dev = d;
    cap = want;
    return ZE_OK;
}

// ze_seq_mark_seen for the n chains of a prefill pass at once: ONE host -> device copy and ONE launch instead of ~10 launches per
// chain (the scheduler's thread spent 1.4 ms per chain in the single-chain call on a busy GPU: 1.8 s per lane of the 15-s stream)
extern "C" int ze_seq_mark_seen_batch(ze_engine* e, const int32_t* seqs, const int32_t* counts, int n, const int32_t* ids, void* stream) {
    if (!e || n < 0 || (n > 0 && (!seqs || !counts))) return ze_fail(e, ZE_ERR_INVALID, "bad arguments");
    if (n == 0) return ZE_OK;
    if (n > e->cfg.max_seqs) return ze_fail(e, ZE_ERR_INVALID, "more chains than slots");
    size_t total = 0;
    int max_count = 0;
    for (int i = 0; i < n; ++i) {
        ZE_TRY(check_seq(e, seqs[i]));
        if (counts[i] < 0 || counts[i] > e->cfg.max_ctx) return ze_fail(e, ZE_ERR_INVALID, "bad id count");
        total += (size_t)counts[i];
        max_count = std::max(max_count, counts[i]);
    }
    if (total > 0 && !ids) return ze_fail(e, ZE_ERR_INVALID, "bad ids");
    for (size_t i = 0; i < total; ++i)
        if (ids[i] < 0 || ids[i] >= e->cfg.vocab) return ze_fail(e, ZE_ERR_INVALID, "token id out of range");
    if (total == 0) return ZE_OK;
    hipSetDevice(e->device);
    hipStream_t s = (hipStream_t)stream;
    // (the event is recorded BEHIND the kernel that reads the device block -- ADVICE r5: behind the copy alone, a second call on
    //  another stream could have overwritten xs_dev under the first call's kernel; acquire waits for that kernel now)
    ZE_TRY(stage_acquire(e, e->xs_staged));
    ZE_TRY(xfer_reserve(e, e->xs_host, e->xs_dev, e->xs_cap, (size_t)(2 * n + 1) + total, e->xs_staged, s));
    int* h = e->xs_host;
    h[0] = 0;
    for (int i = 0; i < n; ++i) {
        h[i + 1] = h[i] + counts[i];
        h[n + 1 + i] = seqs[i];
    }
    memcpy(h + 2 * n + 1, ids, total * sizeof(int));
    ZE_HIP(hipMemcpyAsync(e->xs_dev, h, ((size_t)(2 * n + 1) + total) * sizeof(int), hipMemcpyHostToDevice, s));
    ze_launch_mark_seen_batch(e->seen, e->cfg.vocab, e->xs_dev, e->xs_dev + 2 * n + 1, n, max_count, s);
    ZE_KCHECK();
    ZE_TRY(stage_release(e, e->xs_staged, s));
    return ZE_OK;
}

// ================================================================== prefill
// The chain's SPLIT ROW (round 6; ze_seq_dev::split, ze_attn_batch.hip): the row behind its FIRST image block's <|vision_end|> --
// what the questions about one tile share (system turn + the view's image tokens) ends there.  Found where the tokens are
// prefilled, whatever pass brings them; a chain that copies the block (ze_seq_copy_prefix) inherits it.  A property of the
// chain's tokens alone: the decode attention may cut its parts there without a chain's sums depending on the batch.
static void note_split(ze_engine* e, int seq, const int32_t* ids, int len, int past) {
    if (e->split_host[seq] != 0 || e->cfg.vision_end_token_id < 0) return;
    for (int t = 0; t < len; ++t)
        if (ids[t] == e->cfg.vision_end_token_id) {
            if (past + t + 1 < 65536) e->split_host[seq] = past + t + 1;
            return;
        }
}

// RMSNorm + the projection behind it, on `rows` prefill rows of e->th.  With FP8 activations on a quantised engine the row
// goes out as E4M3 bytes + one scale and the product runs on the block-scaled FP8 MFMA against the FP8 weight rows
// (k_gemm_ring_mx: the values of the fake-quantised bf16 path, another summation order; ze_tune knob 12 = 1 keeps that path).
static void prefill_norm_gemm(ze_engine* e, const bf16_t* norm_w, const ze_linear& lin, const bf16_t* bias, int epi, bf16_t* out,
                              int ldo, int rows, int N, hipStream_t s) {
    const ze_config& c = e->cfg;
    const int H = c.hidden;
    if (e->fp8_act && lin.w8 && ze_gemv_knobs[12] != 1 && H % 128 == 0 && H >= 256 && lin.ld8 % 16 == 0) {
        ze_launch_rmsnorm(e->th, H, norm_w, e->ty, H, rows, H, c.rms_eps, s, 0, 3, e->ty8p, e->ty8p_scale);
        if (ze_launch_gemm_mx(epi, e->ty8p, H, e->ty8p_scale, lin.w8, lin.ld8, lin.scale8, bias, out, ldo, rows, N, H, s)) return;
    }
    ze_launch_rmsnorm(e->th, H, norm_w, e->ty, H, rows, H, c.rms_eps, s, 0, e->fp8_act ? 1 : 0);
    ze_launch_gemm(epi, e->ty, H, lin.w, lin.ld, bias, nullptr, 0, out, ldo, nullptr, rows, N, H, s, e->prefill_ws());
}

static int prefill_impl(ze_engine* e, int seq, const int32_t* input_ids, int len, const void* image_embeds,
                        int n_image_rows, const int32_t* position_ids, int rope_delta, float* out_logits,
                        float* out_logps, void* stream) {
    ZE_TRY(check_seq(e, seq));
    if (!input_ids || !position_ids || len <= 0) return ze_fail(e, ZE_ERR_INVALID, "bad prefill arguments");
    const ze_config& c = e->cfg;
    const int past = e->ctx_host[seq];
    if (past + len > c.max_ctx) return ze_fail(e, ZE_ERR_NOMEM, "sequence exceeds max_ctx");
    hipStream_t s = (hipStream_t)stream;
    hipSetDevice(e->device);
    const int H = c.hidden, hd = e->head_dim, nq = c.heads * hd, nkv = c.kv_heads * hd, nqkv = nq + 2 * nkv;

    ZE_TRY(stage_acquire(e, e->t_staged));  // pinned staging reuse
    int* src = e->t_host_ints;
    int* pos = src + len;
    int* tiles = pos + 3 * len;
    int img = 0;
    for (int t = 0; t < len; ++t) {
        const int id = input_ids[t];
        if (id < 0 || id >= c.vocab) return ze_fail(e, ZE_ERR_INVALID, "token id out of range");
        src[t] = (id == c.image_token_id) ? -1 - img++ : id;
        for (int a = 0; a < 3; ++a) {
            const int p = position_ids[a * len + t];
            if (p < 0 || p >= e->max_pos) return ze_fail(e, ZE_ERR_INVALID, "position id out of range");
            pos[a * len + t] = p;
        }
    }
    if (img != n_image_rows || (img > 0 && !image_embeds))
        return ze_fail(e, ZE_ERR_MISMATCH, "Image features and image tokens do not match, tokens: " +
                                               std::to_string(img) + ", features: " + std::to_string(n_image_rows));
    note_split(e, seq, input_ids, len, past);
    int nt = 0;
    // query rows per attention tile: 128 -- two query tiles per wave, half the LDS fragment reads per MFMA -- on the LDS-DMA
    // staging form of the kernel (no staging registers: 248 VGPRs, two workgroups per CU; with register staging the same tile
    // needs 280 and lost: 120.5 against 116.9 ms per pass pair).  16-chain pass pair: register-staged 64-row tiles 107.0 ms,
    // DMA 64-row 105.3, DMA 128-row 104.3.  ze_tune knob 1 = 9: 64-row tiles, 7: the register-staged form.  Same bits either way.
    const int bq = (ze_gemv_knobs[1] == 9 || ze_gemv_knobs[1] == 7) ? 64 : ZE_FA_BQ_LONG;
    for (int q0 = 0; q0 < len; q0 += bq, ++nt) {
        tiles[4 * nt + 0] = q0;
        tiles[4 * nt + 1] = std::min(q0 + bq, len);
        tiles[4 * nt + 2] = 0;
        tiles[4 * nt + 3] = past + len;
    }
    ZE_HIP(hipMemcpyAsync(e->tsrc, src, (size_t)len * sizeof(int), hipMemcpyHostToDevice, s));
    ZE_HIP(hipMemcpyAsync(e->tpos, pos, (size_t)3 * len * sizeof(int), hipMemcpyHostToDevice, s));
    ZE_HIP(hipMemcpyAsync(e->ttiles, tiles, (size_t)nt * 16, hipMemcpyHostToDevice, s));
    ZE_TRY(stage_release(e, e->t_staged, s));

    const int th = ze_timer_begin(e, 2, s);
    ze_launch_embed_rows(e->tsrc, e->embed, (const bf16_t*)image_embeds, e->th, len, H, s);
    const float scale = 1.0f / sqrtf((float)hd);
    // the queries' M-RoPE inside the flash kernel (k_mrope_kv_vec then moves K and V only: a fifth of its rows); ze_tune knob 22 = 1:
    // the two-launch form of rounds 1-5 (the same bits: tests/test_gpu_model.py)
    const bool q_in_flash = hd == 128 && ze_mrope_vec_ok != 0 && ze_gemv_knobs[22] != 1;
    for (int li = 0; li < c.layers; ++li) {
        const ze_text_layer& L = e->tl[li];
        prefill_norm_gemm(e, L.in_norm, L.qkv, L.qkv.bias, ZE_EPI_NONE, e->tqkv, nqkv, len, nqkv, s);
        ze_launch_mrope_kv(e->tqkv, len, c.heads, c.kv_heads, hd, e->cosT, e->sinT, e->tpos, e->axis_of,
                           e->kc(li, seq), e->vc(li, seq), c.max_ctx, past, nullptr, 0, s, q_in_flash ? 1 : 0);
        ze_launch_flash_attn(hd, 1, e->tqkv, nqkv, hd, e->kc(li, seq), hd, c.max_ctx * hd, e->vc(li, seq), hd,
                             c.max_ctx * hd, e->to, nq, hd, e->ttiles, nt, c.heads, c.heads / c.kv_heads, scale, past,
                             s, nullptr, 0, bq, 0, q_in_flash ? ze_fa_rope{e->cosT, e->sinT, e->tpos, c.mrope_section[0], c.mrope_section[0] + c.mrope_section[1], len} : ze_fa_rope{nullptr, nullptr, nullptr, 0, 0, 0});
        ze_launch_gemm(ZE_EPI_RESIDUAL, e->to, nq, L.o.w, L.o.ld, nullptr, e->th, H, e->th, H, nullptr, len, H, nq, s, e->prefill_ws());
        prefill_norm_gemm(e, L.post_norm, L.gate_up, nullptr, ZE_EPI_SWIGLU, e->ta, e->text_ipad, len, 2 * e->text_ipad, s);
        ze_launch_gemm(ZE_EPI_RESIDUAL, e->ta, e->text_ipad, L.down.w, L.down.ld, nullptr, e->th, H, e->th, H, nullptr,
                       len, H, e->text_ipad, s, e->prefill_ws());
    }
    // last position: final norm fused into the lm_head stream (logits_to_keep = 1, HF:...:1386-1387) -- the kernel of the
    // batched pass with one chain, so the first token does not depend on how the chain was prefilled
    {
        const bf16_t* xr = e->th + (size_t)(len - 1) * H;
        float* lo = e->dlogits + (size_t)seq * c.vocab;
        if (!ze_launch_logits_rows(e->lm_head, H, c.vocab, H, e->final_norm, c.rms_eps, &xr, &lo, 1, s)) {
            ze_gemv_args a;
            memset(&a, 0, sizeof(a));
            a.W = e->lm_head;
            a.ldw = H;
            a.N = c.vocab;
            a.K = H;
            a.x = xr;
            a.norm_w = e->final_norm;
            a.eps = c.rms_eps;
            a.out_f32 = lo;
            a.D = hd;
            ze_launch_gemv(ZE_GV_LOGITS, a, s);
        }
    }
    if (out_logps && len > 1) {
        // every position: final norm, lm_head GEMM in row chunks (bf16 logits as HF's lm_head gives them), then the
        // log-softmax pick of the next id.  The logits live in the (now free) MLP activation workspace.
        const int ldl = (c.vocab + 7) & ~7;
        int chunk = (int)std::min<size_t>((size_t)e->prefill_rows * e->text_ipad / ldl, (size_t)len - 1);
        if (chunk >= 128) chunk &= ~127;
        if (chunk < 1) return ze_fail(e, ZE_ERR_NOMEM, "prefill workspace too small for one row of logits");
        ZE_HIP(hipMemcpyAsync(e->trow_aux, input_ids + 1, (size_t)(len - 1) * sizeof(int), hipMemcpyHostToDevice, s));
        ze_launch_rmsnorm(e->th, H, e->final_norm, e->ty, H, len - 1, H, c.rms_eps, s);
        for (int r0 = 0; r0 < len - 1; r0 += chunk) {
            const int m = std::min(chunk, len - 1 - r0);
            ze_launch_gemm(ZE_EPI_NONE, e->ty + (size_t)r0 * H, H, e->lm_head, H, nullptr, nullptr, 0, e->ta, ldl,
                           nullptr, m, c.vocab, H, s);
            ze_launch_token_logprob(e->ta, ldl, c.vocab, e->trow_aux + r0, out_logps + r0, m, s);
        }
    }
    ze_timer_end(e, th, s);
    ZE_KCHECK();
    if (out_logits)
        ZE_HIP(hipMemcpyAsync(out_logits, e->dlogits + (size_t)seq * c.vocab, (size_t)c.vocab * sizeof(float),
                              hipMemcpyDeviceToDevice, s));
    e->ctx_host[seq] = past + len;
    e->delta_host[seq] = rope_delta;
    return push_state(e, seq, s, input_ids[len - 1], 0, 0);
}

extern "C" int ze_prefill(ze_engine* e, int seq, const int32_t* input_ids, int len, const void* image_embeds,
                          int n_image_rows, const int32_t* position_ids, int rope_delta, float* out_logits,
                          void* stream) {
    return prefill_impl(e, seq, input_ids, len, image_embeds, n_image_rows, position_ids, rope_delta, out_logits,
                        nullptr, stream);
}

extern "C" int ze_score(ze_engine* e, int seq, const int32_t* input_ids, int len, const void* image_embeds,
                        int n_image_rows, const int32_t* position_ids, int rope_delta, float* out_logps,
                        void* stream) {
    if (!out_logps) return ze_fail(e, ZE_ERR_INVALID, "ze_score needs an output buffer");
    return prefill_impl(e, seq, input_ids, len, image_embeds, n_image_rows, position_ids, rope_delta, nullptr,
                        out_logps, stream);
}

// Prefill of n chains in one pass: all rows share every GEMM, attention and the KV append go per chain.
extern "C" int ze_prefill_batch(ze_engine* e, const int32_t* seqs, int n, const int32_t* lens, const int32_t* input_ids,
                                const void* image_embeds, const int32_t* n_image_rows, const int32_t* position_ids,
                                const int32_t* rope_deltas, void* stream) {
    if (!e || !seqs || !lens || !input_ids || !position_ids || !rope_deltas || n <= 0)
        return ze_fail(e, ZE_ERR_INVALID, "bad prefill arguments");
    const ze_config& c = e->cfg;
    hipStream_t s = (hipStream_t)stream;
    hipSetDevice(e->device);
    const int H = c.hidden, hd = e->head_dim, nq = c.heads * hd, nkv = c.kv_heads * hd, nqkv = nq + 2 * nkv;
    int total = 0;
    for (int i = 0; i < n; ++i) {
        ZE_TRY(check_seq(e, seqs[i]));
        for (int k2 = 0; k2 < i; ++k2)
            if (seqs[k2] == seqs[i]) return ze_fail(e, ZE_ERR_INVALID, "a chain appears twice in the batch");
        if (lens[i] <= 0) return ze_fail(e, ZE_ERR_INVALID, "bad prefill arguments");
        if (e->ctx_host[seqs[i]] + lens[i] > c.max_ctx) return ze_fail(e, ZE_ERR_NOMEM, "sequence exceeds max_ctx");
        total += lens[i];
    }
    if (total > e->prefill_rows) return ze_fail(e, ZE_ERR_NOMEM, "batched prefill exceeds max_prefill_rows");

    ZE_TRY(stage_acquire(e, e->t_staged));  // pinned staging reuse
    int* src = e->t_host_ints;
    int* pos = src + total;
    int* row_aux = pos + 3 * total;
    int* tiles = row_aux + 2 * total;
    int img = 0, nt = 0, row0 = 0, img_expected = 0;
    // query rows per attention tile: 128 -- two query tiles per wave, half the LDS fragment reads per MFMA -- on the LDS-DMA
    // staging form of the kernel (no staging registers: 248 VGPRs, two workgroups per CU; with register staging the same tile
    // needs 280 and lost: 120.5 against 116.9 ms per pass pair).  16-chain pass pair: register-staged 64-row tiles 107.0 ms,
    // DMA 64-row 105.3, DMA 128-row 104.3.  ze_tune knob 1 = 9: 64-row tiles, 7: the register-staged form.  Same bits either way.
    const int bq = (ze_gemv_knobs[1] == 9 || ze_gemv_knobs[1] == 7) ? 64 : ZE_FA_BQ_LONG;
    std::vector<int> tile_aux;
    for (int i = 0; i < n; ++i) {
        const int seq = seqs[i], len = lens[i], past = e->ctx_host[seq];
        int img_chain = 0;
        for (int t = 0; t < len; ++t) {
            const int id = input_ids[row0 + t];
            if (id < 0 || id >= c.vocab) return ze_fail(e, ZE_ERR_INVALID, "token id out of range");
            if (id == c.image_token_id) {
                src[row0 + t] = -1 - img++;
                ++img_chain;
            } else {
                src[row0 + t] = id;
            }
            for (int a = 0; a < 3; ++a) {
                const int p = position_ids[(size_t)a * total + row0 + t];
                if (p < 0 || p >= e->max_pos) return ze_fail(e, ZE_ERR_INVALID, "position id out of range");
                pos[(size_t)a * total + row0 + t] = p;
            }
            row_aux[2 * (row0 + t)] = seq;
            row_aux[2 * (row0 + t) + 1] = past + t;
        }
        note_split(e, seq, input_ids + row0, len, past);
        const int want = n_image_rows ? n_image_rows[i] : 0;
        if (img_chain != want || (img_chain > 0 && !image_embeds))
            return ze_fail(e, ZE_ERR_MISMATCH, "Image features and image tokens do not match, tokens: " +
                                                   std::to_string(img_chain) + ", features: " + std::to_string(want));
        img_expected += want;
        for (int q0 = 0; q0 < len; q0 += bq, ++nt) {
            tiles[4 * nt + 0] = row0 + q0;
            tiles[4 * nt + 1] = row0 + std::min(q0 + bq, len);
            tiles[4 * nt + 2] = 0;
            tiles[4 * nt + 3] = past + len;
            tile_aux.push_back(seq);
            tile_aux.push_back(past - row0);  // key index visible to row r: <= r + (past - row0)
        }
        row0 += len;
    }
    (void)img_expected;
    int* taux = tiles + 4 * nt;
    memcpy(taux, tile_aux.data(), tile_aux.size() * sizeof(int));
    ZE_HIP(hipMemcpyAsync(e->tsrc, src, (size_t)total * sizeof(int), hipMemcpyHostToDevice, s));
    ZE_HIP(hipMemcpyAsync(e->tpos, pos, (size_t)3 * total * sizeof(int), hipMemcpyHostToDevice, s));
    ZE_HIP(hipMemcpyAsync(e->trow_aux, row_aux, (size_t)2 * total * sizeof(int), hipMemcpyHostToDevice, s));
    ZE_HIP(hipMemcpyAsync(e->ttiles, tiles, (size_t)nt * 16, hipMemcpyHostToDevice, s));
    ZE_HIP(hipMemcpyAsync(e->ttile_aux, taux, (size_t)nt * 2 * sizeof(int), hipMemcpyHostToDevice, s));
    ZE_TRY(stage_release(e, e->t_staged, s));

    const int th = ze_timer_begin(e, 2, s);
    ze_launch_embed_rows(e->tsrc, e->embed, (const bf16_t*)image_embeds, e->th, total, H, s);
    const float scale = 1.0f / sqrtf((float)hd);
    const size_t seq_stride = (size_t)c.kv_heads * c.max_ctx * hd;
    // the queries' M-RoPE inside the flash kernel (k_mrope_kv_vec then moves K and V only: a fifth of its rows); ze_tune knob 22 = 1:
    // the two-launch form of rounds 1-5 (the same bits: tests/test_gpu_model.py)
    const bool q_in_flash = hd == 128 && ze_mrope_vec_ok != 0 && ze_gemv_knobs[22] != 1;
    for (int li = 0; li < c.layers; ++li) {
        const ze_text_layer& L = e->tl[li];
        prefill_norm_gemm(e, L.in_norm, L.qkv, L.qkv.bias, ZE_EPI_NONE, e->tqkv, nqkv, total, nqkv, s);
        ze_launch_mrope_kv(e->tqkv, total, c.heads, c.kv_heads, hd, e->cosT, e->sinT, e->tpos, e->axis_of, e->kc(li, 0),
                           e->vc(li, 0), c.max_ctx, 0, e->trow_aux, seq_stride, s, q_in_flash ? 1 : 0);
        ze_launch_flash_attn(hd, 1, e->tqkv, nqkv, hd, e->kc(li, 0), hd, c.max_ctx * hd, e->vc(li, 0), hd,
                             c.max_ctx * hd, e->to, nq, hd, e->ttiles, nt, c.heads, c.heads / c.kv_heads, scale, 0, s,
                             e->ttile_aux, seq_stride, bq, 0,
                             q_in_flash ? ze_fa_rope{e->cosT, e->sinT, e->tpos, c.mrope_section[0], c.mrope_section[0] + c.mrope_section[1], total} : ze_fa_rope{nullptr, nullptr, nullptr, 0, 0, 0});
        ze_launch_gemm(ZE_EPI_RESIDUAL, e->to, nq, L.o.w, L.o.ld, nullptr, e->th, H, e->th, H, nullptr, total, H, nq, s, e->prefill_ws());
        prefill_norm_gemm(e, L.post_norm, L.gate_up, nullptr, ZE_EPI_SWIGLU, e->ta, e->text_ipad, total, 2 * e->text_ipad, s);
        ze_launch_gemm(ZE_EPI_RESIDUAL, e->ta, e->text_ipad, L.down.w, L.down.ld, nullptr, e->th, H, e->th, H, nullptr,
                       total, H, e->text_ipad, s, e->prefill_ws());
    }
    // last position of every chain: final norm + lm_head, the weight matrix streamed once per EIGHT chains (ze_gemv_logits.hip;
    // per chain the arithmetic -- and the kernel -- of ze_prefill)
    {
        std::vector<const bf16_t*> xr(n);
        std::vector<float*> lo(n);
        row0 = 0;
        for (int i = 0; i < n; ++i) {
            xr[i] = e->th + (size_t)(row0 + lens[i] - 1) * H;
            lo[i] = e->dlogits + (size_t)seqs[i] * c.vocab;
            row0 += lens[i];
        }
        if (!ze_launch_logits_rows(e->lm_head, H, c.vocab, H, e->final_norm, c.rms_eps, xr.data(), lo.data(), n, s)) {
            for (int i = 0; i < n; ++i) {
                ze_gemv_args a;
                memset(&a, 0, sizeof(a));
                a.W = e->lm_head;
                a.ldw = H;
                a.N = c.vocab;
                a.K = H;
                a.x = xr[i];
                a.norm_w = e->final_norm;
                a.eps = c.rms_eps;
                a.out_f32 = lo[i];
                a.D = hd;
                ze_launch_gemv(ZE_GV_LOGITS, a, s);
            }
        }
    }
    ze_timer_end(e, th, s);
    ZE_KCHECK();
    row0 = 0;
    for (int i = 0; i < n; ++i) {
        e->ctx_host[seqs[i]] += lens[i];
        e->delta_host[seqs[i]] = rope_deltas[i];
        ZE_TRY(push_state(e, seqs[i], s, input_ids[row0 + lens[i] - 1], 0, 0));
        row0 += lens[i];
    }
    return ZE_OK;
}

// ================================================================== decode
// One token for chain `seq`: everything is read from the device-side chain state, so the same launch
// sequence can be captured once into a hipGraph and replayed.
int ze_enqueue_decode_step(ze_engine* e, int seq, float penalty, int ignore_eos, bool sample, const ze_sample_opts& so,
                           hipStream_t s) {
    const ze_config& c = e->cfg;
    const int H = c.hidden, hd = e->head_dim, nq = c.heads * hd, nqkv = nq + 2 * c.kv_heads * hd;
    const ze_seq_dev* st = e->st_dev + seq;
    const float scale = 1.0f / sqrtf((float)hd);
    for (int li = 0; li < c.layers; ++li) {
        const ze_text_layer& L = e->tl[li];
        {
        ze_gemv_args a;
        memset(&a, 0, sizeof(a));
        a.W = L.qkv.w;
        a.ldw = L.qkv.ld;
        if (e->fp8_ready) {
            a.W8 = L.qkv.w8;
            a.scale8 = L.qkv.scale8;
            a.ldw8 = L.qkv.ld8;
        }
        a.N = nqkv;
        a.K = H;
        a.x = e->dh;
        a.norm_w = L.in_norm;
        a.eps = c.rms_eps;
        a.bias = L.qkv.bias;
        a.out_bf16 = e->dq;
        a.st = st;
        a.cosT = e->cosT;
        a.sinT = e->sinT;
        a.kcache = e->kc(li, seq);
        a.vcache = e->vc(li, seq);
        a.heads = c.heads;
        a.kv_heads = c.kv_heads;
        a.D = hd;
        a.max_ctx = c.max_ctx;
        a.act8 = e->fp8_act ? 1 : 0;
        if (li == 0) {
            a.embed = e->embed;
            a.embed_out = e->dh;
        }
        ze_launch_gemv(ZE_GV_QKV_ROPE, a, s);
        ze_launch_attn_decode(e->dq, 0, e->kc(li, seq), e->vc(li, seq), 0, e->dattn, 0, st, nullptr, 1, c.heads,
                              c.kv_heads, hd, c.max_ctx, scale, e->dpartial, e->max_splits, e->atickets, s);
        ze_gemv_args o;
        memset(&o, 0, sizeof(o));
        o.W = L.o.w;
        o.ldw = L.o.ld;
        if (e->fp8_ready) {
            o.W8 = L.o.w8;
            o.scale8 = L.o.scale8;
            o.ldw8 = L.o.ld8;
        }
        o.N = H;
        o.K = nq;
        o.x = e->dattn;
        o.out_bf16 = e->dh;
        o.D = hd;
        ze_launch_gemv(ZE_GV_RESIDUAL, o, s);
        }
        ze_gemv_args g;
        memset(&g, 0, sizeof(g));
        g.W = L.gate_up.w;
        g.ldw = L.gate_up.ld;
        if (e->fp8_ready) {
            g.W8 = L.gate_up.w8;
            g.scale8 = L.gate_up.scale8;
            g.ldw8 = L.gate_up.ld8;
        }
        g.N = 2 * e->text_ipad;
        g.K = H;
        g.x = e->dh;
        g.norm_w = L.post_norm;
        g.eps = c.rms_eps;
        g.out_bf16 = e->dact;
        g.D = hd;
        g.act8 = e->fp8_act ? 1 : 0;
        ze_launch_gemv(ZE_GV_SWIGLU, g, s);
        ze_gemv_args d;
        memset(&d, 0, sizeof(d));
        d.W = L.down.w;
        d.ldw = L.down.ld;
        if (e->fp8_ready) {
            d.W8 = L.down.w8;
            d.scale8 = L.down.scale8;
            d.ldw8 = L.down.ld8;
        }
        d.N = H;
        d.K = e->text_ipad;
        d.x = e->dact;
        d.out_bf16 = e->dh;
        d.D = hd;
        ze_launch_gemv(ZE_GV_RESIDUAL, d, s);
    }
    ze_gemv_args a;
    memset(&a, 0, sizeof(a));
    a.W = e->lm_head;
    a.ldw = H;
    if (e->fp8_ready && e->lm_head8.w8) {
        a.W8 = e->lm_head8.w8;
        a.scale8 = e->lm_head8.scale8;
        a.ldw8 = e->lm_head8.ld8;
    }
    a.N = c.vocab;
    a.K = H;
    a.x = e->dh;
    a.norm_w = e->final_norm;
    a.eps = c.rms_eps;
    a.out_f32 = e->dlogits + (size_t)seq * c.vocab;
    a.D = hd;
    // greedy: the arg-max partials come out of the lm_head launch itself (knob 14 = 1: the separate partial kernel)
    const bool folded = sample && so.temperature <= 0.f && ze_gemv_knobs[14] != 1;
    if (folded) {
        a.seen = e->seen + (size_t)seq * c.vocab;
        a.penalty = penalty;
        a.amax_ws = e->damax;
    }
    const bool gemv_ok = ze_launch_gemv(ZE_GV_LOGITS, a, s);
    if (folded && gemv_ok)
        ze_launch_sample_folded(e->damax, c.vocab, e->seen + (size_t)seq * c.vocab, e->st_dev + seq, e->eos_dev, c.n_eos,
                                c.pad_token_id, ignore_eos, /*advance_ctx=*/1, e->out_tokens + (size_t)seq * c.max_ctx, s);
    else if (sample)
        ze_launch_sample(e->dlogits + (size_t)seq * c.vocab, c.vocab, e->seen + (size_t)seq * c.vocab, penalty, e->st_dev + seq, e->eos_dev,
                         c.n_eos, c.pad_token_id, ignore_eos, /*advance_ctx=*/1,
                         e->out_tokens + (size_t)seq * c.max_ctx, e->dsample, so, s);
    else
        ze_launch_advance_ctx(e->st_dev + seq, s);  // teacher forcing: the caller chooses the next token
    ZE_KCHECK();
    return ZE_OK;
}

extern "C" int ze_decode_step(ze_engine* e, int seq, int token, float* out_logits, void* stream) {
    ZE_TRY(check_seq(e, seq));
    const ze_config& c = e->cfg;
    if (e->ctx_host[seq] + 1 > c.max_ctx) return ze_fail(e, ZE_ERR_NOMEM, "sequence exceeds max_ctx");
    if (token >= c.vocab) return ze_fail(e, ZE_ERR_INVALID, "token id out of range");
    hipStream_t s = (hipStream_t)stream;
    hipSetDevice(e->device);
    if (token >= 0) ze_launch_set_ints(&(e->st_dev + seq)->token, &token, 1, s);
    const int th = ze_timer_begin(e, 3, s);
    ZE_TRY(ze_enqueue_decode_step(e, seq, 1.0f, 1, false, ze_sample_opts{}, s));
    ze_timer_end(e, th, s);
    e->ctx_host[seq] += 1;
    if (out_logits)
        ZE_HIP(hipMemcpyAsync(out_logits, e->dlogits + (size_t)seq * c.vocab, (size_t)c.vocab * sizeof(float),
                              hipMemcpyDeviceToDevice, s));
    return ZE_OK;
}

// one sampling step on caller-supplied logits; `index` plays the role of the generated-token index of the draw
static int op_sample(ze_engine* e, int seq, const float* logits, float repetition_penalty, const ze_sample_opts& so,
                     int index, int32_t* out_token, hipStream_t s) {
    ZE_TRY(check_seq(e, seq));
    if (!logits || !out_token) return ze_fail(e, ZE_ERR_INVALID, "null argument");
    if (index < 0 || index >= e->cfg.max_ctx) return ze_fail(e, ZE_ERR_INVALID, "index out of range");
    const ze_config& c = e->cfg;
    hipSetDevice(e->device);
    // n_gen is set so the sampled token lands in out_tokens[index] of this chain's slot
    ZE_TRY(push_state(e, seq, s, 0, 0, 0));
    if (index) ze_launch_set_ints(&(e->st_dev + seq)->n_gen, &index, 1, s);
    ze_launch_sample(logits, c.vocab, e->seen + (size_t)seq * c.vocab, repetition_penalty, e->st_dev + seq, e->eos_dev,
                     c.n_eos, c.pad_token_id, 1, 0, e->out_tokens + (size_t)seq * c.max_ctx, e->dsample, so, s);
    ZE_KCHECK();
    ZE_HIP(hipMemcpyAsync(out_token, e->out_tokens + (size_t)seq * c.max_ctx + index, sizeof(int), hipMemcpyDeviceToHost, s));
    ZE_HIP(hipStreamSynchronize(s));
    return ZE_OK;
}

extern "C" int ze_op_sample_greedy(ze_engine* e, int seq, const float* logits, float repetition_penalty,
                                   int32_t* out_token, void* stream) {
    return op_sample(e, seq, logits, repetition_penalty, ze_sample_opts{}, 0, out_token, (hipStream_t)stream);
}

extern "C" int ze_op_sample_temperature(ze_engine* e, int seq, const float* logits, float repetition_penalty,
                                        float temperature, uint64_t seed, int index, int32_t* out_token, void* stream) {
    if (!(temperature > 0.f)) return ze_fail(e, ZE_ERR_INVALID, "temperature must be positive");
    ze_sample_opts so;
    so.temperature = temperature;
    so.seed = seed;
    so.slot = seq;
    return op_sample(e, seq, logits, repetition_penalty, so, index, out_token, (hipStream_t)stream);
}

static ze_sample_opts sample_opts_of(const ze_gen_params* p, int slot) {
    ze_sample_opts so;
    if (p->do_sample && p->temperature > 0.f) {
        so.temperature = p->temperature;
        so.seed = p->seed;
    }
    so.slot = slot;
    return so;
}

extern "C" int ze_generate(ze_engine* e, int seq, const ze_gen_params* p, int32_t* out_tokens, int* n_out,
                           void* stream) {
    ZE_TRY(check_seq(e, seq));
    if (!p || !out_tokens || !n_out) return ze_fail(e, ZE_ERR_INVALID, "null argument");
    const ze_config& c = e->cfg;
    hipStream_t s = (hipStream_t)stream;
    hipSetDevice(e->device);
    int max_new = p->max_new_tokens;
    if (max_new <= 0) {
        *n_out = 0;
        return ZE_OK;
    }
    // the last generated token is never fed back, so ctx grows by max_new - 1
    if (e->ctx_host[seq] + max_new - 1 > c.max_ctx) max_new = c.max_ctx - e->ctx_host[seq] + 1;
    if (max_new <= 0) return ze_fail(e, ZE_ERR_NOMEM, "sequence exceeds max_ctx");
    const float pen = p->repetition_penalty > 0.f ? p->repetition_penalty : 1.0f;
    const int ign = p->ignore_eos ? 1 : 0;
    int32_t* dev_out = e->out_tokens + (size_t)seq * c.max_ctx;
    ze_seq_dev* st = e->st_dev + seq;

    const int t_s = ze_timer_begin(e, 4, s);
    // first token from the prefill logits (no cache growth)
    const ze_sample_opts so = sample_opts_of(p, seq);
    ze_launch_sample(e->dlogits + (size_t)seq * c.vocab, c.vocab, e->seen + (size_t)seq * c.vocab, pen, st, e->eos_dev,
                     c.n_eos, c.pad_token_id, ign, 0, dev_out, e->dsample, so, s);
    ze_timer_end(e, t_s, s);
    ZE_KCHECK();

    // decode-step graph for this chain (re-captured when the sampling options change)
    hipGraphExec_t gexec = nullptr;
    if (p->use_graph && max_new > 1) {
        if (!e->graphs[seq] || e->graph_penalty[seq] != pen || e->graph_ignore_eos[seq] != ign ||
            e->graph_variant[seq] != (int)ze_tune_epoch || e->graph_temperature[seq] != so.temperature ||
            e->graph_seed[seq] != so.seed) {
            if (e->graphs[seq]) {
                hipGraphExecDestroy(e->graphs[seq]);
                e->graphs[seq] = nullptr;
            }
            hipStream_t cs;
            ZE_HIP(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
            hipGraph_t graph = nullptr;
            int r = ZE_OK;
            if (hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal) != hipSuccess)
                r = ze_fail(e, ZE_ERR_HIP, "hipStreamBeginCapture failed");
            if (r == ZE_OK) r = ze_enqueue_decode_step(e, seq, pen, ign, true, so, cs);
            if (hipStreamEndCapture(cs, &graph) != hipSuccess && r == ZE_OK)
                r = ze_fail(e, ZE_ERR_HIP, "hipStreamEndCapture failed");
            if (r == ZE_OK && hipGraphInstantiate(&e->graphs[seq], graph, nullptr, nullptr, 0) != hipSuccess)
                r = ze_fail(e, ZE_ERR_HIP, "hipGraphInstantiate failed");
            if (graph) hipGraphDestroy(graph);
            hipStreamDestroy(cs);
            ZE_TRY(r);
            e->graph_penalty[seq] = pen;
            e->graph_ignore_eos[seq] = ign;
            e->graph_variant[seq] = (int)ze_tune_epoch;
            e->graph_temperature[seq] = so.temperature;
            e->graph_seed[seq] = so.seed;
        }
        gexec = e->graphs[seq];
    }

    const int sync_every = std::max(1, p->sync_every);
    int produced = 1;  // tokens sampled so far (device side)
    int finished = 0;
    const int t_d = ze_timer_begin(e, 3, s);
    while (produced < max_new && !finished) {
        const int burst = std::min(sync_every, max_new - produced);
        for (int i = 0; i < burst; ++i) {
            if (gexec)
                ZE_HIP(hipGraphLaunch(gexec, s));
            else
                ZE_TRY(ze_enqueue_decode_step(e, seq, pen, ign, true, so, s));
        }
        produced += burst;
        e->ctx_host[seq] += burst;
        if (!ign && produced < max_new) {
            ZE_HIP(hipMemcpyAsync(e->d_host_ints + 32, &st->finished, sizeof(int), hipMemcpyDeviceToHost, s));
            ZE_HIP(hipStreamSynchronize(s));
            finished = e->d_host_ints[32];
        }
    }
    ze_timer_end(e, t_d, s);
    ZE_HIP(hipMemcpyAsync(out_tokens, dev_out, (size_t)produced * sizeof(int), hipMemcpyDeviceToHost, s));
    ZE_HIP(hipStreamSynchronize(s));
    // trim at the first EOS (tokens after it are pad, as HF emits for finished rows)
    int n = produced;
    if (!ign) {
        for (int i = 0; i < produced; ++i) {
            bool is_eos = false;
            for (int k = 0; k < c.n_eos; ++k) is_eos |= out_tokens[i] == c.eos_token_ids[k];
            if (is_eos) {
                n = i + 1;
                break;
            }
        }
    }
    *n_out = n;
    return ZE_OK;
}

// ================================================================== batched decode (BASELINE configs[2])
// One token for each of n chains per step: the weights are streamed ONCE for the whole batch through the MFMA GEMM
// path (rows = chains), each chain keeps its own KV cache / position / seen-set.  Rows are computed independently of
// the batch composition (same per-element accumulation order for every tile shape), so a chain's tokens do not
// depend on which other chains share its steps.
// Fragment-major copies of the wide, short-K decode projections for the batched step (see ze_engine.h).  Shapes the
// fragment kernel does not cover (rows % 16, K % 32, K > 4096) keep wf = null and stay on the row-major launchers.
static int ensure_fragments(ze_engine* e, hipStream_t s) {
    if (e->frag_ready) return ZE_OK;
    const ze_config& c = e->cfg;
    const int H = c.hidden, hd = e->head_dim, nq = c.heads * hd, nqkv = nq + 2 * c.kv_heads * hd, ip = e->text_ipad;
    const bool ok = H % 32 == 0 && H <= 4096 && nq % 32 == 0 && nq <= 4096 && nqkv % 16 == 0 && (2 * ip) % 32 == 0 && c.vocab % 16 == 0;
    for (int li = 0; li < c.layers; ++li) {
        ze_text_layer& L = e->tl[li];
        L.qkv.wf = L.o.wf = L.gate_up.wf = nullptr;
        L.qkv.wf8 = L.o.wf8 = L.gate_up.wf8 = nullptr;   // (these stayed stale in the row-streaming regime)
        L.qkv.wp = L.qkv.bias_p = nullptr;
    }
    e->lm_head_f = nullptr;
    e->lm_head8.wf8 = nullptr;
    // Each kernel family gets only the copies it reads: the row-streaming regime the permuted qkv rows (0.38 GB at the
    // 3B shape), the fragment family the fragment-major arena (4.5 GB) -- an engine of 64 slots never runs the first, one of 768
    // never the second.  ze_set_decode_regime clears frag_ready, so a regime change rebuilds what the new family needs.
    const bool wide = e->wide_regime();
    // ... and the OTHER family's copies are given back (after ze_set_decode_regime the 4.5-GB fragment arena, its FP8 twin
    // and the permuted qkv rows all stayed allocated).  A regime change is rare and never concurrent with a step: wait for the device.
    if (wide ? (e->arena_f || e->arena_f8) : (e->arena_p != nullptr)) {
        ZE_HIP(hipDeviceSynchronize());
        if (wide) {
            if (e->arena_f) hipFree(e->arena_f);
            if (e->arena_f8) hipFree(e->arena_f8);
            e->arena_f = nullptr;
            e->arena_f8 = nullptr;
        } else {
            hipFree(e->arena_p);
            e->arena_p = nullptr;
        }
    }
    // row-streaming regime: the qkv rows permuted per head, so that M-RoPE + the KV append run as the projection's epilogue
    if (wide && hd == 128 && H % 64 == 0 && H / 64 >= 4 && nqkv % 128 == 0) {
        const size_t per_layer = (size_t)nqkv * H + nqkv;
        if (!e->arena_p) ZE_HIP(hipMalloc((void**)&e->arena_p, per_layer * c.layers * sizeof(bf16_t)));
        if (!e->qkv_epi_dev) ZE_HIP(hipMalloc((void**)&e->qkv_epi_dev, sizeof(ze_qkv_epi) * c.layers));
        std::vector<ze_qkv_epi> host(c.layers);
        for (int li = 0; li < c.layers; ++li) {
            ze_text_layer& L = e->tl[li];
            bf16_t* wp = e->arena_p + per_layer * li;
            bf16_t* bp = wp + (size_t)nqkv * H;
            ze_launch_permute_qkv(L.qkv.w, L.qkv.ld, L.qkv.bias, nqkv / 128, H, wp, L.qkv.bias ? bp : nullptr, s);
            L.qkv.wp = wp;
            L.qkv.bias_p = L.qkv.bias ? bp : nullptr;
            ze_qkv_epi& q = host[li];
            memset(&q, 0, sizeof(q));
            q.st = e->st_dev;
            q.seq_ids = e->bseq;
            q.cosT = e->cosT;
            q.sinT = e->sinT;
            q.kcache = e->kc(li, 0);
            q.vcache = e->vc(li, 0);
            q.cache_seq_stride = (size_t)c.kv_heads * c.max_ctx * hd;
            q.max_ctx = c.max_ctx;
            q.heads = c.heads;
            q.kv_heads = c.kv_heads;
        }
        ZE_HIP(hipMemcpyAsync(e->qkv_epi_dev, host.data(), sizeof(ze_qkv_epi) * c.layers, hipMemcpyHostToDevice, s));
        ZE_HIP(hipStreamSynchronize(s));  // (host is a local)
        ZE_KCHECK();
    }
    if (ok && !wide) {
        const size_t per_layer = (size_t)(nqkv + 2 * ip) * H + (size_t)H * nq;
        const size_t total = per_layer * c.layers + (size_t)c.vocab * H;
        if (!e->arena_f) ZE_HIP(hipMalloc((void**)&e->arena_f, total * sizeof(bf16_t)));
        bf16_t* cur = e->arena_f;
        auto pack = [&](ze_linear& l, int rows, int cols, int rope_dim) {
            ze_launch_pack_fragments(l.w, l.ld, rows, cols, cur, s, rope_dim);
            l.wf = cur;
            cur += (size_t)rows * cols;
        };
        for (int li = 0; li < c.layers; ++li) {
            ze_text_layer& L = e->tl[li];
            pack(L.qkv, nqkv, H, hd);  // rows permuted per head for the fused M-RoPE epilogue (ze_gemm_oneshot.hip)
            pack(L.o, H, nq, 0);
            pack(L.gate_up, 2 * ip, H, 0);
        }
        ze_launch_pack_fragments(e->lm_head, H, c.vocab, H, cur, s);
        e->lm_head_f = cur;
        // FP8 engine: the same fragments at one byte per weight (qkv, o, gate/up, an untied lm_head); the batched step
        // streams these and dequantises in registers -- identical values, half the weight bytes
        for (int li = 0; li < c.layers; ++li) {
            ze_text_layer& L = e->tl[li];
            L.qkv.wf8 = L.o.wf8 = L.gate_up.wf8 = nullptr;
        }
        e->lm_head8.wf8 = nullptr;
        if (e->fp8_ready) {
            const bool head8 = e->lm_head8.w8 != nullptr;
            const size_t bytes = ((size_t)(nqkv + 2 * ip) * H + (size_t)H * nq) * c.layers + (head8 ? (size_t)c.vocab * H : 0);
            if (!e->arena_f8) ZE_HIP(hipMalloc((void**)&e->arena_f8, bytes));
            uint8_t* c8 = e->arena_f8;
            auto pack8 = [&](ze_linear& l, int rows, int cols, int rope_dim) {
                ze_launch_pack_fragments8(l.w8, l.ld8, rows, cols, c8, s, rope_dim);
                l.wf8 = c8;
                c8 += (size_t)rows * cols;
            };
            for (int li = 0; li < c.layers; ++li) {
                ze_text_layer& L = e->tl[li];
                pack8(L.qkv, nqkv, H, hd);
                pack8(L.o, H, nq, 0);
                pack8(L.gate_up, 2 * ip, H, 0);
            }
            if (head8) pack8(e->lm_head8, c.vocab, H, 0);
        }
        ZE_KCHECK();
    }
    e->frag_ready = true;
    return ZE_OK;
}

// the attention grid's extent for the next `steps` decode steps of these chains: the parts of the longest context
static void set_live_parts(ze_engine* e, const int32_t* seqs, int n, int steps) {
    int mx = 0, ml = 0;
    for (int i = 0; i < n; ++i) {
        const int c = e->ctx_host[seqs[i]] + std::max(1, steps) + 1, sp = e->split_host[seqs[i]];
        mx = std::max(mx, e->ctx_host[seqs[i]]);
        ml = std::max(ml, (sp + 383) / 384 + (c - sp + 383) / 384);   // (the pipelined kernel's 384-key parts under the chain's split)
    }
    e->live_parts = (mx + std::max(1, steps) + 1 + 191) / 192;
    e->live_parts_long = ml;
}

// Round 6: which rows of the step share their PREFIX parts (ze_attn_batch.hip).  Two chains pair when they have the same split row and
// read the rows below it from the same holder -- the holder itself included -- under the hints this very step uses (sync_prefix ran on
// this stream just before): first come, first paired, in batch order (the questions of a tile sit next to each other).  Symmetric;
// -1 = alone.  A pairing changes who computes a partial, never its bits.
static void upload_mates(ze_engine* e, const int32_t* seqs, int n, hipStream_t s) {
    std::vector<int> mate(n, -1);
    if (e->prefix_hints && ze_gemv_knobs[17] != 1) {
        std::map<long long, int> open;   // (holder, split) -> the row waiting for a partner
        for (int i = 0; i < n; ++i) {
            const int q = seqs[i], sp = e->split_host[q];
            if (sp <= 0) continue;
            const int h = e->pfx_pushed[q];
            const int holder = (h != 0 && (h & 0xffff) >= sp) ? (h >> 16) : q;
            const long long key = ((long long)holder << 20) | (long long)sp;
            auto it = open.find(key);
            if (it == open.end()) open[key] = i;
            else {
                mate[i] = it->second;
                mate[it->second] = i;
                open.erase(it);
            }
        }
    }
    ze_launch_set_ints(e->bmate, mate.data(), n, s);
}

static int upload_batch(ze_engine* e, const int32_t* seqs, int n, hipStream_t s) {
    if (n <= 0 || n > e->cfg.max_seqs) return ze_fail(e, ZE_ERR_INVALID, "batch size out of range");
    if (n > 64 && !e->wide_regime())
        return ze_fail(e, ZE_ERR_INVALID, "more than 64 chains in a step of an engine pinned to the fragment kernels (ze_set_decode_regime)");
    for (int i = 0; i < n; ++i) {
        if (seqs[i] < 0 || seqs[i] >= e->cfg.max_seqs) return ze_fail(e, ZE_ERR_NOTFOUND, "sequence id out of range");
        for (int j = 0; j < i; ++j)
            if (seqs[j] == seqs[i]) return ze_fail(e, ZE_ERR_INVALID, "duplicate sequence id in a batch");
        if (e->ctx_host[seqs[i]] + 1 > e->cfg.max_ctx) return ze_fail(e, ZE_ERR_NOMEM, "sequence exceeds max_ctx");
    }
    ze_launch_set_ints(e->bseq, seqs, n, s);
    sync_prefix(e, seqs, n, s);
    upload_mates(e, seqs, n, s);
    return ZE_OK;
}

// decode attention of the batched step: the per-wave streaming kernel k_attn_decode_wave (ze_attn_batch.hip); knob 8 = 2:
// its predecessor k_attn_decode_stream (a workgroup-wide LDS-DMA ring); knob 8 = 1: the 64-token-slice kernel of the
// single-chain step with a chain dimension (the round-1 form) -- both kept for A/B measurements
// (q_rows / out_rows: the step's own buffers, or the caller's -- ze_op_attn_decode)
static void launch_batch_attention(ze_engine* e, int li, int n, bool frag_out, hipStream_t s, const bf16_t* q_rows = nullptr,
                                   bf16_t* out_rows = nullptr) {
    const ze_config& c = e->cfg;
    const int hd = e->head_dim, nq = c.heads * hd, nqkv = nq + 2 * c.kv_heads * hd;
    const size_t seq_stride = (size_t)c.kv_heads * c.max_ctx * hd;
    const float scale = 1.0f / sqrtf((float)hd);
    const bf16_t* qb = q_rows ? q_rows : e->bqkv;
    bf16_t* ob = out_rows ? out_rows : e->bo;
    if (ze_gemv_knobs[8] == 1)
        ze_launch_attn_decode(qb, nqkv, e->kc(li, 0), e->vc(li, 0), seq_stride, ob, frag_out ? -(nq / 32) : nq, e->st_dev,
                              e->bseq, n, c.heads, c.kv_heads, hd, c.max_ctx, scale, e->bpartial, e->max_splits, e->atickets, s);
    else {
        // tokens per part (a multiple of 32; knob 11 for measurements): a function of nothing but the build, so a chain's
        // partition depends on its own context length alone
        // the per-wave kernel cuts 192-key parts whatever the context: ceil(max_ctx / 192) of them must fit the partial buffer
        // (max(max_splits, 8) parts per chain) and the merge's 64 lanes
        const int wparts = (c.max_ctx + 191) / 192;
        const bool per_wave = ze_gemv_knobs[8] != 2 && ze_gemv_knobs[11] == 0 && wparts <= std::max(e->max_splits, 8) && wparts <= 64 &&
                              hd == 128 && c.heads / c.kv_heads <= 16;  // (128-wide heads, the q heads of a kv head as MFMA columns)
        const int chunk = ze_gemv_knobs[11] >= 64 ? ze_gemv_knobs[11] / 32 * 32 : 0;  // 0: a sixth of the chain's context
        const int max_parts = chunk ? (c.max_ctx + chunk - 1) / chunk : 8;           // (at most 8 parts: 128-token floor)
        ze_launch_attn_decode_stream(qb, nqkv, e->kc(li, 0), e->vc(li, 0), seq_stride, ob, frag_out ? -(nq / 32) : nq,
                                     e->st_dev, e->bseq, n, c.heads, c.kv_heads, c.max_ctx, scale, e->bpartial, max_parts,
                                     e->atickets, s, chunk, per_wave ? (e->live_parts > 0 ? e->live_parts : wparts) : 0, e->pfx_dev,
                                     q_rows ? nullptr : e->bmate, e->live_parts_long, 1);
    }
}

static int enqueue_decode_batch(ze_engine* e, int n, float penalty, int ignore_eos, int sample, const ze_sample_opts& so,
                                hipStream_t s) {
    const ze_config& c = e->cfg;
    const int H = c.hidden, hd = e->head_dim, nq = c.heads * hd, nkv = c.kv_heads * hd, nqkv = nq + 2 * nkv;
    const size_t seq_stride = (size_t)c.kv_heads * c.max_ctx * hd;
    ze_launch_embed_tokens_batch(e->st_dev, e->bseq, n, e->embed, e->bh, H, s);
    for (int li = 0; li < c.layers; ++li) {
        const ze_text_layer& L = e->tl[li];
        // every projection on fragment-major operands when the copy exists (ensure_fragments) and n <= 64: the norms, the
        // attention merge and the SwiGLU epilogue then write their outputs in that layout too
        const bool fr = L.qkv.wf && !e->wide_regime() && ze_gemv_knobs[5] != 1;
        const ze_gemm_ws ws = e->gemm_ws();
        // Beyond 64 chains (no fragment kernels): the prefill tile policy for qkv / o / gate-up / lm_head (one pass over K
        // per output tile, no split: at 256 chains qkv 12.0 against 24.7 us on the split-K streaming launcher, o 11.9 /
        // 17.7, gate/up 43.6 / 56.9, lm_head 250 / 336; tools/bench_midm.py), the streaming launcher for down (32.4 / 44.3).
        // Every one of them sums an output's K range in an order fixed by (N, K): batch invariance within this path.
        const bool tiled = e->wide_regime() && ze_gemv_knobs[13] != 1;
        // FP8 activations: the fragment path with FP8 weight fragments takes the row as FP8 fragments + a scale (fp8 x
        // fp8 MFMA); any other path takes the same values as bf16
        const bool a8q = e->fp8_act && fr && L.qkv.wf8 && ze_gemv_knobs[10] != 1;
        const bool a8g = e->fp8_act && fr && L.gate_up.wf8 && ze_gemv_knobs[10] != 1;
        ze_launch_rmsnorm(e->bh, H, L.in_norm, e->by, H, n, H, c.rms_eps, s, fr ? 1 : 2, a8q ? 2 : (e->fp8_act ? 1 : 0), e->ty8,
                          e->ty8_scale);
        if (fr) {  // projection + M-RoPE + KV append in one launch (the fragment copy of qkv is packed for it)
            const bool w8 = L.qkv.wf8 && ze_gemv_knobs[10] != 1;  // FP8 fragment stream (quantised engine)
            ze_launch_qkv_rope_oneshot(a8q ? (const bf16_t*)e->ty8 : e->by, w8 ? (const bf16_t*)L.qkv.wf8 : L.qkv.wf, L.qkv.bias,
                                       e->bqkv, nqkv, n, H, c.heads, c.kv_heads, e->cosT, e->sinT, e->st_dev, e->bseq,
                                       e->kc(li, 0), e->vc(li, 0), seq_stride, c.max_ctx, s, w8 ? L.qkv.scale8 : nullptr,
                                       a8q ? e->ty8_scale : nullptr);
        } else {
            // row streaming: projection + M-RoPE + KV append in ONE launch on the permuted rows (same bits as the pair of
            // launches below; knob 13 = 2 keeps the pair, for A/B runs and the bit-equality test)
            if (tiled && L.qkv.wp && ze_gemv_knobs[13] != 2) {
                ze_launch_gemm_qkv_rope(e->by, H, L.qkv.wp, H, L.qkv.bias_p, e->qkv_epi_dev + li, e->bqkv, nqkv, n, nqkv, H, s);
            } else {
                if (tiled) ze_launch_gemm_wide(ZE_EPI_NONE, e->by, H, L.qkv.w, L.qkv.ld, L.qkv.bias, nullptr, 0, e->bqkv, nqkv, n, nqkv, H, ws, s);
                else ze_launch_gemm_stream(ZE_EPI_NONE, e->by, H, L.qkv.w, L.qkv.ld, L.qkv.bias, nullptr, 0, e->bqkv, nqkv, n, nqkv, H, ws, s);
                ze_launch_rope_kv_batch(e->bqkv, n, c.heads, c.kv_heads, hd, e->cosT, e->sinT, e->st_dev, e->bseq, e->kc(li, 0),
                                        e->vc(li, 0), seq_stride, c.max_ctx, s);
            }
        }
        launch_batch_attention(e, li, n, fr, s);
        if (fr && ze_gemv_knobs[9] != 1) {
            const bool w8 = L.o.wf8 && ze_gemv_knobs[10] != 1;
            ze_launch_gemm_oneshot(ZE_EPI_RESIDUAL, e->bo, w8 ? (const bf16_t*)L.o.wf8 : L.o.wf, nullptr, e->bh, H, e->bh, H, n, H,
                                   nq, s, w8 ? L.o.scale8 : nullptr);
        } else if (fr)
            ze_launch_gemm_frag(ZE_EPI_RESIDUAL, e->bo, L.o.wf, nullptr, e->bh, H, e->bh, H, n, H, nq, s);
        else if (tiled)
            ze_launch_gemm_wide(ZE_EPI_RESIDUAL, e->bo, nq, L.o.w, L.o.ld, nullptr, e->bh, H, e->bh, H, n, H, nq, ws, s);
        else
            ze_launch_gemm_stream(ZE_EPI_RESIDUAL, e->bo, nq, L.o.w, L.o.ld, nullptr, e->bh, H, e->bh, H, n, H, nq, ws, s);
        ze_launch_rmsnorm(e->bh, H, L.post_norm, e->by, H, n, H, c.rms_eps, s, fr ? 1 : 2, a8g ? 2 : (e->fp8_act ? 1 : 0), e->ty8,
                          e->ty8_scale);
        if (fr) {
            const bool w8 = L.gate_up.wf8 && ze_gemv_knobs[10] != 1;
            ze_launch_gemm_frag(ZE_EPI_SWIGLU, a8g ? (const bf16_t*)e->ty8 : e->by, w8 ? (const bf16_t*)L.gate_up.wf8 : L.gate_up.wf,
                                nullptr, nullptr, 0, e->ba, e->text_ipad, n, 2 * e->text_ipad, H, s,
                                w8 ? L.gate_up.scale8 : nullptr, a8g ? e->ty8_scale : nullptr);
        } else if (tiled)
            ze_launch_gemm_wide(ZE_EPI_SWIGLU, e->by, H, L.gate_up.w, L.gate_up.ld, nullptr, nullptr, 0, e->ba, e->text_ipad, n,
                                2 * e->text_ipad, H, ws, s);
        else
            ze_launch_gemm_stream(ZE_EPI_SWIGLU, e->by, H, L.gate_up.w, L.gate_up.ld, nullptr, nullptr, 0, e->ba, e->text_ipad, n,
                                  2 * e->text_ipad, H, ws, s);
        // the down projection (K = 11008) stays on the split-K ring: the fragment kernel with K split over 8 x 32
        // workgroups measured 22.6-25.3 us against 17.9 (slab reduction included in both)
        if (tiled)
            ze_launch_gemm_wide(ZE_EPI_RESIDUAL, e->ba, e->text_ipad, L.down.w, L.down.ld, nullptr, e->bh, H, e->bh, H, n, H,
                                e->text_ipad, ws, s);
        else
            ze_launch_gemm_stream(ZE_EPI_RESIDUAL, e->ba, e->text_ipad, L.down.w, L.down.ld, nullptr, e->bh, H, e->bh, H, n, H,
                                  e->text_ipad, ws, s);
    }
    const bool fl = e->lm_head_f && !e->wide_regime() && ze_gemv_knobs[5] != 1;
    ze_launch_rmsnorm(e->bh, H, e->final_norm, e->by, H, n, H, c.rms_eps, s, fl ? 1 : 2);
    if (fl) {
        const bool w8 = e->lm_head8.wf8 && ze_gemv_knobs[10] != 1;
        ze_launch_gemm_frag(ZE_EPI_F32, e->by, w8 ? (const bf16_t*)e->lm_head8.wf8 : e->lm_head_f, nullptr, nullptr, 0,
                            (bf16_t*)e->blogits, c.vocab, n, c.vocab, H, s, w8 ? e->lm_head8.scale8 : nullptr);
    } else if (e->wide_regime() && ze_gemv_knobs[13] != 1)  // (one pass over K whatever the row count: batch invariance)
        ze_launch_gemm_wide(ZE_EPI_F32, e->by, H, e->lm_head, H, nullptr, nullptr, 0, (bf16_t*)e->blogits, c.vocab, n, c.vocab, H,
                            e->gemm_ws(), s);
    else
        ze_launch_gemm_stream(ZE_EPI_F32, e->by, H, e->lm_head, H, nullptr, nullptr, 0, (bf16_t*)e->blogits, c.vocab, n,
                              c.vocab, H, e->gemm_ws(), s);
    ze_launch_sample_batch(e->blogits, c.vocab, e->seen, penalty, e->st_dev, e->bseq, n, e->eos_dev, c.n_eos,
                           c.pad_token_id, ignore_eos, 1, sample, e->out_tokens, c.max_ctx, e->bsample,
                           e->bsample + (size_t)c.max_seqs * 2 * 128, so, s);
    ZE_KCHECK();
    return ZE_OK;
}

extern "C" int ze_decode_batch(ze_engine* e, const int32_t* seqs, int n, const int32_t* tokens, float* out_logits,
                               void* stream) {
    if (!e || !seqs) return ze_fail(e, ZE_ERR_INVALID, "null argument");
    const ze_config& c = e->cfg;
    hipStream_t s = (hipStream_t)stream;
    hipSetDevice(e->device);
    ZE_TRY(ensure_fragments(e, s));
    ZE_TRY(upload_batch(e, seqs, n, s));
    set_live_parts(e, seqs, n, 1);
    if (tokens) {
        for (int i = 0; i < n; ++i) {
            if (tokens[i] >= c.vocab) return ze_fail(e, ZE_ERR_INVALID, "token id out of range");
            if (tokens[i] >= 0) ze_launch_set_ints(&(e->st_dev + seqs[i])->token, &tokens[i], 1, s);
        }
    }
    const int th = ze_timer_begin(e, 3, s);
    ZE_TRY(enqueue_decode_batch(e, n, 1.0f, 1, 0, ze_sample_opts{}, s));
    ze_timer_end(e, th, s);
    for (int i = 0; i < n; ++i) e->ctx_host[seqs[i]] += 1;
    if (out_logits)
        ZE_HIP(hipMemcpyAsync(out_logits, e->blogits, (size_t)n * c.vocab * sizeof(float), hipMemcpyDeviceToDevice, s));
    return ZE_OK;
}

// The captured batched decode step for `na` chains (chain ids / positions live in device memory, so one graph per
// batch size and sampling setting serves every composition); nullptr in *out = run eagerly.
static int batch_step_graph(ze_engine* e, int na, float pen, int ign, const ze_sample_opts& bso, hipGraphExec_t* out) {
    auto key = std::make_tuple(na, pen, ign, bso.temperature, bso.seed, e->live_parts * 64 + e->live_parts_long);
    if (e->bgraph_epoch != ze_tune_epoch) {
        for (auto& kv : e->bgraphs) hipGraphExecDestroy(kv.second);
        e->bgraphs.clear();
        e->bgraph_epoch = ze_tune_epoch;
    }
    auto it = e->bgraphs.find(key);
    if (it == e->bgraphs.end()) {
        hipStream_t cs;
        ZE_HIP(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
        hipGraph_t graph = nullptr;
        int r = ZE_OK;
        if (hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal) != hipSuccess)
            r = ze_fail(e, ZE_ERR_HIP, "hipStreamBeginCapture failed");
        if (r == ZE_OK) r = enqueue_decode_batch(e, na, pen, ign, 1, bso, cs);
        if (hipStreamEndCapture(cs, &graph) != hipSuccess && r == ZE_OK)
            r = ze_fail(e, ZE_ERR_HIP, "hipStreamEndCapture failed");
        hipGraphExec_t ex = nullptr;
        if (r == ZE_OK && hipGraphInstantiate(&ex, graph, nullptr, nullptr, 0) != hipSuccess)
            r = ze_fail(e, ZE_ERR_HIP, "hipGraphInstantiate failed");
        if (graph) hipGraphDestroy(graph);
        hipStreamDestroy(cs);
        ZE_TRY(r);
        it = e->bgraphs.emplace(key, ex).first;
    }
    *out = it->second;
    return ZE_OK;
}

// `steps` sampled decode steps for the chains in `active` (all with room for them)
static int run_burst(ze_engine* e, const std::vector<int>& active, int steps, const ze_gen_params* p, float pen, int ign,
                     const ze_sample_opts& bso, hipStream_t s) {
    const int na = (int)active.size();
    ZE_TRY(upload_batch(e, active.data(), na, s));
    set_live_parts(e, active.data(), na, steps);
    hipGraphExec_t gx = nullptr;
    if (p->use_graph) ZE_TRY(batch_step_graph(e, na, pen, ign, bso, &gx));
    for (int i = 0; i < steps; ++i) {
        if (gx)
            ZE_HIP(hipGraphLaunch(gx, s));
        else
            ZE_TRY(enqueue_decode_batch(e, na, pen, ign, 1, bso, s));
    }
    for (int q : active) e->ctx_host[q] += steps;
    return ZE_OK;
}

// first token of a chain from the logits its prefill left behind; `sample_stream` = the chain's random stream
static int begin_chain(ze_engine* e, int q, const ze_gen_params* p, float pen, int ign, int sample_stream, hipStream_t s) {
    const ze_config& c = e->cfg;
    ze_sample_opts so = sample_opts_of(p, q);
    if (so.temperature > 0.f) ze_launch_set_ints(&(e->st_dev + q)->stream, &sample_stream, 1, s);
    ze_launch_sample(e->dlogits + (size_t)q * c.vocab, c.vocab, e->seen + (size_t)q * c.vocab, pen, e->st_dev + q, e->eos_dev,
                     c.n_eos, c.pad_token_id, ign, 0, e->out_tokens + (size_t)q * c.max_ctx, e->dsample, so, s);
    ZE_KCHECK();
    return ZE_OK;
}

extern "C" int ze_generate_batch(ze_engine* e, const int32_t* seqs, int n, const ze_gen_params* p, int32_t* out_tokens,
                                 int32_t* n_out, void* stream) {
    if (!e || !seqs || !p || !out_tokens || !n_out) return ze_fail(e, ZE_ERR_INVALID, "null argument");
    const ze_config& c = e->cfg;
    hipStream_t s = (hipStream_t)stream;
    hipSetDevice(e->device);
    const int max_new = p->max_new_tokens;
    if (max_new <= 0 || n <= 0) return ze_fail(e, ZE_ERR_INVALID, "max_new_tokens and n must be positive");
    ZE_TRY(ensure_fragments(e, s));
    // Per-chain budget, as ze_generate clamps it: the last generated token is never fed back, so a chain with `ctx`
    // cached tokens may produce max_ctx - ctx + 1 tokens.  A chain that reaches its budget leaves the batch like one
    // that hit EOS: what a request gets never depends on the other requests of its batch.
    std::vector<int> limit(c.max_seqs, 0);
    for (int i = 0; i < n; ++i) {
        if (seqs[i] < 0 || seqs[i] >= c.max_seqs) return ze_fail(e, ZE_ERR_NOTFOUND, "sequence id out of range");
        for (int j = 0; j < i; ++j)
            if (seqs[j] == seqs[i]) return ze_fail(e, ZE_ERR_INVALID, "duplicate sequence id in a batch");
        if (e->ctx_host[seqs[i]] > c.max_ctx) return ze_fail(e, ZE_ERR_NOMEM, "sequence exceeds max_ctx");
        limit[seqs[i]] = std::min(max_new, c.max_ctx - e->ctx_host[seqs[i]] + 1);
    }
    const float pen = p->repetition_penalty > 0.f ? p->repetition_penalty : 1.0f;
    const int ign = p->ignore_eos ? 1 : 0;
    const ze_sample_opts bso = sample_opts_of(p, 0);
    // sampling stream of a chain = its row in this call (reproducible per request)
    for (int i = 0; i < n; ++i) ZE_TRY(begin_chain(e, seqs[i], p, pen, ign, i, s));
    std::vector<int> active;
    std::vector<int> produced(c.max_seqs, 0);
    for (int i = 0; i < n; ++i) {
        produced[seqs[i]] = 1;
        if (limit[seqs[i]] > 1) active.push_back(seqs[i]);
    }
    const int sync_every = std::max(1, p->sync_every);
    const int td = ze_timer_begin(e, 3, s);
    int steps = 1;
    while (steps < max_new && !active.empty()) {
        int burst = std::min(sync_every, max_new - steps);
        for (int q : active) burst = std::min(burst, limit[q] - produced[q]);  // >= 1: exhausted chains were dropped
        ZE_TRY(run_burst(e, active, burst, p, pen, ign, bso, s));
        steps += burst;
        for (int q : active) produced[q] += burst;
        if (steps < max_new) {  // chains that emitted an EOS or used up their budget leave (continuous batching: others go on)
            if (!ign) {
                ZE_HIP(hipMemcpyAsync(e->bstate_host, e->st_dev, sizeof(ze_seq_dev) * c.max_seqs, hipMemcpyDeviceToHost, s));
                ZE_HIP(hipStreamSynchronize(s));
            }
            std::vector<int> still;
            for (int q : active)
                if ((ign || !e->bstate_host[q].finished) && produced[q] < limit[q]) still.push_back(q);
            active.swap(still);
        }
    }
    ze_timer_end(e, td, s);
    ZE_HIP(hipStreamSynchronize(s));
    for (int i = 0; i < n; ++i) {
        const int q = seqs[i];
        int32_t* dst = out_tokens + (size_t)i * max_new;
        // (on the caller's stream: a legacy-stream copy would synchronise with every other engine's stream of the process)
        ZE_HIP(hipMemcpyAsync(dst, e->out_tokens + (size_t)q * c.max_ctx, (size_t)produced[q] * sizeof(int), hipMemcpyDeviceToHost, s));
        ZE_HIP(hipStreamSynchronize(s));
        int cnt = produced[q];
        if (!ign) {
            for (int t = 0; t < produced[q]; ++t) {
                bool is_eos = false;
                for (int k = 0; k < c.n_eos; ++k) is_eos |= dst[t] == c.eos_token_ids[k];
                if (is_eos) {
                    cnt = t + 1;
                    break;
                }
            }
        }
        n_out[i] = cnt;
    }
    return ZE_OK;
}

// ================================================================== continuous batching (chains join and leave between bursts)
extern "C" int ze_chain_begin(ze_engine* e, int seq, const ze_gen_params* p, int sample_stream, void* stream) {
    ZE_TRY(check_seq(e, seq));
    if (!p) return ze_fail(e, ZE_ERR_INVALID, "null argument");
    hipSetDevice(e->device);
    hipStream_t s = (hipStream_t)stream;
    ZE_TRY(ensure_fragments(e, s));
    const float pen = p->repetition_penalty > 0.f ? p->repetition_penalty : 1.0f;
    const int t_s = ze_timer_begin(e, 4, s);
    const int r = begin_chain(e, seq, p, pen, p->ignore_eos ? 1 : 0, sample_stream, s);
    ze_timer_end(e, t_s, s);
    return r;
}

// A burst in two halves, so that the host can enqueue other work (the prefill / ViT round of the next newcomers, on ANOTHER
// stream) while the burst runs: _begin enqueues `steps` captured decode steps for the live chains and returns at once (the
// number of steps actually enqueued: every chain must have room for all of them); _end waits for the burst and reports
// each chain's token count and EOS flag.  ze_decode_burst = _begin + _end.
extern "C" int ze_decode_burst_begin(ze_engine* e, const int32_t* seqs, int n, int steps, const ze_gen_params* p, void* stream) {
    if (!e || !seqs || !p || n <= 0 || steps < 0) return ze_fail(e, ZE_ERR_INVALID, "bad burst arguments");
    const ze_config& c = e->cfg;
    hipSetDevice(e->device);
    hipStream_t s = (hipStream_t)stream;
    ZE_TRY(ensure_fragments(e, s));
    std::vector<int> active(seqs, seqs + n);
    for (int q : active) {
        if (q < 0 || q >= c.max_seqs) return ze_fail(e, ZE_ERR_NOTFOUND, "sequence id out of range");
        steps = std::min(steps, c.max_ctx - e->ctx_host[q]);  // every chain of the burst must have room for all its steps
    }
    if (steps < 0) return ze_fail(e, ZE_ERR_NOMEM, "sequence exceeds max_ctx");
    const float pen = p->repetition_penalty > 0.f ? p->repetition_penalty : 1.0f;
    const int ign = p->ignore_eos ? 1 : 0;
    const int td = ze_timer_begin(e, 3, s);
    if (steps > 0) ZE_TRY(run_burst(e, active, steps, p, pen, ign, sample_opts_of(p, 0), s));
    ze_timer_end(e, td, s);
    return steps;
}

extern "C" int ze_decode_burst_end(ze_engine* e, const int32_t* seqs, int n, int32_t* n_generated, int32_t* finished, void* stream) {
    if (!e || !seqs || n <= 0) return ze_fail(e, ZE_ERR_INVALID, "bad burst arguments");
    const ze_config& c = e->cfg;
    hipSetDevice(e->device);
    hipStream_t s = (hipStream_t)stream;
    ZE_HIP(hipMemcpyAsync(e->bstate_host, e->st_dev, sizeof(ze_seq_dev) * c.max_seqs, hipMemcpyDeviceToHost, s));
    ZE_HIP(hipStreamSynchronize(s));
    for (int i = 0; i < n; ++i) {
        if (seqs[i] < 0 || seqs[i] >= c.max_seqs) return ze_fail(e, ZE_ERR_NOTFOUND, "sequence id out of range");
        if (n_generated) n_generated[i] = e->bstate_host[seqs[i]].n_gen;
        if (finished) finished[i] = e->bstate_host[seqs[i]].finished;
    }
    return ZE_OK;
}

extern "C" int ze_decode_burst(ze_engine* e, const int32_t* seqs, int n, int steps, const ze_gen_params* p,
                               int32_t* n_generated, int32_t* finished, void* stream) {
    const int ran = ze_decode_burst_begin(e, seqs, n, steps, p, stream);
    if (ran < 0) return ran;
    ZE_TRY(ze_decode_burst_end(e, seqs, n, n_generated, finished, stream));
    return ran;
}

extern "C" int ze_chain_tokens(ze_engine* e, int seq, int32_t* out, int cap, int* n_out, void* stream) {
    ZE_TRY(check_seq(e, seq));
    if (!out || !n_out || cap < 0) return ze_fail(e, ZE_ERR_INVALID, "null argument");
    const ze_config& c = e->cfg;
    hipSetDevice(e->device);
    hipStream_t s = (hipStream_t)stream;
    ze_seq_dev st;
    ZE_HIP(hipMemcpyAsync(&st, e->st_dev + seq, sizeof(st), hipMemcpyDeviceToHost, s));
    ZE_HIP(hipStreamSynchronize(s));
    int n = std::min(std::min(st.n_gen, cap), c.max_ctx);
    if (n > 0) {  // (on the caller's stream: a legacy-stream copy would synchronise with every other engine's stream of the process)
        ZE_HIP(hipMemcpyAsync(out, e->out_tokens + (size_t)seq * c.max_ctx, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, s));
        ZE_HIP(hipStreamSynchronize(s));
    }
    // trim at the first EOS (tokens after it are pad, as HF emits for finished rows)
    for (int i = 0; i < n; ++i) {
        bool is_eos = false;
        for (int k = 0; k < c.n_eos; ++k) is_eos |= out[i] == c.eos_token_ids[k];
        if (is_eos && st.finished) {
            n = i + 1;
            break;
        }
    }
    *n_out = n;
    return ZE_OK;
}

// ze_chain_tokens for the n chains a burst retires: one gather launch, one device -> host copy, one wait (the single-chain call
// costs two small copies and two waits per chain -- 1.5 ms each beside the 75-MB tile uploads of the stream).
// out_tokens: host int32 [n, capacity]; n_out [n].
extern "C" int ze_chain_tokens_batch(ze_engine* e, const int32_t* seqs, int n, int32_t* out, int cap, int32_t* n_out, void* stream) {
    if (!e || n < 0 || cap < 0 || (n > 0 && (!seqs || !out || !n_out))) return ze_fail(e, ZE_ERR_INVALID, "null argument");
    if (n == 0) return ZE_OK;
    if (n > e->cfg.max_seqs) return ze_fail(e, ZE_ERR_INVALID, "more chains than slots");
    const ze_config& c = e->cfg;
    for (int i = 0; i < n; ++i) ZE_TRY(check_seq(e, seqs[i]));
    cap = std::min(cap, c.max_ctx);
    hipSetDevice(e->device);
    hipStream_t s = (hipStream_t)stream;
    const size_t words = 2 * (size_t)n + (size_t)n * cap;
    ZE_TRY(xfer_reserve(e, e->xt_host, e->xt_dev, e->xt_cap, words + (size_t)n, nullptr, s));
    // (the call waits for the stream before it returns, so the scratch is free again by the next call)
    int* slots_dev = e->xt_dev + words;
    memcpy(e->xt_host + words, seqs, (size_t)n * sizeof(int));
    ZE_HIP(hipMemcpyAsync(slots_dev, e->xt_host + words, (size_t)n * sizeof(int), hipMemcpyHostToDevice, s));
    ze_launch_gather_chain_tokens(e->st_dev, e->out_tokens, c.max_ctx, slots_dev, n, cap, e->xt_dev, s);
    ZE_KCHECK();
    ZE_HIP(hipMemcpyAsync(e->xt_host, e->xt_dev, words * sizeof(int), hipMemcpyDeviceToHost, s));
    ZE_HIP(hipStreamSynchronize(s));
    for (int i = 0; i < n; ++i) {
        int m = e->xt_host[2 * i];
        const bool finished = e->xt_host[2 * i + 1] != 0;
        const int* row = e->xt_host + 2 * (size_t)n + (size_t)i * cap;
        for (int t = 0; t < m; ++t) {  // trim at the first EOS, as ze_chain_tokens
            out[(size_t)i * cap + t] = row[t];
            bool is_eos = false;
            for (int k = 0; k < c.n_eos; ++k) is_eos |= row[t] == c.eos_token_ids[k];
            if (is_eos && finished) {
                m = t + 1;
                break;
            }
        }
        n_out[i] = m;
    }
    return ZE_OK;
}

// ================================================================== unit ops
extern "C" int ze_op_linear(ze_engine* e, const void* a, const void* w, const void* bias, void* cmat, int M, int N,
                            int K, int act, void* stream) {
    if (!e || !a || !w || !cmat) return ze_fail(e, ZE_ERR_INVALID, "null argument");
    if (K % 8) return ze_fail(e, ZE_ERR_INVALID, "K must be a multiple of 8");
    hipSetDevice(e->device);
    hipStream_t s = (hipStream_t)stream;
    if (M == 1 && act == 0 && N % 2 == 0) {
        ze_gemv_args g;
        memset(&g, 0, sizeof(g));
        g.W = (const bf16_t*)w;
        g.ldw = K;
        g.N = N;
        g.K = K;
        g.x = (const bf16_t*)a;
        g.bias = (const bf16_t*)bias;
        g.out_bf16 = (bf16_t*)cmat;
        g.D = 128;
        if (!ze_launch_gemv(ZE_GV_PLAIN, g, s))
            ze_launch_gemm(ZE_EPI_NONE, (const bf16_t*)a, K, (const bf16_t*)w, K, (const bf16_t*)bias, nullptr, 0,
                           (bf16_t*)cmat, N, nullptr, M, N, K, s);
    } else if (act == 3 || act == 5) {  // the fragment-major kernels of the batched decode step, operands packed here
        if (M > 64 || N % 16 || K % 32 || K > 4096) return ze_fail(e, ZE_ERR_INVALID, "fragment path: M <= 64, N % 16, K % 32, K <= 4096");
        bf16_t *wf = nullptr, *xf = nullptr;
        const size_t mp = (size_t)(M + 15) / 16 * 16;
        ZE_HIP(hipMalloc((void**)&wf, ((size_t)N * K + mp * K) * sizeof(bf16_t)));
        xf = wf + (size_t)N * K;
        ze_launch_pack_fragments((const bf16_t*)w, K, N, K, wf, s);
        ze_launch_pack_fragments((const bf16_t*)a, K, M, K, xf, s);
        if (act == 5)  // the sixteen-wave one-shot kernel (qkv / o projections)
            ze_launch_gemm_oneshot(ZE_EPI_NONE, xf, wf, (const bf16_t*)bias, nullptr, 0, (bf16_t*)cmat, N, M, N, K, s);
        else
            ze_launch_gemm_frag(ZE_EPI_NONE, xf, wf, (const bf16_t*)bias, nullptr, 0, (bf16_t*)cmat, N, M, N, K, s);
        hipStreamSynchronize(s);
        hipFree(wf);
    } else if (act == 4) {  // SwiGLU epilogue on interleaved gate/up rows: C is [M, N/2]
        if (N % 32) return ze_fail(e, ZE_ERR_INVALID, "SwiGLU: N = 2 * width with width % 16 == 0");
        ze_launch_gemm(ZE_EPI_SWIGLU, (const bf16_t*)a, K, (const bf16_t*)w, K, (const bf16_t*)bias, nullptr, 0,
                       (bf16_t*)cmat, N / 2, nullptr, M, N, K, s, e->prefill_ws());
    } else if (act == 6 || act == 7) {  // the launcher of the row-streaming decode regime (7: SwiGLU, C is [M, N/2])
        if (act == 7 && N % 32) return ze_fail(e, ZE_ERR_INVALID, "SwiGLU: N = 2 * width with width % 16 == 0");
        ze_launch_gemm_wide(act == 7 ? ZE_EPI_SWIGLU : ZE_EPI_NONE, (const bf16_t*)a, K, (const bf16_t*)w, K, (const bf16_t*)bias,
                            nullptr, 0, (bf16_t*)cmat, act == 7 ? N / 2 : N, M, N, K, e->gemm_ws(), s);
    } else if (act == 10) {  // the lm_head of the row-streaming regime: fp32 output [M, N] through ze_launch_gemm_wide
        ze_launch_gemm_wide(ZE_EPI_F32, (const bf16_t*)a, K, (const bf16_t*)w, K, nullptr, nullptr, 0, (bf16_t*)cmat, N, M, N, K,
                            e->gemm_ws(), s);
    } else if (act == 8 || act == 9) {  // the eight-phase 256 x 256 kernel whatever the grid (9: SwiGLU), for its tests
        if (act == 9 && N % 32) return ze_fail(e, ZE_ERR_INVALID, "SwiGLU: N = 2 * width with width % 16 == 0");
        if (K % 64 || K < 64) return ze_fail(e, ZE_ERR_INVALID, "eight-phase kernel: K a multiple of 64");
        ze_launch_gemm_p8(act == 9 ? ZE_EPI_SWIGLU : ZE_EPI_NONE, (const bf16_t*)a, K, (const bf16_t*)w, K, (const bf16_t*)bias, nullptr, 0,
                          (bf16_t*)cmat, act == 9 ? N / 2 : N, nullptr, M, N, K, s, e->prefill_ws());
    } else if (act == 2) {  // weight-streaming mode of the batched decode step (rows = chains), for measurements
        ze_launch_gemm_stream(ZE_EPI_NONE, (const bf16_t*)a, K, (const bf16_t*)w, K, (const bf16_t*)bias, nullptr, 0,
                              (bf16_t*)cmat, N, M, N, K, e->gemm_ws(), s);
    } else {
        ze_launch_gemm(act ? ZE_EPI_GELU : ZE_EPI_NONE, (const bf16_t*)a, K, (const bf16_t*)w, K, (const bf16_t*)bias,
                       nullptr, 0, (bf16_t*)cmat, N, nullptr, M, N, K, s, e->prefill_ws());
    }
    ZE_KCHECK();
    return ZE_OK;
}

extern "C" int ze_op_token_logprob(ze_engine* e, const void* logits, int rows, int vocab, int ld, const int32_t* targets,
                                   float* out, void* stream) {
    if (!logits || !targets || !out || rows < 0 || vocab <= 0 || ld < vocab || ld % 8)
        return ze_fail(e, ZE_ERR_INVALID, "bad token_logprob arguments");
    hipSetDevice(e->device);
    ze_launch_token_logprob((const bf16_t*)logits, ld, vocab, targets, out, rows, (hipStream_t)stream);
    ZE_KCHECK();
    return ZE_OK;
}

extern "C" int ze_op_rmsnorm(ze_engine* e, const void* x, const void* weight, void* y, int rows, int cols, float eps,
                             void* stream) {
    if (!e || !x || !weight || !y) return ze_fail(e, ZE_ERR_INVALID, "null argument");
    if (cols % 8) return ze_fail(e, ZE_ERR_INVALID, "cols must be a multiple of 8");
    hipSetDevice(e->device);
    ze_launch_rmsnorm((const bf16_t*)x, cols, (const bf16_t*)weight, (bf16_t*)y, cols, rows, cols, eps,
                      (hipStream_t)stream);
    ZE_KCHECK();
    return ZE_OK;
}

extern "C" int ze_op_attention(ze_engine* e, const void* q, const void* k, const void* v, void* o, int T, int heads,
                               int kv_heads, int D, const int32_t* cu, int n_seg, int causal, void* stream) {
    if (!e || !q || !k || !v || !o || !cu) return ze_fail(e, ZE_ERR_INVALID, "null argument");
    if (D != 80 && D != 128) return ze_fail(e, ZE_ERR_INVALID, "D must be 80 or 128");
    if (heads % kv_heads) return ze_fail(e, ZE_ERR_INVALID, "kv_heads must divide heads");
    hipSetDevice(e->device);
    hipStream_t s = (hipStream_t)stream;
    std::vector<int> tiles((size_t)4 * (T / 64 + n_seg + 2));
    const int bq = ze_gemv_knobs[1] == 9 ? 64 : ZE_FA_BQ_LONG;  // (two query tiles per wave unless knob 1 = 9: the unit op covers both forms)
    const int nt = build_tiles(cu, n_seg, tiles.data(), (int)tiles.size() / 4, bq);
    if (nt < 0) return ze_fail(e, ZE_ERR_NOMEM, "tile overflow");
    int4* dt = nullptr;
    ZE_HIP(hipMalloc((void**)&dt, (size_t)std::max(nt, 1) * 16));
    ZE_HIP(hipMemcpyAsync(dt, tiles.data(), (size_t)nt * 16, hipMemcpyHostToDevice, s));
    // causal inside each segment: key j visible to query i iff j <= i (absolute row indices, offset 0)
    ze_launch_flash_attn(D, causal, (const bf16_t*)q, heads * D, D, (const bf16_t*)k, kv_heads * D, D,
                         (const bf16_t*)v, kv_heads * D, D, (bf16_t*)o, heads * D, D, dt, nt, heads, heads / kv_heads,
                         1.0f / sqrtf((float)D), 0, s, nullptr, 0, bq);
    hipError_t le = hipGetLastError();
    hipStreamSynchronize(s);
    hipFree(dt);
    if (le != hipSuccess) return ze_fail(e, ZE_ERR_HIP, hipGetErrorString(le));
    return ZE_OK;
}

// ---- K4 / K5 + K8 / K13 / K15 + K18 on their own (one entry per kernel), the launchers ze_vit_forward / ze_prefill use
// K4: rows of pixel_values gathered into WINDOW order (+ the cast to bf16 the patch embed reads), and merged rows scattered back
extern "C" int ze_op_window_gather(ze_engine* e, const float* pixel_values, const int32_t* grid_thw, int n_images, void* out_bf16,
                                   void* stream) {
    if (!e || !pixel_values || !grid_thw || !out_bf16 || n_images <= 0) return ze_fail(e, ZE_ERR_INVALID, "null argument");
    const ze_config& c = e->cfg;
    hipStream_t s = (hipStream_t)stream;
    hipSetDevice(e->device);
    const int mu = c.spatial_merge_size * c.spatial_merge_size;
    const int pk = c.in_channels * c.temporal_patch_size * c.patch_size * c.patch_size;
    std::vector<int64_t> widx;
    std::vector<int32_t> cu_win;
    ze_window_index_impl(grid_thw, n_images, c.spatial_merge_size, c.window_size, c.patch_size, widx, cu_win);
    const int n = (int)widx.size() * mu;
    if (n > c.max_patches) return ze_fail(e, ZE_ERR_NOMEM, "too many patches for max_patches");
    std::vector<int> perm(n);
    for (int j = 0; j < n / mu; ++j)
        for (int u = 0; u < mu; ++u) perm[j * mu + u] = (int)widx[j] * mu + u;
    ZE_HIP(hipMemcpyAsync(e->vperm, perm.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, s));
    ZE_HIP(hipStreamSynchronize(s));  // (perm is a local)
    ze_launch_gather_cast_rows(pixel_values, pk, e->vperm, (bf16_t*)out_bf16, pk, n, s);
    ZE_KCHECK();
    return ZE_OK;
}
extern "C" int ze_op_window_scatter(ze_engine* e, const void* x_bf16, int cols, const int32_t* grid_thw, int n_images, void* out_bf16,
                                    void* stream) {
    if (!e || !x_bf16 || !grid_thw || !out_bf16 || n_images <= 0 || cols <= 0 || cols % 8) return ze_fail(e, ZE_ERR_INVALID, "bad argument (cols % 8)");
    const ze_config& c = e->cfg;
    hipStream_t s = (hipStream_t)stream;
    hipSetDevice(e->device);
    std::vector<int64_t> widx;
    std::vector<int32_t> cu_win;
    ze_window_index_impl(grid_thw, n_images, c.spatial_merge_size, c.window_size, c.patch_size, widx, cu_win);
    const int m = (int)widx.size();
    if (m > c.max_patches) return ze_fail(e, ZE_ERR_NOMEM, "too many rows for max_patches");
    std::vector<int> inv(m);
    for (int j = 0; j < m; ++j) inv[j] = (int)widx[j];
    ZE_HIP(hipMemcpyAsync(e->vinv, inv.data(), (size_t)m * sizeof(int), hipMemcpyHostToDevice, s));
    ZE_HIP(hipStreamSynchronize(s));
    ze_launch_scatter_rows((const bf16_t*)x_bf16, cols, e->vinv, (bf16_t*)out_bf16, cols, m, cols, s);
    ZE_KCHECK();
    return ZE_OK;
}
// K5 + K8: the 2-D rotary tables of the grids (fp32, HF:...:125-134,441-446) applied to the q and k thirds of qkv [n, 3 * heads * D]
// in place, fp32 arithmetic (HF:...:160-171); rows in WINDOW order when window_order != 0 (as inside the ViT), else HF order
extern "C" int ze_op_vision_rope(ze_engine* e, void* qkv_bf16, const int32_t* grid_thw, int n_images, int window_order, void* stream) {
    if (!e || !qkv_bf16 || !grid_thw || n_images <= 0) return ze_fail(e, ZE_ERR_INVALID, "null argument");
    const ze_config& c = e->cfg;
    hipStream_t s = (hipStream_t)stream;
    hipSetDevice(e->device);
    const int mu = c.spatial_merge_size * c.spatial_merge_size;
    const int hd = e->vit_head_dim, half = hd / 2, nf = half / 2;
    std::vector<int64_t> widx;
    std::vector<int32_t> cu_win, hw;
    ze_window_index_impl(grid_thw, n_images, c.spatial_merge_size, c.window_size, c.patch_size, widx, cu_win);
    ze_vision_pos_ids_impl(grid_thw, n_images, c.spatial_merge_size, hw);
    const int n = (int)widx.size() * mu;
    if (n > c.max_patches) return ze_fail(e, ZE_ERR_NOMEM, "too many patches for max_patches");
    std::vector<float> invf(nf), hc((size_t)n * half), hs((size_t)n * half);
    for (int i = 0; i < nf; ++i) invf[i] = 1.0f / powf(10000.0f, (float)(2 * i) / (float)half);
    for (int r = 0; r < n; ++r) {
        const int old = window_order ? (int)widx[r / mu] * mu + r % mu : r;
        for (int i = 0; i < nf; ++i) {
            const float fh = (float)hw[2 * old] * invf[i], fw = (float)hw[2 * old + 1] * invf[i];
            hc[(size_t)r * half + i] = cosf(fh);
            hs[(size_t)r * half + i] = sinf(fh);
            hc[(size_t)r * half + nf + i] = cosf(fw);
            hs[(size_t)r * half + nf + i] = sinf(fw);
        }
    }
    ZE_HIP(hipMemcpyAsync(e->vcos, hc.data(), hc.size() * sizeof(float), hipMemcpyHostToDevice, s));
    ZE_HIP(hipMemcpyAsync(e->vsin, hs.data(), hs.size() * sizeof(float), hipMemcpyHostToDevice, s));
    ZE_HIP(hipStreamSynchronize(s));
    ze_launch_vision_rope((bf16_t*)qkv_bf16, e->vcos, e->vsin, n, c.vit_heads, hd, s);
    ZE_KCHECK();
    return ZE_OK;
}
// K13: out[t] = embed_tokens[ids[t]], rows whose id is the image token overwritten by the feature rows in order (HF:...:1206-1215)
extern "C" int ze_op_embed_scatter(ze_engine* e, const int32_t* input_ids, int len, const void* image_embeds, int n_image_rows,
                                   void* out_bf16, void* stream) {
    if (!e || !input_ids || !out_bf16 || len <= 0) return ze_fail(e, ZE_ERR_INVALID, "null argument");
    const ze_config& c = e->cfg;
    if (len > e->prefill_rows) return ze_fail(e, ZE_ERR_NOMEM, "more rows than the prefill workspace holds");
    hipStream_t s = (hipStream_t)stream;
    hipSetDevice(e->device);
    std::vector<int> src(len);
    int img = 0;
    for (int t = 0; t < len; ++t) {
        if (input_ids[t] < 0 || input_ids[t] >= c.vocab) return ze_fail(e, ZE_ERR_INVALID, "token id out of range");
        src[t] = input_ids[t] == c.image_token_id ? -1 - img++ : input_ids[t];
    }
    if (img != n_image_rows || (img > 0 && !image_embeds))
        return ze_fail(e, ZE_ERR_MISMATCH, "Image features and image tokens do not match, tokens: " + std::to_string(img) +
                                               ", features: " + std::to_string(n_image_rows));
    ZE_HIP(hipMemcpyAsync(e->tsrc, src.data(), (size_t)len * sizeof(int), hipMemcpyHostToDevice, s));
    ZE_HIP(hipStreamSynchronize(s));
    ze_launch_embed_rows(e->tsrc, e->embed, (const bf16_t*)image_embeds, (bf16_t*)out_bf16, len, c.hidden, s);
    ZE_KCHECK();
    return ZE_OK;
}
// K15 + K18: M-RoPE on the q / k parts of qkv [T, (heads + 2 kv_heads) * 128] (q in place; cos / sin rounded to bf16, products
// and sum rounded as HF's bf16 tensors round them, HF:...:557-599) and the KV append: the roped k rows and the v rows go
// into `layer`'s cache of chain `seq` at positions past .. past + T - 1 (DynamicCache.update, HF:...:667-668).  The chain's
// length is NOT changed (a unit op: ze_op_kv_read reads rows back whatever the chain's length says).
extern "C" int ze_op_mrope_kv(ze_engine* e, int seq, int layer, void* qkv_bf16, int T, const int32_t* position_ids, int past, void* stream) {
    ZE_TRY(check_seq(e, seq));
    const ze_config& c = e->cfg;
    if (!qkv_bf16 || !position_ids || T <= 0 || layer < 0 || layer >= c.layers || past < 0 || past + T > c.max_ctx || T > e->prefill_rows)
        return ze_fail(e, ZE_ERR_INVALID, "bad mrope_kv arguments");
    for (int i = 0; i < 3 * T; ++i)
        if (position_ids[i] < 0 || position_ids[i] >= e->max_pos) return ze_fail(e, ZE_ERR_INVALID, "position id out of range");
    hipStream_t s = (hipStream_t)stream;
    hipSetDevice(e->device);
    ZE_HIP(hipMemcpyAsync(e->tpos, position_ids, (size_t)3 * T * sizeof(int), hipMemcpyHostToDevice, s));
    ZE_HIP(hipStreamSynchronize(s));
    ze_launch_mrope_kv((bf16_t*)qkv_bf16, T, c.heads, c.kv_heads, e->head_dim, e->cosT, e->sinT, e->tpos, e->axis_of, e->kc(layer, seq),
                       e->vc(layer, seq), c.max_ctx, past, nullptr, 0, s);
    ZE_KCHECK();
    return ZE_OK;
}
// the decode-step form of the same (k_rope_kv_batch: row b = chain seqs[b] at ITS position ctx + rope_delta, K/V appended at ctx)
extern "C" int ze_op_rope_kv_decode(ze_engine* e, const int32_t* seqs, int n, int layer, void* qkv_bf16, void* stream) {
    if (!e || !seqs || !qkv_bf16 || n <= 0 || n > e->cfg.max_seqs || layer < 0 || layer >= e->cfg.layers)
        return ze_fail(e, ZE_ERR_INVALID, "bad rope_kv_decode arguments");
    const ze_config& c = e->cfg;
    for (int i = 0; i < n; ++i) {
        ZE_TRY(check_seq(e, seqs[i]));
        if (e->ctx_host[seqs[i]] + 1 > c.max_ctx) return ze_fail(e, ZE_ERR_NOMEM, "sequence exceeds max_ctx");
    }
    hipStream_t s = (hipStream_t)stream;
    hipSetDevice(e->device);
    ze_launch_set_ints(e->bseq, seqs, n, s);
    ze_launch_rope_kv_batch((bf16_t*)qkv_bf16, n, c.heads, c.kv_heads, e->head_dim, e->cosT, e->sinT, e->st_dev, e->bseq, e->kc(layer, 0),
                            e->vc(layer, 0), (size_t)c.kv_heads * c.max_ctx * e->head_dim, c.max_ctx, s);
    ZE_KCHECK();
    return ZE_OK;
}
// The attention of ONE batched decode step, alone: row b of qkv_bf16 [n, (heads + 2 kv_heads) x 128] is chain seqs[b]'s
// projection AFTER ze_op_rope_kv_decode (which rotated q / k and appended k, v at row ctx of the cache); the kernel of the step
// (k_attn_decode_wave_long in the row-streaming family and from 17 chains on, the fragment family's choice below) attends the
// q heads over rows 0 .. ctx of `layer` and writes out_bf16 [n, heads x 128].  Chain state is not advanced.
extern "C" int ze_op_attn_decode(ze_engine* e, const int32_t* seqs, int n, int layer, const void* qkv_bf16, void* out_bf16, void* stream) {
    if (!e || !seqs || !qkv_bf16 || !out_bf16 || n <= 0 || n > e->cfg.max_seqs || layer < 0 || layer >= e->cfg.layers)
        return ze_fail(e, ZE_ERR_INVALID, "bad attn_decode arguments");
    for (int i = 0; i < n; ++i) {
        ZE_TRY(check_seq(e, seqs[i]));
        if (e->ctx_host[seqs[i]] + 1 > e->cfg.max_ctx) return ze_fail(e, ZE_ERR_NOMEM, "sequence exceeds max_ctx");
    }
    hipStream_t s = (hipStream_t)stream;
    hipSetDevice(e->device);
    ze_launch_set_ints(e->bseq, seqs, n, s);
    sync_prefix(e, seqs, n, s);
    set_live_parts(e, seqs, n, 1);
    launch_batch_attention(e, layer, n, false, s, (const bf16_t*)qkv_bf16, (bf16_t*)out_bf16);
    ZE_KCHECK();
    return ZE_OK;
}
// rows [start, start + n) of `layer`'s K and V cache of chain `seq`, every kv head: out_k / out_v bf16 [kv_heads, n, 128] (device)
extern "C" int ze_op_kv_read(ze_engine* e, int seq, int layer, int start, int n, void* out_k, void* out_v, void* stream) {
    ZE_TRY(check_seq(e, seq));
    const ze_config& c = e->cfg;
    if (!out_k || !out_v || layer < 0 || layer >= c.layers || start < 0 || n <= 0 || start + n > c.max_ctx)
        return ze_fail(e, ZE_ERR_INVALID, "bad kv_read arguments");
    hipStream_t s = (hipStream_t)stream;
    hipSetDevice(e->device);
    const size_t row = (size_t)e->head_dim * sizeof(bf16_t);
    for (int h = 0; h < c.kv_heads; ++h) {
        ZE_HIP(hipMemcpyAsync((char*)out_k + (size_t)h * n * row, e->kc(layer, seq) + ((size_t)h * c.max_ctx + start) * e->head_dim,
                              (size_t)n * row, hipMemcpyDeviceToDevice, s));
        ZE_HIP(hipMemcpyAsync((char*)out_v + (size_t)h * n * row, e->vc(layer, seq) + ((size_t)h * c.max_ctx + start) * e->head_dim,
                              (size_t)n * row, hipMemcpyDeviceToDevice, s));
    }
    return ZE_OK;
}

// ================================================================== fp8 decode weights
static int quantize_linear(ze_engine* e, ze_linear& l, int rows, int cols, uint8_t*& cur8, float*& curs, hipStream_t s) {
    l.ld8 = (cols + 15) / 16 * 16;
    l.w8 = cur8;
    l.scale8 = curs;
    cur8 += (size_t)rows * l.ld8;
    curs += rows;
    ze_launch_quantize_rows(l.w, rows, cols, l.ld, l.w8, l.ld8, l.scale8, s);
    return ZE_OK;
}

extern "C" int ze_weights_quantize_fp8(ze_engine* e, void* stream) {
    if (!e) return ze_fail(e, ZE_ERR_INVALID, "null engine");
    if (e->fp8_ready) return ZE_OK;  // (a weight change clears the flag: ze_weights_changed)
    const ze_config& c = e->cfg;
    hipStream_t s = (hipStream_t)stream;
    hipSetDevice(e->device);
    const int H = c.hidden, hd = e->head_dim, nq = c.heads * hd, nqkv = nq + 2 * c.kv_heads * hd, ip = e->text_ipad;
    auto pad16 = [](int x) { return (size_t)((x + 15) / 16 * 16); };
    size_t bytes = 0, rows = 0;
    for (int li = 0; li < c.layers; ++li) {
        bytes += (size_t)nqkv * pad16(H) + (size_t)H * pad16(nq) + (size_t)2 * ip * pad16(H) + (size_t)H * pad16(ip);
        rows += (size_t)nqkv + H + 2 * ip + H;
    }
    const bool head = !c.tie_word_embeddings;  // a tied lm_head is the embedding table: it stays bf16
    if (head) {
        bytes += (size_t)c.vocab * pad16(H);
        rows += c.vocab;
    }
    if (!e->arena8) ZE_HIP(hipMalloc((void**)&e->arena8, bytes + rows * sizeof(float) + 256));  // re-quantisation reuses it
    ZE_HIP(hipMemsetAsync(e->arena8, 0, bytes + rows * sizeof(float) + 256, s));
    uint8_t* cur8 = e->arena8;
    float* curs = reinterpret_cast<float*>(e->arena8 + (bytes + 255) / 256 * 256);
    for (int li = 0; li < c.layers; ++li) {
        ze_text_layer& L = e->tl[li];
        quantize_linear(e, L.qkv, nqkv, H, cur8, curs, s);
        quantize_linear(e, L.o, H, nq, cur8, curs, s);
        quantize_linear(e, L.gate_up, 2 * ip, H, cur8, curs, s);
        quantize_linear(e, L.down, H, ip, cur8, curs, s);
    }
    if (head) {
        e->lm_head8.w = e->lm_head;
        e->lm_head8.ld = H;
        quantize_linear(e, e->lm_head8, c.vocab, H, cur8, curs, s);
    }
    ZE_KCHECK();
    ZE_HIP(hipStreamSynchronize(s));
    e->fp8_ready = true;
    e->frag_ready = false;  // the bf16 copies were replaced by the dequantised values
    ++ze_tune_epoch;  // captured decode steps hold the bf16 streams
    return ZE_OK;
}

extern "C" int ze_set_fp8_activations(ze_engine* e, int on) {
    if (!e) return ze_fail(e, ZE_ERR_INVALID, "null engine");
    if (on && !e->fp8_ready)
        return ze_fail(e, ZE_ERR_INVALID, "fp8 activations go with fp8 weights: call ze_weights_quantize_fp8 first");
    if (e->fp8_act != (on != 0)) {
        e->fp8_act = on != 0;
        ++ze_tune_epoch;  // captured decode steps bake the choice of kernels in
    }
    return ZE_OK;
}

extern "C" int ze_op_linear_mx(ze_engine* e, const void* a8, const void* sa, const void* w8, const void* sw, const void* bias,
                               void* cmat, int M, int N, int K, int swiglu, void* stream) {
    if (!e || !a8 || !sa || !w8 || !sw || !cmat || M <= 0 || N <= 0 || K <= 0)
        return ze_fail(e, ZE_ERR_INVALID, "null pointer or empty shape");
    if (K % 128 != 0 || K < 256 || (swiglu && N % 32 != 0))
        return ze_fail(e, ZE_ERR_INVALID, "K must be a multiple of 128 (>= 256); with swiglu N a multiple of 32");
    hipSetDevice(e->device);
    if (!ze_launch_gemm_mx(swiglu ? ZE_EPI_SWIGLU : ZE_EPI_NONE, (const uint8_t*)a8, K, (const float*)sa, (const uint8_t*)w8, K,
                           (const float*)sw, (const bf16_t*)bias, (bf16_t*)cmat, swiglu ? N / 2 : N, M, N, K, (hipStream_t)stream))
        return ze_fail(e, ZE_ERR_INVALID, "shape not supported by the block-scaled kernel");
    ZE_KCHECK();
    return ZE_OK;
}

extern "C" int ze_op_quantize_fp8(ze_engine* e, void* w_bf16, int rows, int cols, void* q_out, void* scale_out,
                                  void* stream) {
    if (!e || !w_bf16 || !q_out || !scale_out || rows <= 0 || cols <= 0 || cols % 16)
        return ze_fail(e, ZE_ERR_INVALID, "bad argument (cols must be a multiple of 16)");
    hipSetDevice(e->device);
    ze_launch_quantize_rows((bf16_t*)w_bf16, rows, cols, cols, (uint8_t*)q_out, cols, (float*)scale_out, (hipStream_t)stream);
    ZE_KCHECK();
    return ZE_OK;
}

extern "C" int ze_set_decode_regime(ze_engine* e, int regime) {
    if (!e || regime < -1 || regime > 1) return ze_fail(e, ZE_ERR_INVALID, "regime is -1 (by capacity), 0 (fragment kernels) or 1 (row streaming)");
    if (regime != e->decode_regime) {
        const bool was = e->wide_regime();
        e->decode_regime = regime;
        if (was != e->wide_regime()) e->frag_ready = false;  // the other family's weight copies are built on its first step
        ++ze_tune_epoch;  // captured batched steps bake the kernel family in
    }
    return e->wide_regime() ? 1 : 0;
}

// ================================================================== measurement
extern "C" int ze_tune(int knob, int value) {
    if (knob < 0 || knob >= 24) return ze_fail(nullptr, ZE_ERR_INVALID, "unknown knob");
    ze_gemv_knobs[knob] = value;
    ++ze_tune_epoch;  // captured decode steps bake the launch policy in: engines drop their graphs on the next use
    return ZE_OK;
}
extern "C" int ze_profile_decode_kernel(ze_engine* e, int which, int iters, float* avg_us, double* bytes_per_launch,
                                        void* stream) {
    if (!e || !avg_us || !bytes_per_launch || iters <= 0) return ze_fail(e, ZE_ERR_INVALID, "bad argument");
    const ze_config& c = e->cfg;
    hipStream_t s = (hipStream_t)stream;
    hipSetDevice(e->device);
    const int H = c.hidden, hd = e->head_dim, nq = c.heads * hd, nqkv = nq + 2 * c.kv_heads * hd;
    hipEvent_t a, b;
    ZE_HIP(hipEventCreate(&a));
    ZE_HIP(hipEventCreate(&b));
    ZE_HIP(hipMemsetAsync(e->dh, 0, (size_t)H * 2, s));
    ZE_HIP(hipMemsetAsync(e->dattn, 0, (size_t)nq * 2, s));
    ZE_HIP(hipMemsetAsync(e->dact, 0, (size_t)e->text_ipad * 2, s));
    double bytes = 0;
    auto launch = [&](int it) {
        const ze_text_layer& L = e->tl[it % c.layers];
        ze_gemv_args g;
        memset(&g, 0, sizeof(g));
        g.D = hd;
        // a quantised engine streams the FP8 copy (1 byte per weight + one fp32 scale per row), as its decode step does
        auto fp8 = [&](const ze_linear& l, double rows, double cols) {
            if (e->fp8_ready && l.w8) {
                g.W8 = l.w8;
                g.scale8 = l.scale8;
                g.ldw8 = l.ld8;
                return rows * cols + rows * 4.0;
            }
            return rows * cols * 2.0;
        };
        switch (which) {
            case 0:
                g.W = L.qkv.w; g.ldw = L.qkv.ld; g.N = nqkv; g.K = H; g.x = e->dh; g.norm_w = L.in_norm;
                g.eps = c.rms_eps; g.bias = L.qkv.bias; g.out_bf16 = e->dq; g.st = e->st_dev; g.cosT = e->cosT;
                g.sinT = e->sinT; g.kcache = e->kc(it % c.layers, 0); g.vcache = e->vc(it % c.layers, 0);
                g.heads = c.heads; g.kv_heads = c.kv_heads; g.max_ctx = c.max_ctx;
                bytes = fp8(L.qkv, nqkv, H);
                ze_launch_gemv(ZE_GV_QKV_ROPE, g, s);
                break;
            case 1:
                g.W = L.o.w; g.ldw = L.o.ld; g.N = H; g.K = nq; g.x = e->dattn; g.out_bf16 = e->dh;
                bytes = fp8(L.o, H, nq);
                ze_launch_gemv(ZE_GV_RESIDUAL, g, s);
                break;
            case 2:
                g.W = L.gate_up.w; g.ldw = L.gate_up.ld; g.N = 2 * e->text_ipad; g.K = H; g.x = e->dh;
                g.norm_w = L.post_norm; g.eps = c.rms_eps; g.out_bf16 = e->dact;
                bytes = fp8(L.gate_up, 2.0 * c.intermediate, H);
                ze_launch_gemv(ZE_GV_SWIGLU, g, s);
                break;
            case 3:
                g.W = L.down.w; g.ldw = L.down.ld; g.N = H; g.K = e->text_ipad; g.x = e->dact; g.out_bf16 = e->dh;
                bytes = fp8(L.down, H, c.intermediate);
                ze_launch_gemv(ZE_GV_RESIDUAL, g, s);
                break;
            default:
                g.W = e->lm_head; g.ldw = H; g.N = c.vocab; g.K = H; g.x = e->dh; g.norm_w = e->final_norm;
                g.eps = c.rms_eps; g.out_f32 = e->dlogits;
                bytes = fp8(e->lm_head8, c.vocab, H);
                ze_launch_gemv(ZE_GV_LOGITS, g, s);
                break;
        }
    };
    for (int i = 0; i < std::min(iters, 4); ++i) launch(i);  // warm-up
    ZE_HIP(hipEventRecord(a, s));
    for (int i = 0; i < iters; ++i) launch(i);
    ZE_HIP(hipEventRecord(b, s));
    ZE_HIP(hipEventSynchronize(b));
    float ms = 0.f;
    ZE_HIP(hipEventElapsedTime(&ms, a, b));
    hipEventDestroy(a);
    hipEventDestroy(b);
    ZE_KCHECK();
    *avg_us = ms * 1000.0f / (float)iters;
    *bytes_per_launch = bytes;
    return ZE_OK;
}

// The kernels of the BATCHED decode step (ze_decode_batch / ze_decode_burst) at n chains (slots 0..n-1, with whatever
// context they hold), one kind per call, cycling through the layers' real weights and KV caches, bracketed by HIP events
// on `stream`.  which: 0 qkv, 1 o_proj, 2 gate_up (SwiGLU), 3 down, 4 lm_head, 5 decode attention, 6 RMSNorm,
// 7 rope + KV append.  bytes_per_launch = algorithmic bytes: the weight matrix (0-4), the K/V rows of the n chains (5),
// the activation rows read + written (6, 7).
extern "C" int ze_profile_batch_kernel(ze_engine* e, int which, int n, int iters, float* avg_us, double* bytes_per_launch,
                                       void* stream) {
    if (!e || !avg_us || !bytes_per_launch || iters <= 0 || n <= 0 || n > e->cfg.max_seqs || (n > 64 && !e->wide_regime()))
        return ze_fail(e, ZE_ERR_INVALID, "bad argument");
    const ze_config& c = e->cfg;
    hipStream_t s = (hipStream_t)stream;
    hipSetDevice(e->device);
    ZE_TRY(ensure_fragments(e, s));
    const int H = c.hidden, hd = e->head_dim, nq = c.heads * hd, nkv = c.kv_heads * hd, nqkv = nq + 2 * nkv;
    const size_t seq_stride = (size_t)c.kv_heads * c.max_ctx * hd;
    std::vector<int> seqs(n);
    double kv_bytes = 0;
    for (int i = 0; i < n; ++i) {
        seqs[i] = i;
        kv_bytes += (double)(std::min(e->ctx_host[i] + 1, c.max_ctx)) * nkv * 2 * 2;
    }
    ze_launch_set_ints(e->bseq, seqs.data(), n, s);
    sync_prefix(e, seqs.data(), n, s);
    upload_mates(e, seqs.data(), n, s);
    set_live_parts(e, seqs.data(), n, 1);  // (the grid a decode step of these chains would launch)
    double bytes = 0;
    auto launch = [&](int it) {
        const int li = it % c.layers;
        const ze_text_layer& L = e->tl[li];
        const bool fr = L.qkv.wf && !e->wide_regime() && ze_gemv_knobs[5] != 1;  // (as enqueue_decode_batch)
        const bool tiled = e->wide_regime() && ze_gemv_knobs[13] != 1;
        const ze_gemm_ws ws = e->gemm_ws();
        switch (which) {
            case 0:
                if (fr) ze_launch_qkv_rope_oneshot(e->by, L.qkv.wf, L.qkv.bias, e->bqkv, nqkv, n, H, c.heads, c.kv_heads, e->cosT, e->sinT,
                                                   e->st_dev, e->bseq, e->kc(li, 0), e->vc(li, 0), seq_stride, c.max_ctx, s);
                else if (tiled && L.qkv.wp && ze_gemv_knobs[13] != 2)
                    ze_launch_gemm_qkv_rope(e->by, H, L.qkv.wp, H, L.qkv.bias_p, e->qkv_epi_dev + li, e->bqkv, nqkv, n, nqkv, H, s);
                else if (tiled) ze_launch_gemm_wide(ZE_EPI_NONE, e->by, H, L.qkv.w, L.qkv.ld, L.qkv.bias, nullptr, 0, e->bqkv, nqkv, n, nqkv, H, ws, s);
                else ze_launch_gemm_stream(ZE_EPI_NONE, e->by, H, L.qkv.w, L.qkv.ld, L.qkv.bias, nullptr, 0, e->bqkv, nqkv, n, nqkv, H, ws, s);
                bytes = (double)nqkv * H * 2;
                break;
            case 1:
                if (fr && ze_gemv_knobs[9] != 1) ze_launch_gemm_oneshot(ZE_EPI_RESIDUAL, e->bo, L.o.wf, nullptr, e->bh, H, e->bh, H, n, H, nq, s);
                else if (fr) ze_launch_gemm_frag(ZE_EPI_RESIDUAL, e->bo, L.o.wf, nullptr, e->bh, H, e->bh, H, n, H, nq, s);
                else if (tiled) ze_launch_gemm_wide(ZE_EPI_RESIDUAL, e->bo, nq, L.o.w, L.o.ld, nullptr, e->bh, H, e->bh, H, n, H, nq, ws, s);
                else ze_launch_gemm_stream(ZE_EPI_RESIDUAL, e->bo, nq, L.o.w, L.o.ld, nullptr, e->bh, H, e->bh, H, n, H, nq, ws, s);
                bytes = (double)H * nq * 2;
                break;
            case 2:
                if (fr) ze_launch_gemm_frag(ZE_EPI_SWIGLU, e->by, L.gate_up.wf, nullptr, nullptr, 0, e->ba, e->text_ipad, n, 2 * e->text_ipad, H, s);
                else if (tiled) ze_launch_gemm_wide(ZE_EPI_SWIGLU, e->by, H, L.gate_up.w, L.gate_up.ld, nullptr, nullptr, 0, e->ba, e->text_ipad, n, 2 * e->text_ipad, H, ws, s);
                else ze_launch_gemm_stream(ZE_EPI_SWIGLU, e->by, H, L.gate_up.w, L.gate_up.ld, nullptr, nullptr, 0, e->ba, e->text_ipad, n, 2 * e->text_ipad, H, ws, s);
                bytes = 2.0 * c.intermediate * H * 2;
                break;
            case 3:
                if (tiled) ze_launch_gemm_wide(ZE_EPI_RESIDUAL, e->ba, e->text_ipad, L.down.w, L.down.ld, nullptr, e->bh, H, e->bh, H, n, H, e->text_ipad, ws, s);
                else ze_launch_gemm_stream(ZE_EPI_RESIDUAL, e->ba, e->text_ipad, L.down.w, L.down.ld, nullptr, e->bh, H, e->bh, H, n, H, e->text_ipad, ws, s);
                bytes = (double)H * c.intermediate * 2;
                break;
            case 4:
                if (e->lm_head_f && fr) ze_launch_gemm_frag(ZE_EPI_F32, e->by, e->lm_head_f, nullptr, nullptr, 0, (bf16_t*)e->blogits, c.vocab, n, c.vocab, H, s);
                else if (tiled) ze_launch_gemm_wide(ZE_EPI_F32, e->by, H, e->lm_head, H, nullptr, nullptr, 0, (bf16_t*)e->blogits, c.vocab, n, c.vocab, H, ws, s);
                else ze_launch_gemm_stream(ZE_EPI_F32, e->by, H, e->lm_head, H, nullptr, nullptr, 0, (bf16_t*)e->blogits, c.vocab, n, c.vocab, H, ws, s);
                bytes = (double)c.vocab * H * 2;
                break;
            case 5:
                launch_batch_attention(e, li, n, fr, s);
                bytes = kv_bytes;
                break;
            case 6:
                ze_launch_rmsnorm(e->bh, H, L.in_norm, e->by, H, n, H, c.rms_eps, s, fr ? 1 : 2);
                bytes = (double)n * H * 2 * 2;
                break;
            default:
                ze_launch_rope_kv_batch(e->bqkv, n, c.heads, c.kv_heads, hd, e->cosT, e->sinT, e->st_dev, e->bseq, e->kc(li, 0),
                                        e->vc(li, 0), seq_stride, c.max_ctx, s);
                bytes = (double)n * nqkv * 2 * 2;
                break;
        }
    };
    hipEvent_t a, b;
    ZE_HIP(hipEventCreate(&a));
    ZE_HIP(hipEventCreate(&b));
    for (int i = 0; i < std::min(iters, 4); ++i) launch(i);  // warm-up
    ZE_HIP(hipEventRecord(a, s));
    for (int i = 0; i < iters; ++i) launch(i);
    ZE_HIP(hipEventRecord(b, s));
    ZE_HIP(hipEventSynchronize(b));
    float ms = 0.f;
    ZE_HIP(hipEventElapsedTime(&ms, a, b));
    hipEventDestroy(a);
    hipEventDestroy(b);
    ZE_KCHECK();
    *avg_us = ms * 1000.0f / (float)iters;
    *bytes_per_launch = bytes;
    return ZE_OK;
}

// The projections of a PREFILL pass, one kind per call, on the pass's own operands: the engine's layer weights in rotation and the
// activation rows the last ze_prefill_batch / ze_prefill left in the workspace (normalised hidden rows, attention output, SwiGLU
// output of its last layer -- rows beyond that pass hold older passes' rows or zeros), through the launcher the pass uses
// (ze_launch_gemm: k_gemm_p8 from ~1.5 K rows on).  which: 0 qkv, 1 o_proj (+ residual), 2 gate_up (SwiGLU), 3 down (+ residual).
// Outputs go to scratch rows (the residual stream is read, never written).  flops_per_launch = 2 x rows x N x K of the real shape.
extern "C" int ze_profile_prefill_kernel(ze_engine* e, int which, int rows, int iters, float* avg_us, double* flops_per_launch,
                                         void* stream) {
    if (!e || !avg_us || !flops_per_launch || iters <= 0 || rows <= 0 || rows > e->prefill_rows || which < 0 || which > 3)
        return ze_fail(e, ZE_ERR_INVALID, "bad argument");
    const ze_config& c = e->cfg;
    hipStream_t s = (hipStream_t)stream;
    hipSetDevice(e->device);
    const int H = c.hidden, hd = e->head_dim, nq = c.heads * hd, nqkv = nq + 2 * c.kv_heads * hd, ip = e->text_ipad;
    if (nqkv < H) return ze_fail(e, ZE_ERR_INVALID, "scratch rows too short for this shape");
    double flops = 0;
    auto launch = [&](int it) {
        const ze_text_layer& L = e->tl[it % c.layers];
        switch (which) {
            case 0:
                ze_launch_gemm(ZE_EPI_NONE, e->ty, H, L.qkv.w, L.qkv.ld, L.qkv.bias, nullptr, 0, e->tqkv, nqkv, nullptr, rows, nqkv, H, s, e->prefill_ws());
                flops = 2.0 * rows * nqkv * H;
                break;
            case 1:
                ze_launch_gemm(ZE_EPI_RESIDUAL, e->to, nq, L.o.w, L.o.ld, nullptr, e->th, H, e->tqkv, nqkv, nullptr, rows, H, nq, s, e->prefill_ws());
                flops = 2.0 * rows * H * nq;
                break;
            case 2:
                ze_launch_gemm(ZE_EPI_SWIGLU, e->ty, H, L.gate_up.w, L.gate_up.ld, nullptr, nullptr, 0, e->ta, ip, nullptr, rows, 2 * ip, H, s, e->prefill_ws());
                flops = 2.0 * rows * 2.0 * c.intermediate * H;
                break;
            default:
                ze_launch_gemm(ZE_EPI_RESIDUAL, e->ta, ip, L.down.w, L.down.ld, nullptr, e->th, H, e->tqkv, nqkv, nullptr, rows, H, ip, s, e->prefill_ws());
                flops = 2.0 * rows * H * c.intermediate;
                break;
        }
    };
    hipEvent_t a, b;
    ZE_HIP(hipEventCreate(&a));
    ZE_HIP(hipEventCreate(&b));
    for (int i = 0; i < std::min(iters, 3); ++i) launch(i);  // warm-up
    ZE_HIP(hipEventRecord(a, s));
    for (int i = 0; i < iters; ++i) launch(i);
    ZE_HIP(hipEventRecord(b, s));
    ZE_HIP(hipEventSynchronize(b));
    float ms = 0.f;
    ZE_HIP(hipEventElapsedTime(&ms, a, b));
    hipEventDestroy(a);
    hipEventDestroy(b);
    ZE_KCHECK();
    *avg_us = ms * 1000.0f / (float)iters;
    *flops_per_launch = flops;
    return ZE_OK;
}

// The four projections of a prefill layer in PASS ORDER (qkv, o, gate/up, down; layer after layer, as ze_prefill_batch issues them --
// minus the norm / rope / attention launches between them), every launch bracketed by its own pair of HIP events: per-projection
// averages under the clocks and cache state a pass gives them (twelve back-to-back launches of ONE projection run 5-9 % slower than
// the same kernel inside a pass: rocprofv3 of the replayed pass, profiles/r06_prefill_by_shape.csv).  Operands as
// ze_profile_prefill_kernel.  avg_us / flops: [0] qkv, [1] o, [2] gate/up, [3] down.
extern "C" int ze_profile_prefill_layer(ze_engine* e, int rows, int layers_run, float avg_us[4], double flops[4], void* stream) {
    if (!e || !avg_us || !flops || layers_run <= 0 || layers_run > 256 || rows <= 0 || rows > e->prefill_rows)
        return ze_fail(e, ZE_ERR_INVALID, "bad argument");
    const ze_config& c = e->cfg;
    hipStream_t s = (hipStream_t)stream;
    hipSetDevice(e->device);
    const int H = c.hidden, hd = e->head_dim, nq = c.heads * hd, nqkv = nq + 2 * c.kv_heads * hd, ip = e->text_ipad;
    if (nqkv < H) return ze_fail(e, ZE_ERR_INVALID, "scratch rows too short for this shape");
    auto launch = [&](int li, int which) {
        const ze_text_layer& L = e->tl[li % c.layers];
        switch (which) {
            case 0: ze_launch_gemm(ZE_EPI_NONE, e->ty, H, L.qkv.w, L.qkv.ld, L.qkv.bias, nullptr, 0, e->tqkv, nqkv, nullptr, rows, nqkv, H, s, e->prefill_ws()); break;
            case 1: ze_launch_gemm(ZE_EPI_RESIDUAL, e->to, nq, L.o.w, L.o.ld, nullptr, e->th, H, e->tqkv, nqkv, nullptr, rows, H, nq, s, e->prefill_ws()); break;
            case 2: ze_launch_gemm(ZE_EPI_SWIGLU, e->ty, H, L.gate_up.w, L.gate_up.ld, nullptr, nullptr, 0, e->ta, ip, nullptr, rows, 2 * ip, H, s, e->prefill_ws()); break;
            default: ze_launch_gemm(ZE_EPI_RESIDUAL, e->ta, ip, L.down.w, L.down.ld, nullptr, e->th, H, e->tqkv, nqkv, nullptr, rows, H, ip, s, e->prefill_ws()); break;
        }
    };
    flops[0] = 2.0 * rows * nqkv * H;
    flops[1] = 2.0 * rows * H * nq;
    flops[2] = 2.0 * rows * 2.0 * c.intermediate * H;
    flops[3] = 2.0 * rows * H * c.intermediate;
    std::vector<hipEvent_t> ev((size_t)layers_run * 8);
    for (auto& x : ev) ZE_HIP(hipEventCreate(&x));
    for (int li = 0; li < 2; ++li)   // warm-up
        for (int w = 0; w < 4; ++w) launch(li, w);
    for (int li = 0; li < layers_run; ++li)
        for (int w = 0; w < 4; ++w) {
            ZE_HIP(hipEventRecord(ev[(size_t)(li * 4 + w) * 2], s));
            launch(li, w);
            ZE_HIP(hipEventRecord(ev[(size_t)(li * 4 + w) * 2 + 1], s));
        }
    ZE_HIP(hipStreamSynchronize(s));
    double sum[4] = {0, 0, 0, 0};
    for (int li = 0; li < layers_run; ++li)
        for (int w = 0; w < 4; ++w) {
            float ms = 0.f;
            ZE_HIP(hipEventElapsedTime(&ms, ev[(size_t)(li * 4 + w) * 2], ev[(size_t)(li * 4 + w) * 2 + 1]));
            sum[w] += ms;
        }
    for (auto& x : ev) hipEventDestroy(x);
    ZE_KCHECK();
    for (int w = 0; w < 4; ++w) avg_us[w] = (float)(sum[w] * 1000.0 / layers_run);
    return ZE_OK;
}

// The numeric helpers every epilogue of the library shares (ze_common.h: f32_to_bf16 / pack_bf16x2 = v_cvt_pk_bf16_f32, silu_f =
// x * v_rcp_f32(1 + v_exp_f32(-x))), alone: out[i] = bf16(x[i]) | bf16(bf16(silu(x[i])) * y[i]) << 16, out2[i] = pack(x[i], y[i]).
extern "C" int ze_op_numeric_helpers(ze_engine* e, const float* x, const float* y, uint32_t* out, uint32_t* out2, int n, void* stream) {
    if (!e || !x || !y || !out || !out2 || n < 0) return ze_fail(e, ZE_ERR_INVALID, "null argument");
    hipSetDevice(e->device);
    ze_launch_numeric_helpers(x, y, out, out2, n, (hipStream_t)stream);
    ZE_KCHECK();
    return ZE_OK;
}
