// Host-side integer helpers of the path: Pillow bicubic coefficient tables, smart_resize, vision window
// index / segment lengths / rotary position ids, M-RoPE position ids.  Plain C++ (no HIP), exact integer /
// IEEE-double arithmetic mirroring the Python originals cited at each function.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "ze_host.h"

// Pillow src/libImaging/Resample.c: bicubic_filter (a = -0.5), precompute_coeffs, normalize_coeffs_8bpc.
// (third-party; restated in oracle/frontend.py and pinned by tests/golden/bicubic.npz)
static inline double bicubic_filter(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

void ze_bicubic_coeffs(int in_size, int out_size, ze_coeffs* c) {
    const double scale = (double)in_size / (double)out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 2.0 * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    c->ksize = ksize;
    c->out_size = out_size;
    c->xmin.assign(out_size, 0);
    c->xcnt.assign(out_size, 0);
    c->kk.assign((size_t)out_size * ksize, 0);
    const double ss = 1.0 / filterscale;
    std::vector<double> w(ksize);
    int max_cnt = 0;
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        int lo = (int)(center - support + 0.5);
        if (lo < 0) lo = 0;
        int hi = (int)(center + support + 0.5);
        if (hi > in_size) hi = in_size;
        const int n = hi - lo;
        double ww = 0.0;
        for (int x = 0; x < n; ++x) {
            w[x] = bicubic_filter((x + lo - center + 0.5) * ss);
            ww += w[x];
        }
        for (int x = 0; x < n; ++x) {
            if (ww != 0.0) w[x] /= ww;
            const double v = w[x];
            c->kk[(size_t)xx * ksize + x] = v < 0 ? (int)(-0.5 + v * (1 << 22)) : (int)(0.5 + v * (1 << 22));
        }
        c->xmin[xx] = lo;
        c->xcnt[xx] = n;
        max_cnt = std::max(max_cnt, n);
    }
    c->max_cnt = max_cnt;
}

// HF:models/qwen2_vl/image_processing_pil_qwen2_vl.py:57-83 (Python round() = round-half-even = nearbyint).
int ze_smart_resize_impl(int height, int width, int factor, int64_t min_pixels, int64_t max_pixels, int* out_h,
                         int* out_w) {
    if (height <= 0 || width <= 0) return -1;
    if ((double)std::max(height, width) / (double)std::min(height, width) > 200.0) return -1;
    long h_bar = (long)nearbyint((double)height / factor) * factor;
    long w_bar = (long)nearbyint((double)width / factor) * factor;
    if (h_bar * w_bar > max_pixels) {
        const double beta = sqrt(((double)height * (double)width) / (double)max_pixels);
        h_bar = std::max<long>(factor, (long)floor((double)height / beta / factor) * factor);
        w_bar = std::max<long>(factor, (long)floor((double)width / beta / factor) * factor);
    } else if (h_bar * w_bar < min_pixels) {
        const double beta = sqrt((double)min_pixels / ((double)height * (double)width));
        h_bar = (long)ceil((double)height * beta / factor) * factor;
        w_bar = (long)ceil((double)width * beta / factor) * factor;
    }
    *out_h = (int)h_bar;
    *out_w = (int)w_bar;
    return 0;
}

// HF:vision_utils.py:130-188 get_vision_window_index.
void ze_window_index_impl(const int32_t* grid_thw, int n_images, int merge, int window_size, int patch,
                          std::vector<int64_t>& window_index, std::vector<int32_t>& cu_window) {
    window_index.clear();
    cu_window.assign(1, 0);
    const int vws = window_size / merge / patch;
    const int unit = merge * merge;
    int64_t base = 0;
    for (int im = 0; im < n_images; ++im) {
        const int t = grid_thw[3 * im], h = grid_thw[3 * im + 1], w = grid_thw[3 * im + 2];
        const int lh = h / merge, lw = w / merge;
        const int pad_h = vws - lh % vws, pad_w = vws - lw % vws;
        const int nh = (lh + pad_h) / vws, nw = (lw + pad_w) / vws;
        for (int tt = 0; tt < t; ++tt)
            for (int wy = 0; wy < nh; ++wy)
                for (int wx = 0; wx < nw; ++wx) {
                    int cnt = 0;
                    for (int iy = 0; iy < vws; ++iy)
                        for (int ix = 0; ix < vws; ++ix) {
                            const int y = wy * vws + iy, x = wx * vws + ix;
                            if (y < lh && x < lw) {
                                window_index.push_back(base + ((int64_t)tt * lh + y) * lw + x);
                                ++cnt;
                            }
                        }
                    const int32_t next = cu_window.back() + cnt * unit;
                    // torch.unique_consecutive: drop empty windows
                    if (next != cu_window.back()) cu_window.push_back(next);
                }
        base += (int64_t)t * lh * lw;
    }
}

// HF:vision_utils.py:81-127 get_vision_position_ids: (h, w) per patch in merge-block-major order.
void ze_vision_pos_ids_impl(const int32_t* grid_thw, int n_images, int merge, std::vector<int32_t>& hw) {
    hw.clear();
    for (int im = 0; im < n_images; ++im) {
        const int t = grid_thw[3 * im], h = grid_thw[3 * im + 1], w = grid_thw[3 * im + 2];
        for (int tt = 0; tt < t; ++tt)
            for (int by = 0; by < h / merge; ++by)
                for (int bx = 0; bx < w / merge; ++bx)
                    for (int my = 0; my < merge; ++my)
                        for (int mx = 0; mx < merge; ++mx) {
                            hw.push_back(by * merge + my);
                            hw.push_back(bx * merge + mx);
                        }
    }
}

// HF:models/qwen2_5_vl/modeling_qwen2_5_vl.py:892-1058 get_rope_index for one unpadded sequence (images only).
int ze_rope_index_impl(const int32_t* ids, int len, const int32_t* grid_thw, int n_images, int image_token_id,
                       int merge, int32_t* pos, int32_t* rope_delta) {
    int cur = 0, gi = 0, i = 0, maxpos = -1;
    while (i < len) {
        const bool img = ids[i] == image_token_id;
        int j = i;
        while (j < len && (ids[j] == image_token_id) == img) ++j;
        if (!img) {
            for (int t = i; t < j; ++t) {
                const int p = cur + (t - i);
                pos[t] = pos[len + t] = pos[2 * len + t] = p;
                maxpos = std::max(maxpos, p);
            }
            cur += j - i;
        } else {
            if (gi >= n_images) return -1;
            const int t_ = grid_thw[3 * gi], h = grid_thw[3 * gi + 1], w = grid_thw[3 * gi + 2];
            ++gi;
            const int lh = h / merge, lw = w / merge;
            if (j - i != t_ * lh * lw) return -2;
            int n = i;
            for (int tt = 0; tt < t_; ++tt)
                for (int y = 0; y < lh; ++y)
                    for (int x = 0; x < lw; ++x, ++n) {
                        pos[n] = tt + cur;
                        pos[len + n] = y + cur;
                        pos[2 * len + n] = x + cur;
                        maxpos = std::max(maxpos, std::max(tt + cur, std::max(y + cur, x + cur)));
                    }
            cur += std::max(h, w) / merge;
        }
        i = j;
    }
    *rope_delta = maxpos + 1 - len;
    return 0;
}
