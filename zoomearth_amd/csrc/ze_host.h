// Host-side declarations shared by the engine translation units.
#pragma once
#include <stdint.h>

#include <vector>

struct ze_coeffs {
    int ksize = 0, out_size = 0, max_cnt = 0;
    std::vector<int> xmin, xcnt, kk;
};
void ze_bicubic_coeffs(int in_size, int out_size, ze_coeffs* c);
int ze_smart_resize_impl(int height, int width, int factor, int64_t min_pixels, int64_t max_pixels, int* out_h,
                         int* out_w);
void ze_window_index_impl(const int32_t* grid_thw, int n_images, int merge, int window_size, int patch,
                          std::vector<int64_t>& window_index, std::vector<int32_t>& cu_window);
void ze_vision_pos_ids_impl(const int32_t* grid_thw, int n_images, int merge, std::vector<int32_t>& hw);
int ze_rope_index_impl(const int32_t* ids, int len, const int32_t* grid_thw, int n_images, int image_token_id,
                       int merge, int32_t* pos, int32_t* rope_delta);
