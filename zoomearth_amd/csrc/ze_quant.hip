// FP8 (OCP E4M3, the fp8 of gfx950) weight quantisation for the decode weight stream (BASELINE.json configs[4]).
// One workgroup per weight row: amax -> scale 2^k with k the smallest integer such that amax / 2^k <= 448 ->
// q = cvt_fp8(w / 2^k) (exact division, round to nearest even by v_cvt_pk_fp8_f32) -> the bf16 row is overwritten
// with q * 2^k, which bf16 represents exactly.  The prefill / ViT GEMMs and the batched-decode GEMMs keep reading the
// bf16 arena, the batch-1 decode GEMVs stream the fp8 copy (half the bytes) and dequantise in registers: both
// compute with IDENTICAL weight values.  Restated in numpy by oracle/fp8.py.
#include "ze_kernels.h"

typedef float f32x2_t __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(256) k_quantize_rows(bf16_t* __restrict__ w, int cols, int ld,
                                                       uint8_t* __restrict__ q, int ld8, float* __restrict__ scale) {
    bf16_t* row = w + (size_t)blockIdx.x * ld;
    uint8_t* qrow = q + (size_t)blockIdx.x * ld8;
    const int tid = threadIdx.x;
    float amax = 0.f;
    for (int i = tid; i < cols; i += 256) amax = fmaxf(amax, fabsf(bf16_to_f32(row[i])));
    amax = wave_max(amax);
    __shared__ float red[4];
    if ((tid & 63) == 0) red[tid >> 6] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    int k = 0;
    if (amax > 0.f) {
        int e;
        const float m = frexpf(amax / 448.0f, &e);  // amax / 448 = m * 2^e, m in [0.5, 1)
        k = (m == 0.5f) ? e - 1 : e;
    }
    const float s = ldexpf(1.0f, k), inv = ldexpf(1.0f, -k);
    if (tid == 0) scale[blockIdx.x] = s;
    for (int i = tid * 2; i < ld8; i += 512) {  // two elements per thread-step (one cvt_pk), zero beyond cols
        const float a = i < cols ? bf16_to_f32(row[i]) * inv : 0.f;
        const float b = i + 1 < cols ? bf16_to_f32(row[i + 1]) * inv : 0.f;
        const int packed = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
        *reinterpret_cast<uint16_t*>(qrow + i) = (uint16_t)(packed & 0xffff);
        const f32x2_t back = __builtin_amdgcn_cvt_pk_f32_fp8(packed, false);
        if (i < cols) row[i] = f32_to_bf16(back.x * s);
        if (i + 1 < cols) row[i + 1] = f32_to_bf16(back.y * s);
    }
}

void ze_launch_quantize_rows(bf16_t* w, int rows, int cols, int ld, uint8_t* q, int ld8, float* scale, hipStream_t s) {
    if (rows <= 0) return;
    k_quantize_rows<<<rows, 256, 0, s>>>(w, cols, ld, q, ld8, scale);
}
