// Internal helpers shared by the HIP translation units of libzoomearth_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/zoomearth.h"

typedef uint16_t bf16_t;  // raw bfloat16 bits

#define ZE_WAVE 64

static inline int ze_pad32(int x) { return (x + 31) / 32 * 32; }
static inline int ze_cdiv(int a, int b) { return (a + b - 1) / b; }

// ------------------------------------------------------------------ device numerics
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even, NaN kept a NaN (the integer trick alone turns some NaNs into inf/0)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    return (bf16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ float bf16_round(float f) { return bf16_to_f32(f32_to_bf16(f)); }

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
}
__device__ __forceinline__ float bf16lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ------------------------------------------------------------------ host-side error plumbing
struct ze_error_sink {
    std::string msg;
};
extern thread_local std::string ze_global_error;

#define ZE_HIP(call)                                                                         \
    do {                                                                                     \
        hipError_t _e = (call);                                                              \
        if (_e != hipSuccess) {                                                              \
            return ze_fail(e, ZE_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(_e)); \
        }                                                                                    \
    } while (0)

int ze_fail(ze_engine* e, int code, const std::string& msg);
