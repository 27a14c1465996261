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

// f32 -> bf16, round to nearest even: ONE v_cvt_pk_bf16_f32 per pair (a vector conversion hipcc can see -- never inline asm in front
// of an MFMA: DESIGN.md 3, asm rule).  Rounds 1-5 did it in integer arithmetic (u + 0x7fff + lsb, NaNs patched): 7 VALU instructions
// per element, 400 of the 708 of a 64-row slice of the eight-phase GEMM's SwiGLU epilogue.  Same bits for every finite value and
// infinity; a NaN comes out as the hardware's quiet NaN instead of the input's payload with the quiet bit set.
// ZE_SOFT_BF16: the integer form (A/B builds).
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
#ifdef ZE_SOFT_BF16
    auto one = [](float f) -> uint32_t {
        const uint32_t u = __float_as_uint(f);
        if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40;
        return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
    };
    return one(lo) | (one(hi) << 16);
#else
    typedef __attribute__((ext_vector_type(2))) float ze_f32x2_;
    typedef __attribute__((ext_vector_type(2))) __bf16 ze_bf16x2_;
    const ze_bf16x2_ b = __builtin_convertvector(ze_f32x2_{lo, hi}, ze_bf16x2_);
    uint32_t r;
    __builtin_memcpy(&r, &b, 4);
    return r;
#endif
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return (bf16_t)(pack_bf16x2(f, 0.f) & 0xffffu); }
__device__ __forceinline__ float bf16_round(float f) { return bf16_to_f32(f32_to_bf16(f)); }

__device__ __forceinline__ float bf16lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// x * sigmoid(x) with v_rcp_f32 (1 ulp) instead of the correctly rounded fp32 division (v_div_scale / v_div_fmas / v_div_fixup + a Newton
// chain: 10 VALU instructions per quotient, 160 of the 388 of a 64-row slice of the SwiGLU epilogue).  __expf is a 1-ulp v_exp_f32
// already, and every caller rounds the product to bf16 (8 bits) next: the result differs from the divided one when the exact value
// lies within ~2 fp32 ulps of a bf16 rounding boundary, about 6 elements in 100,000, by one bf16 ulp.  ZE_EXACT_SILU_DIV: the division.
__device__ __forceinline__ float silu_f(float x) {
#ifdef ZE_EXACT_SILU_DIV
    return x / (1.0f + __expf(-x));
#else
    return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x));
#endif
}

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
