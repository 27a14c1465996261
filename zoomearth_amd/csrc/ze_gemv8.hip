// FP8 (OCP E4M3) instantiations of the decode weight-streaming kernel: same shape policy as the bf16 stream,
// 1024-element chunks, per-row power-of-two scales applied once in the epilogue (ze_quant.hip, oracle/fp8.py).
#include "ze_gemv_kernel.h"

bool ze_launch_gemv8(int epi, const ze_gemv_args& a, hipStream_t s) {
    if ((size_t)((a.K + 511) / 512 + 2) * 1024 > 60000) return false;  // x must fit the LDS stage
    const bool long_k = a.K > 4096;
    const bool many_rows = a.N >= 8192;
    if (a.K % 16) return false;
    switch (epi) {
        // two row pairs per wave-iteration where there are many rows: the f32 widening of x is shared by four rows
        case ZE_GV_QKV_ROPE: launch_gemv_cfg<ZE_GV_QKV_ROPE, 1, 1, 4, 8>(a, s); break;
        case ZE_GV_SWIGLU:
            if (many_rows && ze_gemv_knobs[1] != 2) launch_gemv_cfg<ZE_GV_SWIGLU, 2, 1, 4, 8>(a, s);
            else launch_gemv_cfg<ZE_GV_SWIGLU, 1, 1, 4, 8>(a, s);
            break;
        case ZE_GV_RESIDUAL:
            if (long_k) launch_gemv_cfg<ZE_GV_RESIDUAL, 1, 4, 6, 8>(a, s);
            else launch_gemv_cfg<ZE_GV_RESIDUAL, 1, 1, 4, 8>(a, s);
            break;
        case ZE_GV_LOGITS:
            if (many_rows) launch_gemv_cfg<ZE_GV_LOGITS, 2, 1, 4, 8>(a, s);
            else launch_gemv_cfg<ZE_GV_LOGITS, 1, 1, 4, 8>(a, s);
            break;
        default:
            if (long_k) launch_gemv_cfg<ZE_GV_PLAIN, 1, 4, 6, 8>(a, s);
            else launch_gemv_cfg<ZE_GV_PLAIN, 1, 1, 4, 8>(a, s);
            break;
    }
    return true;
}
