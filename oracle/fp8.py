"""OCP FP8 E4M3 (the fp8 of gfx950 / CDNA4: bias 7, 3 mantissa bits, max 448, no infinity) in numpy, and the
per-row power-of-two-scale weight quantisation of the engine's fp8 decode path (BASELINE.json configs[4]: "fp8
weights").  Test infrastructure: restates zoomearth_amd/csrc/ze_quant.hip so the GPU quantiser can be checked bit
for bit and the oracle model can run on the same dequantised weights.

Scheme: for a weight row w, scale = 2^k with k the smallest integer such that max|w| / 2^k <= 448; q = e4m3(w / 2^k)
(round to nearest even; the division is exact); the value the model computes with is q * 2^k, which is exactly
representable in bf16 (3 mantissa bits), so the bf16 prefill GEMMs (on the dequantised copy) and the fp8 decode
stream (dequantised in registers) use IDENTICAL weights."""
from __future__ import annotations

import numpy as np

E4M3_MAX = 448.0


def e4m3_round(x: np.ndarray) -> np.ndarray:
    """Round fp32 values to the nearest E4M3 value (ties to even), saturating at +-448; returns fp32 values."""
    x = np.asarray(x, dtype=np.float32)
    ax = np.abs(x).astype(np.float64)
    with np.errstate(divide="ignore"):
        e = np.floor(np.log2(np.where(ax > 0, ax, 1.0)))
    e = np.maximum(e, -6.0)                      # below 2^-6 the grid is the subnormal one: step 2^-9
    step = np.exp2(e - 3.0)
    q = np.rint(ax / step) * step                # np.rint = round half to even
    q = np.minimum(q, E4M3_MAX)
    return (np.sign(x) * q).astype(np.float32)


def e4m3_bits(v: np.ndarray) -> np.ndarray:
    """Bit patterns (uint8) of values that are already on the E4M3 grid."""
    v = np.asarray(v, dtype=np.float32)
    a = np.abs(v).astype(np.float64)
    sign = (np.signbit(v)).astype(np.uint8) << 7
    out = np.zeros(v.shape, dtype=np.uint8)
    nz = a > 0
    e = np.floor(np.log2(np.where(nz, a, 1.0)))
    sub = e < -6
    mant_n = np.rint((a / np.exp2(e) - 1.0) * 8.0).astype(np.int64)      # normal: 1.mmm
    mant_s = np.rint(a / 2.0 ** -9).astype(np.int64)                      # subnormal: 0.mmm * 2^-6
    bits = np.where(sub, mant_s, ((e + 7).astype(np.int64) << 3) | mant_n)
    out[nz] = bits[nz].astype(np.uint8)
    return out | sign


def row_scale_exponent(w: np.ndarray) -> np.ndarray:
    """k per row: smallest integer with max|row| / 2^k <= 448 (k = 0 for an all-zero row)."""
    amax = np.abs(np.asarray(w, dtype=np.float32)).max(axis=1).astype(np.float64)
    t = amax / E4M3_MAX
    m, e = np.frexp(t)                            # t = m * 2^e, m in [0.5, 1)
    k = np.where(m == 0.5, e - 1, e)
    return np.where(amax > 0, k, 0).astype(np.int32)


def quantize_rows(w: np.ndarray):
    """-> (bits uint8 [N, K], k int32 [N], dequantised fp32 [N, K] = e4m3 value * 2^k)."""
    w = np.asarray(w, dtype=np.float32)
    k = row_scale_exponent(w)
    s = np.exp2(k.astype(np.float64)).astype(np.float32)[:, None]
    q = e4m3_round(w / s)
    return e4m3_bits(q), k, (q * s).astype(np.float32)
