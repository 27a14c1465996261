"""Synthetic inputs for benchmarks and demos: the repo's counter-based integer PRNG (splitmix64 stream),
Irwin-Hall normals, token-id streams and band-limited u8 tiles.

Product-side twin of oracle/prng.py (kept separate so nothing under zoomearth_amd/ or bench.py's timed path
imports the oracle); tests/test_synth_cpu.py asserts both produce identical streams.  The device mirror used
for the weights is csrc/ze_prng.h.
"""
from __future__ import annotations

import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)

# sqrt(4 * (65536**2 - 1) / 12): standard deviation of a sum of four uniform u16
IH4_STD = 37837.22722412452


def mix64(z: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on a uint64 array (wrap-around arithmetic)."""
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def stream64(seed: int, start: int, n: int) -> np.ndarray:
    """Elements start..start+n-1 of the counter stream: mix64(seed + (i+1)*GOLDEN)."""
    idx = np.arange(start + 1, start + n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        return mix64(np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + idx * _GOLDEN)


def fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def tensor_seed(global_seed: int, name: str) -> int:
    """Per-tensor seed: mix64(global_seed ^ fnv1a64(name))."""
    return int(mix64(np.array([(global_seed ^ fnv1a64(name)) & 0xFFFFFFFFFFFFFFFF], dtype=np.uint64))[0])


def normal_ih4(seed: int, n: int, std: float, start: int = 0) -> np.ndarray:
    """n approximately-normal float32 values (Irwin-Hall of four u16), exactly reproducible.

    value_i = float32(int32(s_i) - 131070) * float32(std / IH4_STD),
    s_i = sum of the four 16-bit fields of stream64(seed)[i].
    """
    h = stream64(seed, start, n)
    m = np.uint64(0xFFFF)
    s = (h & m) + ((h >> np.uint64(16)) & m) + ((h >> np.uint64(32)) & m) + (h >> np.uint64(48))
    c = np.float32(std / IH4_STD)
    return (s.astype(np.int64) - 131070).astype(np.float32) * c


def uniform_ints(seed: int, n: int, lo: int, hi: int, start: int = 0) -> np.ndarray:
    """n int64 values in [lo, hi): lo + (stream >> 11) % (hi - lo)."""
    h = stream64(seed, start, n) >> np.uint64(11)
    return (lo + (h % np.uint64(hi - lo))).astype(np.int64)


def bytes_stream(seed: int, n: int) -> np.ndarray:
    """n uint8 values: little-endian bytes of the 64-bit stream."""
    nw = (n + 7) // 8
    h = stream64(seed, 0, nw)
    return h.view(np.uint8)[:n].copy()


def synthetic_tile(seed: int, height: int, width: int) -> np.ndarray:
    """u8 [H, W, 3] synthetic tile, band-limited so bicubic sees natural-ish content
    (SURVEY.md 8d asks for low-passed noise; integer-exact so every box regenerates it).

    coarse grid G[(H>>3)+2, (W>>3)+2, 3] = bytes of stream(seed);
    fine noise  n[y,x,c] = (byte of stream(seed+1) & 15) - 8;
    tile[y,x,c] = clip(((8-fy)*((8-fx)*G00 + fx*G01) + fy*((8-fx)*G10 + fx*G11) + 32 >> 6) + n)
    with (cy,fy) = divmod(y,8), (cx,fx) = divmod(x,8).
    """
    hc, wc = (height >> 3) + 2, (width >> 3) + 2
    g = bytes_stream(seed, hc * wc * 3).reshape(hc, wc, 3).astype(np.int16)
    x = np.arange(width)
    cx, fx = x >> 3, (x & 7).astype(np.int16)[None, :, None]
    a = (8 - fx) * g[:, cx] + fx * g[:, cx + 1]  # [hc, W, 3], scaled by 8
    out = np.empty((height, width, 3), dtype=np.int16)
    for fy in range(8):
        rows = out[fy::8]
        nr = rows.shape[0]
        rows[...] = ((8 - fy) * a[:nr] + fy * a[1 : nr + 1] + 32) >> 6
    n = (bytes_stream(seed + 1, height * width * 3) & 15).astype(np.int16).reshape(height, width, 3) - 8
    out += n
    np.clip(out, 0, 255, out=out)
    return out.astype(np.uint8)
