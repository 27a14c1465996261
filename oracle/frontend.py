"""Oracle for the integer image front-end: Pillow-exact bicubic
resize, crop, smart_resize, rescale+normalize+patchify.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Everything here is integer /
fixed-point or a 3x256 float32 LUT, so the HIP kernels must match it bit for bit.

What it restates:
  * PIL.Image.resize(size, Image.BICUBIC) on RGB u8, as called by the reference at
    /root/reference/src/eval/infer.py:78-85 (resize_image) and src/demo.py:86-93, and by
    transformers' PIL image-processor backend
    (HF:models/qwen2_vl/image_processing_pil_qwen2_vl.py:126-150 -> image_processing_backends.py:521-570).
    Pillow's C source (src/libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc,
    ImagingResampleHorizontal_8bpc / Vertical_8bpc) is a third-party dependency absent from
    /root/reference (Pillow is unpinned at requirements.txt:26; 12.2.0 in this image); its
    published algorithm is restated below and pinned by tests/golden/bicubic_*.npz which were
    produced by Pillow itself.
  * PIL.Image.crop zero-fill semantics, as used by cut_image (src/eval/infer.py:41-76).
  * smart_resize (HF:models/qwen2_vl/image_processing_pil_qwen2_vl.py:57-83).
  * rescale / normalize / patchify (same file :152-187,197-246; HF:image_transforms.py rescale, normalize).
"""
from __future__ import annotations

import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2  # Pillow Resample.c

OPENAI_CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
OPENAI_CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _bicubic(x: float) -> float:
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def bicubic_coeffs(in_size: int, out_size: int):
    """Pillow precompute_coeffs + normalize_coeffs_8bpc for one axis.

    Returns (xmin[int32 out], xcnt[int32 out], k[int32 out, ksize]); taps beyond xcnt are 0.
    """
    scale = in_size / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    xmin = np.zeros(out_size, dtype=np.int32)
    xcnt = np.zeros(out_size, dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        lo = int(center - support + 0.5)  # C truncation; argument is > -1 so int() == trunc
        if lo < 0:
            lo = 0
        hi = int(center + support + 0.5)
        if hi > in_size:
            hi = in_size
        n = hi - lo
        w = [_bicubic((x + lo - center + 0.5) * ss) for x in range(n)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        xmin[xx] = lo
        xcnt[xx] = n
        for x, v in enumerate(w):
            if v < 0:
                kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS))
            else:
                kk[xx, x] = int(0.5 + v * (1 << PRECISION_BITS))
    return xmin, xcnt, kk


def _resample_axis0(img: np.ndarray, out_size: int) -> np.ndarray:
    """Resample along axis 0 of a [N, ...] u8 array (fixed-point, rounded to u8)."""
    in_size = img.shape[0]
    xmin, xcnt, kk = bicubic_coeffs(in_size, out_size)
    out = np.empty((out_size,) + img.shape[1:], dtype=np.uint8)
    src = img.astype(np.int64)
    for xx in range(out_size):
        n = int(xcnt[xx])
        lo = int(xmin[xx])
        acc = np.tensordot(kk[xx, :n].astype(np.int64), src[lo : lo + n], axes=(0, 0))
        acc = (acc + (1 << (PRECISION_BITS - 1))) >> PRECISION_BITS
        out[xx] = np.clip(acc, 0, 255).astype(np.uint8)
    return out


def resize_bicubic(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """PIL `Image.resize((out_w, out_h), Image.BICUBIC)` on a u8 [H, W, C] array.

    Horizontal pass first (result rounded to u8), then vertical; an axis whose size does not
    change is skipped (Pillow ImagingResample need_horizontal / need_vertical).
    """
    assert img.dtype == np.uint8 and img.ndim == 3
    h, w, _ = img.shape
    cur = img
    if out_w != w:
        cur = _resample_axis0(np.ascontiguousarray(cur.transpose(1, 0, 2)), out_w).transpose(1, 0, 2)
    if out_h != h:
        cur = _resample_axis0(np.ascontiguousarray(cur), out_h)
    return np.ascontiguousarray(cur)


def crop_zero_fill(img: np.ndarray, box) -> np.ndarray:
    """PIL `Image.crop((x0, y0, x1, y1))`: region outside the image is zero."""
    x0, y0, x1, y1 = (int(v) for v in box)
    h, w, c = img.shape
    out = np.zeros((max(y1 - y0, 0), max(x1 - x0, 0), c), dtype=img.dtype)
    sx0, sy0, sx1, sy1 = max(x0, 0), max(y0, 0), min(x1, w), min(y1, h)
    if sx1 > sx0 and sy1 > sy0:
        out[sy0 - y0 : sy1 - y0, sx0 - x0 : sx1 - x0] = img[sy0:sy1, sx0:sx1]
    return out


def smart_resize(height: int, width: int, factor: int = 28, min_pixels: int = 56 * 56,
                 max_pixels: int = 14 * 14 * 4 * 1280):
    """HF:models/qwen2_vl/image_processing_pil_qwen2_vl.py:57-83 (Python round = banker's)."""
    if max(height, width) / min(height, width) > 200:
        raise ValueError(
            f"absolute aspect ratio must be smaller than 200, got {max(height, width) / min(height, width)}"
        )
    h_bar = round(height / factor) * factor
    w_bar = round(width / factor) * factor
    if h_bar * w_bar > max_pixels:
        beta = math.sqrt((height * width) / max_pixels)
        h_bar = max(factor, math.floor(height / beta / factor) * factor)
        w_bar = max(factor, math.floor(width / beta / factor) * factor)
    elif h_bar * w_bar < min_pixels:
        beta = math.sqrt(min_pixels / (height * width))
        h_bar = math.ceil(height * beta / factor) * factor
        w_bar = math.ceil(width * beta / factor) * factor
    return h_bar, w_bar


def normalize_lut() -> np.ndarray:
    """float32 [3, 256]: LUT[c][v] = (f32(f64(v) * (1/255)) - f32(mean_c)) / f32(std_c).

    HF rescale multiplies in float64 then casts to float32 (image_transforms.rescale with
    dtype=np.float32); normalize runs in float32 (image_transforms.normalize).
    """
    v = np.arange(256, dtype=np.float64)
    r = (v * (1 / 255)).astype(np.float32)
    mean = np.array(OPENAI_CLIP_MEAN, dtype=np.float32)
    std = np.array(OPENAI_CLIP_STD, dtype=np.float32)
    return ((r[None, :] - mean[:, None]) / std[:, None]).astype(np.float32)


def patchify(img_u8: np.ndarray, patch: int = 14, merge: int = 2, temporal: int = 2):
    """u8 [H, W, 3] (H, W multiples of patch*merge) -> (float32 [gh*gw, 3*temporal*patch*patch], gh, gw).

    Row order (block-row, block-col, 2x2 inside the block); column order (C, T, ph, pw); the
    T copies are a broadcast of the single frame
    (HF:models/qwen2_vl/image_processing_pil_qwen2_vl.py:152-187).
    """
    h, w, c = img_u8.shape
    gh, gw = h // patch, w // patch
    lut = normalize_lut()
    x = np.stack([lut[ch][img_u8[:, :, ch]] for ch in range(c)], axis=0)  # [C, H, W] f32
    x = x.reshape(c, gh // merge, merge, patch, gw // merge, merge, patch)
    x = x.transpose(1, 4, 2, 5, 0, 3, 6)  # gh/m, gw/m, m, m, C, ph, pw
    x = np.broadcast_to(x[:, :, :, :, :, None, :, :], x.shape[:5] + (temporal,) + x.shape[5:])
    return np.ascontiguousarray(x.reshape(gh * gw, c * temporal * patch * patch)), gh, gw


def image_to_pixel_values(img_u8: np.ndarray, min_pixels: int = 3136, max_pixels: int = 128 * 128 * 28 * 28,
                          patch: int = 14, merge: int = 2, temporal: int = 2):
    """Qwen2VLImageProcessor._preprocess for one image: smart_resize -> bicubic -> LUT -> patchify.

    Returns (pixel_values float32 [N, 1176], (1, gh, gw)).
    """
    h, w, _ = img_u8.shape
    rh, rw = smart_resize(h, w, factor=patch * merge, min_pixels=min_pixels, max_pixels=max_pixels)
    if (rh, rw) != (h, w):
        img_u8 = resize_bicubic(img_u8, rw, rh)
    pv, gh, gw = patchify(img_u8, patch, merge, temporal)
    return pv, (1, gh, gw)
