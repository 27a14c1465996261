"""ZoomEarthProcessor: the `Qwen2_5_VLProcessor` call surface on top of the HIP front-end.

replaces: `Qwen2_5_VLProcessor.from_pretrained(model_name, trust_remote_code=True, max_pixels=128*128*28*28)` and
`processor(text=..., images=..., return_tensors="pt", padding="longest")`
(/root/reference/src/eval/infer.py:102-107,153-157; src/demo.py:7-12,127;
HF:models/qwen2_5_vl/processing_qwen2_5_vl.py:59-62, HF:models/qwen2_vl/image_processing_pil_qwen2_vl.py:197-246).
The image part (smart_resize, bicubic, rescale, normalise, patchify) runs in HIP; placeholder expansion and
tokenisation are host-side.
"""
from __future__ import annotations

import json
import os

import numpy as np
import torch

from .image import DeviceImage
from .tokenizer import ZoomEarthTokenizer

IMAGE_TOKEN = "<|image_pad|>"
_DEFAULT_ENGINE = None


def set_default_engine(engine) -> None:
    """The most recently created model registers its engine here so a processor built before / after it
    finds the device front-end at call time."""
    global _DEFAULT_ENGINE
    _DEFAULT_ENGINE = engine


class BatchFeature(dict):
    """dict of tensors with `.to(device)` (HF BatchFeature surface used at src/eval/infer.py:107)."""

    def to(self, device, *a, **kw):
        for k, v in list(self.items()):
            if isinstance(v, torch.Tensor):
                self[k] = v.to(device)
        return self

    def __getattr__(self, item):
        try:
            return self[item]
        except KeyError as e:
            raise AttributeError(item) from e


class ZoomEarthProcessor:
    def __init__(self, tokenizer: ZoomEarthTokenizer, min_pixels: int = 56 * 56, max_pixels: int = 28 * 28 * 1280,
                 merge_size: int = 2, engine=None):
        self.tokenizer = tokenizer
        self.min_pixels, self.max_pixels, self.merge_size = int(min_pixels), int(max_pixels), merge_size
        self.engine = engine
        self.image_token = IMAGE_TOKEN
        self.image_token_id = tokenizer.convert_tokens_to_ids(IMAGE_TOKEN)

    @classmethod
    def from_pretrained(cls, path: str, trust_remote_code=None, min_pixels=None, max_pixels=None, **kw):
        tok = ZoomEarthTokenizer.from_pretrained(path)
        mn, mx = 56 * 56, 28 * 28 * 1280  # HF defaults (image_processing_pil_qwen2_vl.py:89)
        pc = os.path.join(path, "preprocessor_config.json")
        if os.path.exists(pc):
            with open(pc, encoding="utf-8") as f:
                c = json.load(f)
            size = c.get("size") or {}
            mn = c.get("min_pixels", size.get("shortest_edge", mn))
            mx = c.get("max_pixels", size.get("longest_edge", mx))
        return cls(tok, min_pixels if min_pixels is not None else mn, max_pixels if max_pixels is not None else mx)

    # ------------------------------------------------------------------ images
    def _engine(self):
        e = self.engine or _DEFAULT_ENGINE
        if e is None:
            raise RuntimeError("no ze_engine available: create the model (ZoomEarthForConditionalGeneration) first")
        return e

    def _to_device_image(self, img) -> DeviceImage:
        if isinstance(img, DeviceImage):
            return img
        e = self._engine()
        if isinstance(img, np.ndarray):
            return DeviceImage.from_numpy(img, e)
        if isinstance(img, torch.Tensor):
            return DeviceImage(img.to(e.device), e)
        return DeviceImage.from_pil(img, e)  # PIL.Image

    def preprocess_images(self, images):
        """flat list of images -> (pixel_values f32 [sum N, 1176] device, grids [[1,gh,gw],...], keys)."""
        e = self._engine()
        pvs, grids, keys = [], [], []
        for im in images:
            di = self._to_device_image(im)
            cache = getattr(di, "_pv_cache", None)
            if cache is None or cache[0] != (self.min_pixels, self.max_pixels):
                pv, grid = e.preprocess_image(di.tensor(), self.min_pixels, self.max_pixels)
                di._pv_cache = ((self.min_pixels, self.max_pixels), pv, grid)
            _, pv, grid = di._pv_cache
            pvs.append(pv)
            grids.append(list(grid))
            keys.append((di.key, self.min_pixels, self.max_pixels))
        return (torch.cat(pvs) if len(pvs) > 1 else pvs[0]), grids, keys

    # ------------------------------------------------------------------ __call__
    def __call__(self, text=None, images=None, return_tensors="pt", padding=False, **kw):
        texts = [text] if isinstance(text, str) else list(text)
        out = BatchFeature()
        grids = []
        if images is not None:
            flat = []
            for it in (images if isinstance(images, (list, tuple)) else [images]):
                flat.extend(it if isinstance(it, (list, tuple)) else [it])
            pv, grids, keys = self.preprocess_images(flat)
            out["pixel_values"] = pv
            out["image_grid_thw"] = torch.tensor(grids, dtype=torch.long)
            out["image_keys"] = keys
        # placeholder expansion: each <|image_pad|> -> t*h*w/merge^2 copies, images consumed in order
        it = iter(grids)
        expanded = []
        for t in texts:
            parts = t.split(IMAGE_TOKEN)
            s = parts[0]
            for p in parts[1:]:
                try:
                    g = next(it)
                except StopIteration:
                    raise ValueError("more <|image_pad|> placeholders than images") from None
                s += IMAGE_TOKEN * (g[0] * g[1] * g[2] // self.merge_size ** 2) + p
            expanded.append(s)
        enc = self.tokenizer(expanded, padding=padding, return_tensors="pt")
        out["input_ids"] = enc["input_ids"]
        out["attention_mask"] = enc["attention_mask"]
        out["mm_token_type_ids"] = (enc["input_ids"] == self.image_token_id).to(torch.int32)
        return out

    def batch_decode(self, *a, **kw):
        return self.tokenizer.batch_decode(*a, **kw)
