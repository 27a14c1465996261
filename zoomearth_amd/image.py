"""DeviceImage: an RGB u8 image resident in HBM with the PIL methods the zoom chain calls.

replaces: the `PIL.Image` objects handled by cut_image / resize_image in the reference
(/root/reference/src/eval/infer.py:41-85,215,237): `.size`, `.width`, `.height`, `.crop(box)` (zero fill outside),
`.resize((w, h), Image.BICUBIC)`, `.convert("RGB")`.  crop() is lazy (a box on the parent tile); resize() runs ONE
fused crop+bicubic launch of the HIP front-end on the full-resolution tile, so a 5000-px tile is uploaded once
and every view / zoom of every question about it reads it in place.
"""
from __future__ import annotations

import itertools

import numpy as np
import torch

BICUBIC = 3
_ids = itertools.count(1)


class DeviceImage:
    def __init__(self, base: torch.Tensor, engine, box=None, key=None):
        assert base.dtype == torch.uint8 and base.dim() == 3 and base.shape[2] == 3 and base.is_cuda
        self.base = base.contiguous()
        self.engine = engine
        h, w = int(base.shape[0]), int(base.shape[1])
        self.box = tuple(int(v) for v in box) if box is not None else (0, 0, w, h)
        # identity used by the processor / model to reuse ViT features and prompt KV across the two stages
        self.key = key if key is not None else ("img", next(_ids))

    # ---- constructors
    @classmethod
    def from_numpy(cls, arr: np.ndarray, engine):
        arr = np.ascontiguousarray(arr, dtype=np.uint8)
        if arr.ndim != 3 or arr.shape[2] != 3:
            raise ValueError("expected an [H, W, 3] uint8 array")
        return cls(torch.from_numpy(arr).to(engine.device, non_blocking=False), engine)

    @classmethod
    def from_pil(cls, img, engine):
        return cls.from_numpy(np.asarray(img.convert("RGB")), engine)

    @classmethod
    def open(cls, path: str, engine):
        from PIL import Image

        Image.MAX_IMAGE_PIXELS = None  # 5000-px tiles (src/eval/infer.py:18)
        with Image.open(path) as im:
            return cls.from_pil(im, engine)

    # ---- PIL surface
    @property
    def width(self) -> int:
        return self.box[2] - self.box[0]

    @property
    def height(self) -> int:
        return self.box[3] - self.box[1]

    @property
    def size(self):
        return (self.width, self.height)

    def convert(self, mode: str):
        if mode != "RGB":
            raise ValueError("DeviceImage is RGB only")
        return self

    def crop(self, box):
        x0, y0, x1, y1 = (int(v) for v in box)
        bx, by = self.box[0], self.box[1]
        return DeviceImage(self.base, self.engine, (bx + x0, by + y0, bx + x1, by + y1),
                           key=("crop", self.key, (x0, y0, x1, y1)))

    def resize(self, size, resample=BICUBIC, **kw):
        if resample != BICUBIC:
            raise ValueError("only Image.BICUBIC is implemented on the device")
        w, h = int(size[0]), int(size[1])
        out = self.engine.crop_resize(self.base, self.box, (w, h))
        return DeviceImage(out, self.engine, key=("resize", self.key, (w, h)))

    def tensor(self) -> torch.Tensor:
        """Materialised u8 [H, W, 3] device tensor of this view (crop applied, zero fill outside the tile)."""
        h, w = int(self.base.shape[0]), int(self.base.shape[1])
        if self.box == (0, 0, w, h):
            return self.base
        return self.engine.crop_resize(self.base, self.box, (self.width, self.height))

    def numpy(self) -> np.ndarray:
        return self.tensor().cpu().numpy()
