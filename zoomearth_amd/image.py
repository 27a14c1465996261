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


def decode_rgb(path: str) -> np.ndarray:
    """Host decode of a tile file to RGB u8 [H, W, 3] (Image.open(fp).convert("RGB"), src/eval/infer.py:215,237)."""
    from PIL import Image

    Image.MAX_IMAGE_PIXELS = None
    with Image.open(path) as im:
        raw = _raw_rgb_strips(im)
        if raw is None:
            return np.array(im.convert("RGB"), dtype=np.uint8)  # a writable, contiguous copy
        w, h = im.size
    # Uncompressed 8-bit RGB (the usual way a 5000-px TIFF tile is stored): the file's strips ARE the pixels.  Reading them
    # straight into the array keeps the interpreter lock free (readinto releases it) -- through PIL the same tile goes
    # through load() + tobytes(), 64 KB at a time under the lock, and three decode threads then take 0.66 s per tile instead of
    # 0.16 and starve the threads that drive the GPU (measured on the real entry point: tools/bench_infer_e2e.py)
    arr = np.empty((h, w, 3), dtype=np.uint8)
    with open(path, "rb", buffering=0) as f:
        for y0, y1, off in raw:
            f.seek(off)
            mv = memoryview(arr[y0:y1]).cast("B")
            got = 0
            while got < len(mv):
                n = f.readinto(mv[got:])
                if not n:
                    raise OSError(f"{path}: truncated image data")
                got += n
    return arr


def _raw_rgb_strips(im):
    """[(first row, end row, file offset)] when the opened image is uncompressed, tightly packed, top-down 8-bit RGB stored
    in full-width strips (PIL's tile list says so); None otherwise."""
    try:
        if im.mode != "RGB" or not im.tile:
            return None
        w, h = im.size
        out, y = [], 0
        for t in im.tile:
            name, (x0, y0, x1, y1), off, args = t[0], t[1], t[2], t[3]
            rawmode = args if isinstance(args, str) else args[0]
            stride = 0 if isinstance(args, str) or len(args) < 2 else args[1]
            orient = 1 if isinstance(args, str) or len(args) < 3 else args[2]
            if name != "raw" or rawmode != "RGB" or stride not in (0, 3 * w) or orient != 1 or x0 != 0 or x1 != w or y0 != y:
                return None
            out.append((y0, y1, int(off)))
            y = y1
        return out if y == h else None
    except Exception:
        return None


class TilePrefetcher:
    """One tile decode per TILE instead of two per QUESTION, and off the GPU's critical path.

    replaces: the two `Image.open(image_fp).convert("RGB")` calls per question of the reference loop
    (/root/reference/src/eval/infer.py:215,237).  A 5000-px TIFF decodes in 0.2-0.5 s on one
    core -- as long as a whole question takes on the GPU -- so the decode of the NEXT distinct tile of the question
    stream runs in a background thread (PIL releases the GIL inside its decoders) into pinned host memory while the
    current tile's questions run; `get(path)` then only pays the PCIe copy.  Questions arrive grouped by tile
    (accel.shard_by_tile), so one tile ahead is enough; one resident tile + one decoded-ahead tile bound the memory.

        pf = TilePrefetcher(paths_in_question_order, engine)
        for q in questions: tile = pf.get(q.path)
    """

    def __init__(self, paths, engine, decode=decode_rgb, pin: bool = True, depth: int = 1, workers: int = 1):
        """depth: decoded tiles held ahead of the one in use (75 MB of pinned memory each at 5000 px); workers: decode
        threads (a stream that answers 50 questions/s needs ~5 tiles/s: more than one core's worth of TIFF / PNG decode)."""
        import threading

        self._engine = engine
        self._decode = decode
        self._pin = pin
        self._depth, self._workers = max(1, int(depth)), max(1, int(workers))
        order = []
        for p in paths:  # distinct tiles in first-use order
            if not order or order[-1] != p:
                order.append(p)
        self._order = order
        self._next = 0            # index in _order of the next tile to decode ahead
        self._ready = {}          # path -> decoded host array / exception
        self._busy = 0            # decodes in progress
        self._current = (None, None)
        self._cv = threading.Condition()
        self._skipped = set()     # tiles the caller will not ask for after all (claimed by another rank: accel.TileClaims)
        self._in_copy = []        # (event, pinned buffer) of uploads that may still be running
        self._pinned = []         # free pinned staging buffers, reused tile after tile (hipHostMalloc of 75 MB per tile costs
        #                           more than the decode itself and serialises with the GPU's work)
        self.decodes = 0
        self.decode_s = 0.0       # seconds spent inside the decoder (all threads)
        self.wait_s = 0.0         # seconds get() waited for a tile that was not ready
        self._kick()

    def _kick(self):
        import threading

        with self._cv:
            while (self._next < len(self._order) and self._busy < self._workers
                   and len(self._ready) + self._busy < self._depth):
                path = self._order[self._next]
                self._next += 1
                if path in self._skipped:
                    continue
                self._busy += 1
                threading.Thread(target=self._work, args=(path,), daemon=True).start()

    def _to_pinned(self, arr):
        """The decoded tile in a pinned buffer of the pool (allocated on first use, recycled by get() once uploaded)."""
        n = int(arr.size)
        buf = None
        with self._cv:
            for i, b in enumerate(self._pinned):
                if b.numel() >= n:
                    buf = self._pinned.pop(i)
                    break
        if buf is None:
            buf = torch.empty(n, dtype=torch.uint8).pin_memory()
        view = buf[:n].view(arr.shape)
        view.numpy()[...] = arr
        view._ze_pool_buffer = buf
        return view

    def _work(self, path):
        import time
        t0 = time.perf_counter()
        try:
            arr = self._decode(path)
            if self._pin and torch.cuda.is_available():
                arr = self._to_pinned(arr)
        except Exception as ex:  # surfaced by get()
            arr = ex
        with self._cv:
            if path in self._skipped:     # skipped while it was being decoded: the staging buffer goes back to the pool
                buf = getattr(arr, "_ze_pool_buffer", None)
                if buf is not None:
                    self._pinned.append(buf)
            else:
                self._ready[path] = arr
            self._busy -= 1
            self.decodes += 1
            self.decode_s += time.perf_counter() - t0
            self._cv.notify_all()
        self._kick()

    def skip(self, path: str) -> None:
        """The caller will not get() this tile (work stealing: another rank claimed it): a copy decoded ahead is dropped and its
        place in the look-ahead window goes to the next tile."""
        with self._cv:
            self._skipped.add(path)
            arr = self._ready.pop(path, None)
            buf = getattr(arr, "_ze_pool_buffer", None) if arr is not None else None
            if buf is not None:
                self._pinned.append(buf)
        self._kick()

    def ready(self, path: str) -> bool:
        """True when get(path) would not wait for a decode in progress (the caller has better things to do meanwhile)."""
        if self._current[0] == path:
            return True
        with self._cv:
            return path in self._ready or path not in self._order[:self._next]

    def get(self, path: str) -> DeviceImage:
        if self._current[0] == path:
            return self._current[1]
        import time
        t0 = time.perf_counter()
        with self._cv:
            while path not in self._ready:
                queued = path in self._order[:self._next]     # requested from a worker: in progress or done
                if not queued or (self._busy == 0 and path not in self._ready):
                    break
                self._cv.wait(timeout=0.05)
            arr = self._ready.pop(path, None)
        if arr is None:  # out-of-order request: decode here
            arr = self._decode(path)
            self.decodes += 1
        self.wait_s += time.perf_counter() - t0
        if isinstance(arr, Exception):
            self._kick()
            raise arr
        t = arr if isinstance(arr, torch.Tensor) else torch.from_numpy(arr)
        dev = t.to(self._engine.device, non_blocking=True)
        buf = getattr(t, "_ze_pool_buffer", None)
        if buf is not None:  # the staging buffer goes back to the pool once the copy has run (checked at later calls)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self._engine.device))
            self._in_copy.append((ev, buf))
        still = []
        for ev, b in self._in_copy:
            if ev.query():
                with self._cv:
                    self._pinned.append(b)
            else:
                still.append((ev, b))
        self._in_copy = still
        img = DeviceImage(dev, self._engine)
        self._current = (path, img)  # the previous tile's HBM is released with its last reference
        self._kick()
        return img
