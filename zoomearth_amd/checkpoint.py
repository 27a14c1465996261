"""Minimal safetensors reader / writer (checkpoint format of SURVEY.md row a16).

replaces: the HF `from_pretrained` weight loading used at /root/reference/src/eval/infer.py:147-150.  Both key
layouts are accepted downstream by ze_load_weight (5.x `model.visual.* / model.language_model.*` and 4.49-era
`visual.* / model.layers.*`).  Tensors are memory-mapped and handed to the engine without dtype conversion on
the host (bf16 stays raw uint16 bits).
"""
from __future__ import annotations

import glob
import json
import os
import struct

import numpy as np

_DT = {"F32": (np.float32, 4), "F16": (np.float16, 2), "BF16": (np.uint16, 2)}


def read_header(path: str):
    with open(path, "rb") as f:
        (n,) = struct.unpack("<Q", f.read(8))
        header = json.loads(f.read(n))
    return header, 8 + n


def iter_safetensors(path: str):
    """Yields (name, array) -- array is float32/float16, or a (uint16 array, 'bf16') pair for BF16 tensors."""
    header, start = read_header(path)
    mm = np.memmap(path, dtype=np.uint8, mode="r", offset=start)
    for name, info in header.items():
        if name == "__metadata__":
            continue
        if info["dtype"] not in _DT:
            raise ValueError(f"{path}: tensor {name} has unsupported dtype {info['dtype']}")
        dt, _ = _DT[info["dtype"]]
        b, e = info["data_offsets"]
        arr = np.frombuffer(mm[b:e], dtype=dt).reshape(info["shape"])
        yield name, ((arr, "bf16") if info["dtype"] == "BF16" else arr)


def checkpoint_files(model_dir: str):
    idx = os.path.join(model_dir, "model.safetensors.index.json")
    if os.path.exists(idx):
        with open(idx, encoding="utf-8") as f:
            names = sorted(set(json.load(f)["weight_map"].values()))
        return [os.path.join(model_dir, n) for n in names]
    files = sorted(glob.glob(os.path.join(model_dir, "*.safetensors")))
    if not files:
        raise FileNotFoundError(f"no .safetensors files under {model_dir}")
    return files


def iter_checkpoint(model_dir: str):
    for f in checkpoint_files(model_dir):
        yield from iter_safetensors(f)


def write_safetensors(path: str, tensors: dict, bf16: bool = False) -> None:
    """tensors: name -> float32/float16 numpy array. With bf16=True float32 arrays are stored as BF16 (RNE)."""
    header, blobs, off = {}, [], 0
    for name, arr in tensors.items():
        arr = np.ascontiguousarray(arr)
        if bf16 and arr.dtype == np.float32:
            u = arr.view(np.uint32)
            raw = (((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)).tobytes()
            dt = "BF16"
        elif arr.dtype == np.float32:
            raw, dt = arr.tobytes(), "F32"
        elif arr.dtype == np.float16:
            raw, dt = arr.tobytes(), "F16"
        else:
            raise ValueError(f"unsupported dtype {arr.dtype}")
        header[name] = {"dtype": dt, "shape": list(arr.shape), "data_offsets": [off, off + len(raw)]}
        blobs.append(raw)
        off += len(raw)
    hj = json.dumps(header).encode()
    hj += b" " * ((8 - len(hj) % 8) % 8)
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(hj)))
        f.write(hj)
        for b in blobs:
            f.write(b)
