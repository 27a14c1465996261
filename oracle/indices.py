"""Oracle for the integer index builders of the Qwen2.5-VL path (SURVEY.md K4, K5, K9 segments, K14).

TEST INFRASTRUCTURE (see oracle/__init__.py).  All outputs are integers: exact match required.

Restates (transformers 5.15.0 line numbers, `HF:` = site-packages/transformers/):
  * get_vision_cu_seqlens        HF:vision_utils.py:42-65
  * get_vision_position_ids      HF:vision_utils.py:81-127
  * get_vision_window_index      HF:vision_utils.py:130-188
  * Qwen2_5_VLModel.get_vision_position_ids / get_rope_index
                                 HF:models/qwen2_5_vl/modeling_qwen2_5_vl.py:892-1058
    (4.49-era in-tree statement: /root/reference/src/train/RL/src/open-r1-multimodal/src/open_r1/
     model/modeling_qwen2_vl.py:967-1114)
  * placeholder expansion        HF:models/qwen2_5_vl/processing_qwen2_5_vl.py:59-62
"""
from __future__ import annotations

import numpy as np


def vision_cu_seqlens(grid_thw) -> np.ndarray:
    """One full-attention segment per frame: cumsum(repeat(h*w, t)) with a leading 0 (int32)."""
    lens = []
    for t, h, w in grid_thw:
        lens += [int(h) * int(w)] * int(t)
    return np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)


def vision_position_ids(grid_thw, merge: int = 2) -> np.ndarray:
    """(h, w) index of every patch, laid out block-major over merge x merge blocks: int64 [N, 2]."""
    out = []
    for t, h, w in grid_thw:
        t, h, w = int(t), int(h), int(w)
        hp, wp = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
        shp = (h // merge, merge, w // merge, merge)
        hp = hp.reshape(shp).transpose(0, 2, 1, 3).reshape(-1)
        wp = wp.reshape(shp).transpose(0, 2, 1, 3).reshape(-1)
        out.append(np.tile(np.stack([hp, wp], axis=-1), (t, 1)))
    return np.concatenate(out, axis=0).astype(np.int64)


def vision_window_index(grid_thw, merge: int = 2, window_size: int = 112, patch: int = 14):
    """Returns (window_index int64 [N/merge^2], cu_window_seqlens int32 [n_windows+1]).

    window_index permutes merge-units (groups of merge^2 patches) so each window is contiguous;
    cu_window_seqlens counts PATCHES and has consecutive duplicates removed (empty windows).
    """
    window_index = []
    cu = [0]
    base = 0
    vws = window_size // merge // patch
    unit = merge * merge
    for t, h, w in grid_thw:
        t, h, w = int(t), int(h), int(w)
        lh, lw = h // merge, w // merge
        index = np.arange(t * lh * lw).reshape(t, lh, lw)
        pad_h = vws - lh % vws
        pad_w = vws - lw % vws
        nh = (lh + pad_h) // vws
        nw = (lw + pad_w) // vws
        padded = np.full((t, lh + pad_h, lw + pad_w), -100, dtype=np.int64)
        padded[:, :lh, :lw] = index
        padded = padded.reshape(t, nh, vws, nw, vws).transpose(0, 1, 3, 2, 4).reshape(t, nh * nw, vws, vws)
        seqlens = (padded != -100).sum(axis=(2, 3)).reshape(-1)
        flat = padded.reshape(-1)
        window_index.append(flat[flat != -100] + base)
        cu_tmp = np.cumsum(seqlens) * unit + cu[-1]
        cu.extend(cu_tmp.tolist())
        base += t * lh * lw
    window_index = np.concatenate(window_index).astype(np.int64)
    cu = np.asarray(cu, dtype=np.int32)
    keep = np.concatenate([[True], cu[1:] != cu[:-1]])
    return window_index, cu[keep]


def expand_image_placeholders(ids, grid_thw, image_token_id: int, merge: int = 2):
    """Each image placeholder id -> t*h*w/merge^2 copies, consumed in image order."""
    out = []
    it = iter(grid_thw)
    for tok in ids:
        if tok == image_token_id:
            t, h, w = next(it)
            out += [image_token_id] * (int(t) * int(h) * int(w) // (merge * merge))
        else:
            out.append(int(tok))
    return out


def rope_index(input_ids: np.ndarray, image_grid_thw, image_token_id: int, attention_mask=None,
               merge: int = 2):
    """M-RoPE position ids: (int64 [3, B, L], rope_deltas int64 [B, 1]).

    Text runs count up on all three axes; an image run gets (t=start, h=start+row, w=start+col)
    and the next run starts at start + max(h, w)//merge  (images only, time_interval=1).
    Image runs are maximal runs of `image_token_id` (the reference never places two images
    back to back: each is wrapped in <|vision_start|> ... <|vision_end|>).
    """
    input_ids = np.asarray(input_ids)
    b, l = input_ids.shape
    pos = np.zeros((3, b, l), dtype=np.int64)
    deltas = []
    grids = iter(image_grid_thw) if image_grid_thw is not None else iter(())
    for bi in range(b):
        ids = input_ids[bi]
        keep = np.ones(l, dtype=bool) if attention_mask is None else np.asarray(attention_mask[bi]).astype(bool)
        cur = ids[keep]
        is_img = (cur == image_token_id).astype(np.int64)
        cols = []
        current = 0
        i = 0
        n = len(cur)
        while i < n:
            j = i
            while j < n and is_img[j] == is_img[i]:
                j += 1
            if is_img[i] == 0:
                tl = j - i
                cols.append(np.tile(np.arange(tl)[None, :], (3, 1)) + current)
                current += tl
            else:
                t, h, w = (int(v) for v in next(grids))
                lh, lw = h // merge, w // merge
                tt, hh, ww = np.meshgrid(np.arange(t), np.arange(lh) + current, np.arange(lw) + current,
                                         indexing="ij")
                v = np.stack([tt, hh, ww], axis=0).reshape(3, -1)
                v[0] += current
                cols.append(v)
                current += max(h, w) // merge
            i = j
        llm = np.concatenate(cols, axis=1).reshape(3, -1)
        pos[:, bi, keep] = llm
        deltas.append(int(llm.max()) + 1 - len(cur))
    return pos, np.asarray(deltas, dtype=np.int64).reshape(-1, 1)
