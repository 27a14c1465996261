"""Python handle on one ze_engine (one process per GPU).

torch tensors are used only as device-memory containers (allocation, H2D/D2H copies, stream handles);
every computation goes through the C ABI of libzoomearth_hip.so.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Iterable, Sequence

import numpy as np
import torch

from . import _lib
from .config import ModelConfig

_NP2ZE = {np.dtype(np.float32): _lib.ZE_F32, np.dtype(np.float16): _lib.ZE_F16}


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _i32(a):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.int32))
    return a, a.ctypes.data_as(C.POINTER(C.c_int32))


class Engine:
    def __init__(self, config: ModelConfig, device: int = 0, max_seqs: int = 4, max_ctx: int = 4096,
                 max_patches: int = 8192, max_tile_side: int = 8192, max_prefill_rows: int = 0):
        if not torch.cuda.is_available():
            raise RuntimeError("zoomearth_amd needs a ROCm GPU (MI355X); there is no CPU fallback")
        self.lib = _lib.lib()
        self.config = config
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self.zcfg = self._make_zcfg(config, max_seqs, max_ctx, max_patches, max_tile_side)
        self.zcfg.max_prefill_rows = int(max_prefill_rows)
        self.max_prefill_rows = max(int(max_prefill_rows), int(max_ctx))
        h = C.c_void_p()
        _lib.check(self.lib.ze_engine_create(C.byref(self.zcfg), device, C.byref(h)))
        self.h = h
        self.max_seqs, self.max_ctx, self.max_patches = max_seqs, max_ctx, max_patches
        self.patch_dim = (config.vision.in_channels * config.vision.temporal_patch_size
                          * config.vision.patch_size ** 2)

    # ------------------------------------------------------------------ plumbing
    @staticmethod
    def _make_zcfg(c: ModelConfig, max_seqs, max_ctx, max_patches, max_tile_side) -> _lib.ZeConfig:
        z = _lib.ZeConfig()
        v, t = c.vision, c.text
        z.vit_depth, z.vit_hidden, z.vit_heads = v.depth, v.hidden_size, v.num_heads
        z.vit_intermediate, z.vit_out_hidden = v.intermediate_size, v.out_hidden_size
        z.patch_size, z.temporal_patch_size, z.spatial_merge_size = v.patch_size, v.temporal_patch_size, v.spatial_merge_size
        z.window_size, z.in_channels = v.window_size, v.in_channels
        z.n_fullatt = len(v.fullatt_block_indexes)
        for i, b in enumerate(v.fullatt_block_indexes):
            z.fullatt_block_indexes[i] = b
        z.hidden, z.layers, z.heads, z.kv_heads = t.hidden_size, t.num_hidden_layers, t.num_attention_heads, t.num_key_value_heads
        z.intermediate, z.vocab = t.intermediate_size, t.vocab_size
        z.rms_eps, z.rope_theta = t.rms_norm_eps, t.rope_theta
        for i in range(3):
            z.mrope_section[i] = t.mrope_section[i]
        z.tie_word_embeddings = int(t.tie_word_embeddings)
        z.image_token_id, z.vision_start_token_id = c.image_token_id, c.vision_start_token_id
        z.vision_end_token_id, z.pad_token_id = c.vision_end_token_id, c.pad_token_id
        z.n_eos = len(c.eos_token_ids)
        for i, e in enumerate(c.eos_token_ids):
            z.eos_token_ids[i] = e
        z.max_seqs, z.max_ctx, z.max_patches, z.max_tile_side = max_seqs, max_ctx, max_patches, max_tile_side
        return z

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _use(self, *tensors):
        """The kernels about to be enqueued on the CURRENT stream read these tensors.  A tensor allocated on another stream (a
        tile uploaded or a view resized on the caller's stream, then cropped / encoded on the scheduler's admission stream)
        goes back to ITS stream's pool when the last reference drops, and the caching allocator would hand the block out again
        -- e.g. to the next tile's upload -- while this stream's reads are still queued: record_stream keeps the block until
        they are over (a no-op for a tensor of this stream)."""
        st = torch.cuda.current_stream(self.device)
        for t in tensors:
            if t is not None and t.is_cuda:
                t.record_stream(st)

    def _check(self, code):
        return _lib.check(code, self.h)

    def close(self):
        if getattr(self, "h", None):
            self.lib.ze_engine_destroy(self.h)
            self.h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        self._check(self.lib.ze_sync(self.h, self._stream()))

    # ------------------------------------------------------------------ weights
    def load_weight(self, name: str, array) -> None:
        """array: numpy float32/float16, or a (uint16 view, 'bf16') pair for raw bf16 bits."""
        if isinstance(array, tuple):
            arr, dt = np.ascontiguousarray(array[0]), _lib.ZE_BF16
        else:
            arr = np.ascontiguousarray(array)
            if arr.dtype not in _NP2ZE:
                arr = arr.astype(np.float32)
            dt = _NP2ZE[arr.dtype]
        shape = (C.c_int64 * arr.ndim)(*arr.shape)
        self._check(self.lib.ze_load_weight(self.h, name.encode(), dt, arr.ndim, shape,
                                            arr.ctypes.data_as(C.c_void_p)))

    def load_state_dict(self, items: Iterable) -> None:
        for name, arr in items:
            self.load_weight(name, arr)
        self.assert_ready()

    def fill_synthetic(self, seed: int = 0, std: float = 0.02, matrix_gain: float = 1.0, bias_std: float = 0.0,
                       norm_jitter: float = 0.0) -> None:
        self._check(self.lib.ze_weights_fill_synthetic(self.h, seed, std, matrix_gain, bias_std, norm_jitter))

    def assert_ready(self) -> None:
        n = self.lib.ze_weights_missing(self.h)
        if n != 0:
            raise RuntimeError(self.lib.ze_last_error(self.h).decode())

    def weights_arena(self) -> torch.Tensor:
        """uint8 view of the packed weight arena (no copy), e.g. for one torch.distributed.broadcast over RCCL.  Reading it
        has no side effect; whoever WRITES through it calls weights_invalidate() afterwards (accel.broadcast_engine_weights
        and Accelerator.broadcast_weights do): the fragment / FP8 copies and the captured decode graphs derive from it."""
        p, n = C.c_void_p(), C.c_size_t()
        self._check(self.lib.ze_weights_arena(self.h, C.byref(p), C.byref(n)))

        class _Arena:
            __cuda_array_interface__ = {"shape": (n.value,), "typestr": "|u1", "data": (p.value, False), "version": 2}

        holder = _Arena()
        t = torch.as_tensor(holder, device=self.device)
        t._ze_keepalive = self  # the engine owns the memory
        return t

    def weights_invalidate(self) -> None:
        """After writing through weights_arena() (broadcast, weight refresh): derived copies and graphs are rebuilt."""
        self._check(self.lib.ze_weights_invalidate(self.h))

    # ------------------------------------------------------------------ front-end
    def crop_resize(self, tile: torch.Tensor, box: Sequence[int], out_wh: Sequence[int]) -> torch.Tensor:
        """PIL `tile.crop(box).resize(out_wh, BICUBIC)` on a device u8 [H, W, 3] tensor."""
        assert tile.dtype == torch.uint8 and tile.is_cuda and tile.is_contiguous() and tile.shape[-1] == 3
        h, w = int(tile.shape[0]), int(tile.shape[1])
        ow, oh = int(out_wh[0]), int(out_wh[1])
        out = torch.empty((oh, ow, 3), dtype=torch.uint8, device=self.device)
        b = (C.c_int32 * 4)(*[int(v) for v in box])
        self._use(tile)
        self._check(self.lib.ze_op_crop_resize(self.h, _ptr(tile), h, w, b, _ptr(out), oh, ow, self._stream()))
        return out

    def smart_resize(self, h: int, w: int, min_pixels: int, max_pixels: int):
        f = self.config.vision.patch_size * self.config.vision.spatial_merge_size
        oh, ow = C.c_int(), C.c_int()
        _lib.check(self.lib.ze_smart_resize(h, w, f, min_pixels, max_pixels, C.byref(oh), C.byref(ow)))
        return oh.value, ow.value

    def patchify(self, img: torch.Tensor) -> torch.Tensor:
        h, w = int(img.shape[0]), int(img.shape[1])
        p = self.config.vision.patch_size
        out = torch.empty(((h // p) * (w // p), self.patch_dim), dtype=torch.float32, device=self.device)
        self._use(img)
        self._check(self.lib.ze_op_patchify(self.h, _ptr(img), h, w, _ptr(out), self._stream()))
        return out

    def preprocess_image(self, img: torch.Tensor, min_pixels: int = 3136, max_pixels: int = 128 * 128 * 28 * 28):
        """u8 [H, W, 3] device image -> (pixel_values f32 [N, 1176] device, (1, gh, gw))."""
        assert img.dtype == torch.uint8 and img.is_cuda and img.is_contiguous()
        h, w = int(img.shape[0]), int(img.shape[1])
        rh, rw = self.smart_resize(h, w, min_pixels, max_pixels)
        p = self.config.vision.patch_size
        rows = (rh // p) * (rw // p)
        out = torch.empty((rows, self.patch_dim), dtype=torch.float32, device=self.device)
        grid = (C.c_int32 * 3)()
        self._use(img)
        self._check(self.lib.ze_preprocess_image(self.h, _ptr(img), h, w, min_pixels, max_pixels, _ptr(out), rows,
                                                 grid, self._stream()))
        return out, (int(grid[0]), int(grid[1]), int(grid[2]))

    # ------------------------------------------------------------------ index helpers
    def window_index(self, grids):
        g, gp = _i32(np.asarray(grids).reshape(-1, 3))
        n = int((g[:, 0] * g[:, 1] * g[:, 2]).sum()) // (self.config.vision.spatial_merge_size ** 2)
        wi = np.zeros(n, dtype=np.int64)
        cu = np.zeros(n + 2, dtype=np.int32)
        ncu = C.c_int()
        _lib.check(self.lib.ze_vision_window_index(C.byref(self.zcfg), gp, len(g), wi.ctypes.data_as(C.POINTER(C.c_int64)),
                                                   cu.ctypes.data_as(C.POINTER(C.c_int32)), len(cu), C.byref(ncu)))
        return wi, cu[: ncu.value]

    def rope_index(self, input_ids, grids):
        ids, ip = _i32(input_ids)
        g, gp = _i32(np.asarray(grids).reshape(-1, 3)) if len(grids) else (np.zeros((0, 3), np.int32), None)
        pos = np.zeros((3, len(ids)), dtype=np.int32)
        delta = C.c_int32()
        _lib.check(self.lib.ze_rope_index(C.byref(self.zcfg), ip, len(ids), gp, len(g),
                                          pos.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(delta)))
        return pos, int(delta.value)

    # ------------------------------------------------------------------ model
    def vit_forward(self, pixel_values: torch.Tensor, grids) -> torch.Tensor:
        assert pixel_values.dtype == torch.float32 and pixel_values.is_cuda and pixel_values.is_contiguous()
        g, gp = _i32(np.asarray(grids).reshape(-1, 3))
        n = int((g[:, 0] * g[:, 1] * g[:, 2]).sum())
        assert pixel_values.shape[0] == n, (pixel_values.shape, n)
        mu = self.config.vision.spatial_merge_size ** 2
        out = torch.empty((n // mu, self.config.vision.out_hidden_size), dtype=torch.bfloat16, device=self.device)
        self._use(pixel_values)
        self._check(self.lib.ze_vit_forward(self.h, _ptr(pixel_values), gp, len(g), _ptr(out), self._stream()))
        return out

    def seq_reset(self, seq: int):
        self._check(self.lib.ze_seq_reset(self.h, seq, self._stream()))

    def seq_retire(self, seq: int, stream=None):
        """The chain in `seq` is over: chains that read their prompt prefix from its cache move to another holder of the same
        rows.  `stream`: the stream the decode steps run on (default: the current one)."""
        st = C.c_void_p(stream.cuda_stream) if stream is not None else self._stream()
        self._check(self.lib.ze_seq_retire(self.h, seq, st))

    def seq_set_prefix_hint(self, seq: int, src: int, rows: int):
        """Declares that the first `rows` cached tokens of `seq` equal those of chain `src` (rows = 0 clears)."""
        self._check(self.lib.ze_seq_set_prefix_hint(self.h, seq, src, rows, self._stream()))

    def seq_set_split(self, seq: int, rows: int):
        """Measurement / tests: declare the chain's split row by hand (prefilling an image block sets it by itself)."""
        self._check(self.lib.ze_seq_set_split(self.h, int(seq), int(rows), self._stream()))

    def seq_prefix_hint(self, seq: int):
        """(source chain, rows): the decode attention reads the first `rows` cached tokens of `seq` from the source's cache
        (the same bits; one copy per tile in flight); (seq, 0) when it reads its own."""
        h = int(self.lib.ze_seq_prefix_hint(self.h, seq))
        if h < 0:
            self._check(h)
        return (h >> 16, h & 0xffff) if h else (seq, 0)

    def seq_truncate(self, seq: int, keep: int):
        self._check(self.lib.ze_seq_truncate(self.h, seq, keep, self._stream()))

    def seq_copy_prefix(self, dst: int, src: int, n_tokens: int):
        """Chain `dst` becomes the first n_tokens cached tokens of chain `src` (shared prompt prefix: K/V rows copied)."""
        self._check(self.lib.ze_seq_copy_prefix(self.h, dst, src, n_tokens, self._stream()))

    def seq_len(self, seq: int) -> int:
        return self._check(self.lib.ze_seq_len(self.h, seq))

    def mark_seen(self, seq: int, ids):
        a, p = _i32(ids)
        self._check(self.lib.ze_seq_mark_seen(self.h, seq, p, len(a), self._stream()))

    def mark_seen_batch(self, seqs, ids_list):
        """mark_seen for the chains of a prefill pass at once (one copy, one launch)."""
        if not len(seqs):
            return
        sq, sp = _i32(seqs)
        cnt, cp = _i32([len(x) for x in ids_list])
        flat, fp = _i32(np.concatenate([np.asarray(x, dtype=np.int32) for x in ids_list]) if len(ids_list) else [])
        self._check(self.lib.ze_seq_mark_seen_batch(self.h, sp, cp, len(sq), fp, self._stream()))

    def prefill(self, seq: int, new_ids, image_embeds, position_ids, rope_delta: int, want_logits: bool = True):
        """Appends `new_ids` to chain `seq`. position_ids: int32 [3, len(new_ids)]."""
        ids, ip = _i32(new_ids)
        pos, pp = _i32(position_ids)
        assert pos.shape == (3, len(ids))
        n_img = 0 if image_embeds is None else int(image_embeds.shape[0])
        if image_embeds is not None:
            assert image_embeds.dtype == torch.bfloat16 and image_embeds.is_contiguous()
        logits = torch.empty(self.config.text.vocab_size, dtype=torch.float32, device=self.device) if want_logits else None
        self._use(image_embeds)
        self._check(self.lib.ze_prefill(self.h, seq, ip, len(ids), _ptr(image_embeds), n_img, pp, rope_delta,
                                        _ptr(logits), self._stream()))
        return logits

    def score(self, seq: int, new_ids, image_embeds, position_ids, rope_delta: int):
        """prefill() plus the log-probability of every next id: returns f32 [len(new_ids) - 1] with
        out[t] = log_softmax(logits[t])[new_ids[t + 1]] (ze_score; replaces _get_per_token_logps of the reference's
        GRPO trainer)."""
        ids, ip = _i32(new_ids)
        pos, pp = _i32(position_ids)
        assert pos.shape == (3, len(ids))
        n_img = 0 if image_embeds is None else int(image_embeds.shape[0])
        if image_embeds is not None:
            assert image_embeds.dtype == torch.bfloat16 and image_embeds.is_contiguous()
        out = torch.empty(max(len(ids) - 1, 0), dtype=torch.float32, device=self.device)
        self._check(self.lib.ze_score(self.h, seq, ip, len(ids), _ptr(image_embeds), n_img, pp, rope_delta,
                                      _ptr(out) if len(ids) > 1 else _ptr(torch.empty(1, dtype=torch.float32, device=self.device)),
                                      self._stream()))
        return out

    def op_token_logprob(self, logits: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
        """log_softmax(logits, -1).gather(targets) in fp32 for bf16 logits [rows, vocab] (row stride % 8 == 0)."""
        assert logits.dtype == torch.bfloat16 and logits.dim() == 2 and logits.stride(1) == 1
        assert targets.dtype == torch.int32 and targets.is_contiguous()
        out = torch.empty(logits.shape[0], dtype=torch.float32, device=self.device)
        self._check(self.lib.ze_op_token_logprob(self.h, _ptr(logits), logits.shape[0], logits.shape[1],
                                                 logits.stride(0), _ptr(targets), _ptr(out), self._stream()))
        return out

    def prefill_batch(self, seqs, ids_list, embeds_list, pos_list, deltas):
        """One prefill pass for several chains (rows of all chains share every GEMM).  Per chain i: ids_list[i] (new
        token ids), embeds_list[i] (bf16 [rows, hidden] or None), pos_list[i] (int32 [3, len]), deltas[i].  Each
        chain's result is bit-identical to prefill() of that chain alone."""
        sq, sp = _i32(seqs)
        lens, lp = _i32([len(x) for x in ids_list])
        ids, ip = _i32(np.concatenate([np.asarray(x, dtype=np.int32) for x in ids_list]))
        pos = np.concatenate([np.asarray(p, dtype=np.int32).reshape(3, -1) for p in pos_list], axis=1)
        pos, pp = _i32(np.ascontiguousarray(pos))
        embs = [x for x in embeds_list if x is not None and x.shape[0] > 0]
        nrows, nrp = _i32([0 if x is None else int(x.shape[0]) for x in embeds_list])
        emb = (torch.cat(embs) if len(embs) > 1 else embs[0]).contiguous() if embs else None
        if emb is not None:
            assert emb.dtype == torch.bfloat16
        dl, dp = _i32(deltas)
        self._use(emb, *embs)
        self._check(self.lib.ze_prefill_batch(self.h, sp, len(sq), lp, ip, _ptr(emb), nrp, pp, dp, self._stream()))

    def decode_step(self, seq: int, token: int = -1, want_logits: bool = True):
        logits = torch.empty(self.config.text.vocab_size, dtype=torch.float32, device=self.device) if want_logits else None
        self._check(self.lib.ze_decode_step(self.h, seq, token, _ptr(logits), self._stream()))
        return logits

    @staticmethod
    def _gen_params(max_new_tokens, repetition_penalty, ignore_eos, use_graph, sync_every, do_sample, temperature, seed):
        if do_sample and not (temperature and temperature > 0):
            raise ValueError("`temperature` has to be a strictly positive float when sampling")  # as HF raises
        if os.environ.get("ZE_NO_GRAPH") == "1":  # debugging aid: every decode step launched eagerly
            use_graph = False
        return _lib.ZeGenParams(max_new_tokens, repetition_penalty, int(ignore_eos), int(use_graph), sync_every,
                                int(bool(do_sample)), float(temperature or 0.0), int(seed) & (2 ** 64 - 1))

    def generate(self, seq: int, max_new_tokens: int, repetition_penalty: float = 1.0, ignore_eos: bool = False,
                 use_graph: bool = True, sync_every: int = 16, do_sample: bool = False, temperature: float = 1.0,
                 seed: int = 0):
        p = self._gen_params(max_new_tokens, repetition_penalty, ignore_eos, use_graph, sync_every, do_sample,
                             temperature, seed)
        out = (C.c_int32 * max(max_new_tokens, 1))()
        n = C.c_int()
        self._check(self.lib.ze_generate(self.h, seq, C.byref(p), out, C.byref(n), self._stream()))
        return [int(out[i]) for i in range(n.value)]

    def set_decode_regime(self, regime: int = -1) -> int:
        """Kernel family of the batched decode step: 0 fragment kernels (<= 64 chains per step), 1 row streaming (any
        count), -1 by capacity (max_seqs > 64 -> 1).  Returns the family in force."""
        return self._check(self.lib.ze_set_decode_regime(self.h, int(regime)))

    def decode_batch(self, seqs, tokens=None, want_logits: bool = True):
        sq, sp = _i32(seqs)
        tk, tp = _i32(tokens) if tokens is not None else (None, None)
        logits = torch.empty((len(sq), self.config.text.vocab_size), dtype=torch.float32, device=self.device) if want_logits else None
        self._check(self.lib.ze_decode_batch(self.h, sp, len(sq), tp, _ptr(logits), self._stream()))
        return logits

    def generate_batch(self, seqs, max_new_tokens: int, repetition_penalty: float = 1.0, ignore_eos: bool = False,
                       sync_every: int = 16, use_graph: bool = True, do_sample: bool = False, temperature: float = 1.0,
                       seed: int = 0):
        """Generation for several prefilled chains at once; returns one token list per chain."""
        sq, sp = _i32(seqs)
        p = self._gen_params(max_new_tokens, repetition_penalty, ignore_eos, use_graph, sync_every, do_sample,
                             temperature, seed)
        out = (C.c_int32 * (len(sq) * max_new_tokens))()
        n_out = (C.c_int32 * len(sq))()
        self._check(self.lib.ze_generate_batch(self.h, sp, len(sq), C.byref(p), out, n_out, self._stream()))
        return [[int(out[i * max_new_tokens + t]) for t in range(n_out[i])] for i in range(len(sq))]

    # ------------------------------------------------------------------ continuous batching
    def chain_begin(self, seq: int, params, sample_stream: int = 0):
        """First token of a prefilled chain (from the logits its prefill left); `params` from gen_params()."""
        self._check(self.lib.ze_chain_begin(self.h, seq, C.byref(params), int(sample_stream), self._stream()))

    def gen_params(self, repetition_penalty: float = 1.0, ignore_eos: bool = False, use_graph: bool = True,
                   do_sample: bool = False, temperature: float = 1.0, seed: int = 0):
        return self._gen_params(0, repetition_penalty, ignore_eos, use_graph, 1, do_sample, temperature, seed)

    def decode_burst(self, seqs, steps: int, params):
        """`steps` sampled decode steps for the live chains `seqs`; returns (steps run, n_generated[], finished[])."""
        sq, sp = _i32(seqs)
        ng = (C.c_int32 * len(sq))()
        fin = (C.c_int32 * len(sq))()
        ran = self._check(self.lib.ze_decode_burst(self.h, sp, len(sq), int(steps), C.byref(params), ng, fin,
                                                   self._stream()))
        return ran, list(ng), [bool(f) for f in fin]

    def decode_burst_begin(self, seqs, steps: int, params):
        """First half of decode_burst: enqueues the steps on the current stream and returns at once (steps enqueued).  The
        caller may now enqueue a ViT / prefill round for OTHER chains on another stream; decode_burst_end collects."""
        sq, sp = _i32(seqs)
        return self._check(self.lib.ze_decode_burst_begin(self.h, sp, len(sq), int(steps), C.byref(params), self._stream()))

    def decode_burst_end(self, seqs):
        """Waits for the burst begun on the current stream; returns (n_generated[], finished[])."""
        sq, sp = _i32(seqs)
        ng = (C.c_int32 * len(sq))()
        fin = (C.c_int32 * len(sq))()
        self._check(self.lib.ze_decode_burst_end(self.h, sp, len(sq), ng, fin, self._stream()))
        return list(ng), [bool(f) for f in fin]

    def chain_tokens(self, seq: int, capacity: int = 0, stream=None):
        """The tokens chain `seq` has generated.  `stream`: the stream its decode steps ran on (default: the current one) --
        the call copies on it and waits for it, so a scheduler inside its admission stream's context names the decode stream
        (waiting for the admission stream would serialise the next burst behind the prefill pass in flight)."""
        cap = int(capacity) if capacity else self.max_ctx
        out = (C.c_int32 * max(cap, 1))()
        n = C.c_int()
        st = C.c_void_p(stream.cuda_stream) if stream is not None else self._stream()
        self._check(self.lib.ze_chain_tokens(self.h, seq, out, cap, C.byref(n), st))
        return [int(out[i]) for i in range(n.value)]

    def chain_tokens_batch(self, seqs, capacity: int = 0, stream=None):
        """chain_tokens for several chains in one device -> host copy and one wait; returns a list of id lists."""
        if not len(seqs):
            return []
        cap = max(1, min(int(capacity) if capacity else self.max_ctx, self.max_ctx))
        sq, sp = _i32(seqs)
        out = np.empty((len(sq), cap), dtype=np.int32)
        n = (C.c_int32 * len(sq))()
        st = C.c_void_p(stream.cuda_stream) if stream is not None else self._stream()
        self._check(self.lib.ze_chain_tokens_batch(self.h, sp, len(sq), out.ctypes.data_as(C.POINTER(C.c_int32)), cap, n, st))
        return [out[i, :n[i]].tolist() for i in range(len(sq))]

    def tile_upload(self, host_rgb) -> torch.Tensor:
        """Decoded RGB u8 [H, W, 3] host array / tensor (ideally pinned) -> device tensor (ze_tile_upload)."""
        t = host_rgb if isinstance(host_rgb, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(host_rgb, dtype=np.uint8))
        assert t.dtype == torch.uint8 and t.dim() == 3 and t.shape[2] == 3 and t.is_contiguous()
        out = torch.empty(t.shape, dtype=torch.uint8, device=self.device)
        self._check(self.lib.ze_tile_upload(self.h, C.c_void_p(t.data_ptr()), int(t.shape[0]), int(t.shape[1]), _ptr(out),
                                            self._stream()))
        out._ze_host_keepalive = t  # the async copy reads the host buffer until the stream reaches it
        return out

    def sample_greedy(self, seq: int, logits: torch.Tensor, repetition_penalty: float = 1.0) -> int:
        tok = C.c_int32()
        self._check(self.lib.ze_op_sample_greedy(self.h, seq, _ptr(logits), repetition_penalty, C.byref(tok),
                                                 self._stream()))
        return int(tok.value)

    def sample_temperature(self, seq: int, logits: torch.Tensor, temperature: float, seed: int, index: int = 0,
                           repetition_penalty: float = 1.0) -> int:
        tok = C.c_int32()
        self._check(self.lib.ze_op_sample_temperature(self.h, seq, _ptr(logits), repetition_penalty, temperature,
                                                      int(seed) & (2 ** 64 - 1), index, C.byref(tok), self._stream()))
        return int(tok.value)

    def quantize_fp8(self):
        """Switch the decoder's linear layers to FP8 (E4M3, per-row power-of-two scales): see ze_weights_quantize_fp8."""
        self._check(self.lib.ze_weights_quantize_fp8(self.h, self._stream()))

    def set_fp8_activations(self, on: bool = True):
        """FP8 x FP8 batched decode (qkv and gate/up inputs quantised per row): see ze_set_fp8_activations.  Needs
        quantize_fp8() first; a weight change switches it off again."""
        self._check(self.lib.ze_set_fp8_activations(self.h, 1 if on else 0))

    def op_linear_mx(self, a8: torch.Tensor, sa: torch.Tensor, w8: torch.Tensor, sw: torch.Tensor, bias=None, swiglu: bool = False):
        """(a8 * sa[:, None]) @ (w8 * sw[:, None]).T on the block-scaled FP8 MFMA: a8 u8 [M, K], w8 u8 [N, K] (E4M3 bytes),
        sa / sw f32 powers of two per row (what op_quantize_fp8 returns) -> bf16 [M, N] ([M, N / 2] with swiglu)."""
        m, k = a8.shape
        n = w8.shape[0]
        out = torch.empty((m, n // 2 if swiglu else n), dtype=torch.bfloat16, device=self.device)
        self._check(self.lib.ze_op_linear_mx(self.h, _ptr(a8), _ptr(sa), _ptr(w8), _ptr(sw), _ptr(bias), _ptr(out), m, n, k,
                                             1 if swiglu else 0, self._stream()))
        return out

    def op_quantize_fp8(self, w: torch.Tensor):
        """w bf16 [rows, cols] on the device (overwritten with the dequantised values) -> (u8 bits, f32 scales)."""
        rows, cols = w.shape
        q = torch.empty((rows, cols), dtype=torch.uint8, device=self.device)
        sc = torch.empty(rows, dtype=torch.float32, device=self.device)
        self._check(self.lib.ze_op_quantize_fp8(self.h, _ptr(w), rows, cols, _ptr(q), _ptr(sc), self._stream()))
        return q, sc

    # ------------------------------------------------------------------ unit ops (parity tests)
    def op_window_gather(self, pixel_values: torch.Tensor, grids) -> torch.Tensor:
        g, gp = _i32(np.asarray(grids).reshape(-1, 3))
        out = torch.empty(pixel_values.shape, dtype=torch.bfloat16, device=self.device)
        self._check(self.lib.ze_op_window_gather(self.h, _ptr(pixel_values), gp, len(g), _ptr(out), self._stream()))
        return out

    def op_window_scatter(self, x: torch.Tensor, grids) -> torch.Tensor:
        g, gp = _i32(np.asarray(grids).reshape(-1, 3))
        out = torch.empty_like(x)
        self._check(self.lib.ze_op_window_scatter(self.h, _ptr(x), x.shape[1], gp, len(g), _ptr(out), self._stream()))
        return out

    def op_vision_rope(self, qkv: torch.Tensor, grids, window_order: bool = True) -> torch.Tensor:
        """In place on qkv bf16 [n, 3 * heads * 80]; returns it."""
        g, gp = _i32(np.asarray(grids).reshape(-1, 3))
        self._check(self.lib.ze_op_vision_rope(self.h, _ptr(qkv), gp, len(g), int(window_order), self._stream()))
        return qkv

    def op_embed_scatter(self, input_ids, image_embeds=None) -> torch.Tensor:
        ids, ip = _i32(input_ids)
        out = torch.empty((len(ids), self.config.text.hidden_size), dtype=torch.bfloat16, device=self.device)
        n_img = 0 if image_embeds is None else int(image_embeds.shape[0])
        self._check(self.lib.ze_op_embed_scatter(self.h, ip, len(ids), _ptr(image_embeds), n_img, _ptr(out), self._stream()))
        return out

    def op_mrope_kv(self, seq: int, layer: int, qkv: torch.Tensor, position_ids, past: int) -> torch.Tensor:
        """In place on qkv bf16 [T, (heads + 2 kv_heads) * 128] (q roped); K / V rows appended to the cache at `past`."""
        pos, pp = _i32(position_ids)
        assert pos.shape == (3, qkv.shape[0])
        self._check(self.lib.ze_op_mrope_kv(self.h, seq, layer, _ptr(qkv), qkv.shape[0], pp, int(past), self._stream()))
        return qkv

    def op_rope_kv_decode(self, seqs, layer: int, qkv: torch.Tensor) -> torch.Tensor:
        sq, sp = _i32(seqs)
        assert qkv.shape[0] == len(sq)
        self._check(self.lib.ze_op_rope_kv_decode(self.h, sp, len(sq), layer, _ptr(qkv), self._stream()))
        return qkv

    def op_attn_decode(self, seqs, layer: int, qkv: torch.Tensor) -> torch.Tensor:
        """the attention of one batched decode step, alone: qkv as op_rope_kv_decode left it -> [n, heads x 128]"""
        sq, sp = _i32(seqs)
        t = self.config.text
        assert qkv.shape[0] == len(sq) and qkv.dtype == torch.bfloat16 and qkv.is_contiguous()
        out = torch.empty((len(sq), t.hidden_size), dtype=torch.bfloat16, device=self.device)
        self._check(self.lib.ze_op_attn_decode(self.h, sp, len(sq), layer, _ptr(qkv), _ptr(out), self._stream()))
        return out

    def op_kv_read(self, seq: int, layer: int, start: int, n: int):
        t = self.config.text
        shape = (t.num_key_value_heads, n, t.hidden_size // t.num_attention_heads)
        k = torch.empty(shape, dtype=torch.bfloat16, device=self.device)
        v = torch.empty(shape, dtype=torch.bfloat16, device=self.device)
        self._check(self.lib.ze_op_kv_read(self.h, seq, layer, start, n, _ptr(k), _ptr(v), self._stream()))
        return k, v

    def op_numeric_helpers(self, x: torch.Tensor, y: torch.Tensor):
        """(bf16(x) | bf16(bf16(silu(x)) * y) << 16, pack_bf16x2(x, y)) as two int32 tensors (u32 bit patterns): x, y f32 on the device."""
        n = x.numel()
        out = torch.empty(n, dtype=torch.int32, device=self.device)
        out2 = torch.empty(n, dtype=torch.int32, device=self.device)
        self._check(self.lib.ze_op_numeric_helpers(self.h, _ptr(x), _ptr(y), _ptr(out), _ptr(out2), n, self._stream()))
        return out, out2

    def op_linear(self, a, w, bias=None, act: int = 0):
        m, k = a.shape
        n = w.shape[0]
        if act == 10:  # fp32 logits of the row-streaming regime's lm_head
            out = torch.empty((m, n), dtype=torch.float32, device=self.device)
            self._check(self.lib.ze_op_linear(self.h, _ptr(a), _ptr(w), None, _ptr(out), m, n, k, act, self._stream()))
            return out
        out = torch.empty((m, n // 2 if act in (4, 7, 9) else n), dtype=torch.bfloat16, device=self.device)
        self._check(self.lib.ze_op_linear(self.h, _ptr(a), _ptr(w), _ptr(bias), _ptr(out), m, n, k, act, self._stream()))
        return out

    def op_rmsnorm(self, x, w, eps: float):
        out = torch.empty_like(x)
        self._check(self.lib.ze_op_rmsnorm(self.h, _ptr(x), _ptr(w), _ptr(out), x.shape[0], x.shape[1], eps,
                                           self._stream()))
        return out

    def op_attention(self, q, k, v, cu_seqlens, causal: bool):
        t, heads, d = q.shape
        kvh = k.shape[1]
        out = torch.empty_like(q)
        cu, cp = _i32(cu_seqlens)
        self._check(self.lib.ze_op_attention(self.h, _ptr(q), _ptr(k), _ptr(v), _ptr(out), t, heads, kvh, d, cp,
                                             len(cu) - 1, int(causal), self._stream()))
        return out

    # ------------------------------------------------------------------ measurement
    def profile_decode_kernel(self, which: int, iters: int = 72):
        us, by = C.c_float(), C.c_double()
        self._check(self.lib.ze_profile_decode_kernel(self.h, which, iters, C.byref(us), C.byref(by), self._stream()))
        return float(us.value), float(by.value)

    def profile_batch_kernel(self, which: int, n: int, iters: int = 72):
        us, by = C.c_float(), C.c_double()
        self._check(self.lib.ze_profile_batch_kernel(self.h, which, n, iters, C.byref(us), C.byref(by), self._stream()))
        return float(us.value), float(by.value)

    def profile_prefill_kernel(self, which: int, rows: int, iters: int = 12):
        """(avg us, FLOP) of one projection of a prefill pass (0 qkv, 1 o, 2 gate/up, 3 down) at `rows` rows, on the operands the
        last pass left in the workspace and the layers' own weights."""
        us, fl = C.c_float(), C.c_double()
        self._check(self.lib.ze_profile_prefill_kernel(self.h, which, rows, iters, C.byref(us), C.byref(fl), self._stream()))
        return float(us.value), float(fl.value)

    def profile_prefill_layer(self, rows: int, layers_run: int = 36):
        """{projection: (avg us, FLOP)} of a prefill layer's four projections issued in pass order at `rows` rows."""
        us, fl = (C.c_float * 4)(), (C.c_double * 4)()
        self._check(self.lib.ze_profile_prefill_layer(self.h, rows, layers_run, us, fl, self._stream()))
        return {k: (float(us[i]), float(fl[i])) for i, k in enumerate(("qkv", "o", "gate_up", "down"))}

    def phase_timers(self, enable: bool = True, reset: bool = False):
        out = (C.c_float * 5)()
        self._check(self.lib.ze_phase_timers(self.h, int(enable), int(reset), out))
        return dict(zip(("frontend", "vit", "prefill", "decode", "sample"), [float(x) for x in out]))
