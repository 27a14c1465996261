"""ZoomEarthForConditionalGeneration: the `Qwen2_5_VLForConditionalGeneration` call surface used by the
reference entry points, on top of the HIP engine.

replaces: `Qwen2_5_VLForConditionalGeneration.from_pretrained(model_name, torch_dtype=torch.float16)`, `.eval()`,
`.device`, `.generation_config.{temperature,top_p,top_k}` and
`model.generate(**inputs, max_new_tokens=1024, do_sample=..., num_beams=1[, temperature])`
(/root/reference/src/eval/infer.py:109-115,147-151,160-162; src/demo.py:14-19,128).

Documented deviations: arithmetic is bf16 (the reference runs fp16 weights under a bf16 autocast
wrapper); `do_sample=True, temperature=T` draws from softmax(logits / T) with the engine's counter-based
random stream (`seed=` kwarg or `generation_config.seed`); `top_k` / `top_p` other than None (or top_k=1, which is
greedy) raise NotImplementedError, as the reference sets both to None; `num_beams` must be 1.
"""
from __future__ import annotations

import json
import os
from collections import OrderedDict
from types import SimpleNamespace

import numpy as np
import torch

from . import processor as _processor
from .checkpoint import iter_checkpoint
from .config import ModelConfig
from .engine import Engine


class ZoomEarthForConditionalGeneration:
    def __init__(self, config: ModelConfig, engine: Engine, generation_config=None):
        self.config = config
        self.engine = engine
        self.generation_config = generation_config or SimpleNamespace(
            temperature=None, top_p=None, top_k=None, repetition_penalty=1.0, do_sample=False,
            eos_token_id=list(config.eos_token_ids), pad_token_id=config.pad_token_id)
        self._vit_cache = OrderedDict()   # image key -> bf16 features
        self._chains = OrderedDict()      # slot -> (prompt ids tuple, image keys tuple)
        self._next_slot = 0
        self.reuse_prefix = True
        _processor.set_default_engine(engine)

    # ------------------------------------------------------------------ construction
    @classmethod
    def from_pretrained(cls, path: str, torch_dtype=None, device=None, max_seqs: int = 4, max_ctx: int = 4096,
                        max_patches: int = 8192, max_tile_side: int = 8192, max_prefill_rows: int = 0, broadcast=False, **kw):
        """broadcast=True under WORLD_SIZE > 1 (one rank per GPU): only rank 0 reads the safetensors, the other ranks
        receive the packed weight arena in ONE collective (accel.broadcast_engine_weights: RCCL over xGMI) -- where the
        reference has every rank read the checkpoint itself (/root/reference/src/eval/infer.py:147-151).  The seconds the
        collective took are left in `model.weight_broadcast_s`."""
        config = ModelConfig.from_pretrained(path)
        dev = 0 if device is None else (device.index or 0 if isinstance(device, torch.device) else int(device))
        if device is None and "LOCAL_RANK" in os.environ:  # (more local ranks than GPUs: ranks share GPUs, accel.Accelerator)
            dev = int(os.environ["LOCAL_RANK"]) % max(1, torch.cuda.device_count())
        engine = Engine(config, device=dev, max_seqs=max_seqs, max_ctx=max_ctx, max_patches=max_patches,
                        max_tile_side=max_tile_side, max_prefill_rows=max_prefill_rows)
        rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
        broadcast = bool(broadcast) and world > 1
        bcast_s = 0.0
        try:
            if not broadcast or rank == 0:
                synth = None
                with open(os.path.join(path, "config.json"), encoding="utf-8") as f:
                    synth = json.load(f).get("zoomearth_synthetic_weights")
                if synth is not None:
                    # a checkpoint directory WITHOUT weight files whose config.json asks for the repo's synthetic weights
                    # (tools/bench_infer_e2e.py: the 3B shape end to end through src/infer.py with no 7.5-GB file on disk)
                    engine.fill_synthetic(**synth)
                    engine.assert_ready()
                else:
                    engine.load_state_dict(iter_checkpoint(path))
            if broadcast:
                from .accel import broadcast_engine_weights
                bcast_s = broadcast_engine_weights(engine, rank, world, src=0)
                engine.assert_ready()
        except Exception:
            engine.close()
            raise
        gen = SimpleNamespace(temperature=None, top_p=None, top_k=None, repetition_penalty=1.0, do_sample=False,
                              eos_token_id=list(config.eos_token_ids), pad_token_id=config.pad_token_id)
        gp = os.path.join(path, "generation_config.json")
        if os.path.exists(gp):
            with open(gp, encoding="utf-8") as f:
                for k, v in json.load(f).items():
                    setattr(gen, k, v)
        model = cls(config, engine, gen)
        model.weight_broadcast_s = bcast_s
        return model

    @classmethod
    def from_synthetic(cls, config: ModelConfig, seed: int = 0, std: float = 0.02, matrix_gain: float = 1.0,
                       bias_std: float = 0.0, norm_jitter: float = 0.0, device: int = 0, **engine_kw):
        engine = Engine(config, device=device, **engine_kw)
        engine.fill_synthetic(seed, std, matrix_gain, bias_std, norm_jitter)
        return cls(config, engine)

    def eval(self):
        return self

    def clone_lane(self, **engine_kw):
        """A second engine on the same GPU with a copy of this model's weights (device-to-device), for a further LANE of
        question chains: its own KV cache, workspaces, scheduler thread and HIP stream, so that the prefill / ViT rounds of
        one lane overlap the decode bursts of the other (src/eval/infer.py --lanes; bench.py --lanes)."""
        e = self.engine
        kw = dict(device=e.device.index or 0, max_seqs=e.max_seqs, max_ctx=e.max_ctx, max_patches=e.max_patches,
                  max_tile_side=int(e.zcfg.max_tile_side), max_prefill_rows=int(e.zcfg.max_prefill_rows))
        kw.update(engine_kw)
        e2 = Engine(self.config, **kw)
        try:
            e2.weights_arena().copy_(e.weights_arena())
            torch.cuda.synchronize(e.device)
            e2.weights_invalidate()
            e2.assert_ready()
        except Exception:
            e2.close()
            raise
        gen = SimpleNamespace(**vars(self.generation_config))
        lane = ZoomEarthForConditionalGeneration.__new__(ZoomEarthForConditionalGeneration)
        lane.config, lane.engine, lane.generation_config = self.config, e2, gen
        lane._vit_cache, lane._chains, lane._next_slot, lane.reuse_prefix = OrderedDict(), OrderedDict(), 0, True
        lane.weight_broadcast_s = 0.0
        return lane

    @property
    def device(self):
        return self.engine.device

    # ------------------------------------------------------------------ helpers
    def _features(self, pv_rows, grid, key):
        """ViT features of one image, cached by image identity (bit-identical to recomputation)."""
        if key is not None and key in self._vit_cache:
            self._vit_cache.move_to_end(key)
            return self._vit_cache[key]
        f = self.engine.vit_forward(pv_rows.contiguous(), [grid])
        if key is not None:
            self._vit_cache[key] = f
            while len(self._vit_cache) > 8:
                self._vit_cache.popitem(last=False)
        return f

    def _pick_slot(self, ids, keys):
        """Returns (slot, reusable prefix length).  A cached chain is reusable when its prefilled prompt is a
        strict prefix of `ids` and the images inside that prefix are the same objects."""
        if self.reuse_prefix and all(k is not None for k in keys):
            for slot, (pids, pkeys) in self._chains.items():
                n = len(pids)
                if 0 < n < len(ids) and tuple(ids[:n]) == pids and tuple(keys[: len(pkeys)]) == pkeys \
                        and ids[n] != self.config.image_token_id:
                    return slot, n
        slot = self._next_slot
        self._next_slot = (self._next_slot + 1) % self.engine.max_seqs
        return slot, 0

    # ------------------------------------------------------------------ rollout scoring
    @torch.no_grad()
    def per_token_logps(self, input_ids, attention_mask=None, pixel_values=None, image_grid_thw=None,
                        image_keys=None, **kw):
        """Log-probability of every token given its prefix: f32 [B, L - 1], column t = log p(input_ids[:, t + 1]).
        Same result layout as `_get_per_token_logps(model, input_ids, attention_mask, pixel_values=...,
        image_grid_thw=...)` of the reference's GRPO trainer (src/train/RL/src/open-r1-multimodal/src/open_r1/
        trainer/grpo_trainer.py:494-504), which it calls without gradients for the old policy and the reference
        model (:660-683).  Padded positions (attention_mask 0) are skipped, their columns are 0."""
        e, cfg = self.engine, self.config
        ids_cpu = input_ids.cpu().numpy()
        mask = attention_mask.cpu().numpy().astype(bool) if attention_mask is not None else np.ones_like(ids_cpu, bool)
        grids = image_grid_thw.cpu().numpy().tolist() if image_grid_thw is not None else []
        keys = list(image_keys) if image_keys is not None else [None] * len(grids)
        rows_per = [g[0] * g[1] * g[2] for g in grids]
        offs = np.concatenate([[0], np.cumsum(rows_per)]).astype(int)
        out = torch.zeros((ids_cpu.shape[0], max(ids_cpu.shape[1] - 1, 0)), dtype=torch.float32, device=e.device)
        self._chains.clear()  # scoring uses slot 0 as scratch
        gi = 0
        for b in range(ids_cpu.shape[0]):
            valid = np.nonzero(mask[b])[0]
            ids = ids_cpu[b][valid].astype(np.int64).tolist()
            is_img = np.asarray(ids) == cfg.image_token_id
            n_img = int((is_img & ~np.concatenate([[False], is_img[:-1]])).sum())
            my = list(range(gi, gi + n_img))
            gi += n_img
            if gi > len(grids):
                raise ValueError("Image features and image tokens do not match")
            if len(ids) < 2:
                continue
            feats = [self._features(pixel_values[offs[i]:offs[i + 1]], grids[i], keys[i]) for i in my]
            emb = (torch.cat(feats) if len(feats) > 1 else feats[0]) if feats else None
            pos, delta = e.rope_index(ids, [grids[i] for i in my])
            e.seq_reset(0)
            lp = e.score(0, ids, emb, pos, delta)
            out[b, torch.as_tensor(valid[1:] - 1, device=e.device)] = lp
        return out.to(input_ids.device) if input_ids.device.type != "cpu" else out.cpu()

    # ------------------------------------------------------------------ generate
    @torch.no_grad()
    def generate(self, input_ids=None, attention_mask=None, pixel_values=None, image_grid_thw=None,
                 mm_token_type_ids=None, image_keys=None, max_new_tokens: int = 20, do_sample: bool = False,
                 num_beams: int = 1, temperature=None, top_p=None, top_k=None, repetition_penalty=None,
                 ignore_eos: bool = False, **kw):
        if num_beams != 1:
            raise NotImplementedError("beam search is not part of the ZoomEarth path (num_beams=1 everywhere)")
        e, cfg = self.engine, self.config
        ids_cpu = input_ids.cpu().numpy()
        mask = attention_mask.cpu().numpy().astype(bool) if attention_mask is not None else np.ones_like(ids_cpu, bool)
        grids = image_grid_thw.cpu().numpy().tolist() if image_grid_thw is not None else []
        keys = list(image_keys) if image_keys is not None else [None] * len(grids)
        rows_per = [g[0] * g[1] * g[2] for g in grids]
        offs = np.concatenate([[0], np.cumsum(rows_per)]).astype(int)
        pen = repetition_penalty if repetition_penalty is not None else getattr(self.generation_config, "repetition_penalty", 1.0) or 1.0
        gc = self.generation_config
        top_k = top_k if top_k is not None else getattr(gc, "top_k", None)
        top_p = top_p if top_p is not None else getattr(gc, "top_p", None)
        temperature = temperature if temperature is not None else getattr(gc, "temperature", None)
        if do_sample and top_k == 1:
            do_sample = False  # a one-token nucleus is the arg-max
        if do_sample and ((top_k not in (None, 0)) or (top_p not in (None, 1.0))):
            raise NotImplementedError("top_k / top_p filtering is not part of the ZoomEarth path "
                                      "(src/eval/infer.py sets both to None)")
        if do_sample and temperature is None:
            temperature = 1.0
        sample_kw = dict(do_sample=bool(do_sample), temperature=float(temperature or 1.0),
                         seed=int(kw.get("seed", getattr(gc, "seed", 0) or 0)))
        gi = 0
        outs = []
        nrows = ids_cpu.shape[0]
        batched = nrows > 1
        if batched and nrows > e.max_seqs:
            raise ValueError(f"batch of {nrows} rows needs max_seqs >= {nrows} (engine has {e.max_seqs})")
        if batched:  # every row gets its own chain slot; the decode steps then run as one batch
            self._chains.clear()
            self._next_slot = 0
            e.set_decode_regime(-1)  # (a scheduler may have pinned the family to its own capacity)
        slots = []
        pending = []
        for b in range(nrows):
            ids = ids_cpu[b][mask[b]].astype(np.int64).tolist()
            is_img = np.asarray(ids) == cfg.image_token_id
            starts = is_img & ~np.concatenate([[False], is_img[:-1]])
            n_img = int(starts.sum())
            my = list(range(gi, gi + n_img))
            gi += n_img
            if gi > len(grids):
                raise ValueError("Image features and image tokens do not match")
            my_grids = [grids[i] for i in my]
            my_keys = [keys[i] for i in my]
            slot, reuse = (b, 0) if batched else self._pick_slot(ids, my_keys)
            pre = np.asarray(ids[:reuse]) == cfg.image_token_id
            n_img_reused = int((pre & ~np.concatenate([[False], pre[:-1]])).sum()) if reuse else 0
            feats = [self._features(pixel_values[offs[i]:offs[i + 1]], grids[i], keys[i]) for i in my[n_img_reused:]]
            emb = (torch.cat(feats) if len(feats) > 1 else feats[0]) if feats else None
            pos, delta = e.rope_index(ids, my_grids)
            if not batched:
                self._chains.pop(slot, None)  # re-registered only after its prefill succeeded
            if reuse:
                e.seq_truncate(slot, reuse)  # also clears the chain's seen-set
            else:
                e.seq_reset(slot)
            if batched:  # rows of several chains share every GEMM of the prefill (engine.prefill_batch)
                pending.append((slot, ids[reuse:], emb, pos[:, reuse:], delta, ids))
            else:
                e.prefill(slot, ids[reuse:], emb, pos[:, reuse:], delta, want_logits=False)
                if pen != 1.0:
                    e.mark_seen(slot, ids)
            if not batched:
                self._chains[slot] = (tuple(ids), tuple(my_keys))
                self._chains.move_to_end(slot)
                outs.append(e.generate(slot, max_new_tokens, repetition_penalty=pen, ignore_eos=ignore_eos, **sample_kw))
            slots.append(slot)
        if batched:
            group, rows = [], 0
            for item in pending + [None]:
                if group and (item is None or rows + len(item[1]) > e.max_prefill_rows):
                    e.prefill_batch([g[0] for g in group], [g[1] for g in group], [g[2] for g in group],
                                    [g[3] for g in group], [g[4] for g in group])
                    group, rows = [], 0
                if item is not None:
                    group.append(item)
                    rows += len(item[1])
            if pen != 1.0:
                for slot, _, _, _, _, ids in pending:
                    e.mark_seen(slot, ids)
            outs = e.generate_batch(slots, max_new_tokens, repetition_penalty=pen, ignore_eos=ignore_eos, **sample_kw)
        width = max(len(t) for t in outs)
        pad = cfg.pad_token_id
        res = torch.full((ids_cpu.shape[0], ids_cpu.shape[1] + width), pad, dtype=torch.long)
        res[:, : ids_cpu.shape[1]] = torch.from_numpy(ids_cpu)
        for b, t in enumerate(outs):
            res[b, ids_cpu.shape[1]: ids_cpu.shape[1] + len(t)] = torch.tensor(t, dtype=torch.long)
        return res.to(input_ids.device)
