"""Continuous batching of question chains on one GPU.

replaces: the one-sample-at-a-time loop of the reference (`BATCH_SIZE = 1`, /root/reference/src/eval/infer.py:27,
173-249) and the 100 requests its "fast" path keeps in flight against a serving back-end
(/root/reference/src/eval/infer_vllm.py:244-271).  Up to `max_seqs` chains share every decode step -- the weights are
streamed once per step for all of them (`ze_decode_burst`) -- and chains join and leave BETWEEN bursts: a request that
finishes (EOS or its token budget) hands its KV slot to the next waiting request, or keeps it for a follow-up that
extends its own prompt (stage 2 of the zoom chain: the cached stage-1 prompt is reused, only the appended tokens are
prefilled, `ze_seq_truncate`).  Newcomers of one round are prefilled together (`ze_prefill_batch`) after ONE
multi-resolution ViT call over all their images.

A chain's tokens do not depend on which chains share its bursts (the batched kernels accumulate every output element in
an order that is a function of the layer shape alone -- tests/test_gpu_batch.py), and its sampling stream is the
request's own `stream_id`.  The decode step has one kernel family per engine (`Engine.decode_regime`: the fragment kernels
of an engine with at most 64 chain slots, the row-streaming kernels of a larger one), chosen by the engine's CAPACITY and
never by how many chains happen to be live, so results are the same for any `max_batch` of one engine, including 1, and
for every engine of one regime; engines of different regimes agree within bf16 rounding (tests/test_gpu_3b_shape.py).
"""
from __future__ import annotations

import contextlib
import os
import time
from collections import OrderedDict, deque
from dataclasses import dataclass, field
from typing import Any, Callable, List, Optional

import numpy as np
import torch


@dataclass
class Request:
    prompt: str
    images: list                       # flat, in prompt order (DeviceImage / PIL / ndarray)
    max_new_tokens: int = 1024
    stream_id: int = 0                 # random stream of the request when sampling (e.g. the question number)
    on_done: Optional[Callable[["Request", List[int], str], Optional["Request"]]] = None
    on_error: Optional[Callable[["Request", Exception], None]] = None
    tag: Any = None
    # filled by the scheduler
    slot: int = -1
    n_prompt: int = 0
    tokens: List[int] = field(default_factory=list)
    text: str = ""


class _Live:
    __slots__ = ("req", "ids", "keys", "produced")

    def __init__(self, req, ids, keys):
        self.req, self.ids, self.keys, self.produced = req, ids, keys, 1


_PER_CHAIN = os.environ.get("ZE_PER_CHAIN_XFER") == "1"   # measurement only: the per-chain mark_seen / chain_tokens calls (A/B runs)


class ChainScheduler:
    def __init__(self, model, processor, do_sample: bool = False, temperature=None, repetition_penalty=None, seed: int = 0,
                 burst: int = 8, max_batch: Optional[int] = None, ignore_eos: bool = False, use_graph: Optional[bool] = None,
                 feature_cache: int = 64, min_admit: int = 1, max_wait_bursts: int = 2, share_prefix: bool = True,
                 min_shared: int = 64, reuse_generated: bool = True, overlap: Optional[bool] = None, hold_below: int = 0,
                 admit_chunk_rows: int = 0):
        self.model, self.processor, self.engine = model, processor, model.engine
        gc = model.generation_config
        pen = repetition_penalty if repetition_penalty is not None else (getattr(gc, "repetition_penalty", 1.0) or 1.0)
        if do_sample and temperature is None:
            temperature = getattr(gc, "temperature", None) or 1.0
        self.penalty = float(pen)
        # Captured hipGraphs for the decode steps: yes in the fragment regime (at most 64 chains: a step of ~3.4 ms is ~330
        # launches), no in the row-streaming regime -- its steps take 6-25 ms, the host runs far ahead of the GPU, and a graph
        # is keyed by (live chains, attention grid), i.e. captured anew at almost every burst (stream: 72.1 / 72.0 questions/s
        # with graphs, 72.7 / 72.7 without)
        if use_graph is None:
            use_graph = min(int(max_batch or self.engine.max_seqs), self.engine.max_seqs) <= 64
        self.use_graph = bool(use_graph)
        self.params = self.engine.gen_params(repetition_penalty=self.penalty, ignore_eos=ignore_eos, use_graph=use_graph,
                                             do_sample=bool(do_sample), temperature=float(temperature or 1.0), seed=seed)
        # Admission hysteresis: while chains are decoding, newcomers wait until `min_admit` of them can share one ViT call
        # and one prefill pass (a 520-row prefill alone runs its GEMMs at a third of the rate of a 4000-row pass), but
        # never longer than `max_wait_bursts` bursts; with nothing decoding they are admitted at once.
        self.min_admit = max(1, int(min_admit))
        self.max_wait_bursts = max(0, int(max_wait_bursts))
        self._waited = 0
        # Hold: while an admission round still has prefill passes to run and fewer than `hold_below` chains are live, the live
        # chains do not decode -- a step costs almost the same at 60 chains as at 400, so the early members of a round wait
        # for the later ones instead of stepping alone beside the passes.  0: never hold (a server's latency setting).
        self.hold_below = max(0, int(hold_below))
        # Admission in CHUNKS (round 6): with a long queue -- the start of a run: hundreds of prompts at once -- tokenising and planning
        # every waiting request before the first pass is enqueued leaves the GPU idle for the whole of it (0.4 s for 640 prompts).  With
        # `admit_chunk_rows` > 0 one call of _admit takes requests off the queue only until that many new rows are planned (a pass's
        # worth) and the NEXT request is about another first image (the questions of a tile stay together: their shared prefix is
        # planned within one call), enqueues that pass, and prepares the next chunk while the GPU runs it.  Under `hold_below` the
        # requests still admissible count as "round in progress": the early chunks' chains wait for the later ones as before.
        self.admit_chunk_rows = max(0, int(admit_chunk_rows))
        # Shared prompt prefixes: the questions about one tile start with the same system turn and the same view's image
        # tokens (347 of the 802 tokens of a stage-1 prompt).  A fresh chain whose prompt starts like that of a chain that
        # already holds those K/V rows copies them (ze_seq_copy_prefix) and prefills only its own tail; when a round brings
        # several such chains and none exists yet, the first one's prefix is prefilled alone (pass A), copied to the
        # others, and all tails go through one pass (pass B).  Bit-identical to prefilling every prompt in full.
        self.share_prefix = bool(share_prefix)
        self.min_shared = max(1, int(min_shared))
        # Follow-ups keep the K/V rows of the tokens their predecessor generated while the new prompt repeats those ids
        # (exact in real arithmetic; in bf16 the rows come from the decode kernels instead of the prefill kernels, both
        # held to the same tolerance against the oracle).  False: only the predecessor's PROMPT rows are kept, which is
        # bit-identical to prefilling the new prompt in full.
        self.reuse_generated = bool(reuse_generated)
        self.burst = max(1, int(burst))
        self.max_batch = min(int(max_batch or self.engine.max_seqs), self.engine.max_seqs)
        # the decode step's kernel family follows the scheduler's CAPACITY, never the live count (see the module docstring)
        if hasattr(self.engine, "set_decode_regime"):
            self.engine.set_decode_regime(1 if self.max_batch > 64 else 0)
        self.waiting = deque()
        self.live = OrderedDict()          # slot -> _Live
        self.parked = {}                   # slot -> (prompt ids, image keys, generated ids with K/V rows): finished chains whose slot waits for a follow-up
        self.free = list(range(self.max_batch))[::-1]
        self._features = OrderedDict()     # image key -> ViT features (LRU)
        self._feature_cap = feature_cache
        self.stats = dict(bursts=0, steps=0, chain_steps=0, prefill_rows=0, admitted=0, vit_calls=0, shared_rows=0,
                          reused_generated_rows=0, overlapped_passes=0)
        # Overlap of the two kinds of work on one GPU: while a burst of decode steps runs on the caller's stream
        # (ze_decode_burst_begin: enqueued, not awaited), ONE prefill pass of the admission round -- its ViT call included --
        # is enqueued on a side stream; the burst is collected afterwards (ze_decode_burst_end) and the chains whose pass
        # has completed join the next burst.  The batched decode step has activation buffers of its own and the chains of
        # a pass are not in the burst, so the two streams share no state; the matrix-bound pass fills what the bandwidth-
        # and latency-bound decode steps leave idle, and the host work of admission (tokeniser, index builders, callbacks)
        # hides behind the burst.  Results are unchanged: the same kernels on the same operands.
        if overlap is None:
            overlap = hasattr(self.engine, "decode_burst_begin") and os.environ.get("ZE_OVERLAP", "1") != "0"
        self.overlap = bool(overlap)
        self._side = torch.cuda.Stream(device=self.engine.device) if (self.overlap and torch.cuda.is_available()) else None
        self._carry = []                   # (admit_chunk_rows) planned items of a partial pass, waiting for the next chunk's
        self._groups = deque()             # prefill passes of the admission round in progress, one per step
        self._round = None                 # (todo, needed) of that round: images still to encode / features to keep
        self._ready = []                   # prefilled requests waiting to join the live set
        self._pass_done = []               # (overlap) per entry of _ready: the event behind its prefill pass
        self._trace_ev = None
        if os.environ.get("ZE_SCHED_TRACE") and torch.cuda.is_available():   # measurement only: when the first passes really START on the GPU
            self._trace_ev = dict(base=torch.cuda.Event(enable_timing=True), t0=time.perf_counter(), passes=[])
            self._trace_ev["base"].record(torch.cuda.current_stream(self.engine.device))
        model._chains.clear()              # the scheduler owns every chain slot while it runs

    # ------------------------------------------------------------------ queue
    def submit(self, req: Request) -> None:
        self.waiting.append(req)

    def pending_requests(self):
        """Every request the scheduler holds, whatever its state (a caller that gives up on the engine fails them all)."""
        reqs = [l.req for l in self.live.values()] + list(self.waiting) + [r for r, _, _ in self._ready]
        reqs += [it["req"] for g in list(self._groups) + [self._carry] for it in g if it.get("final", True)]
        return reqs

    def busy(self) -> bool:
        return bool(self.waiting or self.live or self._groups or self._ready or self._carry)

    def run(self) -> None:
        while self.busy():
            self.step()
        tr = self._trace_ev
        if tr is not None and tr["passes"]:
            torch.cuda.synchronize()
            with open(os.environ["ZE_SCHED_TRACE"], "a") as f:
                for host_s, a, b in tr["passes"]:
                    f.write(f"{id(self) % 100000} {tr['t0']:.4f} G {host_s:.4f} {tr['base'].elapsed_time(a) / 1e3:.4f} {tr['base'].elapsed_time(b) / 1e3:.4f}\n")

    # ------------------------------------------------------------------ one scheduling round
    def step(self) -> None:
        tp = time.perf_counter
        t0 = tp()
        if self._side is not None:
            # what the caller put on ITS stream before submitting (the tile upload, the view's resize) is read by the
            # admission work on the side stream: order the two here, while the caller's stream holds nothing else
            self._side.wait_stream(torch.cuda.current_stream(self.engine.device))
        hold = self._round_in_progress() and len(self.live) < self.hold_below
        handle = self._burst_begin() if (self.live and self.overlap and not hold) else None
        t1 = tp()
        with self._side_stream():
            if not self._groups:
                self._admit()
            t2 = tp()
            if self._groups:
                self._run_group(self._groups.popleft())
                if handle is not None:
                    self.stats["overlapped_passes"] += 1
        t3 = tp()
        if handle is not None:
            self._burst_end(handle)
        t4 = tp()
        more = bool(self._carry or (self.admit_chunk_rows and self.waiting and self.waiting[0].slot < 0 and self.free))
        self._join_ready(wait=hold and not self._groups and not more)   # (held with nothing left to enqueue: wait for the oldest pass)
        if handle is None:   # (nothing was decoding beside the pass: the round may just have begun, or been completed)
            hold = self._round_in_progress() and len(self.live) < self.hold_below
            if self.live and not hold:
                self._burst()
        if hold:
            self.stats["held_steps"] = self.stats.get("held_steps", 0) + 1
        t5 = tp()
        st = self.stats   # host seconds of the scheduling round by part (burst_end = waiting for the GPU + retiring + the callbacks)
        for k, v in (("host_s_burst_begin", t1 - t0), ("host_s_admit", t2 - t1), ("host_s_pass", t3 - t2), ("host_s_burst_end", t4 - t3),
                     ("host_s_join", t5 - t4)):
            st[k] = st.get(k, 0.0) + v

    @staticmethod
    def _first_image_key(req):
        """What tells two requests about the same view apart from two about different ones, before anything is tokenised: the
        first image object itself (the entry points hand every question of a tile the same view object), by value where it has one."""
        if not req.images:
            return None
        im = req.images[0]
        return im if isinstance(im, (str, int, tuple)) else id(im)

    def _round_in_progress(self) -> bool:
        """Passes still to run, prefilled chains still to join -- or, when admitting in chunks, requests the next call of _admit can take."""
        if self._groups or self._ready or self._carry:
            return True
        # (FRESH requests only: a follow-up that waits for its pass is the steady state of a running stream, not a round being admitted)
        return bool(self.admit_chunk_rows and self.waiting and self.waiting[0].slot < 0 and self.free)

    def _side_stream(self):
        """Admission work (front-end, ViT, prefill, the callbacks' crops) runs on the side stream when overlapping."""
        return torch.cuda.stream(self._side) if self._side is not None else contextlib.nullcontext()

    # -- admission: waiting requests take free slots (a follow-up keeps the slot its predecessor parked)
    def _admit(self) -> None:
        e = self.engine
        # (admitting in chunks: the rest of a queue whose first chunks are already in is not "a few newcomers" -- no waiting for more)
        chunking = bool(self._carry or (self.admit_chunk_rows and self.waiting and self.waiting[0].slot < 0 and self.free))
        if self.live and self.min_admit > 1 and not chunking:
            ready = sum(1 for r in self.waiting if r.slot >= 0) + min(len(self.free), sum(1 for r in self.waiting if r.slot < 0))
            if 0 < ready < self.min_admit and self._waited < self.max_wait_bursts:
                self._waited += 1
                return
        self._waited = 0
        # tokenise / preprocess each newcomer as it leaves the queue; collect the images whose features are not cached
        prepared, todo, needed = [], OrderedDict(), set()
        if self._carry and self._round is not None:   # (items held back from the last chunk keep their images' entries)
            todo, needed = OrderedDict(self._round[0]), set(self._round[1])
        rows_planned, last_img = 0, None
        while self.waiting:
            req = self.waiting[0]
            if self.admit_chunk_rows and rows_planned >= self.admit_chunk_rows and prepared:
                first = self._first_image_key(req)
                if first is None or first != last_img:
                    break                                      # a pass's worth is planned and the next request is about another tile
            if req.slot < 0:
                if not self.free:
                    break
                req.slot = self.free.pop()
            self.waiting.popleft()
            last_img = self._first_image_key(req)
            try:
                inp = self.processor(text=[req.prompt], images=list(req.images) or None, return_tensors="pt")
                ids = inp["input_ids"][0].tolist()
                grids = inp["image_grid_thw"].tolist() if req.images else []
                keys = list(inp.get("image_keys", []))
                mu = self.model.config.vision.spatial_merge_size ** 2
                n_img_tok = sum(g[0] * g[1] * g[2] // mu for g in grids)
                if ids.count(self.model.config.image_token_id) != n_img_tok:
                    raise ValueError("Image features and image tokens do not match")
                if len(ids) + 1 > e.max_ctx:
                    raise ValueError(f"prompt of {len(ids)} tokens exceeds max_ctx = {e.max_ctx}")
                rows = np.concatenate([[0], np.cumsum([g[0] * g[1] * g[2] for g in grids])]).astype(int)
                if len(grids) and int(np.diff(rows).max()) > e.max_patches:
                    raise ValueError(f"image of {int(np.diff(rows).max())} patches exceeds max_patches = {e.max_patches}")
                reuse, n_reused = self._reusable(req.slot, ids, keys)
                for i in range(n_reused, len(grids)):
                    if keys[i] in self._features:
                        self._features.move_to_end(keys[i])   # needed this round: not an eviction candidate
                    elif keys[i] not in todo:
                        todo[keys[i]] = (inp["pixel_values"][rows[i]:rows[i + 1]], grids[i])
                    needed.add(keys[i])
                prepared.append(dict(req=req, ids=ids, grids=grids, keys=keys, reuse=reuse, n_reused=n_reused, copy_from=None,
                                     upto=len(ids), final=True))
                rows_planned += len(ids) - reuse
            except Exception as ex:  # a malformed request must not take the batch down
                self._fail(req, ex)
        if not prepared and not self._carry:
            return
        # The round's prefill passes: rows of several chains share every GEMM, up to max_prefill_rows per pass; pass A (the
        # prefixes that other newcomers of this round will copy) goes first.  One pass runs per step (beside the decode burst
        # when overlapping); each encodes the images it needs that are not cached yet -- ONE multi-resolution ViT call.
        try:
            anchors = self._plan_sharing(prepared) if self.share_prefix else []
        except Exception as ex:
            self._fail_all([p["req"] for p in prepared if p["req"].slot >= 0], ex)
            return
        self._round = (todo, needed)
        carried, self._carry = self._carry, []
        for which, items in (("A", anchors), ("B", carried + prepared)):
            group, rows = [], 0
            for item in items + [None]:
                if group and (item is None or rows + item["upto"] - item["reuse"] > e.max_prefill_rows):
                    # admitting in chunks: the last, partial pass of a chunk waits for the next chunk's items (passes stay full)
                    if (item is None and which == "B" and self.admit_chunk_rows and rows < 0.85 * e.max_prefill_rows
                            and self.waiting and (self.waiting[0].slot >= 0 or self.free)):
                        self._carry = group
                    else:
                        self._groups.append(group)
                    group, rows = [], 0
                if item is not None:
                    group.append(item)
                    rows += item["upto"] - item["reuse"]
        if not self.overlap:  # everything at once: one ViT call per round, then the passes back to back
            while self._groups:
                self._run_group(self._groups.popleft())

    def _run_group(self, group) -> None:
        """One prefill pass of the round, behind the ViT call for its uncached images.  A failure (the ViT call, the pass)
        must not orphan a request that already left `waiting`: whatever escapes fails every request of the pass -- slots
        freed, on_error called."""
        todo, needed = self._round
        try:
            if self.overlap:
                want = OrderedDict()
                for it in group:
                    for k in it["keys"]:
                        if k in todo and k not in want:
                            want[k] = todo.pop(k)
                self._encode(want, needed)
            elif todo:
                self._encode(OrderedDict(todo), needed)
                todo.clear()
            n0 = len(self._ready)
            tr = self._trace_ev
            if tr is not None and len(tr["passes"]) < 4 and self._side is not None:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(self._side)
                self._prefill(group)
                b.record(self._side)
                tr["passes"].append((time.perf_counter() - tr["t0"], a, b))
            else:
                self._prefill(group)
            if self._side is not None and len(self._ready) > n0:
                # the pass's own event: its chains may join the live set as soon as THIS pass is over, whatever else the
                # admission stream has been given since
                ev = torch.cuda.Event()
                ev.record(self._side)
                self._pass_done.extend([ev] * (len(self._ready) - n0))
        except Exception as ex:
            self._fail_all([it["req"] for it in group if it["req"].slot >= 0 and it["req"].slot not in self.live
                            and not any(r is it["req"] for r, _, _ in self._ready)], ex)

    # -- shared prompt prefixes
    def _prefix_len(self, a, b) -> int:
        """Longest common prefix of two id lists that leaves both a non-empty tail and does not end inside a run of image
        tokens (a run is fed by one image's feature rows: it is shared whole or not at all)."""
        n = min(len(a), len(b)) - 1
        i = 0
        while i < n and a[i] == b[i]:
            i += 1
        img = self.model.config.image_token_id
        while i > 0 and a[i - 1] == img and a[i] == img:
            i -= 1
        return i

    def _images_in(self, ids, n) -> int:
        img = self.model.config.image_token_id
        return sum(1 for t in range(n) if ids[t] == img and (t == 0 or ids[t - 1] != img))

    def _plan_sharing(self, prepared):
        """Decides, for every fresh chain of the round, where its leading K/V rows come from; returns the pass-A items."""
        fresh = [p for p in prepared if p["reuse"] == 0 and p["keys"] and all(k is not None for k in p["keys"])]
        if not fresh:
            return []
        donors = {}   # first image key -> [(slot, ids, keys)]: chains that hold their prompt's K/V rows right now
        taken = {p["req"].slot for p in prepared}
        for slot, l in self.live.items():
            if l.keys and slot not in taken:
                donors.setdefault(l.keys[0], []).append((slot, l.ids, l.keys))
        for slot, (pids, pkeys, _gen) in self.parked.items():
            if pkeys:
                donors.setdefault(pkeys[0], []).append((slot, pids, pkeys))
        groups = OrderedDict()
        for p in fresh:
            groups.setdefault(p["keys"][0], []).append(p)
        anchors = []
        for k0, members in groups.items():
            rest = members
            for slot, dids, dkeys in donors.get(k0, []):   # an existing chain: every member that matches it copies from it
                for p in list(rest):
                    n = self._prefix_len(p["ids"], dids)
                    ni = self._images_in(p["ids"], n)
                    if n >= self.min_shared and tuple(p["keys"][:ni]) == tuple(dkeys[:ni]):
                        p.update(copy_from=slot, reuse=n, n_reused=ni)
                        rest = [q for q in rest if q is not p]
                if not rest:
                    break
            if len(rest) < 2:
                continue
            ref = rest[0]                                   # no donor yet: the first member's prefix is prefilled alone
            n = min(self._prefix_len(ref["ids"], p["ids"]) for p in rest[1:])
            ni = self._images_in(ref["ids"], n)
            if n < self.min_shared or any(tuple(p["keys"][:ni]) != tuple(ref["keys"][:ni]) for p in rest[1:]):
                continue
            anchors.append(dict(ref, upto=n, final=False))
            ref.update(reuse=n, n_reused=ni, copy_from=-1)   # -1: the rows are already in its own slot after pass A
            for p in rest[1:]:
                p.update(copy_from=ref["req"].slot, reuse=n, n_reused=ni)
        return anchors

    def _reusable(self, slot, ids, keys):
        """(cached prefix length, images inside it) when the slot's parked chain is a strict prefix of `ids`.  With
        `reuse_generated` the prefix extends over the tokens the parked chain GENERATED for as long as the new prompt
        repeats them id for id (stage 2 re-inserts the stage-1 output: src/eval/infer.py:222): their K/V rows were written
        by the decode steps and are kept instead of being prefilled again."""
        rec = self.parked.pop(slot, None)
        if rec is None:
            return 0, 0
        pids, pkeys, gen = rec
        n = len(pids)
        cfg = self.model.config
        if 0 < n < len(ids) and tuple(ids[:n]) == pids and tuple(keys[: len(pkeys)]) == pkeys \
                and ids[n] != cfg.image_token_id and all(k is not None for k in pkeys):
            special = (cfg.image_token_id, getattr(cfg, "vision_start_token_id", None), getattr(cfg, "vision_end_token_id", None))
            m = 0
            while m < len(gen) and n + m < len(ids) - 1 and ids[n + m] == gen[m] and gen[m] not in special:
                m += 1
            self.stats["reused_generated_rows"] += m
            self.stats["generated_rows_offered"] = self.stats.get("generated_rows_offered", 0) + max(0, min(len(gen), len(ids) - 1 - n))
            return n + m, len(pkeys)
        return 0, 0

    def _encode(self, todo, needed=()) -> None:
        """ONE multi-resolution ViT call (per max_patches worth of images) for the uncached images of this round."""
        e = self.engine
        items = list(todo.items())
        i = 0
        while i < len(items):
            j, n = i, 0
            while j < len(items) and (j == i or n + items[j][1][0].shape[0] <= e.max_patches):
                n += items[j][1][0].shape[0]
                j += 1
            pvs = [it[1][0] for it in items[i:j]]
            grids = [it[1][1] for it in items[i:j]]
            feats = e.vit_forward((torch.cat(pvs) if len(pvs) > 1 else pvs[0]).contiguous(), grids)
            self.stats["vit_calls"] += 1
            self.stats["vit_images"] = self.stats.get("vit_images", 0) + len(grids)
            self.stats["vit_patches"] = self.stats.get("vit_patches", 0) + int(n)
            off = 0
            mu = self.model.config.vision.spatial_merge_size ** 2
            for (key, (_, g)) in items[i:j]:
                k = g[0] * g[1] * g[2] // mu
                self._features[key] = feats[off:off + k]
                off += k
            i = j
        # LRU eviction, never of a feature this round's prefills are about to read
        for key in list(self._features.keys()):
            if len(self._features) <= max(self._feature_cap, len(needed)):
                break
            if key not in needed:
                del self._features[key]

    def _prefill(self, group) -> None:
        e = self.engine
        slots, ids_l, emb_l, pos_l, dl = [], [], [], [], []
        ok = []
        for it in group:
            req, ids, grids, keys, reuse, n_reused, upto = (it["req"], it["ids"], it["grids"], it["keys"], it["reuse"],
                                                           it["n_reused"], it["upto"])
            if req.slot < 0:
                continue                                       # failed earlier in this round (pass A)
            if it["copy_from"] is not None and e.seq_len(it["copy_from"] if it["copy_from"] >= 0 else req.slot) < reuse:
                # the donor's rows are gone (its pass failed; _fail resets the slot, so a stale context of the slot's
                # previous occupant cannot pass for them): prefill in full
                it["copy_from"], reuse, n_reused = None, 0, 0
                it.update(reuse=0, n_reused=0)
            try:
                pos, delta = e.rope_index(ids, grids)
                n_upto = len(keys) if upto == len(ids) else self._images_in(ids, upto)   # pass A stops after the prefix
                feats = [self._features[k] for k in keys[n_reused:n_upto]]
                for k in keys[n_reused:n_upto]:
                    self._features.move_to_end(k)
                emb = (torch.cat(feats) if len(feats) > 1 else feats[0]) if feats else None
                if it["copy_from"] is None:
                    if reuse:
                        e.seq_truncate(req.slot, reuse)       # a follow-up on its own slot: keep the cached prompt
                    else:
                        e.seq_reset(req.slot)
                elif it["copy_from"] >= 0:                     # shared prefix: the rows of another chain
                    e.seq_reset(req.slot)
                    e.seq_copy_prefix(req.slot, it["copy_from"], reuse)
                    self.stats["shared_rows"] += reuse
                # (copy_from == -1: pass A left the prefix in this slot)
                slots.append(req.slot)
                ids_l.append(ids[reuse:upto])
                emb_l.append(emb)
                pos_l.append(pos[:, reuse:upto])
                dl.append(delta)
                ok.append(it)
            except Exception as ex:
                self._fail(req, ex)
        if not ok:
            return
        final = [it for it in ok if it["final"]]                # (pass A: the chain is completed by pass B)
        try:
            if self.penalty != 1.0 and final:
                # the prompts' ids into the repetition-penalty sets (cleared by the reset / truncate above): one copy and one
                # launch for the pass, IN FRONT of it -- behind it, the next call would find its staging buffer busy until the
                # whole pass has run
                if hasattr(e, "mark_seen_batch") and not _PER_CHAIN:
                    e.mark_seen_batch([it["req"].slot for it in final], [it["ids"] for it in final])
                else:
                    for it in final:
                        e.mark_seen(it["req"].slot, it["ids"])
            e.prefill_batch(slots, ids_l, emb_l, pos_l, dl)
        except Exception as ex:
            self._fail_all([it["req"] for it in ok], ex)
            return
        self.stats["prefill_rows"] += sum(len(x) for x in ids_l)
        if os.environ.get("ZE_SCHED_TRACE"):   # measurement only
            with open(os.environ["ZE_SCHED_TRACE"], "a") as f:
                f.write(f"{id(self) % 100000} {time.perf_counter():.4f} P {len(slots)} {sum(len(x) for x in ids_l)}\n")
        for it in final:
            req, ids, keys = it["req"], it["ids"], it["keys"]
            req.n_prompt = len(ids)
            self._ready.append((req, tuple(ids), tuple(keys)))

    def _join_ready(self, wait: bool = False) -> None:
        """The prefilled newcomers draw their first token (from the logits their pass left) and join the live set.  When
        overlapping, a pass that is still running does not hold the live chains up: they go into their next burst, the
        newcomers join behind a later one (with nothing live, the call waits for the oldest pass)."""
        if not self._ready:
            return
        keep_from = len(self._ready)
        if self._side is not None:
            if len(self._pass_done) != len(self._ready):   # (entries without an event: wait for everything, as before)
                self._side.synchronize()
                self._pass_done = [None] * len(self._ready)
            for i, ev in enumerate(self._pass_done):
                if ev is not None and not ev.query():
                    if i == 0 and (wait or not self.live):
                        ev.synchronize()                    # nothing to decode meanwhile
                        continue
                    keep_from = i
                    break
        for req, ids, keys in self._ready[:keep_from]:
            self.engine.chain_begin(req.slot, self.params, req.stream_id)
            self.live[req.slot] = _Live(req, ids, keys)
            self.stats["admitted"] += 1
        if keep_from < len(self._ready):
            self.stats["deferred_joins"] = self.stats.get("deferred_joins", 0) + 1
        self._ready = self._ready[keep_from:]
        self._pass_done = self._pass_done[keep_from:]

    # histogram of the decode steps by their live-chain count (the row count of the step's GEMMs picks their tiles: which
    # buckets the stream spends its steps in says which tile is worth work); keys `steps_le_<bound>`
    _ROW_BOUNDS = (64, 128, 192, 256, 320, 384, 416, 448, 512, 640, 768)

    def _tally_step_rows(self, ran: int, n: int) -> None:
        if ran <= 0:
            return
        if os.environ.get("ZE_SCHED_TRACE"):   # measurement only: one line per burst (tools/sched_trace.py draws the occupancy)
            with open(os.environ["ZE_SCHED_TRACE"], "a") as f:
                f.write(f"{id(self) % 100000} {time.perf_counter():.4f} B {n} {ran} {len(self.waiting)} {len(self._groups)} {len(self._ready)}\n")
        b = next((x for x in self._ROW_BOUNDS if n <= x), None)
        k = f"steps_le_{b}" if b is not None else "steps_gt_768"
        self.stats[k] = self.stats.get(k, 0) + ran

    # -- one burst of decode steps for every live chain, then retire the finished ones
    def _burst(self) -> None:
        e = self.engine
        slots = list(self.live.keys())
        steps = self._burst_steps()
        ran, n_gen, fin = e.decode_burst(slots, steps, self.params)
        self.stats["bursts"] += 1
        self.stats["steps"] += ran
        self.stats["chain_steps"] += ran * len(slots)
        self._tally_step_rows(ran, len(slots))
        self._retire_finished(slots, n_gen, fin)

    def _retire_finished(self, slots, n_gen, fin) -> None:
        """The chains the burst finished leave the live set: their ids come over in ONE copy (chain_tokens_batch), then the
        callbacks run."""
        e = self.engine
        out = []
        for slot, ng, f in zip(slots, n_gen, fin):
            l = self.live[slot]
            l.produced = ng
            if f or ng >= min(l.req.max_new_tokens, e.max_ctx - l.req.n_prompt + 1):
                out.append(slot)
        toks = None
        if len(out) > 1 and hasattr(e, "chain_tokens_batch") and not _PER_CHAIN:
            ds = getattr(self, "_decode_stream", None)
            cap = max(self.live[s].req.max_new_tokens for s in out)
            toks = e.chain_tokens_batch(out, cap, stream=ds) if ds is not None else e.chain_tokens_batch(out, cap)
        for i, slot in enumerate(out):
            self._retire(slot, None if toks is None else toks[i][:self.live[slot].req.max_new_tokens])

    def _burst_steps(self) -> int:
        """Steps of the next burst: never past the chain with the fewest tokens left.  (Letting chains overrun their budget by
        up to burst - 1 steps, as a chain that ends with an EOS does, makes the bursts of a ragged stream 8 steps long instead
        of 1-3 -- and was no faster: 80.4 / 81.7 against 81.9 / 83.3 questions/s at bursts of 8, 84.1 / 83.4 at bursts of 4;
        the waits between bursts are not what the stream spends its time on.)"""
        e = self.engine
        budget = min(min(l.req.max_new_tokens, e.max_ctx - l.req.n_prompt + 1) - l.produced for l in self.live.values())
        return max(0, min(self.burst, budget))

    def _burst_begin(self):
        e = self.engine
        slots = list(self.live.keys())
        steps = self._burst_steps()
        ran = e.decode_burst_begin(slots, steps, self.params)
        return slots, ran

    def _burst_end(self, handle) -> None:
        e = self.engine
        slots, ran = handle
        self._decode_stream = torch.cuda.current_stream(e.device) if self._side is not None else None
        t0 = time.perf_counter()
        n_gen, fin = e.decode_burst_end(slots)
        self.stats["host_s_burst_wait"] = self.stats.get("host_s_burst_wait", 0.0) + time.perf_counter() - t0
        self.stats["bursts"] += 1
        self.stats["steps"] += ran
        self.stats["chain_steps"] += ran * len(slots)
        self._tally_step_rows(ran, len(slots))
        with self._side_stream():   # (the callbacks of finished chains crop / resize on the front-end's stream)
            self._retire_finished(slots, n_gen, fin)

    def _retire(self, slot: int, tokens=None) -> None:
        l = self.live.pop(slot)
        req = l.req
        ds = getattr(self, "_decode_stream", None)
        if tokens is not None:
            req.tokens = tokens
        else:
            req.tokens = (self.engine.chain_tokens(slot, req.max_new_tokens, stream=ds) if ds is not None
                          else self.engine.chain_tokens(slot, req.max_new_tokens))
        req.text = self.processor.tokenizer.decode(req.tokens, skip_special_tokens=True).strip()
        follow = None
        try:
            follow = req.on_done(req, req.tokens, req.text) if req.on_done else None
        except Exception as ex:
            if req.on_error:
                req.on_error(req, ex)
            else:
                raise
        if follow is not None:  # continues on this slot; its cached prompt is reusable
            follow.slot = slot
            # ... and so are the rows of the generated tokens that went through the model (all but the last one sampled)
            cached = min(self.engine.seq_len(slot) - len(l.ids), len(req.tokens) - 1) if self.reuse_generated else 0
            self.parked[slot] = (l.ids, l.keys, tuple(req.tokens[:max(0, cached)]))
            self.waiting.appendleft(follow)
        else:
            self._retire_slot(slot)
            self.free.append(slot)

    def _retire_slot(self, slot: int) -> None:
        """The slot's rows may be overwritten from now on: the chains that read their prompt prefix from it (the decode
        attention streams ONE copy of a tile's image prefix) are moved to another holder -- on the stream the decode steps run
        on, i.e. before the next burst, whatever stream the admission work uses."""
        retire = getattr(self.engine, "seq_retire", None)
        if retire is not None:
            retire(slot, stream=getattr(self, "_decode_stream", None))

    def _release(self, req: Request) -> None:
        if req.slot >= 0:
            self.parked.pop(req.slot, None)
            try:
                self._retire_slot(req.slot)
                self.engine.seq_reset(req.slot)   # nothing may copy a prefix from what the slot held before
            except Exception:
                pass
            self.free.append(req.slot)
            req.slot = -1

    def _fail(self, req: Request, ex: Exception) -> None:
        self._release(req)
        if req.on_error:
            req.on_error(req, ex)
        else:
            raise ex

    def _fail_all(self, reqs, ex: Exception) -> None:
        """Every request of `reqs` gets the error; the ones without an `on_error` re-raise it once all slots are back."""
        unhandled = False
        for req in reqs:
            self._release(req)
            if req.on_error:
                req.on_error(req, ex)
            else:
                unhandled = True
        if unhandled:
            raise ex


def run_requests(model, processor, requests, **kw) -> None:
    """Convenience: every request through one scheduler, to completion."""
    s = ChainScheduler(model, processor, **kw)
    for r in requests:
        s.submit(r)
    s.run()
