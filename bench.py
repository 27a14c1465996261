#!/usr/bin/env python3
"""Headline benchmark: questions/sec end-to-end for the ZoomEarth-3B zoom chain on 5000-px tiles, at N MI355X.

`python bench.py --gpus N --steps K --warmup W` runs the LRS-GRO question stream of BASELINE configs[3] at EVERY N
(1 included): a synthetic question table (tiles of 5000 x 5000 px with 3..18 questions each, 10.7 on average as LRS-GRO's
9734 questions about 908 images) is assigned to the N ranks tile by tile (`accel.shard_by_tile`: a tile never splits,
longest-processing-time packing), and every rank drives its share through the continuous-batching scheduler
(`zoomearth_amd/scheduler.py`, the code path of `src/eval/infer.py`) with two lanes of 768 chain slots.  One "step" = 64 questions per
GPU entering the stream; the K timed steps are ONE stream of K x 64 questions per GPU (filled at the start, drained at the
end, both inside the timed region), bracketed by barrier + device synchronisation; `value` = all questions of all ranks
over the slowest rank's time.  A question = one full two-stage zoom chain:
  K0  tile 5000^2 -> 512^2 bicubic view            K1/K2 smart_resize 504^2 + patchify (1296 patches)
  ViT 1296 patches -> 324 image tokens             prefill L1 = 802 tokens, decode N1 ~ 192 (greedy, penalty 1.05)
  scripted bbox -> 512^2 crop of the FULL-RES tile  ViT on the crop (view features reused: identical bits)
  prefill of the new tokens after the cached stage-1 prompt (L2 ~ 1320), decode N2 ~ 96; N1 / N2 ragged (+-25 %)
Tiles are resident in HBM when the timed region starts (uploaded from pinned host memory before it; the upload rate is
reported).  Text ids are synthetic with the structure of the reference prompt: 21 system-turn ids and 437 instruction
ids that are the same for every question, 18 question ids that differ.  Weights: Qwen2.5-VL-3B shape, bf16, synthetic
N(0, 0.02^2) from the repo PRNG (no checkpoint offline), generated on rank 0 and broadcast once (RCCL over xGMI) to the
other ranks -- the path's only collective.  Control flow is scripted (random weights emit neither EOS nor a bbox).

At N = 1 the same line carries, as named sub-objects with their own rooflines: `configs1` (BASELINE configs[1]: one
chain, batch 1, the single-chain GEMV decode path), `batch64` (BASELINE configs[2]: 64 chain slots) and `cpu_baseline`
(the reference's own transformers CPU path on this box's host cores).  `--batch 1` makes configs[1] the line's value,
`--batch B` configs[2] with B chains.  `--gpus N` without a torchrun environment spawns the N ranks itself (before
anything touches the GPU).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

L_TEXT_A, L_TEXT_B, N1, N2, PENALTY = 21, 455, 192, 96, 1.05  # 21 + 455 = 476 text ids
HBM_PEAK_GBS = 8000.0
BATCH_KERNELS = {0: "qkv", 1: "o_proj", 2: "gate_up", 3: "down", 4: "lm_head", 5: "attention", 6: "rmsnorm", 7: "rope_kv"}
BATCH_KERNEL_NAMES_WIDE = {
    0: "k_gemm_ring<64..128,64,QKV_ROPE> (row-streaming decode qkv projection + M-RoPE + KV append in one launch, rows = chains; the row tile follows the row count: one round of workgroups)",
    1: "k_gemm_ring<64..128,64,RESIDUAL> (row-streaming decode o projection + residual)",
    2: "k_gemm_ring<192,192 / 256,192 / 320,192 / 384,192,SWIGLU> by row count above 256 chains, k_gemm_wstream<BM,96,SWIGLU> below (row-streaming decode gate/up projection)",
    3: "k_gemm_ring split-K x 8 + k_splitk_reduce (row-streaming decode down projection; 192 x 128 / 128 x 256 / 320 x 128 / 384 x 128 tiles by row count)",
    4: "k_gemm_wstream / k_gemm_ring<256,256,F32> (row-streaming decode lm_head)",
    5: "k_attn_decode_wave_long<8, 6, false> (batched decode attention: 384-key parts, every wave streams 16 keys of each 64-key round -- K rows straight into MFMA registers, V rows through its own LDS stages -- three rounds in flight until the part ends)",
}
BATCH_KERNEL_NAMES = {
    0: "k_gemm_oneshot<QKV,4> (batched decode qkv projection + M-RoPE + KV append, sixteen waves per workgroup)",
    1: "k_gemm_oneshot<RESIDUAL,4> (batched decode o projection)",
    2: "k_gemm_skinny<6,SWIGLU,FRAG,BAL> (batched decode gate/up weight stream, one workgroup per CU)",
    3: "k_gemm_ring<64,64,4,RESIDUAL> split-K (batched decode down projection)",
    4: "k_gemm_skinny<2,F32,FRAG> (batched decode lm_head)",
    5: "k_attn_decode_wave_long<8, 6, false> (batched decode attention: 384-key parts, every wave streams 16 keys of each 64-key round -- K rows straight into MFMA registers, V rows through its own LDS stages -- three rounds in flight until the part ends)",
}


L_QUESTION = 18  # of the 455 ids after the image: the question itself; the other 437 are the instruction


def text_ids(q: int):
    """The 476 synthetic text ids of question q with the structure of the reference prompt (src/eval/infer.py:180-214;
    SURVEY 8d: ~21 system-turn ids, ~18 question ids, ~437 instruction ids): the system turn and the instruction are the
    same text for every question, only the question differs."""
    from zoomearth_amd.synth import uniform_ints
    a = uniform_ints(7, L_TEXT_A, 1000, 150000).tolist()
    b = uniform_ints(7_000_003 + q, L_QUESTION, 1000, 150000).tolist() + \
        uniform_ints(8_000_003, L_TEXT_B - L_QUESTION, 1000, 150000).tolist()
    return a, b


def question_ids(cfg, q: int, n_img: int):
    a, b = text_ids(q)
    return a + [cfg.vision_start_token_id] + [cfg.image_token_id] * n_img + [cfg.vision_end_token_id] + b


def scripted_bbox(q: int, tile_side: int):
    """bbox in tile pixels from the rl.jsonl size statistics: 96.8 % small boxes (-> 512^2 crop),
    3.2 % large boxes (up to 2500^2 -> bicubic downscale branch)."""
    from zoomearth_amd.synth import uniform_ints
    r = uniform_ints(99 + q, 4, 0, 1 << 30)
    large = (r[0] % 1000) < 32
    side = 600 + int(r[1] % 1900) if large else 40 + int(r[1] % 420)
    x = int(r[2] % (tile_side - side))
    y = int(r[3] % (tile_side - side))
    return [float(x), float(y), float(x + side), float(y + side)]


def ragged_lengths(q: int):
    """(N1, N2) of question q for the batched workload: +-25 % around the scripted means."""
    from zoomearth_amd.synth import uniform_ints
    r = uniform_ints(555 + q, 2, 0, 1 << 20)
    return int(round(N1 * (0.75 + 0.5 * (r[0] / float(1 << 20))))), int(round(N2 * (0.75 + 0.5 * (r[1] / float(1 << 20)))))


class Chain:
    """The scripted two-stage chain on one engine (all compute through the C ABI)."""

    def __init__(self, engine, tile, use_graph=True):
        from zoomearth_amd import hostloop
        self.e, self.tile, self.H = engine, tile, hostloop
        self.side = int(tile.shape[0])
        self.use_graph = use_graph

    def question(self, q: int):
        e, cfg, H = self.e, self.e.config, self.H
        side = self.side
        # ---- stage 1
        s = 512 / side
        view = e.crop_resize(self.tile, (0, 0, side, side), (int(side * s), int(side * s)))
        pv_v, g_v = e.preprocess_image(view)
        emb_v = e.vit_forward(pv_v, [g_v])
        n_img = g_v[1] * g_v[2] // 4
        ids1 = question_ids(cfg, q, n_img)
        pos, delta = e.rope_index(ids1, [g_v])
        e.seq_reset(0)
        e.prefill(0, ids1, emb_v, pos, delta, want_logits=False)
        e.mark_seen(0, ids1)
        out1 = e.generate(0, N1, repetition_penalty=PENALTY, ignore_eos=True, use_graph=self.use_graph, sync_every=N1)
        # ---- zoom: crop the FULL-RES tile around the (scripted) bbox, <=512 px
        box = H.zoom_box((side, side), scripted_bbox(q, side))
        bw, bh = box[2] - box[0], box[3] - box[1]
        sc = min(1.0, 512 / max(bw, bh))
        crop = e.crop_resize(self.tile, box, (int(bw * sc), int(bh * sc)) if sc < 1 else (bw, bh))
        pv_c, g_c = e.preprocess_image(crop)
        emb_c = e.vit_forward(pv_c, [g_c])  # the view's features are reused (bit-identical, tested)
        # ---- stage 2: cached stage-1 prompt + re-fed stage-1 output + second vision block
        ids2 = ids1 + refeed(cfg, out1) + [cfg.vision_start_token_id] + [cfg.image_token_id] * (g_c[1] * g_c[2] // 4) + [cfg.vision_end_token_id]
        pos2, delta2 = e.rope_index(ids2, [g_v, g_c])
        # the rows of the generated tokens the prompt repeats id for id stay too (all but the last one sampled, which never
        # went through the model), as in the scheduler (ChainScheduler.reuse_generated, the path of src/eval/infer.py)
        keep = len(ids1)
        while keep - len(ids1) < len(out1) - 1 and ids2[keep] == out1[keep - len(ids1)]:
            keep += 1
        e.seq_truncate(0, keep)
        e.prefill(0, ids2[keep:], emb_c, pos2[:, keep:], delta2, want_logits=False)
        e.mark_seen(0, ids2)
        out2 = e.generate(0, N2, repetition_penalty=PENALTY, ignore_eos=True, use_graph=self.use_graph, sync_every=N2)
        self.kept_generated = keep - len(ids1)   # rows of generated tokens kept instead of prefilled again
        return out1, out2, len(ids1), len(ids2)


def refeed(cfg, toks):
    """Generated ids as they re-enter the stage-2 prompt.  The reference decodes with skip_special_tokens=True and
    re-tokenises (src/eval/infer.py:122,222), so special ids never come back; with random weights the argmax can be a
    special id (e.g. <|image_pad|>, which would desynchronise the image-token count): map those to a plain id."""
    lo = min([cfg.image_token_id, cfg.vision_start_token_id, cfg.vision_end_token_id, cfg.pad_token_id, *cfg.eos_token_ids])
    return [t if t < lo else 1000 for t in toks]


# ----------------------------------------------------------------------------- configs[2]: 64 chains, continuous batching
class SynthTokenizer:
    """No tokenizer files offline: "decoding" writes the ids as decimal words (special ids mapped to a plain one, as
    skip_special_tokens + re-tokenisation would never bring them back), "encoding" reads them again."""

    def __init__(self, cfg):
        self.cfg = cfg

    def decode(self, ids, skip_special_tokens=True):
        return " ".join(str(t) for t in refeed(self.cfg, ids))


class SynthProcessor:
    """The processor surface the scheduler calls (`processor(text=[prompt], images=[...])`, `.tokenizer.decode`) with
    synthetic token ids: the real image path (ZoomEarthProcessor.preprocess_images -> HIP front-end), and for the text
    the scripted lengths of SURVEY 8d -- 21 ids before the first image (the system turn: the same for every question),
    455 after it (18 of the question, 437 of the instruction), then whatever decimal words follow `assistant\\n` (the
    re-fed stage-1 output)."""

    def __init__(self, cfg, engine):
        from zoomearth_amd.processor import ZoomEarthProcessor
        self.cfg = cfg
        self.tokenizer = SynthTokenizer(cfg)
        self._img = ZoomEarthProcessor.__new__(ZoomEarthProcessor)
        self._img.engine, self._img.min_pixels, self._img.max_pixels, self._img.merge_size = engine, 3136, 128 * 128 * 28 * 28, 2

    def __call__(self, text, images=None, return_tensors="pt", **kw):
        import torch
        from zoomearth_amd.hostloop import VISION_BLOCK
        cfg = self.cfg
        prompt = text[0]
        q = int(prompt.split("#q", 1)[1].split("#", 1)[0])
        pv, grids, keys = self._img.preprocess_images(list(images))
        parts = prompt.split(VISION_BLOCK)
        assert len(parts) == len(grids) + 1
        ta, tb = text_ids(q)
        ids = list(ta)
        for i, g in enumerate(grids):
            ids += [cfg.vision_start_token_id] + [cfg.image_token_id] * (g[0] * g[1] * g[2] // 4) + [cfg.vision_end_token_id]
            if i == 0:
                ids += tb
                tail = parts[1].split("assistant\n", 1)[1]
                ids += [int(w) for w in tail.split()]
        return dict(input_ids=torch.tensor([ids]), image_grid_thw=torch.tensor(grids), pixel_values=pv, image_keys=keys)


class Model64:
    """What ChainScheduler needs of the model object."""

    def __init__(self, engine):
        from types import SimpleNamespace
        self.engine, self.config = engine, engine.config
        self.generation_config = SimpleNamespace(repetition_penalty=PENALTY, temperature=None)
        self._chains = {}


Q_STEP = 64          # questions per step per GPU of the stream workload
STREAM_SLOTS = 768   # chain slots per LANE of the stream workload (the engine's choice in configs[3]; on the 1280-question stream
                     # of --steps 20, one lane: 256 slots answer 53.0, 384 54.0, 512 56.3, 1024 60.3 questions/s; two lanes: 2 x 256
                     # 62.9, 2 x 512 66.2, 2 x 768 67.6, 2 x 1024 67.7; 58 GB of KV cache per lane at max_ctx 2048)


def question_table(n_questions: int, seed: int = 0):
    """Tile index of every question of a synthetic LRS-GRO-like table: tile t has 3..18 questions (10.5 on average; LRS-GRO:
    9734 questions about 908 images = 10.7), questions grouped by tile as the dataset lists them."""
    from zoomearth_amd.synth import uniform_ints
    counts, total, t = [], 0, 0
    while total < n_questions:
        c = 3 + int(uniform_ints(424_242 + seed * 7919 + t, 1, 0, 16)[0])
        c = min(c, n_questions - total)
        counts.append(c)
        total += c
        t += 1
    return [t for t, c in enumerate(counts) for _ in range(c)]


class TileFeeder:
    """Tile uploads INSIDE the timed region (VERDICT r4 #3 / SURVEY 8d: the metric starts at "tile decoded in host RAM"): a lane's
    tiles go pinned host memory -> HBM (ze_tile_upload) on a stream of the feeder's own, `depth` tiles ahead of their first
    question, so the 75-MB copies run beside the lane's compute; tile(t)() makes the caller's stream wait for tile t's copy and
    hands the DeviceImage over (the same object for every question of the tile).  What image.TilePrefetcher does for
    src/eval/infer.py behind a file decode; here the "decoded" tiles are the synthetic pool."""

    def __init__(self, engine, host_pool, order, depth: int = 3):
        import torch
        self.e, self.pool, self.order, self.depth = engine, host_pool, list(order), depth
        self.pos = {t: i for i, t in enumerate(self.order)}
        self.stream = torch.cuda.Stream(device=engine.device)
        self.img, self.ev, self.timed, self.next = {}, {}, [], 0

    def _kick(self, upto: int) -> None:
        import torch
        from zoomearth_amd.image import DeviceImage
        while self.next < len(self.order) and self.next <= upto:
            t = self.order[self.next]
            with torch.cuda.stream(self.stream):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                d = self.e.tile_upload(self.pool[t % len(self.pool)] if isinstance(t, int) else self.pool[t[1] % len(self.pool)])
                b.record()
            self.img[t], self.ev[t] = DeviceImage(d, self.e), b
            self.timed.append((a, b))
            self.next += 1

    def tile(self, t):
        def get():
            import torch
            self._kick(self.pos[t] + self.depth)
            torch.cuda.current_stream(self.e.device).wait_event(self.ev[t])
            return self.img[t]
        return get

    def upload_ms(self):
        """(total device ms of the copies, number of tiles) -- after the run"""
        return sum(a.elapsed_time(b) for a, b in self.timed), len(self.timed)


REUSE_GENERATED = True   # ChainScheduler.reuse_generated of the streams below (the `reuse_sensitivity` annex runs one stream with it off)


def run_stream(engine, table, q0: int, slots: int, stats=None, use_graph=None):
    """The questions `table` = [(question number offset, tile DeviceImage, view key)] through the continuous-batching
    scheduler and the host code of src/eval/infer.py (hostloop: views, crops, prompts); the "parsed" box is scripted,
    lengths are ragged and EOS is ignored (random weights emit neither).  Every question of a (tile, view key) looks at
    the same <=512-px view, encoded once."""
    from zoomearth_amd import hostloop as H
    from zoomearth_amd.scheduler import ChainScheduler, Request

    model = Model64(engine)
    proc = SynthProcessor(engine.config, engine)
    sched = ChainScheduler(model, proc, do_sample=False, repetition_penalty=PENALTY, ignore_eos=True, burst=int(os.environ.get("ZE_BURST", "8")),
                           use_graph=use_graph, min_admit=int(os.environ.get("ZE_MIN_ADMIT", str(max(1, 3 * slots // 4 if slots > 64 else slots // 2)))), hold_below=int(os.environ.get("ZE_HOLD", str(3 * slots // 4 if slots > 64 else 0))),
                           max_wait_bursts=int(os.environ.get("ZE_MAX_WAIT", "8" if slots > 64 else "12")), max_batch=slots,
                           reuse_generated=REUSE_GENERATED,
                           # round 6: the queue is taken in 1.7 passes' worth of raw prompt rows at a time (the first pass is enqueued ~30 ms
                           # into the run instead of 0.4 s; with the hold at 3/4 of the slots: 92.2 against 91.0 questions/s, same box, five
                           # runs each -- profiles/r06_ab_admit_chunks.txt; ZE_ADMIT_ROWS=0 ZE_HOLD=512: round 5's setting)
                           admit_chunk_rows=int(os.environ.get("ZE_ADMIT_ROWS", str(int(1.7 * engine.max_prefill_rows) if slots > 64 else 0))))
    done = {}
    views = {}
    for b, tile, vkey in table:
        q = q0 + b
        # (as src/eval/infer.py: the queue stays short -- tiles are uploaded shortly before their first question, while the chains
        #  already in advance -- instead of every question of the run being submitted before the first step)
        while len(sched.waiting) >= slots:
            sched.step()
        if callable(tile):
            tile = tile()          # TileFeeder: the upload was requested tiles ahead on a stream of its own
        if vkey not in views:
            views[vkey] = H.resize_image(tile)
        n1, n2 = ragged_lengths(q)
        text = H.stage1_prompt(f"#q{q}#")

        def stage1(req, toks, out1, q=q, tile=tile, view=views[vkey][0], n2=n2, text=text):
            crop, _ = H.resize_image(H.cut_image(tile, scripted_bbox(q, tile.width)))

            def stage2(req2, toks2, out2, q=q, n_p1=req.n_prompt, n_o1=len(toks)):
                done[q] = (n_p1, n_o1, req2.n_prompt, len(toks2))
            return Request(prompt=H.stage2_prompt(text, out1), images=[view, crop], max_new_tokens=n2, on_done=stage2)

        sched.submit(Request(prompt=text, images=[views[vkey][0]], max_new_tokens=n1, on_done=stage1))
    sched.run()
    assert len(done) == len(table)
    if stats is not None:
        stats.update(sched.stats)
        stats["lens"] = [done[q0 + b] for b, _, _ in table]
    return done


def run_stream_lanes(engines, table, q0: int, slots: int, stats=None, use_graph=None, host_pool=None):
    """The same stream on several LANES of one GPU: every lane is an engine of its own (weights, KV cache, workspaces) with
    its own scheduler, host thread and HIP stream; tiles are dealt to the lanes whole (a tile's questions share its view and
    its prompt prefix).  While one lane is in a prefill / ViT round (matrix-bound) the other one decodes (bandwidth- and
    latency-bound): the GPU overlaps them, and the host work of one lane hides behind the device work of the other."""
    import threading

    import torch
    from zoomearth_amd.image import DeviceImage
    lane_of, parts = {}, [[] for _ in engines]
    for b, tile, vkey in table:
        ln = lane_of.setdefault(vkey, len(lane_of) % len(engines))
        parts[ln].append((b, tile, vkey))
    outs, errs, sts = [None] * len(engines), [], [dict() for _ in engines]
    feeders = [None] * len(engines)

    def lane_table(ln):
        """host_pool given: `tile` entries are tile ids, uploaded by the lane's TileFeeder as the stream reaches them."""
        e = engines[ln]
        if host_pool is None:
            own, mine = {}, []
            for b, tile, vkey in parts[ln]:  # the lane's engine runs the tile's front-end launches
                if vkey not in own:
                    own[vkey] = tile if tile.engine is e else DeviceImage(tile.base, e)
                mine.append((b, own[vkey], vkey))
            return mine
        order = []
        for _, tile, _ in parts[ln]:
            if not order or order[-1] != tile:
                if tile not in order:
                    order.append(tile)
        feeders[ln] = TileFeeder(e, host_pool, order)
        return [(b, feeders[ln].tile(tile), vkey) for b, tile, vkey in parts[ln]]

    if len(engines) == 1:
        done = run_stream(engines[0], lane_table(0), q0, slots, stats, use_graph)
        if stats is not None and feeders[0] is not None:
            torch.cuda.synchronize()
            stats["tile_upload_ms_total"], stats["tiles_uploaded"] = feeders[0].upload_ms()
        return done

    def work(ln):
        try:
            e = engines[ln]
            prof = None
            if ln == 0 and os.environ.get("ZE_BENCH_HOSTPROF") == "1":  # measurement only: cProfile of lane 0's host thread -> stderr
                import cProfile
                prof = cProfile.Profile()
                prof.enable()
            with torch.cuda.stream(torch.cuda.Stream(device=e.device)):
                outs[ln] = run_stream(e, lane_table(ln), q0, slots, sts[ln], use_graph)
                torch.cuda.current_stream().synchronize()
            if prof is not None:
                import io
                import pstats
                prof.disable()
                buf = io.StringIO()
                pstats.Stats(prof, stream=buf).sort_stats("tottime").print_stats(45)
                print(buf.getvalue(), file=sys.stderr)
        except BaseException as ex:  # noqa: BLE001
            errs.append(ex)

    threads = [threading.Thread(target=work, args=(ln,)) for ln in range(len(engines))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errs:
        raise errs[0]
    done = {}
    for o in outs:
        done.update(o)
    if stats is not None:
        for st in sts:
            for k, v in st.items():
                if k != "lens":
                    stats[k] = stats.get(k, 0) + v
        stats["lens"] = [done[q0 + b] for b, _, _ in table]
        stats["lanes"] = len(engines)
        if any(f is not None for f in feeders):
            torch.cuda.synchronize()
            ups = [f.upload_ms() for f in feeders if f is not None]
            stats["tile_upload_ms_total"], stats["tiles_uploaded"] = sum(u[0] for u in ups), sum(u[1] for u in ups)
    return done


def batch_step(engine, tiles, q0: int, B: int, stats=None, use_graph=None, slots=None):
    """B questions about len(tiles) tiles (6 : 64 as LRS-GRO's 908 : 9734), passes of 64 questions over the tiles."""
    table = []
    for b in range(B):
        t = b * len(tiles) // min(B, 64) if B <= 64 else (b // 64) * len(tiles) + (b % 64) * len(tiles) // 64
        table.append((b, tiles[t % len(tiles)], t))
    return run_stream(engine, table, q0, slots or B, stats, use_graph)


# ----------------------------------------------------------------------------- CPU baseline (rank 0, N = 1 only)
def cpu_baseline(budget_s: float = 45.0):
    """The oracle (numpy port of the reference arithmetic, fp32 BLAS on all host cores) on a bounded sample,
    extrapolated by layer count to the question AS THE REFERENCE EXECUTES IT (no reuse: 3 view encodes, 802- and
    1320-token prefills, 288 decode steps).  Fallback when the transformers path cannot run."""
    from oracle import qwen25vl as Q
    cores = os.cpu_count() or 1
    full = Q.Config()
    vd, td = 3, 2
    cfg = Q.Config(vision=Q.VisionConfig(depth=vd, fullatt_block_indexes=(2,)), text=Q.TextConfig(num_hidden_layers=td))
    rng = np.random.default_rng(0)
    w = {}
    for name, shape in Q.weight_shapes(cfg).items():
        if name.endswith("norm1.weight") or name.endswith("norm2.weight") or "layernorm" in name or name.endswith("ln_q.weight") or name.endswith("model.norm.weight"):
            w[name] = np.ones(shape, np.float32)
        elif name.endswith(".bias"):
            w[name] = np.zeros(shape, np.float32)
        else:
            w[name] = rng.standard_normal(shape, dtype=np.float32) * np.float32(0.02)
    o = Q.Qwen25VLOracle(cfg, w, "fp32")
    pv = rng.standard_normal((1296, 1176), dtype=np.float32)
    grid = [(1, 36, 36)]
    t_depth = {}
    o.cfg.vision.depth = 1
    o.vit_forward(pv, grid)  # untimed warm-up (imports, page faults)
    for d in (1, 2, 3):
        o.cfg.vision.depth = d
        t0 = time.perf_counter()
        emb = o.vit_forward(pv, grid)
        t_depth[d] = time.perf_counter() - t0
    t_win = max(t_depth[2] - t_depth[1], 0.0)
    t_fullblk = max(t_depth[3] - t_depth[2], 0.0)
    t_over = max(t_depth[1] - t_win, 0.0)  # patch embed + merger
    n_full = len(full.vision.fullatt_block_indexes)
    ids = list(rng.integers(1000, 150000, 21)) + [full.vision_start_token_id] + [full.image_token_id] * 324 + \
        [full.vision_end_token_id] + list(rng.integers(1000, 150000, 455))
    t0 = time.perf_counter()
    lg = o.prefill(ids, image_embeds=emb, grid_thw=grid)
    t_pre = time.perf_counter() - t0
    nd = 4
    t0 = time.perf_counter()
    for _ in range(nd):
        lg = o.decode_step(int(np.argmax(lg)))
    t_dec = (time.perf_counter() - t0) / nd
    h = rng.standard_normal((1, 2048), dtype=np.float32)
    t0 = time.perf_counter()
    for _ in range(4):
        _ = h @ o.w["lm_head.weight"].T
    t_head = (time.perf_counter() - t0) / 4
    vit_full = t_over + t_win * (full.vision.depth - n_full) + t_fullblk * n_full
    pre_layer = max(t_pre - t_head, 0.0) / td
    dec_layer = max(t_dec - t_head, 0.0) / td
    L1, L2 = 802, 1320
    t_question = (3 * vit_full + (pre_layer * full.text.num_hidden_layers) * (L1 + L2) / L1 + 2 * t_head
                  + (N1 + N2) * (dec_layer * full.text.num_hidden_layers + t_head))
    return {
        "value": 1.0 / t_question, "unit": "questions/s", "cores": cores, "threads": cores, "kind": "port",
        "sample": (f"EXTRAPOLATED: oracle (numpy fp32 BLAS) timed on 3 of 32 ViT blocks over 1296 patches (window block {t_win:.3f}s, full-attention block {t_fullblk:.3f}s, embed+merger {t_over:.3f}s), 2 of 36 decoder "
                   f"layers prefilling 802 tokens ({t_pre:.2f}s), {nd} decode steps ({t_dec:.3f}s each incl. lm_head "
                   f"{t_head:.3f}s); scaled by layer count to the as-executed question (3 view encodes, 802+1320 "
                   f"prefill, 288 decode) = {t_question:.1f}s"),
    }


def cpu_baseline_hf(full_depth: bool = True):
    """The reference's own CPU path -- the installed `transformers` Qwen2_5_VLForConditionalGeneration that
    src/eval/infer.py:147-151 instantiates -- in bf16 with random weights at the FULL 3B depth, timed on ONE whole
    question AS THE REFERENCE EXECUTES IT: stage 1 = generate() on the 802-token prompt with the 1296-patch view
    (192 new tokens), stage 2 = a fresh generate() on the 1320-token prompt with view + crop (two more view-sized
    encodes, 96 new tokens); greedy, repetition penalty 1.05, lengths forced (min = max new tokens).
    full_depth=False: the bounded layer sample of round 1 (8 of 32 ViT blocks, 2 of 36 layers), extrapolated."""
    import torch
    import transformers
    from transformers import Qwen2_5_VLConfig, Qwen2_5_VLForConditionalGeneration
    try:
        from transformers.initialization import no_init_weights
    except Exception:  # older layouts
        from transformers.modeling_utils import no_init_weights
    ncpu = os.cpu_count() or 1
    threads = min(ncpu, int(os.environ.get("ZE_HF_THREADS", "16")))  # the best setting found on the GPU box (2 x EPYC 9575F): 8 threads 69 s per question, 16: 37.5 s, 32: 95 s, 64: 247 s, 256: > 15 min -- the small GEMMs of a decode step do not scale
    torch.set_num_threads(threads)
    t_start = time.perf_counter()

    def note(msg):
        print(f"[hf baseline +{time.perf_counter() - t_start:.1f}s] {msg}", file=sys.stderr, flush=True)

    vd, td = (32, 36) if full_depth else (8, 2)
    hc = Qwen2_5_VLConfig(
        vision_config=dict(depth=vd, hidden_size=1280, num_heads=16, intermediate_size=3420, out_hidden_size=2048,
                           patch_size=14, temporal_patch_size=2, spatial_merge_size=2, window_size=112, in_channels=3,
                           fullatt_block_indexes=[7, 15, 23, 31] if full_depth else [7]),
        text_config=dict(hidden_size=2048, num_hidden_layers=td, num_attention_heads=16, num_key_value_heads=2,
                         intermediate_size=11008, vocab_size=151936, rms_norm_eps=1e-6, max_position_embeddings=32768,
                         rope_parameters=dict(rope_type="default", rope_theta=1e6, mrope_section=[16, 24, 24]),
                         eos_token_id=[151645, 151643], pad_token_id=151643, bos_token_id=None),
        image_token_id=151655, video_token_id=151656, vision_start_token_id=151652, vision_end_token_id=151653,
        tie_word_embeddings=True)
    hc._attn_implementation = "sdpa"
    torch.set_default_dtype(torch.bfloat16)  # parameters are created in bf16 directly (7.5 GB, not 15)
    with no_init_weights():
        model = Qwen2_5_VLForConditionalGeneration(hc)
    torch.set_default_dtype(torch.float32)
    model = model.to(torch.bfloat16).eval()
    with torch.no_grad():
        # random values from one 64-M-element pool copied into every matrix (values do not matter for the timing; a
        # per-tensor uniform_ on 3.75 G bf16 elements alone takes over a minute on one core)
        pool = torch.empty(1 << 26, dtype=torch.bfloat16).uniform_(-0.03, 0.03)
        for n, p in model.named_parameters():
            if p.dim() >= 2:
                flat = p.view(-1)
                for o in range(0, flat.numel(), pool.numel()):
                    m = min(pool.numel(), flat.numel() - o)
                    flat[o:o + m].copy_(pool[:m])
            elif "norm" in n or n.endswith("ln_q.weight"):
                p.fill_(1.0)
            else:
                p.zero_()
        del pool
    note(f"model built (ViT depth {vd}, {td} decoder layers)")
    rng = np.random.default_rng(0)
    pv1 = torch.from_numpy(rng.standard_normal((1296, 1176), dtype=np.float32)).to(torch.bfloat16)
    pv2 = torch.from_numpy(rng.standard_normal((2592, 1176), dtype=np.float32)).to(torch.bfloat16)
    txt_a, txt_b = list(rng.integers(1000, 150000, 21)), list(rng.integers(1000, 150000, 455))
    vis = [151652] + [151655] * 324 + [151653]
    ids1 = torch.tensor([txt_a + vis + txt_b])
    ids2 = torch.tensor([txt_a + vis + txt_b + list(rng.integers(1000, 150000, N1)) + vis])
    assert ids1.shape[1] == 802 and ids2.shape[1] == 1320

    def kw(ids, pv, n_img):
        return dict(input_ids=ids, attention_mask=torch.ones_like(ids), pixel_values=pv,
                    image_grid_thw=torch.tensor([[1, 36, 36]] * n_img), mm_token_type_ids=(ids == 151655).int())

    def gen(ids, pv, n_img, n_new):
        t0 = time.perf_counter()
        with torch.no_grad():
            out = model.generate(**kw(ids, pv, n_img), max_new_tokens=n_new, min_new_tokens=n_new, do_sample=False,
                                 num_beams=1, repetition_penalty=PENALTY)
        assert out.shape[1] == ids.shape[1] + n_new
        return time.perf_counter() - t0

    base = {"unit": "questions/s", "cores": ncpu, "threads": threads, "kind": "reference"}
    if full_depth:
        gen(ids1, pv1, 1, 2)  # warm-up: allocator, oneDNN primitive caches
        note("warm-up done")
        t1 = gen(ids1, pv1, 1, N1)
        note(f"stage 1: {t1:.1f}s")
        t2 = gen(ids2, pv2, 2, N2)
        note(f"stage 2: {t2:.1f}s")
        return dict(base, value=1.0 / (t1 + t2),
                    sample=(f"ONE whole question at full depth, as executed by the reference: installed transformers "
                            f"{transformers.__version__} Qwen2_5_VLForConditionalGeneration (src/eval/infer.py:147-151), bf16, "
                            f"random weights, 3B shape (32 ViT blocks, 36 layers), torch on {threads} threads of {ncpu} "
                            f"cores: stage 1 generate (802-token prompt, 1296 patches, {N1} new tokens) {t1:.1f}s + stage 2 "
                            f"generate (1320-token prompt, 2592 patches, {N2} new tokens) {t2:.1f}s = {t1 + t2:.1f}s; not extrapolated"))
    # bounded layer sample, extrapolated by layer count
    n_full, depth_full, layers_full = 4, 32, 36
    k = 24
    with torch.no_grad():
        visual = model.model.visual
        blocks = visual.blocks
        visual(pv1, grid_thw=torch.tensor([[1, 36, 36]]))

        def best(fn, n=3):
            ts = []
            for _ in range(n):
                t0 = time.perf_counter()
                fn()
                ts.append(time.perf_counter() - t0)
            return min(ts)

        t_depth = {}
        for d in (1, 7, 8):  # blocks 0..6: window attention; block 7: full attention
            visual.blocks = blocks[:d]
            t_depth[d] = best(lambda: visual(pv1, grid_thw=torch.tensor([[1, 36, 36]])), 5)
        visual.blocks = blocks
        t_win = max(t_depth[7] - t_depth[1], 0.0) / 6
        t_fullblk = max(t_depth[8] - t_depth[7], 0.0)
        t_over = max(t_depth[1] - t_win, 0.0)
        t_pre_all = best(lambda: model(**kw(ids1, pv1, 1), use_cache=True, logits_to_keep=1), 2)
        h = torch.randn(1, 2048).to(torch.bfloat16)
        t_head = best(lambda: model.lm_head(h), 4)
        t_pre = max(t_pre_all - t_depth[8] - t_head, 0.0)
        g1 = min(gen(ids1, pv1, 1, 1), gen(ids1, pv1, 1, 1))
        gk = min(gen(ids1, pv1, 1, 1 + k), gen(ids1, pv1, 1, 1 + k))
        t_dec = max(gk - g1, 0.0) / k
    vit_full = t_over + t_win * (depth_full - n_full) + t_fullblk * n_full
    t_question = (3 * vit_full + t_pre / td * layers_full * (802 + 1320) / 802 + 2 * t_head
                  + (N1 + N2) * (max(t_dec - t_head, 0.0) / td * layers_full + t_head))
    return dict(base, value=1.0 / t_question,
                sample=(f"EXTRAPOLATED from a layer sample: installed transformers {transformers.__version__}, bf16, random "
                        f"weights, torch on {threads} threads of {ncpu} cores: 8 of 32 ViT blocks over 1296 patches (window "
                        f"block {t_win:.4f}s, full-attention block {t_fullblk:.4f}s, embed+merger {t_over:.4f}s), 2 of 36 "
                        f"decoder layers prefilling 802 tokens ({t_pre:.3f}s), {k} generate() decode steps ({t_dec:.4f}s each "
                        f"incl. lm_head {t_head:.4f}s); scaled by layer count to the as-executed question = {t_question:.1f}s"))


def cpu_baseline_hf_bounded(full_depth: bool, timeout_s: float):
    """cpu_baseline_hf in a child process under a wall-clock limit: a host where the transformers CPU path crawls
    (thread oversubscription, bf16 emulation) must not stall the benchmark; the caller falls back."""
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--hf-baseline-child", "full" if full_depth else "sample"],
                       capture_output=True, text=True, timeout=timeout_s,
                       env=dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""))
    for ln in reversed(r.stdout.strip().splitlines()):
        if ln.startswith("{"):
            return json.loads(ln)
    raise RuntimeError(f"child failed: {r.stderr.strip()[-200:]}")


def measure_cpu_baseline(mode: str):
    notes = []
    if mode in ("auto", "hf"):
        for full, limit in ((True, float(os.environ.get("ZE_HF_FULL_TIMEOUT", "420"))), (False, 150.0)):
            try:
                c = cpu_baseline_hf_bounded(full, limit)
                if notes:
                    c["sample"] += " (" + "; ".join(notes) + ")"
                return c
            except Exception as ex:
                notes.append(f"{'full-depth question' if full else 'layer sample'} unavailable: {type(ex).__name__}: {str(ex)[:100]}")
    try:
        c = cpu_baseline()
        if notes:
            c["sample"] += " (" + "; ".join(notes) + ")"
        return c
    except Exception as ex:  # pragma: no cover
        return {"value": None, "unit": "questions/s", "cores": os.cpu_count(), "threads": None, "kind": "port",
                "sample": f"failed: {ex}; " + "; ".join(notes)}


# ----------------------------------------------------------------------------- launcher
def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` outside torchrun: start the N ranks as child processes (one per GPU) BEFORE this
    process touches the GPU, and exit with their worst return code; rank 0 prints the JSON line."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


def main():
    if "--hf-baseline-child" in sys.argv:
        print(json.dumps(cpu_baseline_hf(full_depth=sys.argv[-1] != "sample")), flush=True)
        return 0
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: 16 stream steps of 64 questions per GPU; 4 with --batch)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed warm-up steps (default 4; 1 with --batch)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline", choices=("auto", "hf", "port"), default="auto",
                    help="auto: one full-depth question through the installed transformers model (the reference's CPU path), "
                         "else its layer sample, else the oracle port")
    ap.add_argument("--tile", type=int, default=5000)
    ap.add_argument("--model", choices=("3b", "7b"), default="3b",
                    help="3b = ZoomEarth-3B shape (the metric's workload); 7b = Qwen2.5-VL-7B backbone swap of BASELINE "
                         "configs[4]: reported without the 3B roofline objects")
    ap.add_argument("--fp8", action="store_true",
                    help="FP8 (E4M3) decoder weights (BASELINE configs[4]); NOT the metric's precision: reported as its own "
                         "configuration")
    ap.add_argument("--fp8-act", action="store_true",
                    help="--fp8 plus FP8 activations at the qkv / gate-up inputs (ze_set_fp8_activations: fp8 x fp8 MFMA in the "
                         "batched decode step)")
    ap.add_argument("--batch", type=int, default=0,
                    help="0 (default) = the configs[3] question stream is the line's value; 1 = BASELINE configs[1] (one chain, "
                         "one question per step); B > 1 = configs[2] with B chains (one step = B questions)")
    ap.add_argument("--slots", type=int, default=int(os.environ.get("ZE_STREAM_SLOTS", str(STREAM_SLOTS))),
                    help="chain slots per GPU of the stream workload")
    ap.add_argument("--lanes", type=int, default=int(os.environ.get("ZE_LANES", "2")),
                    help="engines per GPU of the stream workload, each with its own scheduler thread and HIP stream (prefill of one "
                         "overlaps decode of the other; `src/eval/infer.py --lanes`); --slots chain slots EACH.  Measured on the "
                         "1280-question stream: 1 x 512 slots 57.7, 1 x 1024 60.3, 2 x 256 62.9, 2 x 512 66.2, 2 x 768 67.6, 3 x 512 65.4 questions/s")
    ap.add_argument("--no-batch64", action="store_true", help="skip the batch64 object of the default N = 1 line")
    ap.add_argument("--no-configs1", action="store_true", help="skip the configs1 object of the default N = 1 line")
    ap.add_argument("--no-reuse-sensitivity", action="store_true", help="skip the two short streams behind `value_without_generated_row_reuse`")
    ap.add_argument("--spawn-check", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    args.fp8 = args.fp8 or args.fp8_act
    stream = args.batch == 0
    if args.steps is None:
        args.steps = 16 if stream else 4
    if args.warmup is None:
        args.warmup = 4 if stream else 1

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus)
    if args.spawn_check:  # launcher self-test (tests/test_hostlayer_cpu.py): what this rank was started with, no GPU touched
        if int(os.environ.get("RANK", "0")) == 0:
            print(json.dumps({"n_gpus": int(os.environ.get("WORLD_SIZE", "1")), "rank": 0,
                              "master": os.environ.get("MASTER_ADDR"), "local_rank": os.environ.get("LOCAL_RANK")}), flush=True)
        return 0

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    use_dist = world > 1 or os.environ.get("ZE_BENCH_FORCE_DIST") == "1"  # the env flag exercises RCCL at N=1
    # ZE_DIST_BACKEND=gloo: the same N > 1 code path (sharding, the weight broadcast into ranks that never filled their arena, the
    # barriers, the MAX / SUM reductions of the report) with the ranks SHARING a GPU -- RCCL wants one rank per GPU, and the
    # builder's box has one: tests/test_gpu_bench_contract.py runs `--gpus 2` this way.  Default: nccl = RCCL.
    backend = os.environ.get("ZE_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local = local % max(1, torch.cuda.device_count())
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    torch.cuda.set_device(local)
    red_dev = f"cuda:{local}" if backend == "nccl" else "cpu"  # where the report's reductions live

    from zoomearth_amd.accel import broadcast_engine_weights, shard_by_tile
    from zoomearth_amd.config import ModelConfig
    from zoomearth_amd.engine import Engine
    from zoomearth_amd.image import DeviceImage
    from zoomearth_amd.synth import synthetic_tile

    B = max(1, args.batch)
    SLOTS = max(1, args.slots)
    want1 = stream and world == 1 and args.model == "3b" and not args.fp8 and not args.no_configs1
    want64 = stream and world == 1 and args.model == "3b" and not args.fp8 and not args.no_batch64
    cfg = ModelConfig.zoomearth_3b() if args.model == "3b" else ModelConfig.qwen25vl_7b()
    chains = SLOTS if stream else B
    e = Engine(cfg, device=local, max_seqs=chains, max_ctx=2048, max_patches=max(4096, 1400 * min(max(chains, 64 if want64 else 1), 40)),
               max_prefill_rows=(int(os.environ.get("ZE_PREFILL_ROWS", str(32 * 832 if chains > 64 else 16 * 832))) if chains > 1 else 0),
               max_tile_side=max(args.tile, 1024))
    filled_here = rank == 0 or os.environ.get("ZE_BENCH_EVERY_RANK_FILLS") == "1"
    if filled_here:
        e.fill_synthetic(seed=0, std=0.02)
    for kv in os.environ.get("ZE_TUNE", "").split(","):  # measurement-only A/B knobs, e.g. ZE_TUNE=2:64
        if ":" in kv:
            e.lib.ze_tune(int(kv.split(":")[0]), int(kv.split(":")[1]))
    # the path's only collective: rank 0 holds the weights (a checkpoint read in production, the synthetic fill here), the
    # other ranks receive the packed arena once over RCCL / xGMI (accel.broadcast_engine_weights, what
    # ZoomEarthForConditionalGeneration.from_pretrained(..., broadcast=True) and src/eval/infer.py run)
    bcast_s = broadcast_engine_weights(e, rank, world, src=0, force=True) if use_dist else 0.0
    e.assert_ready()
    if args.fp8:
        e.quantize_fp8()  # after the broadcast: every rank quantises the weights it received
        if args.fp8_act:
            e.set_fp8_activations(True)
    engines = [e]
    for _ in range(1, max(1, args.lanes) if stream else 1):  # further lanes: engines of their own, weights copied on the device
        e2 = Engine(cfg, device=local, max_seqs=chains, max_ctx=2048, max_patches=max(4096, 1400 * min(chains, 40)),
                    max_prefill_rows=int(os.environ.get("ZE_PREFILL_ROWS", str(32 * 832))), max_tile_side=max(args.tile, 1024))
        e2.weights_arena().copy_(e.weights_arena())
        torch.cuda.synchronize()
        e2.weights_invalidate()
        engines.append(e2)
    # (several engines on one GPU: the eight-phase prefill GEMM launches one tile per workgroup by itself -- the library counts its
    #  live engines, zoomearth.h knob 4 -- no process-wide switch is flipped here any more)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    # ---- the workload's tiles: a small pool of distinct synthetic host tiles (generating one takes ~1 s of host time),
    # every logical tile of the rank uploaded from the pool to its own HBM buffer (pinned host memory -> HBM through the C
    # ABI, timed on its own, never inside `value`: tiles are resident when the timed region starts)
    n_pool = 6 if (stream or B > 1 or want64) else 1
    host_pool = [torch.from_numpy(synthetic_tile(1000 + rank * 16 + t, args.tile, args.tile)).pin_memory() for t in range(n_pool)]

    upload_stats = {"n": 0, "s": 0.0}

    def upload(t):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        d = e.tile_upload(host_pool[t % n_pool])
        torch.cuda.synchronize()
        upload_stats["n"] += 1
        upload_stats["s"] += time.perf_counter() - t0
        return d

    line = {}
    bstats = {}
    use_graph = not args.no_graph
    sched_graph = False if args.no_graph else None  # (None: the scheduler's own choice -- graphs up to 64 chain slots only)
    if stream:
        # global question table of the timed region: world x steps x 64 questions, tiles assigned to ranks whole
        n_total = world * args.steps * Q_STEP
        tile_of = question_table(n_total, seed=1)
        mine = shard_by_tile([f"tile{t:05d}.tif" for t in tile_of], rank, world)
        warm_tile_of = question_table(args.warmup * Q_STEP, seed=2)
        my_tiles = sorted({tile_of[i] for i in mine})
        # Round 5: the tiles are NOT resident when the timed region starts.  Every lane uploads its tiles itself, pinned host pool ->
        # HBM on a stream of its own, three tiles ahead of their first question (TileFeeder): `value` includes the uploads, as
        # SURVEY 8d's metric does ("tile decoded in host RAM -> answer token ids").  ZE_BENCH_RESIDENT_TILES=1: round 4's form.
        resident = os.environ.get("ZE_BENCH_RESIDENT_TILES") == "1"
        if resident:
            dev = {t: DeviceImage(upload(t), e) for t in my_tiles}
            table = [(i, dev[tile_of[i]], tile_of[i]) for i in mine]
        else:
            table = [(i, tile_of[i], tile_of[i]) for i in mine]
        warm_table = [(i, ("w", t), ("w", t)) for i, t in enumerate(warm_tile_of)]
        if warm_table:
            run_stream_lanes(engines, warm_table, 9_000_000 + rank * 100_000, SLOTS, use_graph=sched_graph, host_pool=host_pool)
        for en in engines:
            en.phase_timers(enable=True, reset=True)
        barrier()
        t0 = time.perf_counter()
        run_stream_lanes(engines, table, 1_000_000, SLOTS, bstats, use_graph=sched_graph, host_pool=None if resident else host_pool)
        torch.cuda.synchronize()
        my_dt = time.perf_counter() - t0
        barrier()
        dt = time.perf_counter() - t0
        if not resident:  # (the annexes below look at one resident tile)
            upload_stats["n"], upload_stats["s"] = int(bstats.pop("tiles_uploaded", 0)), bstats.pop("tile_upload_ms_total", 0.0) / 1000.0
            dev = {t: DeviceImage(e.tile_upload(host_pool[t % n_pool]), e) for t in my_tiles[:6]}
        n_questions = n_total
        lens_all = bstats.pop("lens")
        # how much of `value` rests on the synthetic tokenizer's identity round trip.  The stream keeps the K/V
        # rows the decode steps wrote for a stage-1 output while the re-tokenised ids repeat the generated ones -- with the decimal-word
        # tokenizer of this bench they always do (191 of 192 rows), with a real BPE they stop at the first re-merged piece.  Two short
        # streams of the warm-up's size, back to back on the warm engines, one with the reuse on and one with it off (every stage-2
        # prompt prefilled from the end of the cached stage-1 PROMPT: bit-identical to a full prefill): their ratio scales `value` to
        # the figure without any generated-row reuse -- the lower bound for any tokenizer.
        reuse_sens = None
        if world == 1 and not args.no_reuse_sensitivity and args.warmup > 0:
            global REUSE_GENERATED
            n_short = max(4, args.warmup) * Q_STEP
            short_of = question_table(n_short, seed=3)
            qps = {}
            for flag in (True, False):
                REUSE_GENERATED = flag
                short = [(i, ("s1" if flag else "s0", t), ("s1" if flag else "s0", t)) for i, t in enumerate(short_of)]
                st_s = {}
                torch.cuda.synchronize()
                ts = time.perf_counter()
                run_stream_lanes(engines, short, (7_000_000 if flag else 8_000_000), SLOTS, st_s, use_graph=sched_graph, host_pool=host_pool)
                torch.cuda.synchronize()
                qps[flag] = (n_short / (time.perf_counter() - ts), st_s.get("prefill_rows", 0) / n_short, st_s.get("reused_generated_rows", 0) / n_short)
            REUSE_GENERATED = True
            reuse_sens = {"questions_per_stream": n_short, "with_reuse_qps": round(qps[True][0], 2), "without_reuse_qps": round(qps[False][0], 2),
                          "ratio": round(qps[False][0] / qps[True][0], 4),
                          "prefill_rows_per_question": [round(qps[True][1], 1), round(qps[False][1], 1)],
                          "generated_rows_kept_per_question": [round(qps[True][2], 1), round(qps[False][2], 1)],
                          "note": ("two short streams back to back on the warm engines (fill and drain weigh more in them than in the timed "
                                   "stream: compare the ratio, not the absolute figures); without the reuse every stage-2 prompt is prefilled "
                                   "from the end of the cached stage-1 prompt, bit-identical to a full prefill")}
        lens = (int(np.mean([l[0] for l in lens_all])), int(np.mean([l[2] for l in lens_all])),
                float(np.mean([l[1] for l in lens_all])), float(np.mean([l[3] for l in lens_all])))
    else:
        dev_tiles = [upload(t) for t in range(max(1, round(B * 6 / 64)) if B > 1 else 1)]
        tiles64 = [DeviceImage(t, e) for t in dev_tiles]
        chain = Chain(e, dev_tiles[0], use_graph=use_graph)
        q0 = rank * 100000

        def run_step(q):
            if B == 1:
                return chain.question(q)
            d = batch_step(e, tiles64, q * B, B, bstats, use_graph=sched_graph)
            l = d[q * B]
            return [0] * l[1], [0] * l[3], l[0], l[2]

        for i in range(args.warmup):
            run_step(q0 + i)
        e.phase_timers(enable=True, reset=True)
        barrier()
        t0 = time.perf_counter()
        lens = None
        for i in range(args.steps):
            out1, out2, l1, l2 = run_step(q0 + args.warmup + i)
            lens = (l1, l2, len(out1), len(out2))
        torch.cuda.synchronize()
        my_dt = time.perf_counter() - t0
        barrier()
        dt = time.perf_counter() - t0
        n_questions = world * args.steps * B
        mine = list(range(args.steps * B))
    phases = e.phase_timers(enable=False)
    for en in engines[1:]:
        for k, v in en.phase_timers(enable=False).items():
            phases[k] += v
    per_rank = [[len(mine), my_dt, 1.0 if filled_here else 0.0]]
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        g = torch.zeros((world, 3), dtype=torch.float64, device=red_dev)
        g[rank, 0], g[rank, 1], g[rank, 2] = len(mine), my_dt, 1.0 if filled_here else 0.0
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
        per_rank = g.cpu().tolist()

    def batch_roofline(n, shared=None):
        """dominant kernel of the batched decode step at n chains (largest device time per step among its launches),
        measured live with HIP events on the launch stream; the chains hold the contexts the run left behind.
        shared = (chains per tile, rows): the attention is timed TWICE -- with independent chains and with the sharing the
        question stream has (the chains of a tile read their common prompt prefix from one holder's cache:
        ze_seq_dev::prefix; declared here with ze_seq_set_prefix_hint, the launch's results are discarded) -- and the
        second is the object's `achieved`: same algorithmic bytes (every chain attends over all its rows), fewer of them
        from HBM (`traffic`)."""
        wide = e.set_decode_regime(1 if n > 64 else 0) == 1
        names = BATCH_KERNEL_NAMES_WIDE if wide else BATCH_KERNEL_NAMES
        rows = {}
        for which in sorted(BATCH_KERNELS):
            if which == 7:
                continue  # (both families fold rope + KV append into the qkv launch: round 4 for the row-streaming one)
            u, by = e.profile_batch_kernel(which, n, iters=72)
            rows[BATCH_KERNELS[which]] = {"us": round(u, 2), "GBps": round(by / (u * 1e-6) / 1e9, 1), "bytes": by}
        per_step = {k: v["us"] * (2 if k == "rmsnorm" else 1) for k, v in rows.items() if k != "lm_head"}
        dom = max((w for w in BATCH_KERNELS if BATCH_KERNELS[w] in per_step and w in names), key=lambda w: per_step[BATCH_KERNELS[w]])
        r = rows[BATCH_KERNELS[dom]]
        ach = r["bytes"] / (r["us"] * 1e-6) / 1e9
        layer_us = sum(per_step.values())
        traffic, traffic_src = committed_traffic("stream" if wide else "batch64", BATCH_KERNELS[dom], r["bytes"])
        obj = {"bound": "hbm", "kernel": names[dom], "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
               # what crosses the HBM interface (the PMC traffic, Infinity-Cache hits included) per second against the peak: below
               # `frac` where chains share rows, above it where a kernel re-reads
               "hbm_interface_frac": (traffic / (r["us"] * 1e-6) / 1e9 / HBM_PEAK_GBS) if traffic else None, "avg_us": r["us"],
               "bytes_per_launch": r["bytes"], "chains": n, "layer_us": round(layer_us, 1),
               "step_kernels": {k: {"us": v["us"], "GBps": v["GBps"]} for k, v in rows.items()}}
        if shared and BATCH_KERNELS[dom] == "attention" and hasattr(e, "seq_set_prefix_hint"):
            group, prows = shared
            hinted = 0
            for s_ in range(n):
                lead = s_ - s_ % group
                if s_ != lead and min(e.seq_len(s_), e.seq_len(lead)) >= prows:
                    e.seq_set_prefix_hint(s_, lead, prows)
                    hinted += 1
            u2, by2 = e.profile_batch_kernel(5, n, iters=72)
            for s_ in range(n):
                e.seq_set_prefix_hint(s_, s_, 0)
            if hinted:
                ach2 = by2 / (u2 * 1e-6) / 1e9
                t2, t2_src = committed_traffic("stream_shared", "attention", by2)
                obj["independent_chains"] = {"achieved": ach, "frac": ach / HBM_PEAK_GBS, "avg_us": r["us"], "traffic": traffic,
                                             "traffic_source": traffic_src}
                obj["independent_chains"]["hbm_interface_frac"] = obj.get("hbm_interface_frac")
                obj.update(achieved=ach2, frac=ach2 / HBM_PEAK_GBS, avg_us=round(u2, 2), traffic=t2, traffic_source=t2_src,
                           hbm_interface_frac=(t2 / (u2 * 1e-6) / 1e9 / HBM_PEAK_GBS) if t2 else None,
                           layer_us=round(layer_us - r["us"] + u2, 1),
                           sharing=(f"{hinted} of the {n} chains read their first {prows} rows (system turn + the view's image tokens) from the "
                                    f"cache of the first chain of their tile ({group} chains per tile, as the stream's 10.5), "
                                    "ze_seq_dev::prefix: the algorithmic bytes are those of independent chains, the prefix "
                                    "crosses the HBM interface once per tile and step"))
                obj["step_kernels"]["attention"] = {"us": round(u2, 2), "GBps": round(ach2, 1)}
        e.set_decode_regime(-1)
        return obj

    def live_like_contexts(n):
        """The stream's end state holds finished stage-2 chains (~1416 cached rows each); a LIVE step holds two stage-1 chains
        (802 .. 994 rows) for every stage-2 chain (1320 .. 1416): mean ~1055.  Cut the first n chains back to lengths drawn like
        that before the decode kernels are timed (the roofline launch ran on the contexts the stream left
        behind).  Returns the mean context of the n chains."""
        tot = 0
        for s_ in range(n):
            want = (L_TEXT_A + 2 + 324 + L_TEXT_B + (s_ * 37) % N1) if s_ % 3 != 2 else (1320 + (s_ * 53) % N2)
            have = e.seq_len(s_)
            if have > want:
                e.seq_truncate(s_, want)
                have = want
            tot += have
        return tot / max(1, n)

    def stream_rooflines(st, roof, live, n_q_rank, wall_s):
        """The whole-stream roofline objects of the line, from ISOLATED kernel-time accounting -- the stream's own
        phase timers sum two lanes' overlapping stream time and cannot be turned into a fraction:
          decode   the step's kernels timed alone at the mean live-chain count on live-like contexts (batch_roofline): decode
                   steps x (layers x layer_us + lm_head) per question, against the algorithmic bytes of those steps at 8 TB/s
                   (weights once per step for all chains of the step + every chain's K/V rows: SURVEY 8d's B_decode);
          prefill  a replay of representative admission passes alone on the GPU (16 chains x 802 new rows, then 330 more rows
          / vit    behind a cached prompt) and of a 24-image ViT call, scaled to the rows / patches the stream actually executed
                   (scheduler stats), against the FLOPs of those rows at 2.5 PFLOP/s;
          question SURVEY 8d's formula at the measured mean batch: T_roof = 12.8 TFLOP / 2.5 PFLOP/s + (6.171 GB x 288 / chains
                   + 11.2 GB) / 8 TB/s against the measured wall time per question of this GPU."""
        import torch
        cfgt = cfg.text
        steps_total = max(1, st.get("steps", 1))
        mean_chains = st.get("chain_steps", 0) / steps_total
        # ---- decode: the steps are spread over a wide range of live-chain counts (two lanes x 768 slots fill and drain: most
        # chain-steps run at 513-640 chains, the MEAN step has ~410), and the step's GEMMs pick their tiles by the row count --
        # so the step is timed alone at a representative count of every bucket of the scheduler's histogram (`steps_le_<bound>`)
        # and weighted by the steps the stream ran there; `step_us_at_mean_chains` stays for reference
        step_us = cfgt.num_hidden_layers * roof["layer_us"] + roof["step_kernels"]["lm_head"]["us"]
        buckets, dec_us_total, lo = [], 0.0, 1
        for bound in (64, 128, 192, 256, 320, 384, 416, 448, 512, 640, 768):
            nsteps = st.get(f"steps_le_{bound}", 0)
            if nsteps > 0:
                n_rep = int(min(e.max_seqs, (lo + bound) // 2))
                if n_rep > 64:
                    live_like_contexts(n_rep)
                    rb = batch_roofline(n_rep, shared=(10, 347))
                    us_b = cfgt.num_hidden_layers * rb["layer_us"] + rb["step_kernels"]["lm_head"]["us"]
                else:   # (the drain tail: few chains on the row-streaming family's small-row kernels)
                    rb = batch_roofline(max(n_rep, 65), shared=(10, 347))
                    us_b = cfgt.num_hidden_layers * rb["layer_us"] + rb["step_kernels"]["lm_head"]["us"]
                buckets.append({"chains": f"{lo}-{bound}", "timed_at": max(n_rep, 65) if n_rep <= 64 else n_rep, "steps": nsteps,
                                "step_us": round(us_b, 1), "layer_us": rb["layer_us"]})
                dec_us_total += nsteps * us_b
            lo = bound + 1
        if not buckets:
            dec_us_total = steps_total * step_us
        dec_ms_q = dec_us_total / 1000.0 / n_q_rank
        w_bytes = 2.0 * (2774532096 + 311164928)
        kv_row = 36864.0
        kv_bytes_q = kv_row * (N1 * (L_TEXT_A + 2 + 324 + L_TEXT_B + N1 / 2.0) + N2 * (1320 + N2 / 2.0))
        dec_bytes_q = w_bytes * (N1 + N2) / max(1.0, mean_chains) + kv_bytes_q
        dec_roof_ms = dec_bytes_q / (HBM_PEAK_GBS * 1e9) * 1000.0
        # ---- replay: prefill passes and a ViT call alone on the GPU
        view = next(iter(dev.values())).resize((512, 512))
        pv, grid = e.preprocess_image(view.tensor())
        n_img = grid[1] * grid[2] // 4
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        k_img = 24 if 24 * pv.shape[0] <= e.max_patches else max(1, e.max_patches // pv.shape[0])
        pvs = torch.cat([pv] * k_img).contiguous()
        e.vit_forward(pvs, [grid] * k_img)  # warm-up
        ev[0].record()
        feats = e.vit_forward(pvs, [grid] * k_img)
        ev[1].record()
        nrep = min(16, e.max_seqs)
        ids1 = [question_ids(cfg, 77_000 + c, n_img) for c in range(nrep)]
        extra = [[1000 + c] * 4 + [cfg.vision_start_token_id] + [cfg.image_token_id] * n_img + [cfg.vision_end_token_id] for c in range(nrep)]
        pl = [e.rope_index(ids1[c] + extra[c], [grid, grid]) for c in range(nrep)]
        emb = feats[:n_img]

        def passes():
            for c in range(nrep):
                e.seq_reset(c)
            ev[2].record()
            e.prefill_batch(list(range(nrep)), ids1, [emb] * nrep, [pl[c][0][:, :len(ids1[c])] for c in range(nrep)], [pl[c][1] for c in range(nrep)])
            ev[3].record()
            e.prefill_batch(list(range(nrep)), extra, [emb] * nrep, [pl[c][0][:, len(ids1[c]):] for c in range(nrep)], [pl[c][1] for c in range(nrep)])
            ev[4].record()
        passes()
        passes()
        torch.cuda.synchronize()
        vit_ms_img = ev[0].elapsed_time(ev[1]) / k_img
        rows_a, rows_b = sum(len(x) for x in ids1), sum(len(x) for x in extra)
        pre_ms_row = (ev[2].elapsed_time(ev[3]) + ev[3].elapsed_time(ev[4])) / (rows_a + rows_b)
        imgs_q = st.get("vit_images", 0) / n_q_rank
        patches_q = st.get("vit_patches", 0) / n_q_rank
        rows_q = st.get("prefill_rows", 0) / n_q_rank
        vit_ms_q = vit_ms_img * imgs_q
        pre_ms_q = pre_ms_row * rows_q
        f_vit_q = patches_q * 1.2813e9 + imgs_q * 45.0e9          # per patch 1.2813 GFLOP, attention 45 GFLOP per 36 x 36 view
        f_pre_q = rows_q * (2.0 * 2774532096 + 36 * 4 * 2048 * 870.0)  # linear layers + causal attention at the rows' mean context
        mfma = 2500.0e12
        wall_ms_q = 1000.0 * wall_s / n_q_rank
        t_roof_ms = (12.8e12 / mfma + (w_bytes * (N1 + N2) / max(1.0, mean_chains) + 11.2e9) / (HBM_PEAK_GBS * 1e9)) * 1000.0
        iso = vit_ms_q + pre_ms_q + dec_ms_q
        return {
            "note": ("isolated kernel-time accounting (the kernels of each phase timed ALONE on this GPU, scaled to the work the stream "
                     "executed); the stream's own phase timers overlap two lanes and several HIP streams"),
            "vit": {"bound": "mfma", "ms_per_question": round(vit_ms_q, 3), "images_per_question": round(imgs_q, 3),
                    "achieved_TFLOPs": round(f_vit_q / (vit_ms_q * 1e-3) / 1e12, 1) if vit_ms_q > 0 else None, "peak_TFLOPs": 2500.0,
                    "frac": (f_vit_q / mfma) / (vit_ms_q * 1e-3) if vit_ms_q > 0 else None,
                    "replay": f"{k_img} views of 36 x 36 patches in one call: {vit_ms_img:.3f} ms per image"},
            "prefill": {"bound": "mfma", "ms_per_question": round(pre_ms_q, 3), "rows_per_question": round(rows_q, 1),
                        "achieved_TFLOPs": round(f_pre_q / (pre_ms_q * 1e-3) / 1e12, 1) if pre_ms_q > 0 else None, "peak_TFLOPs": 2500.0,
                        "frac": (f_pre_q / mfma) / (pre_ms_q * 1e-3) if pre_ms_q > 0 else None,
                        "replay": f"{nrep} chains x {len(ids1[0])} new rows, then {len(extra[0])} rows behind the cached prompt: {1000 * pre_ms_row:.2f} us per row"},
            "decode": {"bound": "hbm", "ms_per_question": round(dec_ms_q, 3), "step_us_at_mean_chains": round(step_us, 1),
                       "steps_by_live_chains": buckets,
                       "mean_chains_per_step": round(mean_chains, 1), "algorithmic_GB_per_question": round(dec_bytes_q / 1e9, 2),
                       "achieved_GBs": round(dec_bytes_q / (dec_ms_q * 1e-3) / 1e9, 1), "peak_GBs": HBM_PEAK_GBS,
                       "frac": dec_roof_ms / dec_ms_q},
            "isolated_ms_per_question": round(iso, 3),
            "question": {"roofline_ms": round(t_roof_ms, 3), "measured_ms": round(wall_ms_q, 3), "frac": t_roof_ms / wall_ms_q,
                         "formula": "SURVEY 8d: 12.8 TFLOP / 2.5 PFLOP/s + (6.171 GB x 288 / mean chains + 11.2 GB) / 8 TB/s",
                         "gpu_busy_with_isolated_kernel_time": round(iso / wall_ms_q, 3)},
        }

    def committed_traffic(section, kernel, alg_bytes=None):
        """HBM bytes per launch from the PMC passes of the latest committed profile (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
        cannot run inside this process): profiles/traffic_latest.json, written by tools/profile_digest.py from the passes
        of tools/profile_round3.sh around tools/pmc_kernel.py (the same launcher this process times).  Where the live
        launch's algorithmic bytes differ from the profiled launch's (the attention kernel: other contexts), the profiled
        traffic / algorithmic RATIO is applied to the live launch's bytes, and the source string says so."""
        tp = os.path.join(ROOT, "profiles", "traffic_latest.json")
        try:
            with open(tp) as f:
                tj = json.load(f)
            ent = tj[section][kernel]
            src = (f"profiles/traffic_latest.json (rocprofv3 --pmc FETCH_SIZE x 2 x 1024 + WRITE_SIZE x 1024 in separate passes, round "
                   f"{tj.get('round')}: {ent['hbm_bytes_per_launch'] / 1e6:.2f} MB per launch = {ent['ratio']:.3f} x algorithmic at "
                   f"{ent.get('chains')} chains)")
            # a figure from a committed profile goes stale silently when the kernels change after profiling -- the
            # profile records the hash of the library's sources it ran on, the line says whether this tree still has them
            from zoomearth_amd._lib import kernel_sources_sha16
            now, then = kernel_sources_sha16(), tj.get("kernel_sources_sha16")
            src += (f"; profiled on kernel sources {then}, this tree has {now}: " + ("the same" if then == now else "DIFFERENT -- the counter figure "
                    "predates later kernel changes") if then else f"; the profile predates the source hash (this tree: {now})")
            if alg_bytes and abs(alg_bytes / ent["algorithmic_bytes_per_launch"] - 1.0) > 0.01:
                return ent["ratio"] * alg_bytes, src + "; that ratio applied to this launch's algorithmic bytes"
            return ent["hbm_bytes_per_launch"], src
        except Exception:
            return None, None

    def top_kernel_by_gpu_time():
        """the line's `roofline` names the dominant kernel of the DECODE STEP; the stream's largest consumer of GPU
        time is another one.  Its name and share come from the latest committed rocprofv3 --kernel-trace --stats summary of this very
        command (profiles/rNN_stream_kernel_stats.csv: a profiler cannot run inside this process); its own roofline fraction is
        measured here, alone on the GPU, per SHAPE (the two pass sizes of the replay: 16 x 802 and 16 x 330 rows -- never a mean over
        both), in BOTH launch forms (persistent workgroups / one tile per workgroup: what several lanes sharing the GPU run), on the
        pass's own operands: the layers' weights in rotation and the activation rows the replayed passes left in the workspace
        (ze_profile_prefill_kernel).  `frac` = the form the stream runs, at the larger pass."""
        import csv
        import glob
        import re as _re
        best = None
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_stream_kernel_stats.csv"))):
            m = _re.match(r"r(\d+)_stream_kernel_stats", os.path.basename(f))
            if m and (best is None or int(m.group(1)) >= best[0]):
                best = (int(m.group(1)), f)
        if best is None:
            return None
        rows = [r for r in csv.DictReader(open(best[1])) if r.get("source") == "kernel_stats" and r.get("pct_time")]
        if not rows:
            return None
        top = max(rows, key=lambda r: float(r["pct_time"]))
        obj = {"kernel": top["kernel"].split("(")[0], "share_of_gpu_time": round(float(top["pct_time"]) / 100.0, 4),
               "mean_us_in_stream": round(float(top["mean"]) / 1000.0, 1), "share_source": "profiles/" + os.path.basename(best[1])}
        if "k_gemm_p8<3" in top["kernel"]:  # the prefill / ViT gate-up (SwiGLU epilogue) on the eight-phase 256 x 256 tiles
            shapes = {}
            in_stream = "tile_granular" if len(engines) > 1 else "persistent"
            for rows_p in (16 * (L_TEXT_A + 2 + 324 + L_TEXT_B), 16 * 330):
                if rows_p > e.max_prefill_rows:
                    continue
                ent = {}
                for form, knob in (("persistent", 2), ("tile_granular", 1)):
                    e.lib.ze_tune(4, knob)
                    lay = e.profile_prefill_layer(rows_p, 36)   # the four projections in pass order, each launch between its own events
                    us, fl = lay["gate_up"]
                    ent[form] = {"us": round(us, 1), "TFLOPs": round(fl / (us * 1e-6) / 1e12, 1), "frac": round(fl / (us * 1e-6) / 2.5e15, 4),
                                 "layer_us": {k: round(v[0], 1) for k, v in lay.items()},
                                 "layer_frac": {k: round(v[1] / (v[0] * 1e-6) / 2.5e15, 4) for k, v in lay.items()}}
                shapes[str(rows_p)] = ent
            e.lib.ze_tune(4, 2)
            big = shapes[max(shapes, key=int)]
            obj.update(bound="mfma", peak_TFLOPs=2500.0, form_in_stream=in_stream, frac=big[in_stream]["frac"],
                       achieved_TFLOPs=big[in_stream]["TFLOPs"], isolated_us=big[in_stream]["us"], by_rows=shapes,
                       shape=(f"gate/up of the replayed prefill passes: rows x {2 * cfg.text.intermediate_size} (gate/up interleaved) x "
                              f"{cfg.text.hidden_size}, SwiGLU epilogue; operands = the pass's own (post-norm rows of its last layer, the 36 "
                              "layers' weights in rotation), the layer's four projections issued in pass order with HIP events around every launch "
                              "(`layer_us`); FLOP = 2 x rows x N x K unpadded; `frac` is gate/up at the larger pass in the form the stream runs; "
                              "rocprofv3 of the replayed pass: profiles/r06_prefill_by_shape.csv"))
        return obj

    def configs1_object(steps, warmup):
        """BASELINE configs[1]: one chain, batch 1 -- `steps` questions through the single-chain path (GEMV decode under a
        captured hipGraph), with the roofline of ITS dominant kernel and the per-phase fractions"""
        tile = next(iter(dev.values())).base if stream else dev_tiles[0]
        ch = Chain(e, tile, use_graph=use_graph)
        for i in range(warmup):
            ch.question(5_000_000 + i)
        e.phase_timers(enable=True, reset=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        l = None
        for i in range(steps):
            o1, o2, l1, l2 = ch.question(5_000_100 + i)
            l = (l1, l2, len(o1), len(o2))
        torch.cuda.synchronize()
        d = time.perf_counter() - t0
        ph = e.phase_timers(enable=False)
        obj = {"workload": "BASELINE configs[1]: ZoomEarth-3B shape, one 5000x5000 tile, full two-stage zoom chain per question, "
                           "greedy, batch 1 (single-chain GEMV decode path, captured hipGraph)",
               "value": steps / d, "unit": "questions/s", "steps": steps, "warmup": warmup, "ms_per_question": 1000.0 * d / steps,
               "L1": l[0], "L2": l[1], "N1": l[2], "N2": l[3], "stage2_rows_kept_from_decode": ch.kept_generated,
               "phase_ms_per_question": {k: round(v / steps, 3) for k, v in ph.items()}}
        obj.update(decode_rooflines(obj["phase_ms_per_question"], l, ch.kept_generated, obj["ms_per_question"]))
        return obj

    def decode_rooflines(pm, l, kept_generated, ms_question):
        # roofline of the dominant kernel (decode gate/up weight stream), measured live with HIP events on the stream
        # the kernel runs on; bytes = algorithmic weight bytes of one launch (2 * 11008 * 2048 * 2 B)
        us, by = e.profile_decode_kernel(2, iters=144)
        ach = by / (us * 1e-6) / 1e9
        others = {}
        for which, name in ((0, "qkv"), (1, "o_proj"), (3, "down"), (4, "lm_head")):
            u, b = e.profile_decode_kernel(which, iters=72)
            others[name] = {"us": round(u, 2), "GBps": round(b / (u * 1e-6) / 1e9, 1)}
        traffic, traffic_src = committed_traffic("configs1", "gate_up")
        out = {"roofline": {"bound": "hbm", "kernel": "k_gemv<EPI=SWIGLU,PAIRS=1,KSPLIT=1,CH=4> = k_gemv<2, 1, 1, 4> (decode gate/up weight stream, 36 launches per token)",
                            "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                            "traffic_source": traffic_src, "avg_us": us, "bytes_per_launch": by, "other_decode_kernels": others}}
        # per-phase roofline fractions: algorithmic work of the as-built question (stage-1 prompt KV and view
        # features reused) over the measured phase time, against the dense bf16 MFMA peak / the HBM peak
        # (prefill: the rows actually prefilled -- the stage-1 prompt and what stage 2 appends beyond the kept rows)
        kept = l[0] + kept_generated
        f_vit = 3.41e12                                            # FLOP per question
        f_pre = (l[0] + l[1] - kept) * 5.549e9 + 36 * 4 * 2048 * (l[0] ** 2 + l[1] ** 2 - kept ** 2) / 2
        b_dec = (N1 + N2) * 6.171e9 + 36864.0 * (N1 * (l[0] + N1 / 2) + N2 * (l[1] + N2 / 2))  # bytes per question
        ph = {
            "vit": {"bound": "mfma", "achieved_TFLOPs": f_vit / (pm["vit"] * 1e-3) / 1e12, "peak_TFLOPs": 2500.0,
                    "frac": f_vit / (pm["vit"] * 1e-3) / 2.5e15},
            "prefill": {"bound": "mfma", "achieved_TFLOPs": f_pre / (pm["prefill"] * 1e-3) / 1e12, "peak_TFLOPs": 2500.0,
                        "frac": f_pre / (pm["prefill"] * 1e-3) / 2.5e15},
            "decode": {"bound": "hbm", "achieved_GBs": b_dec / (pm["decode"] * 1e-3) / 1e9, "peak_GBs": HBM_PEAK_GBS,
                       "frac": b_dec / (pm["decode"] * 1e-3) / (HBM_PEAK_GBS * 1e9)},
            "question": {"roofline_ms": (f_vit + f_pre + (N1 + N2) * 6.171e9) / 2.5e15 * 1e3 + b_dec / (HBM_PEAK_GBS * 1e9) * 1e3,
                         "measured_ms": ms_question},
        }
        ph["question"]["frac"] = ph["question"]["roofline_ms"] / ph["question"]["measured_ms"]
        out["roofline_phases"] = ph
        return out

    if rank == 0:
        n_up = max(1, upload_stats["n"])
        line = {
            "metric": "questions/sec end-to-end, ZoomEarth-3B on 5000px tiles" if args.model == "3b" else
                      "questions/sec end-to-end, Qwen2.5-VL-7B shape (bf16) on 5000px tiles", "value": n_questions / dt,
            "unit": "questions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1000.0 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": (f"BASELINE configs[3] (LRS-GRO question stream, sharded by tile across the GPUs; ZoomEarth-3B shape): "
                                    f"{n_questions} questions about {len(set(tile_of))} tiles of {args.tile}x{args.tile} px ({Q_STEP} questions per GPU "
                                    f"per step), full two-stage zoom chain per question, greedy, {len(engines)} lane(s) x {SLOTS} chain slots per GPU on the "
                                    f"continuous-batching scheduler (the path of src/eval/infer.py --lanes)") if stream else
                                   ("BASELINE configs[1]: ZoomEarth-3B shape, one 5000x5000 tile per GPU, full two-stage "
                                    "zoom chain per question, greedy, batch 1") if B == 1 else
                                   (f"BASELINE configs[2]: ZoomEarth-3B shape, {B} question chains about {len(dev_tiles)} tiles advanced "
                                    f"together per GPU by the continuous-batching scheduler (ragged lengths +-25 %, dynamic-"
                                    f"resolution ViT batch), one step = {B} questions"),
                       "questions_per_step_per_gpu": Q_STEP if stream else B, "chain_slots_per_gpu": chains * len(engines),
                       "lanes_per_gpu": len(engines),
                       "tile": [args.tile, args.tile], "L1": lens[0], "L2": lens[1], "N1": lens[2], "N2": lens[3],
                       "repetition_penalty": PENALTY, "hip_graph": (use_graph and SLOTS <= 64) if stream else (use_graph if B == 1 else (use_graph and B <= 64)),
                       "reuse": ("stage-1 prompt KV and view features reused in stage 2 (bit-identical); the KV rows of the "
                                 "generated tokens that stage 2 re-inserts id for id are kept as the decode steps wrote them; "
                                 "prompt prefixes shared between the questions of a tile (bit-identical) and read from ONE holder's cache "
                                 "by the decode attention"),
                       "parallelism": f"dp{world}", "sharding": "accel.shard_by_tile (tile-level LPT, no data-path collective)",
                       "weight_broadcast_s": round(bcast_s, 4)},
            "weight_broadcast_s": round(bcast_s, 4),
            "per_rank": {"questions": [int(r[0]) for r in per_rank], "seconds": [round(r[1], 3) for r in per_rank],
                         "filled_own_weights": [bool(r[2]) for r in per_rank], "dist_backend": backend if use_dist else None,
                         "imbalance": round(max(r[1] for r in per_rank) / max(1e-9, float(np.mean([r[1] for r in per_rank]))), 4)},
            "tile_upload_ms": round(1000.0 * upload_stats["s"] / n_up, 3),
            "tile_upload_note": ((f"one {args.tile}x{args.tile}x3 u8 tile, pinned host memory -> HBM (ze_tile_upload), device time per TILE; "
                                  f"{upload_stats['n']} tiles uploaded on this rank INSIDE the timed region, each by its lane three tiles ahead "
                                  "of its first question on a stream of its own (TileFeeder): `value` includes them")
                                 if stream and os.environ.get("ZE_BENCH_RESIDENT_TILES") != "1" else
                                 (f"one {args.tile}x{args.tile}x3 u8 tile, pinned host memory -> HBM (ze_tile_upload), per TILE, "
                                  f"{upload_stats['n']} tiles uploaded on this rank outside the timed region (tiles are resident when "
                                  "it starts)")),
            "phase_ms_per_question": {k: round(v / max(1, len(mine)), 3) for k, v in phases.items()},
        }
        if stream:
            st = dict(bstats)
            steps_run = max(1, st.get("steps", 1))
            line["decode_ms_per_step"] = round(phases["decode"] / steps_run, 3)
            line["mean_chains_per_step"] = round(st.get("chain_steps", 0) / steps_run, 1)
            line["scheduler"] = st
            offered = max(1, st.get("generated_rows_offered", 0))
            line["generated_rows_kept_fraction"] = round(st.get("reused_generated_rows", 0) / offered, 4)
            if reuse_sens is not None:
                line["reuse_sensitivity"] = reuse_sens
                line["value_without_generated_row_reuse"] = round(line["value"] * reuse_sens["ratio"], 3)
            if len(engines) > 1:
                e.lib.ze_tune(4, 2)   # the annexes below run ONE engine's kernels alone: the persistent form (restored to 0 = by live-engine count below)
            if args.model == "3b" and not args.fp8:
                live = int(min(SLOTS, max(1, round(line["mean_chains_per_step"]))))
                mean_ctx = live_like_contexts(live)
                line["roofline"] = batch_roofline(live, shared=(10, 347))
                line["roofline"]["contexts"] = (f"the {live} chains cut back to live-like lengths before the launch (two stage-1 chains of "
                                                f"802-994 rows per stage-2 chain of 1320-1416): mean {mean_ctx:.0f} rows")
                try:
                    line["roofline_phases"] = stream_rooflines(st, line["roofline"], live, max(1, len(mine)), my_dt)
                except Exception as ex:  # the line must not die on its annex
                    line["roofline_phases"] = {"error": f"{type(ex).__name__}: {ex}"}
                try:
                    line["roofline"]["top_kernel_by_gpu_time"] = top_kernel_by_gpu_time()
                except Exception as ex:
                    line["roofline"]["top_kernel_by_gpu_time"] = {"error": f"{type(ex).__name__}: {ex}"}
                try:   # what a ROW pays for the four projections of a layer, decode step against prefill pass
                    sk = line["roofline"]["step_kernels"]
                    dec = sum(sk[k]["us"] for k in ("qkv", "o_proj", "gate_up", "down")) / live
                    rp = line["roofline_phases"]["decode"]
                    rp["projections_us_per_row_per_layer"] = round(dec, 4)
                    rp["projections_note"] = (f"qkv + o + gate/up + down (+ its reducer) of one decode step at {live} chains, alone on the GPU, per chain; "
                                              "`prefill_pass` = the same four projections of a prefill pass per row (top_kernel_by_gpu_time.by_rows, the form the stream runs)")
                    tk = line["roofline"]["top_kernel_by_gpu_time"]
                    rp["prefill_pass_projections_us_per_row_per_layer"] = {
                        rows: round(sum(ent[tk["form_in_stream"]]["layer_us"].values()) / int(rows), 4) for rows, ent in tk.get("by_rows", {}).items()}
                except Exception:
                    pass
        if args.model == "3b" and not args.fp8 and not stream:
            if B == 1:
                pm = line["phase_ms_per_question"]
                line.update(decode_rooflines(pm, lens, chain.kept_generated, line["ms_per_step"]))
                line["config"]["stage2_rows_kept_from_decode"] = chain.kept_generated
            else:
                line["roofline"] = batch_roofline(B)
                line["scheduler"] = {k: v for k, v in bstats.items() if k != "lens"}
        elif B == 1 and not stream:
            # configs[4] shapes (7B backbone and / or FP8 decoder weights): the same object for this configuration's own
            # dominant kernel -- the decode gate/up weight stream at this shape and weight width -- measured the same way
            us, by = e.profile_decode_kernel(2, iters=144)
            others = {}
            for which, name in ((0, "qkv"), (1, "o_proj"), (3, "down"), (4, "lm_head")):
                u, b = e.profile_decode_kernel(which, iters=72)
                others[name] = {"us": round(u, 2), "GBps": round(b / (u * 1e-6) / 1e9, 1)}
            ach = by / (us * 1e-6) / 1e9
            line["roofline"] = {"bound": "hbm", "kernel": ("k_gemv<SWIGLU, fp8 weights> " if args.fp8 else "k_gemv<SWIGLU> ") +
                                f"(decode gate/up weight stream of the {cfg.name} shape, {cfg.text.num_hidden_layers} launches per token)",
                                "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                                "avg_us": us, "bytes_per_launch": by, "other_decode_kernels": others}
        if args.fp8:
            line["dtype"] = "fp8-e4m3 decoder weights (per-row power-of-two scales) streamed by the decode GEMVs; bf16 activations, bf16 MFMA prefill on the dequantised copy"
            line["metric"] += " [fp8 weights: reduced precision, not the headline metric]"
        if args.fp8_act:
            line["dtype"] = ("fp8-e4m3 decoder weights + fp8-e4m3 qkv / gate-up inputs (dynamic per-row power-of-two scales): "
                             "fp8 x fp8 MFMA in the batched decode step, the same values as bf16 everywhere else")
        if want64:
            # BASELINE configs[2]: 64 chain slots (the fragment-kernel family of the decode step), a stream of 4 x 64 questions
            # about 6 tiles (every tile questioned in four passes): a chain that finishes hands its slot to the next question,
            # so the batch stays full until the stream runs dry (the drain tail is paid once per 256 questions)
            st = {}
            NQ = 256
            tiles64 = [dev[t] for t in my_tiles[:6]]
            batch_step(e, tiles64, 7_000_000, 64, slots=64)
            e.phase_timers(enable=True, reset=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            batch_step(e, tiles64, 7_100_000, NQ, st, slots=64)
            torch.cuda.synchronize()
            dt64 = time.perf_counter() - t0
            ph64 = e.phase_timers(enable=False)
            lens64 = st.pop("lens")
            dec_steps = max(1, st.get("steps", 1))
            line["batch64"] = {
                "workload": ("BASELINE configs[2]: 64 question chains about 6 tiles advanced together on one GPU by the "
                             "continuous-batching scheduler (zoomearth_amd/scheduler.py: the path of src/eval/infer.py), fed a "
                             "stream of 256 questions (four steps of 64); ragged N1 / N2 (+-25 %), chains leave at their own "
                             "budget and hand their slot to the next question, stage 2 joins the running batch; one multi-"
                             "resolution ViT call per admission round, the view of a tile encoded once per pass over its questions"),
                "value": NQ / dt64, "unit": "questions/s", "steps": NQ // 64, "questions": NQ, "chain_slots": 64, "tiles": len(tiles64),
                "ms_per_step": 1000.0 * dt64 / (NQ // 64),
                "mean_N1": float(np.mean([l[1] for l in lens64])), "mean_N2": float(np.mean([l[3] for l in lens64])),
                "mean_L1": float(np.mean([l[0] for l in lens64])), "mean_L2": float(np.mean([l[2] for l in lens64])),
                "phase_ms": {k: round(v, 2) for k, v in ph64.items()},
                "decode_ms_per_step": round(ph64["decode"] / dec_steps, 3), "scheduler": st,
                "roofline": batch_roofline(64),
            }
        if want1:
            line["configs1"] = configs1_object(4, 1)
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N=1 only
            line["cpu_baseline"] = measure_cpu_baseline(args.cpu_baseline)
        e.lib.ze_tune(4, 0)   # (back to the library's own rule: by live-engine count)
        print(json.dumps(line), flush=True)
    for en in engines:
        en.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
