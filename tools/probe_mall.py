"""How fast is the decode layer when its weights are served by the 256 MiB Infinity Cache instead of HBM?  The 3B layer
shape with a 4096-row vocabulary at 1, 2, 4 and 8 layers: 1 layer = 154 MB of weights re-read every step (resident),
2 layers = 308 MB (beyond the cache: evicted between two uses), and so on; reports the step time per layer for the
single-chain GEMV path and for the batched step at 64 chains.  usage: python tools/probe_mall.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from zoomearth_amd.config import ModelConfig, TextConfig  # noqa: E402
from zoomearth_amd.engine import Engine  # noqa: E402
from zoomearth_amd.synth import uniform_ints  # noqa: E402

for layers in (1, 2, 4, 8):
    cfg = ModelConfig(text=TextConfig(num_hidden_layers=layers, vocab_size=4096), image_token_id=4001,
                      vision_start_token_id=4002, vision_end_token_id=4003, eos_token_ids=(4005, 4004), pad_token_id=4004,
                      name=f"3b-shape-{layers}L")
    B = 64
    e = Engine(cfg, max_seqs=B, max_ctx=2048, max_patches=1024, max_tile_side=512, max_prefill_rows=16 * 1024)
    e.fill_synthetic(0)
    lens = [800 + int(v) for v in uniform_ints(5, B, 0, 640)]
    for g0 in range(0, B, 8):
        gs = list(range(g0, g0 + 8))
        ids = [uniform_ints(100 + s, lens[s], 10, 3990).tolist() for s in gs]
        pl = [e.rope_index(i, []) for i in ids]
        for s in gs:
            e.seq_reset(s)
        e.prefill_batch(gs, ids, [None] * len(gs), [p[0] for p in pl], [p[1] for p in pl])
    out = []
    for n in (1, 64):
        def restore():
            for s in range(n):
                e.seq_truncate(s, lens[s])
        N = 65
        if n == 1:
            e.generate(0, 8, ignore_eos=True)
            restore()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e.generate(0, N, ignore_eos=True, sync_every=N)
        else:
            e.generate_batch(list(range(n)), 4, ignore_eos=True)
            restore()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e.generate_batch(list(range(n)), N, ignore_eos=True, sync_every=N)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / N
        restore()
        out.append((n, dt))
    print(f"{layers} layers ({154 * layers} MB of layer weights per step): " +
          ", ".join(f"{n} chain(s) {1e6 * dt:8.1f} us/step" for n, dt in out), flush=True)
    e.close()
