"""Rollout scoring (ze_score) on the 3B shape: ms per sequence and rows/s, against a plain prefill of the same ids."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zoomearth_amd.config import ModelConfig
from zoomearth_amd.engine import Engine
from zoomearth_amd.synth import uniform_ints

e = Engine(ModelConfig.zoomearth_3b(), max_seqs=2, max_ctx=2048, max_patches=2048, max_tile_side=1024)
e.fill_synthetic(0)
for L in (512, 1416, 2048):
    ids = uniform_ints(100, L, 1000, 150000).tolist()
    pos, delta = e.rope_index(ids, [])
    res = {}
    for name, fn in (("prefill", lambda: e.prefill(0, ids, None, pos, delta, want_logits=False)),
                     ("score", lambda: e.score(0, ids, None, pos, delta))):
        for it in range(6):
            if it == 1:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            e.seq_reset(0)
            fn()
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / 5
    extra = res["score"] - res["prefill"]
    flop = 2.0 * (L - 1) * 2048 * 151936
    print(f"L={L}: prefill {1e3 * res['prefill']:.2f} ms, score {1e3 * res['score']:.2f} ms "
          f"(+{1e3 * extra:.2f} ms for {L - 1} rows of logits = {flop / extra / 1e12:.0f} TFLOP/s incl. the log-softmax pick)",
          flush=True)
e.close()
