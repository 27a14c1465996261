import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from zoomearth_amd.config import ModelConfig
from zoomearth_amd.engine import Engine
from zoomearth_amd import synth as prng
n = 8
e = Engine(ModelConfig.tiny(), device=0, max_seqs=n, max_ctx=1024, max_patches=256, max_tile_side=256)
e.fill_synthetic(seed=1, std=0.02, matrix_gain=4.0, bias_std=0.02, norm_jitter=0.1)
for L in (5, 63, 64, 65, 191, 192, 193, 255, 256, 257, 383, 384, 385, 511, 512, 513, 700, 767, 769, 1000):
    ids = prng.uniform_ints(70, L, 10, 1990).tolist()
    res = {}
    for knob in (0, 5, 6, 7, 8, 9):
        e.lib.ze_tune(8, knob)
        e.seq_reset(0)
        e.prefill(0, ids, None, *e.rope_index(ids, []), want_logits=False)
        a = e.decode_batch([0], [77]).cpu().numpy()
        res[knob] = a
    print(L, {k: (bool(np.isfinite(v).all()), float(np.abs(v - res[0]).max()) if np.isfinite(v).all() else None) for k, v in res.items()}, flush=True)
e.close()
