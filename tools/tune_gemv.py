"""Measurement-only A/B of decode GEMV launch shapes (ze_tune knobs) on the real 3B weights, one process.
knob 2 = explicit grid size (0 = shipped policy: one resident round)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zoomearth_amd.config import ModelConfig
from zoomearth_amd.engine import Engine
e = Engine(ModelConfig.zoomearth_3b(), max_seqs=1, max_ctx=2048, max_patches=2048, max_tile_side=1024)
e.fill_synthetic(0)
names = {0: "qkv", 1: "o", 2: "gate_up", 3: "down", 4: "lm_head"}
sweeps = ((2, (0, 512, 704, 768, 832, 1024, 2048)), (3, (0, 256, 384, 512, 576, 1024)), (0, (0, 256, 320)), (1, (0, 256)),
          (4, (0, 768, 1024, 2048)))
for rnd in range(2):
    for which, caps in sweeps:
        row = []
        for cap in caps:
            e.lib.ze_tune(2, cap)
            us, b = e.profile_decode_kernel(which, 144)
            row.append(f"{cap}:{us:.2f}")
        print(names[which], " ".join(row), flush=True)
e.close()
