"""Measurement-only A/B of the down-projection GEMV variants (ze_tune knob 0) and grids (knob 2), 3B weights."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zoomearth_amd.config import ModelConfig
from zoomearth_amd.engine import Engine
e = Engine(ModelConfig.zoomearth_3b(), max_seqs=1, max_ctx=2048, max_patches=2048, max_tile_side=1024)
e.fill_synthetic(0)
for rnd in range(2):
    for variant in (0, 1, 4, 5, 6):
        row = []
        for cap in (0, 256, 512, 1024):
            e.lib.ze_tune(0, variant)
            e.lib.ze_tune(2, cap)
            us, b = e.profile_decode_kernel(3, 144)
            row.append(f"{cap}:{us:.2f}")
        print("down variant", variant, " ".join(row), flush=True)
    e.lib.ze_tune(0, 0)
    for v1 in (0, 1):
        row = []
        for cap in (0, 512, 688, 704, 768, 920, 1376):
            e.lib.ze_tune(1, v1)
            e.lib.ze_tune(2, cap)
            us, b = e.profile_decode_kernel(2, 144)
            row.append(f"{cap}:{us:.2f}")
        print("gate_up variant", v1, " ".join(row), flush=True)
    e.lib.ze_tune(1, 0)
e.close()
