"""The eight-phase 256 x 256 kernel (ze_tune 7:8) against the two-stage 256 x 256 ring (7:4): same bits (race screen: many
repeats at several shapes), device time.  usage: python tools/bench_p8.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from zoomearth_amd.config import ModelConfig  # noqa: E402
from zoomearth_amd.engine import Engine  # noqa: E402

e = Engine(ModelConfig.tiny(), max_seqs=1, max_ctx=256, max_patches=1024, max_tile_side=1024)
shapes = [("4096^3", 4096, 4096, 4096, 0), ("8192^3", 8192, 8192, 8192, 0), ("gate_up 5800", 5800, 22016, 2048, 4), ("down 5800", 5800, 2048, 11008, 0),
          ("qkv 5800", 5800, 2560, 2048, 0), ("o 5800", 5800, 2048, 2048, 0), ("gate_up 2304", 2304, 22016, 2048, 4),
          ("gate_up 12832", 12832, 22016, 2048, 4), ("down 12832", 12832, 2048, 11008, 0), ("v.qkv 18144", 18144, 3840, 1280, 0),
          ("v.proj 18144", 18144, 1280, 1280, 0), ("v.gate_up 18144", 18144, 6912, 1280, 4), ("v.down 18144", 18144, 1280, 3456, 0),
          ("ragged 777x1000x640", 777, 1000, 640, 0), ("one tile K=64", 200, 256, 64, 0), ("K=128", 300, 520, 128, 0)]
for name, m, n, k, act in shapes:
    a = (torch.randn(m, k, device="cuda") * 0.5).to(torch.bfloat16)
    w = (torch.randn(n, k, device="cuda") * 0.05).to(torch.bfloat16)
    b = (torch.randn(n, device="cuda") * 0.5).to(torch.bfloat16) if act == 0 else None
    res = {}
    for kn in (4, 8):
        e.lib.ze_tune(7, kn)
        ref = e.op_linear(a, w, b, act)
        bad = 0
        for _ in range(12 if m * n * k < 1 << 37 else 4):
            bad += int(not torch.equal(e.op_linear(a, w, b, act), ref))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            e.op_linear(a, w, b, act)
        torch.cuda.synchronize()
        res[kn] = (ref, (time.perf_counter() - t0) / 10, bad)
    same = torch.equal(res[4][0], res[8][0])
    fl = 2.0 * m * n * k
    print(f"{name:22s} ring {res[4][1] * 1e6:8.1f}us {fl / res[4][1] / 1e12:6.0f}TF | p8 {res[8][1] * 1e6:8.1f}us {fl / res[8][1] / 1e12:6.0f}TF "
          f"({res[4][1] / res[8][1]:.3f}x) | same bits {same}, unstable repeats ring {res[4][2]} p8 {res[8][2]}", flush=True)
e.lib.ze_tune(7, 0)
e.close()
