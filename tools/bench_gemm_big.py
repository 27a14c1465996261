"""Many-row (batched prefill) GEMM shapes: shipped policy against tile experiments (ze_tune knob 7), TFLOP/s, and
bit-identity of the results (every kernel accumulates an output element in the same K order)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zoomearth_amd.config import ModelConfig
from zoomearth_amd.engine import Engine

e = Engine(ModelConfig.tiny(), max_seqs=1, max_ctx=256, max_patches=1024, max_tile_side=1024)
shapes = [("big", 4096, 4096, 4096), ("gate_up 16x802", 12832, 22016, 2048), ("down 16x802", 12832, 2048, 11008),
          ("qkv 16x802", 12832, 2560, 2048), ("gate_up 16x518", 8288, 22016, 2048), ("vit gate_up x8", 10368, 6912, 1280)]
knobs = [int(x) for x in os.environ.get("KNOB7", "0,4").split(",")]
for name, m, n, k in shapes:
    a = (torch.randn(m, k, device="cuda") * 0.5).to(torch.bfloat16)
    w = (torch.randn(n, k, device="cuda") * 0.05).to(torch.bfloat16)
    ref = None
    for kn in knobs:
        e.lib.ze_tune(7, kn)
        for _ in range(2):
            out = e.op_linear(a, w)
        torch.cuda.synchronize()
        it = 10
        t0 = time.perf_counter()
        for _ in range(it):
            e.op_linear(a, w)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / it
        same = "" if ref is None else ("  identical" if torch.equal(out, ref) else "  DIFFERENT")
        ref = out if ref is None else ref
        print(f"knob7={kn} {name:16s} M={m:6d} N={n:6d} K={k:6d}  {dt * 1e6:9.1f} us  {2 * m * n * k / dt / 1e12:7.1f} TFLOP/s{same}", flush=True)
e.lib.ze_tune(7, 0)
e.close()
