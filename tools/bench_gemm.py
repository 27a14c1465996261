"""Per-shape throughput of the prefill / ViT MFMA GEMM through ze_op_linear (TFLOP/s), MI355X."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zoomearth_amd.config import ModelConfig
from zoomearth_amd.engine import Engine

e = Engine(ModelConfig.tiny(), max_seqs=1, max_ctx=256, max_patches=1024, max_tile_side=1024)
shapes = [("llm qkv", 802, 2560, 2048), ("llm o", 802, 2048, 2048), ("llm gate_up", 802, 22016, 2048), ("llm down", 802, 2048, 11008),
          ("llm2 gate_up", 518, 22016, 2048), ("llm2 down", 518, 2048, 11008),
          ("llm2 qkv", 518, 2560, 2048), ("llm2 o", 518, 2048, 2048),
          ("vit qkv", 1296, 3840, 1280), ("vit proj", 1296, 1280, 1280), ("vit gate_up", 1296, 6912, 1280), ("vit down", 1296, 1280, 3456),
          ("merger0", 324, 5120, 5120), ("merger2", 324, 2048, 5120), ("patch", 1296, 1280, 1176), ("big", 4096, 4096, 4096)]
tot = 0.0
import itertools
variants = [int(x) for x in os.environ.get("KNOB7", "0").split(",")]
for (name, m, n, k), old in itertools.product(shapes, variants):
    e.lib.ze_tune(7, old)
    a = (torch.randn(m, k, device="cuda") * 0.5).to(torch.bfloat16)
    w = (torch.randn(n, k, device="cuda") * 0.05).to(torch.bfloat16)
    for _ in range(3):
        e.op_linear(a, w)
    torch.cuda.synchronize()
    it = 20
    t0 = time.perf_counter()
    for _ in range(it):
        e.op_linear(a, w)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / it
    print(f"knob7={old} {name:14s} M={m:5d} N={n:6d} K={k:6d}  {dt * 1e6:8.1f} us  {2 * m * n * k / dt / 1e12:7.1f} TFLOP/s", flush=True)
e.close()
