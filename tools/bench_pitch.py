"""Does a power-of-two row pitch of the weights (K = 2048: 4096 B) cost the GEMMs anything?  The same GEMM at K and K + 64.
usage: python tools/bench_pitch.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from zoomearth_amd.config import ModelConfig  # noqa: E402
from zoomearth_amd.engine import Engine  # noqa: E402

e = Engine(ModelConfig.tiny(), max_seqs=1, max_ctx=256, max_patches=1024, max_tile_side=1024)
for name, m, n, k0, act in (("gate_up 5800", 5800, 22016, 2048, 4), ("qkv 5800", 5800, 2560, 2048, 0), ("4096^3", 4096, 4096, 4096, 0),
                            ("gate_up 384 (decode)", 384, 22016, 2048, 7), ("gate_up 256 (decode)", 256, 22016, 2048, 7),
                            ("qkv 384 (decode)", 384, 2560, 2048, 6), ("o 384 (decode)", 384, 2048, 2048, 6)):
    row = []
    for k in (k0, k0 + 64, k0 + 128):
        a = (torch.randn(m, k, device="cuda") * 0.5).to(torch.bfloat16)
        w = (torch.randn(n, k, device="cuda") * 0.05).to(torch.bfloat16)
        for _ in range(3):
            e.op_linear(a, w, None, act)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            e.op_linear(a, w, None, act)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        row.append(f"K={k}: {dt * 1e6:7.1f}us {2.0 * m * n * k / dt / 1e12:6.0f}TF")
    print(f"{name:22s} | " + " | ".join(row), flush=True)
e.close()
