"""GEMMs of a batched decode step at 65..256 chains: the weight-streaming launcher the step uses today beyond 64 chains
(ze_launch_gemm_stream: 64 x 64 tiles, split-K) against the prefill tile policy (ze_launch_gemm), per layer shape.
usage: python tools/bench_midm.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from zoomearth_amd.config import ModelConfig  # noqa: E402
from zoomearth_amd.engine import Engine  # noqa: E402

e = Engine(ModelConfig.tiny(), max_seqs=1, max_ctx=256, max_patches=1024, max_tile_side=1024)
shapes = [("qkv", 2560, 2048), ("o", 2048, 2048), ("gate_up", 22016, 2048), ("down", 2048, 11008), ("lm_head", 151936, 2048)]
for m in (64, 128, 192, 256):
    row = []
    for name, n, k in shapes:
        ws = [(torch.randn(n, k, device="cuda") * 0.05).to(torch.bfloat16) for _ in range(4 if n < 100000 else 1)]  # rotate: no L2 reuse
        a = (torch.randn(m, k, device="cuda") * 0.5).to(torch.bfloat16)
        cell = []
        for act in (2, 0):
            for w in ws:
                e.op_linear(a, w, act=act)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            it = 24
            for i in range(it):
                e.op_linear(a, ws[i % len(ws)], act=act)
            torch.cuda.synchronize()
            cell.append((time.perf_counter() - t0) / it * 1e6)
        row.append(f"{name} {cell[0]:6.1f} / {cell[1]:6.1f}")
    print(f"M={m:3d}  stream / prefill-policy us:  " + " | ".join(row), flush=True)
e.close()
