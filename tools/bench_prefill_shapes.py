"""TFLOP/s of ze_launch_gemm on the GEMMs of a prefill / ViT pass at the row counts the stream's passes have (1-6 K rows),
alone on the GPU.  usage: python tools/bench_prefill_shapes.py [knob:value,...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from zoomearth_amd.config import ModelConfig  # noqa: E402
from zoomearth_amd.engine import Engine  # noqa: E402

e = Engine(ModelConfig.tiny(), max_seqs=1, max_ctx=256, max_patches=1024, max_tile_side=1024)
for kv in (sys.argv[1].split(",") if len(sys.argv) > 1 else []):
    e.lib.ze_tune(int(kv.split(":")[0]), int(kv.split(":")[1]))
LLM = [("qkv", 2560, 2048, 0), ("o", 2048, 2048, 0), ("gate_up", 22016, 2048, 4), ("down", 2048, 11008, 0)]
VIT = [("v.qkv", 3840, 1280, 0), ("v.proj", 1280, 1280, 0), ("v.gate_up", 6912, 1280, 4), ("v.down", 1280, 3456, 0)]
tot = {}
for group, rows in ((LLM, (1024, 1820, 2304, 3072, 4096, 6144)), (VIT, (1296, 2592, 5184, 12960))):
    for m in rows:
        line, t_all, f_all = [], 0.0, 0.0
        for name, n, k, act in group:
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
            fl = 2.0 * m * n * k
            t_all += dt
            f_all += fl
            line.append(f"{name} {dt * 1e6:7.1f}us {fl / dt / 1e12:6.0f}TF")
        print(f"M={m:6d} | " + " | ".join(line) + f" | layer GEMMs {t_all * 1e6:8.1f}us {f_all / t_all / 1e12:6.0f}TF", flush=True)
e.close()
