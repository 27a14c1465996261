"""Fixed cost per 256 x 256 output tile of k_gemm_p8 (prologue + epilogue) against its K loop: the ViT's qkv shape (M = 31104,
N = 3840) at K = 64 .. 2560 through ze_op_linear act 8 (the eight-phase kernel whatever the grid), HIP events.
usage: python tools/bench_p8_epilogue.py [M=31104] [N=3840]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from zoomearth_amd.config import ModelConfig  # noqa: E402
from zoomearth_amd.engine import Engine  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 31104
N = int(sys.argv[2]) if len(sys.argv) > 2 else 3840
e = Engine(ModelConfig.tiny(), max_seqs=1, max_ctx=256, max_patches=1024, max_tile_side=1024)
for kv in os.environ.get("ZE_TUNE", "").split(","):  # e.g. ZE_TUNE=7:10 (the plain two-byte epilogue)
    if ":" in kv:
        e.lib.ze_tune(int(kv.split(":")[0]), int(kv.split(":")[1]))
for K in (64, 128, 256, 640, 1280, 2560):
    a = (torch.randn(M, K, device="cuda") * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    b = torch.zeros(N, device="cuda", dtype=torch.bfloat16)
    for act, name in ((8, "bias"), (9, "swiglu")):
        for _ in range(3):
            e.op_linear(a, w, b, act)
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(10):
            e.op_linear(a, w, b, act)
        t1.record()
        torch.cuda.synchronize()
        us = t0.elapsed_time(t1) * 100.0
        print(f"M={M} N={N} K={K:5d} {name:6s}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s", flush=True)
e.close()
