"""The block-scaled FP8 GEMM (ze_op_linear_mx) against the bf16 ring on the prefill shapes.  usage: python tools/bench_gemm_mx.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from zoomearth_amd.config import ModelConfig  # noqa: E402
from zoomearth_amd.engine import Engine  # noqa: E402

e = Engine(ModelConfig.tiny(), max_seqs=1, max_ctx=256, max_patches=1024, max_tile_side=1024)
for name, m, n, k in (("4096^3", 4096, 4096, 4096), ("8192^3", 8192, 8192, 8192), ("llm qkv 16x802", 12832, 2560, 2048),
                      ("llm gate_up 16x802", 12832, 22016, 2048), ("7b gate_up 16x802", 12832, 37888, 3584), ("llm qkv 802", 802, 2560, 2048),
                      ("llm gate_up 802", 802, 22016, 2048)):
    a = (torch.randn(m, k, device="cuda") * 0.5).to(torch.bfloat16)
    w = (torch.randn(n, k, device="cuda") * 0.05).to(torch.bfloat16)
    a8, sa = e.op_quantize_fp8(a.clone())
    w8, sw = e.op_quantize_fp8(w.clone())

    def timeit(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 10

    t16 = timeit(lambda: e.op_linear(a, w))
    t8 = timeit(lambda: e.op_linear_mx(a8, sa, w8, sw))
    fl = 2.0 * m * n * k
    print(f"{name:20s} M={m:6d} N={n:6d} K={k:5d} | bf16 {t16 * 1e6:8.1f} us {fl / t16 / 1e12:7.1f} TF | fp8 block-scaled {t8 * 1e6:8.1f} us "
          f"{fl / t8 / 1e12:7.1f} TF | x{t16 / t8:.2f}", flush=True)
e.close()
