"""Device time of the weight-streaming GEMM (rows = chains of a batched decode step) per shape, by HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zoomearth_amd.config import ModelConfig
from zoomearth_amd.engine import Engine
e = Engine(ModelConfig.tiny(), max_seqs=1, max_ctx=256, max_patches=1024, max_tile_side=1024)
for m, old in ((8, 1), (8, 0), (64, 1), (64, 0)):
    e.lib.ze_tune(5, old)
    for name, n, k in (("qkv", 2560, 2048), ("o", 2048, 2048), ("gate_up", 22016, 2048), ("down", 2048, 11008), ("lm_head", 151936, 2048)):
        a = (torch.randn(m, k, device="cuda") * 0.5).to(torch.bfloat16)
        ws = [(torch.randn(n, k, device="cuda") * 0.05).to(torch.bfloat16) for _ in range(4 if n < 100000 else 1)]  # rotate: cold weights
        for w in ws:
            e.op_linear(a, w, act=2)
        torch.cuda.synchronize()
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        it = 40
        st.record()
        for i in range(it):
            e.op_linear(a, ws[i % len(ws)], act=2)
        en.record()
        torch.cuda.synchronize()
        us = st.elapsed_time(en) * 1000 / it
        print(f"{['skinny', 'ring  '][old]} M={m:3d} {name:8s} N={n:6d} K={k:6d}: {us:7.1f} us  {n * k * 2 / us / 1e6:6.2f} TB/s", flush=True)
e.close()
