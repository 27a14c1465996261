"""Does the power-of-two row stride of the K = 2048 operands (4096 B: every row's chunk of a K-step in ONE L2 channel) bound the
LDS-DMA ingest of the decode GEMMs?  The same M x N problem at K = 2048 and at neighbouring K (row strides of 16.5 / 17 / 15 x
256 B), through the row-streaming launcher (act 7: SwiGLU, act 6: plain), device time per launch normalised to K = 2048.
usage: python tools/bench_kpad.py [M ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from zoomearth_amd.config import ModelConfig  # noqa: E402
from zoomearth_amd.engine import Engine  # noqa: E402

e = Engine(ModelConfig.tiny(), max_seqs=1, max_ctx=256, max_patches=1024, max_tile_side=1024)
Ms = [int(x) for x in sys.argv[1:]] or [576]
for m in Ms:
    for name, n, act in (("gate_up", 22016, 7), ("qkv", 2560, 6), ("o", 2048, 6)):
        for k in (2048, 2112, 2176, 1920, 2304):
            a = (torch.randn(m, k, device="cuda") * 0.5).to(torch.bfloat16)
            ws = [(torch.randn(n, k, device="cuda") * 0.05).to(torch.bfloat16) for _ in range(6)]  # rotate: cold weights
            for w in ws:
                e.op_linear(a, w, act=act)
            torch.cuda.synchronize()
            st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            it = 60
            st.record()
            for i in range(it):
                e.op_linear(a, ws[i % len(ws)], act=act)
            en.record()
            torch.cuda.synchronize()
            us = st.elapsed_time(en) * 1000 / it
            print(f"M={m:4d} {name:8s} N={n:6d} K={k:5d} (row stride {k * 2 / 256:5.1f} x 256 B): {us:7.1f} us = {us * 2048 / k:7.1f} us per 2048 of K", flush=True)
e.close()
