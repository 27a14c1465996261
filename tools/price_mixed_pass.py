"""VERDICT r5 #1, design (a) priced on the shipped kernels: the four projections of a text layer through ze_launch_gemm (what a prefill
pass runs: k_gemm_p8 from ~1.5 K rows on) at a pass's row count ALONE and with a decode step's rows APPENDED, next to what the
decode step pays for the same rows on its own kernels (tools/bench_wide.py).  One line per (pass rows, decode rows):
the marginal microseconds per appended row per layer and projection.  Operands: N(0, 0.5^2) activations, N(0, 0.05^2) weights.
usage: python tools/price_mixed_pass.py [pass_rows,... [decode_rows,...]]     -> stdout + gpurun_out/mixed_pass_pricing.json"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from zoomearth_amd.config import ModelConfig  # noqa: E402
from zoomearth_amd.engine import Engine  # noqa: E402

PASS = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0,1728,2880,5280,12832").split(",")]
DEC = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "489,576,1152").split(",")]
LLM = [("qkv", 2560, 2048, 0), ("o", 2048, 2048, 0), ("gate_up", 22016, 2048, 4), ("down", 2048, 11008, 0)]
e = Engine(ModelConfig.tiny(), max_seqs=1, max_ctx=256, max_patches=1024, max_tile_side=1024)
W = {name: (torch.randn(n, k, device="cuda") * 0.05).to(torch.bfloat16) for name, n, k, _ in LLM}
cache = {}


def t_us(name, n, k, act, m):
    if m <= 0:
        return 0.0
    if (name, m) in cache:
        return cache[(name, m)]
    a = (torch.randn(m, k, device="cuda") * 0.5).to(torch.bfloat16)
    for _ in range(3):
        e.op_linear(a, W[name], None, act)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        e.op_linear(a, W[name], None, act)
    torch.cuda.synchronize()
    cache[(name, m)] = (time.perf_counter() - t0) / 20 * 1e6
    return cache[(name, m)]


out = []
for mp in PASS:
    for md in DEC:
        row = dict(pass_rows=mp, decode_rows=md)
        tot_a = tot_b = 0.0
        for name, n, k, act in LLM:
            alone, mixed = t_us(name, n, k, act, mp), t_us(name, n, k, act, mp + md)
            row[name] = dict(pass_alone_us=round(alone, 1), mixed_us=round(mixed, 1), marginal_us_per_row=round((mixed - alone) / md, 4))
            tot_a += alone
            tot_b += mixed
        row["layer_marginal_us_per_row"] = round((tot_b - tot_a) / md, 4)
        out.append(row)
        print(f"pass {mp:6d} + decode {md:5d} rows | " + " | ".join(f"{nm} {row[nm]['pass_alone_us']:7.1f} -> {row[nm]['mixed_us']:7.1f} ({row[nm]['marginal_us_per_row']:.4f}/row)"
                                                               for nm, *_ in LLM) + f" | layer +{row['layer_marginal_us_per_row']:.4f} us/row", flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/mixed_pass_pricing.json", "w"), indent=1)
e.close()
