"""Spot check of ze_op_attention against a float32 torch reference for every (D, causal) instantiation."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zoomearth_amd.config import ModelConfig
from zoomearth_amd.engine import Engine
e = Engine(ModelConfig.tiny(), max_seqs=1, max_ctx=256, max_patches=1024, max_tile_side=1024)
torch.manual_seed(0)
for (T, heads, kvh, d, causal) in [(500, 4, 2, 128, True), (500, 4, 4, 128, False), (500, 4, 2, 80, True), (500, 4, 4, 80, False), (130, 2, 2, 80, True)]:
    q = torch.randn(T, heads, d, device="cuda").to(torch.bfloat16)
    k = torch.randn(T, kvh, d, device="cuda").to(torch.bfloat16)
    v = torch.randn(T, kvh, d, device="cuda").to(torch.bfloat16)
    got = e.op_attention(q, k, v, [0, T], causal).float()
    g = heads // kvh
    kk = k.float().repeat_interleave(g, dim=1)
    vv = v.float().repeat_interleave(g, dim=1)
    s = torch.einsum("qhd,khd->hqk", q.float(), kk) / d ** 0.5
    if causal:
        s = s.masked_fill(torch.triu(torch.ones(T, T, device="cuda", dtype=torch.bool), 1), float("-inf"))
    want = torch.einsum("hqk,khd->qhd", torch.softmax(s, -1), vv)
    err = (got - want).abs().max().item()
    print(f"T={T} heads={heads}/{kvh} D={d} causal={causal}: max err {err:.4f} {'OK' if err < 0.03 else 'WRONG'}", flush=True)
e.close()
