"""Time of the prefill / ViT flash-attention kernel through ze_op_attention (one segment of T tokens)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zoomearth_amd.config import ModelConfig
from zoomearth_amd.engine import Engine
e = Engine(ModelConfig.tiny(), max_seqs=1, max_ctx=256, max_patches=1024, max_tile_side=1024)
for (T, heads, kvh, d, causal) in [(64, 16, 2, 128, True), (128, 16, 2, 128, True), (256, 16, 2, 128, True), (802, 16, 2, 128, True),
                                   (1600, 16, 2, 128, True), (3200, 16, 2, 128, True), (1296, 16, 16, 80, False)]:
    q = torch.randn(T, heads, d, device="cuda").to(torch.bfloat16)
    k = torch.randn(T, kvh, d, device="cuda").to(torch.bfloat16)
    v = torch.randn(T, kvh, d, device="cuda").to(torch.bfloat16)
    for _ in range(3):
        e.op_attention(q, k, v, [0, T], causal)
    torch.cuda.synchronize()
    it = 20
    t0 = time.perf_counter()
    for _ in range(it):
        e.op_attention(q, k, v, [0, T], causal)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / it
    fl = 4.0 * T * T * d * heads * (0.5 if causal else 1.0)
    print(f"T={T:5d} heads={heads} kv={kvh} D={d} causal={causal}: {dt * 1e6:8.1f} us  {fl / dt / 1e12:6.1f} TFLOP/s", flush=True)
e.close()
