"""Throughput of batched decode (BASELINE configs[2]) on the 3B shape: tokens/s per decode step at batch B."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zoomearth_amd.config import ModelConfig
from zoomearth_amd.engine import Engine
from zoomearth_amd.synth import uniform_ints

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L, N = 802, 48
e = Engine(ModelConfig.zoomearth_3b(), max_seqs=B, max_ctx=1024, max_patches=2048, max_tile_side=1024)
e.fill_synthetic(0)
for kv in os.environ.get('ZE_TUNE', '').split(','):  # e.g. ZE_TUNE=7:7
    if ':' in kv:
        e.lib.ze_tune(int(kv.split(':')[0]), int(kv.split(':')[1]))
for b in (1, 8, 16, 32, 64):
    if b > B:
        break
    for s in range(b):
        ids = uniform_ints(100 + s, L, 1000, 150000).tolist()
        pos, delta = e.rope_index(ids, [])
        e.seq_reset(s)
        e.prefill(s, ids, None, pos, delta, want_logits=False)
    e.generate_batch(list(range(b)), 4, ignore_eos=True)  # warm-up
    for s in range(b):
        e.seq_truncate(s, L)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e.generate_batch(list(range(b)), N, ignore_eos=True, sync_every=N)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"batch {b:3d}: {1000 * dt / (N - 1):7.3f} ms/step  {b * (N - 1) / dt:9.0f} tokens/s  ({6.171 / (dt / (N - 1)) / 1000:.2f} TB/s of weights)", flush=True)
e.close()
