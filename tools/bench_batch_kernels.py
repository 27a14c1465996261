"""Per-kernel device time of the batched decode step (ze_profile_batch_kernel, HIP events) on the 3B shape with ragged
contexts, plus the step time of generate_batch, for a list of ze_tune settings.

usage: python tools/bench_batch_kernels.py [chains=64] [tune ...]      e.g.  ... 64 "" 8:1
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from zoomearth_amd.config import ModelConfig  # noqa: E402
from zoomearth_amd.engine import Engine  # noqa: E402
from zoomearth_amd.synth import uniform_ints  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
tunes = sys.argv[2:] or [""]
NAMES = ["qkv", "o_proj", "gate_up", "down", "lm_head", "attention", "rmsnorm", "rope_kv"]
cfg = ModelConfig.qwen25vl_7b() if os.environ.get("ZE_MODEL") == "7b" else ModelConfig.zoomearth_3b()
e = Engine(cfg, max_seqs=B, max_ctx=2048, max_patches=2048, max_tile_side=1024, max_prefill_rows=16 * 1024)
e.fill_synthetic(0)
if os.environ.get("ZE_FP8") in ("1", "2"):  # FP8 decoder weights: the batched step streams fp8 fragments (knob 10 = 1: bf16 copies)
    e.quantize_fp8()
if os.environ.get("ZE_FP8") == "2":  # ... and FP8 activations at the two norm sites (fp8 x fp8 MFMA)
    e.set_fp8_activations(True)
print(f"model {cfg.name}, fp8 = {os.environ.get('ZE_FP8', '0')} (1 weights, 2 weights + activations)", flush=True)
lens = [800 + int(v) for v in uniform_ints(5, B, 0, 640)]  # ragged: 800 .. 1440 tokens (mean ~1120)
for g0 in range(0, B, 8):
    gs = list(range(g0, min(B, g0 + 8)))
    ids = [uniform_ints(100 + s, lens[s], 1000, 150000).tolist() for s in gs]
    pl = [e.rope_index(i, []) for i in ids]
    for s in gs:
        e.seq_reset(s)
    e.prefill_batch(gs, ids, [None] * len(gs), [p[0] for p in pl], [p[1] for p in pl])
print(f"{B} chains, contexts {min(lens)}..{max(lens)} (mean {sum(lens) / B:.0f})", flush=True)
for tune in tunes:
    for k in range(16):
        e.lib.ze_tune(k, 0)
    for kv in tune.split(","):
        if ":" in kv:
            e.lib.ze_tune(int(kv.split(":")[0]), int(kv.split(":")[1]))
    for n in sorted({min(8, B), min(32, B), B}):
        row = []
        for w in range(8):
            us, by = e.profile_batch_kernel(w, n, 72)
            row.append(f"{NAMES[w]} {us:6.2f}us {by / us / 1e6:5.2f}TB/s")
        for s in range(n):
            e.seq_truncate(s, lens[s])
        e.generate_batch(list(range(n)), 4, ignore_eos=True)
        for s in range(n):
            e.seq_truncate(s, lens[s])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        N = 33
        e.generate_batch(list(range(n)), N, ignore_eos=True, sync_every=N)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / (N - 1)
        for s in range(n):
            e.seq_truncate(s, lens[s])
        print(f"tune[{tune}] n={n:3d} step {1000 * dt:6.3f} ms | " + " | ".join(row), flush=True)
e.close()
