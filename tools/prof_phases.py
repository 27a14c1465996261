"""One phase of the question stream alone on the GPU, for rocprofv3 --kernel-trace --stats (per-kernel shares of the ViT call
and of the batched prefill pass at the sizes bench.py's roofline_phases replays).
usage: python tools/prof_phases.py vit|prefill [reps=6]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench as B  # noqa: E402
from zoomearth_amd.config import ModelConfig  # noqa: E402
from zoomearth_amd.engine import Engine  # noqa: E402
from zoomearth_amd.synth import synthetic_tile  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "vit"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
cfg = ModelConfig.zoomearth_3b()
e = Engine(cfg, max_seqs=16, max_ctx=2048, max_patches=1400 * 24, max_tile_side=1024, max_prefill_rows=16 * 832)
e.fill_synthetic(0)
for kv in os.environ.get("ZE_TUNE", "").split(","):  # A/B knobs, e.g. ZE_TUNE=1:9 (64-query attention tiles)
    if ":" in kv:
        e.lib.ze_tune(int(kv.split(":")[0]), int(kv.split(":")[1]))
img = e.tile_upload(torch.from_numpy(synthetic_tile(3, 512, 512)))
pv, grid = e.preprocess_image(img)
n_img = grid[1] * grid[2] // 4
k = 24
pvs = torch.cat([pv] * k).contiguous()
feats = e.vit_forward(pvs, [grid] * k)
ids1 = [B.question_ids(cfg, 77_000 + c, n_img) for c in range(16)]
extra = [[1000 + c] * 4 + [cfg.vision_start_token_id] + [cfg.image_token_id] * n_img + [cfg.vision_end_token_id] for c in range(16)]
pl = [e.rope_index(ids1[c] + extra[c], [grid, grid]) for c in range(16)]
emb = feats[:n_img]
torch.cuda.synchronize()
t0 = torch.cuda.Event(enable_timing=True)
t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(reps):
    if what == "vit":
        e.vit_forward(pvs, [grid] * k)
    else:
        for c in range(16):
            e.seq_reset(c)
        e.prefill_batch(list(range(16)), ids1, [emb] * 16, [pl[c][0][:, :len(ids1[c])] for c in range(16)], [pl[c][1] for c in range(16)])
        e.prefill_batch(list(range(16)), extra, [emb] * 16, [pl[c][0][:, len(ids1[c]):] for c in range(16)], [pl[c][1] for c in range(16)])
t1.record()
torch.cuda.synchronize()
print(f"{what}: {t0.elapsed_time(t1) / reps:.3f} ms per rep ({k} images / 16 chains x (802 + 330) rows)")
e.close()
