"""Per-kernel device time of the batched decode step in the row-streaming regime (ze_profile_batch_kernel, HIP events, the 36
layers' real weights in rotation) on the 3B shape at several chain counts, for a list of ze_tune settings.
usage: python tools/bench_wide.py [slots=256] [tune ...]     e.g. ... 256 "" 15:99 15:3   (ZE_COUNTS=512,580,640: the chain counts)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from zoomearth_amd.config import ModelConfig  # noqa: E402
from zoomearth_amd.engine import Engine  # noqa: E402
from zoomearth_amd.synth import uniform_ints  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
tunes = sys.argv[2:] or [""]
NAMES = ["qkv", "o_proj", "gate_up", "down", "lm_head", "attention", "rmsnorm", "rope_kv"]
e = Engine(ModelConfig.zoomearth_3b(), max_seqs=B, max_ctx=int(os.environ.get("ZE_MAX_CTX", "2048")), max_patches=2048, max_tile_side=1024, max_prefill_rows=16 * 1024)
e.fill_synthetic(0)
lens = [800 + int(v) for v in uniform_ints(5, B, 0, 640)]  # ragged: 800 .. 1440 tokens (mean ~1120)
GROUP = int(os.environ.get("ZE_GROUP", "1"))      # chains per tile: the first ZE_SHARED prompt tokens are the tile's (one prefill,
SHARED = int(os.environ.get("ZE_SHARED", "347"))  # copied to the others: ze_seq_copy_prefix, as the scheduler does)
for g0 in range(0, B, 8):
    gs = list(range(g0, min(B, g0 + 8)))
    ids = [uniform_ints(100 + s, lens[s], 1000, 150000).tolist() for s in gs]
    if GROUP > 1:
        ids = [uniform_ints(7000 + s // GROUP, SHARED, 1000, 150000).tolist() + i[SHARED:] for s, i in zip(gs, ids)]
    pl = [e.rope_index(i, []) for i in ids]
    for s in gs:
        e.seq_reset(s)
    lead = [s for s in gs if GROUP == 1 or s % GROUP == 0]
    if lead:
        e.prefill_batch(lead, [ids[s - g0] for s in lead], [None] * len(lead), [pl[s - g0][0] for s in lead], [pl[s - g0][1] for s in lead])
    rest = [s for s in gs if s not in lead]
    for s in rest:
        e.seq_copy_prefix(s, s - s % GROUP, SHARED)
    if rest:
        e.prefill_batch(rest, [ids[s - g0][SHARED:] for s in rest], [None] * len(rest), [pl[s - g0][0][:, SHARED:] for s in rest],
                        [pl[s - g0][1] for s in rest])
if GROUP > 1:   # (round 6) chains that share an image prefix have their split row at its end
    for s in range(B):
        e.seq_set_split(s, SHARED)
print(f"{B} chain slots, contexts {min(lens)}..{max(lens)} (mean {sum(lens) / B:.0f}); family {e.set_decode_regime(-1)}", flush=True)
for tune in tunes:
    for k in range(24):
        e.lib.ze_tune(k, 0)
    for kv in tune.split(","):
        if ":" in kv:
            e.lib.ze_tune(int(kv.split(":")[0]), int(kv.split(":")[1]))
    counts = [int(x) for x in os.environ.get("ZE_COUNTS", "64,128,217,256,261,344,384,440").split(",")] + [B]
    for n in sorted({min(c, B) for c in counts}):
        row = []
        for w in (0, 1, 2, 3, 4, 5):
            us, by = e.profile_batch_kernel(w, n, 72)
            row.append(f"{NAMES[w]} {us:6.2f}us {by / us / 1e6:5.2f}TB/s")
        print(f"tune[{tune}] n={n:3d} | " + " | ".join(row), flush=True)
e.close()
