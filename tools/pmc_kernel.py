"""The kernels whose roofline bench.py reports, launched exactly as bench.py's live measurement launches them
(ze_profile_batch_kernel / ze_profile_decode_kernel: the layers' real weights and KV caches in rotation), for a rocprofv3
--pmc pass around this script: every dispatch of a kernel name is then ONE configuration, so the mean FETCH_SIZE /
WRITE_SIZE per dispatch is the HBM traffic of the launch whose duration bench.py divides the algorithmic bytes by.
usage: rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- python3 tools/pmc_kernel.py wide|wide_shared|wide<N>|wide<N>_shared|batch64|configs1
(wide_shared: the attention launch only, ten chains per tile reading their first 347 rows from one cache, as the stream;
 wide<N> / wide<N>_shared, round 4: N chains -- 410 = the mean step of the benchmark line, 580 = the bucket (513-640 chains) in
 which the stream runs 62 % of its chain-steps -- on contexts drawn like the live ones, two stage-1 chains of 802-994 rows per
 stage-2 chain of 1320-1416, mean ~1055, and EVERY kernel of the layer)
prints one JSON line: {kernel key: {"us": ..., "bytes_per_launch": ...}} (HIP-event time under the profiler, for reference)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from zoomearth_amd.config import ModelConfig  # noqa: E402
from zoomearth_amd.engine import Engine  # noqa: E402
from zoomearth_amd.synth import uniform_ints  # noqa: E402

import re  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "wide"
live = re.fullmatch(r"wide(\d+)(_shared)?", mode)   # wide410, wide580_shared, ...
if live:
    n = int(live.group(1))
    slots = min(768, (n + 127) // 128 * 128)
else:
    slots = {"wide": 256, "wide_shared": 256, "batch64": 64, "configs1": 1}[mode]
    n = {"wide": 220, "wide_shared": 220, "batch64": 64, "configs1": 1}[mode]
e = Engine(ModelConfig.zoomearth_3b(), max_seqs=slots, max_ctx=2048, max_patches=2048, max_tile_side=1024, max_prefill_rows=16 * 1024)
e.fill_synthetic(0)
from zoomearth_amd._lib import kernel_sources_sha16  # noqa: E402

out = {"mode": mode, "chains": n, "kernel_sources_sha16": kernel_sources_sha16()}   # (the tree this profile was taken with)
if mode == "configs1":
    ids = uniform_ints(100, 1200, 1000, 150000).tolist()
    pos, delta = e.rope_index(ids, [])
    e.seq_reset(0)
    e.prefill(0, ids, None, pos, delta, want_logits=False)
    for which, name in ((2, "gate_up"), (3, "down"), (0, "qkv"), (1, "o_proj"), (4, "lm_head")):
        us, by = e.profile_decode_kernel(which, iters=72)
        out[name] = {"us": round(us, 2), "bytes_per_launch": by}
else:
    lens = [800 + int(v) for v in uniform_ints(5, slots, 0, 640)]  # ragged: 800 .. 1440 tokens (mean ~1120), as the stream holds
    if live:  # live-like: bench.py's live_like_contexts
        lens = [(802 + (s * 37) % 192) if s % 3 != 2 else (1320 + (s * 53) % 96) for s in range(slots)]
    for g0 in range(0, slots, 8):
        gs = list(range(g0, min(slots, g0 + 8)))
        ids = [uniform_ints(100 + s, lens[s], 1000, 150000).tolist() for s in gs]
        pl = [e.rope_index(i, []) for i in ids]
        for s in gs:
            e.seq_reset(s)
        e.prefill_batch(gs, ids, [None] * len(gs), [p[0] for p in pl], [p[1] for p in pl])
    e.set_decode_regime(1 if mode.startswith("wide") else 0)
    if live:   # (round 6) the stream's chains are image prompts: their split row is the end of the view's image block, shared or not
        for s in range(slots):
            e.seq_set_split(s, 347)
    if mode.endswith("_shared"):
        for s in range(n):
            if s % 10:
                e.seq_set_prefix_hint(s, s - s % 10, 347)
    out["mean_context"] = sum(lens[:n]) / n
    for which, name in (((5, "attention"),) if mode.endswith("_shared") else ((5, "attention"), (2, "gate_up"), (3, "down"), (0, "qkv"), (1, "o_proj"), (4, "lm_head"), (6, "rmsnorm"))):
        us, by = e.profile_batch_kernel(which, n, iters=72)
        out[name] = {"us": round(us, 2), "bytes_per_launch": by}
print(json.dumps(out), flush=True)
e.close()
