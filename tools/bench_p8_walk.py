"""The tile walk of k_gemm_p8 (ze_tune knob 21: 1 = down a column of tiles, the order of rounds 3-4; R << 8 | C = blocks of R x C tiles)
at the row counts of the stream's prefill passes and ViT calls, interleaved repeats (same process, same operands), device time."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from zoomearth_amd.config import ModelConfig  # noqa: E402
from zoomearth_amd.engine import Engine  # noqa: E402

e = Engine(ModelConfig.tiny(), max_seqs=1, max_ctx=256, max_patches=1024, max_tile_side=1024)


def t(a, w, act, tune):
    e.lib.ze_tune(21, tune)
    for _ in range(2):
        e.op_linear(a, w, act=act)
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(6):
        e.op_linear(a, w, act=act)
    en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) * 1000 / 6


walks = [("column", 1), ("8x4", (8 << 8) | 4), ("4x8", (4 << 8) | 8), ("2x16", (2 << 8) | 16), ("6x6", (6 << 8) | 6), ("3x11", (3 << 8) | 11)]
for m in [int(x) for x in (sys.argv[1:] or ["12832", "5280", "17312"])]:
    for name, n, k, act in (("gate_up", 22016, 2048, 4), ("down", 2048, 11008, 0), ("v.gate_up", 6912, 1280, 4)):
        a = (torch.randn(m, k, device="cuda") * 0.5).to(torch.bfloat16)
        w = (torch.randn(n, k, device="cuda") * 0.05).to(torch.bfloat16)
        res = {nm: [] for nm, _ in walks}
        for rep in range(3):
            for nm, tune in (walks if rep % 2 == 0 else walks[::-1]):
                res[nm].append(t(a, w, act, tune))
        print(f"M={m:6d} {name:10s} " + "  ".join(f"{nm} {min(v):7.1f}" for nm, v in res.items()), flush=True)
e.lib.ze_tune(21, 0)
e.close()
