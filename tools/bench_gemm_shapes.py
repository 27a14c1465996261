import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from zoomearth_amd.config import ModelConfig
from zoomearth_amd.engine import Engine
e = Engine(ModelConfig.tiny(), max_seqs=1, max_ctx=256, max_patches=1024, max_tile_side=1024)
shapes = [("vit qkv x16", 20736, 3840, 1280), ("vit proj x16", 20736, 1280, 1280), ("vit gate_up x16", 20736, 6912, 1280),
          ("vit down x16", 20736, 1280, 3456), ("vit qkv x4", 5184, 3840, 1280), ("vit gate_up x4", 5184, 6912, 1280),
          ("4096^3", 4096, 4096, 4096), ("llm gate_up 16x802", 12832, 22016, 2048), ("llm down 16x802", 12832, 2048, 11008),
          ("llm o 16x518", 8288, 2048, 2048), ("llm down 16x518", 8288, 2048, 11008), ("llm qkv 16x518", 8288, 2560, 2048)]
for name, m, n, k in shapes:
    a = (torch.randn(m, k, device="cuda") * 0.5).to(torch.bfloat16)
    w = (torch.randn(n, k, device="cuda") * 0.05).to(torch.bfloat16)
    row = []
    ref = None
    for kn in (0, 4, 6):
        e.lib.ze_tune(7, kn)
        for _ in range(2):
            out = e.op_linear(a, w)
        if kn == 4:
            ref = out.clone()
        if kn == 6:  # the tile choices accumulate in the same K order
            assert torch.equal(out, ref), f"{name}: 128 x 256 and 256 x 256 tiles differ"
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            e.op_linear(a, w)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        row.append(f"knob7={kn}: {dt * 1e6:8.1f} us {2 * m * n * k / dt / 1e12:7.1f} TF")
    print(f"{name:16s} M={m:6d} N={n:5d} K={k:5d} | " + " | ".join(row), flush=True)
e.close()
