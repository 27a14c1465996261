import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zoomearth_amd.config import ModelConfig
from zoomearth_amd.engine import Engine
e = Engine(ModelConfig.tiny(), max_seqs=1, max_ctx=256, max_patches=1024, max_tile_side=1024)
torch.manual_seed(0)
worst = 0
for m in (1, 3, 8, 16, 17, 33, 64):
    for n, k in ((2560, 2048), (2048, 2048), (22016, 2048), (2048, 11008), (200, 128), (72, 352), (4608, 3584), (3584, 18944)):
        a = (torch.randn(m, k, device="cuda") * 0.5).to(torch.bfloat16)
        w = (torch.randn(n, k, device="cuda") * 0.05).to(torch.bfloat16)
        b = (torch.randn(n, device="cuda") * 0.1).to(torch.bfloat16)
        e.lib.ze_tune(5, 0)
        got = e.op_linear(a, w, b, act=2).float()
        got2 = e.op_linear(a, w, b, act=2).float()
        want = (a.double() @ w.double().T + b.double()).float()
        err = (got - want).abs().max().item() / max(want.abs().max().item(), 1e-6)
        assert torch.equal(got, got2), (m, n, k)
        # batch-composition invariance: row 0 alone
        one = e.op_linear(a[:1].contiguous(), w, b, act=2).float()
        assert torch.equal(one[0], got[0]), ("invariance", m, n, k)
        worst = max(worst, err)
        assert err < 8e-3, (m, n, k, err)
print("skinny ok, worst rel err", worst)
e.close()
