"""GPU: rollout scoring (ze_score / ze_op_token_logprob) vs the oracle and the transformers fixture
(tests/golden/score.npz, made by tests/golden/make_fixtures.py score).

Tolerances: the log-softmax pick alone is fp32 arithmetic on given bf16 logits -> 2e-5 absolute against a float64
evaluation.  The whole path inherits the bf16 model error: yardstick E_hf = max |HF-bf16 - HF-fp32| per-token
log-probs from the fixture (HF's own bf16 logits, log-softmax in fp32); the engine must stay within 2 x E_hf of HF-fp32
(the factor the logits tests use).
"""
import json

import numpy as np
import pytest

import parity_ledger
import torch

from conftest import npz_str
from gpu_util import CHAIN_W, tiny_engine, tiny_weights  # noqa: F401
from oracle import prng
from oracle import qwen25vl as Q
from test_gpu_model import chain, run_prefill  # noqa: F401

pytestmark = pytest.mark.gpu


def test_token_logprob_kernel(tiny_engine):
    g = torch.Generator().manual_seed(3)
    for rows, vocab, pad in ((1, 8, 0), (5, 1000, 0), (3, 1003, 5), (130, 151936, 0), (2, 77, 3), (2, 77, 0)):
        ld = vocab + pad
        buf = (torch.randn(rows, ld, generator=g) * 4).to(torch.bfloat16).cuda()
        logits = buf[:, :vocab]
        tg = torch.randint(0, vocab, (rows,), generator=g, dtype=torch.int32).cuda()
        if ld % 8:
            with pytest.raises(RuntimeError):
                tiny_engine.op_token_logprob(logits, tg)
            continue
        got = tiny_engine.op_token_logprob(logits, tg).cpu().double()
        want = torch.log_softmax(logits.cpu().double(), -1).gather(1, tg.cpu().long()[:, None])[:, 0]
        assert (got - want).abs().max().item() < 2e-5, (rows, vocab)


def _score_inputs(e, chain, golden_npz):
    s = golden_npz("score.npz")
    ids = s["ids"].tolist()
    pv = torch.cat([chain["pv_v"], chain["pv_c"]])
    grids = [chain["g_v"], chain["g_c"]]
    emb = e.vit_forward(pv, grids)
    pos, delta = e.rope_index(ids, grids)
    return s, ids, emb, pos, delta, grids


def test_score_vs_reference_and_oracle(tiny_engine, tiny_weights, chain, golden_npz):
    e = tiny_engine
    s, ids, emb, pos, delta, grids = _score_inputs(e, chain, golden_npz)
    e.seq_reset(0)
    got = e.score(0, ids, emb, pos, delta).cpu().numpy()
    ref32, ref16 = s["logps_fp32"], s["logps_bf16_logits_fp32_softmax"]
    e_hf = np.abs(ref16 - ref32).max()
    e_me = np.abs(got - ref32).max()
    rms_hf, rms_me = np.sqrt(np.mean((ref16 - ref32) ** 2)), np.sqrt(np.mean((got - ref32) ** 2))
    print(f"max|engine-fp32|={e_me:.4f} (HF bf16 {e_hf:.4f}); rms {rms_me:.4f} ({rms_hf:.4f})")
    assert got.shape == (len(ids) - 1,)
    parity_ledger.record(e_me, e_hf, "test_gpu_score.py:61")
    assert e_me <= 2.0 * e_hf and rms_me <= 2.0 * rms_hf
    o = Q.Qwen25VLOracle(Q.tiny_config(), tiny_weights, "bf16")
    want = o.per_token_logps(ids, torch.cat([chain["pv_v"], chain["pv_c"]]).cpu().numpy(), grids)
    parity_ledger.record(np.abs(got - want).max(), e_hf, "test_gpu_score.py:64")
    assert np.abs(got - want).max() <= 2.0 * e_hf


def test_score_rows_agree_with_stepwise_logits(tiny_engine, chain, golden_npz):
    """Row t of the score equals the log-softmax pick on the logits a prefill of ids[:t+1] leaves, up to the bf16
    rounding of a logit (GEMM vs GEMV accumulation order): 2 bf16 ulps of the largest |logit|."""
    e = tiny_engine
    s, ids, emb, pos, delta, grids = _score_inputs(e, chain, golden_npz)
    e.seq_reset(0)
    got = e.score(0, ids, emb, pos, delta).cpu().numpy()
    n_img = int((np.asarray(ids) == e.config.image_token_id).sum())
    for t in (len(ids) - 2, len(ids) - 9, int(s["prompt_len"]) - 1):
        assert (np.asarray(ids[: t + 1]) == e.config.image_token_id).sum() == n_img  # both images inside the prefix
        e.seq_reset(1)
        lg = e.prefill(1, ids[: t + 1], emb, pos[:, : t + 1], delta).cpu().double()
        want = (lg[ids[t + 1]] - torch.logsumexp(lg, 0)).item()
        tol = 2 * float(lg.abs().max()) * 2.0 ** -8
        assert abs(got[t] - want) <= tol, (t, got[t], want, tol)


def test_chain_continues_after_score(tiny_engine, chain, golden_npz):
    e = tiny_engine
    s, ids, emb, pos, delta, grids = _score_inputs(e, chain, golden_npz)
    e.seq_reset(0)
    e.score(0, ids, emb, pos, delta)
    a = e.generate(0, 12, repetition_penalty=1.0)
    e.seq_reset(1)
    e.prefill(1, ids, emb, pos, delta, want_logits=False)
    b = e.generate(1, 12, repetition_penalty=1.0)
    assert a == b and len(a) > 0


def test_model_per_token_logps_layout_and_padding(tiny_engine, chain, golden_npz):
    from zoomearth_amd.modeling import ZoomEarthForConditionalGeneration as M

    e = tiny_engine
    s, ids, emb, pos, delta, grids = _score_inputs(e, chain, golden_npz)
    e.seq_reset(0)
    flat = e.score(0, ids, emb, pos, delta).cpu()
    m = M(e.config, e)
    pad = e.config.pad_token_id
    L = len(ids)
    row0 = [pad] * 3 + ids + [pad] * 2          # left and right padding
    row1 = ids[: L - 4] + [pad] * 9             # a shorter sequence (both images still inside)
    inp = torch.tensor([row0, row1])
    mask = torch.tensor([[0] * 3 + [1] * L + [0] * 2, [1] * (L - 4) + [0] * 9])
    pv = torch.cat([chain["pv_v"], chain["pv_c"]])
    g = torch.tensor([list(x) for x in grids] * 2)
    out = m.per_token_logps(inp, mask, torch.cat([pv, pv]), g).cpu()
    assert out.shape == (2, L + 4)
    assert torch.equal(out[0, 3: 3 + L - 1], flat)
    assert torch.equal(out[0, :3], torch.zeros(3)) and torch.equal(out[0, 3 + L - 1:], torch.zeros(2))
    assert torch.equal(out[1, : L - 5], flat[: L - 5])  # causal: a prefix scores the same
    assert torch.equal(out[1, L - 5:], torch.zeros(9))
