"""GPU: BASELINE configs[4] at full size on one GPU -- the whole Qwen2.5-VL-7B backbone shape (28 layers, 28 q / 4 kv heads,
MLP 18944, untied 152,064-row lm_head) with FP8 (E4M3) decoder weights, with and without FP8 activations -- through
size-independent properties (the numpy oracle is far too slow at this size; the small-shape oracle comparisons are in
tests/test_gpu_7b_shape.py and tests/test_gpu_fp8.py):
  * quantising moves the model (the logits change by far more than bf16 noise) and leaves it ONE model: prefill (bf16 MFMA on
    the dequantised copy), single-chain decode (FP8 GEMV stream) and batched decode (FP8 fragments) agree within the distance
    between the two decode paths;
  * the FP8 fragment stream of the batched step equals, bit for bit, the same kernels on the dequantised bf16 fragments;
  * batch invariance (bitwise) in every mode; FP8 activations switched off return the W8A16 logits exactly."""
import numpy as np
import pytest

import parity_ledger

from oracle import prng

pytestmark = pytest.mark.gpu
W = dict(seed=4, std=0.02, matrix_gain=2.0, bias_std=0.02, norm_jitter=0.1)


@pytest.fixture(scope="module")
def eng7b():
    from zoomearth_amd.config import ModelConfig
    from zoomearth_amd.engine import Engine
    e = Engine(ModelConfig.qwen25vl_7b(), device=0, max_seqs=16, max_ctx=2048, max_patches=2048, max_tile_side=1024,
               max_prefill_rows=4096)
    e.fill_synthetic(**W)
    yield e
    e.close()


def text_ids(seed, n):
    return prng.uniform_ints(seed, n, 1000, 150000).tolist()


def three_paths(e, ids, nxt):
    pos, delta = e.rope_index(ids + [nxt], [])
    e.seq_reset(0)
    whole = e.prefill(0, ids + [nxt], None, pos, delta).cpu().numpy()
    e.seq_reset(1)
    e.prefill(1, ids, None, pos[:, :-1], delta, want_logits=False)
    gemv = e.decode_step(1, nxt).cpu().numpy()
    e.seq_reset(2)
    e.prefill(2, ids, None, pos[:, :-1], delta, want_logits=False)
    mfma = e.decode_batch([2], [nxt]).cpu().numpy()[0]
    return whole, gemv, mfma


def check_one_model(tag, whole, gemv, mfma):
    yard = float(np.abs(gemv - mfma).max())
    for name, got in (("gemv", gemv), ("batched", mfma)):
        err = float(np.abs(got - whole).max())
        print(f"{tag}: |{name} decode - prefill| = {err:.4f}, |gemv - batched| = {yard:.4f}, logit scale {np.abs(whole).max():.2f}")
        parity_ledger.record(err, yard, "test_gpu_full_size_7b.py:52")
        assert np.isfinite(got).all() and err <= 2.0 * yard + 0.02, (tag, name, err, yard)


def test_7b_fp8_is_one_model_in_every_mode(eng7b):
    e = eng7b
    ids, nxt = text_ids(11, 640), int(text_ids(12, 1)[0])
    bf16 = three_paths(e, ids, nxt)
    check_one_model("7B bf16", *bf16)
    e.quantize_fp8()
    w8 = three_paths(e, ids, nxt)
    check_one_model("7B fp8 weights", *w8)
    moved = float(np.abs(w8[0] - bf16[0]).max())
    assert moved > 2.0 * float(np.abs(bf16[1] - bf16[2]).max()), moved       # the quantisation is really in effect
    # the FP8 fragment stream against the same kernels on the dequantised bf16 fragments: bit for bit
    try:
        e.lib.ze_tune(10, 1)
        e.seq_reset(2)
        pos, delta = e.rope_index(ids + [nxt], [])
        e.prefill(2, ids, None, pos[:, :-1], delta, want_logits=False)
        deq = e.decode_batch([2], [nxt]).cpu().numpy()[0]
    finally:
        e.lib.ze_tune(10, 0)
    assert np.array_equal(deq, w8[2])
    e.set_fp8_activations(True)
    a8 = three_paths(e, ids, nxt)
    check_one_model("7B fp8 weights + fp8 activations", *a8)
    assert float(np.abs(a8[0] - w8[0]).max()) > 0.0
    e.set_fp8_activations(False)
    again = three_paths(e, ids, nxt)
    assert all(np.array_equal(a, b) for a, b in zip(again, w8))               # off again: the W8A16 model exactly


def test_7b_fp8_batch_invariance(eng7b):
    e = eng7b
    e.quantize_fp8()
    prompts = [text_ids(30 + s, 200 + 37 * s) for s in range(16)]
    tok = [int(t) for t in text_ids(50, 16)]
    for act in (False, True):
        e.set_fp8_activations(act)

        def run(slots):
            for s in slots:
                e.seq_reset(s)
                e.prefill(s, prompts[s], None, *e.rope_index(prompts[s], []), want_logits=False)
            return e.decode_batch(list(slots), [tok[s] for s in slots]).cpu().numpy()

        crowd, alone, pair = run(list(range(16))), run([9]), run([9, 3])
        assert np.isfinite(crowd).all()
        assert np.array_equal(alone[0], crowd[9]) and np.array_equal(pair[0], crowd[9]) and np.array_equal(pair[1], crowd[3])
    e.set_fp8_activations(False)
