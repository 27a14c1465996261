"""GPU: FP8 (OCP E4M3) decode weights (BASELINE configs[4], "fp8 weights") through the C ABI.
  * the quantiser kernel vs oracle/fp8.py: bits, scales and dequantised bf16 values, exact;
  * the whole model on the dequantised weights vs the oracle on the SAME weights (teacher-forced logits within 2x the
    oracle's own bf16-vs-fp32 error): prefill runs bf16 MFMA GEMMs on the dequantised copy, decode streams the fp8
    bytes -- the two paths share no kernel but must describe one model;
  * fp8 decode vs the bf16 GEMV decode of the same (dequantised) weights: same products, another summation order."""
import numpy as np
import pytest

import parity_ledger
import torch

from gpu_util import CHAIN_W
from oracle import fp8, prng
from oracle import qwen25vl as Q

pytestmark = pytest.mark.gpu


@pytest.fixture()
def fresh_tiny():
    from zoomearth_amd.config import ModelConfig
    from zoomearth_amd.engine import Engine
    e = Engine(ModelConfig.tiny(), device=0, max_seqs=2, max_ctx=1024, max_patches=1024, max_tile_side=1024)
    e.fill_synthetic(**CHAIN_W)
    yield e
    e.close()


def test_quantiser_kernel_is_exact(fresh_tiny):
    e = fresh_tiny
    rng = np.random.default_rng(3)
    w = (rng.normal(size=(97, 256)) * rng.uniform(0.001, 2.0, size=(97, 1))).astype(np.float32)
    w[7] = 0.0
    w[8, :] = 0.0
    w[8, 5] = 448.0 * 2.0 ** -3
    wb = torch.from_numpy(w).cuda().to(torch.bfloat16).contiguous()
    w_in = wb.float().cpu().numpy()                      # the bf16 values the kernel sees
    q, sc = e.op_quantize_fp8(wb)
    bits, k, dq = fp8.quantize_rows(w_in)
    assert np.array_equal(sc.cpu().numpy(), np.exp2(k.astype(np.float64)).astype(np.float32))
    got = q.cpu().numpy()
    same = (got == bits) | (((got & 0x7F) == 0) & ((bits & 0x7F) == 0))
    assert same.all()
    assert np.array_equal(wb.float().cpu().numpy(), dq)  # in place: the dequantised values, exactly


def text_ids(seed, n):
    return prng.uniform_ints(seed, n, 10, 1990).tolist()


def test_fp8_model_vs_oracle_on_dequantised_weights(fresh_tiny):
    e = fresh_tiny
    oc = Q.tiny_config()
    w = Q.synthetic_weights(oc, **CHAIN_W)
    ids = text_ids(7, 120)
    forced = [int(t) for t in text_ids(8, 10)]
    pos, delta = e.rope_index(ids, [])

    def engine_run():
        e.seq_reset(0)
        return [e.prefill(0, ids, None, pos, delta).cpu().numpy()] + [e.decode_step(0, t).cpu().numpy() for t in forced]

    bf16_run = engine_run()
    e.quantize_fp8()
    got = engine_run()
    # oracle on the dequantised weights (decoder linears only; tiny is tied, so lm_head / embedding stay bf16)
    wq = dict(w)
    for name, v in w.items():
        if name.startswith("model.language_model.layers") and name.endswith("proj.weight"):
            wq[name] = fp8.quantize_rows(v.reshape(v.shape[0], -1))[2].reshape(v.shape)
    # q/k/v are quantised as ONE stacked matrix row-wise, which equals per-matrix row-wise quantisation
    o32, o16 = Q.Qwen25VLOracle(oc, wq, "fp32"), Q.Qwen25VLOracle(oc, wq, "bf16")
    ref32 = [o32.prefill(ids)] + [o32.decode_step(t) for t in forced]
    ref16 = [o16.prefill(ids)] + [o16.decode_step(t) for t in forced]
    yard = max(float(np.abs(a - b).max()) for a, b in zip(ref16, ref32))
    worst = max(float(np.abs(a - b).max()) for a, b in zip(got, ref32))
    moved = max(float(np.abs(a - b).max()) for a, b in zip(got, bf16_run))
    print(f"fp8 model: max|engine - fp32 oracle(dq)| = {worst:.4f}, oracle bf16-vs-fp32 = {yard:.4f}, "
          f"fp8 vs unquantised engine = {moved:.4f}")
    parity_ledger.record(worst, yard, "test_gpu_fp8.py:79")
    assert worst <= 2.0 * yard
    assert moved > 4.0 * yard  # the quantisation is really in effect (it moves the logits far more than bf16 noise)
    # decode stream (fp8 GEMV) vs prefill path (bf16 GEMM on the dequantised copy): one model
    e.seq_reset(1)
    full = e.prefill(1, ids + forced[:1], None, *e.rope_index(ids + forced[:1], [])).cpu().numpy()
    parity_ledger.record(float(np.abs(full - got[1]).max()), yard, "test_gpu_fp8.py:84")
    assert float(np.abs(full - got[1]).max()) <= 2.0 * yard
    # generation runs (graph path) and batched decode (bf16 GEMM on the dequantised copy) agree with the fp8 GEMV path
    e.seq_reset(0)
    e.prefill(0, ids, None, pos, delta, want_logits=False)
    toks = e.generate(0, 6, ignore_eos=True)
    assert len(toks) == 6
    e.seq_reset(1)
    e.prefill(1, ids, None, pos, delta, want_logits=False)
    lb = e.decode_batch([1], [forced[0]]).cpu().numpy()[0]
    parity_ledger.record(float(np.abs(lb - got[1]).max()), yard, "test_gpu_fp8.py:93")
    assert float(np.abs(lb - got[1]).max()) <= 2.0 * yard


def test_fp8_fragment_stream_of_the_batched_step_is_the_dequantised_model_bit_for_bit(fresh_tiny):
    """Batched decode of the quantised engine streams FP8 weight fragments (k_pack_fragments8) and dequantises them in
    registers (v_cvt_scalef32_pk_bf16_fp8 with the row's power-of-two scale); q * 2^k is exact in bf16, so the logits
    must equal, bit for bit, those of the same kernels fed the dequantised bf16 fragments (ze_tune knob 10 = 1)."""
    e = fresh_tiny
    e.quantize_fp8()
    ids = [text_ids(20 + s, 60 + 17 * s) for s in range(2)]
    forced = [[int(t) for t in text_ids(30 + s, 3)] for s in range(2)]

    def run(knob):
        e.lib.ze_tune(10, knob)
        out = []
        for s in range(2):
            e.seq_reset(s)
            e.prefill(s, ids[s], None, *e.rope_index(ids[s], []), want_logits=False)
        for step in range(3):
            out.append(e.decode_batch([0, 1], [forced[0][step], forced[1][step]]).cpu().numpy())
        return out

    try:
        fp8_stream, bf16_frags = run(0), run(1)
    finally:
        e.lib.ze_tune(10, 0)
    for a, b in zip(fp8_stream, bf16_frags):
        assert np.array_equal(a, b)


def test_weight_reload_after_quantisation_drops_the_fp8_stream(fresh_tiny):
    """load / fill / arena hand-out after quantize_fp8() must not leave the decode GEMVs on stale fp8 copies:
    after new weights arrive, decode (GEMV path) and prefill (GEMM path) describe the NEW model again."""
    e = fresh_tiny
    ids = text_ids(41, 90)
    tok = int(text_ids(42, 1)[0])
    pos, delta = e.rope_index(ids, [])
    e.quantize_fp8()
    e.seq_reset(0)
    e.prefill(0, ids, None, pos, delta, want_logits=False)
    e.generate(0, 4, ignore_eos=True)                     # captures a decode graph on the fp8 streams
    e.fill_synthetic(seed=9, std=0.02, matrix_gain=4.0, bias_std=0.02, norm_jitter=0.1)   # new weights, unquantised
    e.seq_reset(0)
    e.prefill(0, ids, None, pos, delta, want_logits=False)
    dec = e.decode_step(0, tok).cpu().numpy()             # GEMV path: must read the new bf16 weights
    e.seq_reset(1)
    full = e.prefill(1, ids + [tok], None, *e.rope_index(ids + [tok], [])).cpu().numpy()
    assert float(np.abs(dec - full).max()) < 0.2, float(np.abs(dec - full).max())
    e.seq_reset(0)
    e.prefill(0, ids, None, pos, delta, want_logits=False)
    toks = e.generate(0, 4, ignore_eos=True)              # the graph was re-captured
    assert len(toks) == 4
    e.quantize_fp8()                                      # quantising again works (the fp8 arena is reused)
    e.seq_reset(0)
    e.prefill(0, ids, None, pos, delta, want_logits=False)
    assert len(e.generate(0, 4, ignore_eos=True)) == 4


# ----------------------------------------------------------------------------- fp8 activations (ze_set_fp8_activations)
def _quantised_oracles(oc, w, act_fp8):
    wq = dict(w)
    for name, v in w.items():
        if name.startswith("model.language_model.layers") and name.endswith("proj.weight"):
            wq[name] = fp8.quantize_rows(v.reshape(v.shape[0], -1))[2].reshape(v.shape)
    return (Q.Qwen25VLOracle(oc, wq, "fp32", act_fp8=act_fp8), Q.Qwen25VLOracle(oc, wq, "bf16", act_fp8=act_fp8))


def test_fp8_activations_need_fp8_weights(fresh_tiny):
    e = fresh_tiny
    with pytest.raises(RuntimeError, match="ze_weights_quantize_fp8"):
        e.set_fp8_activations(True)
    e.quantize_fp8()
    e.set_fp8_activations(True)
    e.set_fp8_activations(False)


def test_fp8_activation_model_vs_oracle_and_across_kernels(fresh_tiny):
    """One model whichever kernel serves a token: prefill (bf16 MFMA on fake-quantised rows), single-chain GEMV decode
    (fake-quantised row in LDS) and batched decode (FP8 fragments on v_mfma_f32_16x16x32_fp8_fp8) against the oracle
    with act_fp8 (teacher-forced logits within 2x the oracle's own bf16-vs-fp32 error)."""
    e = fresh_tiny
    oc = Q.tiny_config()
    w = Q.synthetic_weights(oc, **CHAIN_W)
    ids = text_ids(57, 130)
    forced = [int(t) for t in text_ids(58, 8)]
    pos, delta = e.rope_index(ids, [])
    e.quantize_fp8()

    def gemv_run():
        e.seq_reset(0)
        return [e.prefill(0, ids, None, pos, delta).cpu().numpy()] + [e.decode_step(0, t).cpu().numpy() for t in forced]

    def batch_run():
        e.seq_reset(1)
        out = [e.prefill(1, ids, None, pos, delta).cpu().numpy()]
        return out + [e.decode_batch([1], [t]).cpu().numpy()[0] for t in forced]

    w8a16 = gemv_run()
    e.set_fp8_activations(True)
    got_gemv, got_batch = gemv_run(), batch_run()
    o32, o16 = _quantised_oracles(oc, w, True)
    ref32 = [o32.prefill(ids)] + [o32.decode_step(t) for t in forced]
    ref16 = [o16.prefill(ids)] + [o16.decode_step(t) for t in forced]
    yard = max(float(np.abs(a - b).max()) for a, b in zip(ref16, ref32))
    worst_gemv = max(float(np.abs(a - b).max()) for a, b in zip(got_gemv, ref32))
    worst_batch = max(float(np.abs(a - b).max()) for a, b in zip(got_batch, ref32))
    moved = max(float(np.abs(a - b).max()) for a, b in zip(got_gemv, w8a16))
    print(f"fp8 activations: |gemv - fp32 oracle| = {worst_gemv:.4f}, |batched - fp32 oracle| = {worst_batch:.4f}, "
          f"oracle bf16-vs-fp32 = {yard:.4f}, moved vs bf16 activations = {moved:.4f}")
    parity_ledger.record(worst_gemv, yard, "test_gpu_fp8.py:202")
    assert worst_gemv <= 2.0 * yard and worst_batch <= 2.0 * yard
    assert moved > 0.0                                    # the mode is really in effect
    # prefill of prompt + first forced token == decode of that token (both kernels), within the same yardstick
    e.seq_reset(1)
    full = e.prefill(1, ids + forced[:1], None, *e.rope_index(ids + forced[:1], [])).cpu().numpy()
    parity_ledger.record(float(np.abs(full - got_gemv[1]).max()), yard, "test_gpu_fp8.py:207")
    assert float(np.abs(full - got_gemv[1]).max()) <= 2.0 * yard
    parity_ledger.record(float(np.abs(full - got_batch[1]).max()), yard, "test_gpu_fp8.py:208")
    assert float(np.abs(full - got_batch[1]).max()) <= 2.0 * yard
    # switching the mode off returns the W8A16 model exactly
    e.set_fp8_activations(False)
    again = gemv_run()
    for a, b in zip(again, w8a16):
        assert np.array_equal(a, b)


def test_fp8_activation_batched_step_is_batch_invariant(fresh_tiny):
    """fp8 x fp8 batched decode keeps the batch-invariance contract: a chain's logits are the same bits alone, in a
    batch of 2 and in a batch large enough for the balanced gate/up launch (> 32 rows)."""
    from zoomearth_amd.config import ModelConfig
    from zoomearth_amd.engine import Engine
    e = Engine(ModelConfig.tiny(), device=0, max_seqs=40, max_ctx=256, max_patches=256, max_tile_side=256)
    try:
        e.fill_synthetic(**CHAIN_W)
        e.quantize_fp8()
        e.set_fp8_activations(True)
        ids = [text_ids(70 + s, 40 + 3 * s) for s in range(40)]
        tok = [int(text_ids(90 + s, 1)[0]) for s in range(40)]

        def run(slots):
            for s in slots:
                e.seq_reset(s)
                e.prefill(s, ids[s], None, *e.rope_index(ids[s], []), want_logits=False)
            out = e.decode_batch(list(slots), [tok[s] for s in slots]).cpu().numpy()
            return {s: out[i] for i, s in enumerate(slots)}

        alone, pair, crowd = run([3]), run([3, 17]), run(list(range(40)))
        assert np.array_equal(alone[3], pair[3]) and np.array_equal(alone[3], crowd[3])
        assert np.array_equal(pair[17], crowd[17])
    finally:
        e.close()


@pytest.mark.parametrize("m,n,k,swiglu", [(37, 96, 256, False), (802, 2560, 2048, False), (300, 2 * 1376, 512, True),
                                         (2100, 4096, 3584, False), (1296, 22016, 2048, True)])
def test_block_scaled_fp8_gemm_vs_fp64(fresh_tiny, m, n, k, swiglu):
    """ze_op_linear_mx (k_gemm_ring_mx: v_mfma_scale_f32_16x16x128_f8f6f4, both tile shapes, ragged M / N tails) against
    fp64 arithmetic on the SAME quantised operands: every product q_a 2^ka q_w 2^kw is exact in fp32, so only the fp32
    accumulation order and the final bf16 rounding separate the two."""
    e = fresh_tiny
    rng = np.random.default_rng(m + n)
    a = (rng.normal(size=(m, k)) * rng.uniform(0.2, 3.0, size=(m, 1))).astype(np.float32)
    w = (rng.normal(size=(n, k)) * 0.05 * rng.uniform(0.5, 2.0, size=(n, 1))).astype(np.float32)
    bias = None if swiglu else (rng.normal(size=n) * 0.1).astype(np.float32)
    ab = torch.from_numpy(a).cuda().to(torch.bfloat16).contiguous()
    wb = torch.from_numpy(w).cuda().to(torch.bfloat16).contiguous()
    a8, sa = e.op_quantize_fp8(ab)                      # ab / wb now hold the dequantised values q 2^k
    w8, sw = e.op_quantize_fp8(wb)
    bb = None if bias is None else torch.from_numpy(bias).cuda().to(torch.bfloat16).contiguous()
    got = e.op_linear_mx(a8, sa, w8, sw, bias=bb, swiglu=swiglu).float().cpu().numpy()
    ref = ab.double().cpu().numpy() @ wb.double().cpu().numpy().T
    if bb is not None:
        ref = ref + bb.double().cpu().numpy()[None, :]
    if swiglu:                                          # packed rows: blocks of 16 gate rows then 16 up rows
        r = ref.reshape(m, n // 32, 2, 16)
        g, u = r[:, :, 0, :].reshape(m, -1), r[:, :, 1, :].reshape(m, -1)
        gb = torch.from_numpy(g).to(torch.bfloat16).double().numpy()
        ub = torch.from_numpy(u).to(torch.bfloat16).double().numpy()
        sil = torch.from_numpy(gb / (1.0 + np.exp(-gb))).to(torch.bfloat16).double().numpy()
        ref = sil * ub
    scale = float(np.abs(ref).max())
    err = float(np.abs(got - ref).max())
    assert got.shape == ref.shape and np.isfinite(got).all()
    assert err <= 0.012 * scale + 1e-3, (err, scale)    # bf16 output rounding (2^-8 relative) + fp32 accumulation
