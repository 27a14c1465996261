"""CPU: the E4M3 / row-scale restatement (oracle/fp8.py) against torch's float8_e4m3fn and its defining properties."""
import numpy as np
import pytest

from oracle import fp8


def test_e4m3_round_and_bits_match_torch():
    torch = pytest.importorskip("torch")
    if not hasattr(torch, "float8_e4m3fn"):
        pytest.skip("torch without float8_e4m3fn")
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.normal(size=4000) * 30, rng.normal(size=4000) * 0.01, rng.uniform(-448, 448, 4000),
                        [0.0, -0.0, 448.0, -448.0, 2.0 ** -9, 2.0 ** -10, 1.5 * 2.0 ** -9, 2.5 * 2.0 ** -9, 0.0625, 240.0]]).astype(np.float32)
    x = x[np.abs(x) <= 448.0]
    t = torch.from_numpy(x).to(torch.float8_e4m3fn)
    assert np.array_equal(fp8.e4m3_round(x), t.float().numpy())
    got, want = fp8.e4m3_bits(fp8.e4m3_round(x)), t.view(torch.uint8).numpy()
    same = (got == want) | ((got & 0x7F) == 0) & ((want & 0x7F) == 0)   # +-0 may differ in sign after rounding to zero
    assert same.all()


def test_row_quantisation_properties():
    rng = np.random.default_rng(1)
    w = (rng.normal(size=(64, 96)) * rng.uniform(0.001, 3.0, size=(64, 1))).astype(np.float32)
    w[5] = 0.0
    w[6, :] = 0.0
    w[6, 3] = 448.0 * 2.0 ** -4          # amax exactly on a power-of-two boundary: k = -4, not -3
    bits, k, dq = fp8.quantize_rows(w)
    assert k[5] == 0 and np.all(dq[5] == 0) and k[6] == -4 and dq[6, 3] == w[6, 3]
    amax = np.abs(w).max(axis=1)
    nz = amax > 0
    assert np.all(amax[nz] / np.exp2(k[nz]) <= 448.0) and np.all(amax[nz] / np.exp2(k[nz] - 1) > 448.0)
    # the dequantised value is on the bf16 grid (3 mantissa bits times a power of two)
    as_bits = dq.view(np.uint32)
    assert np.all((as_bits & 0xFFFF) == 0)
    # relative error of a normal-range element is at most 2^-4
    big = np.abs(w) > np.exp2(k)[:, None] * 2.0 ** -6
    assert np.all(np.abs(dq - w)[big] <= np.abs(w)[big] * 2.0 ** -4 + 1e-12)


def test_act_fp8_oracle_mode_quantises_the_norm_outputs_and_stays_causal():
    """Oracle act_fp8 (ze_set_fp8_activations): per-row dynamic scales make the quantisation of a token independent of
    the other rows of the pass, so prefill of n + 1 tokens == prefill of n then one decode step, to fp32 rounding."""
    from oracle import prng
    from oracle import qwen25vl as Q
    oc = Q.tiny_config()
    w = Q.synthetic_weights(oc, seed=3, std=0.02, matrix_gain=4.0, bias_std=0.02, norm_jitter=0.1)
    ids = prng.uniform_ints(5, 24, 10, 1990).tolist()
    plain, quant = Q.Qwen25VLOracle(oc, w, "fp32"), Q.Qwen25VLOracle(oc, w, "fp32", act_fp8=True)
    a, b = plain.prefill(ids), quant.prefill(ids)
    assert float(np.abs(a - b).max()) > 1e-3              # the mode does something
    step = quant.decode_step(ids[0])
    whole = Q.Qwen25VLOracle(oc, w, "fp32", act_fp8=True).prefill(ids + ids[:1])
    assert float(np.abs(step - whole).max()) < 2e-3
