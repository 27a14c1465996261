"""GPU: unit ops through the C ABI (GEMM / GEMV / RMSNorm / attention) vs float64 numpy on the same bf16 inputs.

Tolerance: inputs are exact bf16 values, accumulation is fp32, the output is rounded once to bf16, so the result
must be within one bf16 ulp (2^-8 relative) of the exact value plus fp32 accumulation noise.
"""
import numpy as np
import pytest
import torch

from gpu_util import tiny_engine, to_dev_bf16  # noqa: F401
from oracle import prng
from oracle.qwen25vl import bf16_round

pytestmark = pytest.mark.gpu


def rnd(seed, shape, std=1.0):
    n = int(np.prod(shape))
    return bf16_round(prng.normal_ih4(seed, n, std)).reshape(shape)


def close_bf16(got, want, scale=None, ulps=1.5):
    scale = np.maximum(np.abs(want), scale if scale is not None else 1e-3)
    err = np.abs(got - want) / scale
    assert err.max() <= ulps * 2.0 ** -8 + 1e-6, (err.max(), np.unravel_index(err.argmax(), err.shape))


@pytest.mark.parametrize("m,n,k,bias,act", [
    (64, 64, 64, False, 0), (130, 200, 96, True, 0), (1296, 480, 160, True, 0), (24, 512, 1176, False, 0),
    (300, 1280, 224, True, 0), (77, 640, 640, True, 1), (1, 512, 512, True, 0), (1, 2048, 1376, False, 0),
    (1, 96, 5632, True, 0), (257, 130, 40, True, 0), (1000, 2752, 512, False, 0),
])
def test_linear(tiny_engine, m, n, k, bias, act):
    a, w = rnd(1, (m, k)), rnd(2, (n, k), 0.05)
    b = rnd(3, (n,), 0.5) if bias else None
    got = tiny_engine.op_linear(to_dev_bf16(a), to_dev_bf16(w), to_dev_bf16(b) if bias else None, act).float().cpu().numpy()
    want = a.astype(np.float64) @ w.astype(np.float64).T + (b.astype(np.float64) if bias else 0.0)
    if act:
        import math
        x = bf16_round(want.astype(np.float32)).astype(np.float64)
        want = 0.5 * x * (1.0 + np.vectorize(math.erf)(x / math.sqrt(2.0)))
        # one bf16 ulp of the pre-activation (|gelu'| <= 1.13) plus the output rounding
        tol = 2.0 ** -8 * (1.2 * np.maximum(np.abs(x), 0.05) + 1.5 * np.maximum(np.abs(want), 0.05))
        assert (np.abs(got - want) <= tol).all(), np.abs(got - want).max()
    else:
        close_bf16(got, want, scale=0.05 * np.sqrt(k) * 0.05)


# The GEMM shapes of the 3B benchmark path, one per tile policy of ze_launch_gemm (ze_gemm.hip): prefill at M = 802
# (qkv on the 128 x 128 spread ring, o / down on the eight-wave 64 x 128 ring, gate/up on 128 x 256), the ViT at
# M = 1296 (patch embed K = 1176 on the register-staged kernel, qkv, padded MLP width 3456), the merger, the second-stage
# prefill (M = 518) and the 16-chain batched prefill (M = 12832: 256 x 256 tiles), each against float64.
@pytest.mark.parametrize("m,n,k,bias,swiglu", [
    (802, 2560, 2048, True, False), (802, 2048, 2048, False, False), (802, 22016, 2048, False, True),
    (802, 2048, 11008, False, False), (518, 22016, 2048, False, True), (518, 2048, 11008, False, False),
    (1296, 1280, 1176, False, False), (1296, 3840, 1280, True, False), (1296, 1280, 1280, True, False),
    (1296, 6912, 1280, True, True), (1296, 1280, 3456, True, False), (324, 5120, 5120, True, False),
    (324, 2048, 5120, True, False), (12832, 22016, 2048, False, True), (12832, 2048, 11008, False, False),
    (12832, 2560, 2048, True, False),
])
def test_linear_3b_shapes(tiny_engine, m, n, k, bias, swiglu):
    a, w = rnd(21, (m, k)), rnd(22, (n, k), 0.05)
    b = rnd(23, (n,), 0.5) if bias else None
    got_t = tiny_engine.op_linear(to_dev_bf16(a), to_dev_bf16(w), to_dev_bf16(b) if bias else None, 4 if swiglu else 0)
    # float64 reference on a row sample that touches every 64-row tile of the grid (all columns): keeps the host side
    # of the 12832-row case at a few seconds
    rows = np.unique(np.concatenate([np.arange(0, m, 64) + (np.arange(0, m, 64) // 64 * 37) % 64, [0, m - 1],
                                     prng.uniform_ints(24, 192, 0, m)]).clip(0, m - 1))
    got = got_t[torch.from_numpy(rows).cuda()].float().cpu().numpy()
    full = a[rows].astype(np.float64) @ w.astype(np.float64).T + (b.astype(np.float64) if bias else 0.0)
    sc = 0.05 * np.sqrt(k) * 0.05
    if not swiglu:
        close_bf16(got, full, scale=sc)
        return
    # packed rows: [gate 0..15 | up 0..15 | gate 16..31 | ...]; C = bf16(bf16(silu(bf16(g))) * bf16(u))
    blk = full.reshape(len(rows), n // 32, 2, 16)
    g, u = blk[:, :, 0, :].reshape(len(rows), -1), blk[:, :, 1, :].reshape(len(rows), -1)
    want = g / (1.0 + np.exp(-g)) * u
    # four roundings (g, u, silu, product), each up to 2^-8 relative; |d silu / dg| <= 1.1 carries g's into the product
    tol = 2.0 ** -8 * (3.0 * np.abs(want) + 1.2 * np.maximum(np.abs(g), sc) * np.abs(u) + sc * sc)
    bad = np.abs(got - want) > tol
    assert not bad.any(), (int(bad.sum()), float((np.abs(got - want) / tol).max()))


@pytest.mark.parametrize("m,n,k,bias", [
    (1, 2560, 2048, True), (8, 2048, 2048, False), (17, 200, 128, True), (33, 72, 352, True), (64, 2560, 2048, True),
    (64, 3584, 3584, False), (64, 22016, 2048, False), (48, 2048, 11008, False), (65, 512, 256, True),
])
def test_linear_weight_streaming(tiny_engine, m, n, k, bias):
    """The launcher of the batched decode step (rows = chains): the skinny MFMA kernel for short narrow matrices,
    the split-K ring otherwise.  Same one-rounding contract as test_linear; a row's result must not depend on which
    other rows share the launch (bit-identical alone and in the batch), and repeats are bit-identical."""
    a, w = rnd(6, (m, k)), rnd(7, (n, k), 0.05)
    b = rnd(8, (n,), 0.5) if bias else None
    da, dw, db = to_dev_bf16(a), to_dev_bf16(w), to_dev_bf16(b) if bias else None
    got_t = tiny_engine.op_linear(da, dw, db, 2)
    got = got_t.float().cpu().numpy()
    want = a.astype(np.float64) @ w.astype(np.float64).T + (b.astype(np.float64) if bias else 0.0)
    close_bf16(got, want, scale=0.05 * np.sqrt(k) * 0.05)
    assert torch.equal(got_t, tiny_engine.op_linear(da, dw, db, 2))
    for r in {0, m // 2, m - 1}:
        alone = tiny_engine.op_linear(da[r:r + 1].contiguous(), dw, db, 2)
        assert torch.equal(alone[0], got_t[r]), r


# The row-streaming family of the batched decode step (engines with more than 64 chain slots): the projections of the 3B
# layer (gate/up with the SwiGLU epilogue, down with its eight K slices, qkv with bias, a 4096-wide one) at the row counts
# of the wide regime -- 65, 128, the 217 live chains of a 256-slot stream's average step, 256, 261 / 300 / 384 (the 384-row
# instance of the weight-streaming kernel), 385 / 512 (its 512-row instance), 640 (a 512- and a 128-row block) -- plus a
# drained batch; 513 / 580 / 640: gate/up on 320 x 192 ring tiles with K-steps of 32 (round 4), 768: a 512- and a 256-row block.
# 1152 (round 6): beyond 768 rows -- an engine with more than 768 chain slots, the one-lane A/B of DESIGN 7i -- the one-pass projections
# take the prefill policy's tile for that row count (same K order: same bits), the down projection stays on its eight slices.
@pytest.mark.parametrize("m", [9, 65, 128, 217, 256, 261, 300, 384, 385, 512, 513, 580, 640, 768, 1152])
@pytest.mark.parametrize("n,k,bias,swiglu", [(22016, 2048, False, True), (2048, 11008, False, False),
                                             (2560, 2048, True, False), (4096, 2048, False, False)])
def test_linear_wide_decode(tiny_engine, m, n, k, bias, swiglu):
    """ze_launch_gemm_wide against float64 (one bf16 rounding; SwiGLU: four), a row's result independent of the rows that
    share the launch -- alone, in a 64-row batch and in the full batch: the same bits -- and repeats bit-identical."""
    a, w = rnd(31, (m, k)), rnd(32, (n, k), 0.05)
    b = rnd(33, (n,), 0.5) if bias else None
    da, dw, db = to_dev_bf16(a), to_dev_bf16(w), to_dev_bf16(b) if bias else None
    act = 7 if swiglu else 6
    got_t = tiny_engine.op_linear(da, dw, db, act)
    got = got_t.float().cpu().numpy()
    full = a.astype(np.float64) @ w.astype(np.float64).T + (b.astype(np.float64) if bias else 0.0)
    sc = 0.05 * np.sqrt(k) * 0.05
    if swiglu:
        blk = full.reshape(m, n // 32, 2, 16)
        g, u = blk[:, :, 0, :].reshape(m, -1), blk[:, :, 1, :].reshape(m, -1)
        want = g / (1.0 + np.exp(-g)) * u
        tol = 2.0 ** -8 * (3.0 * np.abs(want) + 1.2 * np.maximum(np.abs(g), sc) * np.abs(u) + sc * sc)
        bad = np.abs(got - want) > tol
        assert not bad.any(), (int(bad.sum()), float((np.abs(got - want) / tol).max()))
    else:
        close_bf16(got, full, scale=sc)
    assert torch.equal(got_t, tiny_engine.op_linear(da, dw, db, act))
    for r in {0, m // 2, m - 1}:
        alone = tiny_engine.op_linear(da[r:r + 1].contiguous(), dw, db, act)
        assert torch.equal(alone[0], got_t[r]), r
    if m > 64:
        part = tiny_engine.op_linear(da[m - 64:].contiguous(), dw, db, act)
        assert torch.equal(part, got_t[m - 64:])


def test_lm_head_of_the_row_streaming_regime_at_full_vocabulary(tiny_engine):
    """the lm_head of the wide regime, N = 151,936 (HF:modeling_qwen2_5_vl.py:1386-1387 + the fp32 copy
    of HF:generation/utils.py:2894), at the row counts either side of its switches -- k_gemm_wstream up to 160 rows, ring /
    eight-phase tiles above, 376 = the stream's mean live chains, 768 = a lane's slots -- against float64 on EVERY column.
    The weight is a 4096-row random block repeated 38 times, each copy rolled by 17 rows and scaled by its own signed power of
    two (exact in bf16), so float64 needs one [M, 4096] product while a read of a wrong tile still shows.  Rows of a batch
    do not depend on the batch: the first 65 rows are the same bits at every row count."""
    n, k, blk = 151936, 2048, 4096
    base = rnd(41, (blk, k), 0.05)
    a_full = rnd(42, (768, k))
    copies = (n + blk - 1) // blk
    scale = [(-1.0) ** c * 2.0 ** ((c % 5) - 2) for c in range(copies)]
    db = to_dev_bf16(base)
    dw = torch.cat([torch.roll(db, shifts=-17 * c, dims=0) * scale[c] for c in range(copies)])[:n].contiguous()
    assert dw.dtype == torch.bfloat16 and dw.shape == (n, k)
    ref = a_full.astype(np.float64) @ base.astype(np.float64).T                # [768, 4096]
    sc = 0.05 * np.sqrt(k) * 0.05
    first = None
    for m in (65, 160, 161, 376, 768):
        got_t = tiny_engine.op_linear(to_dev_bf16(a_full[:m]), dw, None, 10)
        assert got_t.dtype == torch.float32 and got_t.shape == (m, n)
        got = got_t.cpu().numpy()
        assert np.array_equal(got, bf16_round(got)), "logits are the fp32 copy of bf16 values"
        for c in range(copies):
            cols = min(blk, n - c * blk)
            want = np.roll(ref[:m], -17 * c, axis=1)[:, :cols] * scale[c]
            close_bf16(got[:, c * blk: c * blk + cols], want, scale=sc * abs(scale[c]))
        if first is None:
            first = got_t[:65].clone()
        else:
            assert torch.equal(got_t[:65], first), m
        del got_t, got
    torch.cuda.empty_cache()


@pytest.mark.parametrize("m,n,k,bias", [
    (1, 2560, 2048, True), (8, 2048, 2048, False), (17, 208, 128, True), (33, 80, 352, True), (64, 2560, 2048, True),
    (64, 3584, 3584, False), (40, 22016, 2048, False), (64, 4608, 3584, True), (3, 16, 32, False),
])
def test_linear_fragment_major(tiny_engine, m, n, k, bias):
    """The fragment-major kernels of the batched decode step (k_pack_fragments + k_gemm_skinny<..., FRAG>): same
    one-rounding contract, a row's result independent of the other rows, repeats bit-identical."""
    a, w = rnd(9, (m, k)), rnd(10, (n, k), 0.05)
    b = rnd(11, (n,), 0.5) if bias else None
    da, dw, db = to_dev_bf16(a), to_dev_bf16(w), to_dev_bf16(b) if bias else None
    got_t = tiny_engine.op_linear(da, dw, db, 3)
    got = got_t.float().cpu().numpy()
    want = a.astype(np.float64) @ w.astype(np.float64).T + (b.astype(np.float64) if bias else 0.0)
    close_bf16(got, want, scale=0.05 * np.sqrt(k) * 0.05)
    assert torch.equal(got_t, tiny_engine.op_linear(da, dw, db, 3))
    for r in {0, m // 2, m - 1}:
        alone = tiny_engine.op_linear(da[r:r + 1].contiguous(), dw, db, 3)
        assert torch.equal(alone[0], got_t[r]), r


@pytest.mark.parametrize("m,n,k,bias", [
    (1, 2560, 2048, True), (8, 2048, 2048, False), (17, 208, 128, True), (33, 80, 352, True), (64, 2560, 2048, True),
    (64, 3584, 3584, False), (64, 4608, 3584, True), (3, 16, 32, False), (40, 2048, 4096, False),
])
def test_linear_one_shot(tiny_engine, m, n, k, bias):
    """The sixteen-wave one-shot kernel of the batched step's qkv / o projections (ze_gemm_oneshot.hip): one-rounding
    contract vs float64, a row's result independent of the other rows, repeats bit-identical."""
    a, w = rnd(12, (m, k)), rnd(13, (n, k), 0.05)
    b = rnd(14, (n,), 0.5) if bias else None
    da, dw, db = to_dev_bf16(a), to_dev_bf16(w), to_dev_bf16(b) if bias else None
    got_t = tiny_engine.op_linear(da, dw, db, 5)
    got = got_t.float().cpu().numpy()
    want = a.astype(np.float64) @ w.astype(np.float64).T + (b.astype(np.float64) if bias else 0.0)
    close_bf16(got, want, scale=0.05 * np.sqrt(k) * 0.05)
    assert torch.equal(got_t, tiny_engine.op_linear(da, dw, db, 5))
    for r in {0, m // 2, m - 1}:
        alone = tiny_engine.op_linear(da[r:r + 1].contiguous(), dw, db, 5)
        assert torch.equal(alone[0], got_t[r]), r


@pytest.mark.parametrize("rows,cols", [(5, 160), (1296, 1280), (3, 2048), (64, 512)])
def test_rmsnorm(tiny_engine, rows, cols):
    x, w = rnd(4, (rows, cols), 2.0), bf16_round(1.0 + rnd(5, (cols,), 0.1))
    got = tiny_engine.op_rmsnorm(to_dev_bf16(x), to_dev_bf16(w), 1e-6).float().cpu().numpy()
    xf = x.astype(np.float64)
    xn = bf16_round((xf / np.sqrt((xf * xf).mean(-1, keepdims=True) + 1e-6)).astype(np.float32))
    want = w.astype(np.float64) * xn
    # the intermediate bf16 rounding may flip one ulp when the fp32 rsqrt differs in the last bit
    close_bf16(got, want, ulps=3.0)


def ref_attention(q, k, v, cu, causal):
    t, h, d = q.shape
    g = h // k.shape[1]
    out = np.zeros((t, h, d))
    for s0, s1 in zip(cu[:-1], cu[1:]):
        for hh in range(h):
            s = q[s0:s1, hh].astype(np.float64) @ k[s0:s1, hh // g].astype(np.float64).T / np.sqrt(d)
            if causal:
                s = np.where(np.tril(np.ones_like(s)) > 0, s, -np.inf)
            p = np.exp(s - s.max(-1, keepdims=True))
            p /= p.sum(-1, keepdims=True)
            out[s0:s1, hh] = p @ v[s0:s1, hh // g].astype(np.float64)
    return out


@pytest.mark.parametrize("d,heads,kvh,cu,causal", [
    (80, 2, 2, [0, 64, 128, 160, 176], False), (80, 16, 16, [0, 1296], False), (80, 3, 3, [0, 4, 68, 100, 356], False),
    (128, 4, 2, [0, 300], True), (128, 16, 2, [0, 64, 193], True), (128, 8, 1, [0, 1, 66], True),
])
def test_attention(tiny_engine, d, heads, kvh, cu, causal):
    t = cu[-1]
    q, k, v = rnd(6, (t, heads, d)), rnd(7, (t, kvh, d)), rnd(8, (t, kvh, d))
    got = tiny_engine.op_attention(to_dev_bf16(q), to_dev_bf16(k), to_dev_bf16(v), cu, causal).float().cpu().numpy()
    want = ref_attention(q, k, v, cu, causal)
    # P is rounded to bf16 before the PV product (as HF eager does): allow 2^-7 of the value scale
    assert np.abs(got - want).max() <= 2.0 ** -6 * max(1.0, np.abs(want).max()), np.abs(got - want).max()
    assert np.sqrt(np.mean((got - want) ** 2)) <= 4e-3


@pytest.mark.parametrize("d,heads,kvh,t,causal", [(128, 4, 2, 500, True), (128, 4, 4, 500, False), (80, 4, 2, 500, True),
                                                  (80, 4, 4, 500, False), (80, 2, 2, 130, True)])
def test_attention_every_instantiation(tiny_engine, d, heads, kvh, t, causal):
    """All four (D, causal) instantiations at a length with full, partial and (causal) diagonal key tiles.  The D = 80
    causal one is not on the model's path but caught a compiler problem: hipcc's SLP vectoriser produced wrong rows for
    it (the kernel is now built with -fno-slp-vectorize)."""
    q, k, v = rnd(16, (t, heads, d)), rnd(17, (t, kvh, d)), rnd(18, (t, kvh, d))
    got = tiny_engine.op_attention(to_dev_bf16(q), to_dev_bf16(k), to_dev_bf16(v), [0, t], causal).float().cpu().numpy()
    want = ref_attention(q, k, v, [0, t], causal)
    assert np.abs(got - want).max() <= 2.0 ** -6 * max(1.0, np.abs(want).max()), np.abs(got - want).max()


@pytest.mark.parametrize("d,heads,kvh,cu,causal", [(128, 4, 2, [0, 500], True), (128, 16, 2, [0, 130, 131, 400, 1202], True),
                                                    (80, 4, 4, [0, 1296], False), (80, 2, 2, [0, 64, 128, 144, 400], False),
                                                    (128, 4, 4, [0, 700], False), (80, 2, 2, [0, 333], True)])
def test_attention_two_query_tiles_per_wave_give_the_same_bits(tiny_engine, d, heads, kvh, cu, causal):
    """The long-segment form of the flash kernel (128-query tiles: two 16-query tiles per wave, every K / V^T fragment read from
    LDS feeds two MFMAs) against the 64-query form (ze_tune knob 1 = 9): a row's arithmetic does not depend on the tile it sits
    in -- same key tiles, same online softmax -- so the outputs are the same bits (what keeps prefix reuse and cross-chain
    prefill exact), for ragged segments, segment ends inside a tile, GQA and both head sizes."""
    t = cu[-1]
    q, k, v = rnd(26, (t, heads, d)), rnd(27, (t, kvh, d)), rnd(28, (t, kvh, d))
    dq, dk, dv = to_dev_bf16(q), to_dev_bf16(k), to_dev_bf16(v)
    two = tiny_engine.op_attention(dq, dk, dv, cu, causal)
    try:
        tiny_engine.lib.ze_tune(1, 9)
        one = tiny_engine.op_attention(dq, dk, dv, cu, causal)
        tiny_engine.lib.ze_tune(1, 7)   # D = 128 causal: the register-staged form instead of the LDS-DMA staging (two query tiles)
        reg = tiny_engine.op_attention(dq, dk, dv, cu, causal)
    finally:
        tiny_engine.lib.ze_tune(1, 0)
    assert torch.equal(one, two) and torch.equal(reg, two)
    want = ref_attention(q, k, v, cu, causal)
    got = two.float().cpu().numpy()
    assert np.abs(got - want).max() <= 2.0 ** -6 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("m,n,k,bias,act", [
    (200, 256, 64, True, 0), (300, 520, 128, False, 0), (777, 1000, 640, True, 0), (600, 1024, 192, False, 4),
    (1000, 2560, 2048, True, 0), (2304, 4096, 2048, False, 4), (513, 257 * 8, 320, True, 0), (4096, 2048, 2752, False, 0),
    (5, 16, 256, True, 0),
])
def test_gemm_eight_phase_kernel(tiny_engine, m, n, k, bias, act):
    """k_gemm_p8 (256 x 256 tiles, half-tiles restaged three deep behind counted waits; the many-round GEMMs of the batched
    prefill and of the ViT), launched directly (op_linear act 8 / 9), against float64 and, bit for bit, against the policy's
    choice for the shape (the ring / register-staged kernels: same MFMA, same K order): one K-tile, two, three, an odd count,
    ragged edges in both dimensions, a single partial tile; repeats bit-identical (the race screen of a hand-placed wait)."""
    a, w = rnd(41, (m, k)), rnd(42, (n, k), 0.05)
    b = rnd(43, (n,), 0.5) if bias else None
    da, dw, db = to_dev_bf16(a), to_dev_bf16(w), to_dev_bf16(b) if bias else None
    try:
        tiny_engine.lib.ze_tune(7, 4)  # (never the eight-phase kernel)
        ref = tiny_engine.op_linear(da, dw, db, act)
    finally:
        tiny_engine.lib.ze_tune(7, 0)
    got_t = tiny_engine.op_linear(da, dw, db, 9 if act == 4 else 8)
    assert torch.equal(got_t, ref)
    for _ in range(8):
        assert torch.equal(tiny_engine.op_linear(da, dw, db, 9 if act == 4 else 8), got_t)
    try:   # one tile per workgroup instead of persistent workgroups (knob 4: what lanes sharing a GPU run): the same bits
        tiny_engine.lib.ze_tune(4, 1)
        for _ in range(3):
            assert torch.equal(tiny_engine.op_linear(da, dw, db, 9 if act == 4 else 8), got_t)
    finally:
        tiny_engine.lib.ze_tune(4, 0)
    if act == 0:
        want = a.astype(np.float64) @ w.astype(np.float64).T + (b.astype(np.float64) if bias else 0.0)
        close_bf16(got_t.float().cpu().numpy(), want, scale=0.05 * np.sqrt(k) * 0.05)


def test_prefill_split_k_is_one_sum_order_on_every_kernel(tiny_engine):
    """Round 6 (built, measured slower, off by default -- knob 20 = 3 turns it on): the long-K projection of a prefill pass
    in three K slices.  What made it admissible at all is that the split is a function of K alone: the eight-phase kernel (its new
    slab / ticket tail), the 64 x 64 ring tiles the policy picks at this size and the register-staged kernel give the SAME bits, a
    64-row slice of the rows alone too -- and other bits than the one-run sum, both within one bf16 rounding of float64.  (With the
    knob on for the whole process the GPU suite was green in round 6: batched = single prefill, stage-2 reuse = fresh prefill.)"""
    m, n, k = 700, 512, 4160          # 65 K-tiles: slices of 22 / 22 / 21
    a, w = rnd(51, (m, k)), rnd(52, (n, k), 0.05)
    da, dw = to_dev_bf16(a), to_dev_bf16(w)
    e = tiny_engine
    one_run = e.op_linear(da, dw, None, 8)
    try:
        e.lib.ze_tune(20, 3)
        p8 = e.op_linear(da, dw, None, 8)                      # k_gemm_p8, three slices
        assert torch.equal(p8, e.op_linear(da, dw, None, 8))   # (tickets reset themselves: repeats are the same bits)
        policy = e.op_linear(da, dw, None, 0)                  # ze_launch_gemm's own choice at 700 rows: 64 x 64 ring tiles, three slices
        e.lib.ze_tune(6, 1)
        staged = e.op_linear(da, dw, None, 0)                  # ... and the register-staged kernel (k_gemm_tn) on the same slices
        e.lib.ze_tune(6, 0)
        part = e.op_linear(da[300:364].contiguous(), dw, None, 0)   # 64 rows of it alone
    finally:
        e.lib.ze_tune(20, 0)
        e.lib.ze_tune(6, 0)
    assert torch.equal(p8, policy) and torch.equal(p8, staged) and torch.equal(part, p8[300:364])
    assert not torch.equal(p8, one_run)                         # three partial sums are not one running sum
    want = a.astype(np.float64) @ w.astype(np.float64).T
    for got in (p8, one_run):
        close_bf16(got.float().cpu().numpy(), want, scale=0.05 * np.sqrt(k) * 0.05)
