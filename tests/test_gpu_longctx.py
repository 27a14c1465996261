"""GPU: long contexts and slice-boundary contexts of the decode attention (flash-decoding slices of 64..128 tokens,
up to 64 slices, merged in the launch) against the prefill path (MFMA flash attention), which shares no kernel with
it: prefilling ids[:n] and reading the last position's logits must agree with prefilling ids[:n-1] and decoding
ids[n-1], within the bf16 noise of the two accumulation orders.  Contexts straddle every structural boundary: one
slice, 64-token slice edges, 16 / 17 slices (the merge's first batch), 64 slices, two rounds per slice."""
import numpy as np
import pytest
import torch

from gpu_util import CHAIN_W
from oracle import prng

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def long_engine():
    from zoomearth_amd.config import ModelConfig
    from zoomearth_amd.engine import Engine
    e = Engine(ModelConfig.tiny(), device=0, max_seqs=2, max_ctx=6144, max_patches=1024, max_tile_side=1024)
    e.fill_synthetic(**CHAIN_W)
    yield e
    e.close()


@pytest.mark.parametrize("n", [2, 63, 64, 65, 129, 1024, 1025, 1088, 1089, 4096, 4097, 5003])
def test_decode_matches_prefill_at_context(long_engine, n):
    e = long_engine
    ids = prng.uniform_ints(1000 + n, n, 10, 1990).tolist()
    pos, delta = e.rope_index(ids, [])
    e.seq_reset(0)
    full = e.prefill(0, ids, None, pos, delta).cpu().numpy()
    e.seq_reset(1)
    e.prefill(1, ids[:-1], None, pos[:, :-1], delta, want_logits=False)
    step = e.decode_step(1, ids[-1]).cpu().numpy()
    scale = float(np.abs(full).max())
    err = float(np.abs(full - step).max())
    assert np.isfinite(step).all()
    assert err <= 0.03 * scale + 0.02, (n, err, scale)
    # and the in-launch merge is deterministic: the same step again is bit-identical
    e.seq_truncate(1, n - 1)
    again = e.decode_step(1, ids[-1]).cpu().numpy()
    assert np.array_equal(step, again)
