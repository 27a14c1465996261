"""GPU: the fused per-layer decode attention block (ze_mega.hip; ze_tune knob 3, off by default) against the four stand-alone
kernels it replaces (QKV GEMV + flash-decoding slices + merge + O-proj), through the C ABI.

The fused launch keeps the stand-alone arithmetic (same row ownership, same accumulation order), so the bar is
BIT-exact: teacher-forced logits of every step and greedy tokens, on the tiny fixture shape and on a shallow model of
the ZoomEarth-3B layer shape (hidden 2048, 16/2 heads) with a context long enough for several slices per kv head."""
import numpy as np
import pytest
import torch

from gpu_util import CHAIN_W, tiny_engine  # noqa: F401
from oracle import prng

from zoomearth_amd import _lib

# the fused kernels are experimental and not in the default library (csrc/Makefile: `make MEGA=1`)
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not _lib.lib().ze_mega_available(), reason="library built without MEGA=1")]


def text_ids(seed, n, hi=1990):
    return prng.uniform_ints(seed, n, 10, hi).tolist()


def prefill_text(e, seq, ids):
    pos, delta = e.rope_index(ids, [])
    e.seq_reset(seq)
    e.prefill(seq, ids, None, pos, delta, want_logits=True)


KNOB = {"attn": 3, "mlp": 4}
WHICH = "attn"


def fused(e, on):
    assert e.lib.ze_tune(KNOB[WHICH], 1 if on else 0) == 0


def run_forced(e, ids, forced):
    prefill_text(e, 0, ids)
    return [e.decode_step(0, t).cpu().numpy() for t in forced]


def check_engine(e, prompt_len, n_forced, n_gen, vocab_hi):
    ids = text_ids(5, prompt_len, vocab_hi)
    forced = [int(t) for t in text_ids(6, n_forced, vocab_hi)]
    try:
        fused(e, False)
        ref = run_forced(e, ids, forced)
        prefill_text(e, 0, ids)
        ref_tok = e.generate(0, n_gen, repetition_penalty=1.3, ignore_eos=True)
        fused(e, True)
        got = run_forced(e, ids, forced)
        prefill_text(e, 0, ids)
        got_tok = e.generate(0, n_gen, repetition_penalty=1.3, ignore_eos=True)
        prefill_text(e, 0, ids)
        got_tok_eager = e.generate(0, n_gen, repetition_penalty=1.3, ignore_eos=True, use_graph=False)
    finally:
        fused(e, False)  # the shipped default
    for i, (a, b) in enumerate(zip(ref, got)):
        assert np.array_equal(a, b), f"step {i}: max |diff| {np.abs(a - b).max()}"
    assert list(ref_tok) == list(got_tok) == list(got_tok_eager)


def test_fused_chain_slot_reuse_with_different_prompts(tiny_engine):
    """Regression: captured decode steps replayed on a chain slot that held a DIFFERENT chain before (the first fused
    version zeroed its barrier counters with a memset node that ran unordered inside the graph; identical prompts
    hid it because stale data and fresh data were equal)."""
    e = tiny_engine
    e.fill_synthetic(**CHAIN_W)
    prompts = {k: text_ids(40 + i, n) for i, (k, n) in enumerate((("a", 284), ("b", 295), ("c", 308)))}

    def gen(seq, name, on, graph=True, sync_every=16):
        fused(e, on)
        prefill_text(e, seq, prompts[name])
        return e.generate(seq, 12, ignore_eos=True, use_graph=graph, sync_every=sync_every)

    try:
        ref = {k: gen(0, k, False) for k in prompts}
        for seq, graph, sync_every in ((1, True, 16), (2, True, 1), (1, False, 16), (2, True, 3)):
            for name in "abcab":
                assert gen(seq, name, True, graph, sync_every) == ref[name], (seq, graph, sync_every, name)
    finally:
        fused(e, False)


def test_fused_attention_block_tiny(tiny_engine):
    tiny_engine.fill_synthetic(**CHAIN_W)
    check_engine(tiny_engine, prompt_len=150, n_forced=40, n_gen=48, vocab_hi=1990)


def test_fused_attention_block_3b_layer_shape():
    from zoomearth_amd.config import ModelConfig
    from zoomearth_amd.engine import Engine
    import dataclasses
    cfg = ModelConfig.zoomearth_3b()
    cfg = dataclasses.replace(cfg, text=dataclasses.replace(cfg.text, num_hidden_layers=4),
                              vision=dataclasses.replace(cfg.vision, depth=1))
    e = Engine(cfg, device=0, max_seqs=1, max_ctx=2048, max_patches=1024, max_tile_side=1024)
    try:
        e.fill_synthetic(**CHAIN_W)
        check_engine(e, prompt_len=700, n_forced=24, n_gen=40, vocab_hi=150000)
    finally:
        e.close()
        torch.cuda.empty_cache()


def test_fused_mlp_block_3b_layer_shape():
    """O-proj + gate/up + down in one launch (ze_tune knob 4) against the three stand-alone GEMV launches."""
    global WHICH
    from zoomearth_amd.config import ModelConfig
    from zoomearth_amd.engine import Engine
    import dataclasses
    cfg = ModelConfig.zoomearth_3b()
    cfg = dataclasses.replace(cfg, text=dataclasses.replace(cfg.text, num_hidden_layers=4),
                              vision=dataclasses.replace(cfg.vision, depth=1))
    e = Engine(cfg, device=0, max_seqs=2, max_ctx=2048, max_patches=1024, max_tile_side=1024)
    WHICH = "mlp"
    try:
        e.fill_synthetic(**CHAIN_W)
        check_engine(e, prompt_len=300, n_forced=24, n_gen=40, vocab_hi=150000)
        # chain-slot re-use with different prompts through captured steps
        prompts = {k: text_ids(60 + i, n, 150000) for i, (k, n) in enumerate((("a", 120), ("b", 131), ("c", 99)))}

        def gen(seq, name, on):
            fused(e, on)
            prefill_text(e, seq, prompts[name])
            return e.generate(seq, 12, ignore_eos=True)

        ref = {k: gen(0, k, False) for k in prompts}
        for name in "abcab":
            assert gen(1, name, True) == ref[name], name
    finally:
        WHICH = "attn"
        fused(e, False)
        e.lib.ze_tune(4, 0)
        e.close()
        torch.cuda.empty_cache()
