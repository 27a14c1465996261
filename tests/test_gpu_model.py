"""GPU: tiny-config Qwen2.5-VL forward / generation through the C ABI vs the oracle and the transformers
fixture (tests/golden/tiny_chain.npz).

Floating-point protocol, tolerances written here:
  * yardstick E_hf = max |HF-bf16 logits - HF-fp32 logits| on the same teacher-forced path (from the fixture);
  * the engine's raw fp32 logits must satisfy max |engine - HF-fp32| <= 2.0 * E_hf and rms <= 2.0 * rms_hf;
  * token ids must match the HF-fp32 greedy token wherever the fp32 top-1/top-2 margin exceeds 2 * (2.0 * E_hf);
    sub-margin steps are counted and reported, not hidden;
  * integer / control-flow behaviour (graph vs eager, prefix reuse, EOS, repetition-penalty set) is exact.
"""
import json

import numpy as np
import pytest

import parity_ledger
import torch

from conftest import npz_str
from gpu_util import CHAIN_W, tiny_engine, tiny_weights  # noqa: F401
from oracle import frontend, indices, prng
from oracle import qwen25vl as Q

pytestmark = pytest.mark.gpu
TOL_FACTOR = 2.0


@pytest.fixture(scope="module")
def chain(golden_npz, tiny_engine):
    z = golden_npz("tiny_chain.npz")
    c = json.loads(npz_str(z["chain_json"]))
    assert {k: c[k] for k in ("std", "matrix_gain", "bias_std", "norm_jitter")} == {k: CHAIN_W[k] for k in ("std", "matrix_gain", "bias_std", "norm_jitter")}
    tiny_engine.fill_synthetic(**CHAIN_W)
    tile = prng.synthetic_tile(c["tile_seed"], c["tile_h"], c["tile_w"])
    t = torch.from_numpy(tile).to("cuda")
    h, w, _ = tile.shape
    s = 512 / max(w, h)
    view = tiny_engine.crop_resize(t, (0, 0, w, h), (int(w * s), int(h * s)))
    x1, y1, x2, y2 = (int(v) for v in c["bbox"])
    cx, cy = (x1 + x2) // 2, (y1 + y2) // 2
    crop = tiny_engine.crop_resize(t, (cx - 256, cy - 256, cx + 256, cy + 256), (512, 512))
    pv_v, g_v = tiny_engine.preprocess_image(view)
    pv_c, g_c = tiny_engine.preprocess_image(crop)
    assert list(g_v) == z["grid_view"].tolist() and list(g_c) == z["grid_crop"].tolist()
    return dict(z=z, c=c, pv_v=pv_v, g_v=g_v, pv_c=pv_c, g_c=g_c)


def run_prefill(e, seq, ids, embeds, grids, past=0):
    """Whole-sequence rope index, then prefill of ids[past:]."""
    pos, delta = e.rope_index(ids, grids)
    return e.prefill(seq, ids[past:], embeds, pos[:, past:], delta)


def test_synthetic_fill_equals_loaded_weights(tiny_engine, tiny_weights, chain):
    e = tiny_engine
    ids = chain["z"]["ids1"].tolist()
    emb = e.vit_forward(chain["pv_v"], [chain["g_v"]])
    e.seq_reset(0)
    a = run_prefill(e, 0, ids, emb, [chain["g_v"]]).cpu().numpy()
    e.load_state_dict(tiny_weights.items())  # same values through ze_load_weight (fp32 -> bf16 pack path)
    emb2 = e.vit_forward(chain["pv_v"], [chain["g_v"]])
    e.seq_reset(0)
    b = run_prefill(e, 0, ids, emb2, [chain["g_v"]]).cpu().numpy()
    assert torch.equal(emb, emb2)
    assert np.array_equal(a, b)
    e.fill_synthetic(**CHAIN_W)


def test_vit_vs_oracle_and_fixture(tiny_engine, tiny_weights, chain):
    z = chain["z"]
    got = tiny_engine.vit_forward(chain["pv_v"], [chain["g_v"]]).float().cpu().numpy()
    sel = got[:: max(1, got.shape[0] // 16)][:20]
    e_hf = np.abs(z["s1_vit_bf16"] - z["s1_vit_fp32"]).max()
    e_me = np.abs(sel - z["s1_vit_fp32"]).max()
    parity_ledger.record(e_me, e_hf, "test_gpu_model.py:73", bar=TOL_FACTOR)
    assert e_me <= TOL_FACTOR * e_hf, (e_me, e_hf)
    o = Q.Qwen25VLOracle(Q.tiny_config(), tiny_weights, "bf16")
    want = o.vit_forward(chain["pv_v"].cpu().numpy(), [chain["g_v"]])
    parity_ledger.record(np.abs(got - want).max(), e_hf, "test_gpu_model.py:76", bar=TOL_FACTOR)
    assert np.abs(got - want).max() <= TOL_FACTOR * e_hf


def _stage(e, z, tag, seq, ids, embeds, grids, penalty):
    """Teacher-forced pass along the HF-fp32 greedy path; returns engine logits [steps, vocab] and argmax."""
    forced = z[f"{tag}_tokens_fp32"].tolist()
    e.seq_reset(seq)
    lg = run_prefill(e, seq, ids, embeds, grids)
    e.mark_seen(seq, ids)
    logits, picks = [], []
    for step, tok in enumerate(forced):
        logits.append(lg.cpu().numpy())
        picks.append(e.sample_greedy(seq, lg, penalty))
        e.mark_seen(seq, [tok])
        if step + 1 < len(forced):
            lg = e.decode_step(seq, tok)
    return np.stack(logits), picks


def _check_stage(z, tag, logits, picks, penalty):
    ref32, ref16 = z[f"{tag}_logits_fp32"], z[f"{tag}_logits_bf16"]
    forced = z[f"{tag}_tokens_fp32"].tolist()
    e_hf, rms_hf = np.abs(ref16 - ref32).max(), np.sqrt(np.mean((ref16 - ref32) ** 2))
    e_me, rms_me = np.abs(logits - ref32).max(), np.sqrt(np.mean((logits - ref32) ** 2))
    parity_ledger.record(e_me, e_hf, "test_gpu_model.py:100", bar=TOL_FACTOR)
    assert e_me <= TOL_FACTOR * e_hf, (tag, e_me, e_hf)
    parity_ledger.record(rms_me, rms_hf, "test_gpu_model.py:101", bar=TOL_FACTOR)
    assert rms_me <= TOL_FACTOR * rms_hf, (tag, rms_me, rms_hf)
    tol = TOL_FACTOR * e_hf
    sub, mism = 0, 0
    seen = list(z["ids1"].tolist() if tag == "s1" else z["ids2"].tolist())
    for step, tok in enumerate(forced):
        sc = Q.apply_repetition_penalty(ref32[step], seen, penalty)
        top2 = np.partition(sc, -2)[-2:]
        margin = float(top2[1] - top2[0])
        if margin > 2 * tol:
            assert picks[step] == tok, (tag, step, margin)
        else:
            sub += 1
            mism += picks[step] != tok
        seen.append(tok)
    print(f"[{tag}] max|engine-fp32|={e_me:.4f} (HF bf16: {e_hf:.4f}), rms {rms_me:.4f} ({rms_hf:.4f}); "
          f"sub-margin steps {sub}/{len(forced)}, of which token differs {mism}")
    return e_me, e_hf


def test_two_stage_chain_teacher_forced(tiny_engine, chain):
    e, z, c = tiny_engine, chain["z"], chain["c"]
    pen = c["repetition_penalty"]
    emb_v = e.vit_forward(chain["pv_v"], [chain["g_v"]])
    ids1 = z["ids1"].tolist()
    lg1, p1 = _stage(e, z, "s1", 0, ids1, emb_v, [chain["g_v"]], pen)
    _check_stage(z, "s1", lg1, p1, pen)
    # stage 2: [view, crop] as a fresh sequence (what the reference does)
    pv2 = torch.cat([chain["pv_v"], chain["pv_c"]])
    grids2 = [chain["g_v"], chain["g_c"]]
    emb2 = e.vit_forward(pv2, grids2)
    assert torch.equal(emb2[: emb_v.shape[0]], emb_v)  # stage-1 view features are reusable bit for bit
    ids2 = z["ids2"].tolist()
    lg2, p2 = _stage(e, z, "s2", 1, ids2, emb2, grids2, pen)
    _check_stage(z, "s2", lg2, p2, pen)


def test_generate_graph_eager_and_stepwise_agree(tiny_engine, chain):
    e, z, c = tiny_engine, chain["z"], chain["c"]
    pen = c["repetition_penalty"]
    ids = z["ids1"].tolist()
    emb = e.vit_forward(chain["pv_v"], [chain["g_v"]])
    outs = []
    for use_graph in (False, True, True):
        e.seq_reset(0)
        run_prefill(e, 0, ids, emb, [chain["g_v"]])
        e.mark_seen(0, ids)
        outs.append(e.generate(0, 24, repetition_penalty=pen, ignore_eos=True, use_graph=use_graph, sync_every=5))
        assert e.seq_len(0) == len(ids) + 23
    assert outs[0] == outs[1] == outs[2]
    assert len(set(outs[0])) >= 16
    # stepwise free-running with the host-side sampler op
    e.seq_reset(2)
    lg = run_prefill(e, 2, ids, emb, [chain["g_v"]])
    e.mark_seen(2, ids)
    toks = []
    for _ in range(24):
        t = e.sample_greedy(2, lg, pen)
        toks.append(t)
        e.mark_seen(2, [t])
        lg = e.decode_step(2, t)
    assert toks == outs[0]
    # free-running greedy vs the HF-fp32 path: report the first divergence (not asserted beyond step 0)
    ref = z["s1_tokens_fp32"].tolist()
    first = next((i for i, (a, b) in enumerate(zip(toks, ref)) if a != b), len(ref))
    print(f"free-running greedy matches HF-fp32 for {first}/{len(ref)} steps (HF-bf16 itself: "
          f"{next((i for i, (a, b) in enumerate(zip(z['s1_tokens_bf16_free'].tolist(), ref)) if a != b), len(ref))})")
    assert first >= 1


def test_prefix_reuse_is_bit_identical(tiny_engine, chain):
    """Stage 2 re-prefilled from scratch == stage 2 appended after the cached stage-1 prompt (SURVEY 8a note)."""
    e, z = tiny_engine, chain["z"]
    ids1, ids2 = z["ids1"].tolist(), z["ids2"].tolist()
    grids2 = [chain["g_v"], chain["g_c"]]
    emb2 = e.vit_forward(torch.cat([chain["pv_v"], chain["pv_c"]]), grids2)
    n_view = chain["g_v"][1] * chain["g_v"][2] // 4
    e.seq_reset(0)
    full = run_prefill(e, 0, ids2, emb2, grids2).cpu().numpy()
    e.seq_reset(1)
    run_prefill(e, 1, ids1, emb2[:n_view].contiguous(), [chain["g_v"]])
    e.generate(1, 5, ignore_eos=True)            # pollute the cache past the prompt, then roll back
    e.seq_truncate(1, len(ids1))
    part = run_prefill(e, 1, ids2, emb2[n_view:].contiguous(), grids2, past=len(ids1)).cpu().numpy()
    assert np.array_equal(full, part)
    assert e.seq_len(1) == len(ids2)


def test_eos_stops_and_pads(tiny_engine, chain):
    e, z = tiny_engine, chain["z"]
    ids = z["ids1"].tolist()
    emb = e.vit_forward(chain["pv_v"], [chain["g_v"]])
    e.seq_reset(0)
    run_prefill(e, 0, ids, emb, [chain["g_v"]])
    free = e.generate(0, 12, ignore_eos=True, use_graph=False)
    # make the 4th generated token an EOS by construction: rebuild an engine config? -> use the sampler contract
    from zoomearth_amd.engine import Engine
    from zoomearth_amd.config import ModelConfig
    cfg = ModelConfig.tiny()
    cfg.eos_token_ids = (free[3], cfg.pad_token_id)
    e2 = Engine(cfg, max_seqs=1, max_ctx=1024, max_patches=4096)
    try:
        e2.fill_synthetic(**CHAIN_W)
        emb2 = e2.vit_forward(chain["pv_v"], [chain["g_v"]])
        for sync_every in (1, 16):
            e2.seq_reset(0)
            run_prefill(e2, 0, ids, emb2, [chain["g_v"]])
            got = e2.generate(0, 12, ignore_eos=False, use_graph=True, sync_every=sync_every)
            assert got == free[:4]
    finally:
        e2.close()


def test_image_token_mismatch_raises(tiny_engine, chain):
    from zoomearth_amd._lib import ZoomEarthError
    e, z = tiny_engine, chain["z"]
    ids = z["ids1"].tolist()
    emb = e.vit_forward(chain["pv_v"], [chain["g_v"]])
    e.seq_reset(0)
    pos, delta = e.rope_index(ids, [chain["g_v"]])
    with pytest.raises(ZoomEarthError, match="Image features and image tokens do not match"):
        e.prefill(0, ids, emb[:-1].contiguous(), pos, delta)
    with pytest.raises(ZoomEarthError):
        e.rope_index(ids[:40], [chain["g_v"]])


# ---------------------------------------------------------------------------------------------------------------------
# The 3B HEAD STRUCTURE against transformers (tests/golden/heads_chain.npz; VERDICT r4 missing #4): 16 query / 2 key-value heads
# x 128 (GQA group 8 -- HF's repeat_kv at the real group size), 16 ViT heads x 80, hidden 2048 / 1280.  Same protocol and
# tolerances as the tiny fixture above; through the single-chain path and through the batched step.
def test_engine_matches_transformers_at_the_3b_head_structure(golden_npz):
    from zoomearth_amd.config import ModelConfig
    from zoomearth_amd.engine import Engine

    z = golden_npz("heads_chain.npz")
    c = json.loads(npz_str(z["chain_json"]))
    cfg = ModelConfig.heads()
    assert (cfg.text.num_attention_heads, cfg.text.num_key_value_heads, cfg.vision.num_heads) == (16, 2, 16)
    e = Engine(cfg, device=0, max_seqs=4, max_ctx=512, max_patches=1024, max_tile_side=1024)
    try:
        e.fill_synthetic(seed=c["weight_seed"], std=c["std"], matrix_gain=c["matrix_gain"], bias_std=c["bias_std"], norm_jitter=c["norm_jitter"])
        tile = prng.synthetic_tile(c["tile_seed"], c["tile_h"], c["tile_w"])
        pv, g = e.preprocess_image(torch.from_numpy(tile).to("cuda"))   # (a tile of at most 512 px is its own view)
        assert list(g) == z["grid"].tolist()
        emb = e.vit_forward(pv, [g])
        got = emb.float().cpu().numpy()
        sel = got[:: max(1, got.shape[0] // 16)][:20]
        e_hf = np.abs(z["vit_bf16"] - z["vit_fp32"]).max()
        parity_ledger.record(np.abs(sel - z["vit_fp32"]).max(), e_hf, "test_gpu_model.py:248", bar=TOL_FACTOR)
        assert np.abs(sel - z["vit_fp32"]).max() <= TOL_FACTOR * e_hf, (np.abs(sel - z["vit_fp32"]).max(), e_hf)
        ids, forced, pen = z["ids"].tolist(), z["tokens_fp32"].tolist(), c["repetition_penalty"]
        ref32, ref16 = z["logits_fp32"], z["logits_bf16"]
        e_hf, rms_hf = np.abs(ref16 - ref32).max(), np.sqrt(np.mean((ref16 - ref32) ** 2))
        tol = TOL_FACTOR * e_hf

        def check(logits, picks, tag):
            e_me, rms_me = np.abs(logits - ref32).max(), np.sqrt(np.mean((logits - ref32) ** 2))
            parity_ledger.record(e_me, e_hf, f"3B head structure vs transformers: {tag}", bar=TOL_FACTOR)
            assert e_me <= tol and rms_me <= TOL_FACTOR * rms_hf, (tag, e_me, e_hf, rms_me, rms_hf)
            seen, sub = list(ids), 0
            for step, tok in enumerate(forced):
                sc = Q.apply_repetition_penalty(ref32[step], seen, pen)
                top2 = np.partition(sc, -2)[-2:]
                if float(top2[1] - top2[0]) > 2 * tol:
                    assert picks[step] == tok, (tag, step)
                else:
                    sub += 1
                seen.append(tok)
            print(f"[heads/{tag}] max|engine-fp32|={e_me:.4f} (HF bf16: {e_hf:.4f}), rms {rms_me:.4f} ({rms_hf:.4f}); sub-margin steps {sub}/{len(forced)}")
            assert sub < len(forced)         # (the gate leaves some steps decided; every one of them matched above)

        # (1) the single-chain path: prefill + GEMV decode steps, teacher-forced along the HF-fp32 greedy path
        e.seq_reset(0)
        lg = run_prefill(e, 0, ids, emb, [g])
        e.mark_seen(0, ids)
        logits, picks = [], []
        for step, tok in enumerate(forced):
            logits.append(lg.cpu().numpy())
            picks.append(e.sample_greedy(0, lg, pen))
            e.mark_seen(0, [tok])
            if step + 1 < len(forced):
                lg = e.decode_step(0, tok)
        check(np.stack(logits), picks, "single")
        # (2) the batched step, two chains advancing together on the same forced tokens: the fragment family (what an engine of
        # at most 64 slots runs) and the row-streaming family (the stream's: qkv + M-RoPE + KV append as one GEMM epilogue)
        for regime, tag in ((0, "batched/fragment"), (1, "batched/row-streaming")):
            assert e.set_decode_regime(regime) == regime
            for s in (1, 2):
                e.seq_reset(s)
            first = [run_prefill(e, s, ids, emb, [g]).cpu().numpy() for s in (1, 2)]
            assert np.array_equal(first[0], first[1]) and np.array_equal(first[0], logits[0])
            blog = [first[0]]
            for step, tok in enumerate(forced[:-1]):
                o = e.decode_batch([1, 2], [tok, tok], want_logits=True).cpu().numpy()
                assert np.array_equal(o[0], o[1])   # a chain's row does not depend on its place in the batch
                blog.append(o[0])
            bl = np.stack(blog)
            bp = [int(np.argmax(Q.apply_repetition_penalty(bl[i], ids + forced[:i], pen))) for i in range(len(forced))]
            check(bl, bp, tag)
        e.set_decode_regime(-1)
    finally:
        e.close()
