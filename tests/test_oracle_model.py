"""CPU: the oracle's Qwen2.5-VL restatement against the transformers tiny-config fixture.

Protocol: fp32 oracle must reproduce the reference fp32 logits (float rounding
only) and greedy tokens; the bf16 oracle's teacher-forced logits must be as close to the reference
fp32 logits as the reference's OWN bf16 run is (x1.5).
"""
import json

import numpy as np
import pytest

from conftest import npz_str, sha
from oracle import frontend, prng, qwen25vl


@pytest.fixture(scope="module")
def chain(golden_npz):
    z = golden_npz("tiny_chain.npz")
    c = json.loads(npz_str(z["chain_json"]))
    cfg = qwen25vl.tiny_config()
    w = qwen25vl.synthetic_weights(cfg, seed=c["weight_seed"], std=c["std"], matrix_gain=c["matrix_gain"],
                                   bias_std=c["bias_std"], norm_jitter=c["norm_jitter"])
    tile = prng.synthetic_tile(c["tile_seed"], c["tile_h"], c["tile_w"])
    return z, c, cfg, w, tile


def _views(c, tile):
    # reference host code restated: view = resize to <=512; crop = 512x512 box around the bbox
    h, w, _ = tile.shape
    s = 512 / max(w, h)
    view = frontend.resize_bicubic(tile, int(w * s), int(h * s))
    x1, y1, x2, y2 = (int(v) for v in c["bbox"])
    cx, cy = (x1 + x2) // 2, (y1 + y2) // 2
    crop = frontend.crop_zero_fill(tile, (cx - 256, cy - 256, cx + 256, cy + 256))
    return view, crop


def test_chain_inputs_pinned(chain):
    z, c, cfg, w, tile = chain
    view, crop = _views(c, tile)
    assert sha(view) == npz_str(z["view_sha256"])
    assert sha(crop) == npz_str(z["crop_sha256"])


def test_fp32_oracle_matches_reference(chain):
    z, c, cfg, w, tile = chain
    view, crop = _views(c, tile)
    pv_v, g_v = frontend.image_to_pixel_values(view)
    pv_c, g_c = frontend.image_to_pixel_values(crop)
    o = qwen25vl.Qwen25VLOracle(cfg, w, "fp32")
    vit = o.vit_forward(pv_v, [g_v])
    ref = z["s1_vit_fp32"]
    assert np.abs(vit[:: max(1, vit.shape[0] // 16)][:20] - ref).max() < 2e-4
    ids1 = z["ids1"].tolist()
    r = qwen25vl.greedy_generate(o, ids1, pv_v, [g_v], c["n1"], c["repetition_penalty"], eos_token_ids=())
    assert r["tokens"] == z["s1_tokens_fp32"].tolist()
    assert np.abs(r["logits"] - z["s1_logits_fp32"]).max() < 1e-4
    assert len(set(r["tokens"])) >= 20  # non-degenerate output
    ids2 = z["ids2"].tolist()
    r2 = qwen25vl.greedy_generate(o, ids2, np.concatenate([pv_v, pv_c]), [g_v, g_c], c["n2"],
                                  c["repetition_penalty"], eos_token_ids=())
    assert r2["tokens"] == z["s2_tokens_fp32"].tolist()
    assert np.abs(r2["logits"] - z["s2_logits_fp32"]).max() < 1e-4


def test_bf16_oracle_within_reference_bf16_error(chain):
    z, c, cfg, w, tile = chain
    view, crop = _views(c, tile)
    pv_v, g_v = frontend.image_to_pixel_values(view)
    o = qwen25vl.Qwen25VLOracle(cfg, w, "bf16")
    forced = z["s1_tokens_fp32"].tolist()
    r = qwen25vl.greedy_generate(o, z["ids1"].tolist(), pv_v, [g_v], c["n1"], c["repetition_penalty"],
                                 eos_token_ids=(), forced_tokens=forced)
    ref32, ref16 = z["s1_logits_fp32"], z["s1_logits_bf16"]
    hf_err = np.abs(ref16 - ref32).max()
    my_err = np.abs(r["logits"] - ref32).max()
    hf_rms = np.sqrt(np.mean((ref16 - ref32) ** 2))
    my_rms = np.sqrt(np.mean((r["logits"] - ref32) ** 2))
    assert my_err <= 1.5 * hf_err, (my_err, hf_err)
    assert my_rms <= 1.5 * hf_rms, (my_rms, hf_rms)


def test_repetition_penalty_and_argmax_ties():
    lg = np.array([1.0, -2.0, 1.0, 0.5], dtype=np.float32)
    out = qwen25vl.apply_repetition_penalty(lg, [1, 1, 3], 2.0)
    assert out.tolist() == [1.0, -4.0, 1.0, 0.25]
    assert int(np.argmax(out)) == 0  # lowest index wins ties, as torch.argmax


def test_bf16_round():
    x = np.array([1.0, 1.00390625, 1.01171875, -3.1415927, 65504.0, np.inf], dtype=np.float32)
    y = qwen25vl.bf16_round(x)
    assert y.tolist() == [1.0, 1.0, 1.015625, -3.140625, 65536.0, np.inf]
    assert np.isnan(qwen25vl.bf16_round(np.array([np.nan], dtype=np.float32)))[0]


def test_per_token_logps_match_reference(chain, golden_npz):
    """Rollout scoring (`_get_per_token_logps`, grpo_trainer.py:494-504): oracle vs the HF fixture score.npz."""
    z, c, cfg, w, tile = chain
    s = golden_npz("score.npz")
    view, crop = _views(c, tile)
    pv_v, g_v = frontend.image_to_pixel_values(view)
    pv_c, g_c = frontend.image_to_pixel_values(crop)
    ids = s["ids"].tolist()
    assert ids == z["ids2"].tolist() + z["s2_tokens_fp32"].tolist()
    pv = np.concatenate([pv_v, pv_c])
    got32 = qwen25vl.Qwen25VLOracle(cfg, w, "fp32").per_token_logps(ids, pv, [g_v, g_c])
    assert got32.shape == (len(ids) - 1,)
    assert np.abs(got32 - s["logps_fp32"]).max() < 5e-5
    got16 = qwen25vl.Qwen25VLOracle(cfg, w, "bf16").per_token_logps(ids, pv, [g_v, g_c])
    hf_err = np.abs(s["logps_bf16_logits_fp32_softmax"] - s["logps_fp32"]).max()
    assert np.abs(got16 - s["logps_fp32"]).max() <= 1.5 * hf_err
    # HF's own bf16 log-softmax output is the fp32-softmax value rounded to bf16
    assert np.abs(s["logps_bf16"] - s["logps_bf16_logits_fp32_softmax"]).max() <= 2.0 ** -7 * np.abs(s["logps_bf16"]).max()


# ---------------------------------------------------------------------------------------------------------------------
# The 3B HEAD STRUCTURE (16 query / 2 key-value heads x 128: GQA group 8; 16 ViT heads x 80) against transformers
# (tests/golden/heads_chain.npz, VERDICT r4 missing #4: the tiny fixture has 4 / 2 and 2 heads)
@pytest.fixture(scope="module")
def heads(golden_npz):
    z = golden_npz("heads_chain.npz")
    c = json.loads(npz_str(z["chain_json"]))
    cfg = qwen25vl.heads_config()
    assert (cfg.text.num_attention_heads, cfg.text.num_key_value_heads, cfg.head_dim) == (16, 2, 128)
    assert (cfg.vision.num_heads, cfg.vision.hidden_size // cfg.vision.num_heads) == (16, 80)
    w = qwen25vl.synthetic_weights(cfg, seed=c["weight_seed"], std=c["std"], matrix_gain=c["matrix_gain"], bias_std=c["bias_std"],
                                   norm_jitter=c["norm_jitter"])
    tile = prng.synthetic_tile(c["tile_seed"], c["tile_h"], c["tile_w"])
    assert max(tile.shape[:2]) <= 512          # the reference's resize_image leaves a tile of at most 512 px as it is
    view = tile
    assert sha(view) == npz_str(z["view_sha256"])
    pv, g = frontend.image_to_pixel_values(view)
    assert list(g) == z["grid"].tolist()
    return z, c, cfg, w, pv, g


def test_fp32_oracle_matches_reference_at_the_3b_head_structure(heads):
    z, c, cfg, w, pv, g = heads
    o = qwen25vl.Qwen25VLOracle(cfg, w, "fp32")
    vit = o.vit_forward(pv, [g])
    ref = z["vit_fp32"]
    assert np.abs(vit[:: max(1, vit.shape[0] // 16)][:20] - ref).max() < 2e-5 * max(1.0, float(np.abs(ref).max()))
    r = qwen25vl.greedy_generate(o, z["ids"].tolist(), pv, [g], c["n1"], c["repetition_penalty"], eos_token_ids=())
    assert r["tokens"] == z["tokens_fp32"].tolist()
    ref32 = z["logits_fp32"]
    assert np.abs(r["logits"] - ref32).max() < 1e-4, np.abs(r["logits"] - ref32).max()   # float rounding only (the tiny fixture's bar)
    assert len(set(r["tokens"])) >= 16


def test_bf16_oracle_within_reference_bf16_error_at_the_3b_head_structure(heads):
    z, c, cfg, w, pv, g = heads
    o = qwen25vl.Qwen25VLOracle(cfg, w, "bf16")
    r = qwen25vl.greedy_generate(o, z["ids"].tolist(), pv, [g], c["n1"], c["repetition_penalty"], eos_token_ids=(),
                                 forced_tokens=z["tokens_fp32"].tolist())
    ref32, ref16 = z["logits_fp32"], z["logits_bf16"]
    hf_err, my_err = np.abs(ref16 - ref32).max(), np.abs(r["logits"] - ref32).max()
    hf_rms, my_rms = np.sqrt(np.mean((ref16 - ref32) ** 2)), np.sqrt(np.mean((r["logits"] - ref32) ** 2))
    assert my_err <= 1.5 * hf_err, (my_err, hf_err)
    assert my_rms <= 1.5 * hf_rms, (my_rms, hf_rms)
    vit = o.vit_forward(pv, [g])
    sel = vit[:: max(1, vit.shape[0] // 16)][:20]
    assert np.abs(sel - z["vit_fp32"]).max() <= 1.5 * np.abs(z["vit_bf16"] - z["vit_fp32"]).max()
