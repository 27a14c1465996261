"""GPU: the Qwen2.5-VL-7B layer shape (BASELINE configs[4] backbone swap: hidden 3584, 28 q / 4 kv heads x 128,
MLP 18944, untied lm_head) through the C ABI against the oracle, at reduced depth / vocabulary so the numpy oracle
finishes in seconds.  None of the 3B-specific fast paths applies here (7 q heads per kv head, K = 3584 = 7 chunks,
K-split down projection over 37 chunks), so this pins the generic paths of the decode GEMV family, the decode
attention with an odd group size and the GEMM tile policy.

Tolerance (no HF fixture for this shape): the engine's fp32 logits against the fp32 oracle must stay within 2x the
oracle's own bf16-vs-fp32 error on the same teacher-forced path -- the protocol of test_gpu_model.py."""
import dataclasses

import numpy as np
import pytest

import parity_ledger
import torch

from oracle import prng
from oracle import qwen25vl as Q

pytestmark = pytest.mark.gpu
W7 = dict(seed=3, std=0.02, matrix_gain=2.0, bias_std=0.02, norm_jitter=0.1)


def test_7b_layer_shape_text_path_vs_oracle():
    from zoomearth_amd.config import ModelConfig
    from zoomearth_amd.engine import Engine
    mc = ModelConfig.qwen25vl_7b()
    mc = dataclasses.replace(mc, text=dataclasses.replace(mc.text, num_hidden_layers=2, vocab_size=4096),
                             vision=dataclasses.replace(mc.vision, depth=1, fullatt_block_indexes=(0,)),
                             image_token_id=4000, vision_start_token_id=4001, vision_end_token_id=4002,
                             eos_token_ids=(4003,), pad_token_id=4004)
    oc = Q.Config(vision=Q.VisionConfig(depth=1, fullatt_block_indexes=(0,), out_hidden_size=3584),
                  text=Q.TextConfig(hidden_size=3584, num_hidden_layers=2, num_attention_heads=28, num_key_value_heads=4,
                                    intermediate_size=18944, vocab_size=4096, tie_word_embeddings=False),
                  image_token_id=4000, vision_start_token_id=4001, vision_end_token_id=4002, eos_token_ids=(4003,),
                  pad_token_id=4004)
    w = {k: v for k, v in Q.synthetic_weights(oc, **W7).items() if not k.startswith("model.visual")}
    e = Engine(mc, device=0, max_seqs=2, max_ctx=512, max_patches=256, max_tile_side=512)
    try:
        e.fill_synthetic(**W7)
        ids = prng.uniform_ints(8, 150, 10, 3990).tolist()
        forced = [int(t) for t in prng.uniform_ints(9, 6, 10, 3990)]
        o32, o16 = Q.Qwen25VLOracle(oc, w, "fp32"), Q.Qwen25VLOracle(oc, w, "bf16")
        ref32 = [o32.prefill(ids)] + [o32.decode_step(t) for t in forced]
        ref16 = [o16.prefill(ids)] + [o16.decode_step(t) for t in forced]
        pos, delta = e.rope_index(ids, [])
        e.seq_reset(0)
        got = [e.prefill(0, ids, None, pos, delta).cpu().numpy()] + [e.decode_step(0, t).cpu().numpy() for t in forced]
        yard = max(float(np.abs(a - b).max()) for a, b in zip(ref16, ref32))
        worst = max(float(np.abs(a - b).max()) for a, b in zip(got, ref32))
        print(f"7B layer shape: max|engine - fp32 oracle| = {worst:.4f}, oracle bf16-vs-fp32 = {yard:.4f}")
        parity_ledger.record(worst, yard, "test_gpu_7b_shape.py:50")
        assert worst <= 2.0 * yard
        # batched decode of two chains on this shape agrees with the single-chain path within the same yardstick
        for s in (0, 1):
            e.seq_reset(s)
            e.prefill(s, ids[: 100 + 30 * s], None, pos[:, : 100 + 30 * s], delta, want_logits=False)
        lb = e.decode_batch([0, 1], [forced[0], forced[1]]).cpu().numpy()
        for s in (0, 1):
            e.seq_reset(s)
            e.prefill(s, ids[: 100 + 30 * s], None, pos[:, : 100 + 30 * s], delta, want_logits=False)
            ls = e.decode_step(s, forced[s]).cpu().numpy()
            parity_ledger.record(float(np.abs(ls - lb[s]).max()), yard, "test_gpu_7b_shape.py:60")
            assert float(np.abs(ls - lb[s]).max()) <= 2.0 * yard
        toks = e.generate(0, 8, ignore_eos=True)
        assert len(toks) == 8
    finally:
        e.close()
        torch.cuda.empty_cache()


def test_7b_shape_through_the_vit():
    """configs[4] end to end at reduced depth: a 28 x 28 view through the ViT whose merger projects to the 7B hidden size
    (out_hidden 3584: merger.mlp.2 is [3584, 5120]), image tokens scattered into a 7B-shape prompt, prefill and three
    teacher-forced decode steps -- engine vs fp32 oracle within 2x the oracle's own bf16-vs-fp32 error."""
    from oracle import frontend
    from zoomearth_amd.config import ModelConfig
    from zoomearth_amd.engine import Engine
    mc = ModelConfig.qwen25vl_7b()
    mc = dataclasses.replace(mc, text=dataclasses.replace(mc.text, num_hidden_layers=2, vocab_size=4096),
                             vision=dataclasses.replace(mc.vision, depth=2, fullatt_block_indexes=(1,)),
                             image_token_id=4000, vision_start_token_id=4001, vision_end_token_id=4002,
                             eos_token_ids=(4003,), pad_token_id=4004)
    oc = Q.Config(vision=Q.VisionConfig(depth=2, fullatt_block_indexes=(1,), out_hidden_size=3584),
                  text=Q.TextConfig(hidden_size=3584, num_hidden_layers=2, num_attention_heads=28, num_key_value_heads=4,
                                    intermediate_size=18944, vocab_size=4096, tie_word_embeddings=False),
                  image_token_id=4000, vision_start_token_id=4001, vision_end_token_id=4002, eos_token_ids=(4003,),
                  pad_token_id=4004)
    w = Q.synthetic_weights(oc, **W7)
    e = Engine(mc, device=0, max_seqs=1, max_ctx=512, max_patches=1024, max_tile_side=512)
    try:
        e.fill_synthetic(**W7)
        img = prng.synthetic_tile(21, 392, 392)
        pv, grid = e.preprocess_image(torch.from_numpy(img).cuda())
        want_pv, want_grid = frontend.image_to_pixel_values(img)
        assert tuple(grid) == tuple(want_grid) == (1, 28, 28) and np.array_equal(pv.cpu().numpy(), want_pv)
        emb = e.vit_forward(pv, [grid])
        o32, o16 = Q.Qwen25VLOracle(oc, w, "fp32"), Q.Qwen25VLOracle(oc, w, "bf16")
        v32, v16 = o32.vit_forward(want_pv, [want_grid]), o16.vit_forward(want_pv, [want_grid])
        assert emb.shape == (196, 3584)
        parity_ledger.record(float(np.abs(emb.float().cpu().numpy() - v32).max()), float(np.abs(v16 - v32).max()), "7B-shape ViT")
        assert float(np.abs(emb.float().cpu().numpy() - v32).max()) <= 2.0 * float(np.abs(v16 - v32).max())
        ids = prng.uniform_ints(8, 20, 10, 3990).tolist() + [4001] + [4000] * 196 + [4002] + prng.uniform_ints(9, 60, 10, 3990).tolist()
        forced = [int(t) for t in prng.uniform_ints(10, 3, 10, 3990)]
        ref32 = [o32.prefill(ids, pixel_values=want_pv, grid_thw=[want_grid])] + [o32.decode_step(t) for t in forced]
        ref16 = [o16.prefill(ids, pixel_values=want_pv, grid_thw=[want_grid])] + [o16.decode_step(t) for t in forced]
        pos, delta = e.rope_index(ids, [grid])
        e.seq_reset(0)
        got = [e.prefill(0, ids, emb, pos, delta).cpu().numpy()] + [e.decode_step(0, t).cpu().numpy() for t in forced]
        yard = max(float(np.abs(a - b).max()) for a, b in zip(ref16, ref32))
        worst = max(float(np.abs(a - b).max()) for a, b in zip(got, ref32))
        print(f"7B shape through the ViT: max|engine - fp32 oracle| = {worst:.4f}, oracle bf16-vs-fp32 = {yard:.4f}")
        parity_ledger.record(worst, yard, "test_gpu_7b_shape.py:108")
        assert worst <= 2.0 * yard
    finally:
        e.close()
        torch.cuda.empty_cache()
