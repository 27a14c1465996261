"""GPU: the WHOLE depth of the ZoomEarth-3B shape -- 32 ViT blocks, 36 decoder layers, every per-layer dimension of
the benchmark -- against the oracle (every other oracle comparison runs 2-4 layers, and a bias that
grows with depth and is common to the engine's two decode paths would be invisible to the self-consistency checks of
test_gpu_full_size.py).  Only the vocabulary is reduced (4096: the lm_head is one GEMM whatever its width), so that the
numpy oracle finishes in a few minutes.

Path checked (HF:models/qwen2_5_vl/modeling_qwen2_5_vl.py:408-471 ViT, :761-872 text model, :1386-1387 logits):
the 36 x 36 view through the front-end and the ViT, the benchmark's 802-token prompt through `ze_prefill`, then four
teacher-forced decode steps through (a) the single-chain GEMV path `ze_decode_step`, (b) `ze_decode_batch` on the
fragment kernels and (c) `ze_decode_batch` on the row-streaming family (the regime of bench.py's stream figure).

Weights: numpy.random (PCG64, seed 2024) pools, every tensor a window of a pool at an offset derived from its name --
loaded into the engine through `ze_load_weight` and handed to BOTH oracles as they are (bf16-representable float32, no
per-oracle copy: one 14-GB set).  Bar: max|engine - oracle fp32| <= 2 x max|oracle bf16 - oracle fp32| per step (and the
same for the rms), the protocol of SURVEY.md 8 c.2."""
import dataclasses
import time
import zlib

import numpy as np
import pytest

import parity_ledger
import torch

from oracle import frontend, prng
from oracle import qwen25vl as Q

pytestmark = pytest.mark.gpu
IMG, VS, VE, EOS, PAD = 4000, 4001, 4002, 4003, 4004


def configs():
    from zoomearth_amd.config import ModelConfig
    mc = ModelConfig.zoomearth_3b()
    mc = dataclasses.replace(mc, text=dataclasses.replace(mc.text, vocab_size=4096),
                             image_token_id=IMG, vision_start_token_id=VS, vision_end_token_id=VE,
                             eos_token_ids=(EOS,), pad_token_id=PAD)
    oc = Q.Config(text=Q.TextConfig(vocab_size=4096), image_token_id=IMG, vision_start_token_id=VS,
                  vision_end_token_id=VE, eos_token_ids=(EOS,), pad_token_id=PAD)
    assert oc.text.num_hidden_layers == 36 and oc.vision.depth == 32
    return mc, oc


def pooled_weights(oc, seed=2024):
    """HF-keyed float32 tensors with bf16-representable values, every one a VIEW into one of four pools."""
    rng = np.random.default_rng(seed)
    n_pool = 1 << 26  # 64 M values: the largest matrix of the shape is 22.5 M
    mat = Q.bf16_round(rng.standard_normal(n_pool, dtype=np.float32) * np.float32(0.03))
    emb = Q.bf16_round(rng.standard_normal(1 << 24, dtype=np.float32) * np.float32(0.02))
    small = Q.bf16_round(rng.standard_normal(1 << 16, dtype=np.float32) * np.float32(0.02))
    norm = Q.bf16_round(np.float32(1.0) + np.float32(0.1) * rng.standard_normal(1 << 16, dtype=np.float32))
    out = {}
    for name, shape in Q.weight_shapes(oc).items():
        n = int(np.prod(shape))
        if name.endswith("norm1.weight") or name.endswith("norm2.weight") or name.endswith("ln_q.weight") \
                or name.endswith("layernorm.weight") or name.endswith("language_model.norm.weight"):
            pool = norm
        elif name.endswith(".bias"):
            pool = small
        elif name.endswith("embed_tokens.weight") or name == "lm_head.weight":
            pool = emb
        else:
            pool = mat
        off = zlib.crc32(name.encode()) % (len(pool) - n)
        out[name] = pool[off:off + n].reshape(shape)
    return out


def test_full_depth_vit_prefill_and_every_decode_path_vs_oracle():
    from zoomearth_amd.engine import Engine
    mc, oc = configs()
    t0 = time.time()
    w = pooled_weights(oc)
    o32 = Q.Qwen25VLOracle(oc, w, "fp32", share_weights=True)
    o16 = Q.Qwen25VLOracle(oc, w, "bf16", share_weights=True)
    e = Engine(mc, device=0, max_seqs=66, max_ctx=1024, max_patches=2048, max_tile_side=1024)
    try:
        e.load_state_dict(w.items())
        print(f"weights: {time.time() - t0:.1f}s")
        img = prng.synthetic_tile(11, 504, 504)
        pv, grid = e.preprocess_image(torch.from_numpy(img).cuda())
        want_pv, want_grid = frontend.image_to_pixel_values(img)
        assert tuple(grid) == tuple(want_grid) == (1, 36, 36) and np.array_equal(pv.cpu().numpy(), want_pv)

        # ---- ViT, all 32 blocks
        t0 = time.time()
        emb = e.vit_forward(pv, [grid])
        v32, v16 = o32.vit_forward(want_pv, [want_grid]), o16.vit_forward(want_pv, [want_grid])
        got_v = emb.float().cpu().numpy()
        yard_v, err_v = float(np.abs(v16 - v32).max()), float(np.abs(got_v - v32).max())
        rms = lambda a: float(np.sqrt(np.mean(np.square(a, dtype=np.float64))))  # noqa: E731
        print(f"full-depth ViT (32 blocks, 1296 patches): max|engine - fp32| = {err_v:.4f}, oracle bf16-vs-fp32 = {yard_v:.4f}; "
              f"rms {rms(got_v - v32):.5f} / {rms(v16 - v32):.5f}; scale {float(np.abs(v32).max()):.2f} ({time.time() - t0:.0f}s)")
        parity_ledger.record(err_v, yard_v, "test_gpu_full_depth.py:93")
        assert err_v <= 2.0 * yard_v and rms(got_v - v32) <= 2.0 * rms(v16 - v32)

        # ---- prefill of the benchmark prompt through all 36 layers + 4 teacher-forced steps.  Both oracles take the
        # fp32 oracle's image features (as the engine takes its own): the text model is measured on its own inputs.
        n_img = grid[1] * grid[2] // 4
        ids = prng.uniform_ints(21, 21, 10, 3990).tolist() + [VS] + [IMG] * n_img + [VE] + \
            prng.uniform_ints(22, 455, 10, 3990).tolist()
        assert len(ids) == 802
        forced = [int(t) for t in prng.uniform_ints(23, 4, 10, 3990)]
        t0 = time.time()
        ref32 = [o32.prefill(ids, image_embeds=v32, grid_thw=[want_grid])] + [o32.decode_step(t) for t in forced]
        ref16 = [o16.prefill(ids, image_embeds=v16, grid_thw=[want_grid])] + [o16.decode_step(t) for t in forced]
        print(f"oracle prefill + 4 steps, fp32 and bf16: {time.time() - t0:.0f}s")
        yard = max(float(np.abs(a - b).max()) for a, b in zip(ref16, ref32))
        yard_rms = max(rms(a - b) for a, b in zip(ref16, ref32))
        pos, delta = e.rope_index(ids, [grid])

        def check(name, got):
            worst = [float(np.abs(a - b).max()) for a, b in zip(got, ref32)]
            worst_rms = [rms(a - b) for a, b in zip(got, ref32)]
            print(f"full depth, {name}: max|engine - fp32| per step = {[round(x, 4) for x in worst]} (oracle bf16-vs-fp32 "
                  f"{yard:.4f}), rms {[round(x, 5) for x in worst_rms]} ({yard_rms:.5f}); logit scale {float(np.abs(ref32[0]).max()):.2f}")
            parity_ledger.record(max(worst), yard, "test_gpu_full_depth.py:115")
            assert max(worst) <= 2.0 * yard and max(worst_rms) <= 2.0 * yard_rms, name
            for a, b in zip(got, ref32):  # greedy token wherever the fp32 margin is decidable
                top2 = np.partition(b, -2)[-2:]
                if top2[1] - top2[0] > 2.0 * 2.0 * yard:
                    assert int(np.argmax(a)) == int(np.argmax(b)), name

        e.seq_reset(0)
        pre = e.prefill(0, ids, emb, pos, delta).cpu().numpy()
        check("prefill + GEMV decode (ze_decode_step)", [pre] + [e.decode_step(0, t).cpu().numpy() for t in forced])
        for regime, label in ((0, "fragment kernels"), (1, "row-streaming family")):
            e.set_decode_regime(regime)
            e.seq_truncate(0, len(ids))
            got = [pre] + [e.decode_batch([0], [t]).cpu().numpy()[0] for t in forced]
            check(f"ze_decode_batch, {label}, chain alone", got)
            # ... and the same chain among 65 (row streaming) / 33 (fragment) others: the same bits
            n_other = 65 if regime else 33
            for c in range(1, n_other):
                e.seq_reset(c)
                e.seq_copy_prefix(c, 0, 700 + c)
            e.seq_truncate(0, len(ids))
            chains = list(range(n_other))
            among = []
            for t in forced:
                among.append(e.decode_batch(chains, [t] + [17 + c for c in chains[1:]]).cpu().numpy()[0])
            for a, b in zip(among, got[1:]):
                assert np.array_equal(a, b), label
        e.set_decode_regime(-1)
    finally:
        e.close()
        torch.cuda.empty_cache()
