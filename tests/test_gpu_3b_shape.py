"""GPU: the ZoomEarth-3B layer shape -- the dims `bench.py` runs -- through the C ABI against the oracle.

Every other oracle comparison uses the tiny config or the 7B text shape; this one pins the 3B-specific fast paths
(HF:models/qwen2_5_vl/modeling_qwen2_5_vl.py:408-471 ViT, :761-872 text model):
  * ViT: patch-embed K = 1176 on the register-staged GEMM, MLP width 3420 zero-padded to 3456, 16 heads x 80, one
    window-attention and one full-attention block, merger -- a 36 x 36 view (1296 patches -> 324 image tokens);
  * prefill of the benchmark's 802-token prompt: every `k_gemm_ring` tile policy at M = 802, GQA group 8 flash attention;
  * the batch-1 decode path: `k_gemv<2,1,1,4>` at K = 2048 / N = 22016 with the 16-row gate/up interleave, the K-split
    down projection, `k_attn_decode_split<24,1>` with 8 q heads per kv head;
  * the batched decode step (`ze_decode_batch`) at 1, 33 and 64 chains with ragged contexts: fragment-major qkv / o /
    gate-up / lm_head kernels, split-K ring down projection, `k_attn_decode_split<8,1>`;
  * the row-streaming family of that step (engines with more than 64 chain slots: `ze_launch_gemm_wide`, `k_rope_kv_batch`,
    the streaming attention kernel) at 1, 65, 128 and 256 chains -- the regime of the `stream` figure of bench.py.
Depth is reduced (2 ViT blocks, 2 decoder layers) and the vocabulary is 4096 so the numpy oracle finishes in seconds;
the per-layer arithmetic is the full-size one.

Tolerance (no HF fixture at this shape): the engine's output against the fp32 oracle must stay within 2x the oracle's
own bf16-vs-fp32 error on the same teacher-forced path -- the protocol of test_gpu_model.py / SURVEY.md 8 c.2."""
import dataclasses

import numpy as np
import pytest

import parity_ledger
import torch

from oracle import frontend, prng
from oracle import qwen25vl as Q

pytestmark = pytest.mark.gpu
W3 = dict(seed=5, std=0.02, matrix_gain=2.0, bias_std=0.02, norm_jitter=0.1)
IMG, VS, VE, EOS, PAD = 4000, 4001, 4002, 4003, 4004


def configs():
    from zoomearth_amd.config import ModelConfig
    mc = ModelConfig.zoomearth_3b()
    mc = dataclasses.replace(mc, text=dataclasses.replace(mc.text, num_hidden_layers=2, vocab_size=4096),
                             vision=dataclasses.replace(mc.vision, depth=2, fullatt_block_indexes=(1,)),
                             image_token_id=IMG, vision_start_token_id=VS, vision_end_token_id=VE,
                             eos_token_ids=(EOS,), pad_token_id=PAD)
    oc = Q.Config(vision=Q.VisionConfig(depth=2, fullatt_block_indexes=(1,)),
                  text=Q.TextConfig(num_hidden_layers=2, vocab_size=4096),
                  image_token_id=IMG, vision_start_token_id=VS, vision_end_token_id=VE, eos_token_ids=(EOS,),
                  pad_token_id=PAD)
    return mc, oc


def snapshot(o):
    return list(o.k_cache), list(o.v_cache), o.ctx, o.rope_delta


def restore(o, snap):
    o.k_cache, o.v_cache, o.ctx, o.rope_delta = list(snap[0]), list(snap[1]), snap[2], snap[3]


def test_3b_layer_shape_vit_prefill_decode_vs_oracle():
    from zoomearth_amd.engine import Engine
    mc, oc = configs()
    w = Q.synthetic_weights(oc, **W3)
    o32, o16 = Q.Qwen25VLOracle(oc, w, "fp32"), Q.Qwen25VLOracle(oc, w, "bf16")
    e = Engine(mc, device=0, max_seqs=256, max_ctx=1024, max_patches=2048, max_tile_side=1024)
    try:
        e.fill_synthetic(**W3)
        # ---- front-end + ViT on a 36 x 36 grid
        img = prng.synthetic_tile(11, 504, 504)
        pv, grid = e.preprocess_image(torch.from_numpy(img).cuda())
        want_pv, want_grid = frontend.image_to_pixel_values(img)
        assert tuple(grid) == tuple(want_grid) == (1, 36, 36)
        assert np.array_equal(pv.cpu().numpy(), want_pv)
        emb = e.vit_forward(pv, [grid])
        v32, v16 = o32.vit_forward(want_pv, [want_grid]), o16.vit_forward(want_pv, [want_grid])
        got_v = emb.float().cpu().numpy()
        yard_v = float(np.abs(v16 - v32).max())
        err_v = float(np.abs(got_v - v32).max())
        rms_v = float(np.sqrt(np.mean((got_v - v32) ** 2))), float(np.sqrt(np.mean((v16 - v32) ** 2)))
        print(f"3B ViT (1296 patches): max|engine - fp32| = {err_v:.4f} (oracle bf16-vs-fp32 {yard_v:.4f}), rms {rms_v[0]:.5f} ({rms_v[1]:.5f})")
        parity_ledger.record(err_v, yard_v, "3B ViT, 1296 patches")
        assert err_v <= 2.0 * yard_v and rms_v[0] <= 2.0 * rms_v[1]

        # ---- prefill of the benchmark prompt (21 + 1 + 324 + 1 + 455 = 802 tokens) and 6 teacher-forced decode steps
        n_img = grid[1] * grid[2] // 4
        ids = prng.uniform_ints(21, 21, 10, 3990).tolist() + [VS] + [IMG] * n_img + [VE] + \
            prng.uniform_ints(22, 455, 10, 3990).tolist()
        assert len(ids) == 802
        forced = [int(t) for t in prng.uniform_ints(23, 6, 10, 3990)]
        ref32 = [o32.prefill(ids, pixel_values=want_pv, grid_thw=[want_grid])]
        snap32 = snapshot(o32)
        ref32 += [o32.decode_step(t) for t in forced]
        ref16 = [o16.prefill(ids, pixel_values=want_pv, grid_thw=[want_grid])] + [o16.decode_step(t) for t in forced]
        pos, delta = e.rope_index(ids, [grid])
        e.seq_reset(0)
        got = [e.prefill(0, ids, emb, pos, delta).cpu().numpy()] + [e.decode_step(0, t).cpu().numpy() for t in forced]
        yard = max(float(np.abs(a - b).max()) for a, b in zip(ref16, ref32))
        worst = [float(np.abs(a - b).max()) for a, b in zip(got, ref32)]
        print(f"3B prefill(802) + 6 decode steps (batch-1 GEMV path): max|engine - fp32| per step = "
              f"{[round(x, 4) for x in worst]}, oracle bf16-vs-fp32 = {yard:.4f}")
        parity_ledger.record(max(worst), yard, "3B prefill(802) + 6 GEMV decode steps")
        assert max(worst) <= 2.0 * yard
        # greedy token agrees with the fp32 oracle wherever its top-1 / top-2 margin is decidable
        undecidable = 0
        for a, b in zip(got, ref32):
            top2 = np.partition(b, -2)[-2:]
            if top2[1] - top2[0] > 2.0 * 2.0 * yard:
                assert int(np.argmax(a)) == int(np.argmax(b))
            else:
                undecidable += 1
        print(f"3B greedy tokens: {len(got) - undecidable} of {len(got)} steps decidable and equal")
        # free-running greedy (SURVEY 8 c.2 protocol item iii): engine and fp32 oracle each follow their OWN tokens
        # (repetition penalty 1.3: diverse output); reported: the first step where they part and the oracle's top-1 / top-2 margin
        # there -- on random weights a margin below the bf16 noise is expected within a few dozen steps (the reference's own bf16 run
        # parts from its fp32 run after 15-43 steps in SURVEY's probe); asserted: every step BEFORE the first sub-margin step agrees
        n_free, pen = 24, 1.3
        restore(o32, snap32)
        seen, lg, ref_toks, margins = list(ids), ref32[0], [], []
        for _ in range(n_free):
            sc = Q.apply_repetition_penalty(lg, seen, pen)
            top2 = np.partition(sc, -2)[-2:]
            margins.append(float(top2[1] - top2[0]))
            tok = int(np.argmax(sc))
            ref_toks.append(tok)
            seen.append(tok)
            lg = o32.decode_step(tok)
        e.seq_reset(0)
        e.prefill(0, ids, emb, pos, delta)
        e.mark_seen(0, ids)
        got_toks = e.generate(0, n_free, repetition_penalty=pen, ignore_eos=True)
        first = next((i for i, (a, b) in enumerate(zip(got_toks, ref_toks)) if a != b), None)
        first_sub = next((i for i, m in enumerate(margins) if m <= 2.0 * 2.0 * yard), n_free)
        print(f"3B free-running greedy, {n_free} tokens, penalty {pen}: " +
              (f"identical to the fp32 oracle's ({len(set(ref_toks))} distinct tokens)" if first is None else
               f"first divergence at step {first}, oracle margin there {margins[first]:.4f} (bf16 yardstick {yard:.4f}; first sub-margin step {first_sub})"))
        parity_ledger.record(0.0 if first is None else margins[first], yard, f"free-running greedy, {n_free} tokens: oracle top-1/top-2 margin at the first "
                             f"divergence (step {first}; first sub-margin step {first_sub})", sub_margin_steps=int(sum(m <= 4.0 * yard for m in margins)), bar=4.0)
        assert first is None or first >= first_sub, (first, first_sub, margins[:first + 1])

        # ---- the same chain through the batched decode step at 1, 33 and 64 chains.  Chain c = the prompt (every
        # eighth chain a shorter prefix of it: ragged contexts), then two teacher-forced steps with its own tokens.
        nch = 256
        lens = [len(ids) if c % 8 != 1 else 640 + 2 * (c % 64) for c in range(nch)]
        t1 = [int(t) for t in prng.uniform_ints(31, nch, 10, 3990)]
        t2 = [int(t) for t in prng.uniform_ints(32, nch, 10, 3990)]
        want = {}
        snaps = {len(ids): snap32}
        for c in range(nch):
            if lens[c] not in snaps:
                o32.prefill(ids[: lens[c]], pixel_values=want_pv, grid_thw=[want_grid])
                snaps[lens[c]] = snapshot(o32)
            restore(o32, snaps[lens[c]])
            want[c] = (o32.decode_step(t1[c]), o32.decode_step(t2[c]))
        for c in range(nch):
            e.seq_reset(c)
            e.prefill(c, ids[: lens[c]], emb, pos[:, : lens[c]], delta, want_logits=False)
        # regime 0: the fragment kernels (what an engine with at most 64 slots runs); regime 1: the row-streaming family
        # (what this 256-slot engine runs by default) -- each at every batch size it serves, each batch-invariant
        for regime, sizes in ((0, (1, 33, 64)), (1, (1, 65, 128, 256))):
            e.set_decode_regime(regime)
            first = {}
            for n in sizes:
                chains = list(range(n))
                for c in chains:
                    e.seq_truncate(c, lens[c])
                l1 = e.decode_batch(chains, [t1[c] for c in chains]).cpu().numpy()
                l2 = e.decode_batch(chains, [t2[c] for c in chains]).cpu().numpy()
                errs = [max(float(np.abs(l1[c] - want[c][0]).max()), float(np.abs(l2[c] - want[c][1]).max())) for c in chains]
                print(f"3B batched decode, family {regime}, {n} chains: max|engine - fp32| = {max(errs):.4f} "
                      f"(2 x yardstick = {2 * yard:.4f})")
                parity_ledger.record(max(errs), yard, f"batched decode, regime {regime}, {n} chains")
                assert max(errs) <= 2.0 * yard, (regime, n, int(np.argmax(errs)), max(errs))
                # greedy token against the fp32 oracle wherever its margin is decidable
                for c in chains:
                    for got_l, ref_l in ((l1[c], want[c][0]), (l2[c], want[c][1])):
                        top2 = np.partition(ref_l, -2)[-2:]
                        if top2[1] - top2[0] > 2.0 * 2.0 * yard:
                            assert int(np.argmax(got_l)) == int(np.argmax(ref_l)), (regime, n, c)
                # batch invariance: chain 0 (and the ragged chain 1) alone and inside every larger batch, bit for bit
                for c in (0, 1):
                    if c < n:
                        if c in first:
                            assert np.array_equal(first[c][0], l1[c]) and np.array_equal(first[c][1], l2[c]), (regime, n, c)
                        else:
                            first[c] = (l1[c].copy(), l2[c].copy())
        e.set_decode_regime(-1)
    finally:
        e.close()
        torch.cuda.empty_cache()


def test_3b_layer_shape_row_streaming_at_the_headline_chain_counts():
    """bench.py's stream runs 2 x 768 chain slots, ~376 live chains per step.  Above 256 rows the step
    switches instances -- 384- / 512-row k_gemm_wstream passes for gate/up, the 128 x 256 two-launch down projection, the
    lm_head on ring tiles above 160 rows, the attention grid cut by live_parts -- and until round 4 none of that ran against
    the oracle inside a model.  Here: an engine with 768 slots, `ze_decode_batch` at 1 / 261 / 384 / 512 / 600 / 768 ragged chains
    (600, 768: gate/up and down on the 320- / 384-row ring tiles with K-steps of 32, where the stream runs most of its steps),
    two teacher-forced steps each, against the fp32 oracle (2 x the oracle's own bf16-vs-fp32 error), greedy token where
    the oracle's margin is decidable, chains 0 and 1 bit-identical alone and inside every batch."""
    from zoomearth_amd.engine import Engine
    mc, oc = configs()
    w = Q.synthetic_weights(oc, **W3)
    o32, o16 = Q.Qwen25VLOracle(oc, w, "fp32"), Q.Qwen25VLOracle(oc, w, "bf16")
    nch = 768
    e = Engine(mc, device=0, max_seqs=nch, max_ctx=1024, max_patches=2048, max_tile_side=1024)
    try:
        e.fill_synthetic(**W3)
        assert e.set_decode_regime(-1) == 1
        img = prng.synthetic_tile(11, 504, 504)
        pv, grid = e.preprocess_image(torch.from_numpy(img).cuda())
        want_pv, want_grid = frontend.image_to_pixel_values(img)
        emb = e.vit_forward(pv, [grid])
        n_img = grid[1] * grid[2] // 4
        ids = prng.uniform_ints(21, 21, 10, 3990).tolist() + [VS] + [IMG] * n_img + [VE] + \
            prng.uniform_ints(22, 455, 10, 3990).tolist()
        pos, delta = e.rope_index(ids, [grid])
        # ragged contexts: the full 802-token prompt, or one of 24 shorter prefixes of it (every third chain)
        lens = [len(ids) if c % 3 != 1 else 610 + 8 * (c % 24) for c in range(nch)]
        t1 = [int(t) for t in prng.uniform_ints(41, nch, 10, 3990)]
        t2 = [int(t) for t in prng.uniform_ints(42, nch, 10, 3990)]
        feats32 = o32.vit_forward(want_pv, [want_grid])
        feats16 = o16.vit_forward(want_pv, [want_grid])
        snaps, want, yard = {}, {}, 0.0
        for c in range(nch):
            if lens[c] not in snaps:
                o32.prefill(ids[: lens[c]], image_embeds=feats32, grid_thw=[want_grid])
                o16.prefill(ids[: lens[c]], image_embeds=feats16, grid_thw=[want_grid])
                snaps[lens[c]] = (snapshot(o32), snapshot(o16))
            restore(o32, snaps[lens[c]][0])
            want[c] = (o32.decode_step(t1[c]), o32.decode_step(t2[c]))
            if c < 48:  # the yardstick from the first chains (every distinct length is among them)
                restore(o16, snaps[lens[c]][1])
                b1, b2 = o16.decode_step(t1[c]), o16.decode_step(t2[c])
                yard = max(yard, float(np.abs(b1 - want[c][0]).max()), float(np.abs(b2 - want[c][1]).max()))
        for c in range(nch):
            e.seq_reset(c)
            e.prefill(c, ids[: lens[c]], emb, pos[:, : lens[c]], delta, want_logits=False)
        first = {}
        for n in (1, 261, 384, 512, 600, 768):
            chains = list(range(n))
            for c in chains:
                e.seq_truncate(c, lens[c])
            l1 = e.decode_batch(chains, [t1[c] for c in chains]).cpu().numpy()
            l2 = e.decode_batch(chains, [t2[c] for c in chains]).cpu().numpy()
            errs = [max(float(np.abs(l1[c] - want[c][0]).max()), float(np.abs(l2[c] - want[c][1]).max())) for c in chains]
            undecided = 0
            for c in chains:
                for got_l, ref_l in ((l1[c], want[c][0]), (l2[c], want[c][1])):
                    top2 = np.partition(ref_l, -2)[-2:]
                    if top2[1] - top2[0] > 2.0 * 2.0 * yard:
                        assert int(np.argmax(got_l)) == int(np.argmax(ref_l)), (n, c)
                    else:
                        undecided += 1
            print(f"3B row-streaming decode at {n} chains: max|engine - fp32| = {max(errs):.4f} (2 x yardstick = {2 * yard:.4f}), "
                  f"greedy token equal on {2 * n - undecided} of {2 * n} decidable steps")
            parity_ledger.record(max(errs), yard, f"row-streaming decode at {n} chains")
            assert max(errs) <= 2.0 * yard, (n, int(np.argmax(errs)), max(errs), yard)
            for c in (0, 1):
                if c < n:
                    if c in first:
                        assert np.array_equal(first[c][0], l1[c]) and np.array_equal(first[c][1], l2[c]), (n, c)
                    else:
                        first[c] = (l1[c].copy(), l2[c].copy())
            # rows of a batch do not depend on the batch: chain 300 at 384 = at 512 = at 768
            if n >= 384:
                if 300 in first:
                    assert np.array_equal(first[300][0], l1[300]) and np.array_equal(first[300][1], l2[300]), n
                else:
                    first[300] = (l1[300].copy(), l2[300].copy())
    finally:
        e.close()
        torch.cuda.empty_cache()


def test_vit_call_on_sixteen_images_equals_the_single_image_calls():
    """the stream calls the ViT on ~25 images at once (one multi-resolution call per admission round)
    while the oracle checks ran on one or two.  One call on 16 images of four resolutions (17,312 patches: the many-round
    GEMM grids, hundreds of attention segments): every image's features are the same bits as in a call of its own, and two
    of them (a square and a ragged grid) are compared with the fp32 oracle."""
    from zoomearth_amd.engine import Engine
    mc, oc = configs()
    w = Q.synthetic_weights(oc, **W3)
    o32, o16 = Q.Qwen25VLOracle(oc, w, "fp32"), Q.Qwen25VLOracle(oc, w, "bf16")
    e = Engine(mc, device=0, max_seqs=1, max_ctx=256, max_patches=17408, max_tile_side=1024)
    try:
        e.fill_synthetic(**W3)
        sizes = [(504, 504)] * 10 + [(504, 308)] * 2 + [(280, 504)] * 2 + [(504, 504)] + [(112, 56)]
        pvs, grids, hosts = [], [], []
        for i, (h, wd) in enumerate(sizes):
            img = prng.synthetic_tile(300 + i, h, wd)
            pv, grid = e.preprocess_image(torch.from_numpy(img).cuda())
            pvs.append(pv)
            grids.append(tuple(grid))
            hosts.append(img)
        assert sum(g[1] * g[2] for g in grids) == 17312
        both = e.vit_forward(torch.cat(pvs).contiguous(), grids)
        off = 0
        for i, (pv, g) in enumerate(zip(pvs, grids)):
            k = g[1] * g[2] // 4
            alone = e.vit_forward(pv, [g])
            assert torch.equal(alone, both[off: off + k]), i
            if i in (3, 11):
                want_pv, want_grid = frontend.image_to_pixel_values(hosts[i])
                v32, v16 = o32.vit_forward(want_pv, [want_grid]), o16.vit_forward(want_pv, [want_grid])
                got = both[off: off + k].float().cpu().numpy()
                err, yard = float(np.abs(got - v32).max()), float(np.abs(v16 - v32).max())
                print(f"image {i} of the 16-image call (grid {g}): max|engine - fp32| = {err:.4f} (oracle bf16-vs-fp32 {yard:.4f})")
                parity_ledger.record(err, yard, "test_gpu_3b_shape.py:299")
                assert err <= 2.0 * yard
            off += k
    finally:
        e.close()
        torch.cuda.empty_cache()


def test_demo_view_1036_px_vit_and_prefill_vs_oracle():
    """VERDICT r3 missing #4: src/demo.py looks at a <= 1024-px view (/root/reference/src/demo.py:86-93), which the
    processor turns into 1036 x 1036 = grid (1, 74, 74): 5476 patches, 100 windows of ragged sizes (74 is not a multiple of
    the 8-patch window), 1369 image tokens, a full-attention segment of 5476 keys and a 1847-token prompt -- against the
    oracle at the 3B layer shape (reduced depth)."""
    from zoomearth_amd.engine import Engine
    mc, oc = configs()
    w = Q.synthetic_weights(oc, **W3)
    o32, o16 = Q.Qwen25VLOracle(oc, w, "fp32"), Q.Qwen25VLOracle(oc, w, "bf16")
    e = Engine(mc, device=0, max_seqs=1, max_ctx=2048, max_patches=8192, max_tile_side=1100)
    try:
        e.fill_synthetic(**W3)
        img = prng.synthetic_tile(77, 1024, 1024)
        pv, grid = e.preprocess_image(torch.from_numpy(img).cuda())
        want_pv, want_grid = frontend.image_to_pixel_values(img)
        assert tuple(grid) == tuple(want_grid) == (1, 74, 74) and np.array_equal(pv.cpu().numpy(), want_pv)
        wi, cu = e.window_index([grid])
        assert len(cu) - 1 == 100
        emb = e.vit_forward(pv, [grid])
        v32, v16 = o32.vit_forward(want_pv, [want_grid]), o16.vit_forward(want_pv, [want_grid])
        got_v = emb.float().cpu().numpy()
        err_v, yard_v = float(np.abs(got_v - v32).max()), float(np.abs(v16 - v32).max())
        rms_v = float(np.sqrt(np.mean((got_v - v32) ** 2))), float(np.sqrt(np.mean((v16 - v32) ** 2)))
        print(f"demo view ViT (5476 patches, 100 windows): max|engine - fp32| = {err_v:.4f} (oracle bf16-vs-fp32 {yard_v:.4f}), "
              f"rms {rms_v[0]:.5f} ({rms_v[1]:.5f})")
        parity_ledger.record(err_v, yard_v, "test_gpu_3b_shape.py:331")
        assert err_v <= 2.0 * yard_v and rms_v[0] <= 2.0 * rms_v[1]
        ids = prng.uniform_ints(51, 21, 10, 3990).tolist() + [VS] + [IMG] * 1369 + [VE] + prng.uniform_ints(52, 455, 10, 3990).tolist()
        forced = [int(t) for t in prng.uniform_ints(53, 3, 10, 3990)]
        ref32 = [o32.prefill(ids, image_embeds=v32, grid_thw=[want_grid])] + [o32.decode_step(t) for t in forced]
        ref16 = [o16.prefill(ids, image_embeds=v16, grid_thw=[want_grid])] + [o16.decode_step(t) for t in forced]
        pos, delta = e.rope_index(ids, [grid])
        e.seq_reset(0)
        got = [e.prefill(0, ids, emb, pos, delta).cpu().numpy()] + [e.decode_step(0, t).cpu().numpy() for t in forced]
        yard = max(float(np.abs(a - b).max()) for a, b in zip(ref16, ref32))
        worst = [float(np.abs(a - b).max()) for a, b in zip(got, ref32)]
        print(f"demo view prefill({len(ids)}) + 3 decode steps: max|engine - fp32| = {[round(x, 4) for x in worst]}, "
              f"oracle bf16-vs-fp32 = {yard:.4f}")
        parity_ledger.record(max(worst), yard, "test_gpu_3b_shape.py:343")
        assert max(worst) <= 2.0 * yard
        for a, b in zip(got, ref32):
            top2 = np.partition(b, -2)[-2:]
            if top2[1] - top2[0] > 2.0 * 2.0 * yard:
                assert int(np.argmax(a)) == int(np.argmax(b))
    finally:
        e.close()
        torch.cuda.empty_cache()
