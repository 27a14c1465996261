#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE in the build container.

This script is the only place that touches /root/reference or imports `transformers`' Qwen2.5-VL
model.  It is run by hand in the build container (`python tests/golden/make_fixtures.py`); its
outputs (small .json / .npz files) are committed, and every test reads only those.  It never runs on
the GPU box (no /root/reference there).

Fixture sets:
  1. host_helpers.json   resize_image / cut_image / extract_bbox / extract_answer / prompts / record,
                         produced by importing /root/reference/src/eval/infer.py and src/demo.py.
  2. bicubic.npz         Pillow `Image.resize(..., BICUBIC)` outputs for seeded synthetic tiles
                         (inputs are regenerated from oracle/prng.py, so only outputs are stored;
                         the 5000x5000 -> 512x512 case stores SHA-256 + sampled rows).
  3. indices.json        smart_resize, get_vision_window_index, get_vision_cu_seqlens,
                         get_vision_position_ids, get_rope_index from the installed transformers.
  4. pixel_values.npz    Qwen2VLImageProcessorPil outputs (SHA-256 + sampled rows).
  5. tiny_chain.npz      tiny-config Qwen2_5_VLForConditionalGeneration (weights from the repo PRNG):
                         ViT output, prefill logits, greedy tokens and per-step logits of a scripted
                         two-stage zoom chain, in fp32 and bf16.
  5b. heads_chain.npz    the same model class at the 3B HEAD STRUCTURE (16 q / 2 kv heads x 128, 16 ViT heads x 80;
                         two layers): sampled ViT rows, per-step logits and tokens of one stage (`heads`, not in the default list).
"""
from __future__ import annotations

import ast
import hashlib
import importlib.util
import io
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REF = "/root/reference"

from oracle import frontend, indices, prng, qwen25vl  # noqa: E402


def sha(a) -> str:
    if isinstance(a, str):
        a = a.encode("utf-8")
    elif isinstance(a, np.ndarray):
        a = np.ascontiguousarray(a).tobytes()
    return hashlib.sha256(a).hexdigest()


def load_ref(rel, name):
    sys.modules.setdefault("shortuuid", types.ModuleType("shortuuid"))  # imported, unused (infer.py:6)
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def coord_image(w, h):
    """RGB image whose pixel encodes its own coordinates (tests rebuild it identically)."""
    y, x = np.mgrid[0:h, 0:w]
    return np.stack([x & 255, y & 255, (((x >> 8) & 15) << 4) | ((y >> 8) & 15)], axis=-1).astype(np.uint8)


# ----------------------------------------------------------------------------- 1. host helpers
def make_host_helpers():
    from PIL import Image

    infer = load_ref("src/eval/infer.py", "ref_infer")
    demo = load_ref("src/demo.py", "ref_demo")
    out = {}

    sizes = [(5000, 5000), (5000, 3000), (4999, 5000), (300, 200), (512, 512), (513, 100), (1024, 1024),
             (1025, 7), (2048, 1536), (640, 900), (1100, 900)]
    rows = []
    for w, h in sizes:
        img = Image.new("RGB", (w, h))
        a, s = infer.resize_image(img)
        b = demo.resize_image(img)
        rows.append(dict(size=[w, h], infer_size=list(a.size), infer_scale=s, demo_size=list(b.size)))
    out["resize_image"] = rows

    cases = [
        ((5000, 5000), (100, 100, 200, 200)), ((5000, 5000), (1000, 1000, 1600, 1100)),
        ((5000, 5000), (1000, 1000, 2000, 2200)), ((5000, 5000), (-50, -50, 900, 900)),
        ((5000, 5000), (4000, 4000, 6000, 6000)), ((300, 300), (10, 10, 50, 50)),
        ((5000, 5000), (4900.7, 4800.2, 4990.9, 4999.5)), ((5000, 3000), (10.5, 2900.9, 80.1, 2990.0)),
        ((700, 400), (600, 300, 690, 390)), ((5000, 5000), (2500, 2500, 3012, 3011)),
        ((5000, 5000), (2500, 2500, 3012, 3012)), ((640, 900), (100.0, 200.0, 180.0, 260.0)),
        ((5000, 5000), (0, 0, 0, 0)), ((5000, 5000), (300, 200, 100, 50)),
    ]
    rows = []
    for (w, h), bbox in cases:
        src = coord_image(w, h)
        img = Image.fromarray(src)
        got = np.array(infer.cut_image(img, list(bbox)))
        got2 = np.array(demo.cut_image(img, list(bbox)))
        assert np.array_equal(got, got2)
        rows.append(dict(size=[w, h], bbox=list(bbox), out_size=[int(got.shape[1]), int(got.shape[0])],
                         sha256=sha(got)))
    out["cut_image"] = rows

    texts = [
        'x [{"bbox_2d": [10, 20, 30, 40], "label": "a"}] y',
        '"bbox_2d": [1,2,3]',
        '"bbox_2d" :\n [ 1.5 , 2 , 3.25, 4 ]',
        '"bbox_2d": [a, b]',
        'no box here',
        '"bbox_2d": [1,2,3,4] and "bbox_2d": [5, 6, 7, 8]',
        '"bbox_2d": []',
        '"bbox_2d": [ 100,200 ,\n300, 400 ] trailing ] bracket',
        '<think>"bbox_2d": [-5, 1e2, 3, 4]</think>',
    ]
    rows = []
    for t in texts:
        for scale in (1, 9.765625, 5000 / 512):
            rows.append(dict(text=t, scale=scale, infer=infer.extract_bbox(t, scale), demo=demo.extract_bbox(t, scale)))
    out["extract_bbox"] = rows

    texts = ["<answer> yes </answer>", "a<answer>two words</answer>b<answer>second</answer>", "<answer>\nmulti\nline\n</answer>",
             "none", "<answer></answer>", "<answer>  </answer>", "<think>x</think><answer>Bridge</answer>"]
    out["extract_answer"] = [dict(text=t, answer=infer.extract_answer(t)) for t in texts]

    # prompt constants: demo.py module constants; infer.py locals of eval_model_lora (via ast)
    src = open(os.path.join(REF, "src/eval/infer.py"), encoding="utf-8").read()
    consts = {}
    for node in ast.walk(ast.parse(src)):
        if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name) \
                and node.targets[0].id in ("prefix", "instruction") and isinstance(node.value, ast.Constant):
            consts[node.targets[0].id] = node.value.value
    assert consts["prefix"] == demo.PREFIX and consts["instruction"] == demo.INSTRUCTION
    q = "Are there any building on the top-right island?"
    stage1 = consts["prefix"] + q + consts["instruction"]
    o1 = '<think>scene. [{"bbox_2d": [10,20,30,40], "label": "island"}]</think><answer>yes</answer>'
    stage2 = stage1 + o1.split("<answer>")[0] + "<|vision_start|><|image_pad|><|vision_end|>"
    out["prompts"] = dict(prefix_sha256=sha(consts["prefix"]), prefix_len=len(consts["prefix"]),
                          instruction_sha256=sha(consts["instruction"]), instruction_len=len(consts["instruction"]),
                          question=q, stage1_sha256=sha(stage1), output1=o1, stage2_sha256=sha(stage2),
                          stage2_len=len(stage2))

    sample = dict(question_id=17, ground_truth="yes", category="object", type="region",
                  image_name="dir/abc.tif", bbox=[1, 2, 3, 4], question=q)
    buf = io.StringIO()
    infer.record(buf, q, sample, sample, o1, "<think>z</think><answer>yeés</answer>", False)
    buf2 = io.StringIO()
    infer.record(buf2, q, sample, sample, "no box", "", True)
    out["record"] = dict(sample=sample, output1=o1, output2="<think>z</think><answer>yeés</answer>",
                         line=buf.getvalue(), line_error=buf2.getvalue())
    with open(os.path.join(HERE, "host_helpers.json"), "w", encoding="utf-8") as f:
        json.dump(out, f, indent=1, ensure_ascii=False)
    print("host_helpers.json", len(json.dumps(out)))


# ----------------------------------------------------------------------------- 2. bicubic
BICUBIC_CASES = [  # (seed, in_h, in_w, out_w, out_h)
    (21, 100, 80, 29, 37), (22, 500, 500, 51, 51), (23, 512, 512, 504, 504), (24, 517, 300, 308, 504),
    (25, 1000, 1000, 512, 512), (26, 307, 512, 504, 308), (27, 64, 64, 128, 100), (28, 777, 1001, 512, 397),
    (29, 600, 1000, 512, 307), (30, 33, 500, 500, 33),
]


def make_bicubic():
    from PIL import Image

    out = {}
    for seed, h, w, ow, oh in BICUBIC_CASES:
        img = prng.synthetic_tile(seed, h, w)
        ref = np.array(Image.fromarray(img).resize((ow, oh), Image.BICUBIC))
        assert np.array_equal(ref, frontend.resize_bicubic(img, ow, oh)), (seed, h, w)
        key = f"s{seed}_{h}x{w}_to_{oh}x{ow}"
        if ref.size <= 60000:
            out[key] = ref
        else:  # keep the fixture small: digest + every 16th row
            out[key + "_sha256"] = np.frombuffer(sha(ref).encode(), dtype=np.uint8)
            out[key + "_rows16"] = ref[::16].copy()
    img = prng.synthetic_tile(1000, 5000, 5000)
    ref = np.array(Image.fromarray(img).resize((512, 512), Image.BICUBIC))
    out["big_5000_sha256"] = np.frombuffer(sha(ref).encode(), dtype=np.uint8)
    out["big_5000_rows"] = ref[::64].copy()
    img2 = img[:3000]
    ref2 = np.array(Image.fromarray(img2).resize((512, 307), Image.BICUBIC))
    out["big_5000x3000_sha256"] = np.frombuffer(sha(ref2).encode(), dtype=np.uint8)
    # crop + resize of a large box (cut_image "large bbox" branch then resize_image)
    crop = np.array(Image.fromarray(img).crop((1000, 1200, 3500, 3300)).resize((512, 430), Image.BICUBIC))
    out["crop_1000_1200_3500_3300_to_430x512_sha256"] = np.frombuffer(sha(crop).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "bicubic.npz"), **out)
    print("bicubic.npz", os.path.getsize(os.path.join(HERE, "bicubic.npz")))


# ----------------------------------------------------------------------------- 3. indices
def make_indices():
    import torch
    from transformers import vision_utils as vu
    from transformers.models.qwen2_vl.image_processing_pil_qwen2_vl import smart_resize

    from hf_bridge import hf_model

    out = {}
    rows = []
    for h, w, mx in [(5000, 5000, 12845056), (5000, 5000, 1003520), (512, 512, 12845056), (1024, 1024, 12845056),
                     (448, 448, 12845056), (300, 512, 12845056), (307, 512, 12845056), (20, 30, 12845056),
                     (511, 512, 12845056), (14, 2000, 12845056), (42, 70, 12845056), (98, 126, 12845056)]:
        rows.append(dict(h=h, w=w, max_pixels=mx, out=list(smart_resize(h, w, factor=28, min_pixels=3136, max_pixels=mx))))
    out["smart_resize"] = rows

    grids = [[[1, 32, 32]], [[1, 36, 36]], [[1, 36, 22]], [[1, 74, 74]], [[1, 36, 36], [1, 36, 36]],
             [[1, 8, 12]], [[1, 36, 30], [1, 22, 36]], [[1, 2, 2]], [[1, 16, 16]], [[1, 18, 10]]]
    rows = []
    for g in grids:
        gt = torch.tensor(g)
        wi, cw = vu.get_vision_window_index(gt, spatial_merge_size=2, window_size=112, patch_size=14)
        cu = vu.get_vision_cu_seqlens(gt)
        pid = vu.get_vision_position_ids(gt, 2)
        rows.append(dict(grid=g, window_index=wi.tolist(), cu_window_seqlens=cw.tolist(), cu_seqlens=cu.tolist(),
                         position_ids_sha256=sha(pid.numpy().astype(np.int64)), position_ids_head=pid[:24].tolist()))
    out["vision"] = rows

    cfg = qwen25vl.tiny_config()
    model = hf_model(cfg, qwen25vl.synthetic_weights(cfg, seed=1))
    I, S, E = cfg.image_token_id, cfg.vision_start_token_id, cfg.vision_end_token_id
    rows = []

    def case(ids_rows, grids, pad_left):
        L = max(len(r) for r in ids_rows) + pad_left
        ids = np.full((len(ids_rows), L), cfg.pad_token_id, dtype=np.int64)
        am = np.zeros((len(ids_rows), L), dtype=np.int64)
        for i, r in enumerate(ids_rows):
            ids[i, L - len(r):] = r
            am[i, L - len(r):] = 1
        pos, delta = model.model.get_rope_index(torch.from_numpy(ids), mm_token_type_ids=torch.from_numpy((ids == I).astype(np.int32)),
                                                image_grid_thw=torch.tensor(grids), attention_mask=torch.from_numpy(am))
        rows.append(dict(input_ids=ids.tolist(), attention_mask=am.tolist(), grids=grids, position_ids=pos.tolist(),
                         rope_deltas=delta.tolist()))

    one = [5, 6, S] + [I] * 6 + [E, 7, 8, 9]
    two = [5, S] + [I] * 6 + [E, 11, 12, 13, S] + [I] * 20 + [E, 14]
    case([one], [[1, 4, 6]], 0)
    case([two], [[1, 6, 4], [1, 8, 10]], 0)
    case([one, one[:-2]], [[1, 4, 6], [1, 6, 4]], 0)
    case([two, one], [[1, 4, 6], [1, 10, 8], [1, 6, 4]], 3)
    out["rope_index"] = rows
    with open(os.path.join(HERE, "indices.json"), "w") as f:
        json.dump(out, f)
    print("indices.json", os.path.getsize(os.path.join(HERE, "indices.json")))


# ----------------------------------------------------------------------------- 4. pixel_values
PIXEL_CASES = [(41, 512, 512), (42, 300, 512), (43, 448, 448), (44, 112, 168), (45, 511, 512), (46, 40, 50)]


def make_pixel_values():
    from PIL import Image
    from transformers.models.qwen2_vl.image_processing_pil_qwen2_vl import Qwen2VLImageProcessorPil

    ip = Qwen2VLImageProcessorPil(max_pixels=128 * 128 * 28 * 28)
    out = {}
    for seed, h, w in PIXEL_CASES:
        img = prng.synthetic_tile(seed, h, w)
        r = ip(images=[Image.fromarray(img)], return_tensors="np")
        pv, grid = r["pixel_values"], r["image_grid_thw"]
        mine, g = frontend.image_to_pixel_values(img)
        assert np.array_equal(pv, mine) and list(g) == grid[0].tolist()
        key = f"s{seed}_{h}x{w}"
        out[key + "_grid"] = grid
        out[key + "_sha256"] = np.frombuffer(sha(pv.astype(np.float32)).encode(), dtype=np.uint8)
        out[key + "_rows"] = pv[:: max(1, pv.shape[0] // 7)][:8].astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "pixel_values.npz"), **out)
    print("pixel_values.npz", os.path.getsize(os.path.join(HERE, "pixel_values.npz")))


# ----------------------------------------------------------------------------- 5. tiny chain
CHAIN = dict(weight_seed=1, std=0.02, matrix_gain=4.0, bias_std=0.02, norm_jitter=0.1, tile_seed=77, tile_h=900,
             tile_w=1100, text_seed=5, n_text_a=6, n_text_b=18, n1=24, n2=16, repetition_penalty=1.3,
             bbox=[300.0, 200.0, 420.0, 330.0])


def chain_inputs(cfg, n_img_tokens_view):
    a = prng.uniform_ints(CHAIN["text_seed"], CHAIN["n_text_a"], 10, 2000).tolist()
    b = prng.uniform_ints(CHAIN["text_seed"] + 1, CHAIN["n_text_b"], 10, 2000).tolist()
    return a + [cfg.vision_start_token_id] + [cfg.image_token_id] * n_img_tokens_view + [cfg.vision_end_token_id] + b


def make_tiny_chain():
    import torch
    from PIL import Image

    from hf_bridge import hf_model

    infer = load_ref("src/eval/infer.py", "ref_infer")
    cfg = qwen25vl.tiny_config()
    w = qwen25vl.synthetic_weights(cfg, seed=CHAIN["weight_seed"], std=CHAIN["std"], matrix_gain=CHAIN["matrix_gain"],
                                   bias_std=CHAIN["bias_std"], norm_jitter=CHAIN["norm_jitter"])
    tile = prng.synthetic_tile(CHAIN["tile_seed"], CHAIN["tile_h"], CHAIN["tile_w"])
    pil_tile = Image.fromarray(tile)
    view, scale = infer.resize_image(pil_tile)  # reference host code: <=512 px view
    crop, _ = infer.resize_image(infer.cut_image(pil_tile, CHAIN["bbox"]))
    view, crop = np.array(view), np.array(crop)
    pv_v, g_v = frontend.image_to_pixel_values(view)
    pv_c, g_c = frontend.image_to_pixel_values(crop)
    ids1 = chain_inputs(cfg, g_v[1] * g_v[2] // 4)
    out = dict(view_sha256=np.frombuffer(sha(view).encode(), dtype=np.uint8),
               crop_sha256=np.frombuffer(sha(crop).encode(), dtype=np.uint8),
               grid_view=np.array(g_v), grid_crop=np.array(g_c), ids1=np.array(ids1), scale=np.array(scale))

    m32 = hf_model(cfg, w, torch.float32)
    m16 = hf_model(cfg, w, torch.bfloat16)

    def hf_inputs(ids, pvs, grids):
        t = torch.tensor([ids])
        return dict(input_ids=t, attention_mask=torch.ones_like(t), pixel_values=torch.from_numpy(np.concatenate(pvs)),
                    image_grid_thw=torch.tensor(grids), mm_token_type_ids=(t == cfg.image_token_id).int())

    def run_stage(tag, ids, pvs, grids, n_new):
        inp = hf_inputs(ids, pvs, grids)
        with torch.no_grad():
            g = m32.generate(**inp, max_new_tokens=n_new, do_sample=False, num_beams=1,
                             repetition_penalty=CHAIN["repetition_penalty"], output_logits=True,
                             return_dict_in_generate=True)
        toks = g.sequences[0, len(ids):].tolist()
        logits32 = torch.stack([x[0] for x in g.logits]).float().numpy()
        # teacher-forced bf16 (and fp32 cross-check) logits along the fp32 greedy path: one full forward
        full = ids + toks[:-1]
        for name, m in (("bf16", m16), ("fp32full", m32)):
            fi = hf_inputs(full, pvs, grids)
            with torch.no_grad():
                lg = m(**fi).logits[0, len(ids) - 1:].float().numpy()
            out[f"{tag}_logits_{name}"] = lg.astype(np.float32)
        with torch.no_grad():
            vit32 = m32.model.visual(torch.from_numpy(np.concatenate(pvs)), grid_thw=torch.tensor(grids)).pooler_output.numpy()
            vit16 = m16.model.visual(torch.from_numpy(np.concatenate(pvs)).bfloat16(), grid_thw=torch.tensor(grids)
                                     ).pooler_output.float().numpy()
            g16 = m16.generate(**inp, max_new_tokens=n_new, do_sample=False, num_beams=1,
                               repetition_penalty=CHAIN["repetition_penalty"])
        out[f"{tag}_tokens_fp32"] = np.array(toks)
        out[f"{tag}_tokens_bf16_free"] = g16[0, len(ids):].numpy()
        out[f"{tag}_logits_fp32"] = logits32
        out[f"{tag}_vit_fp32"] = vit32[:: max(1, vit32.shape[0] // 16)][:20].astype(np.float32)
        out[f"{tag}_vit_fp32_sha256"] = np.frombuffer(sha(vit32.astype(np.float32)).encode(), dtype=np.uint8)
        out[f"{tag}_vit_bf16"] = vit16[:: max(1, vit16.shape[0] // 16)][:20].astype(np.float32)
        print(tag, "distinct tokens", len(set(toks)), "of", len(toks),
              "max|fp32full-fp32|", float(np.abs(out[f"{tag}_logits_fp32full"] - logits32).max()),
              "max|bf16-fp32|", float(np.abs(out[f"{tag}_logits_bf16"] - logits32).max()))
        return toks

    t1 = run_stage("s1", ids1, [pv_v], [list(g_v)], CHAIN["n1"])
    ids2 = ids1 + t1 + [cfg.vision_start_token_id] + [cfg.image_token_id] * (g_c[1] * g_c[2] // 4) + [cfg.vision_end_token_id]
    out["ids2"] = np.array(ids2)
    run_stage("s2", ids2, [pv_v, pv_c], [list(g_v), list(g_c)], CHAIN["n2"])

    # oracle self-check against the reference outputs (the fixture pins the oracle)
    o32 = qwen25vl.Qwen25VLOracle(cfg, w, "fp32")
    r = qwen25vl.greedy_generate(o32, ids1, pv_v, [g_v], CHAIN["n1"], CHAIN["repetition_penalty"], eos_token_ids=())
    print("oracle fp32 stage1 tokens equal:", r["tokens"] == t1, "max logit err", float(np.abs(r["logits"] - out["s1_logits_fp32"]).max()))
    out["chain_json"] = np.frombuffer(json.dumps(CHAIN).encode(), dtype=np.uint8)
    for k in list(out):
        if "logits" in k:
            out[k] = out[k].astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "tiny_chain.npz"), **out)
    print("tiny_chain.npz", os.path.getsize(os.path.join(HERE, "tiny_chain.npz")))


# ----------------------------------------------------------------------------- 5b. the 3B head structure
HEADS = dict(weight_seed=3, std=0.02, matrix_gain=1.5, bias_std=0.02, norm_jitter=0.1, tile_seed=78, tile_h=224, tile_w=308,
             text_seed=9, n_text_a=5, n_text_b=14, n1=20, repetition_penalty=1.3)


def make_heads_chain():
    """tests/golden/heads_chain.npz: transformers on oracle.heads_config() -- 16 q / 2 kv heads x 128, 16 ViT heads x 80 (VERDICT r4
    missing #4: the tiny fixture has 4 q / 2 kv and 2 ViT heads, so GQA at group 8 was only ever oracle-vs-engine).  One stage:
    ViT rows (sampled), the prompt's per-step fp32 logits along the fp32 greedy path, the reference's own bf16 logits on the same
    path (the yardstick) and its free-running bf16 tokens.  Weights come from the repo PRNG; only outputs are stored."""
    import torch
    from PIL import Image

    from hf_bridge import hf_model

    infer = load_ref("src/eval/infer.py", "ref_infer")
    cfg = qwen25vl.heads_config()
    c = HEADS
    w = qwen25vl.synthetic_weights(cfg, seed=c["weight_seed"], std=c["std"], matrix_gain=c["matrix_gain"], bias_std=c["bias_std"],
                                   norm_jitter=c["norm_jitter"])
    tile = prng.synthetic_tile(c["tile_seed"], c["tile_h"], c["tile_w"])
    view, scale = infer.resize_image(Image.fromarray(tile))
    view = np.array(view)
    pv, g = frontend.image_to_pixel_values(view)
    a = prng.uniform_ints(c["text_seed"], c["n_text_a"], 10, 2000).tolist()
    b = prng.uniform_ints(c["text_seed"] + 1, c["n_text_b"], 10, 2000).tolist()
    ids = a + [cfg.vision_start_token_id] + [cfg.image_token_id] * (g[1] * g[2] // 4) + [cfg.vision_end_token_id] + b
    out = dict(view_sha256=np.frombuffer(sha(view).encode(), dtype=np.uint8), grid=np.array(g), ids=np.array(ids))
    m32, m16 = hf_model(cfg, w, torch.float32), hf_model(cfg, w, torch.bfloat16)

    def hf_inputs(seq):
        t = torch.tensor([seq])
        return dict(input_ids=t, attention_mask=torch.ones_like(t), pixel_values=torch.from_numpy(pv), image_grid_thw=torch.tensor([list(g)]),
                    mm_token_type_ids=(t == cfg.image_token_id).int())

    with torch.no_grad():
        gen = m32.generate(**hf_inputs(ids), max_new_tokens=c["n1"], do_sample=False, num_beams=1, repetition_penalty=c["repetition_penalty"],
                           output_logits=True, return_dict_in_generate=True)
        toks = gen.sequences[0, len(ids):].tolist()
        logits32 = torch.stack([x[0] for x in gen.logits]).float().numpy()
        full = ids + toks[:-1]
        lg16 = m16(**hf_inputs(full)).logits[0, len(ids) - 1:].float().numpy()
        lg32f = m32(**hf_inputs(full)).logits[0, len(ids) - 1:].float().numpy()
        vit32 = m32.model.visual(torch.from_numpy(pv), grid_thw=torch.tensor([list(g)])).pooler_output.numpy()
        vit16 = m16.model.visual(torch.from_numpy(pv).bfloat16(), grid_thw=torch.tensor([list(g)])).pooler_output.float().numpy()
        g16 = m16.generate(**hf_inputs(ids), max_new_tokens=c["n1"], do_sample=False, num_beams=1, repetition_penalty=c["repetition_penalty"])
    step = max(1, vit32.shape[0] // 16)
    out.update(tokens_fp32=np.array(toks), tokens_bf16_free=g16[0, len(ids):].numpy(), logits_fp32=logits32.astype(np.float32),
               logits_bf16=lg16.astype(np.float32), vit_fp32=vit32[::step][:20].astype(np.float32), vit_bf16=vit16[::step][:20].astype(np.float32),
               vit_fp32_sha256=np.frombuffer(sha(vit32.astype(np.float32)).encode(), dtype=np.uint8),
               chain_json=np.frombuffer(json.dumps(c).encode(), dtype=np.uint8))
    print("heads: grid", g, "prompt", len(ids), "distinct tokens", len(set(toks)), "of", len(toks), "max|fp32full-fp32|",
          float(np.abs(lg32f - logits32).max()), "max|bf16-fp32|", float(np.abs(lg16 - logits32).max()))
    o32 = qwen25vl.Qwen25VLOracle(cfg, w, "fp32")
    r = qwen25vl.greedy_generate(o32, ids, pv, [g], c["n1"], c["repetition_penalty"], eos_token_ids=())
    print("oracle fp32 tokens equal:", r["tokens"] == toks, "max logit err", float(np.abs(r["logits"] - logits32).max()))
    np.savez_compressed(os.path.join(HERE, "heads_chain.npz"), **out)
    print("heads_chain.npz", os.path.getsize(os.path.join(HERE, "heads_chain.npz")))


# ----------------------------------------------------------------------------- 6. rollout scoring
def make_score():
    """Per-token log-probs of the tiny chain's stage-2 sequence, computed with the HF model the way
    `_get_per_token_logps` does (src/train/RL/src/open-r1-multimodal/src/open_r1/trainer/grpo_trainer.py:494-504)."""
    import torch
    from PIL import Image

    from hf_bridge import hf_model

    infer = load_ref("src/eval/infer.py", "ref_infer")
    cfg = qwen25vl.tiny_config()
    w = qwen25vl.synthetic_weights(cfg, seed=CHAIN["weight_seed"], std=CHAIN["std"], matrix_gain=CHAIN["matrix_gain"],
                                   bias_std=CHAIN["bias_std"], norm_jitter=CHAIN["norm_jitter"])
    chain = np.load(os.path.join(HERE, "tiny_chain.npz"))
    tile = prng.synthetic_tile(CHAIN["tile_seed"], CHAIN["tile_h"], CHAIN["tile_w"])
    pil_tile = Image.fromarray(tile)
    view, _ = infer.resize_image(pil_tile)
    crop, _ = infer.resize_image(infer.cut_image(pil_tile, CHAIN["bbox"]))
    pv_v, g_v = frontend.image_to_pixel_values(np.array(view))
    pv_c, g_c = frontend.image_to_pixel_values(np.array(crop))
    ids = chain["ids2"].tolist() + chain["s2_tokens_fp32"].tolist()
    t = torch.tensor([ids])
    out = dict(ids=np.array(ids), prompt_len=np.array(len(chain["ids2"])))
    for name, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        m = hf_model(cfg, w, dt)
        with torch.no_grad():
            logits = m(input_ids=t, attention_mask=torch.ones_like(t),
                       pixel_values=torch.from_numpy(np.concatenate([pv_v, pv_c])),
                       image_grid_thw=torch.tensor([list(g_v), list(g_c)]),
                       mm_token_type_ids=(t == cfg.image_token_id).int()).logits
        lp = torch.gather(logits[0, :-1].log_softmax(dim=-1), 1, t[0, 1:].unsqueeze(1)).squeeze(1)
        out[f"logps_{name}"] = lp.float().numpy()
        if dt == torch.bfloat16:  # the same bf16 logits with the log-softmax in fp32 (what ze_score returns)
            lp32 = torch.gather(logits[0, :-1].float().log_softmax(dim=-1), 1, t[0, 1:].unsqueeze(1)).squeeze(1)
            out["logps_bf16_logits_fp32_softmax"] = lp32.numpy()
    o32 = qwen25vl.Qwen25VLOracle(cfg, w, "fp32")
    mine = o32.per_token_logps(ids, np.concatenate([pv_v, pv_c]), [g_v, g_c])
    print("oracle fp32 vs HF fp32 max|d|", float(np.abs(mine - out["logps_fp32"]).max()),
          "HF bf16 vs fp32 max|d|", float(np.abs(out["logps_bf16"] - out["logps_fp32"]).max()))
    np.savez_compressed(os.path.join(HERE, "score.npz"), **out)
    print("score.npz", os.path.getsize(os.path.join(HERE, "score.npz")))


if __name__ == "__main__":
    which = sys.argv[1:] or ["host", "bicubic", "indices", "pixels", "chain"]
    if "host" in which:
        make_host_helpers()
    if "bicubic" in which:
        make_bicubic()
    if "indices" in which:
        make_indices()
    if "pixels" in which:
        make_pixel_values()
    if "chain" in which:
        make_tiny_chain()
    if "score" in which:
        make_score()
    if "heads" in which:
        make_heads_chain()
