"""GPU: the Python call surface of the reference (processor(...), model.generate(...), chat_batch, zoom chain,
from_pretrained on a safetensors checkpoint in both key layouts) on the tiny config, checked against direct
engine calls / the oracle."""
import json
import os

import numpy as np
import pytest
import torch

from gpu_util import CHAIN_W
from oracle import frontend, prng
from oracle import qwen25vl as Q
from tiny_tok import make_tokenizer
from zoomearth_amd import checkpoint
from zoomearth_amd import hostloop as H
from zoomearth_amd.config import ModelConfig
from zoomearth_amd.image import DeviceImage
from zoomearth_amd.modeling import ZoomEarthForConditionalGeneration
from zoomearth_amd.processor import ZoomEarthProcessor

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def stack():
    model = ZoomEarthForConditionalGeneration.from_synthetic(ModelConfig.tiny(), **CHAIN_W, max_seqs=4, max_ctx=2048,
                                                            max_patches=4096, max_tile_side=2048)
    proc = ZoomEarthProcessor(make_tokenizer(), min_pixels=3136, max_pixels=128 * 128 * 28 * 28)
    proc.tokenizer.padding_side = "left"
    tile_np = prng.synthetic_tile(77, 900, 1100)
    tile = DeviceImage.from_numpy(tile_np, model.engine)
    yield model, proc, tile, tile_np
    model.engine.close()


def words(seed, n):
    return " ".join(f"w{int(v)}" for v in prng.uniform_ints(seed, n, 10, 1990))


def prompt1(q):
    return "<|im_start|> " + words(1, 4) + " <|vision_start|><|image_pad|><|vision_end|> " + q + " <|im_start|>"


def test_device_image_matches_pil_semantics(stack):
    model, proc, tile, tile_np = stack
    assert tile.size == (1100, 900) and tile.width == 1100 and tile.height == 900
    view, scale = H.resize_image(tile)
    assert view.size == (512, 418) and scale == 1100 / 512
    assert np.array_equal(view.numpy(), frontend.resize_bicubic(tile_np, 512, 418))
    crop = H.cut_image(tile, [300.0, 200.0, 420.0, 330.0])
    assert crop.size == (512, 512)
    assert np.array_equal(crop.numpy(), frontend.crop_zero_fill(tile_np, H.zoom_box((1100, 900), [300, 200, 420, 330])))
    big = H.cut_image(tile, [-100, -50, 700, 800])  # "large bbox" branch leaves the image: zero fill
    assert np.array_equal(big.numpy(), frontend.crop_zero_fill(tile_np, (-100, -50, 700, 800)))
    small, _ = H.resize_image(big)  # fused crop + resize in ONE launch on the full-res tile
    assert np.array_equal(small.numpy(), frontend.resize_bicubic(frontend.crop_zero_fill(tile_np, (-100, -50, 700, 800)),
                                                                  small.size[0], small.size[1]))
    sub = big.crop((10, 20, 110, 220))  # crop of a crop composes in tile coordinates
    assert np.array_equal(sub.numpy(), frontend.crop_zero_fill(tile_np, (-90, -30, 10, 170)))


def test_processor_outputs(stack):
    model, proc, tile, tile_np = stack
    view, _ = H.resize_image(tile)
    out = proc(text=[prompt1(words(2, 5))], images=[view], return_tensors="pt", padding="longest").to(model.device)
    pv_ref, grid = frontend.image_to_pixel_values(frontend.resize_bicubic(tile_np, 512, 418))
    assert out["image_grid_thw"].tolist() == [list(grid)]
    assert np.array_equal(out["pixel_values"].cpu().numpy(), pv_ref)
    ids = out["input_ids"][0].tolist()
    n_img = grid[1] * grid[2] // 4
    assert ids.count(2005) == n_img and out["mm_token_type_ids"].sum().item() == n_img
    assert ids[ids.index(2005) - 1] == 2002 and ids[ids.index(2005) + n_img] == 2003
    assert out["attention_mask"].shape == out["input_ids"].shape
    with pytest.raises(ValueError):
        proc(text=["<|image_pad|> <|image_pad|>"], images=[view])


def test_generate_matches_engine_and_batches(stack):
    model, proc, tile, tile_np = stack
    e = model.engine
    view, _ = H.resize_image(tile)
    p_a, p_b = prompt1(words(3, 6)), prompt1(words(4, 17))
    # single rows
    outs = {}
    for name, p in (("a", p_a), ("b", p_b)):
        inp = proc(text=[p], images=[view], return_tensors="pt", padding="longest").to(model.device)
        g = model.generate(**inp, max_new_tokens=12, do_sample=False, num_beams=1, ignore_eos=True)
        assert g.shape == (1, inp["input_ids"].shape[1] + 12)
        assert torch.equal(g[:, : inp["input_ids"].shape[1]].cpu(), inp["input_ids"].cpu())
        outs[name] = g[0, inp["input_ids"].shape[1]:].tolist()
        # direct engine path
        ids = inp["input_ids"][0].tolist()
        emb = e.vit_forward(inp["pixel_values"], inp["image_grid_thw"].tolist())
        pos, delta = e.rope_index(ids, inp["image_grid_thw"].tolist())
        e.seq_reset(3)
        e.prefill(3, ids, emb, pos, delta, want_logits=False)
        assert e.generate(3, 12, ignore_eos=True) == outs[name]
    # left-padded batch of both == the single-row results; sampling flags are accepted (greedy)
    inp = proc(text=[p_a, p_b], images=[view, view], return_tensors="pt", padding="longest").to(model.device)
    assert (inp["attention_mask"][0] == 0).sum() == 11
    g = model.generate(**inp, max_new_tokens=12, do_sample=True, temperature=0.01, num_beams=1, ignore_eos=True)
    L = inp["input_ids"].shape[1]
    assert g[0, L:].tolist() == outs["a"] and g[1, L:].tolist() == outs["b"]
    with pytest.raises(NotImplementedError):
        model.generate(**inp, max_new_tokens=2, num_beams=4)


def test_two_stage_chain_with_reuse_is_identical(stack):
    model, proc, tile, tile_np = stack
    view, scale = H.resize_image(tile)
    crop, _ = H.resize_image(H.cut_image(tile, [300.0, 200.0, 420.0, 330.0]))
    p1 = prompt1(words(5, 9))
    results = []
    for reuse in (True, False):
        model.reuse_prefix = reuse
        model._chains.clear()
        model._vit_cache.clear()
        o1 = H.chat_batch([p1], [view], proc, model, max_new_tokens=10)
        p2 = p1 + " " + o1[0] + " <|vision_start|><|image_pad|><|vision_end|>"
        o2 = H.chat_batch([p2], [[view, crop]], proc, model, max_new_tokens=10)
        results.append((o1, o2))
    model.reuse_prefix = True
    assert results[0] == results[1]
    assert len(results[0][0][0].split()) >= 5


def test_zoom_chain_scripted(stack):
    """Control flow of the chain with a scripted stage-1 text (random weights never emit a bbox)."""
    model, proc, tile, tile_np = stack
    seen = []

    def chat(prompts, images):
        seen.append(images)
        out = H.chat_batch(prompts, images, proc, model, max_new_tokens=6)
        if len(seen) == 1:
            return ['w5 [{"bbox_2d": [140, 93, 195, 153], "label": "w9"}] <answer>w1</answer>']
        return out

    r = H.zoom_chain("w11 w12 w13", tile, chat)
    assert not r["error"] and r["bbox"] == [v * (1100 / 512) for v in (140, 93, 195, 153)]
    assert seen[1][0][1].size == (512, 512) and isinstance(r["output2"], str)


def test_from_pretrained_both_key_layouts(stack, tmp_path):
    model, proc, tile, tile_np = stack
    cfg = Q.tiny_config()
    w = Q.synthetic_weights(cfg, **CHAIN_W)
    hf_cfg = {"vision_config": dict(depth=4, hidden_size=160, num_heads=2, intermediate_size=220, out_hidden_size=512,
                                    fullatt_block_indexes=[1, 3]),
              "hidden_size": 512, "num_hidden_layers": 3, "num_attention_heads": 4, "num_key_value_heads": 2,
              "intermediate_size": 1376, "vocab_size": 2048, "rms_norm_eps": 1e-6, "rope_theta": 1000000.0,
              "rope_scaling": {"type": "mrope", "mrope_section": [16, 24, 24]}, "tie_word_embeddings": True,
              "image_token_id": 2005, "vision_start_token_id": 2002, "vision_end_token_id": 2003,
              "eos_token_id": [2045, 2043], "pad_token_id": 2043}
    view, _ = H.resize_image(tile)
    inp = proc(text=[prompt1(words(6, 5))], images=[view], return_tensors="pt", padding="longest")
    ref = model.generate(**inp, max_new_tokens=8, ignore_eos=True)

    def rename_449(k):  # 4.49-era layout
        if k.startswith("model.visual."):
            return k[len("model."):]
        if k.startswith("model.language_model."):
            return "model." + k[len("model.language_model."):]
        return k

    for tag, rename, bf16 in (("v5_f32", lambda k: k, False), ("v449_bf16", rename_449, True)):
        d = tmp_path / tag
        os.makedirs(d)
        with open(d / "config.json", "w") as f:
            json.dump(hf_cfg, f)
        with open(d / "generation_config.json", "w") as f:
            json.dump({"eos_token_id": [2045, 2043], "pad_token_id": 2043, "repetition_penalty": 1.0}, f)
        tensors = {rename(k): v for k, v in w.items()}
        if tag == "v449_bf16":
            tensors["lm_head.weight"] = w["model.language_model.embed_tokens.weight"]  # tied copy present in the file
        checkpoint.write_safetensors(str(d / "model.safetensors"), tensors, bf16=bf16)
        m2 = ZoomEarthForConditionalGeneration.from_pretrained(str(d), max_seqs=1, max_ctx=1024, max_patches=2048,
                                                               max_tile_side=1024)
        try:
            inp2 = proc(text=[prompt1(words(6, 5))], images=[DeviceImage.from_numpy(view.numpy(), m2.engine)],
                        return_tensors="pt", padding="longest")
            got = m2.generate(**inp2, max_new_tokens=8, ignore_eos=True)
            assert torch.equal(got.cpu(), ref.cpu()), tag
        finally:
            m2.engine.close()
    from zoomearth_amd import processor as P
    P.set_default_engine(model.engine)
    # a checkpoint with a missing tensor fails loudly
    d = tmp_path / "broken"
    os.makedirs(d)
    with open(d / "config.json", "w") as f:
        json.dump(hf_cfg, f)
    checkpoint.write_safetensors(str(d / "model.safetensors"), {k: v for k, v in w.items() if "layers.1.mlp.up_proj" not in k})
    with pytest.raises(RuntimeError, match="missing weights"):
        ZoomEarthForConditionalGeneration.from_pretrained(str(d), max_seqs=1, max_ctx=256, max_patches=1024, max_tile_side=1024)
    P.set_default_engine(model.engine)


def test_openai_compatible_server_matches_generate(stack):
    """POST /v1/chat/completions (the request src/eval/infer_vllm.py sends) == processor + generate + decode on the
    chat-templated prompt; greedy and seeded sampling; error mapping."""
    import base64
    import io
    from fastapi.testclient import TestClient
    from PIL import Image
    from zoomearth_amd import serve
    model, proc, tile, tile_np = stack
    small = frontend.resize_bicubic(tile_np, 256, 209)
    buf = io.BytesIO()
    Image.fromarray(small).save(buf, format="PNG")  # lossless, so the server sees exactly `small`
    url = "data:image/png;base64," + base64.b64encode(buf.getvalue()).decode()
    text = words(21, 7)
    msgs = [{"role": "user", "content": [{"type": "image_url", "image_url": {"url": url}}, {"type": "text", "text": " " + text + " "}]}]
    client = TestClient(serve.create_app(serve.ChatServer(model, proc, "ZoomEarth")))
    assert client.get("/health").json() == {"status": "ok"}
    assert client.get("/v1/models").json()["data"][0]["id"] == "ZoomEarth"
    r = client.post("/v1/chat/completions", json={"model": "ZoomEarth", "messages": msgs, "max_tokens": 10})
    assert r.status_code == 200, r.text
    body = r.json()
    # the same request by hand
    prompt, _ = serve.build_prompt(msgs)
    img = DeviceImage.from_numpy(small, model.engine)
    inp = proc(text=[prompt], images=[img], return_tensors="pt", padding="longest").to(model.device)
    g = model.generate(**inp, max_new_tokens=10, do_sample=False, num_beams=1)[0, inp["input_ids"].shape[1]:].tolist()
    eos = set(model.config.eos_token_ids)
    cut = next((i + 1 for i, t in enumerate(g) if t in eos), len(g))
    want = proc.tokenizer.decode(g[:cut], skip_special_tokens=True).strip()
    assert body["choices"][0]["message"] == {"role": "assistant", "content": want}
    assert body["usage"]["prompt_tokens"] == inp["input_ids"].shape[1] and body["usage"]["completion_tokens"] == cut
    assert body["choices"][0]["finish_reason"] == ("stop" if cut < len(g) or g[-1] in eos else "length")
    # seeded sampling is reproducible and seed-dependent
    a = client.post("/v1/chat/completions", json={"messages": msgs, "max_tokens": 12, "temperature": 1.0, "seed": 4}).json()
    b = client.post("/v1/chat/completions", json={"messages": msgs, "max_tokens": 12, "temperature": 1.0, "seed": 4}).json()
    c = client.post("/v1/chat/completions", json={"messages": msgs, "max_tokens": 12, "temperature": 1.0, "seed": 5}).json()
    assert a["choices"][0]["message"] == b["choices"][0]["message"] != c["choices"][0]["message"]
    # errors
    assert client.post("/v1/chat/completions", json={"messages": msgs, "stream": True}).status_code == 400
    assert client.post("/v1/chat/completions", json={"messages": [{"role": "user", "content": [{"type": "image_url", "image_url": {"url": "http://example.com/x.png"}}]}]}).status_code == 400


def test_openai_server_batches_concurrent_requests(stack):
    """Concurrent /v1/chat/completions requests (the reference client keeps many in flight, src/eval/infer_vllm.py:
    244-271) are admitted by the dispatcher into one running batch (zoomearth_amd/scheduler.py).  A request's text must
    equal the same row of model.generate on the batch built by hand (the batched kernels give a chain the same tokens
    whatever shares its steps), and the scheduler's counters must show that chains really shared decode steps."""
    import base64
    import io
    import threading
    from fastapi.testclient import TestClient
    from PIL import Image
    from zoomearth_amd import serve
    model, proc, tile, tile_np = stack
    reqs, prompts, imgs = [], [], []
    for i, (w, h) in enumerate(((256, 209), (224, 224), (308, 252))):
        small = frontend.resize_bicubic(tile_np, w, h)
        buf = io.BytesIO()
        Image.fromarray(small).save(buf, format="PNG")
        url = "data:image/png;base64," + base64.b64encode(buf.getvalue()).decode()
        msgs = [{"role": "user", "content": [{"type": "image_url", "image_url": {"url": url}},
                                             {"type": "text", "text": " " + words(40 + i, 5 + i) + " "}]}]
        reqs.append({"model": "ZoomEarth", "messages": msgs, "max_tokens": 8 + i})
        prompts.append(serve.build_prompt(msgs)[0])
        imgs.append(DeviceImage.from_numpy(small, model.engine))

    def decode(row, budget):
        row = row[:budget]
        eos = set(model.config.eos_token_ids)
        cut = next((j + 1 for j, t in enumerate(row) if t in eos), len(row))
        ids = row[:cut]
        while cut == len(row) and ids and ids[-1] == model.config.pad_token_id and ids[-1] not in eos:
            ids = ids[:-1]
        return proc.tokenizer.decode(ids, skip_special_tokens=True).strip()

    inp = proc(text=prompts, images=imgs, return_tensors="pt", padding="longest").to(model.device)
    g = model.generate(**inp, max_new_tokens=10, do_sample=False, num_beams=1)[:, inp["input_ids"].shape[1]:].tolist()
    want_batched = [decode(g[i], 8 + i) for i in range(3)]
    want_single = []
    for i in range(3):
        one = proc(text=[prompts[i]], images=[imgs[i]], return_tensors="pt", padding="longest").to(model.device)
        gi = model.generate(**one, max_new_tokens=8 + i, do_sample=False, num_beams=1)[0, one["input_ids"].shape[1]:].tolist()
        want_single.append(decode(gi, 8 + i))

    srv = serve.ChatServer(model, proc, "ZoomEarth", batch_window_s=1.0)
    # the explicit batch entry point first: exactly the hand-built batch
    many = srv.complete_many(reqs)
    assert [m["choices"][0]["message"]["content"] for m in many] == want_batched
    client = TestClient(serve.create_app(srv))
    out = [None] * 3
    gate = threading.Barrier(3)

    def post(i):
        gate.wait()
        out[i] = client.post("/v1/chat/completions", json=reqs[i])

    ts = [threading.Thread(target=post, args=(i,)) for i in range(3)]
    [t.start() for t in ts]
    [t.join(timeout=120) for t in ts]
    assert all(o is not None and o.status_code == 200 for o in out), [o and o.text for o in out]
    st = srv.scheduler.stats  # the dispatcher's continuous-batching scheduler: chains really shared decode steps
    assert st["admitted"] == 3 and st["chain_steps"] > st["steps"], st
    for i, o in enumerate(out):
        got = o.json()["choices"][0]["message"]["content"]
        assert got in (want_batched[i], want_single[i]), (i, got, want_batched[i], want_single[i])
    # sampled requests that share temperature and seed run as ONE batch too, and a request draws
    # what it draws alone: the same text as when it is the dispatcher's only request (a batch of one chain)
    sreqs = [dict(r, temperature=0.9, seed=11) for r in reqs]
    alone = [srv.submit(r).result(timeout=120)["choices"][0]["message"]["content"] for r in sreqs]
    futs = [srv.submit(r) for r in sreqs] + [srv.submit(dict(reqs[0], temperature=0.9, seed=12))]
    got = [f.result(timeout=120)["choices"][0]["message"]["content"] for f in futs]
    assert got[:3] == alone and got[3] != alone[0]
    srv.close()


def test_weights_broadcast_through_a_caller_owned_rccl_communicator(stack):
    """ze_weights_broadcast: the entry point for hosts that own an ncclComm_t.  A one-rank communicator created through
    RCCL's C API (ctypes) -- the broadcast is then a device-side no-op copy, which still exercises symbol lookup, argument
    passing and the invalidation of derived weight copies; the arena must be unchanged and generation must still work."""
    import ctypes as C
    import glob
    model, proc, tile, tile_np = stack
    e = model.engine
    libs = glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so*")) + ["/opt/rocm/lib/librccl.so"]
    rccl = C.CDLL(libs[0], mode=C.RTLD_GLOBAL)

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]

    uid = UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        before = e.weights_arena().to(torch.int64).sum().item()
        rc = e.lib.ze_weights_broadcast(e.h, comm, 0, C.c_void_p(torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        assert rc == 0, e.lib.ze_last_error(e.h)
        assert e.weights_arena().to(torch.int64).sum().item() == before
        assert e.lib.ze_weights_broadcast(e.h, None, 0, None) < 0          # null communicator: an error, not a crash
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)
    view, _ = H.resize_image(tile)
    out = H.chat_batch([prompt1(words(61, 5))], [view], proc, model, max_new_tokens=4)
    assert isinstance(out[0], str)
