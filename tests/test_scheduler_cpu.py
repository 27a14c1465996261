"""CPU: host logic of continuous batching (zoomearth_amd/scheduler.py) and of the OpenAI shim's dispatcher on top of
it, against a stub engine that speaks the Engine methods the scheduler calls.  The stub's "model": token i of a chain
= 100 + (first prompt id + i) % 7; chains whose first prompt id is even emit EOS (id 3) as their third token.

Covers (the control flow /root/reference/src/eval/infer.py:173-249 runs one sample at a time, and the in-flight request
stream of src/eval/infer_vllm.py:244-271): more requests than KV slots, chains leaving at EOS / at their own token budget
while others go on, slots handed to waiting requests between bursts, the two-stage follow-up on the SAME slot with the
cached stage-1 prompt kept (seq_truncate to its length, only the appended tokens prefilled), one ViT call for all new
images of a round with per-tile feature reuse, error isolation per request, results independent of the slot count.
"""
import threading
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from zoomearth_amd import hostloop as H
from zoomearth_amd.scheduler import ChainScheduler, Request

EOS, PAD, IMG = 3, 0, 7


class StubEngine:
    def __init__(self, max_seqs=4, max_ctx=256, max_prefill_rows=256, max_patches=64):
        self.max_seqs, self.max_ctx, self.max_prefill_rows, self.max_patches = max_seqs, max_ctx, max_prefill_rows, max_patches
        self.chains = {}
        self.log = []
        self.fail_burst = False

    def gen_params(self, **kw):
        return kw

    def rope_index(self, ids, grids):
        return np.zeros((3, len(ids)), np.int32), 0

    def vit_forward(self, pv, grids):
        self.log.append(("vit", [tuple(g) for g in grids]))
        return torch.zeros((sum(g[0] * g[1] * g[2] for g in grids) // 4, 8))

    def seq_reset(self, slot):
        self.chains[slot] = dict(ids=[], out=[], fin=False)

    def seq_len(self, slot):
        c = self.chains.get(slot)
        return 0 if c is None else len(c["ids"]) + max(len(c["out"]) - 1, 0)

    def seq_copy_prefix(self, dst, src, n):
        assert n <= len(self.chains[src]["ids"]) and dst != src
        self.log.append(("copy", dst, src, n))
        self.chains[dst] = dict(ids=list(self.chains[src]["ids"][:n]), out=[], fin=False)

    def seq_truncate(self, slot, keep):
        c = self.chains[slot]
        assert keep <= len(c["ids"]) + max(len(c["out"]) - 1, 0)
        self.log.append(("truncate", slot, keep))
        c["ids"], c["out"], c["fin"] = (c["ids"] + c["out"][:-1])[:keep], [], False   # rows of fed tokens stay

    def prefill_batch(self, slots, ids_l, emb_l, pos_l, dl):
        self.log.append(("prefill", list(slots), [len(x) for x in ids_l]))
        for s, ids, emb in zip(slots, ids_l, emb_l):
            assert ids.count(IMG) == (0 if emb is None else emb.shape[0])
            self.chains[s]["ids"] += list(ids)

    def mark_seen(self, slot, ids):
        pass

    def _next(self, c):
        i = len(c["out"])
        first = c["ids"][0]
        tok = PAD if c["fin"] else (EOS if (first % 2 == 0 and i == 2) else 100 + (first + i) % 7)
        c["out"].append(tok)
        c["fin"] = c["fin"] or tok == EOS

    def chain_begin(self, slot, params, stream):
        self._next(self.chains[slot])

    def decode_burst(self, slots, steps, params):
        if self.fail_burst:
            raise RuntimeError("engine exploded")
        assert len(set(slots)) == len(slots)
        for s in slots:
            steps = min(steps, self.max_ctx - (len(self.chains[s]["ids"]) + len(self.chains[s]["out"]) - 1))
        self.log.append(("burst", len(slots), steps))
        for _ in range(steps):
            for s in slots:
                self._next(self.chains[s])
        return steps, [len(self.chains[s]["out"]) for s in slots], [self.chains[s]["fin"] for s in slots]

    def chain_tokens(self, slot, cap=0):
        out = self.chains[slot]["out"][: cap or None]
        return out[: out.index(EOS) + 1] if EOS in out else out


class Tok:
    def decode(self, ids, skip_special_tokens=True):
        return " ".join(str(i) for i in ids if not (skip_special_tokens and i in (PAD, EOS)))


class Proc:
    """text -> ids: one id per whitespace word (`<img>` expands to 4 image tokens per image)."""
    tokenizer = Tok()

    def __call__(self, text, images=None, return_tensors="pt", **kw):
        out, n = [], 0
        for w in text[0].split():
            if w == "<img>":
                out += [IMG] * 4
                n += 1
            else:
                out.append(int(w))
        d = dict(input_ids=torch.tensor([out]))
        if images:
            assert n == len(images)
            d.update(image_grid_thw=torch.tensor([[1, 4, 4]] * n), pixel_values=torch.zeros((16 * n, 3)),
                     image_keys=[("k", im) for im in images])
        return d


def make_model(**kw):
    e = StubEngine(**kw)
    cfg = SimpleNamespace(image_token_id=IMG, eos_token_ids=(EOS,), pad_token_id=PAD,
                          vision=SimpleNamespace(spatial_merge_size=2))
    return SimpleNamespace(engine=e, config=cfg, generation_config=SimpleNamespace(repetition_penalty=1.0, temperature=None),
                           _chains={}, device="cpu")


def expected(first, budget):
    out = []
    for i in range(budget):
        t = EOS if (first % 2 == 0 and i == 2) else 100 + (first + i) % 7
        out.append(t)
        if t == EOS:
            break
    return out


@pytest.mark.parametrize("slots", [1, 2, 4])
def test_continuous_batching_more_requests_than_slots(slots):
    model = make_model(max_seqs=slots)
    sched = ChainScheduler(model, Proc(), burst=3)
    got, order = {}, []
    reqs = []
    for q in range(9):
        first = 11 + q
        reqs.append(Request(prompt=f"{first} 50 51", images=[], max_new_tokens=2 + q,
                            on_done=lambda r, toks, text, q=q: (got.__setitem__(q, (toks, text)), order.append(q))[0]))
        sched.submit(reqs[-1])
    sched.run()
    assert sorted(got) == list(range(9))
    for q in range(9):
        want = expected(11 + q, 2 + q)
        assert got[q][0] == want and got[q][1] == " ".join(str(t) for t in want if t != EOS)
        assert reqs[q].n_prompt == 3
    bursts = [x for x in model.engine.log if x[0] == "burst"]
    assert max(b[1] for b in bursts) == min(slots, 9)          # the slots were really shared
    assert sched.stats["admitted"] == 9 and not sched.live and sorted(sched.free) == list(range(slots))
    if slots == 4:  # even-first chains stop at their EOS and their slot is re-used while longer chains go on
        assert order.index(1) < order.index(0) or order.index(3) < order.index(2)


@pytest.mark.parametrize("mode", ["generated", "prompt", "edited"])
def test_two_stage_follow_up_reuses_slot_and_prefix_and_view_features(mode):
    """mode "generated": the follow-up keeps the rows of the prompt AND of the generated tokens it repeats (all but the last
    one sampled, which never went through the model); "prompt": reuse_generated=False keeps the prompt rows only;
    "edited": the re-inserted output differs from the generated ids at its second word -- reuse stops there."""
    model = make_model(max_seqs=2)
    e = model.engine
    sched = ChainScheduler(model, Proc(), burst=2, reuse_generated=mode != "prompt")
    results = {}

    def chain(q, first, view, crop):
        p1 = f"{first} <img> 60"

        def stage1(req, toks, text):
            if mode == "edited":
                words = text.split()
                words[1] = "99"
                text = " ".join(words)
            p2 = p1 + " " + text + " <img>"

            def stage2(req2, toks2, text2):
                results[q] = (req.slot, req2.slot, text, text2, req2.n_prompt)
            return Request(prompt=p2, images=[view, crop], max_new_tokens=3, on_done=stage2)
        sched.submit(Request(prompt=p1, images=[view], max_new_tokens=4, on_done=stage1))

    chain(0, 21, "viewA", "crop0")
    chain(1, 23, "viewA", "crop1")   # same tile: the view is encoded once
    chain(2, 25, "viewB", "crop2")
    sched.run()
    assert sorted(results) == [0, 1, 2]
    for q, first in ((0, 21), (1, 23), (2, 25)):
        s1, s2, t1, t2, n2 = results[q]
        assert s1 == s2                                      # stage 2 continued on the slot of stage 1
        if mode != "edited":
            assert t1 == " ".join(str(t) for t in expected(first, 4))
        assert n2 == 6 + 4 + 4                               # stage-1 prompt + its 4 output words + second image
    kept = {"generated": 6 + 3, "prompt": 6, "edited": 6 + 1}[mode]   # cached stage-1 prompt = 1 + 4 + 1 tokens
    trunc = [x for x in e.log if x[0] == "truncate"]
    assert len(trunc) == 3 and all(t[2] == kept for t in trunc)
    pre = [x for x in e.log if x[0] == "prefill"]
    assert sorted(n for p in pre for n in p[2]) == sorted([6, 6, 6] + [14 - kept] * 3)   # stage 2 prefills only what is not cached
    assert sched.stats["reused_generated_rows"] == 3 * (kept - 6)
    vit = [g for x in e.log if x[0] == "vit" for g in x[1]]
    assert len(vit) == 5                                     # viewA once, viewB once, three crops


def test_shared_prompt_prefix_is_copied_not_recomputed():
    """Questions about one tile start alike (system turn + the view's image tokens): the first newcomer's prefix is
    prefilled alone, the others copy its K/V rows and prefill only their tails; a later question copies from a chain that
    is still alive; outputs equal those of a scheduler that prefills every prompt in full."""
    def run(share):
        model = make_model(max_seqs=4)
        sched = ChainScheduler(model, Proc(), burst=2, share_prefix=share, min_shared=3)
        out = {}
        for q in range(4):                                    # 71 72 73 <img x4> : shared;  then the question
            sched.submit(Request(prompt=f"71 72 73 <img> {20 + q} 50", images=["viewA"], max_new_tokens=40 if q == 0 else 12,
                                 on_done=lambda r, t, x, q=q: out.__setitem__(q, (t, r.n_prompt))))
        sched.step()                                          # all four admitted in one round
        sched.submit(Request(prompt="71 72 73 <img> 33 50 51", images=["viewA"], max_new_tokens=3,
                             on_done=lambda r, t, x: out.__setitem__(4, (t, r.n_prompt))))
        sched.submit(Request(prompt="71 99 <img> 34", images=["viewA"], max_new_tokens=3,   # shares only 1 token: full prefill
                             on_done=lambda r, t, x: out.__setitem__(5, (t, r.n_prompt))))
        sched.run()
        return out, model.engine.log, sched.stats

    shared, log, st = run(True)
    plain, log0, st0 = run(False)
    assert shared == plain and sorted(shared) == [0, 1, 2, 3, 4, 5]
    assert [x for x in log0 if x[0] == "copy"] == []
    copies = [x for x in log if x[0] == "copy"]
    assert len(copies) == 4 and all(c[3] == 7 for c in copies)            # 3 text ids + 4 image tokens
    assert {c[2] for c in copies} == {0} and {c[1] for c in copies[:3]} == {1, 2, 3}   # the later one copies from live chain 0
    pre = [x for x in log if x[0] == "prefill"]
    assert pre[0][2] == [7] and sorted(pre[1][2]) == [2, 2, 2, 2]         # pass A: the prefix once; pass B: four tails
    assert st["shared_rows"] == 28 and st["prefill_rows"] == st0["prefill_rows"] - 28
    assert sum(len(x[1]) for x in log if x[0] == "vit") == 1              # the view encoded once


def test_feature_cache_never_evicts_what_the_round_needs():
    """A round that encodes more new images than the cache holds must keep the already-cached view it also needs."""
    model = make_model(max_seqs=6, max_patches=16)
    sched = ChainScheduler(model, Proc(), burst=2, feature_cache=2)
    done = {}
    sched.submit(Request(prompt="41 <img> 60", images=["view"], max_new_tokens=2, on_done=lambda r, t, x: done.__setitem__("a", t)))
    sched.run()                                              # "view" is cached now
    for q in range(5):                                       # five prompts: the cached view + a new crop each
        sched.submit(Request(prompt=f"{43 + 2 * q} <img> 60 <img>", images=["view", f"crop{q}"], max_new_tokens=2,
                             on_done=lambda r, t, x, q=q: done.__setitem__(q, t)))
    sched.run()
    assert sorted(k for k in done if k != "a") == [0, 1, 2, 3, 4]
    assert sum(len(x[1]) for x in model.engine.log if x[0] == "vit") == 6   # the view once, five crops


def test_errors_are_isolated_per_request():
    model = make_model(max_seqs=2, max_ctx=12)
    sched = ChainScheduler(model, Proc(), burst=4)
    ok, bad = {}, {}
    sched.submit(Request(prompt="31 <img> <img>", images=["only-one"], on_done=lambda r, t, x: ok.__setitem__(0, t),
                         on_error=lambda r, ex: bad.__setitem__(0, str(ex))))
    sched.submit(Request(prompt=" ".join(["33"] * 20), images=[], on_done=lambda r, t, x: ok.__setitem__(1, t),
                         on_error=lambda r, ex: bad.__setitem__(1, str(ex))))
    sched.submit(Request(prompt="35 36 37", images=[], max_new_tokens=50, on_done=lambda r, t, x: ok.__setitem__(2, t),
                         on_error=lambda r, ex: bad.__setitem__(2, str(ex))))
    sched.run()
    assert set(bad) == {0, 1} and "max_ctx" in bad[1]
    assert ok[2] == expected(35, 12 - 3 + 1)                 # clamped to the KV capacity, like ze_generate
    assert sorted(sched.free) == [0, 1]


class OverlapEngine(StubEngine):
    """The stub with the two-half burst of the real engine (ze_decode_burst_begin / _end): what is enqueued by _begin only
    becomes visible at _end, and touching a chain of the burst in between is a bug the stub reports."""

    def __init__(self, **kw):
        super().__init__(**kw)
        self.in_flight = None

    def decode_burst_begin(self, slots, steps, params):
        assert self.in_flight is None
        for s in slots:
            steps = min(steps, self.max_ctx - (len(self.chains[s]["ids"]) + len(self.chains[s]["out"]) - 1))
        self.in_flight = (list(slots), steps)
        self.log.append(("begin", len(slots), steps))
        return steps

    def decode_burst_end(self, slots):
        live, steps = self.in_flight
        assert list(slots) == live
        self.in_flight = None
        for _ in range(steps):
            for s in live:
                self._next(self.chains[s])
        self.log.append(("end", len(live), steps))
        return [len(self.chains[s]["out"]) for s in live], [self.chains[s]["fin"] for s in live]

    def _guard(self, slot):
        assert self.in_flight is None or slot not in self.in_flight[0], f"slot {slot} touched while its burst is in flight"

    def seq_reset(self, slot):
        self._guard(slot)
        super().seq_reset(slot)

    def seq_truncate(self, slot, keep):
        self._guard(slot)
        super().seq_truncate(slot, keep)

    def prefill_batch(self, slots, *a):
        for s in slots:
            self._guard(s)
        if self.in_flight is not None:
            self.log.append(("overlapped",))
        super().prefill_batch(slots, *a)

    def chain_begin(self, slot, params, stream):
        assert self.in_flight is None
        super().chain_begin(slot, params, stream)


def test_overlapped_admission_gives_the_same_tokens_and_never_touches_a_chain_in_flight():
    """overlap=True: a burst is begun, ONE prefill pass of the admission round runs beside it, the burst is collected, the
    pass's chains join the next burst.  Same tokens as the sequential loop; follow-ups keep their slot and cached prompt."""
    def run(overlap):
        model = make_model(max_seqs=3, max_prefill_rows=12)
        if overlap:
            e = OverlapEngine(max_seqs=3, max_prefill_rows=12)
            model.engine = e
        sched = ChainScheduler(model, Proc(), burst=3, overlap=overlap)
        out = {}

        def first(i):
            def cb(req, toks, text):
                out[(i, 1)] = toks
                return Request(prompt=req.prompt + " " + text + " <img>", images=list(req.images) + [f"crop{i}"], max_new_tokens=4,
                               on_done=lambda r, t, x: out.__setitem__((i, 2), t))
            return cb

        for i in range(7):
            sched.submit(Request(prompt=f"{41 + 2 * i} <img> 60 61", images=[f"view{i % 2}"], max_new_tokens=5 + i % 3, on_done=first(i)))
        sched.run()
        return out, model.engine.log, sched.stats

    seq, _, st0 = run(False)
    ovl, log, st1 = run(True)
    assert seq == ovl and len(ovl) == 14
    assert st1["overlapped_passes"] > 0 and any(x[0] == "overlapped" for x in log)
    assert st1["admitted"] == st0["admitted"] == 14 and st1["chain_steps"] == st0["chain_steps"]


def test_a_failing_vit_call_or_prefill_pass_orphans_nobody():
    """a ViT error (or any failure after the round's requests left `waiting`) must fail every request
    of that round that is not live yet -- on_error called, slot back in `free` and reset -- instead of escaping step() with
    them in neither queue; an oversized image is rejected per request, before the ViT call; the scheduler keeps serving."""
    model = make_model(max_seqs=3, max_patches=16)
    e = model.engine
    sched = ChainScheduler(model, Proc(), burst=4)
    ok, bad = {}, {}

    def req(i, prompt, images):
        return Request(prompt=prompt, images=images, max_new_tokens=3, on_done=lambda r, t, x: ok.__setitem__(i, t),
                       on_error=lambda r, ex: bad.__setitem__(i, str(ex)))

    real_vit = e.vit_forward
    e.vit_forward = lambda pv, grids: (_ for _ in ()).throw(RuntimeError("too many patches"))
    sched.submit(req(0, "41 <img> 60", ["a"]))
    sched.submit(req(1, "43 44", []))
    sched.step()                                             # must not raise: both requests carry an on_error
    assert set(bad) == {0, 1} and "too many patches" in bad[0] and not sched.live and not sched.waiting
    assert sorted(sched.free) == [0, 1, 2]
    e.vit_forward = real_vit
    e.prefill_batch = lambda *a: (_ for _ in ()).throw(RuntimeError("prefill exploded"))
    sched.submit(req(2, "45 46", []))
    sched.step()
    assert "prefill exploded" in bad[2] and sorted(sched.free) == [0, 1, 2]
    del e.prefill_batch                                      # back to the class method
    # an image with more patches than the engine's ViT workspace: this request only
    class BigProc(Proc):
        def __call__(self, text, images=None, **kw):
            d = super().__call__(text, images, **kw)
            if images and images[0] == "huge":
                d["image_grid_thw"] = torch.tensor([[1, 8, 8]])
                d["pixel_values"] = torch.zeros((64, 3))
                d["input_ids"] = torch.tensor([[47] + [IMG] * 16 + [60]])
            return d
    sched.processor = BigProc()
    sched.submit(req(3, "47 <img> 60", ["huge"]))
    sched.submit(req(4, "49 50", []))
    sched.run()
    assert "max_patches" in bad[3] and ok[4] == expected(49, 3) and sorted(sched.free) == [0, 1, 2]
    # without an on_error the failure still surfaces as an exception, with every slot back
    e.vit_forward = lambda pv, grids: (_ for _ in ()).throw(RuntimeError("boom"))
    sched.submit(Request(prompt="51 <img> 60", images=["b"], max_new_tokens=2))
    with pytest.raises(RuntimeError):
        sched.step()
    assert sorted(sched.free) == [0, 1, 2] and not sched.waiting


def test_submit_zoom_chain_on_the_scheduler():
    """hostloop.submit_zoom_chain: the reference's per-question control flow (no box -> error record after stage 1,
    malformed box -> error record, box -> stage 2) as linked requests."""
    model = make_model(max_seqs=3)

    class P2(Proc):
        class tokenizer:  # stage-1 "text" carries a box for chains whose first token is 104 (first id % 7 == 4)
            @staticmethod
            def decode(ids, skip_special_tokens=True):
                if ids and ids[0] == 104:
                    return '{"bbox_2d": [10, 20, 30, 40]} <answer>a</answer>'
                if ids and ids[0] == 105:
                    return '{"bbox_2d": [10, 20, 30]}'
                return "nothing here"

        def __call__(self, text, images=None, **kw):
            n = text[0].count("<|image_pad|>")
            first = int(text[0].split("Q")[1].split()[0])
            d = dict(input_ids=torch.tensor([[first] + [IMG] * (4 * n) + [9]]))
            if images:
                d.update(image_grid_thw=torch.tensor([[1, 4, 4]] * n), pixel_values=torch.zeros((16 * n, 3)),
                         image_keys=[("k", id(im)) for im in images])
            return d

    class Img:
        width, height, size = 2000, 1500, (2000, 1500)

        def crop(self, box):
            self.box = box
            return self

        def resize(self, size, resample=None):
            return self

    sched = ChainScheduler(model, P2(), burst=4)
    res = {}
    for q, first in enumerate((11, 12, 13)):   # 11 % 7 == 4 -> box; 12 % 7 == 5 -> 3-number box; 13 -> no box
        H.submit_zoom_chain(sched, f"Q{first} ?", Img(), lambda r, q=q: res.__setitem__(q, r), stream_id=q, max_new_tokens=5)
    sched.run()
    assert res[0]["error"] is False and res[0]["bbox"] == [10 * 2000 / 512, 20 * 2000 / 512, 30 * 2000 / 512, 40 * 2000 / 512]
    assert res[0]["output2"] == '{"bbox_2d": [10, 20, 30, 40]} <answer>a</answer>'
    assert res[1]["error"] is True and res[1]["output1"].startswith("Error: ") and res[1]["output2"] == ""
    assert res[2] == dict(prompt=H.stage1_prompt("Q13 ?"), output1="nothing here", output2="", error=True, bbox=None)


def test_serve_dispatcher_admits_into_the_running_batch():
    """The OpenAI shim's dispatcher (src/eval/infer_vllm.py:244-271 keeps up to 100 requests in flight): greedy
    requests join the running batch, each future resolves when its own chain ends, per-request max_tokens / EOS
    trimming, sampled requests batched per sampling configuration, malformed requests rejected at submit, an engine failure
    reaches every request of the running batch and the server recovers."""
    from zoomearth_amd import serve

    model = make_model(max_seqs=3)
    model.calls = []

    def generate(input_ids=None, attention_mask=None, max_new_tokens=8, do_sample=False, **kw):
        model.calls.append((input_ids.shape[0], max_new_tokens, bool(do_sample), kw.get("seed")))
        first = int(input_ids[0][0])
        return torch.cat([input_ids, torch.tensor([expected(first, max_new_tokens)])], dim=1)

    model.generate = generate

    class SProc(Proc):
        def __call__(self, text, images=None, return_tensors="pt", padding=None, **kw):
            body = text[0].split("user\n")[1].split("<|im_end|>")[0]
            ids = torch.tensor([[ord(c) % 50 + 10 for c in body[:4]]])

            class F(dict):
                def to(self, d):
                    return self
            return F(input_ids=ids, attention_mask=torch.ones_like(ids))

    def req(text, **kw):
        return dict(messages=[{"role": "user", "content": text}], **kw)

    srv = serve.ChatServer(model, SProc(), "stub", batch_window_s=0.2)
    with pytest.raises(serve.BadRequest):
        srv.submit(req("x", stream=True))
    futs = [srv.submit(req("aaaa", max_tokens=5)), srv.submit(req("bbbb", max_tokens=3)),
            srv.submit(req("cccc", max_tokens=6, temperature=0.7, seed=9)), srv.submit(req("dd", max_tokens=4)),
            srv.submit(req("eeee")), srv.submit(req("ffff", max_tokens=2))]
    res = [f.result(timeout=10) for f in futs]
    assert model.calls == []                                 # the sampled request ran on a scheduler too (a batch of its own kind)
    firsts = [ord(c) % 50 + 10 for c in "abcdef"]
    budgets = [5, 3, 6, 4, 1024, 2]
    for i, r in enumerate(res):
        n_in = 2 if i == 3 else 4
        want = expected(firsts[i], min(budgets[i], 256 - n_in + 1))  # the KV capacity clamps the budget per request
        assert r["usage"]["completion_tokens"] == len(want) and r["usage"]["prompt_tokens"] == n_in
        assert r["choices"][0]["finish_reason"] == ("stop" if want[-1] == EOS else "length")
        assert r["choices"][0]["message"]["content"] == " ".join(str(t) for t in want if t != EOS)
    assert srv.complete(req("aaaa", max_tokens=5))["choices"] == res[0]["choices"]
    # engine failure: every request of the running batch gets it, then the server serves again
    model.engine.fail_burst = True
    f1, f2 = srv.submit(req("gggg")), srv.submit(req("hhhh"))
    for f in (f1, f2):
        with pytest.raises(RuntimeError, match="exploded"):
            f.result(timeout=10)
    model.engine.fail_burst = False
    out = {}

    def worker(i):
        out[i] = srv.submit(req("h" * (i + 1), max_tokens=4)).result(timeout=10)

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(7)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert len(out) == 7 and all(o["object"] == "chat.completion" for o in out.values())
    assert srv.scheduler.stats["admitted"] >= 7
    srv.close()


class OverlapStub(StubEngine):
    """The stub with the calls the overlapping scheduler uses: the two halves of a burst, and the batched forms of mark_seen /
    chain_tokens (one call per prefill pass / per burst)."""

    def decode_burst_begin(self, slots, steps, params):
        self._pending = self.decode_burst(slots, steps, params)
        return self._pending[0]

    def decode_burst_end(self, slots):
        return self._pending[1], self._pending[2]

    def mark_seen_batch(self, slots, ids_list):
        assert len(slots) == len(ids_list) and all(self.chains[s]["ids"] == [] or True for s in slots)
        self.log.append(("seen", list(slots), [len(x) for x in ids_list]))

    def chain_tokens_batch(self, slots, cap=0, stream=None):
        self.log.append(("tokens", list(slots)))
        return [self.chain_tokens(s, cap) for s in slots]


@pytest.mark.parametrize("hold", [0, 8])
def test_hold_keeps_the_early_members_of_an_admission_round_from_stepping_alone(hold):
    """hold_below: while the round still has prefill passes to run and fewer than `hold_below` chains are live, nobody decodes
    (a step costs almost the same at 2 chains as at 8).  Same answers either way; with the hold the first burst already has
    eight chains, without it the first two chains step beside the second pass.  The scheduler makes ONE mark_seen call per
    pass and ONE token fetch per burst that retires several chains."""
    model = make_model(max_seqs=12, max_prefill_rows=6)
    model.engine.__class__ = OverlapStub
    model.generation_config.repetition_penalty = 1.1
    sched = ChainScheduler(model, Proc(), burst=2, hold_below=hold, share_prefix=False)
    assert sched.overlap
    got = {}
    for q in range(12):
        sched.submit(Request(prompt=f"{11 + 2 * q} 50 51", images=[], max_new_tokens=5,
                             on_done=lambda r, toks, text, q=q: got.__setitem__(q, toks)))
    sched.run()
    assert got == {q: expected(11 + 2 * q, 5) for q in range(12)}
    log = model.engine.log
    bursts = [x[1] for x in log if x[0] == "burst" and x[2] > 0]
    passes = [x for x in log if x[0] == "prefill"]
    assert len(passes) == 6 and all(len(p[1]) == 2 for p in passes)          # two 3-token prompts per 6-row pass
    assert [x for x in log if x[0] == "seen"] == [("seen", p[1], [3, 3]) for p in passes]
    if hold:
        assert bursts[0] >= 8 and sched.stats["held_steps"] >= 3
        assert any(x[0] == "tokens" and len(x[1]) >= 8 for x in log)          # the chains that began together retire together
    else:
        assert bursts[0] == 2 and "held_steps" not in sched.stats
    assert not sched.live and sorted(sched.free) == list(range(12))


def test_admission_in_chunks_gives_the_same_answers_and_enqueues_the_first_pass_early():
    """Round 6, `admit_chunk_rows`: a long queue is tokenised and planned a pass's worth at a time -- the first pass is enqueued
    when a FRACTION of the queue has been through the processor, not all of it (the GPU idled 0.4 s behind 640 prompts) -- while
    (1) the questions of a tile stay in one chunk (their prefix is shared inside it: one pass-A row set per tile, as unchunked),
    (2) a chunk's last, partial pass waits for the next chunk's items (passes stay full), (3) under `hold_below` nobody decodes
    until the whole queue is in, and (4) every answer is what the unchunked scheduler gives."""
    def run(chunk):
        model = make_model(max_seqs=24, max_prefill_rows=24)
        model.engine.__class__ = OverlapStub
        calls = []

        class CountingProc(Proc):
            def __call__(self, text, images=None, return_tensors="pt", **kw):
                calls.append(len(model.engine.log))    # how much the engine had been asked to do when this prompt was tokenised
                return Proc.__call__(self, text, images=images, return_tensors=return_tensors, **kw)

        sched = ChainScheduler(model, CountingProc(), burst=2, hold_below=20, min_shared=3, admit_chunk_rows=chunk)
        got = {}
        for q in range(24):                                   # six tiles x four questions: "7t 72 73 <img> q 50"
            t = q // 4
            sched.submit(Request(prompt=f"{71 + 2 * t} 72 73 <img> {20 + q} 50", images=[f"view{t}"], max_new_tokens=4,
                                 on_done=lambda r, toks, text, q=q: got.__setitem__(q, toks)))
        sched.run()
        assert not sched.live and sorted(sched.free) == list(range(24)) and not sched._carry
        return got, model.engine.log, calls, sched.stats

    whole, log0, calls0, st0 = run(0)
    parts, log1, calls1, st1 = run(24)
    assert parts == whole == {q: expected(71 + 2 * (q // 4), 4) for q in range(24)}
    # unchunked: every prompt is tokenised before the engine hears of anything; chunked: the engine is at work from the second chunk on
    assert max(calls0) == 0 and calls1[-1] > 0 and sum(1 for c in calls1 if c == 0) < 24
    # the same rows are shared and prefilled, one copy of a tile's prefix per tile
    assert st1["shared_rows"] == st0["shared_rows"] and st1["prefill_rows"] == st0["prefill_rows"] and st1["admitted"] == 24
    copies0, copies1 = [x for x in log0 if x[0] == "copy"], [x for x in log1 if x[0] == "copy"]
    assert len(copies1) == len(copies0) == 18
    # full passes: pass B never runs with fewer than 0.85 x max_prefill_rows rows while more requests are waiting
    passes1 = [sum(x[2]) for x in log1 if x[0] == "prefill"]
    big = [r for r in passes1 if r > 8]
    assert all(r >= 20 for r in big[:-1]), passes1
    # the hold: no decode step before at least 20 chains are live
    first_burst = next(x for x in log1 if x[0] == "burst" and x[2] > 0)
    assert first_burst[1] >= 20
