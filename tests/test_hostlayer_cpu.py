"""CPU: checkpoint reader/writer, tokenizer shim, config parsing, question sharding (incl. a 2-process gloo run)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from zoomearth_amd import accel, checkpoint
from zoomearth_amd.config import ModelConfig


def test_safetensors_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    t = {"a.weight": rng.standard_normal((5, 7)).astype(np.float32), "b.bias": rng.standard_normal(9).astype(np.float16)}
    p = str(tmp_path / "m.safetensors")
    checkpoint.write_safetensors(p, t)
    got = dict(checkpoint.iter_safetensors(p))
    assert np.array_equal(got["a.weight"], t["a.weight"]) and np.array_equal(got["b.bias"], t["b.bias"])
    # interoperable with the `safetensors` package
    from safetensors.numpy import load_file, save_file
    assert np.array_equal(load_file(p)["a.weight"], t["a.weight"])
    save_file({"c": t["a.weight"]}, str(tmp_path / "n.safetensors"))
    assert np.array_equal(dict(checkpoint.iter_safetensors(str(tmp_path / "n.safetensors")))["c"], t["a.weight"])
    # bf16 storage comes back as raw bits
    checkpoint.write_safetensors(str(tmp_path / "h.safetensors"), {"a": t["a.weight"]}, bf16=True)
    arr, tag = dict(checkpoint.iter_safetensors(str(tmp_path / "h.safetensors")))["a"]
    assert tag == "bf16" and arr.dtype == np.uint16 and arr.shape == (5, 7)
    back = (arr.astype(np.uint32) << 16).view(np.float32)
    assert np.abs(back - t["a.weight"]).max() <= np.abs(t["a.weight"]).max() * 2.0 ** -8


def test_checkpoint_index(tmp_path):
    checkpoint.write_safetensors(str(tmp_path / "model-00001-of-00002.safetensors"), {"x": np.zeros(3, np.float32)})
    checkpoint.write_safetensors(str(tmp_path / "model-00002-of-00002.safetensors"), {"y": np.ones(2, np.float32)})
    with open(tmp_path / "model.safetensors.index.json", "w") as f:
        json.dump({"weight_map": {"x": "model-00001-of-00002.safetensors", "y": "model-00002-of-00002.safetensors"}}, f)
    assert sorted(k for k, _ in checkpoint.iter_checkpoint(str(tmp_path))) == ["x", "y"]
    with pytest.raises(FileNotFoundError):
        checkpoint.checkpoint_files(str(tmp_path / "nope"))


def test_config_from_hf_json_both_layouts():
    flat = {"hidden_size": 2048, "num_hidden_layers": 36, "num_attention_heads": 16, "num_key_value_heads": 2,
            "intermediate_size": 11008, "vocab_size": 151936, "rms_norm_eps": 1e-6, "rope_theta": 1000000.0,
            "rope_scaling": {"type": "mrope", "mrope_section": [16, 24, 24]}, "tie_word_embeddings": True,
            "image_token_id": 151655, "eos_token_id": 151645,
            "vision_config": {"depth": 32, "hidden_size": 1280, "num_heads": 16, "intermediate_size": 3420,
                              "out_hidden_size": 2048, "fullatt_block_indexes": [7, 15, 23, 31]}}
    a = ModelConfig.from_hf_dict(flat)
    nested = {"text_config": {k: v for k, v in flat.items() if k not in ("vision_config", "rope_scaling", "image_token_id")},
              "vision_config": flat["vision_config"], "tie_word_embeddings": True, "image_token_id": 151655}
    nested["text_config"]["rope_parameters"] = {"rope_type": "default", "rope_theta": 1000000.0, "mrope_section": [16, 24, 24]}
    b = ModelConfig.from_hf_dict(nested)
    ref = ModelConfig.zoomearth_3b()
    for c in (a, b):
        assert c.text == ref.text and c.vision == ref.vision and c.image_token_id == 151655


def test_tokenizer_shim():
    from tiny_tok import make_tokenizer
    tok = make_tokenizer()
    ids = tok.encode("w1 w2<|vision_start|><|image_pad|><|image_pad|><|vision_end|> w7")
    assert ids == [1, 2, 2002, 2005, 2005, 2003, 7]
    tok.padding_side = "left"
    enc = tok(["w1 w2 w3", "w9"], padding="longest", return_tensors="pt")
    assert enc["input_ids"].tolist() == [[1, 2, 3], [2043, 2043, 9]]
    assert enc["attention_mask"].tolist() == [[1, 1, 1], [0, 0, 1]]
    assert tok.decode([1, 2045, 2, 2043], skip_special_tokens=True) == "w1 w2"
    assert "<|im_end|>" in tok.decode([1, 2045], skip_special_tokens=False)


def test_sharding_covers_every_question_once():
    names = [f"t{(i * 7) % 13}.tif" for i in range(200)]
    for world in (1, 2, 4, 8):
        parts = [accel.shard_by_tile(names, r, world) for r in range(world)]
        assert sorted(sum(parts, [])) == list(range(200))
        owners = {}
        for r, p in enumerate(parts):
            for i in p:
                assert owners.setdefault(names[i], r) == r  # a tile never splits across ranks
        sizes = [len(p) for p in parts]
        assert max(sizes) - min(sizes) <= 16
        rr = [accel.shard_round_robin(200, r, world) for r in range(world)]
        assert sorted(sum(rr, [])) == list(range(200))


def test_merge_results(tmp_path):
    for r, ids in enumerate(([3, 1], [2, 0])):
        with open(tmp_path / f"exp{r}.jsonl", "w") as f:
            for i in ids:
                f.write(json.dumps({"question_id": i, "answer1": "é"}, ensure_ascii=False) + "\n")
    n = accel.merge_results(str(tmp_path / "exp"), 2, str(tmp_path / "exp.jsonl"))
    rows = [json.loads(l) for l in open(tmp_path / "exp.jsonl", encoding="utf-8")]
    assert n == 4 and [r["question_id"] for r in rows] == [0, 1, 2, 3]


WORKER = r"""
import os, sys, json
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from zoomearth_amd import accel
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
acc = accel.Accelerator()
assert acc.process_index == rank and acc.num_processes == world
dist.init_process_group("gloo", rank=rank, world_size=world)
# the path's only collective: one-time broadcast of the packed weight arena (here a CPU stand-in buffer)
# both forms: scatter + all-gather (pieces of 256-byte multiples + a tail broadcast) and the plain broadcast
for mode, size in (("scatter_allgather", 4096 + 777), ("broadcast", 4096), ("scatter_allgather", 300)):
    ref = (torch.arange(size, dtype=torch.int64) * 7 % 251).to(torch.uint8)
    arena = ref.clone() if rank == 0 else torch.zeros(size, dtype=torch.uint8)
    acc.broadcast_weights(arena, src=0, mode=mode)
    assert torch.equal(arena, ref), (mode, size)
    chk = torch.tensor([int(arena.to(torch.int64).sum())])
    gathered = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(gathered, chk)
    assert all(int(g) == int(gathered[0]) for g in gathered)
# ... and the engine-level form the entry points call (from_pretrained(broadcast=True), bench.py): rank 0 holds the
# weights, the other rank receives them and is told that its arena was rewritten (weights_invalidate)
class StubEngine:
    device = torch.device("cpu")
    def __init__(self, fill):
        self.arena = (torch.arange(5000, dtype=torch.int64) * 13 % 241).to(torch.uint8) if fill else torch.zeros(5000, dtype=torch.uint8)
        self.invalidated = 0
    def weights_arena(self):
        return self.arena
    def weights_invalidate(self):
        self.invalidated += 1
for mode in ("broadcast", "scatter_allgather"):
    eng = StubEngine(rank == 0)
    secs = accel.broadcast_engine_weights(eng, rank, world, src=0, mode=mode, backend="gloo")
    assert secs > 0 and eng.invalidated == 1 and torch.equal(eng.arena, StubEngine(True).arena), mode
assert accel.broadcast_engine_weights(StubEngine(True), 0, 1) == 0.0
names = [f"t{(i * 5) % 11}.tif" for i in range(97)]
class DS(list):
    pass
ds = DS({"image_name": n, "question_id": i} for i, n in enumerate(names))
class DL:
    dataset, batch_size, collate_fn = ds, 1, staticmethod(lambda x: x)
_, dl = acc.prepare(object(), DL())
mine = [b[0]["question_id"] for b in dl]
out = [None] * world
dist.all_gather_object(out, mine)
if rank == 0:
    assert sorted(sum(out, [])) == list(range(97)), out
    print("OK", [len(o) for o in out])
dist.destroy_process_group()
"""


@pytest.mark.parametrize("world", [2, 8])
def test_two_process_gloo_sharding_and_broadcast(tmp_path, world):
    """(world = 8, round 5: the eight-rank shape -- seven receivers of the plain broadcast, the scatter + all-gather form with seven
    real peers, the tile sharding over eight ranks -- had never executed; here on CPU over gloo.)"""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = 29600 + os.getpid() % 200 + (world == 8) * 211
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "OK" in outs[0]


STEAL_WORKER = """
import json, os, sys, time
sys.path.insert(0, sys.argv[1])
from zoomearth_amd.accel import TileClaims, shard_by_tile, tile_groups
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
names = [f"tile{t:03d}.tif" for t in range(40) for _ in range(3 + (t * 7) % 9)]
lists = [tile_groups(names, shard_by_tile(names, r, world)) for r in range(world)]
claims = TileClaims.connect(rank, world, lists)
claims.store.add("test_ready", 1)                         # start together: an interpreter that came up late would find its list stolen
while claims.store.add("test_ready", 0) < world:
    time.sleep(0.01)
mine = []
for pos, (tile, idx) in enumerate(lists[rank]):          # own tiles, front to back; rank 2 is ten times slower per question
    if claims.claim(rank, pos):
        mine.append((rank, pos, tile, len(idx)))
        time.sleep((0.05 if rank == 2 else 0.002) * len(idx))
while True:                                              # own list exhausted: whole tiles of the others, from the back
    t = claims.steal()
    if t is None:
        break
    tile, idx = lists[t[0]][t[1]]
    mine.append((t[0], t[1], tile, len(idx)))
    time.sleep(0.002 * len(idx))
claims.finish()
print("RESULT " + json.dumps(dict(rank=rank, tiles=mine, stolen=claims.stolen, questions=len(names))), flush=True)
"""


def test_tile_work_stealing_covers_every_tile_exactly_once(tmp_path):
    """accel.TileClaims (work stealing of whole tiles for the drain tail): three processes over a TCPStore, the
    static LPT lists as the starting point, rank 2 ten times slower.  Every tile is processed exactly once, never split; the
    fast ranks take tiles off the slow rank's list from the back; nobody steals while its own list still has work."""
    script = tmp_path / "steal_worker.py"
    script.write_text(STEAL_WORKER)
    port = 29900 + os.getpid() % 90
    procs = []
    for r in range(3):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="3", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    res = [json.loads(next(ln for ln in o.splitlines() if ln.startswith("RESULT "))[7:]) for o in outs]
    done = [tuple(t[:3]) for r in res for t in r["tiles"]]
    assert len(done) == len(set(done)) == 40                                  # every tile once
    assert sum(t[3] for r in res for t in r["tiles"]) == res[0]["questions"]    # every question once
    by = {r["rank"]: r for r in res}
    assert by[2]["stolen"] == 0 and by[0]["stolen"] + by[1]["stolen"] >= 2       # the fast ranks relieve the slow one
    stolen_from_2 = sorted(t[1] for r in (by[0], by[1]) for t in r["tiles"] if t[0] == 2)
    own_2 = sorted(t[1] for t in by[2]["tiles"])
    assert own_2 and stolen_from_2 and max(own_2) < min(stolen_from_2)          # owner from the front, thieves from the back


def test_resume_after_an_interrupted_stolen_run_counts_every_question_once(tmp_path, monkeypatch):
    """with --steal a rank's file holds tiles it took from other ranks and lacks its own tiles that were
    taken from it.  A --resume run must skip what ANY rank's file holds (done_question_ids_all), and the merge keeps a
    question_id once even when two files hold it (files of a run resumed by the older code)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ze_infer", os.path.join(ROOT, "src", "eval", "infer.py"))
    inf = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(inf)
    monkeypatch.chdir(tmp_path)
    os.makedirs("results")
    rec = lambda q, who: json.dumps(dict(question_id=q, output1=f"by rank {who}"))  # noqa: E731
    # the interrupted run: rank 0 answered its own 0..3 and STOLE 10, 11 from rank 1; rank 1 answered 12 and died in 13 (torn line)
    with open("results/exp0.jsonl", "w") as f:
        f.write("\n".join(rec(q, 0) for q in (0, 1, 2, 3, 10, 11)) + "\n")
    with open("results/exp1.jsonl", "w") as f:
        f.write(rec(12, 1) + "\n" + '{"question_id": 13, "outp')
    assert inf.done_question_ids("results/exp1.jsonl") == {12}
    assert inf.done_question_ids_all("exp", 2) == {0, 1, 2, 3, 10, 11, 12}     # rank 1 will NOT answer 10, 11 again
    # files as the older resume left them (rank 1 re-answered the stolen 10, 11): the merge still counts them once, first record wins
    with open("results/exp1.jsonl", "w") as f:
        f.write("\n".join(rec(q, 1) for q in (12, 13, 10, 11)) + "\n")
    n = accel.merge_results("results/exp", 2, "results/exp.jsonl")
    rows = [json.loads(ln) for ln in open("results/exp.jsonl")]
    assert n == len(rows) == 8 and [r["question_id"] for r in rows] == [0, 1, 2, 3, 10, 11, 12, 13]
    assert {r["question_id"]: r["output1"] for r in rows}[10] == "by rank 0"


CLAIMS_WORKER = """
import os, sys, time
sys.path.insert(0, sys.argv[1])
from zoomearth_amd.accel import TileClaims
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
lists = [[(f"t{r}_{i}.tif", [0]) for i in range(3)] for r in range(world)]
claims = TileClaims.connect(rank, world, lists)
assert claims.claim(rank, 0)
if rank == 1 and sys.argv[2] == "crash":
    os._exit(3)                                  # dies without a word: rank 0 must not hang
if rank == 1 and sys.argv[2] == "abandon":
    assert claims.claim(rank, 1)
    claims.mark_finished(rank, 0)                # tile 0's last record has left for the file: not "unfinished" (ADVICE r5)
    claims.abandon()                             # the error path of infer.py: says so, names its unfinished tile
    sys.exit(4)
try:
    claims.finish(deadline_s=3.0)
    print("FINISHED", flush=True)
except RuntimeError as ex:
    print("DEADLINE " + str(ex), flush=True)
    sys.exit(5)
"""


@pytest.mark.parametrize("mode", ["crash", "abandon"])
def test_tile_claims_finish_does_not_hang_on_a_dead_rank(tmp_path, mode):
    """rank 0 polled `ze_ranks_done` forever.  A rank that crashes: rank 0 fails after its deadline and names the
    missing rank.  A rank that takes infer.py's error path (abandon): rank 0 finishes at once and the failing rank prints the
    tile it had claimed."""
    script = tmp_path / "claims_worker.py"
    script.write_text(CLAIMS_WORKER)
    port = 29700 + os.getpid() % 90
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, mode], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=120)[0].decode() for p in procs]
    if mode == "crash":
        assert procs[0].returncode == 5 and "DEADLINE" in outs[0] and "[1]" in outs[0], outs
        assert procs[1].returncode == 3
    else:
        assert procs[0].returncode == 0 and "FINISHED" in outs[0], outs
        assert procs[1].returncode == 4 and "t1_1.tif" in outs[1] and "t1_0.tif" not in outs[1] and "--resume" in outs[1], outs


def test_scorer(tmp_path):
    import importlib.util
    spec = importlib.util.spec_from_file_location("ze_eval", os.path.join(ROOT, "src", "eval", "eval.py"))
    ev = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ev)
    recs = [
        {"ground_truth": "Yes", "answer1": " yes ", "answer2": "no", "type": "region"},      # broken by stage 2
        {"ground_truth": "bridge", "answer1": "road", "answer2": "Bridge", "type": "object"},  # fixed by stage 2
        {"ground_truth": "3", "answer1": "3", "answer2": None, "type": "global"},             # answer2 None -> answer1
        {"ground_truth": "car", "answer1": None, "answer2": None, "type": "object"},
    ]
    r = ev.score_records(recs)
    assert (r["total"], r["correct1"], r["correct2"]) == (4, 2, 2)
    assert r["by_type"]["object"] == (2, 0, 1) and r["by_type"]["global"] == (1, 1, 1)
    assert len(r["fixed"]) == 1 and len(r["broken"]) == 1
    p = tmp_path / "r.jsonl"
    p.write_text("\n".join(json.dumps(x) for x in recs) + "\n")
    assert ev.evaluation_metrics(str(p))["total"] == 4


def test_tile_prefetcher_decodes_each_tile_once_and_ahead(tmp_path):
    """one decode per tile (the reference decodes twice per question), next tile decoded while
    the current one is in use; errors surface at get(); out-of-order requests still work."""
    import threading
    import time
    import numpy as np
    import torch
    from PIL import Image
    from zoomearth_amd.image import TilePrefetcher, decode_rgb

    paths = []
    for i in range(3):
        fp = tmp_path / f"t{i}.png"
        Image.fromarray(np.full((40 + i, 50, 3), 10 * i, dtype=np.uint8)).save(fp)
        paths.append(str(fp))
    log = []

    def decode(p):
        log.append((p, threading.current_thread().name))
        if p.endswith("bad.png"):
            raise FileNotFoundError(p)
        return decode_rgb(p)

    class Eng:
        device = torch.device("cpu")

    import zoomearth_amd.image as I
    orig = I.DeviceImage.__init__

    def cpu_init(self, base, engine, box=None, key=None):  # DeviceImage insists on a CUDA tensor: relax for the CPU test
        self.base, self.engine = base, engine
        self.box = (0, 0, int(base.shape[1]), int(base.shape[0])) if box is None else box
        self.key = key or ("img", id(base))

    I.DeviceImage.__init__ = cpu_init
    try:
        stream = [paths[0]] * 3 + [paths[1]] * 2 + [paths[2]] + [paths[0]]  # tile 0 comes back at the end
        pf = TilePrefetcher(stream, Eng(), decode=decode, pin=False)
        seen = []
        for p in stream:
            img = pf.get(p)
            seen.append((p, img.size))
            time.sleep(0.02)
        assert [s for _, s in seen] == [(50, 40)] * 3 + [(50, 41)] * 2 + [(50, 42)] + [(50, 40)]
        assert [p for p, _ in log] == [paths[0], paths[1], paths[2], paths[0]]          # one decode per tile visit
        assert all(name != threading.main_thread().name for _, name in log)             # all of them off-thread
        assert pf.get(paths[0]) is pf.get(paths[0])
        bad = str(tmp_path / "bad.png")
        pf2 = TilePrefetcher([bad, paths[1]], Eng(), decode=decode, pin=False)
        with pytest.raises(FileNotFoundError):
            pf2.get(bad)
        assert pf2.get(paths[1]).size == (50, 41)
        assert pf2.get(paths[2]).size == (50, 42)  # never announced: decoded on demand
        # several decode threads, several tiles ahead (a 50 questions/s stream needs ~5 tiles/s): same tiles, one decode each
        log.clear()
        pf3 = TilePrefetcher(stream, Eng(), decode=decode, pin=False, depth=3, workers=2)
        assert [pf3.get(p).size for p in stream] == [(50, 40)] * 3 + [(50, 41)] * 2 + [(50, 42)] + [(50, 40)]
        assert sorted(p for p, _ in log) == sorted([paths[0], paths[1], paths[2], paths[0]]) and pf3.decodes == 4
        assert pf3.decode_s > 0
    finally:
        I.DeviceImage.__init__ = orig


def test_serve_prompt_template_and_data_urls():
    """OpenAI messages -> Qwen2.5-VL chat template (what `vllm serve` applies for src/eval/infer_vllm.py)."""
    import base64
    import io
    import numpy as np
    from PIL import Image
    from zoomearth_amd import serve

    buf = io.BytesIO()
    Image.fromarray(np.arange(12 * 10 * 3, dtype=np.uint8).reshape(12, 10, 3)).save(buf, format="PNG")
    url = "data:image/png;base64," + base64.b64encode(buf.getvalue()).decode()
    msgs = [{"role": "user", "content": [{"type": "image_url", "image_url": {"url": url}}, {"type": "text", "text": "how many?"}]},
            {"role": "assistant", "content": [{"type": "text", "text": "<think>x</think>"}, {"type": "image_url", "image_url": {"url": url}}]}]
    prompt, images = serve.build_prompt(msgs)
    ph = "<|vision_start|><|image_pad|><|vision_end|>"
    assert prompt == ("<|im_start|>system\nYou are a helpful assistant.<|im_end|>\n"
                      f"<|im_start|>user\n{ph}how many?<|im_end|>\n"
                      f"<|im_start|>assistant\n<think>x</think>{ph}<|im_end|>\n"
                      "<|im_start|>assistant\n")
    assert [im.size for im in images] == [(10, 12), (10, 12)] and images[0].mode == "RGB"
    p2, im2 = serve.build_prompt([{"role": "system", "content": "s"}, {"role": "user", "content": "q"}])
    assert p2 == "<|im_start|>system\ns<|im_end|>\n<|im_start|>user\nq<|im_end|>\n<|im_start|>assistant\n" and im2 == []
    for bad in ([], [{"role": "tool", "content": "x"}], [{"role": "user", "content": [{"type": "audio"}]}],
                [{"role": "user", "content": [{"type": "image_url", "image_url": {"url": "http://x/y.png"}}]}]):
        with pytest.raises(serve.BadRequest):
            serve.build_prompt(bad)


def test_bench_gpus_flag_spawns_the_ranks():
    """`python bench.py --gpus N` outside torchrun starts N ranks itself (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*
    in their environment, rendezvous on 127.0.0.1) and rank 0 alone prints the line; under torchrun (WORLD_SIZE set)
    it does not spawn again."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--spawn-check"], capture_output=True,
                       text=True, timeout=120, env=env)
    assert r.returncode == 0, r.stderr[-1000:]
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert lines == [{"n_gpus": 3, "rank": 0, "master": "127.0.0.1", "local_rank": "0"}]
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--spawn-check"], capture_output=True,
                       text=True, timeout=120, env=dict(env, RANK="1", LOCAL_RANK="1", WORLD_SIZE="2"))
    assert r.returncode == 0 and r.stdout.strip() == ""  # a torchrun rank 1: no spawn, no line


def test_decode_rgb_reads_uncompressed_tiff_strips_directly(tmp_path):
    """zoomearth_amd/image.py decode_rgb: an uncompressed 8-bit RGB TIFF is read strip by strip straight into the array (no
    interpreter lock held: the decode threads of the tile prefetcher must not starve the threads that drive the GPU); every
    other format goes through PIL.  Same pixels either way, writable and contiguous."""
    from PIL import Image

    from zoomearth_amd.image import _raw_rgb_strips, decode_rgb
    rng = np.random.default_rng(7)
    a = rng.integers(0, 255, (257, 131, 3), dtype=np.uint8)
    for name, kw, direct in (("u.tif", {}, True), ("l.tif", {"compression": "tiff_lzw"}, False), ("p.png", {}, False)):
        fp = str(tmp_path / name)
        Image.fromarray(a).save(fp, **kw)
        with Image.open(fp) as im:
            assert (_raw_rgb_strips(im) is not None) == direct, name
        got = decode_rgb(fp)
        assert np.array_equal(got, a) and got.flags.writeable and got.flags.c_contiguous, name
    g = rng.integers(0, 255, (40, 30), dtype=np.uint8)
    fp = str(tmp_path / "g.tif")
    Image.fromarray(g).save(fp)
    assert np.array_equal(decode_rgb(fp), np.repeat(g[:, :, None], 3, axis=2))   # not RGB on disk: PIL converts


def test_trained_byte_level_bpe_round_trip_is_not_the_identity():
    """the tokenizer the GPU e2e tests run with (tests/tiny_tok.py: train_bpe) -- specials at the tiny config's ids,
    prompts of a few hundred tokens, and a decode -> strip -> re-encode round trip that DIVERGES from the generated ids on most random
    continuations (pieces re-merge, invalid UTF-8 comes back as U+FFFD), while `"bbox_2d":[...]` fragments still parse."""
    from tiny_tok import SPECIALS, make_bpe_tokenizer
    from zoomearth_amd import hostloop
    tok = make_bpe_tokenizer()
    for name, idx in SPECIALS.items():
        if name != "<unk>":
            assert tok.convert_tokens_to_ids(name) == idx
    p1 = hostloop.stage1_prompt('w3 w6 "bbox_2d":[37,91,98,132]')
    assert 150 < len(tok.encode(p1)) < 700
    text = 'w3 "bbox_2d":[37,91,98,132] w9'
    assert tok.decode(tok.encode(text), skip_special_tokens=True) == text          # text -> ids -> text is exact
    rng = np.random.default_rng(1)
    same = boxes = 0
    prefix = []
    for _ in range(200):
        ids = rng.integers(0, 2048, 14).tolist()
        out = tok.decode(ids, skip_special_tokens=True).strip()
        back = tok.encode(out)
        m = 0
        while m < min(len(back), len(ids)) and back[m] == ids[m]:
            m += 1
        prefix.append(m)
        same += back == ids
        b = hostloop.extract_bbox(out, 1.0)
        boxes += bool(b) and len(b[0]) == 4
    assert same < 100 and 1.0 < np.mean(prefix) < 12.0, (same, np.mean(prefix))       # ids -> text -> ids mostly diverges, after a few ids
    assert boxes > 60, boxes                                                           # ... and stage 2 still has boxes to zoom into
