"""The REAL host path of BASELINE configs[3] on one GPU: `src/infer.py` (the drop-in of /root/reference/src/eval/infer.py:
145-252) on an Arrow dataset of synthetic questions about synthetic 5000 x 5000 tiles that live on disk as TIFF files --
tile decode, pinned staging and upload, tokeniser, processor, scheduler, JSONL records, all inside the timed run -- with
the 3B-shape synthetic checkpoint (config.json asks for the repo's synthetic weights: no 7.5-GB file).

The tokenizer is word-level over the whole 151,936-id vocabulary; every 40th id decodes to a `"bbox_2d":[...]` fragment,
so a stage-1 output of random weights almost always holds a parsable box and stage 2 really runs (full-resolution crop,
second ViT pass, follow-up prefill).  Random weights never emit EOS: both stages run to --max_new_tokens.

usage: python tools/bench_infer_e2e.py [--questions 640] [--batch_size 256] [--max_new_tokens 192] [--workdir DIR]
prints one JSON line: questions/s of the entry point (model load excluded and included), TilePrefetcher tiles/s alone,
time the stream waited for tiles, scheduler statistics."""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "src", "eval"))

import numpy as np  # noqa: E402


def word(i: int) -> str:
    if i % 40 == 7:  # (distinct numbers per id: a word-level vocabulary needs distinct words)
        k = i // 40
        x, y = k % 380, (k // 380) * 37 % 380
        return f'"bbox_2d":[{x},{y},{x + 20 + k % 90},{y + 20 + (k // 90) % 90}]'
    return f"w{i}"


def build(d, n_questions, tile_side, n_pool):
    from datasets import Dataset
    from PIL import Image
    from tokenizers import AddedToken, Tokenizer
    from tokenizers.models import WordLevel
    from tokenizers.pre_tokenizers import WhitespaceSplit
    from zoomearth_amd.synth import synthetic_tile, uniform_ints

    ck = os.path.join(d, "ckpt")
    os.makedirs(ck, exist_ok=True)
    cfg = {"vision_config": dict(depth=32, hidden_size=1280, num_heads=16, intermediate_size=3420, out_hidden_size=2048,
                                 fullatt_block_indexes=[7, 15, 23, 31]),
           "hidden_size": 2048, "num_hidden_layers": 36, "num_attention_heads": 16, "num_key_value_heads": 2,
           "intermediate_size": 11008, "vocab_size": 151936, "rms_norm_eps": 1e-6, "rope_theta": 1000000.0,
           "rope_scaling": {"type": "mrope", "mrope_section": [16, 24, 24]}, "tie_word_embeddings": True,
           "image_token_id": 151655, "vision_start_token_id": 151652, "vision_end_token_id": 151653,
           "eos_token_id": [151645, 151643], "pad_token_id": 151643,
           "zoomearth_synthetic_weights": {"seed": 0, "std": 0.02}}
    with open(os.path.join(ck, "config.json"), "w") as f:
        json.dump(cfg, f)
    with open(os.path.join(ck, "generation_config.json"), "w") as f:
        json.dump({"eos_token_id": [151645, 151643], "pad_token_id": 151643, "repetition_penalty": 1.05}, f)
    specials = {"<|endoftext|>": 151643, "<|im_start|>": 151644, "<|im_end|>": 151645, "<|vision_start|>": 151652,
                "<|vision_end|>": 151653, "<|image_pad|>": 151655, "<unk>": 151935}
    vocab = {word(i): i for i in range(151936) if i not in specials.values()}
    vocab.update(specials)
    tok = Tokenizer(WordLevel(vocab, unk_token="<unk>"))
    tok.pre_tokenizer = WhitespaceSplit()
    tok.add_special_tokens([AddedToken(t, special=True) for t in specials if t != "<unk>"])
    tok.save(os.path.join(ck, "tokenizer.json"))
    with open(os.path.join(ck, "tokenizer_config.json"), "w") as f:
        json.dump({"pad_token": "<|endoftext|>"}, f)
    # tiles: 3..18 questions each (10.5 on average, LRS-GRO: 10.7); pixel content from a small pool of synthetic tiles,
    # every file written out in full (uncompressed TIFF, as remote-sensing tiles usually ship)
    os.makedirs(os.path.join(d, "image"), exist_ok=True)
    pool = [synthetic_tile(1000 + t, tile_side, tile_side) for t in range(n_pool)]
    rows, t, t0 = [], 0, time.perf_counter()
    while len(rows) < n_questions:
        c = min(3 + int(uniform_ints(424_242 + t, 1, 0, 16)[0]), n_questions - len(rows))
        arr = np.roll(pool[t % n_pool], shift=(37 * t) % tile_side, axis=1)  # distinct bytes per tile at memcpy cost
        Image.fromarray(arr).save(os.path.join(d, "image", f"tile{t:04d}.tif"))
        for k in range(c):
            q = len(rows)
            rows.append({"question": " ".join(word(int(v)) for v in uniform_ints(500 + q, 12 + q % 9, 0, 151000)),
                         "image_name": f"LRS_GRO/tile{t:04d}.tif", "question_id": 100000 + q, "ground_truth": word(3 * q),
                         "category": "cat%d" % (q % 2), "type": ("count", "object", "relation")[q % 3],
                         "bbox": [1.0 * k, 2.0, 30.0 + k, 40.0]})
        t += 1
    Dataset.from_list(rows).save_to_disk(os.path.join(d, "LRS_GRO", "test"))
    return t, time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--questions", type=int, default=640)
    ap.add_argument("--batch_size", type=int, default=256)
    ap.add_argument("--max_new_tokens", type=int, default=192)
    ap.add_argument("--tile", type=int, default=5000)
    ap.add_argument("--decode_workers", type=int, default=3)
    ap.add_argument("--decode_ahead", type=int, default=6)
    ap.add_argument("--lanes", type=int, default=1)
    ap.add_argument("--hold", type=int, default=0)
    ap.add_argument("--admit_rows", type=int, default=0)
    ap.add_argument("--workdir", default=None)
    args = ap.parse_args()
    d = args.workdir or tempfile.mkdtemp(prefix="ze_e2e_")
    n_tiles, build_s = build(d, args.questions, args.tile, 4)
    os.chdir(d)
    import torch
    from zoomearth_amd.image import TilePrefetcher, decode_rgb
    paths = sorted(os.path.join(d, "image", f) for f in os.listdir(os.path.join(d, "image")))

    class Eng:
        device = torch.device("cuda", 0)

    # TilePrefetcher alone: decode (+ pinning) + upload of every tile, no model
    out = {"questions": args.questions, "tiles": n_tiles, "tile_side": args.tile, "files": "uncompressed TIFF, %.0f MB each" %
           (os.path.getsize(paths[0]) / 1e6), "build_s": round(build_s, 1)}
    for workers, ahead in ((1, 1), (args.decode_workers, args.decode_ahead), (4, 6)):
        sub = paths[: min(len(paths), 24)]
        t0 = time.perf_counter()
        pf = TilePrefetcher(sub, Eng(), depth=ahead, workers=workers)
        for p in sub:
            pf.get(p)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[f"prefetcher_tiles_per_s_w{workers}_a{ahead}"] = round(len(sub) / dt, 2)
        out[f"decode_s_per_tile_w{workers}_a{ahead}"] = round(pf.decode_s / max(1, pf.decodes), 3)
    t0 = time.perf_counter()
    one = decode_rgb(paths[0])
    out["decode_rgb_s"] = round(time.perf_counter() - t0, 3)
    del one
    import infer  # src/eval/infer.py
    t0 = time.perf_counter()
    marks = {}
    real = infer.ChainScheduler

    class Timed(real):  # the clock starts when the model is loaded and the scheduler exists
        def __init__(self, *a, **kw):
            marks.setdefault("loaded", time.perf_counter())
            super().__init__(*a, **kw)
            marks.setdefault("engines", []).append(self.engine)
            self.engine.phase_timers(enable=True, reset=True)

    infer.ChainScheduler = Timed
    prof = None
    if os.environ.get("ZE_E2E_PROFILE") == "1":  # host-side profile of the calling thread (meaningful with --lanes 1)
        import cProfile
        prof = cProfile.Profile()
        prof.enable()
    stats = infer.eval_model_lora("ckpt", "e2e_", "./LRS_GRO/test", "./image/", args.max_new_tokens, args.batch_size, 2048,
                                  do_sample=False, decode_workers=args.decode_workers, decode_ahead=args.decode_ahead, lanes=args.lanes, hold=args.hold, admit_rows=args.admit_rows)
    if prof is not None:
        import io
        import pstats
        prof.disable()
        buf = io.StringIO()
        pstats.Stats(prof, stream=buf).sort_stats("tottime").print_stats(28)
        sys.stderr.write(buf.getvalue())
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    phases = {}
    for en in marks["engines"]:
        if en.h:
            for k, v in en.phase_timers(enable=False).items():
                phases[k] = phases.get(k, 0.0) + v
    recs = [json.loads(l) for l in open("results/e2e_0.jsonl")]
    out.update({
        "entry_point": "src/eval/infer.py eval_model_lora, --batch_size %d --max_new_tokens %d --greedy --lanes %d --hold %d --admit_rows %d" % (args.batch_size, args.max_new_tokens, args.lanes, args.hold, args.admit_rows),
        "model_load_s": round(marks["loaded"] - t0, 2),
        "stream_s": round(t1 - marks["loaded"], 2),
        "questions_per_s": round(len(recs) / (t1 - marks["loaded"]), 2),
        "questions_per_s_incl_load": round(len(recs) / (t1 - t0), 2),
        "records": len(recs), "stage2_ran": sum(1 for r in recs if not r["error"]),
        "mean_words_stage1": float(np.mean([len(r["stage1"].split()) for r in recs])),
        "gpu_phase_s": {k: round(v / 1000.0, 2) for k, v in phases.items()},
        "host_gap_s": round((t1 - marks["loaded"]) - sum(phases.values()) / 1000.0, 2),
        "scheduler": stats,
    })
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
