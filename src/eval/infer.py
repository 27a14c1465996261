#!/usr/bin/env python3
"""LRS-GRO batch inference on the MI355X engine: drop-in for the reference's `src/eval/infer.py`
(/root/reference/src/eval/infer.py: same CLI `--model_name --exp_name`, same ./LRS_GRO/test + ./image/ layout,
same two-stage chain, same `results/{exp_name}{rank}.jsonl` records in the rank's dataset order).

Differences, all documented in DESIGN.md: up to `--batch_size` question chains advance together on the GPU (continuous
batching, `zoomearth_amd/scheduler.py`; the reference runs `BATCH_SIZE = 1`, :27) -- a chain's output does not depend
on which chains share its steps, and is the same for every `--batch_size` up to 64 (above 64 the decode step runs on
another kernel family: same for every size above 64, equal to the first within bf16 rounding); a 5000-px tile is decoded ONCE per tile by a prefetch thread, uploaded to HBM and cropped / resized
by the HIP front-end (the reference decodes it twice per question on the CPU), and the <=512-px view of a tile is
encoded once for all its questions; sampling at T=0.01 draws from the same distribution with the engine's own random
stream; bf16 arithmetic; questions are sharded by tile across ranks; a question whose box does not parse into four
numbers is recorded as an error instead of killing the run (the reference crashes on a 3-number box).
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from torch.utils.data import DataLoader  # noqa: E402
from tqdm import tqdm  # noqa: E402

from zoomearth_amd import hostloop as H  # noqa: E402
from zoomearth_amd.accel import Accelerator  # noqa: E402
from zoomearth_amd.image import TilePrefetcher  # noqa: E402
from zoomearth_amd.modeling import ZoomEarthForConditionalGeneration  # noqa: E402
from zoomearth_amd.processor import ZoomEarthProcessor  # noqa: E402
from zoomearth_amd.scheduler import ChainScheduler  # noqa: E402

BATCH_SIZE = 64  # question chains advanced together per GPU (the reference: 1)


def collate_fn(examples):
    return examples


def stream_of(sample) -> int:
    """Random stream of a question when sampling: a function of the question alone (its id), so its output does not depend
    on the rank it landed on, its place in the rank's stream or the batch it shared."""
    import zlib
    qid = sample.get("question_id")
    return int(qid) & 0x3FFFFFFF if isinstance(qid, int) else zlib.crc32(str(qid).encode("utf-8")) & 0x3FFFFFFF


def prepare_dataloader(ds_path, collate_fn):
    from datasets import load_from_disk
    return DataLoader(load_from_disk(ds_path), batch_size=1, collate_fn=collate_fn, shuffle=False, num_workers=0)


def done_question_ids(path):
    """question_ids already recorded in a rank file (a run that was interrupted): complete JSON lines only."""
    import json
    ids = set()
    if os.path.exists(path):
        with open(path, encoding="utf-8") as f:
            for line in f:
                try:
                    ids.add(json.loads(line)["question_id"])
                except Exception:
                    pass  # a torn last line: that question runs again
    return ids


def done_question_ids_all(exp_name, world):
    """--resume: what ANY rank's file of the interrupted run holds.  With --steal a rank's file holds tiles it took from other
    ranks and lacks its own tiles that were taken from it, so a rank that looked at its own file alone would answer the stolen
    questions a second time (and the merged file would count them twice)."""
    ids = set()
    for r in range(max(1, world)):
        ids |= done_question_ids(f"results/{exp_name}{r}.jsonl")
    return ids


def eval_model_lora(model_name, exp_name, ds_path="./LRS_GRO/test", image_dir="./image/", max_new_tokens=1024,
                    batch_size=BATCH_SIZE, max_ctx=4096, do_sample=True, resume=False, decode_workers=3, decode_ahead=6, lanes=1,
                    steal=False, hold=0, admit_rows=0):
    # capacity: the stage-2 prompt holds the stage-1 prompt, its output and a second image (<= ~3200 tokens at the
    # default budgets); prefill passes of up to 16 prompts share their GEMMs
    # one rank per GPU (torchrun / accelerate launch): rank 0 reads the checkpoint, the others receive the packed weight
    # arena in one RCCL broadcast over xGMI (the reference: every rank reads it, :147-151).  ZE_WEIGHT_BROADCAST=0, or more
    # ranks than GPUs (ranks sharing a GPU cannot form an RCCL communicator): every rank loads the checkpoint itself.
    import threading

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    share = world > 1 and os.environ.get("ZE_WEIGHT_BROADCAST", "1") != "0" and \
        (os.environ.get("ZE_DIST_BACKEND", "nccl") != "nccl" or world <= torch.cuda.device_count())
    model = ZoomEarthForConditionalGeneration.from_pretrained(
        model_name, max_seqs=batch_size, max_ctx=max_ctx, max_prefill_rows=max(max_ctx, min(batch_size, 16) * 1024),
        max_patches=max(8192, min(batch_size, 32) * 1400), broadcast=share)
    if share and int(os.environ.get("RANK", "0")) == 0:
        print(f"weights broadcast to {world} ranks in {model.weight_broadcast_s:.2f} s")
    model.eval()
    processor = ZoomEarthProcessor.from_pretrained(model_name, trust_remote_code=True, max_pixels=128 * 128 * 28 * 28)
    processor.tokenizer.padding_side = "left"
    accelerator = Accelerator(mixed_precision="bf16", project_dir="checkpoints", log_with=[])
    model.generation_config.temperature = 0.01
    model.generation_config.top_p = None
    model.generation_config.top_k = None

    os.makedirs("results", exist_ok=True)
    out_path = f"results/{exp_name}{accelerator.process_index}.jsonl"
    # --resume: keep what an interrupted run of this rank wrote and skip those question_ids (the reference opens the file
    # with "w" and starts over, :167)
    skip = frozenset(done_question_ids_all(exp_name, world)) if resume else frozenset()
    if resume:
        # every rank has READ every rank's file before any rank rewrites its own (a peer caught mid-rewrite looked
        # shorter than it was, and the reader answered again the questions of its tiles that the peer had stolen)
        accelerator.wait_for_everyone()
    if resume and os.path.exists(out_path):  # drop a torn last line before appending -- also when it is the ONLY line
        with open(out_path, encoding="utf-8") as f:
            good = [ln for ln in f if ln.endswith("\n")]
        tmp = out_path + ".resume.tmp"       # (through a temporary file and os.replace: the file is never seen truncated)
        with open(tmp, "w", encoding="utf-8") as f:
            f.writelines(ln for ln in good if _is_json(ln))
        os.replace(tmp, out_path)
    fout = open(out_path, "a" if resume else "w", encoding="utf-8")
    model, dl = accelerator.prepare(model, prepare_dataloader(ds_path, collate_fn))

    def tile_path(name):
        return os.path.join(image_dir, name.split("/")[-1])

    # LANES: the rank's tiles are dealt to `lanes` engines on its GPU (each with a copy of the weights, its own KV cache,
    # scheduler, tile prefetcher, host thread and HIP stream): while one lane is in a prefill / ViT round the other decodes.
    # A chain's tokens do not depend on the lane (same weights, same kernels, its random stream keyed by the question id),
    # so the records are those of --lanes 1, in the same order.
    lanes = max(1, int(lanes))
    models = [model] + [model.clone_lane() for _ in range(lanes - 1)]
    procs = []
    for m in models:
        p = ZoomEarthProcessor(processor.tokenizer, processor.min_pixels, processor.max_pixels, processor.merge_size, engine=m.engine)
        procs.append(p)
    samples = [(i, sample) for i, sample in enumerate(s for examples in dl for s in examples)]
    # the rank's questions as TILE GROUPS (the loader lists a tile's questions together), dealt to the lanes whole
    groups = []
    for idx, sample in samples:
        if groups and groups[-1][0] == sample["image_name"]:
            groups[-1][1].append((idx, sample))
        else:
            groups.append((sample["image_name"], [(idx, sample)]))
    work = [[(g, name, items) for g, (name, items) in enumerate(groups) if g % lanes == ln] for ln in range(lanes)]
    done, lock, next_out, errors = {}, threading.Lock(), [0], []
    for idx, sample in samples:
        if sample.get("question_id") in skip:
            done[idx] = (sample, None)  # recorded by an earlier run
    extra_idx = [len(samples)]  # records of STOLEN tiles follow the rank's own, in the order they were taken
    # --steal: tile-level work stealing for the drain tail (accel.TileClaims).  Every rank knows every rank's LPT
    # list (shard_by_tile is deterministic); a claim flag per tile in a TCPStore decides who runs it.  Off with --resume (a
    # stolen tile's records live in the thief's file) and for a single rank.
    claims = None
    if steal and world > 1 and not resume:
        from zoomearth_amd.accel import TileClaims, shard_by_tile, tile_groups
        ds_all = dl.dl.dataset
        names_all = list(ds_all["image_name"]) if not isinstance(ds_all, list) else [r["image_name"] for r in ds_all]
        lists = [tile_groups(names_all, shard_by_tile(names_all, r, world)) for r in range(world)]
        assert [n for n, _ in lists[accelerator.process_index]] == [n for n, _ in groups], "the loader's order is the LPT list's"
        claims = TileClaims.connect(accelerator.process_index, world, lists)
        # (the straggler of the stealing tests is made by the claims object itself -- TileClaims.connect reads ZE_TEST_SLOW_RANK --
        #  not by a hook in this loop)
    bar = tqdm(total=len(samples), desc="Evaluating")
    # which claimed tile every output index belongs to, and how many of its records have not left yet: a tile whose last record
    # has been written is FINISHED, and an error path names only the others
    tile_of_idx, tile_left = {}, {}
    if claims is not None:
        for g, (_name, items) in enumerate(groups):
            for idx, _s in items:
                tile_of_idx[idx] = (accelerator.process_index, g)
            tile_left[(accelerator.process_index, g)] = len(items)

    def flush():  # records leave in the rank's dataset order, whatever order (and on whatever lane) the chains finish
        with lock:
            while next_out[0] in done:
                sample, r = done.pop(next_out[0])
                if r is not None:
                    H.record(fout, sample["question"], sample, sample, r["output1"], r["output2"], r["error"])
                key = tile_of_idx.get(next_out[0])
                if key is not None:
                    tile_left[key] -= 1
                    if tile_left[key] == 0:
                        claims.mark_finished(*key)
                next_out[0] += 1
                bar.update(1)

    all_stats = [None] * lanes

    def run_lane(ln):
        m, proc, todo = models[ln], procs[ln], work[ln]
        # the lane's questions arrive grouped by tile: decode the next tiles while the current one is being questioned
        tiles = TilePrefetcher([tile_path(name) for _, name, _items in todo], m.engine, depth=decode_ahead, workers=decode_workers)
        sched = ChainScheduler(m, proc, do_sample=do_sample, temperature=0.01 if do_sample else None, burst=8,
                               min_admit=max(1, batch_size // 2), max_wait_bursts=12, hold_below=hold, admit_chunk_rows=admit_rows)

        def finish(idx, sample, r):
            with lock:
                done[idx] = (sample, r)

        def submit(idx, sample, tile, view, scale):
            try:
                H.submit_zoom_chain(sched, sample["question"], tile, lambda r, idx=idx, sample=sample: finish(idx, sample, r),
                                    view=view, scale=scale, stream_id=stream_of(sample), max_new_tokens=max_new_tokens)
            except Exception as ex:  # keep going; the record marks the failure
                finish(idx, sample, dict(output1=f"Error: {ex}", output2="", error=True))

        for g, name, items in todo:
            path = tile_path(name)
            if claims is not None and not claims.claim(accelerator.process_index, g):
                tiles.skip(path)                       # another rank took this tile off the back of the list: its file has it
                for idx, sample in items:
                    finish(idx, sample, None)
                continue
            view = None  # (view, scale): every question of a tile looks at the same <=512-px view
            for idx, sample in items:
                if sample.get("question_id") in skip:  # recorded by an earlier run (--resume): the immutable set, not `done`,
                    continue                           # whose pre-populated entries another lane's flush() may already have popped
                # keep the queue short (tiles stay resident only while needed) -- and while the next tile is still being
                # decoded, advance the chains that are already in: the GPU never idles behind a decode
                while len(sched.waiting) >= batch_size or (sched.busy() and not tiles.ready(path)):
                    sched.step()
                    flush()
                try:
                    tile = tiles.get(path)
                    if view is None:
                        view = tuple(H.resize_image(tile))
                except Exception as ex:
                    finish(idx, sample, dict(output1=f"Error: {ex}", output2="", error=True))
                    continue
                submit(idx, sample, tile, view[0], view[1])
        # the lane's own list is through: whole tiles of other ranks, from the back of their lists, while the tail drains
        while claims is not None:
            while len(sched.waiting) >= batch_size:
                sched.step()
                flush()
            with lock:
                took = claims.steal()
            if took is None:
                break
            name, idxs = claims.lists[took[0]][took[1]]
            try:
                from zoomearth_amd.image import DeviceImage
                tile = DeviceImage.open(tile_path(name), m.engine)
                view = tuple(H.resize_image(tile))
            except Exception as ex:
                tile, view, err = None, None, ex
            with lock:
                tile_left[took] = len(idxs)
            for j in idxs:
                sample = ds_all[j]
                with lock:
                    idx = extra_idx[0]
                    extra_idx[0] += 1
                    bar.total += 1
                    tile_of_idx[idx] = took
                if tile is None:
                    finish(idx, sample, dict(output1=f"Error: {err}", output2="", error=True))
                else:
                    submit(idx, sample, tile, view[0], view[1])
        while sched.busy():
            sched.step()
            flush()
        st = dict(sched.stats)
        st.update(tile_decodes=tiles.decodes, tile_decode_s=round(tiles.decode_s, 3), tile_wait_s=round(tiles.wait_s, 3))
        all_stats[ln] = st

    def lane_thread(ln):
        try:
            with torch.cuda.stream(torch.cuda.Stream(device=models[ln].engine.device)):
                run_lane(ln)
                torch.cuda.current_stream().synchronize()
        except BaseException as ex:  # noqa: BLE001
            errors.append(ex)

    try:
        if lanes == 1:
            run_lane(0)
        else:
            threads = [threading.Thread(target=lane_thread, args=(ln,), name=f"ze-lane{ln}") for ln in range(lanes)]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
            if errors:
                raise errors[0]
    except BaseException:
        # a failing rank still says it will claim nothing more (rank 0 would otherwise wait for it until its deadline) and
        # names the tiles it had claimed but not finished: their questions are in no file
        if claims is not None:
            fout.flush()
            claims.abandon()
        raise
    flush()
    bar.close()
    fout.close()
    if claims is not None:
        claims.finish()
    for m in models[1:]:
        m.engine.close()
    accelerator.wait_for_everyone()
    if accelerator.is_main_process:
        print("Done! Predictions has been written to: ", out_path)
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
    stats = {}
    for st in all_stats:
        for k, v in (st or {}).items():
            stats[k] = stats.get(k, 0) + v
    stats["lanes"] = lanes
    stats["stolen_tiles"] = claims.stolen if claims is not None else 0
    if os.environ.get("ZE_PRINT_STATS") == "1":   # measurement / tests only: the schedulers' counters as one JSON line
        import json as _json
        print("[stats] " + _json.dumps({k: v for k, v in stats.items() if isinstance(v, (int, float))}), flush=True)
    if claims is not None:
        print(f"[rank {accelerator.process_index}] tiles run: {len(claims.mine)} ({claims.stolen} of them taken from other ranks' lists)", flush=True)
    return stats


def _is_json(line):
    import json
    try:
        json.loads(line)
        return True
    except Exception:
        return False


if __name__ == "__main__":
    parser = argparse.ArgumentParser(description="Evaluate ZoomEarth on LRS-GRO (MI355X engine)")
    parser.add_argument("--model_name", type=str, required=True, help="Path of the HF checkpoint directory")
    parser.add_argument("--exp_name", type=str, required=True, help="Experiment name")
    parser.add_argument("--dataset", type=str, default="./LRS_GRO/test")
    parser.add_argument("--image_dir", type=str, default="./image/")
    parser.add_argument("--max_new_tokens", type=int, default=1024)
    parser.add_argument("--batch_size", type=int, default=BATCH_SIZE, help="question chains advanced together per GPU (up to 64: a chain's output does not depend on the "
                                                                           "batch; 256 gives about 1.5x the questions/s, DESIGN.md 7b)")
    parser.add_argument("--max_ctx", type=int, default=4096, help="tokens per chain (KV capacity)")
    parser.add_argument("--greedy", action="store_true", help="arg-max instead of the reference's T=0.01 sampling")
    parser.add_argument("--resume", action="store_true", help="keep the records results/{exp_name}{rank}.jsonl already holds and "
                                                              "run only the questions that are missing")
    parser.add_argument("--lanes", type=int, default=1, help="engines per GPU, each with its own scheduler thread and --batch_size "
                                                             "chain slots: the prefill rounds of one overlap the decode bursts of the "
                                                             "other (2 x 512 slots answer 13 %% more questions/s than 1 x 512)")
    parser.add_argument("--steal", action="store_true", default=os.environ.get("ZE_STEAL") == "1",
                        help="several ranks: a rank whose own tiles are through takes whole tiles off the back of the other ranks' "
                             "lists (one claim flag per tile in a TCPStore on MASTER_PORT + 17; needs the ranks to run at once)")
    parser.add_argument("--admit_rows", type=int, default=int(os.environ.get("ZE_ADMIT_ROWS", "0")),
                        help="throughput setting: take a long queue of waiting questions in chunks of about this many prompt rows (a prefill "
                             "pass's worth and more) instead of tokenising all of it before the first pass; 0 = the whole queue at once")
    parser.add_argument("--hold", type=int, default=int(os.environ.get("ZE_HOLD", "0")),
                        help="throughput setting: while an admission round still has prefill passes to run and fewer than this many "
                             "chains of a lane are live, the live ones wait for the newcomers instead of stepping alone (a decode step "
                             "costs almost the same at 60 chains as at 400); 0 = never hold")
    parser.add_argument("--decode_workers", type=int, default=3, help="tile decode threads per lane")
    parser.add_argument("--decode_ahead", type=int, default=6, help="tiles decoded ahead of the one in use (75 MB pinned each)")
    args = parser.parse_args()
    eval_model_lora(args.model_name, args.exp_name, args.dataset, args.image_dir, args.max_new_tokens, args.batch_size,
                    args.max_ctx, do_sample=not args.greedy, resume=args.resume, decode_workers=args.decode_workers,
                    decode_ahead=args.decode_ahead, lanes=args.lanes, steal=args.steal, hold=args.hold, admit_rows=args.admit_rows)
