#!/usr/bin/env python3
"""LRS-GRO batch inference on the MI355X engine: drop-in for the reference's `src/eval/infer.py`
(/root/reference/src/eval/infer.py: same CLI `--model_name --exp_name`, same ./LRS_GRO/test + ./image/ layout,
same two-stage chain, same `results/{exp_name}{rank}.jsonl` records).

Differences, all documented in DESIGN.md: a 5000-px tile is decoded ONCE per tile by a prefetch thread, uploaded to
HBM and cropped/resized by the HIP front-end (the reference decodes it twice per question on the CPU); sampling at
T=0.01 draws from the same distribution with the engine's own random stream; bf16 arithmetic; questions are
sharded by tile across ranks; every question is wrapped in try/except so one malformed bbox does not kill the run
(the reference crashes on a 3-number box).
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

BATCH_SIZE = 1


def collate_fn(examples):
    return examples


def prepare_dataloader(ds_path, collate_fn):
    from datasets import load_from_disk
    return DataLoader(load_from_disk(ds_path), batch_size=BATCH_SIZE, collate_fn=collate_fn, shuffle=False, num_workers=0)


def eval_model_lora(model_name, exp_name, ds_path="./LRS_GRO/test", image_dir="./image/", max_new_tokens=1024):
    model = ZoomEarthForConditionalGeneration.from_pretrained(model_name)
    model.eval()
    processor = ZoomEarthProcessor.from_pretrained(model_name, trust_remote_code=True, max_pixels=128 * 128 * 28 * 28)
    processor.tokenizer.padding_side = "left"
    accelerator = Accelerator(mixed_precision="bf16", project_dir="checkpoints", log_with=[])
    model.generation_config.temperature = 0.01
    model.generation_config.top_p = None
    model.generation_config.top_k = None

    os.makedirs("results", exist_ok=True)
    out_path = f"results/{exp_name}{accelerator.process_index}.jsonl"
    fout = open(out_path, "w", encoding="utf-8")
    model, dl = accelerator.prepare(model, prepare_dataloader(ds_path, collate_fn))

    def chat(prompts, images):
        return H.chat_batch(prompts, images, processor, model, device=accelerator.device, do_sample=True,
                            temperature=0.01, max_new_tokens=max_new_tokens)

    def tile_path(sample):
        return os.path.join(image_dir, sample["image_name"].split("/")[-1])

    # the rank's questions arrive grouped by tile: decode the next tile while the current one is being questioned
    tiles = TilePrefetcher([tile_path(s) for ex in dl for s in ex], model.engine)
    for examples in tqdm(dl, desc="Evaluating"):
        for sample in examples:
            try:
                r = H.zoom_chain(sample["question"], tiles.get(tile_path(sample)), chat)
                H.record(fout, sample["question"], sample, sample, r["output1"], r["output2"], r["error"])
            except Exception as ex:  # keep going; the record marks the failure
                H.record(fout, sample["question"], sample, sample, f"Error: {ex}", "", True)
    fout.close()
    accelerator.wait_for_everyone()
    if accelerator.is_main_process:
        print("Done! Predictions has been written to: ", out_path)


if __name__ == "__main__":
    parser = argparse.ArgumentParser(description="Evaluate ZoomEarth on LRS-GRO (MI355X engine)")
    parser.add_argument("--model_name", type=str, required=True, help="Path of the HF checkpoint directory")
    parser.add_argument("--exp_name", type=str, required=True, help="Experiment name")
    parser.add_argument("--dataset", type=str, default="./LRS_GRO/test")
    parser.add_argument("--image_dir", type=str, default="./image/")
    parser.add_argument("--max_new_tokens", type=int, default=1024)
    args = parser.parse_args()
    eval_model_lora(args.model_name, args.exp_name, args.dataset, args.image_dir, args.max_new_tokens)
