"""Two-stage rollout driver: the generation half of one GRPO step of the reference, on the continuous-batching scheduler.

replaces: the rollout section of `Qwen2VLGRPOTrainer._generate_and_score_completions`
(/root/reference/src/train/RL/src/open-r1-multimodal/src/open_r1/trainer/grpo_trainer.py:561-683): the trainer's sampler
repeats every prompt G = num_generations times; stage 1 is one sampled `generate` over that batch; then, ONE SAMPLE AT A
TIME (`customized_funcs.chat`, :617), the first box of the completion (the whole view when none parses, :604-607) is
scaled by `max(max(w, h) / 512, 1)` to tile pixels, cut (>= 512 px window) and resized to <= 512 px, and stage 2
generates on `stage_1_prompt + completion.split("<answer>")[0] + <vision block>` with [view, crop]; samples whose
`bbox` field is empty skip stage 2 (:634-640).  The old-policy / reference-model log-probabilities are then computed on
the final prompt + completion ids from the stage-1 prompt length on (`_get_per_token_logps(...)[:, prompt_length - 1:]`,
:660-683).

Here all G x len(samples) chains advance together (sampled decoding at a temperature, one random stream per chain:
`seed`, stream = sample * G + g for stage 1 and the same + G * len(samples) for stage 2), stage 2 continues on the chain
slot of stage 1 with the cached stage-1 prompt reused, and scoring runs through `ze_score` (`model.per_token_logps`).
The gradient side of the step is out of scope.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

from . import hostloop as H
from .scheduler import ChainScheduler, Request


@dataclass
class Rollout:
    sample: int                      # index into `samples`
    generation: int                  # 0 .. G-1
    prompt1: str
    completion1: str = ""
    completion1_ids: List[int] = field(default_factory=list)
    prompt2: Optional[str] = None    # None: the sample has no box to zoom into (stage 2 skipped)
    completion2: str = ""
    completion2_ids: List[int] = field(default_factory=list)
    bbox: Optional[list] = None      # box in tile pixels actually cut (None when stage 2 was skipped)
    scale: float = 1.0
    n_prompt1: int = 0               # stage-1 prompt length in tokens
    images: list = field(default_factory=list)   # images of the FINAL prompt, in order
    logps: Optional[object] = None   # f32 tensor: log p of every token of the final sequence from position n_prompt1 on
    error: Optional[str] = None


def rollout_two_stage(model, processor, samples, num_generations: int = 4, temperature: float = 0.7,
                      max_new_tokens: int = 800, seed: int = 0, max_view: int = 512, with_logps: bool = True,
                      burst: int = 8) -> List[Rollout]:
    """samples: dicts with `prompt` (the stage-1 prompt text, one `<|vision_start|><|image_pad|><|vision_end|>` block),
    `image` (the tile: DeviceImage or PIL) and `bbox` (the dataset's reference box; empty = non-cropping question).
    Returns len(samples) * num_generations rollouts, sample-major."""
    sched = ChainScheduler(model, processor, do_sample=True, temperature=temperature, seed=seed, burst=burst)
    n, G = len(samples), int(num_generations)
    out = [Rollout(sample=i, generation=g, prompt1=samples[i]["prompt"]) for i in range(n) for g in range(G)]
    views = {}
    for i, s in enumerate(samples):
        img = s["image"]
        view = H.resize_image_demo(img, max_view)             # customized_funcs.resize_image: <= 512 px, image only
        views[i] = (img, view, max(max(img.width, img.height) / max_view, 1))

    def fail(ro):
        def on_error(req, ex):
            ro.error = f"{type(ex).__name__}: {ex}"
        return on_error

    def stage1_done(ro):
        def done(req, tokens, text):
            ro.completion1, ro.completion1_ids, ro.n_prompt1 = text, list(tokens), req.n_prompt
            img, view, scale = views[ro.sample]
            ro.images = [view]
            if not samples[ro.sample].get("bbox"):             # non-cropping question: the chain ends here
                return None
            boxes = H.extract_bbox(text, 1)
            # no parsable box -> the whole view (:604-607); a box that is not four numbers (which crashes the reference in
            # cut_image) is treated the same way
            box = boxes[0] if boxes and len(boxes[0]) == 4 else [0, 0, view.width, view.height]
            ro.scale = scale
            ro.bbox = [p * scale for p in box]
            crop = H.resize_image_demo(H.cut_image(img, ro.bbox), max_view)
            ro.prompt2 = H.stage2_prompt(ro.prompt1, text)
            ro.images = [view, crop]

            def done2(req2, tokens2, text2):
                ro.completion2, ro.completion2_ids = text2, list(tokens2)
                return None
            return Request(prompt=ro.prompt2, images=[view, crop], max_new_tokens=max_new_tokens,
                           stream_id=n * G + ro.sample * G + ro.generation, on_done=done2, on_error=fail(ro))
        return done

    for ro in out:
        sched.submit(Request(prompt=ro.prompt1, images=[views[ro.sample][1]], max_new_tokens=max_new_tokens,
                             stream_id=ro.sample * G + ro.generation, on_done=stage1_done(ro), on_error=fail(ro)))
    sched.run()

    if with_logps:
        import torch
        for ro in out:
            if ro.error:
                continue
            # the final sequence as the trainer scores it: prompt ids re-tokenised from the final prompt text, then the
            # generated ids of the last stage
            prompt = ro.prompt2 if ro.prompt2 is not None else ro.prompt1
            tail = ro.completion2_ids if ro.prompt2 is not None else ro.completion1_ids
            inp = processor(text=[prompt], images=list(ro.images), return_tensors="pt")
            ids = torch.cat([inp["input_ids"], torch.tensor([tail], dtype=torch.long)], dim=1)
            lp = model.per_token_logps(ids, torch.ones_like(ids), inp["pixel_values"], inp["image_grid_thw"],
                                       image_keys=inp.get("image_keys"))
            ro.logps = lp[0, max(ro.n_prompt1 - 1, 0):]
    return out
