#!/usr/bin/env python3
"""One question, two-stage zoom chain, greedy: drop-in for the reference's `src/demo.py`
(/root/reference/src/demo.py:126-154: `chat(prompt, image_fp)`, view <= 1024 px, int bbox parsing).

Fixes of reference defects, documented in DESIGN.md: the stage-1 prompt includes the vision prefix (the reference
defines PREFIX but never uses it, so HF raises "Image features and image tokens do not match"); the model is
loaded once, not on every call; the model class is the Qwen2.5-VL conditional-generation one.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from zoomearth_amd import hostloop as H  # noqa: E402
from zoomearth_amd.image import DeviceImage  # noqa: E402
from zoomearth_amd.modeling import ZoomEarthForConditionalGeneration  # noqa: E402
from zoomearth_amd.processor import ZoomEarthProcessor  # noqa: E402

MODEL_PATH = os.environ.get("ZOOMEARTH_MODEL", "")
_STATE = {}


def _load():
    if "model" not in _STATE:
        _STATE["model"] = ZoomEarthForConditionalGeneration.from_pretrained(MODEL_PATH)
        _STATE["processor"] = ZoomEarthProcessor.from_pretrained(MODEL_PATH)
    return _STATE["processor"], _STATE["model"]


def chat_batch(prompts, imgs, processor, model):
    return H.chat_batch(prompts, imgs, processor, model, do_sample=False)


def chat(prompt, image_fp):
    processor, model = _load()
    image = DeviceImage.open(image_fp, model.engine)
    scale = max(1, max(image.width, image.height) / 1024)
    view = H.resize_image_demo(image)
    text = H.PREFIX + prompt + H.INSTRUCTION
    output1 = chat_batch([text], [view], processor, model)[0]
    bboxs = H.extract_bbox_int(output1, scale)
    if bboxs != []:
        crop = H.resize_image_demo(H.cut_image(image, bboxs[0]))
        new_prompt = text + output1.split("<answer>")[0] + H.VISION_BLOCK
        return chat_batch([new_prompt], [[view, crop]], processor, model)[0]
    return output1


if __name__ == "__main__":
    prompt = "Are there any building on the top-right island?"
    image_fp = sys.argv[1] if len(sys.argv) > 1 else "./images/demo3.png"
    print(chat(prompt=prompt, image_fp=image_fp))
