"""CPU: product-side synthetic generator == oracle PRNG; host helpers == reference golden table."""
import io
import json

import numpy as np

from conftest import sha
from oracle import frontend, prng
from zoomearth_amd import hostloop as H
from zoomearth_amd import synth


def test_synth_matches_oracle_prng():
    assert np.array_equal(synth.stream64(5, 3, 100), prng.stream64(5, 3, 100))
    assert np.array_equal(synth.normal_ih4(9, 1000, 0.02), prng.normal_ih4(9, 1000, 0.02))
    assert np.array_equal(synth.uniform_ints(4, 50, 10, 99), prng.uniform_ints(4, 50, 10, 99))
    assert np.array_equal(synth.synthetic_tile(8, 70, 90), prng.synthetic_tile(8, 70, 90))
    assert synth.tensor_seed(3, "a.b.weight") == prng.tensor_seed(3, "a.b.weight")


class FakeImage:
    """Minimal PIL-like image over a numpy array (crop zero-fills, resize via the oracle)."""

    def __init__(self, arr):
        self.arr = arr

    @property
    def size(self):
        return (self.arr.shape[1], self.arr.shape[0])

    width = property(lambda self: self.arr.shape[1])
    height = property(lambda self: self.arr.shape[0])

    def crop(self, box):
        return FakeImage(frontend.crop_zero_fill(self.arr, box))

    def resize(self, size, resample=None):
        assert resample == H.BICUBIC
        return FakeImage(np.zeros((size[1], size[0], 3), np.uint8))


def coord_image(w, h):
    y, x = np.mgrid[0:h, 0:w]
    return np.stack([x & 255, y & 255, (((x >> 8) & 15) << 4) | ((y >> 8) & 15)], axis=-1).astype(np.uint8)


def test_resize_image_table(golden_json):
    for row in golden_json("host_helpers.json")["resize_image"]:
        w, h = row["size"]
        img = FakeImage(np.zeros((h, w, 3), np.uint8))
        out, scale = H.resize_image(img)
        assert list(out.size) == row["infer_size"] and scale == row["infer_scale"]
        assert list(H.resize_image_demo(img).size) == row["demo_size"]


def test_cut_image_table(golden_json):
    rows = golden_json("host_helpers.json")["cut_image"]
    assert len(rows) >= 14
    for row in rows:
        w, h = row["size"]
        out = H.cut_image(FakeImage(coord_image(w, h)), row["bbox"])
        assert list(out.size) == row["out_size"], row
        assert sha(out.arr) == row["sha256"], row


def test_cut_image_needs_four_numbers():
    import pytest
    with pytest.raises(ValueError):
        H.cut_image(FakeImage(np.zeros((600, 600, 3), np.uint8)), [1, 2, 3])


def test_extract_tables(golden_json):
    g = golden_json("host_helpers.json")
    for row in g["extract_bbox"]:
        assert H.extract_bbox(row["text"], row["scale"]) == row["infer"], row
        assert H.extract_bbox_int(row["text"], row["scale"]) == row["demo"], row
    for row in g["extract_answer"]:
        assert H.extract_answer(row["text"]) == row["answer"], row


def test_prompts_and_record(golden_json):
    g = golden_json("host_helpers.json")
    p = g["prompts"]
    assert sha(H.PREFIX) == p["prefix_sha256"] and len(H.PREFIX) == p["prefix_len"]
    assert sha(H.INSTRUCTION) == p["instruction_sha256"] and len(H.INSTRUCTION) == p["instruction_len"]
    s1 = H.stage1_prompt(p["question"])
    assert sha(s1) == p["stage1_sha256"]
    s2 = H.stage2_prompt(s1, p["output1"])
    assert sha(s2) == p["stage2_sha256"] and len(s2) == p["stage2_len"]
    r = g["record"]
    buf = io.StringIO()
    H.record(buf, r["sample"]["question"], r["sample"], r["sample"], r["output1"], r["output2"], False)
    assert buf.getvalue() == r["line"]
    buf = io.StringIO()
    H.record(buf, r["sample"]["question"], r["sample"], r["sample"], "no box", "", True)
    assert buf.getvalue() == r["line_error"]
    assert list(json.loads(r["line"]).keys()) == list(H.make_record("q", r["sample"], r["sample"], "", "", True).keys())


def test_zoom_chain_control_flow():
    calls = []

    def chat(prompts, images):
        calls.append((prompts, images))
        if len(calls) == 1:
            return ['<think>x [{"bbox_2d": [100, 120, 140, 160], "label": "t"}]</think><answer>a</answer>']
        return ["<think>y</think><answer>final</answer>"]

    img = FakeImage(coord_image(2000, 1500))
    out = H.zoom_chain("What?", img, chat)
    assert not out["error"] and H.extract_answer(out["output2"]) == "final"
    assert out["bbox"] == [v * (2000 / 512) for v in (100, 120, 140, 160)]
    assert calls[1][0][0].endswith('</think>' + H.VISION_BLOCK) and len(calls[1][1][0]) == 2
    assert calls[1][1][0][1].size == (512, 512)
    out = H.zoom_chain("What?", img, lambda p, i: ["no box at all"])
    assert out["error"] and out["output2"] == ""
