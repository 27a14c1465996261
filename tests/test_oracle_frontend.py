"""CPU: the oracle's integer front-end against the Pillow / transformers golden vectors."""
import re

import numpy as np

from conftest import npz_str, sha
from oracle import frontend, prng


def test_bicubic_matches_pillow_goldens(golden_npz):
    z = golden_npz("bicubic.npz")
    pat = re.compile(r"s(\d+)_(\d+)x(\d+)_to_(\d+)x(\d+)(_sha256|_rows16)?$")
    seen = 0
    for key in z.files:
        m = pat.match(key)
        if not m or m.group(6) == "_rows16":
            continue
        seed, h, w, oh, ow = (int(m.group(i)) for i in range(1, 6))
        got = frontend.resize_bicubic(prng.synthetic_tile(seed, h, w), ow, oh)
        if m.group(6) == "_sha256":
            assert sha(got) == npz_str(z[key]), key
            assert np.array_equal(got[::16], z[key.replace("_sha256", "_rows16")])
        else:
            assert np.array_equal(got, z[key]), key
        seen += 1
    assert seen >= 10


def test_bicubic_5000_to_512(golden_npz, big_tile):
    z = golden_npz("bicubic.npz")
    got = frontend.resize_bicubic(big_tile, 512, 512)
    assert np.array_equal(got[::64], z["big_5000_rows"])
    assert sha(got) == npz_str(z["big_5000_sha256"])
    got2 = frontend.resize_bicubic(big_tile[:3000], 512, 307)
    assert sha(got2) == npz_str(z["big_5000x3000_sha256"])
    crop = frontend.crop_zero_fill(big_tile, (1000, 1200, 3500, 3300))
    assert sha(frontend.resize_bicubic(crop, 512, 430)) == npz_str(z["crop_1000_1200_3500_3300_to_430x512_sha256"])


def test_smart_resize(golden_json):
    for row in golden_json("indices.json")["smart_resize"]:
        assert list(frontend.smart_resize(row["h"], row["w"], 28, 3136, row["max_pixels"])) == row["out"], row


def test_pixel_values(golden_npz):
    z = golden_npz("pixel_values.npz")
    keys = sorted({k.rsplit("_", 1)[0] for k in z.files})
    assert len(keys) >= 6
    for key in keys:
        seed, hw = key[1:].split("_")
        h, w = (int(v) for v in hw.split("x"))
        pv, grid = frontend.image_to_pixel_values(prng.synthetic_tile(int(seed), h, w))
        assert list(grid) == z[key + "_grid"][0].tolist()
        assert pv.dtype == np.float32
        assert sha(pv) == npz_str(z[key + "_sha256"]), key
        assert np.array_equal(pv[:: max(1, pv.shape[0] // 7)][:8], z[key + "_rows"])


def test_crop_zero_fill_edges():
    img = prng.synthetic_tile(3, 20, 30)
    assert frontend.crop_zero_fill(img, (0, 0, 30, 20)).tobytes() == img.tobytes()
    c = frontend.crop_zero_fill(img, (-5, -3, 10, 8))
    assert c.shape == (11, 15, 3) and c[:3].sum() == 0 and c[:, :5].sum() == 0
    assert np.array_equal(c[3:, 5:], img[:8, :10])
    assert frontend.crop_zero_fill(img, (5, 5, 5, 9)).shape == (4, 0, 3)
    assert frontend.crop_zero_fill(img, (40, 40, 50, 50)).sum() == 0
