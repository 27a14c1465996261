"""GPU: image front-end (K0-K2) through the C ABI vs the oracle and the Pillow/transformers goldens. Bit-exact."""
import numpy as np
import pytest
import torch

from conftest import npz_str, sha
from gpu_util import tiny_engine  # noqa: F401
from oracle import frontend, prng

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda")


CASES = [  # seed, h, w, box, out_w, out_h
    (21, 100, 80, None, 29, 37), (23, 512, 512, None, 504, 504), (24, 517, 300, None, 308, 504),
    (25, 1000, 1000, None, 512, 512), (27, 64, 64, None, 128, 100), (30, 33, 500, None, 500, 33),
    (31, 900, 1100, (100, 50, 612, 562), 512, 512),            # pure crop
    (32, 900, 1100, (-40, -30, 500, 700), 300, 411),           # crop leaving the image + resize both axes
    (33, 900, 1100, (600, 500, 1300, 1000), 350, 500),         # out of image right/bottom, vertical pass skipped
    (34, 900, 1100, (10, 20, 522, 700), 512, 340),             # horizontal pass skipped
    (35, 700, 3000, None, 64, 15),                             # very wide taps (scale 46: unstaged kernel)
]


@pytest.mark.parametrize("seed,h,w,box,ow,oh", CASES)
def test_crop_resize_matches_oracle(tiny_engine, seed, h, w, box, ow, oh):
    img = prng.synthetic_tile(seed, h, w)
    box = box or (0, 0, w, h)
    want = frontend.crop_zero_fill(img, box)
    if (want.shape[1], want.shape[0]) != (ow, oh):
        want = frontend.resize_bicubic(want, ow, oh)
    got = tiny_engine.crop_resize(dev(img), box, (ow, oh)).cpu().numpy()
    assert got.shape == want.shape
    assert np.array_equal(got, want), np.abs(got.astype(int) - want).max()


def test_5000px_tile_against_pillow_golden(tiny_engine, golden_npz, big_tile):
    z = golden_npz("bicubic.npz")
    t = dev(big_tile)
    got = tiny_engine.crop_resize(t, (0, 0, 5000, 5000), (512, 512)).cpu().numpy()
    assert np.array_equal(got[::64], z["big_5000_rows"])
    assert sha(got) == npz_str(z["big_5000_sha256"])
    got2 = tiny_engine.crop_resize(t, (0, 0, 5000, 3000), (512, 307)).cpu().numpy()
    assert sha(got2) == npz_str(z["big_5000x3000_sha256"])
    got3 = tiny_engine.crop_resize(t, (1000, 1200, 3500, 3300), (512, 430)).cpu().numpy()
    assert sha(got3) == npz_str(z["crop_1000_1200_3500_3300_to_430x512_sha256"])


def test_pixel_values_against_transformers_golden(tiny_engine, golden_npz):
    z = golden_npz("pixel_values.npz")
    for key in sorted({k.rsplit("_", 1)[0] for k in z.files}):
        seed, hw = key[1:].split("_")
        h, w = (int(v) for v in hw.split("x"))
        img = prng.synthetic_tile(int(seed), h, w)
        pv, grid = tiny_engine.preprocess_image(dev(img))
        pv = pv.cpu().numpy()
        assert list(grid) == z[key + "_grid"][0].tolist()
        assert sha(pv) == npz_str(z[key + "_sha256"]), key
        want, _ = frontend.image_to_pixel_values(img)
        assert np.array_equal(pv, want)


def test_patchify_op(tiny_engine):
    img = prng.synthetic_tile(77, 56, 84)
    got = tiny_engine.patchify(dev(img)).cpu().numpy()
    want, gh, gw = frontend.patchify(img)
    assert (gh, gw) == (4, 6) and np.array_equal(got, want)


def test_front_end_errors(tiny_engine):
    from zoomearth_amd._lib import ZoomEarthError
    img = dev(prng.synthetic_tile(1, 64, 64))
    with pytest.raises(ZoomEarthError):
        tiny_engine.crop_resize(img, (10, 10, 10, 40), (5, 5))
    with pytest.raises(ZoomEarthError):
        tiny_engine.patchify(dev(prng.synthetic_tile(1, 30, 56)))
    with pytest.raises(ZoomEarthError):
        tiny_engine.preprocess_image(dev(prng.synthetic_tile(1, 4, 900)))


def test_front_end_from_two_streams(tiny_engine, big_tile):
    """The front-end workspace (coefficient tables, pinned staging, intermediate images) is ONE per engine, and the
    scheduler calls it on its side stream while the loop that feeds it resizes the next tile's view on its own stream
    (zoomearth_amd/scheduler.py step(); src/eval/infer.py run_lane).  Calls alternating between two streams, with nothing
    but the engine's own ordering between them, give the results of the same calls made one after the other."""
    t = dev(big_tile)
    small = dev(prng.synthetic_tile(5, 300, 280))
    jobs = [(t, (0, 0, 5000, 5000), (512, 512)), (small, (10, 20, 250, 260), (140, 84)), (t, (1000, 1200, 3500, 3300), (512, 430)),
            (small, (0, 0, 280, 300), (308, 336)), (t, (40, 4000, 900, 4990), (448, 476)), (small, (100, 100, 180, 190), (28, 56))]
    want = [tiny_engine.crop_resize(*j).cpu() for j in jobs]
    want_pv = tiny_engine.preprocess_image(small)[0].cpu()
    torch.cuda.synchronize()
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    for rep in range(12):
        got = []
        for i, j in enumerate(jobs):
            with torch.cuda.stream(a if (i + rep) % 2 else b):
                got.append(tiny_engine.crop_resize(*j))
        with torch.cuda.stream(a if rep % 2 else b):
            pv = tiny_engine.preprocess_image(small)[0]
        torch.cuda.synchronize()
        for g, w in zip(got, want):
            assert torch.equal(g.cpu(), w), rep
        assert torch.equal(pv.cpu(), want_pv), rep
