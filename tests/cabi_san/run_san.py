"""Runs the host-side integer helpers of ze_index.cpp, built with ASan + UBSan, against the committed golden vectors.
Started by tests/test_cabi_cpu.py as a child process with libasan preloaded; any sanitizer report aborts the process."""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import frontend  # noqa: E402  (the checker of the tap tables)

I32P, I64P = C.POINTER(C.c_int32), C.POINTER(C.c_int64)


def p32(a):
    return a.ctypes.data_as(I32P)


def main(lib_path):
    lib = C.CDLL(lib_path)
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "indices.json")))
    n = 0
    for row in gold["smart_resize"]:
        oh, ow = C.c_int(), C.c_int()
        lib.san_smart_resize.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        assert lib.san_smart_resize(row["h"], row["w"], 28, 3136, row["max_pixels"], C.byref(oh), C.byref(ow)) == 0
        assert [oh.value, ow.value] == row["out"], row
        n += 1
    for row in gold["vision"]:
        g = np.ascontiguousarray(row["grid"], dtype=np.int32)
        patches = int((g[:, 0] * g[:, 1] * g[:, 2]).sum())
        wi = np.zeros(patches // 4, dtype=np.int64)
        cu = np.zeros(patches // 4 + 2, dtype=np.int32)
        k = lib.san_window_index(p32(g), len(g), 2, 112, 14, wi.ctypes.data_as(I64P), len(wi), p32(cu), len(cu))
        assert k > 0 and wi.tolist() == row["window_index"] and cu[:k].tolist() == row["cu_window_seqlens"]
        hw = np.zeros((patches, 2), dtype=np.int32)
        assert lib.san_vision_pos_ids(p32(g), len(g), 2, p32(hw), patches) == 0
        import hashlib
        assert hashlib.sha256(np.ascontiguousarray(hw.astype(np.int64)).tobytes()).hexdigest() == row["position_ids_sha256"]
        assert hw[:24].tolist() == row["position_ids_head"]
        n += 1
    for row in gold["rope_index"]:
        ids, am = np.asarray(row["input_ids"]), np.asarray(row["attention_mask"])
        grids = np.asarray(row["grids"], dtype=np.int32)
        img = 2005  # image_token_id of the tiny config the fixtures were generated with (oracle/qwen25vl.py: tiny_config)
        gi = 0
        for b in range(ids.shape[0]):
            cur = np.ascontiguousarray(ids[b][am[b].astype(bool)], dtype=np.int32)
            n_img = int(((cur == img) & (np.roll(cur, 1) != img)).sum())
            g = np.ascontiguousarray(grids[gi: gi + n_img])
            gi += n_img
            pos = np.zeros((3, len(cur)), dtype=np.int32)
            d = C.c_int32()
            assert lib.san_rope_index(p32(cur), len(cur), p32(g), n_img, img, 2, p32(pos), C.byref(d)) == 0
            want = np.asarray(row["position_ids"])[:, b][:, am[b].astype(bool)]
            assert pos.tolist() == want.tolist() and d.value == row["rope_deltas"][b][0]
            n += 1
    # malformed inputs must be refused, not read out of bounds: more image runs than grids, an empty prompt
    bad = np.array([5, 9, 9, 7, 9], dtype=np.int32)
    g = np.array([[1, 4, 4]], dtype=np.int32)
    pos = np.zeros((3, 5), dtype=np.int32)
    d = C.c_int32()
    assert lib.san_rope_index(p32(bad), 5, p32(g), 1, 9, 2, p32(pos), C.byref(d)) != 0
    # bicubic tap tables (Pillow's fixed-point coefficients) at the sizes the path uses and a few awkward ones
    for in_size, out_size in ((5000, 512), (5000, 511), (3000, 307), (512, 504), (307, 308), (1024, 1036), (7, 3), (3, 7), (1, 1), (2500, 512)):
        xmin = np.zeros(out_size, dtype=np.int32)
        xcnt = np.zeros(out_size, dtype=np.int32)
        kk = np.zeros(out_size * 64, dtype=np.int32)
        ks = lib.san_bicubic_coeffs(in_size, out_size, p32(xmin), p32(xcnt), p32(kk), len(kk))
        assert ks > 0
        w_min, w_cnt, w_kk = frontend.bicubic_coeffs(in_size, out_size)
        assert xmin.tolist() == w_min.tolist() and xcnt.tolist() == w_cnt.tolist()
        for i in range(out_size):
            assert kk[i * ks: i * ks + xcnt[i]].tolist() == w_kk[i][: xcnt[i]].tolist(), (in_size, out_size, i)
        n += 1
    print(f"sanitized host helpers ok: {n} cases")


if __name__ == "__main__":
    main(sys.argv[1])
