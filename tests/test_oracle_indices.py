"""CPU: the oracle's integer index builders against transformers golden vectors (exact)."""
import numpy as np

from conftest import sha
from oracle import indices, qwen25vl


def test_vision_indices(golden_json):
    rows = golden_json("indices.json")["vision"]
    assert len(rows) >= 10
    for row in rows:
        g = row["grid"]
        wi, cw = indices.vision_window_index(g)
        assert wi.tolist() == row["window_index"], g
        assert cw.tolist() == row["cu_window_seqlens"], g
        assert indices.vision_cu_seqlens(g).tolist() == row["cu_seqlens"], g
        pid = indices.vision_position_ids(g)
        assert sha(pid.astype(np.int64)) == row["position_ids_sha256"], g
        assert pid[:24].tolist() == row["position_ids_head"]


def test_rope_index(golden_json):
    cfg = qwen25vl.tiny_config()
    rows = golden_json("indices.json")["rope_index"]
    assert len(rows) == 4
    for row in rows:
        pos, delta = indices.rope_index(np.array(row["input_ids"]), row["grids"], cfg.image_token_id,
                                        attention_mask=np.array(row["attention_mask"]))
        assert pos.tolist() == row["position_ids"]
        assert delta.tolist() == row["rope_deltas"]


def test_placeholder_expansion():
    ids = indices.expand_image_placeholders([1, 9, 2, 9, 3], [(1, 4, 6), (1, 2, 2)], 9)
    assert ids == [1] + [9] * 6 + [2] + [9] + [3]
