"""Model configuration of the Qwen2.5-VL backbone used by ZoomEarth.

Mirrors the fields of HF `Qwen2_5_VLConfig` (HF:models/qwen2_5_vl/configuration_qwen2_5_vl.py:32-207) that the
hot path needs; `from_hf_json` accepts both the 4.49-era flat `config.json` (text fields at top level,
`rope_scaling.mrope_section`) and the 5.x nested one (`text_config`, `rope_parameters`).
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, field


@dataclass
class VisionConfig:
    depth: int = 32
    hidden_size: int = 1280
    num_heads: int = 16
    intermediate_size: int = 3420
    out_hidden_size: int = 2048
    patch_size: int = 14
    temporal_patch_size: int = 2
    spatial_merge_size: int = 2
    window_size: int = 112
    in_channels: int = 3
    fullatt_block_indexes: tuple = (7, 15, 23, 31)


@dataclass
class TextConfig:
    hidden_size: int = 2048
    num_hidden_layers: int = 36
    num_attention_heads: int = 16
    num_key_value_heads: int = 2
    intermediate_size: int = 11008
    vocab_size: int = 151936
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1000000.0
    mrope_section: tuple = (16, 24, 24)
    tie_word_embeddings: bool = True


@dataclass
class ModelConfig:
    vision: VisionConfig = field(default_factory=VisionConfig)
    text: TextConfig = field(default_factory=TextConfig)
    image_token_id: int = 151655
    vision_start_token_id: int = 151652
    vision_end_token_id: int = 151653
    eos_token_ids: tuple = (151645, 151643)
    pad_token_id: int = 151643
    name: str = "zoomearth-3b"

    @property
    def head_dim(self) -> int:
        return self.text.hidden_size // self.text.num_attention_heads

    # -------- presets
    @staticmethod
    def zoomearth_3b() -> "ModelConfig":
        """Qwen2.5-VL-3B shape (parameter count 3,754,622,976)."""
        return ModelConfig()

    @staticmethod
    def qwen25vl_7b() -> "ModelConfig":
        """Qwen2.5-VL-7B shape (BASELINE.json configs[4]; dims [upstream, unverified] per SURVEY.md 8d)."""
        return ModelConfig(
            vision=VisionConfig(out_hidden_size=3584),
            text=TextConfig(hidden_size=3584, num_hidden_layers=28, num_attention_heads=28, num_key_value_heads=4,
                            intermediate_size=18944, vocab_size=152064, tie_word_embeddings=False),
            name="qwen2.5-vl-7b")

    @staticmethod
    def tiny() -> "ModelConfig":
        """Parity-fixture config (same as oracle.qwen25vl.tiny_config)."""
        return ModelConfig(
            vision=VisionConfig(depth=4, hidden_size=160, num_heads=2, intermediate_size=220, out_hidden_size=512,
                                fullatt_block_indexes=(1, 3)),
            text=TextConfig(hidden_size=512, num_hidden_layers=3, num_attention_heads=4, num_key_value_heads=2,
                            intermediate_size=1376, vocab_size=2048, tie_word_embeddings=True),
            image_token_id=2005, vision_start_token_id=2002, vision_end_token_id=2003,
            eos_token_ids=(2045, 2043), pad_token_id=2043, name="tiny")

    @staticmethod
    def heads() -> "ModelConfig":
        """The 3B model's head structure at small depth (same as oracle.qwen25vl.heads_config; tests/golden/heads_chain.npz)."""
        return ModelConfig(
            vision=VisionConfig(depth=2, hidden_size=1280, num_heads=16, intermediate_size=220, out_hidden_size=2048,
                                fullatt_block_indexes=(1,)),
            text=TextConfig(hidden_size=2048, num_hidden_layers=2, num_attention_heads=16, num_key_value_heads=2,
                            intermediate_size=1376, vocab_size=2048, tie_word_embeddings=True),
            image_token_id=2005, vision_start_token_id=2002, vision_end_token_id=2003,
            eos_token_ids=(2045, 2043), pad_token_id=2043, name="heads")

    # -------- HF config.json
    @staticmethod
    def from_hf_dict(d: dict) -> "ModelConfig":
        vd = d.get("vision_config", {}) or {}
        td = d.get("text_config") or d
        rope = td.get("rope_parameters") or td.get("rope_scaling") or d.get("rope_scaling") or {}
        v = VisionConfig(
            depth=vd.get("depth", 32), hidden_size=vd.get("hidden_size", 1280), num_heads=vd.get("num_heads", 16),
            intermediate_size=vd.get("intermediate_size", 3420),
            out_hidden_size=vd.get("out_hidden_size", td.get("hidden_size", 2048)),
            patch_size=vd.get("patch_size", vd.get("spatial_patch_size", 14)),
            temporal_patch_size=vd.get("temporal_patch_size", 2), spatial_merge_size=vd.get("spatial_merge_size", 2),
            window_size=vd.get("window_size", 112), in_channels=vd.get("in_channels", vd.get("in_chans", 3)),
            fullatt_block_indexes=tuple(vd.get("fullatt_block_indexes", (7, 15, 23, 31))))
        t = TextConfig(
            hidden_size=td["hidden_size"], num_hidden_layers=td["num_hidden_layers"],
            num_attention_heads=td["num_attention_heads"],
            num_key_value_heads=td.get("num_key_value_heads") or td["num_attention_heads"],
            intermediate_size=td["intermediate_size"], vocab_size=td["vocab_size"],
            rms_norm_eps=td.get("rms_norm_eps", 1e-6),
            rope_theta=float(rope.get("rope_theta", td.get("rope_theta", d.get("rope_theta", 1000000.0)))),
            mrope_section=tuple(rope.get("mrope_section", (16, 24, 24))),
            tie_word_embeddings=bool(d.get("tie_word_embeddings", td.get("tie_word_embeddings", False))))
        eos = td.get("eos_token_id", d.get("eos_token_id", 151645))
        eos = tuple(eos) if isinstance(eos, (list, tuple)) else (eos,)
        pad = td.get("pad_token_id", d.get("pad_token_id"))
        return ModelConfig(vision=v, text=t, image_token_id=d.get("image_token_id", 151655),
                           vision_start_token_id=d.get("vision_start_token_id", 151652),
                           vision_end_token_id=d.get("vision_end_token_id", 151653),
                           eos_token_ids=eos, pad_token_id=pad if pad is not None else 151643,
                           name=d.get("_name_or_path", "") or "qwen2_5_vl")

    @staticmethod
    def from_pretrained(path: str) -> "ModelConfig":
        with open(os.path.join(path, "config.json"), encoding="utf-8") as f:
            cfg = ModelConfig.from_hf_dict(json.load(f))
        # generation_config.json carries the eos list / pad id used by generate()
        gpath = os.path.join(path, "generation_config.json")
        if os.path.exists(gpath):
            with open(gpath, encoding="utf-8") as f:
                g = json.load(f)
            eos = g.get("eos_token_id")
            if eos is not None:
                cfg.eos_token_ids = tuple(eos) if isinstance(eos, (list, tuple)) else (eos,)
            if g.get("pad_token_id") is not None:
                cfg.pad_token_id = g["pad_token_id"]
        return cfg
