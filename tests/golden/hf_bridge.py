"""Build the installed `transformers` Qwen2.5-VL model from an oracle Config + weight dict.

Used ONLY by tests/golden/make_fixtures.py in the build container (transformers is third-party
code; this file does not travel as a dependency of any test that runs on the GPU box).
"""
from __future__ import annotations

import numpy as np
import torch


def hf_config(cfg):
    from transformers import Qwen2_5_VLConfig

    v, t = cfg.vision, cfg.text
    return Qwen2_5_VLConfig(
        vision_config=dict(depth=v.depth, hidden_size=v.hidden_size, num_heads=v.num_heads,
                           intermediate_size=v.intermediate_size, out_hidden_size=v.out_hidden_size,
                           patch_size=v.patch_size, temporal_patch_size=v.temporal_patch_size,
                           spatial_merge_size=v.spatial_merge_size, window_size=v.window_size,
                           in_channels=v.in_channels, fullatt_block_indexes=list(v.fullatt_block_indexes)),
        text_config=dict(hidden_size=t.hidden_size, num_hidden_layers=t.num_hidden_layers,
                         num_attention_heads=t.num_attention_heads, num_key_value_heads=t.num_key_value_heads,
                         intermediate_size=t.intermediate_size, vocab_size=t.vocab_size,
                         rms_norm_eps=t.rms_norm_eps, max_position_embeddings=32768,
                         rope_parameters=dict(rope_type="default", rope_theta=t.rope_theta,
                                              mrope_section=list(t.mrope_section)),
                         eos_token_id=list(cfg.eos_token_ids), pad_token_id=cfg.pad_token_id,
                         bos_token_id=None),
        image_token_id=cfg.image_token_id, video_token_id=cfg.image_token_id + 1,
        vision_start_token_id=cfg.vision_start_token_id, vision_end_token_id=cfg.vision_end_token_id,
        tie_word_embeddings=t.tie_word_embeddings,
    )


def hf_model(cfg, weights: dict, dtype=torch.float32, attn="sdpa"):
    from transformers import Qwen2_5_VLForConditionalGeneration

    hc = hf_config(cfg)
    hc._attn_implementation = attn
    with torch.device("cpu"):
        model = Qwen2_5_VLForConditionalGeneration(hc)
    sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in weights.items()}
    if "lm_head.weight" not in sd:
        sd["lm_head.weight"] = sd["model.language_model.embed_tokens.weight"]
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("inv_freq" in m for m in missing), missing
    model = model.to(dtype).eval()
    model.generation_config.eos_token_id = list(cfg.eos_token_ids)
    model.generation_config.pad_token_id = cfg.pad_token_id
    return model
