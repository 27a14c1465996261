"""Shared helpers for the -m gpu parity tests (all compute goes through the C ABI)."""
import numpy as np
import pytest
import torch

from oracle import qwen25vl as Q


def bf16_bits_to_f32(t: torch.Tensor) -> np.ndarray:
    return t.float().cpu().numpy()


def to_dev_bf16(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to("cuda").to(torch.bfloat16).contiguous()


def oracle_cfg_to_model_cfg():
    from zoomearth_amd.config import ModelConfig
    return ModelConfig.tiny()


@pytest.fixture(scope="session")
def tiny_engine():
    from zoomearth_amd.engine import Engine
    e = Engine(oracle_cfg_to_model_cfg(), device=0, max_seqs=3, max_ctx=1024, max_patches=4096, max_tile_side=5200)
    yield e
    e.close()


CHAIN_W = dict(seed=1, std=0.02, matrix_gain=4.0, bias_std=0.02, norm_jitter=0.1)


@pytest.fixture(scope="session")
def tiny_weights():
    return Q.synthetic_weights(Q.tiny_config(), **CHAIN_W)
