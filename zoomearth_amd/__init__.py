"""zoomearth_amd: MI355X-native engine for ZoomEarth's zoom-crop-reason inference path.

Host code (this package) mirrors the reference's call surface -- `src/demo.py`, `src/eval/infer.py`, the
Qwen2.5-VL `processor(...)` / `model.generate(...)` API -- on top of the C ABI of libzoomearth_hip.so
(include/zoomearth.h), which holds the hand-written gfx950 HIP kernels.  There is no CPU fallback.
"""
from .config import ModelConfig, TextConfig, VisionConfig  # noqa: F401

__version__ = "0.1.0"
