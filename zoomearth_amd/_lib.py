"""ctypes binding of libzoomearth_hip.so (the C ABI declared in include/zoomearth.h).

The product path fails loudly when the HIP library is missing: there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ZE_LIB_PATH") or os.path.join(_HERE, "libzoomearth_hip.so")  # ZE_LIB_PATH: A/B builds
CSRC = os.path.join(_HERE, "csrc")

ZE_F32, ZE_F16, ZE_BF16 = 0, 1, 2
ZE_MAX_FULLATT, ZE_MAX_EOS = 16, 4


class ZeConfig(C.Structure):
    _fields_ = [
        ("vit_depth", C.c_int32), ("vit_hidden", C.c_int32), ("vit_heads", C.c_int32),
        ("vit_intermediate", C.c_int32), ("vit_out_hidden", C.c_int32),
        ("patch_size", C.c_int32), ("temporal_patch_size", C.c_int32), ("spatial_merge_size", C.c_int32),
        ("window_size", C.c_int32), ("in_channels", C.c_int32),
        ("n_fullatt", C.c_int32), ("fullatt_block_indexes", C.c_int32 * ZE_MAX_FULLATT),
        ("hidden", C.c_int32), ("layers", C.c_int32), ("heads", C.c_int32), ("kv_heads", C.c_int32),
        ("intermediate", C.c_int32), ("vocab", C.c_int32),
        ("rms_eps", C.c_float), ("rope_theta", C.c_float), ("mrope_section", C.c_int32 * 3),
        ("tie_word_embeddings", C.c_int32),
        ("image_token_id", C.c_int32), ("vision_start_token_id", C.c_int32), ("vision_end_token_id", C.c_int32),
        ("pad_token_id", C.c_int32), ("n_eos", C.c_int32), ("eos_token_ids", C.c_int32 * ZE_MAX_EOS),
        ("max_seqs", C.c_int32), ("max_ctx", C.c_int32), ("max_patches", C.c_int32), ("max_tile_side", C.c_int32),
        ("max_prefill_rows", C.c_int32),
    ]


class ZeGenParams(C.Structure):
    _fields_ = [("max_new_tokens", C.c_int32), ("repetition_penalty", C.c_float), ("ignore_eos", C.c_int32),
                ("use_graph", C.c_int32), ("sync_every", C.c_int32), ("do_sample", C.c_int32),
                ("temperature", C.c_float), ("seed", C.c_uint64)]


class ZoomEarthError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libzoomearth_hip error {code}: {msg}")
        self.code = code


def kernel_sources_sha16() -> str:
    """sha256 (first 16 hex digits) over the kernel and host sources of the library (csrc/*.hip, *.h, *.cpp, the Makefile, include/
    zoomearth.h), in name order: what a committed profile records so that a figure quoted from it can be checked against the tree
    that quotes it (bench.py `roofline.traffic_source`; the .git directory does not travel to the GPU box)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.cpp")) +
                   [os.path.join(CSRC, "Makefile"), os.path.join(_HERE, "..", "include", "zoomearth.h")])
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def build(force: bool = False, jobs: int = 8) -> str:
    """Compile libzoomearth_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], check=True, stdout=subprocess.DEVNULL)
    subprocess.run(["make", "-C", CSRC, f"-j{jobs}"], check=True, stdout=subprocess.DEVNULL)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("build finished but libzoomearth_hip.so is missing")
    return LIB_PATH


_lib = None

_P = C.c_void_p
_SIGS = {
    "ze_engine_create": (C.c_int, [C.POINTER(ZeConfig), C.c_int, C.POINTER(_P)]),
    "ze_engine_destroy": (C.c_int, [_P]),
    "ze_last_error": (C.c_char_p, [_P]),
    "ze_version": (C.c_int, []),
    "ze_sync": (C.c_int, [_P, _P]),
    "ze_load_weight": (C.c_int, [_P, C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_int64), _P]),
    "ze_weights_fill_synthetic": (C.c_int, [_P, C.c_uint64, C.c_float, C.c_float, C.c_float, C.c_float]),
    "ze_weights_missing": (C.c_int, [_P]),
    "ze_weights_arena": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "ze_weights_invalidate": (C.c_int, [_P]),
    "ze_set_decode_regime": (C.c_int, [_P, C.c_int]),
    "ze_weights_broadcast": (C.c_int, [_P, _P, C.c_int, _P]),
    "ze_tile_upload": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P]),
    "ze_op_crop_resize": (C.c_int, [_P, _P, C.c_int, C.c_int, C.POINTER(C.c_int32), _P, C.c_int, C.c_int, _P]),
    "ze_smart_resize": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.POINTER(C.c_int),
                                  C.POINTER(C.c_int)]),
    "ze_op_patchify": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P]),
    "ze_preprocess_image": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int64, C.c_int64, _P, C.c_int64,
                                      C.POINTER(C.c_int32), _P]),
    "ze_vision_window_index": (C.c_int, [C.POINTER(ZeConfig), C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_int64),
                                         C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_int)]),
    "ze_rope_index": (C.c_int, [C.POINTER(ZeConfig), C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_int32), C.c_int,
                                C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "ze_vit_forward": (C.c_int, [_P, _P, C.POINTER(C.c_int32), C.c_int, _P, _P]),
    "ze_seq_reset": (C.c_int, [_P, C.c_int, _P]),
    "ze_seq_retire": (C.c_int, [_P, C.c_int, _P]),
    "ze_seq_set_split": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "ze_seq_prefix_hint": (C.c_int, [_P, C.c_int]),
    "ze_seq_set_prefix_hint": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P]),
    "ze_seq_truncate": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "ze_seq_len": (C.c_int, [_P, C.c_int]),
    "ze_seq_copy_prefix": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P]),
    "ze_prefill": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int32), C.c_int, _P, C.c_int, C.POINTER(C.c_int32), C.c_int,
                             _P, _P]),
    "ze_score": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int32), C.c_int, _P, C.c_int, C.POINTER(C.c_int32), C.c_int,
                           _P, _P]),
    "ze_op_token_logprob": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "ze_prefill_batch": (C.c_int, [_P, C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _P,
                                   C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), _P]),
    "ze_decode_step": (C.c_int, [_P, C.c_int, C.c_int, _P, _P]),
    "ze_generate": (C.c_int, [_P, C.c_int, C.POINTER(ZeGenParams), C.POINTER(C.c_int32), C.POINTER(C.c_int), _P]),
    "ze_decode_batch": (C.c_int, [_P, C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_int32), _P, _P]),
    "ze_generate_batch": (C.c_int, [_P, C.POINTER(C.c_int32), C.c_int, C.POINTER(ZeGenParams), C.POINTER(C.c_int32),
                                   C.POINTER(C.c_int32), _P]),
    "ze_chain_begin": (C.c_int, [_P, C.c_int, C.POINTER(ZeGenParams), C.c_int, _P]),
    "ze_decode_burst": (C.c_int, [_P, C.POINTER(C.c_int32), C.c_int, C.c_int, C.POINTER(ZeGenParams),
                                 C.POINTER(C.c_int32), C.POINTER(C.c_int32), _P]),
    "ze_decode_burst_begin": (C.c_int, [_P, C.POINTER(C.c_int32), C.c_int, C.c_int, C.POINTER(ZeGenParams), _P]),
    "ze_decode_burst_end": (C.c_int, [_P, C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _P]),
    "ze_chain_tokens": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_int), _P]),
    "ze_seq_mark_seen": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int32), C.c_int, _P]),
    "ze_seq_mark_seen_batch": (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_int32), _P]),
    "ze_chain_tokens_batch": (C.c_int, [_P, C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_int32), _P]),
    "ze_op_sample_greedy": (C.c_int, [_P, C.c_int, _P, C.c_float, C.POINTER(C.c_int32), _P]),
    "ze_op_sample_temperature": (C.c_int, [_P, C.c_int, _P, C.c_float, C.c_float, C.c_uint64, C.c_int,
                                           C.POINTER(C.c_int32), _P]),
    "ze_weights_quantize_fp8": (C.c_int, [_P, _P]),
    "ze_set_fp8_activations": (C.c_int, [_P, C.c_int]),
    "ze_op_quantize_fp8": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P, _P]),
    "ze_op_linear_mx": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "ze_op_linear": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "ze_op_rmsnorm": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_float, _P]),
    "ze_op_attention": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32),
                                  C.c_int, C.c_int, _P]),
    "ze_op_window_gather": (C.c_int, [_P, _P, C.POINTER(C.c_int32), C.c_int, _P, _P]),
    "ze_op_window_scatter": (C.c_int, [_P, _P, C.c_int, C.POINTER(C.c_int32), C.c_int, _P, _P]),
    "ze_op_vision_rope": (C.c_int, [_P, _P, C.POINTER(C.c_int32), C.c_int, C.c_int, _P]),
    "ze_op_embed_scatter": (C.c_int, [_P, C.POINTER(C.c_int32), C.c_int, _P, C.c_int, _P, _P]),
    "ze_op_mrope_kv": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int, C.POINTER(C.c_int32), C.c_int, _P]),
    "ze_op_rope_kv_decode": (C.c_int, [_P, C.POINTER(C.c_int32), C.c_int, C.c_int, _P, _P]),
    "ze_op_attn_decode": (C.c_int, [_P, C.POINTER(C.c_int32), C.c_int, C.c_int, _P, _P, _P]),
    "ze_op_kv_read": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "ze_op_numeric_helpers": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, _P]),
    "ze_profile_decode_kernel": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_double), _P]),
    "ze_profile_batch_kernel": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_double), _P]),
    "ze_profile_prefill_kernel": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_double), _P]),
    "ze_profile_prefill_layer": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_double), _P]),
    "ze_phase_timers": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "ze_tune": (C.c_int, [C.c_int, C.c_int]),
}
EXPORTS = tuple(_SIGS)


def lib():
    """Load the shared library (after torch, so both share torch's bundled HIP runtime)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C zoomearth_amd/csrc`). zoomearth_amd has no CPU fallback.")
    try:
        import torch  # noqa: F401  (loads libamdhip64.so.7 first; ours resolves to the same runtime)
    except Exception:  # pragma: no cover - torch is plumbing; the C ABI itself does not need it
        pass
    handle = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in _SIGS.items():
        fn = getattr(handle, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = handle
    return _lib


def check(code: int, engine=None):
    if code < 0:
        msg = lib().ze_last_error(engine)
        raise ZoomEarthError(code, msg.decode("utf-8", "replace") if msg else "")
    return code
