"""ctypes binding of the C ABI declared in include/chimeralm_hip.h (csrc/libchimeralm_hip.so).

There is no fallback: if the shared library is missing or does not export the full ABI, importing the
engine fails loudly.  Build it with `python -m chimeralm_amd.build` (hipcc, gfx950).
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

# CLM_LIB points development A/B runs at another build of the same ABI; the product default is the in-tree library
LIB_PATH = Path(os.environ.get("CLM_LIB") or Path(__file__).resolve().parent / "csrc" / "libchimeralm_hip.so")

# error codes / enums (mirror include/chimeralm_hip.h)
OK, E_INVALID, E_HIP, E_MISSING, E_UNSUPPORTED, E_STATE = 0, -1, -2, -3, -4, -5
DT_F32, DT_F64, DT_BF16, DT_F16, DT_U8, DT_I32, DT_I64 = range(7)
PREC_F32, PREC_BF16, PREC_F16, PREC_F16C, PREC_F16X3 = 0, 1, 2, 3, 4
BAM_INPUT_SAM, BAM_KEEP_UNPLACED = 1, 2
PRECISIONS = {"fp32": PREC_F32, "f32": PREC_F32, "bf16": PREC_BF16, "fp16": PREC_F16, "f16": PREC_F16, "fp16c": PREC_F16C, "fp16x3": PREC_F16X3}
STAGES = ["embed", "ln1_in_proj", "short_long_conv", "out_proj", "ln2_fc1_gelu", "fc2", "lnf_pool_score",
          "softmax_pool", "head_mlp", "filter", "out_proj_ln2_mlp", "ln2_mlp"]
N_STAGES = len(STAGES)
ABI_VERSION = 5


class ClmConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("d_model", C.c_int32), ("n_layer", C.c_int32), ("d_inner", C.c_int32),
        ("vocab_rows", C.c_int32), ("filter_order", C.c_int32), ("emb_dim", C.c_int32), ("max_seq_len", C.c_int32),
        ("head_hidden", C.c_int32), ("n_classes", C.c_int32), ("ln_eps", C.c_float), ("precision", C.c_int32),
        ("chunk_reads", C.c_int32),
    ]


class ClmFeederConfig(C.Structure):          # include/chimeralm_feed.h: struct clm_feeder_config
    _fields_ = [("struct_size", C.c_int32), ("batch_size", C.c_int32), ("max_tokens", C.c_int32), ("slots", C.c_int32),
                ("rank", C.c_int32), ("world", C.c_int32), ("pad_left", C.c_int32), ("pinned", C.c_int32),
                ("max_reads", C.c_int64), ("inflate_threads", C.c_int32), ("reserved", C.c_int32)]


class ClmFeedBatch(C.Structure):              # include/chimeralm_feed.h: struct clm_feed_batch
    _fields_ = [("slot", C.c_int32), ("n_reads", C.c_int32), ("n_tokens", C.c_int32), ("reserved", C.c_int32),
                ("row_stride", C.c_int64), ("ids", C.c_void_p), ("names", C.c_void_p), ("first_index", C.c_int64)]


# every symbol include/chimeralm_hip.h and include/chimeralm_feed.h declare: name -> (restype, argtypes)
_H = C.c_void_p
SYMBOLS = {
    "clm_abi_version": (C.c_int, []),
    "clm_default_config": (C.c_int, [C.POINTER(ClmConfig)]),
    "clm_create": (C.c_int, [C.POINTER(ClmConfig), C.c_int, C.POINTER(_H)]),
    "clm_load_weight": (C.c_int, [_H, C.c_char_p, C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.c_int]),
    "clm_finalize": (C.c_int, [_H]),
    "clm_reserve": (C.c_int, [_H, C.c_int, C.c_int]),
    "clm_forward": (C.c_int, [_H, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "clm_stage_ids": (C.c_int, [_H, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "clm_forward_staged": (C.c_int, [_H, C.c_int, C.c_void_p, C.c_void_p]),
    "clm_stage_wait": (C.c_int, [_H, C.c_int]),
    "clm_check": (C.c_int, [_H, C.c_void_p]),
    "clm_selfcheck": (C.c_int, [_H, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_float),
                                C.POINTER(C.c_int)]),
    "clm_logit_deviation": (C.c_int, [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.c_int, C.POINTER(C.c_float),
                                      C.POINTER(C.c_int)]),
    "clm_set_fallback": (C.c_int, [_H, C.c_int]),
    "clm_effective_precision": (C.c_int, [_H, C.c_int]),
    "clm_set_short_read_len": (C.c_int, [_H, C.c_int]),
    "clm_set_mlp_compensation": (C.c_int, [_H, C.c_int]),
    "clm_attention_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "clm_tf_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(_H)]),
    "clm_tf_load_weight": (C.c_int, [_H, C.c_char_p, C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.c_int]),
    "clm_tf_finalize": (C.c_int, [_H]),
    "clm_tf_forward": (C.c_int, [_H, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "clm_tf_selfcheck": (C.c_int, [_H, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_float),
                                   C.POINTER(C.c_int)]),
    "clm_tf_set_fallback": (C.c_int, [_H, C.c_int]),
    "clm_tf_debug_fetch": (C.c_int, [_H, C.c_char_p, C.c_void_p, C.c_size_t]),
    "clm_tf_profile_enable": (C.c_int, [_H, C.c_int]),
    "clm_tf_profile_read": (C.c_int, [_H, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int]),
    "clm_tf_last_error": (C.c_char_p, [_H]),
    "clm_tf_destroy": (C.c_int, [_H]),
    "clm_debug_fetch": (C.c_int, [_H, C.c_char_p, C.c_void_p, C.c_size_t]),
    "clm_debug_stop_after": (C.c_int, [_H, C.c_int, C.c_int]),
    "clm_profile_enable": (C.c_int, [_H, C.c_int]),
    "clm_profile_read": (C.c_int, [_H, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int]),
    "clm_profile_stage_name": (C.c_char_p, [C.c_int]),
    "clm_last_error": (C.c_char_p, [_H]),
    "clm_destroy": (C.c_int, [_H]),
    "clm_feeder_default_config": (C.c_int, [C.POINTER(ClmFeederConfig)]),
    "clm_feeder_open": (C.c_int, [C.c_char_p, C.POINTER(ClmFeederConfig), C.POINTER(_H)]),
    "clm_feeder_next": (C.c_int, [_H, C.POINTER(ClmFeedBatch)]),
    "clm_feeder_release": (C.c_int, [_H, C.c_int32]),
    "clm_feeder_stats": (C.c_int, [_H, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                   C.POINTER(C.c_int64)]),
    "clm_feeder_last_error": (C.c_char_p, [_H]),
    "clm_feeder_close": (C.c_int, [_H]),
    "clm_bam_filter": (C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), C.c_int64, C.POINTER(C.c_int64),
                                 C.POINTER(C.c_int64)]),
    "clm_bam_filter_ex": (C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), C.c_int64, C.c_int, C.POINTER(C.c_int64),
                                  C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "clm_bam_sort_index": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_int64)]),
    "clm_bam_last_error": (C.c_char_p, []),
}

_lib = None


def load() -> C.CDLL:
    """Load the engine library and bind every ABI symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} not found: the ChimeraLM MI355X engine has no CPU or PyTorch fallback. "
            "Build it with `python -m chimeralm_amd.build` (needs hipcc, targets gfx950)."
        )
    # The process must hold ONE HIP runtime.  PyTorch-ROCm ships its own libamdhip64; if this library were opened first it
    # would pull in /opt/rocm's copy and device enumeration then fails in whichever runtime comes second
    # ("no such HIP device 0").  Importing torch first makes the dynamic loader resolve our dependency to the copy torch
    # has already mapped, whatever order the caller imports things in.
    import torch  # noqa: F401

    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise ImportError(f"{LIB_PATH} does not export {name}; rebuild with `python -m chimeralm_amd.build --force`") from e
        fn.restype = res
        fn.argtypes = args
    if lib.clm_abi_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH}: ABI version {lib.clm_abi_version()} != {ABI_VERSION}; rebuild the library")
    _lib = lib
    return lib
