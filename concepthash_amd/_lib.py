"""ctypes binding of libconcepthash_hip.so (the C-ABI declared in include/concepthash_hip.h; test / bench taps in
include/concepthash_hip_debug.h).

The product path has NO fallback: if the library is missing, or an entry point is absent, importing callers get a
loud ``RuntimeError`` -- never a silent PyTorch/CPU path.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import (POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_uint32,
                    c_uint64, c_ulonglong, c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libconcepthash_hip.so")
ABI_VERSION = 3


class ModelConfig(Structure):
    _fields_ = [(n, c_int32) for n in ("image_size", "patch", "dim", "layers", "heads", "ffn", "adapter_dim", "ncontext",
                                       "nbit", "nclass", "proj_dim", "center_dim", "upt_heads", "act", "max_batch")] + \
               [("ln_eps", c_float), ("bn_eps", c_float)]


class ImageDesc(Structure):
    _fields_ = [("src_offset", c_int64), ("tmp_offset", c_int64)] + \
               [(n, c_int32) for n in ("h", "w", "nh", "nw", "top", "left", "row0", "nrows", "stride", "flip")]


class JpegDesc(Structure):
    _fields_ = [("coef_offset", c_int64), ("pix_offset", c_int64), ("plane_offset", c_int64)] + \
               [(n, c_int32) for n in ("width", "height", "ncomp", "hs", "vs", "mcu_w", "mcu_h", "status", "nblocks", "reserved")] + \
               [("quant", (ctypes.c_uint16 * 64) * 3)]


class Tensor(Structure):
    _fields_ = [("name", c_char_p), ("data", POINTER(c_float)), ("numel", c_int64)]


# name -> (restype, argtypes); the single source of truth for the symbol-export test
SIGNATURES = {
    "ch_abi_version": (c_int, []),
    "ch_last_error": (c_char_p, []),
    "ch_model_create": (c_int, [POINTER(ModelConfig), POINTER(Tensor), c_int32, POINTER(c_void_p)]),
    "ch_model_destroy": (None, [c_void_p]),
    "ch_model_device_bytes": (c_size_t, [c_void_p]),
    "ch_model_flops_per_image": (c_double, [c_void_p]),
    "ch_model_set_option": (c_int, [c_void_p, c_char_p, c_int64]),
    "ch_model_get_option": (c_int, [c_void_p, c_char_p, POINTER(c_int64)]),
    "ch_model_profile_begin": (c_int, [c_void_p, c_int32]),
    "ch_model_profile_end": (c_int, [c_void_p, c_int32, POINTER(c_double), POINTER(c_int64), POINTER(c_double)]),
    "ch_debug_gemm": (c_int, [c_int32, c_void_p, c_int64, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p,
                              c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_void_p]),
    "ch_debug_gemm_ln": (c_int, [c_int32, c_void_p, c_int64, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p,
                                 c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                                 c_void_p]),
    "ch_debug_set_gemm_variant": (None, [c_int32]),
    "ch_debug_gemm_dispatch_count": (c_int64, [c_int32]),
    "ch_debug_experiments_built": (c_int32, []),
    "ch_debug_set_gemm_splitk": (None, [c_int32]),
    "ch_debug_copy_buffer": (c_int, [c_void_p, c_int32, c_void_p, c_int64, c_void_p]),
    "ch_debug_adapter": (c_int, [c_void_p] * 2 + [c_int32] * 3 + [c_void_p] * 10 + [c_int32, c_void_p]),
    "ch_debug_attention": (c_int, [c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "ch_encode": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                          c_void_p, c_void_p, c_void_p, c_int32, c_void_p]),
    "ch_encode_hidden": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "ch_preprocess": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, POINTER(c_float), POINTER(c_float), c_void_p,
                              c_int32, c_void_p, c_void_p]),
    "ch_preprocess_max_taps": (c_int32, []),
    "ch_jpeg_plan": (c_int, [c_void_p, c_void_p, c_int32, c_void_p, POINTER(c_int64), POINTER(c_int64), POINTER(c_int64)]),
    "ch_jpeg_entropy_decode": (c_int, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int32]),
    "ch_jpeg_plan_packed": (c_int, [c_void_p, c_void_p, c_int32, c_void_p, POINTER(c_int64), POINTER(c_int64), POINTER(c_int64)]),
    "ch_jpeg_entropy_decode_packed": (c_int, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int32]),
    "ch_io_file_sizes": (c_int, [c_void_p, c_int32, c_void_p]),
    "ch_io_read_files": (c_int, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int32]),
    "ch_jpeg_reconstruct": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p]),
    "ch_pack_sign": (c_int, [c_void_p, c_int64, c_int32, c_float, c_void_p, c_void_p]),
    "ch_hamming_dist": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int32, c_void_p, c_void_p]),
    "ch_hamming_topk_workspace": (c_size_t, [c_int64, c_int64, c_int32, c_int32]),
    "ch_hamming_topk": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int32, c_int32, c_int64, c_void_p, c_void_p,
                                c_void_p, c_size_t, c_void_p]),
    "ch_topk_merge": (c_int, [c_void_p, c_void_p, c_int32, c_int64, c_int32, c_void_p, c_void_p, c_void_p]),
    "ch_hamming_hist": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_int32, c_int32,
                                c_void_p, c_void_p]),
    "ch_hamming_ap": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_int32, c_int32,
                              c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ch_hamming_ap_multi": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_int32, c_int32,
                                    c_void_p, POINTER(c_int64), c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ch_hamming_rec_workgroups": (c_size_t, [c_int64, c_int64, c_int32, c_int32]),
    "ch_hamming_rec_block": (c_int32, [c_int32]),
    "ch_hamming_hist_rec": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_int32, c_int32,
                                    c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p]),
    "ch_hamming_ap_rec": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_int32, c_int32,
                                  c_void_p, c_void_p, c_int32, c_void_p, c_void_p, POINTER(c_int64), c_int32, c_void_p, c_void_p,
                                  c_void_p, c_void_p]),
    "ch_adapter_arena_numel": (c_int64, [c_void_p]),
    "ch_trainer_create": (c_int, [c_void_p, c_int32, c_void_p, c_void_p, POINTER(c_void_p)]),
    "ch_trainer_destroy": (None, [c_void_p]),
    "ch_trainer_bytes": (c_int64, [c_void_p]),
    "ch_trainer_refresh": (c_int, [c_void_p, c_void_p]),
    "ch_train_forward": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p]),
    "ch_train_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ch_sgd_step": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float, c_int32, c_int32, c_void_p]),
    "ch_debug_attention_bwd": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_int32, c_void_p]),
    "ch_debug_wgrad": (c_int, [c_void_p, c_int32, c_void_p, c_int32, c_int64, c_int64, c_int32, c_int32, c_void_p, c_void_p]),
    "ch_debug_ln_bwd": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ch_debug_act": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_int32, c_void_p, c_void_p]),
    "ch_hamming_hist_prefix": (c_int, [c_void_p, c_int32, c_int64, c_int32, c_void_p, c_void_p, c_void_p]),
    "ch_debug_set_hamming_scalar_loads": (None, [c_int32]),
}

# ch_model_set_option keys (include/concepthash_hip.h) and the DEBUG environment overrides the Python wrapper maps onto them when a
# handle is created.  The library itself reads no environment variable; code that needs a setting passes `options=` / set_option().
OPTION_KEYS = ("streams", "chain_auto", "ln_fold", "prune_last", "pp_min_k", "small_kernel", "serpentine", "pp_sched", "fused_adapter", "resid_nt",
               "nt_out", "group_n", "splitk", "gemm_rows", "wide_kernel", "graph_max_batch", "train_chains", "train_chain_min_rows", "train_prune_last",
               "train_batched_grads")
_ENV_OVERRIDES = {  # env name -> (option key, value map)
    "CH_STREAMS": ("streams", int), "CH_LN_FOLD": ("ln_fold", int), "CH_PRUNE_LAST": ("prune_last", int),
    "CH_GEMM_PP_MIN_K": ("pp_min_k", int), "CH_GEMM_SMALL": ("small_kernel", int), "CH_SERPENTINE": ("serpentine", int),
    "CH_GEMM_PP_SCHED": ("pp_sched", int), "CH_FUSED_ADAPTER": ("fused_adapter", int),
    "CH_RESID_NT": ("resid_nt", lambda v: 1 if int(v) else -1), "CH_NT_OUT": ("nt_out", lambda v: 1 if int(v) else -1),
    "CH_GEMM_GROUP_N": ("group_n", int), "CH_GEMM_SPLITK": ("splitk", int), "CH_GEMM_ROWS": ("gemm_rows", int), "CH_GEMM_WIDE": ("wide_kernel", int), "CH_GRAPH_MAX_BATCH": ("graph_max_batch", int),
    "CH_CHAIN_AUTO": ("chain_auto", int), "CH_TRAIN_STREAMS": ("train_chains", int), "CH_TRAIN_CHAIN_MIN_ROWS": ("train_chain_min_rows", int),
    "CH_TRAIN_PRUNE_LAST": ("train_prune_last", int),
}


def env_option_overrides() -> dict:
    """Debug overrides: CH_* environment variables of earlier rounds, read HERE (never in the library) when a handle is created."""
    out = {}
    for env, (key, conv) in _ENV_OVERRIDES.items():
        if env in os.environ:
            out[key] = conv(os.environ[env])
    return out

# launch-profiler categories, in the order of the CH_CAT_* enum; "*_pruned" = the final layer's launches on the compact head rows
CATEGORIES = ("im2col", "gemm_patch", "rowops", "gemm_qkv", "attention", "gemm_out", "gemm_down", "gemm_up", "gemm_fc1",
              "gemm_fc2", "head", "adapter_fused", "attention_pruned", "gemm_out_pruned", "gemm_down_pruned", "gemm_up_pruned",
              "gemm_fc1_pruned", "gemm_fc2_pruned", "end")

_lib = None


def load() -> ctypes.CDLL:
    """Load the HIP library, or raise.  There is deliberately no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the ConceptHash MI355X path needs the HIP extension. "
            "Build it with `python -m concepthash_amd.build` (hipcc, --offload-arch=gfx950). There is no CPU fallback.")
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # e.g. libamdhip64 not found
        raise RuntimeError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RuntimeError(f"{LIB_PATH} does not export {name}; rebuild it") from e
        fn.restype = res
        fn.argtypes = args
    if lib.ch_abi_version() != ABI_VERSION:
        raise RuntimeError(f"ABI mismatch: library {lib.ch_abi_version()} vs binding {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def check(status: int, what: str) -> None:
    if status != 0:
        msg = load().ch_last_error()
        raise RuntimeError(f"{what} failed (status {status}): {msg.decode() if msg else 'unknown error'}")


def stream_ptr(stream=None) -> c_void_p:
    """hipStream_t of a torch stream (default: torch's current stream)."""
    import torch
    s = stream if stream is not None else torch.cuda.current_stream()
    return c_void_p(s.cuda_stream)


def ptr(t) -> c_void_p:
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)
