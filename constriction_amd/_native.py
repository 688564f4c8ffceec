"""ctypes binding of the C ABI declared in include/constriction_amd.h.

There is NO CPU fallback: if the shared library is missing or no gfx950 device is visible the
product path raises (BackendUnavailable).  Nothing here imports the test oracle."""
from __future__ import annotations

import ctypes as C
from pathlib import Path

PKG = Path(__file__).resolve().parent
LIB_PATH = PKG / "lib" / "libconstriction_amd.so"

CST_OK = 0
CST_ERR_INVALID_ARGUMENT = -1
CST_ERR_HIP = -2
CST_ERR_NO_DEVICE = -3
CST_ERR_MODEL = -4
CST_ERR_OUT_OF_MEMORY = -5

STREAM_OK = 0
STREAM_IMPOSSIBLE_SYMBOL = 1
STREAM_CAPACITY = 2
STREAM_INVALID_DATA = 3
STREAM_OUT_OF_DATA = 4

LAYOUT_STREAM_MAJOR = 0
LAYOUT_SYMBOL_MAJOR = 1

FAMILY_LAPLACE, FAMILY_CAUCHY, FAMILY_BINOMIAL = 1, 2, 3

FLAG_NONE = 0
FLAG_RAW_STATE = 1
FLAG_COLD_WORDS = 2
FLAG_PACKED_W16 = 4

CODER_ANS, CODER_RANGE = 0, 1


class BackendUnavailable(RuntimeError):
    """The HIP extension (or a GPU) is missing.  The product path never falls back to the CPU."""


class BackendError(RuntimeError):
    pass


class RangeState(C.Structure):
    """cst_range_state"""
    _fields_ = [("lower", C.c_uint64), ("range", C.c_uint64), ("point", C.c_uint64), ("inverted_n", C.c_uint32),
                ("inverted_first", C.c_uint32), ("position", C.c_uint64)]


class CoderConfig(C.Structure):
    _fields_ = [("word_bits", C.c_int32), ("state_bits", C.c_int32), ("precision", C.c_int32)]


_vp, _z, _i32, _u32, _f64 = C.c_void_p, C.c_size_t, C.c_int32, C.c_uint32, C.c_double

# name -> (restype, argtypes); one entry per symbol of include/constriction_amd.h
SIGNATURES = {
    "cst_abi_version": (_i32, []),
    "cst_device_count": (_i32, []),
    "cst_last_hip_error": (C.c_char_p, []),
    "cst_last_kernel_name": (C.c_char_p, []),
    "cst_ans_max_words": (_z, [_z, CoderConfig]),
    "cst_range_max_words": (_z, [_z, CoderConfig]),
    "cst_model_create_table": (_i32, [_i32, _i32, _i32, _vp, C.POINTER(_vp)]),
    "cst_model_create_gaussian": (_i32, [_i32, _i32, _i32, _f64, _f64, _vp, C.POINTER(_vp)]),
    "cst_model_create_gaussian_per_stream": (_i32, [_i32, _i32, _i32, _vp, _vp, _z, _vp, C.POINTER(_vp)]),
    "cst_model_create_table_noncontiguous": (_i32, [_i32, _i32, _vp, _vp, C.POINTER(_vp)]),
    "cst_symbols_to_indices": (_i32, [_vp, _vp, _z, _vp, _vp]),
    "cst_indices_to_symbols": (_i32, [_vp, _vp, _z, _vp, _vp]),
    "cst_model_destroy": (_i32, [_vp]),
    "cst_model_precision": (_i32, [_vp]),
    "cst_model_min_symbol": (_i32, [_vp]),
    "cst_model_n_symbols": (_i32, [_vp]),
    "cst_model_n_tables": (_z, [_vp]),
    "cst_model_get_cdf": (_i32, [_vp, _z, _vp, _vp]),
    "cst_model_copy_cdfs": (_i32, [_vp, _z, _z, _vp, _vp]),
    "cst_ans_encode_batch": (_i32, [_vp, CoderConfig, _vp, _z, _z, _i32, _vp, _z, _vp, _vp, _vp, _u32, _vp]),
    "cst_ans_decode_batch": (_i32, [_vp, CoderConfig, _vp, _vp, _z, _z, _vp, _vp, _z, _z, _i32, _vp, _vp, _vp, _u32, _vp]),
    "cst_ans_encode_batch_ckpt": (_i32, [_vp, CoderConfig, _vp, _z, _z, _i32, _vp, _z, _vp, _z, _vp, _vp, _vp, _vp]),
    "cst_ckpt_scratch_bytes": (_z, [_z, _z, _z]),
    "cst_ckpt_status_per_stream": (_i32, [_vp, _z, _z, _vp, _vp]),
    "cst_ans_decode_batch_ckpt": (_i32, [_vp, CoderConfig, _vp, _vp, _z, _z, _z, _vp, _vp, _vp, _z, _z, _vp, _vp, _vp]),
    "cst_ans_encode_batch_ckpt_sym": (_i32, [_vp, CoderConfig, _vp, _i32, _z, _z, _i32, _vp, _z, _vp, _z, _vp, _vp, _vp, _vp, _vp]),
    "cst_ckpt_sym_scratch_bytes": (_z, [_z, _z, _z, _i32]),
    "cst_ans_decode_batch_ckpt_sym": (_i32, [_vp, CoderConfig, _vp, _vp, _z, _z, _z, _vp, _vp, _vp, _i32, _z, _z, _vp, _vp, _vp]),
    "cst_compact_words16": (_i32, [_vp, _z, _vp, _z, _vp, _vp, _z, _vp, _vp]),
    "cst_ans_encode_batch_ckpt_packed16": (_i32, [_vp, CoderConfig, _vp, _z, _z, _vp, _z, _vp, _z, _vp, _vp, _vp, _vp]),
    "cst_symbols_widen": (_i32, [_vp, _i32, _z, _vp, _vp]),
    "cst_symbols_narrow": (_i32, [_vp, _z, _vp, _i32, _vp]),
    "cst_symbols_scratch_bytes": (_z, [_z, _z, _i32]),
    "cst_ans_encode_batch_sym": (_i32, [_vp, CoderConfig, _vp, _i32, _z, _z, _i32, _vp, _z, _vp, _vp, _vp, _u32, _vp, _vp]),
    "cst_ans_decode_batch_sym": (_i32, [_vp, CoderConfig, _vp, _vp, _z, _z, _vp, _vp, _i32, _z, _z, _i32, _vp, _vp, _vp, _u32, _vp, _vp]),
    "cst_ans_encode_gaussian_batch_ckpt": (_i32, [CoderConfig, _i32, _i32, _vp, _vp, _vp, _z, _z, _i32, _vp, _z, _vp, _z, _vp, _vp, _vp, _vp]),
    "cst_ans_decode_gaussian_batch_ckpt": (_i32, [CoderConfig, _i32, _i32, _vp, _vp, _z, _z, _z, _vp, _vp, _vp, _vp, _vp, _z, _z, _vp, _vp, _vp]),
    "cst_range_encode_gaussian_batch_ckpt": (_i32, [CoderConfig, _i32, _i32, _vp, _vp, _vp, _z, _z, _i32, _vp, _z, _vp, _z, _vp, _vp, _vp, _vp, _vp]),
    "cst_range_gaussian_ckpt_scratch_bytes": (_z, [_z, _z, _z]),
    "cst_range_decode_gaussian_batch_ckpt": (_i32, [CoderConfig, _i32, _i32, _vp, _vp, _z, _z, _vp, _z, _vp, _vp, _vp, _vp, _vp, _vp, _z, _z, _vp, _vp, _vp]),
    "cst_range_encode_batch_ckpt": (_i32, [_vp, CoderConfig, _vp, _z, _z, _i32, _vp, _z, _vp, _z, _vp, _vp, _vp, _vp, _vp]),
    "cst_range_ckpt_scratch_bytes": (_z, [_z, _z, _z]),
    "cst_range_sym_scratch_bytes": (_z, [_z, _z, _z, _i32]),
    "cst_range_encode_batch_sym": (_i32, [_vp, CoderConfig, _vp, _i32, _z, _z, _i32, _vp, _z, _vp, _vp, _vp, _u32, _vp, _vp]),
    "cst_range_decode_batch_sym": (_i32, [_vp, CoderConfig, _vp, _vp, _z, _z, _vp, _vp, _i32, _z, _z, _i32, _vp, _vp, _u32, _vp, _vp]),
    "cst_range_encode_batch_ckpt_sym": (_i32, [_vp, CoderConfig, _vp, _i32, _z, _z, _i32, _vp, _z, _vp, _z, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cst_range_decode_batch_ckpt_sym": (_i32, [_vp, CoderConfig, _vp, _vp, _z, _z, _vp, _z, _vp, _vp, _vp, _vp, _i32, _z, _z, _vp, _vp, _vp]),
    "cst_jump_points_auto": (_z, [_vp, CoderConfig, _i32, _i32, _vp, _z, _z, _i32, _vp, _z]),
    "cst_jump_points_auto_gaussian": (_z, [CoderConfig, _i32, _z, _z, _i32]),
    "cst_debug_reload_knobs": (None, []),
    "cst_range_decode_batch_ckpt": (_i32, [_vp, CoderConfig, _vp, _vp, _z, _z, _vp, _z, _vp, _vp, _vp, _vp, _z, _z, _vp, _vp, _vp]),
    "cst_ans_encode_ragged": (_i32, [_vp, CoderConfig, _vp, _vp, _z, _vp, _vp, _z, _vp, _vp, _vp]),
    "cst_ans_decode_ragged": (_i32, [_vp, CoderConfig, _vp, _vp, _z, _z, _vp, _vp, _vp, _z, _vp, _vp]),
    "cst_ans_count_until": (_i32, [_vp, CoderConfig, _vp, _vp, _z, _z, _vp, _z, _i32, _z, _vp, _vp, _vp]),
    "cst_ans_encode_ragged_ordered": (_i32, [_vp, CoderConfig, _vp, _vp, _z, _vp, _vp, _vp, _z, _vp, _vp, _vp]),
    "cst_ans_decode_ragged_ordered": (_i32, [_vp, CoderConfig, _vp, _vp, _z, _z, _vp, _vp, _vp, _z, _vp, _vp, _vp]),
    "cst_ans_count_until_ordered": (_i32, [_vp, CoderConfig, _vp, _vp, _z, _z, _vp, _z, _vp, _i32, _z, _vp, _vp, _vp]),
    "cst_ragged_jump_scratch_bytes": (_z, [_z]),
    "cst_ans_encode_ragged_jump": (_i32, [_vp, CoderConfig, _vp, _vp, _z, _vp, _vp, _vp, _z, _vp, _z, _vp, _vp, _vp, _vp, _vp]),
    "cst_ans_decode_ragged_jump": (_i32, [_vp, CoderConfig, _vp, _vp, _z, _z, _vp, _vp, _vp, _z, _z, _vp, _z, _vp, _vp, _vp, _vp, _vp]),
    "cst_compact_scratch_bytes": (_z, [_z]),
    "cst_compact_words": (_i32, [_vp, _z, _vp, _z, _vp, _vp, _z, _vp, _vp]),
    "cst_words_reverse": (_i32, [_vp, _vp, _z, _vp, _z, _vp, _vp, _z, _vp]),
    "cst_rccl_get_unique_id": (_i32, [_vp]),
    "cst_rccl_comm_init": (_i32, [_vp, _i32, _i32, C.POINTER(_vp)]),
    "cst_rccl_comm_destroy": (_i32, [_vp]),
    "cst_gather_sizes_rccl": (_i32, [_vp, _i32, _i32, _vp, _z, _vp, _vp]),
    "cst_gather_rccl": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cst_scatter_rccl": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cst_ans_encode_gaussian_batch": (_i32, [CoderConfig, _i32, _i32, _vp, _vp, _vp, _z, _z, _i32, _vp, _z, _vp, _vp, _vp, _u32, _vp]),
    "cst_ans_decode_gaussian_batch": (_i32, [CoderConfig, _i32, _i32, _vp, _vp, _z, _z, _vp, _vp, _vp, _vp, _z, _z, _i32, _vp, _vp, _vp, _u32, _vp]),
    "cst_ans_encode_cp_batch": (_i32, [CoderConfig, _vp, _vp, _z, _z, _i32, _vp, _z, _vp, _vp, _vp, _u32, _vp]),
    "cst_ans_decode_rows_batch": (_i32, [CoderConfig, _vp, _vp, _z, _z, _vp, _vp, _i32, _i32, _vp, _z, _z, _i32, _vp, _vp, _vp, _u32, _vp]),
    "cst_range_encode_batch": (_i32, [_vp, CoderConfig, _vp, _z, _z, _i32, _vp, _z, _vp, _vp, _vp, _u32, _vp]),
    "cst_range_decode_batch": (_i32, [_vp, CoderConfig, _vp, _vp, _z, _z, _vp, _vp, _z, _z, _i32, _vp, _vp, _u32, _vp]),
    "cst_range_encode_cp_batch": (_i32, [CoderConfig, _vp, _vp, _z, _z, _i32, _vp, _z, _vp, _vp, _vp, _u32, _vp]),
    "cst_range_encode_gaussian_batch": (_i32, [CoderConfig, _i32, _i32, _vp, _vp, _vp, _z, _z, _i32, _vp, _z, _vp, _vp, _vp, _u32, _vp]),
    "cst_range_decode_gaussian_batch": (_i32, [CoderConfig, _i32, _i32, _vp, _vp, _z, _z, _vp, _vp, _vp, _vp, _z, _z, _i32, _vp, _vp, _u32, _vp]),
    "cst_range_decode_rows_batch": (_i32, [CoderConfig, _vp, _vp, _z, _z, _vp, _vp, _i32, _i32, _vp, _z, _z, _i32, _vp, _vp, _u32, _vp]),
    "cst_chain_encode_cp_batch": (_i32, [CoderConfig, _vp, _vp, _z, _z, _i32, _vp, _vp, _z, _vp, _vp, _z, _vp, _vp, _vp, _vp]),
    "cst_chain_encode_gaussian_batch": (_i32, [CoderConfig, _i32, _i32, _vp, _vp, _vp, _z, _z, _i32, _vp, _vp, _z, _vp, _vp, _z, _vp,
                                               _vp, _vp, _vp]),
    "cst_chain_decode_gaussian_batch": (_i32, [CoderConfig, _i32, _i32, _vp, _vp, _z, _vp, _vp, _vp, _vp, _z, _z, _i32, _vp, _z, _vp,
                                               _vp, _vp, _vp]),
    "cst_chain_decode_rows_batch": (_i32, [CoderConfig, _vp, _vp, _z, _vp, _vp, _z, _i32, _i32, _vp, _z, _z, _i32, _vp, _z, _vp, _vp,
                                           _vp, _vp]),
    "cst_family_cdf_rows": (_i32, [_i32, _i32, _i32, _i32, _vp, _vp, _vp, _z, _vp, _vp, _vp]),
    "cst_categorical_perfect_cdf": (_i32, [_vp, _z, _i32, _vp]),
    "cst_debug_family_fn": (_i32, [_i32, _vp, _vp, _z, _vp]),
    "cst_debug_host_log1p": (_f64, [_f64]),
    "cst_release_scratch": (_i32, []),
    "cst_debug_erf": (_i32, [_vp, _vp, _z, _vp]),
    "cst_debug_erf_tab": (_i32, [_vp, _vp, _z, _vp]),
    "cst_debug_erf_fast": (_i32, [_i32, _vp, _vp, _z, _vp]),
    "cst_debug_gaussian_left_quick": (_i32, [_i32, _i32, _i32, _vp, _vp, _vp, _z, _vp, _vp]),
    "cst_debug_gaussian_lcp": (_i32, [_i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _z, _vp]),
}

_lib = None


def load_library():
    """Loads libconstriction_amd.so and types every entry point.  Does not need a GPU."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise BackendUnavailable(
            f"{LIB_PATH} is missing: build it with `python -m constriction_amd.build` "
            "(or __graft_entry__.build()).  There is no CPU fallback.")
    # torch first: it brings its own HIP runtime, and a process must not end up with two of them (the library loaded
    # before torch binds /opt/rocm's runtime and then sees no device once torch has initialised its bundled one)
    import torch  # noqa: F401
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here means the ABI and the binding disagree
        fn.restype = res
        fn.argtypes = args
    if lib.cst_abi_version() != 5:
        raise BackendUnavailable("ABI version mismatch between _native.py and libconstriction_amd.so")
    _lib = lib
    return lib


def lib():
    """The library, after checking that a GPU is actually present."""
    l = load_library()
    n = l.cst_device_count()
    if n < 1:
        raise BackendUnavailable("no MI355X (gfx950) device visible; constriction_amd has no CPU fallback")
    return l


def reload_knobs():
    """cst_debug_reload_knobs: the library re-reads its CST_* debug switches from the environment (it reads them once, when it
    is loaded); for tests that set one inside the process"""
    load_library().cst_debug_reload_knobs()


def check(status: int, what: str = ""):
    if status == CST_OK:
        return
    l = load_library()
    msg = {CST_ERR_INVALID_ARGUMENT: "invalid argument", CST_ERR_HIP: "HIP error: " + l.cst_last_hip_error().decode(),
           CST_ERR_NO_DEVICE: "no device", CST_ERR_MODEL: "model cannot be built",
           CST_ERR_OUT_OF_MEMORY: "out of memory"}.get(status, f"status {status}")
    if status == CST_ERR_NO_DEVICE:
        raise BackendUnavailable(f"{what}: {msg}")
    if status == CST_ERR_MODEL:
        raise ValueError(f"{what}: {msg}")
    raise BackendError(f"{what}: {msg}")
