"""ctypes binding of libvbnmf_hip.so (the C ABI declared in include/vbnmf.h).

The library is REQUIRED: importing this module never falls back to a CPU path.  If the
shared object is missing, ``load()`` raises with the build command; if it loads but no
gfx950 device is present, every compute entry point returns VBNMF_ERR_NO_DEVICE and the
wrappers raise ``VBNMFError``.
"""
from __future__ import annotations

import ctypes
import sys
import importlib.util
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# VBNMF_LIB overrides the path (A/B runs of experimental builds); the default is the in-tree build.
LIB_PATH = os.environ.get("VBNMF_LIB") or os.path.join(_HERE, "lib", "libvbnmf_hip.so")

c_double_p = ctypes.POINTER(ctypes.c_double)
c_int32_p = ctypes.POINTER(ctypes.c_int32)
c_int64_p = ctypes.POINTER(ctypes.c_int64)
c_uint32_p = ctypes.POINTER(ctypes.c_uint32)

OK, ERR_BAD_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_OOM, ERR_STATE = range(6)
MAX_RANK = 128
MAX_SVD_COLUMNS = 64            # vbnmf_engine_svd (device-resident subspace)
COMM_ID_BYTES = 128


class VBNMFError(RuntimeError):
    """A C-ABI call returned a non-zero vbnmf_status."""

    def __init__(self, code: int, message: str):
        super().__init__(f"[vbnmf status {code}] {message}")
        self.code = code


class LayoutView(ctypes.Structure):
    _fields_ = [
        ("side", ctypes.c_int32), ("wide", ctypes.c_int32),
        ("n_major", ctypes.c_int64), ("n_minor", ctypes.c_int64),
        ("block_width", ctypes.c_int32), ("n_blocks", ctypes.c_int32), ("max_len", ctypes.c_int32), ("n_wg", ctypes.c_int32),
        ("row_slots", ctypes.c_int32),
        ("n_tasks", ctypes.c_int64), ("n_slices", ctypes.c_int64), ("n_slots", ctypes.c_int64), ("n_segs", ctypes.c_int64),
        ("task_major", c_uint32_p), ("slice_width", c_int32_p), ("slice_off", c_int64_p), ("slice_block", c_int32_p), ("slice_fast", c_int32_p),
        ("block_start", c_int64_p), ("seg_block", c_int32_p), ("wg_seg0", c_int32_p), ("seg_ptr", c_int32_p),
        ("inv_ptr", c_int32_p), ("inv_task", c_uint32_p),
        ("packed", c_uint32_p), ("wide_idx", c_uint32_p), ("wide_val", c_double_p),
        ("cell_perm", c_int32_p),
    ]


# name -> (restype, argtypes); every symbol include/vbnmf.h declares
_D, _I32, _I64, _VP = ctypes.c_double, ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p
_VPP = ctypes.POINTER(ctypes.c_void_p)
SIGNATURES = {
    "vbnmf_last_error": (ctypes.c_char_p, []),
    "vbnmf_version": (ctypes.c_char_p, []),
    "vbnmf_device_count": (_I32, []),
    "vbnmf_matrix_from_dense": (ctypes.c_int, [_I64, _I64, c_double_p, _VPP]),
    "vbnmf_matrix_from_csc": (ctypes.c_int, [_I64, _I64, c_int32_p, c_int32_p, c_double_p, _VPP]),
    "vbnmf_matrix_from_csr": (ctypes.c_int, [_I64, _I64, c_int32_p, c_int32_p, c_double_p, _VPP]),
    "vbnmf_matrix_from_mtx": (ctypes.c_int, [ctypes.c_char_p, _VPP]),
    "vbnmf_matrix_write_mtx": (ctypes.c_int, [_VP, ctypes.c_char_p]),
    "vbnmf_matrix_csc": (ctypes.c_int, [_VP, ctypes.POINTER(c_int64_p), ctypes.POINTER(c_int32_p), ctypes.POINTER(c_double_p)]),
    "vbnmf_matrix_info": (ctypes.c_int, [_VP, c_int64_p, c_int64_p, c_int64_p, c_double_p]),
    "vbnmf_matrix_empty_counts": (ctypes.c_int, [_VP, c_int64_p, c_int64_p]),
    "vbnmf_matrix_plan_ranks": (ctypes.c_int, [_VP, c_int32_p, _I32, _I32]),
    "vbnmf_plan_classes": (ctypes.c_int, [c_int32_p, _I32, _I32, c_int32_p, c_int32_p]),
    "vbnmf_padded_rank": (_I32, [_I32]),
    "vbnmf_host_threads": (_I32, []),
    "vbnmf_set_host_threads": (_I32, [_I32]),
    "vbnmf_matrix_get_meta": (ctypes.c_int, [_VP, c_double_p]),
    "vbnmf_matrix_shell": (ctypes.c_int, [c_double_p, _VPP]),
    "vbnmf_matrix_is_shell": (ctypes.c_int, [_VP]),
    "vbnmf_matrix_prepare": (ctypes.c_int, [_VP]),
    "vbnmf_matrix_prepare_async": (ctypes.c_int, [_VP]),
    "vbnmf_matrix_preload_layout": (ctypes.c_int, [_VP, _I32, _I32, _I32, _I32]),
    "vbnmf_matrix_export_layout": (ctypes.c_int, [_VP, _I32, _I32, _I32, _VP, _I64, c_int64_p]),
    "vbnmf_matrix_import_layout": (ctypes.c_int, [_VP, _VP, _I64]),
    "vbnmf_matrix_share_layout": (ctypes.c_int, [_VP, _I32, _I32, _I32, ctypes.c_char_p]),
    "vbnmf_matrix_attach_layout": (ctypes.c_int, [_VP, ctypes.c_char_p]),
    "vbnmf_device_sweep_workgroups": (ctypes.c_int, [_I32, c_int32_p]),
    "vbnmf_device_warmup": (ctypes.c_int, [_I32]),
    "vbnmf_matrix_destroy": (None, [_VP]),
    "vbnmf_engine_create": (ctypes.c_int, [_VP, _I32, _I32, _VPP]),
    "vbnmf_engine_create_part": (ctypes.c_int, [_VP, _I64, _I64, _I64, _I32, _I32, _VPP]),
    "vbnmf_engine_create_geom": (ctypes.c_int, [_VP, _I64, _I64, _I64, _I32, _I32, _I32, _VPP]),
    "vbnmf_engine_destroy": (None, [_VP]),
    "vbnmf_engine_dims": (ctypes.c_int, [_VP, c_int64_p, c_int64_p, c_int32_p]),
    "vbnmf_engine_set_state": (ctypes.c_int, [_VP, c_double_p, c_double_p, c_double_p]),
    "vbnmf_engine_step": (ctypes.c_int, [_VP, _D, _D, _D, _D, _D, c_double_p, c_double_p]),
    "vbnmf_engine_step_local": (ctypes.c_int, [_VP, _D, _D, _D, _D, _D]),
    "vbnmf_engine_reduce_buffer": (ctypes.c_int, [_VP, _VPP, c_int64_p]),
    "vbnmf_engine_step_finish": (ctypes.c_int, [_VP, c_double_p, c_double_p]),
    "vbnmf_engine_state_finish": (ctypes.c_int, [_VP]),
    "vbnmf_engine_run": (ctypes.c_int, [_VP, c_double_p, _D, _I32, _D, _I32, _I32, c_int32_p, c_int32_p, c_double_p,
                                        c_double_p, c_int32_p, c_double_p, _I64]),
    "vbnmf_set_engine_grid": (ctypes.c_int, [_I32, _I32]),
    "vbnmf_set_engine_padding": (ctypes.c_int, [_I32]),
    "vbnmf_batch_run": (ctypes.c_int, [_VPP, _I32, c_double_p, _D, _I32, _D, _I32, _I32, c_int32_p, c_int32_p, c_double_p,
                                       c_double_p, c_int32_p, c_double_p, _I64]),
    "vbnmf_batch_ml_run": (ctypes.c_int, [_VPP, _I32, _I32, _D, _D, _I32, _D, c_int32_p, c_double_p, c_int32_p, c_double_p, _I64]),
    "vbnmf_comm_unique_id": (ctypes.c_int, [_VP, _I64]),
    "vbnmf_comm_create": (ctypes.c_int, [_VP, _I64, _I32, _I32, _I32, _VPP]),
    "vbnmf_comm_create_local": (ctypes.c_int, [_I32, _I32, _VPP]),
    "vbnmf_comm_info": (ctypes.c_int, [_VP, c_int32_p, c_int32_p, c_int32_p]),
    "vbnmf_comm_destroy": (None, [_VP]),
    "vbnmf_engine_attach_comm": (ctypes.c_int, [_VP, _VP]),
    "vbnmf_engine_allreduce": (ctypes.c_int, [_VP]),
    "vbnmf_group_state_finish": (ctypes.c_int, [_VP]),
    "vbnmf_group_run": (ctypes.c_int, [_VP, c_double_p, _D, _I32, _D, _I32, _I32, c_int32_p, c_int32_p, c_double_p,
                                       c_double_p, c_int32_p, c_double_p, _I64]),
    "vbnmf_engine_get_state": (ctypes.c_int, [_VP] + [c_double_p] * 6),
    "vbnmf_engine_get_stream": (ctypes.c_int, [_VP, _VPP]),
    "vbnmf_engine_set_stream": (ctypes.c_int, [_VP, _VP]),
    "vbnmf_engine_timing_enable": (ctypes.c_int, [_VP, _I32]),
    "vbnmf_engine_timing_get": (ctypes.c_int, [_VP, c_double_p, c_int64_p]),
    "vbnmf_engine_layout_info": (ctypes.c_int, [_VP] + [c_int64_p] * 6),
    "vbnmf_engine_debug_times": (ctypes.c_int, [_VP, ctypes.POINTER(ctypes.c_uint64), _I64, c_int32_p, c_int32_p]),
    "vbnmf_stateless_cache_clear": (None, []),
    "vbnmf_pool_trim": (None, []),
    "vbnmf_update_dense": (ctypes.c_int, [_I64, _I64, _I32, c_double_p, c_double_p, c_double_p, c_double_p]
                           + [_D] * 5 + [c_double_p] * 7),
    "vbnmf_update_csc": (ctypes.c_int, [_I64, _I64, _I32, c_int32_p, c_int32_p, c_double_p,
                                        c_double_p, c_double_p, c_double_p] + [_D] * 5 + [c_double_p] * 7),
    "vbnmf_engine_ml_set_state": (ctypes.c_int, [_VP, c_double_p, c_double_p]),
    "vbnmf_engine_ml_step": (ctypes.c_int, [_VP, _I32, _D, _D, c_double_p]),
    "vbnmf_engine_ml_run": (ctypes.c_int, [_VP, _I32, _D, _D, _I32, _D, c_int32_p, c_double_p, c_int32_p, c_double_p, _I64]),
    "vbnmf_engine_ml_likelihood": (ctypes.c_int, [_VP, c_double_p]),
    "vbnmf_engine_ml_get_state": (ctypes.c_int, [_VP, c_double_p, c_double_p]),
    "vbnmf_ml_update_dense": (ctypes.c_int, [_I64, _I64, _I32, c_double_p, c_double_p, c_double_p, _I32, _D, _D,
                                             c_double_p, c_double_p, c_double_p]),
    "vbnmf_ml_update_csc": (ctypes.c_int, [_I64, _I64, _I32, c_int32_p, c_int32_p, c_double_p, c_double_p, c_double_p,
                                           _I32, _D, _D, c_double_p, c_double_p, c_double_p]),
    "vbnmf_engine_cluster_ids": (ctypes.c_int, [_VP, c_int32_p]),
    "vbnmf_engine_spmm": (ctypes.c_int, [_VP, _I32, c_double_p, c_double_p]),
    "vbnmf_engine_cluster_changes": (ctypes.c_int, [_VP, c_int64_p, c_int32_p]),
    "vbnmf_engine_random_state": (ctypes.c_int, [_VP, _D, _D, _D, _D, ctypes.c_uint64]),
    "vbnmf_engine_svd": (ctypes.c_int, [_VP, _I32, _D, _I32, ctypes.c_uint64, c_double_p, c_double_p, c_double_p, c_int32_p]),
    "vbnmf_layout_build": (ctypes.c_int, [_VP, _I64, _I64, _I32, _I32, _VPP, ctypes.POINTER(LayoutView)]),
    "vbnmf_layout_destroy": (None, [_VP]),
    "vbnmf_test_special_host": (ctypes.c_int, [_I32, _I64, c_double_p, c_double_p]),
    "vbnmf_test_special_device": (ctypes.c_int, [_I32, _I64, c_double_p, c_double_p]),
    "vbnmf_test_stream_sleep": (ctypes.c_int, [_VP, _D]),
    "vbnmf_test_hash_bytes": (ctypes.c_uint64, [_VP, _I64, ctypes.c_uint64]),
}

_lib = None


def load():
    """Load libvbnmf_hip.so and bind every symbol; raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make` (or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "ccfindr_amd has no CPU fallback.")
    # One HIP runtime per process.  PyTorch ships its own libamdhip64; a process that maps this library (and with it
    # /opt/rocm's runtime) first and torch's afterwards ends with two runtimes, and the second one finds no device
    # ("No HIP GPUs are available" from torch.cuda).  With torch mapped first both use torch's copy.  So: torch first,
    # when it is installed (it is plumbing here -- device tensors, torch.distributed -- and optional).
    if "torch" not in sys.modules and importlib.util.find_spec("torch") is not None:
        try:
            import torch  # noqa: F401
        except Exception:  # noqa: BLE001 -- a broken torch must not keep the engine from loading
            pass
    L = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)          # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def check(rc: int):
    if rc != 0:
        raise VBNMFError(rc, load().vbnmf_last_error().decode("utf-8", "replace"))


def dptr(a):
    return None if a is None else a.ctypes.data_as(c_double_p)


def fcol(a):
    """float64 column-major copy/view: how R and Eigen store a matrix."""
    return np.asfortranarray(np.asarray(a, dtype=np.float64))
