"""ctypes binding of include/vdb_index.h (vectordb_amd/libvdb_amd.so).

The library is built in-tree by ``make -C vectordb_amd/csrc`` (``__graft_entry__.build()``).
There is no Python or CPU fallback: if the library is missing, importing a search entry
point fails.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libvdb_amd.so")

VDB_OK, VDB_ERR_INVALID, VDB_ERR_RUNTIME, VDB_ERR_NO_DEVICE, VDB_ERR_UNSUPPORTED = range(5)
L2, IP = 0, 1
SUM_SEQ, SUM_TREE64 = 0, 1
BF_FP32, BF_BF16, BF_BF16_RERANK = 0, 1, 2

_f32p = C.POINTER(C.c_float)
_u64p = C.POINTER(C.c_uint64)
_u32p = C.POINTER(C.c_uint32)
_vp = C.c_void_p

# name -> (restype, argtypes); one entry per declaration of include/vdb_index.h
SIGNATURES = {
    "vdb_last_error": (C.c_char_p, []),
    "vdb_device_count": (C.c_int, []),
    "vdb_index_create": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t]),
    "vdb_index_load": (C.c_int, [C.POINTER(_vp), C.c_char_p, C.c_int, C.c_size_t, C.c_size_t]),
    "vdb_index_save": (C.c_int, [_vp, C.c_char_p]),
    "vdb_index_free": (None, [_vp]),
    "vdb_index_add": (C.c_int, [_vp, _f32p, C.c_uint64, C.c_int]),
    "vdb_index_add_point": (C.c_int, [_vp, _f32p, C.c_uint64, C.c_int]),
    "vdb_index_set_reuse_deleted": (C.c_int, [_vp, C.c_int]),
    "vdb_index_add_batch": (C.c_int, [_vp, _f32p, _u64p, C.c_size_t, C.c_int]),
    "vdb_index_add_batch_device": (C.c_int, [_vp, _f32p, _u64p, C.c_size_t, C.c_int, C.c_int]),
    "vdb_index_set_build_sum_order": (C.c_int, [_vp, C.c_int]),
    "vdb_index_set_ef": (C.c_int, [_vp, C.c_size_t]),
    "vdb_index_set_spill_mode": (C.c_int, [_vp, C.c_int]),
    "vdb_index_mark_delete": (C.c_int, [_vp, C.c_uint64]),
    "vdb_index_unmark_delete": (C.c_int, [_vp, C.c_uint64]),
    "vdb_index_get_data_by_label": (C.c_int, [_vp, C.c_uint64, _f32p]),
    "vdb_index_resize": (C.c_int, [_vp, C.c_size_t]),
    "vdb_index_check_integrity": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "vdb_index_capacity": (C.c_size_t, [_vp]),
    "vdb_index_count": (C.c_size_t, [_vp]),
    "vdb_index_deleted_count": (C.c_size_t, [_vp]),
    "vdb_index_dim": (C.c_size_t, [_vp]),
    "vdb_index_M": (C.c_size_t, [_vp]),
    "vdb_index_ef": (C.c_size_t, [_vp]),
    "vdb_index_max_level": (C.c_int, [_vp]),
    "vdb_index_entry": (C.c_uint32, [_vp]),
    "vdb_index_level_of": (C.c_int, [_vp, C.c_uint32]),
    "vdb_index_links": (C.c_int, [_vp, C.c_uint32, C.c_int, _u32p, C.c_int]),
    "vdb_index_to_device": (C.c_int, [_vp, C.c_int]),
    "vdb_index_device_bytes": (C.c_size_t, [_vp]),
    "vdb_search_batch_device": (C.c_int, [_vp, _vp, C.c_size_t, C.c_size_t, _vp, _vp, _vp, _vp]),
    "vdb_search_batch_device_noalloc": (C.c_int, [_vp, _vp, C.c_size_t, C.c_size_t, _vp, _vp, _vp, _vp]),
    "vdb_search_batch": (C.c_int, [_vp, _f32p, C.c_size_t, C.c_size_t, _f32p, _u64p, _u32p]),
    "vdb_search_batch_device_filtered": (C.c_int, [_vp, _vp, C.c_size_t, C.c_size_t, _vp, _vp, _vp, _vp, _vp]),
    "vdb_search_batch_filtered": (C.c_int, [_vp, _f32p, C.c_size_t, C.c_size_t, _u32p, _f32p, _u64p, _u32p]),
    "vdb_search_epsilon_batch_device": (C.c_int, [_vp, _vp, C.c_size_t, C.c_float, C.c_size_t, C.c_size_t, _vp, _vp, _vp, _vp, _vp]),
    "vdb_search_epsilon_batch": (C.c_int, [_vp, _f32p, C.c_size_t, C.c_float, C.c_size_t, C.c_size_t, _u32p, _f32p, _u64p, _u32p]),
    "vdb_index_load_to_device": (C.c_int, [C.POINTER(_vp), C.c_char_p, C.c_int, C.c_size_t, C.c_int]),
    "vdb_index_labels": (C.c_int, [_vp, _u64p, C.c_size_t]),
    "vdb_bf_create": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_size_t, C.c_size_t]),
    "vdb_bf_load": (C.c_int, [C.POINTER(_vp), C.c_char_p, C.c_int, C.c_size_t]),
    "vdb_bf_save": (C.c_int, [_vp, C.c_char_p]),
    "vdb_bf_free": (None, [_vp]),
    "vdb_bf_add_batch": (C.c_int, [_vp, _f32p, _u64p, C.c_size_t]),
    "vdb_bf_remove": (C.c_int, [_vp, C.c_uint64]),
    "vdb_bf_count": (C.c_size_t, [_vp]),
    "vdb_bf_to_device": (C.c_int, [_vp, C.c_int]),
    "vdb_bf_search_batch_device": (C.c_int, [_vp, _vp, C.c_size_t, C.c_size_t, C.c_int, _vp, _vp, _vp]),
    "vdb_bf_search_batch": (C.c_int, [_vp, _f32p, C.c_size_t, C.c_size_t, C.c_int, _f32p, _u64p]),
    "vdb_bf_search_batch_device_filtered": (C.c_int, [_vp, _vp, C.c_size_t, C.c_size_t, _vp, _vp, _vp, _vp]),
    "vdb_bf_search_batch_filtered": (C.c_int, [_vp, _f32p, C.c_size_t, C.c_size_t, _u32p, _f32p, _u64p]),
    "vdb_bf_labels": (C.c_int, [_vp, _u64p, C.c_size_t]),
    "vdb_comm_unique_id": (C.c_int, [_vp]),
    "vdb_comm_create": (C.c_int, [C.POINTER(_vp), _vp, C.c_int, C.c_int, C.c_int]),
    "vdb_comm_wrap": (C.c_int, [C.POINTER(_vp), _vp, C.c_int]),
    "vdb_comm_free": (None, [_vp]),
    "vdb_comm_size": (C.c_int, [_vp]),
    "vdb_comm_rank": (C.c_int, [_vp]),
    "vdb_sharded_search_batch_device": (C.c_int, [_vp, _vp, _vp, C.c_size_t, C.c_size_t, _vp, _vp, _vp]),
    "vdb_merge_topk_device": (C.c_int, [_vp, _vp, C.c_size_t, C.c_size_t, C.c_size_t, _vp, _vp, _vp]),
}

_lib = None


class VdbError(RuntimeError):
    """What the reference reports as std::runtime_error, carrying the C-ABI status code."""

    def __init__(self, code: int, message: str):
        super().__init__(message)
        self.code = code


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `make -C vectordb_amd/csrc` "
                              "(there is no CPU fallback for the search path)")
        # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64.so.7.  Loaded
        # first, it is the copy this library binds to as well (same SONAME); loaded second,
        # beside /opt/rocm's, torch finds no GPU.  So when torch is installed, import it first.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc: int) -> None:
    if rc != VDB_OK:
        raise VdbError(rc, lib().vdb_last_error().decode(errors="replace"))
