"""ctypes binding of the C-ABI in include/msckf_mi355x.h.  No fallback: if the
HIP library has not been built, `load()` raises."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
FLAG_TREE_PLAN = 1
FLAG_BAND_ONLY = 2      # msckf_config.flags: long tracks are not split (one Householder plan for every track: 90-column tiles / merge tree)
LIB_PATH = os.environ.get("MSCKF_LIB") or os.path.join(_HERE, "libmsckf_mi355x.so")   # MSCKF_LIB: A/B builds
ABI_VERSION = 2
DTYPE_F64, DTYPE_F32 = 0, 1

OK, NOOP = 0, 1
ERR_ARG, ERR_HIP, ERR_NO_DEVICE, ERR_NOT_SPD, ERR_STATE, ERR_DUP_SLOT, ERR_COMM = -1, -2, -3, -4, -5, -6, -7
COMM_ID_BYTES = 128
MAX_TRACK = 31

# every symbol include/msckf_mi355x.h declares
SYMBOLS = [
    "msckf_create", "msckf_destroy", "msckf_strerror", "msckf_last_error", "msckf_device_count",
    "msckf_update", "msckf_set_state", "msckf_set_features", "msckf_run", "msckf_run_timed", "msckf_sync",
    "msckf_get_result", "msckf_commit_covariance", "msckf_run_compress", "msckf_block_doubles",
    "msckf_export_block", "msckf_run_merge_gain", "msckf_set_group_exchange", "msckf_band_rule", "msckf_group_record_doubles",
    "msckf_export_groups", "msckf_run_merge_groups", "msckf_run_merge_groups_flags", "msckf_export_result", "msckf_import_covariance", "msckf_debug_gate", "msckf_debug_compressed", "msckf_debug_fold_stamps",
    "msckf_device_pointer", "msckf_stream",
    "msckf_set_tracks", "msckf_run_select", "msckf_clear_selection", "msckf_get_selection",
    "msckf_debug_time_select", "msckf_replan", "msckf_run_associate",
    "msckf_propagate", "msckf_augment", "msckf_remove_clones", "msckf_set_poses", "msckf_get_covariance",
    "msckf_comm_unique_id", "msckf_comm_init", "msckf_comm_destroy", "msckf_comm_gather", "msckf_comm_broadcast",
    "msckf_comm_allreduce", "msckf_comm_buffer", "msckf_comm_put", "msckf_comm_get",
    "msckf_set_exchange_mask", "msckf_result_range_doubles", "msckf_get_shared_result", "msckf_set_exchange_span",
    "msckf_debug_split", "msckf_debug_set_rem_direct_rows",
]


class Config(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("device", C.c_int32), ("max_clones", C.c_int32),
                ("max_features", C.c_int32), ("max_track", C.c_int32), ("leaf_rows", C.c_int32),
                ("merge_arity", C.c_int32), ("flags", C.c_int32), ("dtype", C.c_int32), ("reserved0", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("n_features", C.c_int32), ("n_accepted", C.c_int32), ("n_rejected", C.c_int32),
                ("stacked_rows", C.c_int32), ("n_leaves", C.c_int32), ("n_levels", C.c_int32),
                ("not_spd", C.c_int32), ("k5_launches", C.c_int32),
                ("us_total", C.c_float), ("us_feature", C.c_float), ("us_qr", C.c_float), ("us_gain", C.c_float),
                ("us_host_prep", C.c_float), ("us_h2d", C.c_float), ("us_d2h", C.c_float), ("reserved2", C.c_float)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if not k.startswith("reserved")}


class AssocParamsC(C.Structure):
    _fields_ = [("K", C.c_double * 9), ("R_cur", C.c_double * 9), ("t_cur", C.c_double * 3),
                ("epipolar_threshold", C.c_double), ("homography_threshold", C.c_double)]


class SelectParamsC(C.Structure):
    _fields_ = [("min_frames_lost", C.c_int32), ("min_frames_tracked", C.c_int32), ("use_parallax", C.c_int32),
                ("width", C.c_int32), ("height", C.c_int32), ("reserved", C.c_int32),
                ("min_parallax_deg", C.c_double), ("K", C.c_double * 9)]


SEL_VALID, SEL_LOST, SEL_REFRESHED = 1, 2, 4

_lib = None

# Array arguments travel as plain addresses (c_void_p): `ndarray.ctypes.data_as(POINTER(...))` costs 3-5 us per
# argument, which was 60+ us of the 23-argument one-shot call.
_dp = C.c_void_p
_ip = C.c_void_p
_up = C.c_void_p


def load():
    """Load libmsckf_mi355x.so and set prototypes.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `make -C monocular-visual-inertial-msckf_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    lib.msckf_create.argtypes = [C.POINTER(vp), C.POINTER(Config)]
    lib.msckf_create.restype = C.c_int
    lib.msckf_destroy.argtypes = [vp]
    lib.msckf_destroy.restype = None
    lib.msckf_strerror.argtypes = [C.c_int]
    lib.msckf_strerror.restype = C.c_char_p
    lib.msckf_last_error.argtypes = [vp]
    lib.msckf_last_error.restype = C.c_char_p
    lib.msckf_device_count.argtypes = []
    lib.msckf_device_count.restype = C.c_int
    lib.msckf_update.argtypes = [vp, C.c_int32, _dp, _dp, _dp, _dp, _dp, _dp, _dp, C.c_double, C.c_int32, _ip, _dp,
                                 _ip, _dp, _dp, _dp, _dp, C.c_int32, _dp, _dp, _up, C.POINTER(Stats)]
    lib.msckf_update.restype = C.c_int
    lib.msckf_set_state.argtypes = [vp, C.c_int32, _dp, _dp, _dp, _dp, _dp, _dp, _dp, C.c_double, _dp, C.c_int32]
    lib.msckf_set_state.restype = C.c_int
    lib.msckf_set_features.argtypes = [vp, C.c_int32, _ip, _dp, _ip, _dp, _dp, _dp]
    lib.msckf_set_features.restype = C.c_int
    for name in ("msckf_run", "msckf_sync", "msckf_run_compress", "msckf_commit_covariance", "msckf_replan"):
        getattr(lib, name).argtypes = [vp]
        getattr(lib, name).restype = C.c_int
    lib.msckf_run_timed.argtypes = [vp, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.msckf_run_timed.restype = C.c_int
    lib.msckf_get_result.argtypes = [vp, _dp, _dp, _up, C.POINTER(Stats)]
    lib.msckf_get_result.restype = C.c_int
    lib.msckf_block_doubles.argtypes = [vp]
    lib.msckf_block_doubles.restype = C.c_size_t
    lib.msckf_export_block.argtypes = [vp, vp, C.c_int, _ip]
    lib.msckf_export_block.restype = C.c_int
    lib.msckf_run_merge_gain.argtypes = [vp, vp, C.c_int32, C.c_int, C.c_int32]
    lib.msckf_run_merge_gain.restype = C.c_int
    lib.msckf_set_group_exchange.argtypes = [vp, C.c_int]
    lib.msckf_set_group_exchange.restype = C.c_int
    lib.msckf_band_rule.argtypes = [vp, C.c_int32, C.c_int32]
    lib.msckf_band_rule.restype = C.c_int
    lib.msckf_group_record_doubles.argtypes = [vp]
    lib.msckf_group_record_doubles.restype = C.c_size_t
    lib.msckf_export_groups.argtypes = [vp, vp, C.c_int, _ip]
    lib.msckf_export_groups.restype = C.c_int
    lib.msckf_run_merge_groups.argtypes = [vp, vp, C.c_int32, C.c_int, C.c_int32]
    lib.msckf_run_merge_groups.restype = C.c_int
    lib.msckf_run_merge_groups_flags.argtypes = [vp, vp, C.c_int32, C.c_int, vp]
    lib.msckf_run_merge_groups_flags.restype = C.c_int
    lib.msckf_comm_unique_id.argtypes = [vp]
    lib.msckf_comm_unique_id.restype = C.c_int
    lib.msckf_comm_init.argtypes = [vp, C.c_int32, C.c_int32, vp]
    lib.msckf_comm_init.restype = C.c_int
    lib.msckf_comm_destroy.argtypes = [vp]
    lib.msckf_comm_destroy.restype = C.c_int
    lib.msckf_comm_gather.argtypes = [vp, vp, vp, C.c_size_t, C.c_int32]
    lib.msckf_comm_gather.restype = C.c_int
    lib.msckf_comm_broadcast.argtypes = [vp, vp, C.c_size_t, C.c_int32]
    lib.msckf_comm_broadcast.restype = C.c_int
    lib.msckf_comm_allreduce.argtypes = [vp, vp, C.c_size_t, C.c_int32]
    lib.msckf_comm_allreduce.restype = C.c_int
    lib.msckf_comm_put.argtypes = [vp, vp, vp, C.c_size_t]
    lib.msckf_comm_put.restype = C.c_int
    lib.msckf_comm_get.argtypes = [vp, vp, vp, C.c_size_t]
    lib.msckf_comm_get.restype = C.c_int
    lib.msckf_set_exchange_span.argtypes = [vp, C.c_int32]
    lib.msckf_set_exchange_span.restype = C.c_int
    lib.msckf_set_exchange_mask.argtypes = [vp, C.c_int32, _ip]
    lib.msckf_set_exchange_mask.restype = C.c_int
    lib.msckf_result_range_doubles.argtypes = [vp]
    lib.msckf_result_range_doubles.restype = C.c_size_t
    lib.msckf_get_shared_result.argtypes = [vp, _dp, _dp, _up, C.POINTER(Stats)]
    lib.msckf_get_shared_result.restype = C.c_int
    lib.msckf_comm_buffer.argtypes = [vp, C.c_size_t]
    lib.msckf_comm_buffer.restype = vp
    lib.msckf_export_result.argtypes = [vp, vp, vp, C.c_int]
    lib.msckf_export_result.restype = C.c_int
    lib.msckf_import_covariance.argtypes = [vp, vp, C.c_int]
    lib.msckf_import_covariance.restype = C.c_int
    lib.msckf_debug_gate.argtypes = [vp, _dp, _ip]
    lib.msckf_debug_gate.restype = C.c_int
    lib.msckf_debug_compressed.argtypes = [vp, _dp, _dp]
    lib.msckf_debug_compressed.restype = C.c_int
    lib.msckf_debug_fold_stamps.argtypes = [vp, C.POINTER(C.c_longlong), C.c_int32]
    lib.msckf_debug_fold_stamps.restype = C.c_int
    lib.msckf_device_pointer.argtypes = [vp, C.c_int]
    lib.msckf_device_pointer.restype = C.c_uint64
    lib.msckf_stream.argtypes = [vp]
    lib.msckf_stream.restype = vp
    lib.msckf_set_tracks.argtypes = [vp, _dp, _dp, _dp, _ip, _ip]
    lib.msckf_set_tracks.restype = C.c_int
    lib.msckf_run_select.argtypes = [vp, C.POINTER(SelectParamsC)]
    lib.msckf_run_select.restype = C.c_int
    lib.msckf_clear_selection.argtypes = [vp]
    lib.msckf_clear_selection.restype = C.c_int
    lib.msckf_get_selection.argtypes = [vp, _up, _dp, _dp, _dp]
    lib.msckf_get_selection.restype = C.c_int
    lib.msckf_run_associate.argtypes = [vp, C.POINTER(AssocParamsC), _dp, _up, _ip]
    lib.msckf_run_associate.restype = C.c_int
    lib.msckf_propagate.argtypes = [vp, _dp, _dp]
    lib.msckf_propagate.restype = C.c_int
    lib.msckf_augment.argtypes = [vp, _dp, _dp, _dp]
    lib.msckf_augment.restype = C.c_int
    lib.msckf_remove_clones.argtypes = [vp, C.c_int32, _ip]
    lib.msckf_remove_clones.restype = C.c_int
    lib.msckf_set_poses.argtypes = [vp, _dp, _dp, _dp, _dp]
    lib.msckf_set_poses.restype = C.c_int
    lib.msckf_get_covariance.argtypes = [vp, _dp, _ip]
    lib.msckf_get_covariance.restype = C.c_int
    lib.msckf_debug_split.argtypes = [vp, C.POINTER(C.c_int32)]
    lib.msckf_debug_split.restype = C.c_int
    lib.msckf_debug_set_rem_direct_rows.argtypes = [vp, C.c_int32]
    lib.msckf_debug_set_rem_direct_rows.restype = C.c_int
    lib.msckf_debug_time_select.argtypes = [vp, C.c_int32, C.POINTER(C.c_float)]
    lib.msckf_debug_time_select.restype = C.c_int
    _lib = lib
    return lib


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


_c_char, _addressof = C.c_char, C.addressof


def dptr(a):
    # (the address through the buffer protocol: 0.3 us; `a.__array_interface__` builds a dict per call, 0.9 us, `a.ctypes` a helper
    #  object, 1 us -- times 17 arguments of the drop-in call.  Read-only and empty arrays do not export a writable buffer.)
    try:
        return _addressof(_c_char.from_buffer(a))
    except (TypeError, ValueError):
        return a.__array_interface__["data"][0]


iptr = dptr
uptr = dptr


class EngineError(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f"msckf engine error {code}: {text}")
        self.code = code
