"""ctypes binding of libnautilus_hip.so (include/nautilus_hip.h).

The library is the product: there is no Python or CPU fallback.  If the shared object is
missing this module raises at import of the symbol table; if no GPU is visible every
compute entry point returns NHIP_ERR_NODEV and `check()` raises NhipError.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# NHIP_LIB lets kernel A/B experiments point at another build of the same library
LIB_PATH = os.environ.get("NHIP_LIB") or os.path.join(_HERE, "lib", "libnautilus_hip.so")

NHIP_OK, NHIP_ERR_ARG, NHIP_ERR_NODEV, NHIP_ERR_HIP, NHIP_ERR_ALLOC, NHIP_ERR_STATE = 0, -1, -2, -3, -4, -5
NHIP_LIDAR_NORMAL, NHIP_LIDAR_POINT = 0, 1
NHIP_SEARCH_EXHAUSTIVE, NHIP_SEARCH_DENSE, NHIP_SEARCH_SHORT_SCANS, NHIP_SEARCH_EXACT_SCORE, NHIP_SEARCH_LATENCY = 1, 2, 4, 8, 16
NHIP_SHORT_SCAN_POINTS = 1088
NHIP_GRID_SKIP_MAP, NHIP_GRID_NO_IMAGE = 1, 2
NHIP_TIMER_CSM, NHIP_TIMER_GRID, NHIP_TIMER_RESID, NHIP_TIMER_CORR, NHIP_TIMER_NORMEQ, NHIP_TIMER_GRID_CLEAR = 0, 1, 2, 3, 4, 5
NHIP_TIMER_CSM_BOUNDS, NHIP_TIMER_CSM_CAND, NHIP_TIMER_EXACT_SCORE = 6, 7, 8


class NhipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("nautilus_hip error %d: %s" % (code, msg))
        self.code = code


class GridSpec(C.Structure):
    _fields_ = [("range", C.c_double), ("res", C.c_double), ("sigma", C.c_double),
                ("floor_p", C.c_double), ("max_shift", C.c_int32), ("cell_bits", C.c_int32),
                ("flags", C.c_int32), ("reserved", C.c_int32)]


class GridLayout(C.Structure):
    _fields_ = [("side", C.c_int32), ("pad", C.c_int32), ("pitch", C.c_int32), ("rows", C.c_int32),
                ("blur_radius", C.c_int32), ("cell_bytes", C.c_int32), ("tap_sum", C.c_int64),
                ("grid_bytes", C.c_int64), ("score_floor", C.c_double), ("score_step", C.c_double),
                ("skip_bytes", C.c_int64), ("slot_bytes", C.c_int64), ("pool_bytes", C.c_int64),
                ("pool_pitch", C.c_int32), ("pool_rows", C.c_int32), ("pool4_bytes", C.c_int64),
                ("pool4_pitch", C.c_int32), ("pool4_rows", C.c_int32), ("hi_bytes", C.c_int64),
                ("hi_pitch", C.c_int32), ("hits_pitch", C.c_int32), ("hits_bytes", C.c_int64)]


class Search(C.Structure):
    _fields_ = [("n_theta", C.c_int32), ("nx", C.c_int32), ("ny", C.c_int32), ("flags", C.c_int32),
                ("theta_step", C.c_double)]


class CsmParams(C.Structure):
    _fields_ = [("scanner_range", C.c_double), ("trans_range", C.c_double), ("low_res", C.c_double),
                ("high_res", C.c_double), ("sigma", C.c_double), ("floor_p", C.c_double), ("cell_bits", C.c_int32),
                ("reserved", C.c_int32)]


class Match(C.Structure):
    _fields_ = [("itheta", C.c_int32), ("ix", C.c_int32), ("iy", C.c_int32), ("score", C.c_float)]


_vp, _i32, _i64, _f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_double
_P = C.POINTER

# name -> (restype, argtypes); the list is also what tests check against the header.
PROTOTYPES = {
    "nhip_init": (C.c_int, [_P(C.c_int)]),
    "nhip_set_device": (C.c_int, [C.c_int]),
    "nhip_last_error": (C.c_char_p, []),
    "nhip_version": (C.c_char_p, []),
    "nhip_grid_layout": (C.c_int, [_P(GridSpec), _P(GridLayout)]),
    "nhip_grids_bytes": (_i64, [_P(GridSpec), _i64]),
    "nhip_grid_workspace_bytes": (_i64, [_P(GridSpec), _i32]),
    "nhip_grid_tables": (C.c_int, [_P(GridSpec), _vp, _vp]),
    "nhip_csm_rot0": (C.c_int, [_vp, _vp, _i32, _vp]),
    "nhip_csm_delta_table": (C.c_int, [_P(Search), _vp]),
    "nhip_match_to_transform": (C.c_int, [_P(Match), _P(GridSpec), _P(Search), _f64, _i32, _i32,
                                          _P(C.c_float), _P(C.c_float), _P(C.c_float)]),
    "nhip_score_from_sum": (_f64, [_P(GridSpec), _i64, _i32]),
    "nhip_dev_status": (C.c_int, [_vp, _P(_i32)]),
    "nhip_host_phases": (C.c_int, [_P(_f64)]),
    "nhip_device_pool_configure": (C.c_int, [_i64]),
    "nhip_device_pool_release": (C.c_int, []),
    "nhip_device_pool_stats": (C.c_int, [_P(_i64), _P(_i64)]),
    "nhip_grid_build_dev": (C.c_int, [_vp, _vp, _i32, _vp, _i32, _P(GridSpec), _vp, _vp, _i64, _vp]),
    "nhip_grid_rebuild_dev": (C.c_int, [_vp, _vp, _i32, _vp, _i32, _P(GridSpec), _vp, _vp, _i64, _vp]),
    "nhip_csm_match_dev": (C.c_int, [_vp, _vp, _i32, _vp, _i32, _P(GridSpec), _vp, _vp, _vp, _vp, _vp, _i32,
                                     _P(Search), _vp, _vp, _vp, _vp, _i64, _vp]),
    "nhip_csm_workspace_bytes": (_i64, [_i32]),
    "nhip_csm_last_launch": (C.c_int, [_P(_i32)]),
    "nhip_csm_get_transformation_info": (C.c_int, [_vp]),
    "nhip_bnb_stats": (C.c_int, [_P(C.c_uint64), _P(C.c_uint64)]),
    "nhip_bnb_stats_per_pair": (C.c_int, [_vp, _i32]),
    "nhip_bnb_timeline": (C.c_int, [_vp, _i32]),
    "nhip_bnb_timeline_candidates": (C.c_int, [_vp, _i32]),
    "nhip_bnb_stats_levels": (C.c_int, [_P(C.c_uint64)]),
    "nhip_csm_scores_dev": (C.c_int, [_vp, _vp, _i32, _vp, _i32, _P(GridSpec), _i32, _i32, _vp, _vp, _i32, _i32,
                                      _P(Search), _vp, _vp]),
    "nhip_resid_lidar_dev": (C.c_int, [C.c_int, _vp, _vp, _i64, _vp, _vp, _i32, _vp, _i32, _vp,
                                       _vp, _vp, _vp, _vp]),
    "nhip_resid_lidar_normal_eq_dev": (C.c_int, [C.c_int, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _vp, _vp, _vp]),
    "nhip_pose_affines": (C.c_int, [_vp, _i32, _vp]),
    "nhip_corr_search_dev": (C.c_int, [_vp, _vp, _vp, _i32, _vp, _vp, _i32, _vp, C.c_float, _vp, _vp, _vp, _vp]),
    "nhip_corr_search_normals_dev": (C.c_int, [_vp, _vp, _vp, _i32, _vp, _vp, _i32, _vp, C.c_float, C.c_float, _vp, _vp, _vp,
                                               _vp]),
    "nhip_corr_compact_dev": (C.c_int, [_vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    "nhip_resid_point_to_line_dev": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _vp, _i32, _vp, _i32, _vp, _i32,
                                               _vp, _vp, _vp, _vp]),
    "nhip_resid_odometry_dev": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _f64, _f64, _vp, _i32, _vp, _vp,
                                          _vp, _vp]),
    "nhip_scans_upload": (C.c_int, [_vp, _vp, _i32, _P(_vp)]),
    "nhip_scans_free": (C.c_int, [_vp]),
    "nhip_grids_build": (C.c_int, [_vp, _vp, _i32, _P(GridSpec), _P(_vp)]),
    "nhip_grids_free": (C.c_int, [_vp]),
    "nhip_grids_was_rebuilt": (C.c_int, [_vp]),
    "nhip_grids_download": (C.c_int, [_vp, _i32, _vp]),
    "nhip_grids_download_skip_map": (C.c_int, [_vp, _i32, _vp]),
    "nhip_grids_download_hi_plane": (C.c_int, [_vp, _i32, _vp]),
    "nhip_grids_download_hi_plane_copy": (C.c_int, [_vp, _i32, _i32, _vp]),
    "nhip_grids_download_tiled16": (C.c_int, [_vp, _i32, _vp]),
    "nhip_grids_download_pool": (C.c_int, [_vp, _i32, _vp]),
    "nhip_grids_download_pool4": (C.c_int, [_vp, _i32, _vp]),
    "nhip_grids_download_hits": (C.c_int, [_vp, _i32, _vp]),
    "nhip_csm_match": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _P(Search), _vp, _vp]),
    "nhip_csm_scores": (C.c_int, [_vp, _vp, _i32, _i32, _f64, _i32, _i32, _P(Search), _vp]),
    "nhip_lc_scatter_scores_dev": (C.c_int, [_vp, _vp, _i32, _vp, _vp]),
    "nhip_lc_pair_gate_dev": (C.c_int, [_vp, _i32, _vp, _i32, _f64, _i32, _vp, _vp]),
    "nhip_lc_chi_square_gate_dev": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _i32, _f64, _vp, _vp, _vp]),
    "nhip_lc_chi_square_gate": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _i32, _f64, _vp, _vp]),
    "nhip_lc_scatter_scores": (C.c_int, [_vp, _vp]),
    "nhip_lc_pair_gate": (C.c_int, [_vp, _i32, _vp, _i32, _f64, _i32, _vp]),
    "nhip_csm_cache_configure": (C.c_int, [_i64]),
    "nhip_csm_cache_clear": (C.c_int, []),
    "nhip_csm_cache_stats": (C.c_int, [_P(_i64), _P(_i64), _P(_i64), _P(_i64)]),
    "nhip_csm_get_transformation": (C.c_int, [_P(CsmParams), _vp, _i32, _vp, _i32, _f64, _f64, _f64, _P(_f64),
                                              _P(C.c_float), _P(C.c_float), _P(C.c_float)]),
    "nhip_resid_batch_create": (C.c_int, [C.c_int, _vp, _vp, _vp, _vp, _i32, _i32, _P(_vp)]),
    "nhip_resid_batch_eval": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "nhip_resid_batch_eval_compact": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "nhip_resid_batch_eval_q": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "nhip_resid_jacobians_from_q": (C.c_int, [C.c_int, _vp, _vp, _vp, _i64, _vp, _vp]),
    "nhip_resid_batch_eval_block": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    "nhip_resid_batch_free": (C.c_int, [_vp]),
    "nhip_host_alloc": (C.c_int, [C.c_size_t, _P(_vp)]),
    "nhip_host_free": (C.c_int, [_vp]),
    "nhip_resid_odometry": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _f64, _f64, _vp, _i32, _vp, _vp, _vp]),
    "nhip_resid_point_to_line": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _vp, _i32, _vp, _i32, _vp, _i32,
                                           _vp, _vp, _vp]),
    "nhip_allgather_matches": (C.c_int, [_vp, _vp, _i32, _vp, _vp]),
    "nhip_timing_enable": (C.c_int, [C.c_int]),
    "nhip_timing_reset": (C.c_int, []),
    "nhip_timing_get": (C.c_int, [C.c_int, _P(_f64), _P(_i32)]),
}

_lib = None


def load():
    """Load the shared library (once) and bind every prototype.  Fails loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "nautilus_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C nautilus_amd/csrc`; there is no CPU fallback" % LIB_PATH)
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 (same SONAME as
    # /opt/rocm's).  If ours were resolved first, torch would later bind to the system runtime
    # and see no GPUs.  Importing torch first makes both share torch's copy; without torch
    # (a C++ host) the library binds to /opt/rocm as linked.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != NHIP_OK:
        raise NhipError(rc, load().nhip_last_error().decode("utf-8", "replace"))


def device_count():
    n = C.c_int(0)
    load().nhip_init(C.byref(n))
    return n.value


def ptr(a):
    """void* of a numpy array (None -> NULL)."""
    return None if a is None else a.ctypes.data_as(C.c_void_p)
