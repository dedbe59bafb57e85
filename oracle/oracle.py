"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the product
package (nautilus_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")


class GridSpec(C.Structure):
    _fields_ = [("range", C.c_double), ("res", C.c_double), ("sigma", C.c_double), ("floor_p", C.c_double),
                ("cell_bits", C.c_int32), ("reserved", C.c_int32)]


class SearchSpec(C.Structure):
    _fields_ = [("n_theta", C.c_int32), ("nx", C.c_int32), ("ny", C.c_int32), ("theta_step", C.c_double)]


class OMatch(C.Structure):
    _fields_ = [("itheta", C.c_int32), ("ix", C.c_int32), ("iy", C.c_int32), ("sum", C.c_int32),
                ("score", C.c_double)]


OMATCH_DTYPE = np.dtype([("itheta", "<i4"), ("ix", "<i4"), ("iy", "<i4"), ("sum", "<i4"), ("score", "<f8")])
assert OMATCH_DTYPE.itemsize == C.sizeof(OMatch)
OMATCH_F64_DTYPE = np.dtype([("itheta", "<i4"), ("ix", "<i4"), ("iy", "<i4"), ("pad", "<i4"), ("score", "<f8")])

_lib = None
_vp, _i32, _f64 = C.c_void_p, C.c_int32, C.c_double


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    lib = C.CDLL(LIB_PATH)
    lib.orc_grid_side.restype = _i32
    lib.orc_grid_side.argtypes = [_f64, _f64]
    lib.orc_score_floor.restype = _f64
    lib.orc_score_floor.argtypes = [C.POINTER(GridSpec)]
    lib.orc_score_step.restype = _f64
    lib.orc_score_step.argtypes = [C.POINTER(GridSpec)]
    lib.orc_grid_build.argtypes = [_vp, _i32, C.POINTER(GridSpec), _vp]
    lib.orc_csm_match.argtypes = [_vp, _i32, _vp, C.POINTER(GridSpec), _f64, _i32, _i32,
                                  C.POINTER(SearchSpec), C.POINTER(OMatch)]
    lib.orc_csm_scores.argtypes = [_vp, _i32, _vp, C.POINTER(GridSpec), _f64, _i32, _i32,
                                   C.POINTER(SearchSpec), _vp]
    lib.orc_csm_match_batch.argtypes = [_vp, _vp, _vp, C.POINTER(GridSpec), _vp, _vp, _vp, _vp, _i32,
                                        C.POINTER(SearchSpec), _vp, _i32]
    lib.orc_grid_build_batch.argtypes = [_vp, _vp, _vp, _i32, C.POINTER(GridSpec), _vp, _i32]
    lib.orc_num_threads.restype = C.c_int
    lib.orc_grid_build_f64.argtypes = [_vp, _i32, C.POINTER(GridSpec), _vp]
    lib.orc_csm_match_f64.argtypes = [_vp, _i32, _vp, C.POINTER(GridSpec), _f64, C.POINTER(SearchSpec), _vp, _vp]
    lib.orc_csm_match_f64_batch.argtypes = [_vp, _vp, _vp, _vp, _vp, _i32, C.POINTER(GridSpec), C.POINTER(SearchSpec),
                                            _vp, _vp, _vp, _i32]
    lib.orc_csm_pose_score_exact.argtypes = [_vp, _i32, _vp, _i32, C.POINTER(GridSpec), _f64, C.POINTER(SearchSpec), _i32, _i32,
                                             _i32, _i32, _i32, C.POINTER(_f64)]
    lib.orc_two_level_match.argtypes = [_vp, _i32, _vp, _i32, _f64, _f64, _f64, _f64, _f64, _f64, _f64, _f64, _f64,
                                        _i32, C.POINTER(_f64), C.POINTER(C.c_float), C.POINTER(C.c_float),
                                        C.POINTER(C.c_float)]
    lib.orc_lidar_batch_analytic.argtypes = [C.c_int, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32]
    lib.orc_dist_to_segment_f.restype = C.c_float
    lib.orc_dist_to_segment_f.argtypes = [C.c_float] * 6
    lib.orc_dist_to_segment_d.restype = _f64
    lib.orc_dist_to_segment_d.argtypes = [_f64] * 6
    lib.orc_lidar_block.argtypes = [C.c_int, _vp, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp]
    lib.orc_point_to_line_block.argtypes = [_vp, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp]
    lib.orc_odometry_block.argtypes = [_vp, C.c_float, _f64, _f64, _vp, _vp, _vp, _vp, _vp]
    lib.orc_lidar_batch.argtypes = [C.c_int, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32]
    lib.orc_pose_affines.argtypes = [_vp, _i32, _vp]
    lib.orc_pose_affines.restype = None
    lib.orc_corr_search_block.restype = _i32
    lib.orc_corr_search_block.argtypes = [_vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp, C.c_float, _vp, _vp]
    lib.orc_corr_search_batch.argtypes = [_vp, _vp, _vp, _vp, _vp, _i32, _vp, C.c_float, _vp, _vp, _vp, _i32]
    _lib = lib
    return lib


def pose_affines(poses):
    poses = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 3)
    out = np.zeros((len(poses), 4), dtype=np.float32)
    load().orc_pose_affines(_p(poses), len(poses), _p(out))
    return out


def corr_search_block(src_xy, src_nrm, tgt_xy, tgt_nrm, src_aff, tgt_aff, thr=0.25):
    """One block of Solver::GetPointToPointMatching: (rows (n, 8) float32, matched target indices)."""
    sx, sn, tx, tn = _f32(src_xy).reshape(-1, 2), _f32(src_nrm).reshape(-1, 2), _f32(tgt_xy).reshape(-1, 2), _f32(tgt_nrm).reshape(-1, 2)
    out = np.zeros((max(len(sx), 1), 8), dtype=np.float32)
    idx = np.zeros(max(len(sx), 1), dtype=np.int32)
    n = load().orc_corr_search_block(_p(sx), _p(sn), len(sx), _p(tx), _p(tn), len(tx), _p(_f32(src_aff)),
                                     _p(_f32(tgt_aff)), float(thr), _p(out), _p(idx))
    return out[:n].copy(), idx[:n].copy()


def corr_search_batch(xy, normals, offsets, block_src, block_tgt, pose_aff, thr=0.25, n_threads=0):
    """Padded output like the product: (corr (cap, 8), counts, cap_offsets)."""
    xy, normals = _f32(xy).reshape(-1, 2), _f32(normals).reshape(-1, 2)
    offsets = np.ascontiguousarray(offsets, dtype=np.int32)
    bs, bt = np.ascontiguousarray(block_src, dtype=np.int32), np.ascontiguousarray(block_tgt, dtype=np.int32)
    cap = np.zeros(len(bs) + 1, dtype=np.int64)
    cap[1:] = np.cumsum(offsets[bs + 1] - offsets[bs])
    corr = np.zeros((int(cap[-1]), 8), dtype=np.float32)
    counts = np.zeros(len(bs), dtype=np.int32)
    _chk(load().orc_corr_search_batch(_p(xy), _p(normals), _p(offsets), _p(bs), _p(bt), len(bs),
                                      _p(_f32(pose_aff)), float(thr), _p(cap), _p(corr), _p(counts), n_threads),
         "corr_search_batch")
    return corr, counts, cap


def corr_search_gated_batch(xy, normals, offsets, block_src, block_tgt, pose_aff, thr, min_abs_cosine, n_threads=0):
    """Solver::GetPointToNormalMatching (nearest target within thr with a similar normal); padded output."""
    xy, normals = _f32(xy).reshape(-1, 2), _f32(normals).reshape(-1, 2)
    offsets = np.ascontiguousarray(offsets, dtype=np.int32)
    bs, bt = np.ascontiguousarray(block_src, dtype=np.int32), np.ascontiguousarray(block_tgt, dtype=np.int32)
    cap = np.zeros(len(bs) + 1, dtype=np.int64)
    cap[1:] = np.cumsum(offsets[bs + 1] - offsets[bs])
    corr = np.zeros((int(cap[-1]), 8), dtype=np.float32)
    counts = np.zeros(len(bs), dtype=np.int32)
    fn = load().orc_corr_search_gated_batch
    fn.argtypes = [_vp, _vp, _vp, _vp, _vp, _i32, _vp, C.c_float, C.c_float, _vp, _vp, _vp, _i32]
    _chk(fn(_p(xy), _p(normals), _p(offsets), _p(bs), _p(bt), len(bs), _p(_f32(pose_aff)), float(thr),
            float(min_abs_cosine), _p(cap), _p(corr), _p(counts), n_threads), "corr_search_gated_batch")
    return corr, counts, cap


def scatter_matrix_scores(xy, offsets):
    """ComputeScatterMatrixScore (lc_candidate_filter.cc:35-51) of every scan."""
    xy = np.ascontiguousarray(xy, dtype=np.float32)
    offsets = np.ascontiguousarray(offsets, dtype=np.int32)
    out = np.zeros(len(offsets) - 1)
    fn = load().orc_scatter_matrix_scores
    fn.argtypes, fn.restype = [_vp, _vp, _i32, _vp], None
    fn(_p(xy), _p(offsets), len(offsets) - 1, _p(out))
    return out


def pair_gate(poses, candidates, max_range, min_separation):
    poses = np.ascontiguousarray(poses, dtype=np.float64)
    cand = np.ascontiguousarray(candidates, dtype=np.int32)
    flags = np.zeros((len(cand), len(cand)), dtype=np.uint8)
    fn = load().orc_pair_gate
    fn.argtypes, fn.restype = [_vp, _vp, _i32, _f64, _i32, _vp], None
    fn(_p(poses), _p(cand), len(cand), float(max_range), int(min_separation), _p(flags))
    return flags


def chi_square_gate(poses, pair_src, pair_tgt, cov, max_score=5000.0):
    """(scores float64, flags uint8) of LCMatcher's chi-square test for n (source, candidate) pairs
    (lc_matcher.cc:50-74); cov: (n, 2, 2) float32 cross-covariance blocks."""
    poses = np.ascontiguousarray(poses, dtype=np.float64)
    src = np.ascontiguousarray(pair_src, dtype=np.int32)
    tgt = np.ascontiguousarray(pair_tgt, dtype=np.int32)
    cov = np.ascontiguousarray(cov, dtype=np.float32).reshape(-1, 4)
    assert len(cov) == len(src) == len(tgt)
    scores, flags = np.zeros(len(src)), np.zeros(len(src), dtype=np.uint8)
    fn = load().orc_chi_square_gate
    fn.argtypes, fn.restype = [_vp, _vp, _vp, _vp, _i32, _f64, _vp, _vp], None
    fn(_p(poses), _p(src), _p(tgt), _p(cov), len(src), float(max_score), _p(scores), _p(flags))
    return scores, flags


def _p(a):
    return None if a is None else a.ctypes.data_as(_vp)


def _chk(rc, what):
    if rc != 0:
        raise RuntimeError("oracle %s failed: %d" % (what, rc))


def grid_spec(range_m=30.0, res=0.05, sigma=2.0, floor_p=1e-10, cell_bits=16):
    return GridSpec(float(range_m), float(res), float(sigma), float(floor_p), int(cell_bits), 0)


def _cell_dtype(gs):
    return np.uint16 if gs.cell_bits == 16 else np.uint8


def search_spec(n_theta, nx, ny, theta_step):
    return SearchSpec(int(n_theta), int(nx), int(ny), float(theta_step))


def grid_side(gs):
    return load().orc_grid_side(gs.range, gs.res)


def grid_build(points, gs):
    pts = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 2)
    S = grid_side(gs)
    out = np.zeros((S, S), dtype=_cell_dtype(gs))
    _chk(load().orc_grid_build(_p(pts), len(pts), C.byref(gs), _p(out)), "grid_build")
    return out


def grid_build_f64(points, gs):
    """Unquantised table of double log-likelihoods (test-only precision reference)."""
    pts = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 2)
    S = grid_side(gs)
    out = np.zeros((S, S), dtype=np.float64)
    _chk(load().orc_grid_build_f64(_p(pts), len(pts), C.byref(gs), _p(out)), "grid_build_f64")
    return out


def csm_match_f64(src_points, grid_f64, gs, theta0, ss, want_scores=False):
    pts = np.ascontiguousarray(src_points, dtype=np.float32).reshape(-1, 2)
    g = np.ascontiguousarray(grid_f64, dtype=np.float64)
    out = np.zeros(1, dtype=OMATCH_F64_DTYPE)
    vol = np.zeros((ss.n_theta, ss.nx, ss.ny)) if want_scores else None
    _chk(load().orc_csm_match_f64(_p(pts), len(pts), _p(g), C.byref(gs), float(theta0), C.byref(ss), _p(out), _p(vol)),
         "csm_match_f64")
    return (out[0], vol) if want_scores else out[0]


def pose_score_exact(src_points, tgt_points, gs, theta0, ss, k, ix, iy, origin=(0, 0)):
    """The exact score (unquantised log-likelihoods) of ONE pose of the lattice: what NHIP_SEARCH_EXACT_SCORE reports."""
    a = np.ascontiguousarray(src_points, dtype=np.float32).reshape(-1, 2)
    b = np.ascontiguousarray(tgt_points, dtype=np.float32).reshape(-1, 2)
    out = C.c_double(0.0)
    _chk(load().orc_csm_pose_score_exact(_p(a), len(a), _p(b), len(b), C.byref(gs), float(theta0), C.byref(ss), int(origin[0]),
                                         int(origin[1]), int(k), int(ix), int(iy), C.byref(out)), "csm_pose_score_exact")
    return out.value


def csm_match_f64_batch(xy, offsets, pair_src, pair_tgt, theta0, gs, ss, probe=None, n_threads=0):
    """Unquantised exhaustive match of every pair (its table is built and dropped per pair).  probe: (n, 3)
    lattice indices (itheta, ix, iy) at which the f64 score is also returned (e.g. a quantised argmax)."""
    xy = np.ascontiguousarray(xy, dtype=np.float32)
    offsets = np.ascontiguousarray(offsets, dtype=np.int32)
    ps, pt = np.ascontiguousarray(pair_src, dtype=np.int32), np.ascontiguousarray(pair_tgt, dtype=np.int32)
    th = np.ascontiguousarray(theta0, dtype=np.float64)
    out = np.zeros(len(ps), dtype=OMATCH_F64_DTYPE)
    pr = None if probe is None else np.ascontiguousarray(probe, dtype=np.int32).reshape(len(ps), 3)
    prs = None if probe is None else np.zeros(len(ps))
    _chk(load().orc_csm_match_f64_batch(_p(xy), _p(offsets), _p(ps), _p(pt), _p(th), len(ps), C.byref(gs), C.byref(ss),
                                        _p(out), _p(pr), _p(prs), n_threads), "csm_match_f64_batch")
    return (out, prs) if probe is not None else out


def two_level_match(pc_a, pc_b, rot_a, rot_b, rot_restriction, scanner_range=30.0, trans_range=2.0, low_res=0.3,
                    high_res=0.01, sigma=2.0, floor_p=1e-10, cell_bits=16):
    """(score, ((tx, ty), theta)) of the reference-shaped single-pair call (solver.cc:633-644)."""
    a = np.ascontiguousarray(pc_a, dtype=np.float32).reshape(-1, 2)
    b = np.ascontiguousarray(pc_b, dtype=np.float32).reshape(-1, 2)
    sc, tx, ty, th = C.c_double(0), C.c_float(0), C.c_float(0), C.c_float(0)
    _chk(load().orc_two_level_match(_p(a), len(a), _p(b), len(b), float(rot_a), float(rot_b), float(rot_restriction),
                                    float(scanner_range), float(trans_range), float(low_res), float(high_res),
                                    float(sigma), float(floor_p), int(cell_bits), C.byref(sc), C.byref(tx), C.byref(ty),
                                    C.byref(th)), "two_level_match")
    return sc.value, ((np.float32(tx.value), np.float32(ty.value)), np.float32(th.value))


def csm_match(src_points, grid, gs, theta0, ss, origin=(0, 0)):
    pts = np.ascontiguousarray(src_points, dtype=np.float32).reshape(-1, 2)
    g = np.ascontiguousarray(grid, dtype=_cell_dtype(gs))
    m = OMatch()
    _chk(load().orc_csm_match(_p(pts), len(pts), _p(g), C.byref(gs), float(theta0), int(origin[0]),
                              int(origin[1]), C.byref(ss), C.byref(m)), "csm_match")
    return m


def csm_scores(src_points, grid, gs, theta0, ss, origin=(0, 0)):
    pts = np.ascontiguousarray(src_points, dtype=np.float32).reshape(-1, 2)
    g = np.ascontiguousarray(grid, dtype=_cell_dtype(gs))
    out = np.zeros((ss.n_theta, ss.nx, ss.ny), dtype=np.int32)
    _chk(load().orc_csm_scores(_p(pts), len(pts), _p(g), C.byref(gs), float(theta0), int(origin[0]),
                               int(origin[1]), C.byref(ss), _p(out)), "csm_scores")
    return out


def grid_build_batch(xy, offsets, target_ids, gs, n_threads=0):
    xy = np.ascontiguousarray(xy, dtype=np.float32)
    offsets = np.ascontiguousarray(offsets, dtype=np.int32)
    target_ids = np.ascontiguousarray(target_ids, dtype=np.int32)
    S = grid_side(gs)
    out = np.zeros((len(target_ids), S, S), dtype=_cell_dtype(gs))
    _chk(load().orc_grid_build_batch(_p(xy), _p(offsets), _p(target_ids), len(target_ids), C.byref(gs),
                                     _p(out), n_threads), "grid_build_batch")
    return out


def csm_match_batch(xy, offsets, grids, gs, pair_src, pair_slot, theta0, ss, pair_origin=None, n_threads=0):
    xy = np.ascontiguousarray(xy, dtype=np.float32)
    offsets = np.ascontiguousarray(offsets, dtype=np.int32)
    grids = np.ascontiguousarray(grids, dtype=_cell_dtype(gs))
    pair_src = np.ascontiguousarray(pair_src, dtype=np.int32)
    pair_slot = np.ascontiguousarray(pair_slot, dtype=np.int32)
    theta0 = np.ascontiguousarray(theta0, dtype=np.float64)
    org = None if pair_origin is None else np.ascontiguousarray(pair_origin, dtype=np.int32)
    out = np.zeros(len(pair_src), dtype=OMATCH_DTYPE)
    _chk(load().orc_csm_match_batch(_p(xy), _p(offsets), _p(grids), C.byref(gs), _p(pair_src), _p(pair_slot),
                                    _p(theta0), _p(org), len(pair_src), C.byref(ss), _p(out), n_threads),
         "csm_match_batch")
    return out


_NUM_THREADS = None


def num_threads():
    """OpenMP threads available to the oracle (read once: the batch drivers call omp_set_num_threads)."""
    global _NUM_THREADS
    if _NUM_THREADS is None:
        _NUM_THREADS = load().orc_num_threads()
    return _NUM_THREADS


def dist_to_segment_f(p, a, b):
    return float(load().orc_dist_to_segment_f(p[0], p[1], a[0], a[1], b[0], b[1]))


def dist_to_segment_d(p, a, b):
    return float(load().orc_dist_to_segment_d(p[0], p[1], a[0], a[1], b[0], b[1]))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def lidar_block(kind, sp, tp, sn, tn, source_pose, target_pose, jac=(True, True)):
    sp, tp, sn, tn = _f32(sp), _f32(tp), _f32(sn), _f32(tn)
    n = len(sp)
    a = np.ascontiguousarray(source_pose, dtype=np.float64)
    b = np.ascontiguousarray(target_pose, dtype=np.float64)
    r = np.zeros(2 * n)
    j0 = np.zeros((2 * n, 3)) if jac[0] else None
    j1 = np.zeros((2 * n, 3)) if jac[1] else None
    _chk(load().orc_lidar_block(kind, _p(sp), _p(tp), _p(sn), _p(tn), n, _p(a), _p(b), _p(r), _p(j0), _p(j1)),
         "lidar_block")
    return r, j0, j1


def point_to_line_block(seg, pts, pose, line_pose, jac=(True, True)):
    seg, pts = _f32(seg).reshape(4), _f32(pts).reshape(-1, 2)
    n = len(pts)
    a = np.ascontiguousarray(pose, dtype=np.float64)
    b = np.ascontiguousarray(line_pose, dtype=np.float64)
    r = np.zeros(n)
    j0 = np.zeros((n, 3)) if jac[0] else None
    j1 = np.zeros((n, 3)) if jac[1] else None
    _chk(load().orc_point_to_line_block(_p(seg), _p(pts), n, _p(a), _p(b), _p(r), _p(j0), _p(j1)), "p2l")
    return r, j0, j1


def odometry_block(t_odom, r_odom, tw, rw, pose_i, pose_j, jac=(True, True)):
    t = _f32(t_odom).reshape(2)
    a = np.ascontiguousarray(pose_i, dtype=np.float64)
    b = np.ascontiguousarray(pose_j, dtype=np.float64)
    r = np.zeros(3)
    j0 = np.zeros((3, 3)) if jac[0] else None
    j1 = np.zeros((3, 3)) if jac[1] else None
    _chk(load().orc_odometry_block(_p(t), float(r_odom), float(tw), float(rw), _p(a), _p(b), _p(r), _p(j0),
                                   _p(j1)), "odometry")
    return r, j0, j1


def lidar_batch(kind, corr, block_offsets, block_src, block_tgt, poses, want_jac=True, n_threads=0, analytic=False):
    corr = _f32(corr).reshape(-1, 8)
    bo = np.ascontiguousarray(block_offsets, dtype=np.int32)
    bs = np.ascontiguousarray(block_src, dtype=np.int32)
    bt = np.ascontiguousarray(block_tgt, dtype=np.int32)
    poses = np.ascontiguousarray(poses, dtype=np.float64)
    n = len(corr)
    r = np.zeros(2 * n)
    j0 = np.zeros((2 * n, 3)) if want_jac else None
    j1 = np.zeros((2 * n, 3)) if want_jac else None
    fn = load().orc_lidar_batch_analytic if analytic else load().orc_lidar_batch
    _chk(fn(kind, _p(corr), _p(bo), _p(bs), _p(bt), len(bs), _p(poses), _p(r), _p(j0), _p(j1), n_threads), "lidar_batch")
    return r, j0, j1
