"""End-to-end: ICP window solve + GPU loop closure + re-solve on a 1.5-lap synthetic bag.
Exercises K1-K5 and the normal-equation reduction together; the assertion is geometric (trajectory
error against ground truth), so a silent error in any stage shows up."""
import pytest

pytestmark = pytest.mark.gpu


def _slam_loop():
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    import slam_loop
    return slam_loop


def test_loop_closure_and_hitl_reduce_trajectory_error(gpu):
    """BASELINE configs[4] shape on one GPU at test size: growing window 1..10 (solver.cc:335-356), candidate scans
    from the GPU scatter score + geometric pair gate, batched scan matching, constraints, re-solve, then a HITL
    message (two segments on one wall) handled as Solver::HitlCallback does (solver.cc:479-559)."""
    out = _slam_loop().run(n_scans=420, window=10, min_scatter_score=0.3)
    print(out)
    assert out["window"] == 10 and out["icp_correspondences"] > 500000
    assert out["lc_candidate_scans"] >= 10 and out["lc_candidates"] >= 5 and out["lc_accepted"] >= 3
    assert out["lc_rel_err_m"] < 0.08           # matcher recovers the relative transform to about a cell
    assert out["err_icp_m"] < 0.25 * out["err_odometry_m"]   # growing-window ICP on point-to-plane residuals
    assert out["err_lc_m"] < 0.25 * out["err_odometry_m"]    # constraints at one-cell (5 cm) resolution do no harm
    # the HITL constraint found the poses that see the marked wall (with the drift already below the 5 cm line width
    # both segments select the same points, and a point on line a is not tested against line b: solver.cc:497-503),
    # its blocks moved the shared line pose, without harm
    assert out["hitl_line_a_poses"] + out["hitl_line_b_poses"] >= 50 and out["hitl_points"] >= 2000
    assert any(abs(v) > 1e-6 for v in out["hitl_chosen_line_pose"])
    assert out["err_hitl_m"] < 0.25 * out["err_odometry_m"]


def test_same_loop_on_the_cpu_backend_agrees(gpu):
    """The loop driven through the oracle's CPU backend (tests / bench only) and through the product: the same
    host code, the same correspondences and candidate pairs, trajectories equal to solver precision."""
    from oracle.cpu_backend import OracleBackend
    sl = _slam_loop()
    kw = dict(n_scans=200, window=3, min_scatter_score=0.3, cell_bits=8, spacing=0.4)  # 1.3 laps: the loop closes
    a = sl.run(**kw)
    b = sl.run(backend=OracleBackend(), **kw)
    print(a, b)
    assert a["backend"] == "hip" and b["backend"] == "oracle"
    assert a["icp_correspondences"] == b["icp_correspondences"] and a["lc_candidates"] == b["lc_candidates"] > 0
    assert a["lc_accepted"] == b["lc_accepted"] and a["hitl_points"] == b["hitl_points"] > 0
    for k in ("err_icp_m", "err_lc_m", "err_hitl_m"):
        assert abs(a[k] - b[k]) < 1e-6, k


def test_cross_covariance_blocks_match_dense_inverse(gpu):
    """PoseGraph.cross_covariances (LCMatcher::GetCovarianceMatrix, lc_matcher.cc:28-46) against a dense
    inverse of the same normal matrix, and the chi-square gate built on it (lc_matcher.cc:48-74)."""
    import numpy as np
    from nautilus_amd import _lib, csm, hostside, posegraph, synth
    bag = synth.SynthBag(40, dense=True)
    xy, off = csm.pack_scans(bag.scans)
    nrm = np.concatenate(bag.normals).astype(np.float32)
    pg = posegraph.PoseGraph(xy, nrm, off, bag.odom, window=3, kind=_lib.NHIP_LIDAR_NORMAL)
    pg.solve(iterations=3)
    pairs = [(30, 5), (5, 30), (12, 13), (1, 39), (7, 0)]
    got = pg.cross_covariances(pairs)
    H, _, _ = pg._assemble(pg.poses, pg._lines(), research=False)
    Hd = H.toarray()
    for (s_, t_), g in zip(pairs, got):
        gauge = max(min(s_, t_) - 1, 0)
        if gauge in (s_, t_):
            assert not g.any()
            continue
        free = np.r_[0:3 * gauge, 3 * gauge + 3:3 * pg.n]
        inv = np.linalg.inv(Hd[np.ix_(free, free)])
        pos = -np.ones(3 * pg.n, int)
        pos[free] = np.arange(len(free))
        want = inv[np.ix_([pos[3 * s_], pos[3 * s_ + 1]], [pos[3 * t_], pos[3 * t_ + 1]])]
        assert np.allclose(g, want, rtol=1e-4, atol=1e-9)
    matches = hostside.lc_possible_matches(30, [5, 12, 30, 39], pg.poses, pg.cross_covariances)
    assert 30 not in matches and set(matches) <= {5, 12, 39}
    # the same walk with the scores taken on the GPU (nhip_lc_chi_square_gate) keeps the same scans
    assert hostside.lc_possible_matches(30, [5, 12, 30, 39], pg.poses, pg.cross_covariances, backend=pg.backend) == matches
    s = hostside.chi_square_score(np.eye(2) * 0.01, [0.0, 0.0], [1.0, 2.0])
    assert abs(s - 500.0) < 1e-2
