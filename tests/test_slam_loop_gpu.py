"""End-to-end: ICP window solve + GPU loop closure + re-solve on a 1.5-lap synthetic bag.
Exercises K1-K5 and the normal-equation reduction together; the assertion is geometric (trajectory
error against ground truth), so a silent error in any stage shows up."""
import pytest

pytestmark = pytest.mark.gpu


def test_loop_closure_reduces_trajectory_error(gpu):
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    import slam_loop
    out = slam_loop.run(n_scans=320, window=5)
    print(out)
    assert out["icp_correspondences"] > 100000
    assert out["lc_accepted"] >= 10
    assert out["lc_rel_err_m"] < 0.08           # matcher recovers the relative transform to about a cell
    assert out["err_icp_m"] < 0.25 * out["err_odometry_m"]   # growing-window ICP on point-to-plane residuals
    assert out["err_lc_m"] < 0.25 * out["err_odometry_m"]    # constraints at one-cell (5 cm) resolution do no harm


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
    H, _, _ = pg._assemble(pg.poses, research=False)
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
    s = hostside.chi_square_score(np.eye(2) * 0.01, [0.0, 0.0], [1.0, 2.0])
    assert abs(s - 500.0) < 1e-2
