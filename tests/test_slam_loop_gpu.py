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
