"""The end-to-end loop (examples/slam_loop.py + nautilus_amd/posegraph.py) driven on CPU through the oracle's backend:
the host logic -- growing-window solve, candidate walk, pair gate, acceptance by csm_score_threshold, HITL selection
and the shared chosen_line_pose block -- runs without a GPU; the product backend is exercised by the -m gpu twin
(tests/test_slam_loop_gpu.py), which also checks that both backends give the same trajectory."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examples"))


def test_loop_with_hitl_on_the_oracle_backend():
    import slam_loop
    from oracle.cpu_backend import OracleBackend
    out = slam_loop.run(n_scans=270, window=2, iterations=2, backend=OracleBackend(), min_scatter_score=0.3, cell_bits=8)
    assert out["backend"] == "oracle" and out["icp_correspondences"] > 400000
    assert out["err_icp_m"] < out["err_odometry_m"]
    # a lap and a bit: a handful of candidate pairs; false matches (true offset outside the +-2 m window) fall below
    # csm_score_threshold = -5 (default_config.lua:84-85) and are not turned into constraints
    assert out["lc_candidate_scans"] >= 8 and 1 <= out["lc_candidates"] <= 20 and out["lc_accepted"] <= out["lc_candidates"]
    assert out["err_lc_m"] < 0.6 * out["err_odometry_m"]
    # HITL: both groups of poses found on the marked wall, the shared line pose moved, the trajectory improved
    assert out["hitl_line_a_poses"] >= 20 and out["hitl_line_b_poses"] >= 20 and out["hitl_points"] > 5000
    assert abs(out["hitl_chosen_line_pose"][0]) > 1e-4
    assert out["err_hitl_m"] < out["err_lc_m"]
