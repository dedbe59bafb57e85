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


def _loop_rank(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "4"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import slam_loop
        from oracle.cpu_backend import OracleBackend
        out = slam_loop.run(backend=OracleBackend(), rank=rank, world=world, device="cpu", **_LOOP_KW)
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


_LOOP_KW = dict(n_scans=130, window=1, iterations=1, hitl=False, min_scatter_score=0.3, cell_bits=8, spacing=0.55)


def test_loop_with_two_ranks_over_gloo_matches_one_rank():
    """BASELINE configs[4]'s shape on CPU: `examples/slam_loop.py` under torch.distributed with world size 2 (gloo) --
    the window solve replicated, the loop-closure pairs sharded by target across the ranks, matched, all-gathered
    (nautilus_amd/sharding.py) -- ends on every rank with the trajectory of the one-rank run."""
    import socket
    import torch.multiprocessing as mp
    import slam_loop
    from oracle.cpu_backend import OracleBackend
    one = slam_loop.run(backend=OracleBackend(), **_LOOP_KW)
    assert one["lc_candidates"] >= 2 and one["lc_accepted"] >= 1, one
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_loop_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r in (0, 1):
        for k in ("lc_candidates", "lc_accepted", "icp_correspondences"):
            assert res[r][k] == one[k], (r, k)
        for k in ("err_icp_m", "err_lc_m", "lc_rel_err_m"):
            assert abs(res[r][k] - one[k]) < 1e-12, (r, k, res[r][k], one[k])
