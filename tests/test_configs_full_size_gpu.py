"""BASELINE configs[3] and configs[4] at their stated size on ONE MI355X (the driver's GPU suite has one GPU; the 8-GPU
forms shard the same lists by target, tests/test_sharding.py, tests/test_bench_sharded.py).

configs[3]: 10,000 scans, 1,000,000 candidate pairs -- the pair list Solver::SolveAutoLC builds
(/root/reference/src/optimization/solver.cc:676-700) -- through bench.py's own sharded step at world size 1.  A list of
more than 131,072 pairs goes through the matcher in rounds, the candidates of a round on the library's helper stream
beside the next round's bounds (nhip_bnb.hip, launch_csm_bnb): the production geometry of that path.
configs[4]: the end-to-end loop on a 10,000-scan bag (examples/slam_loop.py)."""
import ctypes as C
import math
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from nautilus_amd import _lib, csm  # noqa: E402
from oracle import oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu
DEG = math.radians(1.0)


def test_config3_full_list_takes_the_overlapped_rounds_and_equals_every_add(gpu):
    import torch
    import bench
    from nautilus_amd import sharding
    wl = bench.Workload("config4", 1)
    assert wl.n_pairs == 1000000 and wl.n_scans == 10000
    # self pairs and duplicates replace the first pairs of the list (same targets: the partition is unchanged)
    wl.src[:50] = wl.tgt[:50]
    wl.th0[:50] = 0.0
    wl.src[50:100], wl.th0[50:100] = wl.src[100:150], wl.th0[100:150]
    wl.tgt[50:100] = wl.tgt[100:150]
    plan = sharding.ShardPlan(wl.src, wl.tgt, wl.th0, 1)
    dev = torch.device("cuda", 0)
    m = bench.HipMatcher(wl, plan.shard(0), dev, 16, no_image=False)  # (the every-add cross-check below reads the image)
    lib = _lib.load()
    lib.nhip_timing_reset()
    lib.nhip_timing_enable(1)
    elapsed, full = bench.run_sharded(plan, 0, 1, dev, m, steps=1, warmup=0)
    lib.nhip_timing_enable(0)
    # the form the list took: rounds of 131,072 pairs, two rounds' state, candidates on the helper stream
    info = csm.last_launch()
    print(info, "%.0f ms for the step (first build included)" % (1e3 * elapsed))
    assert info["form_id"] == 3 and info["pairs_per_round"] == 131072 and info["rounds"] == 8, info
    assert info["rounds_of_state"] >= 2 and info["short_scans"] and info["n_pairs"] == 1000000, info
    n_b, n_c = C.c_int32(0), C.c_int32(0)
    ms = C.c_double(0)
    _lib.check(lib.nhip_timing_get(_lib.NHIP_TIMER_CSM_BOUNDS, C.byref(ms), C.byref(n_b)))
    _lib.check(lib.nhip_timing_get(_lib.NHIP_TIMER_CSM_CAND, C.byref(ms), C.byref(n_c)))
    assert n_b.value == 8 and n_c.value == 8
    rec_sh, sums_sh = m.records()                         # shard order = the matcher's order (no weights: by target)
    rec = full.cpu().numpy().view(csm.MATCH_DTYPE).reshape(-1)
    sums = np.empty(wl.n_pairs, np.int32)
    sums[plan.order] = sums_sh.cpu().numpy()

    # every add of the exhaustive definition (csm_correlate16_kernel; these grids carry no skip map: every strip) on
    # 20,000 consecutive pairs of the matcher's list that straddle the boundary between its first two rounds
    a, b = 131072 - 10000, 131072 + 10000
    n = b - a
    ex = csm.search_spec(61, 81, 81, DEG, exhaustive=True, short_scans=True)
    keys = torch.empty(n, dtype=torch.int64, device=dev)
    out = torch.empty((n, 4), dtype=torch.int32, device=dev)
    osum = torch.empty(n, dtype=torch.int32, device=dev)
    _lib.check(lib.nhip_csm_match_dev(m.d_xy.data_ptr(), m.d_off.data_ptr(), m.n_scans, m.d_grids.data_ptr(), m.n_targets, C.byref(m.spec),
                                      m.d_src.data_ptr() + 4 * a, m.d_slot.data_ptr() + 4 * a, m.d_rot0.data_ptr() + 16 * a,
                                      m.d_delta.data_ptr(), None, n, C.byref(ex), keys.data_ptr(), out.data_ptr(),
                                      osum.data_ptr(), None, 0, m.sp))
    torch.cuda.synchronize()
    assert torch.equal(out, rec_sh[a:b]), "branch and bound (overlapped rounds) differs from the kernel that performs every add"
    assert torch.equal(osum, sums_sh[a:b])
    m.free_grids()

    # properties that hold at any size
    assert np.all(rec["itheta"][:50] == 30) and np.all(rec["ix"][:50] == 40) and np.all(rec["iy"][:50] == 40)
    assert rec[50:100].tobytes() == rec[100:150].tobytes()
    Lf, step = math.log(1e-10), -math.log(1e-10) / 65535.0
    assert np.array_equal(rec["score"], (Lf + step * sums.astype(np.float64) / 1081.0).astype(np.float32))
    assert np.all((rec["itheta"] >= 0) & (rec["itheta"] < 61) & (rec["ix"] >= 0) & (rec["ix"] < 81) & (rec["iy"] >= 0) & (rec["iy"] < 81))
    # oracle sample: 40 pairs anywhere in the list + pairs around every round boundary
    rng = np.random.default_rng(4)
    edge = plan.order[np.r_[131071, 131072, 262143, 262144, 917503, 917504]]
    sel = np.r_[0:2, 60:62, rng.choice(wl.n_pairs, 30, replace=False), edge]
    ospec, oss = O.grid_spec(cell_bits=16), O.search_spec(61, 81, 81, DEG)
    ids = np.unique(wl.tgt[sel])
    og = O.grid_build_batch(wl.xy, wl.off, ids, ospec)
    want = O.csm_match_batch(wl.xy, wl.off, og, ospec, wl.src[sel], np.searchsorted(ids, wl.tgt[sel]), wl.th0[sel], oss)
    for f in ("itheta", "ix", "iy"):
        assert np.array_equal(rec[f][sel], want[f]), f
    assert np.array_equal(sums[sel], want["sum"])
    assert np.array_equal(rec["score"][sel], want["score"].astype(np.float32))


def test_config4_loop_on_a_10000_scan_bag(gpu):
    """The end-to-end loop at configs[4]'s size on one GPU: counts and errors of the builder's runs
    (profiles/r03_slam_loop_10000_scans.json: 3,275 candidate pairs, 3,239 accepted, 3.2 cm relative error), and the
    wall-clock by owner -- this repo's path on the GPU beside the host's sparse solves (the reference's Ceres)."""
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import slam_loop
    out = slam_loop.run(n_scans=10000, window=10)  # (LCCandidateFilter's own threshold, 0.70: `python examples/slam_loop.py --scans 10000`)
    print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in out.items()})
    print("GPU path %.2f s | host sparse solves %.2f s | other host work %.2f s (marshalling %.2f, sparse assembly %.2f, HITL "
          "selection %.2f, harness %.2f) | total %.2f s"
          % (out["gpu_path_s"], out["host_solver_s"], out["host_other_s"], out["marshal_s"], out["host_assembly_s"],
             out["hitl_select_s"], out["harness_s"], out["t_total_s"]))
    # the path's boundary marshalling (block lists, HITL input arrays, records -> transforms) is not where the time goes either
    assert out["marshal_s"] < 1.0, out["marshal_s"]
    assert out["icp_correspondences"] > 90e6
    assert out["lc_candidate_scans"] == 153 and out["lc_candidates"] == 3275 and out["lc_accepted"] == 3239
    assert out["lc_rel_err_m"] < 0.04
    assert out["err_lc_m"] < 0.04 and out["err_hitl_m"] < 0.04 and out["err_odometry_m"] > 0.5
    assert out["hitl_points"] > 1000000
    # the path is not where the time goes: the sparse solves are
    assert out["gpu_path_s"] < 0.5 * out["host_solver_s"]
