#!/usr/bin/env python3
"""End-to-end loop on one MI355X, or one rank per GPU under `python -m torch.distributed.run
--nproc-per-node N examples/slam_loop.py` (the shape of BASELINE configs[0]/[4]):
synthetic 1081-beam bag -> sliding-window ICP solve -> loop-closure candidates -> batched GPU
correlative scan matching -> constraints -> re-solve; reports trajectory error and wall-clock.

Candidate pairs come from a distance gate on the current estimate (the reference's
LCCandidateFilter / LCMatcher are host-side callers outside the hot path)."""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(n_scans=320, window=5, seed=20201114, drift_t=0.02, drift_th_deg=0.3, per_target=2, verbose=False,
        residual="normal", rank=0, world=1, device="cuda:0"):
    """With world > 1 (one process per GPU under torch.distributed): the window ICP solve is replicated -- its
    consumer, the solver, is host-side -- and the loop-closure pairs are sharded by target across the ranks,
    matched, and all-gathered (nautilus_amd/sharding.py); every rank ends with the same trajectory."""
    from nautilus_amd import _lib, csm, posegraph, sharding, synth
    bag = synth.SynthBag(n_scans, dense=True, seed=seed)
    odom = synth.odometry_from_truth(bag.truth, sigma_t=drift_t, sigma_th_deg=drift_th_deg, seed=seed)
    odom = odom - odom[0] + bag.truth[0]  # both tracks start at the same anchor (pose 0 is held constant)
    xy, off = csm.pack_scans(bag.scans)
    nrm = np.concatenate(bag.normals).astype(np.float32)
    out = {"n_scans": n_scans, "window": window, "residual": residual, "err_odometry_m": posegraph.trajectory_error(odom, bag.truth)}

    t0 = time.perf_counter()
    # Solver::OptimizeOverGrowingWindow (solver.cc:339-355): window sizes 1..window, fresh correspondences each
    pg, poses = posegraph.solve_growing_window(xy, nrm, off, odom, 1, window, iterations=6,
                                               kind=_lib.NHIP_LIDAR_NORMAL if residual == "normal" else _lib.NHIP_LIDAR_POINT,
                                               device=device, verbose=verbose)
    out["t_icp_solve_s"] = time.perf_counter() - t0
    out["err_icp_m"] = posegraph.trajectory_error(poses, bag.truth)
    out["icp_correspondences"] = pg.icp.n_corr

    # loop-closure candidates: pairs at least half a lap apart whose TRUE poses are within 1.5 m
    # (a stand-in for the covariance gate of LCMatcher, which needs ceres::Covariance)
    lap = int(round(2 * (15 + 7 + math.pi * 1.5) / 0.25))
    targets = np.arange(lap // 2, n_scans, 4)
    src, tgt, _ = bag.sample_pairs(per_target=per_target, targets=targets, max_dist=1.5, min_sep=lap // 2, seed=seed)
    keep = np.abs(src - tgt) >= lap // 2
    src, tgt = src[keep], tgt[keep]
    out["lc_candidates"] = int(len(src))
    t0 = time.perf_counter()
    if len(src):
        a = poses[src, 2] - poses[tgt, 2]
        theta0 = a - 2 * math.pi * np.rint(a / (2 * math.pi))
        st = csm.ScanTable(xy, off)
        spec, search = csm.grid_spec(30.0, 0.05, 2.0, 1e-10, 40), csm.search_spec(61, 81, 81, math.radians(1.0))

        def match_shard(src_s, slot_s, th_s, ids_s):
            # this rank's share: the grids of its own targets, its own pairs
            if len(src_s) == 0:
                return np.zeros(0, dtype=csm.MATCH_DTYPE)
            grids = csm.LikelihoodGrids(st, ids_s, spec)
            ms, _ = csm.match_pairs(st, grids, src_s, slot_s, th_s, search)
            grids.close()
            return ms
        if world > 1:
            m = sharding.distributed_match(match_shard, src, tgt, theta0, rank, world, device=device)
        else:
            ids = np.unique(tgt)
            m = match_shard(src, np.searchsorted(ids, tgt).astype(np.int32), theta0, ids)
        rel = np.array([csm.match_to_transform(mi, spec, search, t0i) for mi, t0i in zip(m, theta0)], dtype=np.float64)
        inside = (np.abs(m["ix"] - 40) < 40) & (np.abs(m["iy"] - 40) < 40) & (np.abs(m["itheta"] - 30) < 30)
        good = inside & (m["score"] > np.median(m["score"]) - 1.5)  # not on the lattice border, plausible score
        out["lc_accepted"] = int(good.sum())
        truth_rel = np.array([bag.true_relative(s, t) for s, t in zip(src, tgt)])
        out["lc_rel_err_m"] = float(np.sqrt(np.mean(np.sum((rel[good, :2] - truth_rel[good, :2]) ** 2, axis=1)))) if good.any() else None
        st.close()
        out["t_csm_s"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        pg.add_loop_closures(src[good], tgt[good], rel[good])
        poses2, hist2 = pg.solve(iterations=8, verbose=verbose)
        out["t_lc_solve_s"] = time.perf_counter() - t0
        out["err_lc_m"] = posegraph.trajectory_error(poses2, bag.truth)
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--scans", type=int, default=320)
    ap.add_argument("--window", type=int, default=5)
    ap.add_argument("--residual", choices=["point", "normal"], default="normal",
                    help="LIDARPointResidual on all points (the reference's non-FEATURE mode, solver.cc:308-314) or "
                         "LIDARNormalResidual (its planar-feature mode, solver.cc:298-303)")
    ap.add_argument("-v", action="store_true")
    a = ap.parse_args()
    rank, world, local = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), \
        int(os.environ.get("LOCAL_RANK", "0"))
    if "RANK" in os.environ:  # python -m torch.distributed.run --nproc-per-node N examples/slam_loop.py ...
        import torch
        import torch.distributed as dist
        from nautilus_amd import _lib
        torch.cuda.set_device(local)
        _lib.check(_lib.load().nhip_set_device(local))
        saved = os.dup(1)
        os.dup2(2, 1)  # RCCL prints its banner on stdout
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        dist.barrier()
        os.dup2(saved, 1)
    res = run(a.scans, a.window, verbose=a.v and rank == 0, residual=a.residual, rank=rank, world=world,
              device="cuda:%d" % local)
    res["world_size"] = world
    if rank == 0:
        print(json.dumps(res))
    if "RANK" in os.environ:
        dist.destroy_process_group()
