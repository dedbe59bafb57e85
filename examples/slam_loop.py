#!/usr/bin/env python3
"""End-to-end loop on one MI355X, or one rank per GPU under `python -m torch.distributed.run
--nproc-per-node N examples/slam_loop.py` (the shape of BASELINE configs[0] / configs[4]):

  synthetic 1081-beam bag
  -> growing-window ICP solve, windows 1..10     Solver::OptimizeOverGrowingWindow   solver.cc:335-356
  -> loop-closure candidates                     LCCandidateFilter (scatter score, GPU) + geometric pair gate
                                                 (lc_candidate_filter.cc:35-81, in place of lc_matcher.cc:28-74)
  -> batched correlative scan matching           GetRelativeTransform                solver.cc:630-649
  -> loop-closure constraints + re-solve         AddLCConstraints (TODO body)         solver.cc:651-673
  -> a HITL message (two segments on one wall)   HitlCallback: GetRelevantPosesForHITL, AddHITLResiduals, SolveSLAM
                                                                                     solver.cc:479-559
and reports trajectory error and wall-clock per stage.  Every evaluation goes through the backend handed to run():
the product's HipBackend by default; tests and bench.py's cpu_baseline inject the oracle's CPU backend to time the
same loop on the CPU restatement."""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def synthetic_hitl_message(bag, poses, early, late):
    """What a user would draw: ONE piece of the outer wall, once where the scan of node `early` puts it under the
    current estimate and once where the scan of node `late` puts it (the two differ by the accumulated drift).
    Both segments cover the same physical piece: PointToLineResidual measures the distance to the SEGMENT
    (slam_util.h:92-110), so points beyond an end of line a would be pulled along the wall.
    HitlSlamInputMsg fields (msg/HitlSlamInputMsg.msg:1-4)."""
    seg = bag.segs[0]  # bottom wall of the outer rectangle, world frame (x0 y0 x1 y1), along x
    lo = max(min(seg[0], seg[2]), bag.truth[early, 0] - 8.0, bag.truth[late, 0] - 8.0)
    hi = min(max(seg[0], seg[2]), bag.truth[early, 0] + 8.0, bag.truth[late, 0] + 8.0)

    def seen_from(i):
        # true wall piece -> node i's frame (truth) -> world under the estimate of node i
        out = []
        for x, y in ((lo, seg[1]), (hi, seg[3])):
            c, s = math.cos(-bag.truth[i, 2]), math.sin(-bag.truth[i, 2])
            lx, ly = c * (x - bag.truth[i, 0]) - s * (y - bag.truth[i, 1]), s * (x - bag.truth[i, 0]) + c * (y - bag.truth[i, 1])
            c, s = math.cos(poses[i, 2]), math.sin(poses[i, 2])
            out.append((c * lx - s * ly + poses[i, 0], s * lx + c * ly + poses[i, 1], 0.0))
        return out
    a, b = seen_from(early), seen_from(late)
    return {"line_a_start": a[0], "line_a_end": a[1], "line_b_start": b[0], "line_b_end": b[1]}


def run(n_scans=320, window=10, seed=20201114, drift_t=0.02, drift_th_deg=0.3, verbose=False, residual="normal",
        rank=0, world=1, device="cuda:0", backend=None, iterations=4, hitl=True, cell_bits=16, gate="scatter",
        min_scatter_score=0.70, csm_score_threshold=-5.0, spacing=0.25):
    """min_scatter_score: LCCandidateFilter's threshold is 0.70 (lc_candidate_filter.cc:76); scans of the synthetic
    24 m x 16 m room score ~0.4, so callers on that world pass a lower one.
    With world > 1 (one process per GPU under torch.distributed): the window ICP solve is replicated -- its
    consumer, the solver, is host-side -- and the loop-closure pairs are sharded by target across the ranks,
    matched, and all-gathered (nautilus_amd/sharding.py); every rank ends with the same trajectory."""
    from nautilus_amd import _lib, csm, hostside, posegraph, sharding, synth
    if backend is None:
        backend = posegraph.HipBackend(device)
    bag = synth.SynthBag(n_scans, dense=True, seed=seed, spacing=spacing)
    odom = synth.odometry_from_truth(bag.truth, sigma_t=drift_t, sigma_th_deg=drift_th_deg, seed=seed)
    odom = odom - odom[0] + bag.truth[0]  # both tracks start at the same anchor (pose 0 is held constant)
    xy, off = csm.pack_scans(bag.scans)
    nrm = np.concatenate(bag.normals).astype(np.float32)
    kind = _lib.NHIP_LIDAR_NORMAL if residual == "normal" else _lib.NHIP_LIDAR_POINT
    out = {"backend": backend.name, "n_scans": n_scans, "window": window, "residual": residual,
           "err_odometry_m": posegraph.trajectory_error(odom, bag.truth)}
    posegraph.clock_reset()

    t0 = time.perf_counter()
    pg, poses = posegraph.solve_growing_window(xy, nrm, off, odom, 1, window, iterations=iterations, kind=kind,
                                               device=device, verbose=verbose, backend=backend)
    out["t_icp_solve_s"] = time.perf_counter() - t0
    out["err_icp_m"] = posegraph.trajectory_error(poses, bag.truth)
    out["icp_correspondences"] = pg.icp.n_corr

    # ---- loop-closure candidates
    t0 = time.perf_counter()
    if gate == "scatter":
        # LCCandidateFilter::GetLCCandidates: scatter-matrix score of every scan (one GPU pass), nodes >= 5 m apart ...
        with posegraph.clocked("path"):
            scores = hostside.scatter_scores(backend, xy, off)
        cand = hostside.lc_candidates_from_scores(poses, scores, min_score=min_scatter_score)
        # ... then the pair gate: |dt| < lc_base_max_range (3.5 m, default_config.lua:122) on the current estimate and
        # more than 20 nodes apart (geometric stand-in for LCMatcher's per-pair ceres::Covariance, lc_matcher.cc:28-74)
        with posegraph.clocked("path"):
            src, tgt = hostside.geometric_pair_gate(poses, cand, max_range=3.5, min_separation=20, backend=backend)
        out["lc_candidate_scans"] = len(cand)
    else:
        idx = np.arange(n_scans)
        d = np.linalg.norm(poses[:, None, :2] - poses[None, :, :2], axis=2)
        s_, t_ = np.nonzero((d < 3.5) & (np.abs(idx[:, None] - idx[None, :]) > 20))
        keep = s_ > t_
        src, tgt = s_[keep].astype(np.int32), t_[keep].astype(np.int32)
    out["t_gate_s"] = time.perf_counter() - t0
    out["lc_candidates"] = int(len(src))
    lc = None
    if len(src):
        t0 = time.perf_counter()
        a = poses[src, 2] - poses[tgt, 2]
        theta0 = a - 2 * math.pi * np.rint(a / (2 * math.pi))
        if world > 1:
            spec_search = {}

            def match_shard(src_s, slot_s, th_s, ids_s):
                if len(src_s) == 0:
                    return np.zeros(0, dtype=csm.MATCH_DTYPE)
                m_, spec_search["spec"], spec_search["search"] = backend.match(xy, off, src_s, ids_s[slot_s], th_s, cell_bits)
                return m_
            with posegraph.clocked("path"):
                m = sharding.distributed_match(match_shard, src, tgt, theta0, rank, world, device=device)
            spec, search = csm.grid_spec(30.0, 0.05, 2.0, 1e-10, 40, cell_bits), csm.search_spec()
        else:
            with posegraph.clocked("path"):
                m, spec, search = backend.match(xy, off, src, tgt, theta0, cell_bits)
        out["t_csm_s"] = time.perf_counter() - t0
        with posegraph.clocked("marshal"):  # (records -> (tx, ty, theta) as consumed at solver.cc:640-644, vectorised)
            rel = np.stack([(m["ix"].astype(np.float64) - (search.nx - 1) // 2) * spec.res,
                            (m["iy"].astype(np.float64) - (search.ny - 1) // 2) * spec.res,
                            theta0 + (m["itheta"].astype(np.float64) - (search.n_theta - 1) // 2) * search.theta_step], axis=1)
            rel = rel.astype(np.float32).astype(np.float64)  # (the reference's floats)
        inside = (np.abs(m["ix"] - 40) < 40) & (np.abs(m["iy"] - 40) < 40) & (np.abs(m["itheta"] - 30) < 30)
        # "Anything above this threshold for CSM is deemed a successful local loop closure"
        # (csm_score_threshold = -5.0, config/default_config.lua:84-85); optima on the lattice border are open-ended
        good = inside & (m["score"] > csm_score_threshold)
        out["lc_accepted"] = int(good.sum())
        truth_rel = np.array([bag.true_relative(s, t) for s, t in zip(src, tgt)])
        out["lc_rel_err_m"] = float(np.sqrt(np.mean(np.sum((rel[good, :2] - truth_rel[good, :2]) ** 2, axis=1)))) if good.any() else None
        lc = (src[good], tgt[good], rel[good])
        t0 = time.perf_counter()
        with posegraph.clocked("marshal"):
            pg.add_loop_closures(*lc)
        poses, _ = pg.solve(iterations=2 * iterations, verbose=verbose)
        out["t_lc_solve_s"] = time.perf_counter() - t0
        out["err_lc_m"] = posegraph.trajectory_error(poses, bag.truth)

    # ---- HITL: the user marks the same wall twice; HitlCallback re-solves with the point-to-line blocks
    if hitl:
        t0 = time.perf_counter()
        early, late = n_scans // 20, n_scans - 1 - n_scans // 20
        msg = synthetic_hitl_message(bag, poses, early, late)
        lines = hostside.hitl_segments(msg)
        with posegraph.clocked("hitl_select"):
            a_poses, b_poses = hostside.hitl_relevant_poses(poses, bag.scans, lines[0], lines[1])
        with posegraph.clocked("marshal"):
            con = posegraph.HitlConstraint(lines[0], lines[1], a_poses, b_poses)
        out["hitl_line_a_poses"], out["hitl_line_b_poses"] = con.n_a, con.n_b
        out["hitl_points"] = int(sum(len(p) for _, p in con.blocks))
        # state_->problem.odometry_factors = GetSolvedOdomFactors() (solver.cc:535, 406-427): the solved trajectory
        # becomes the odometry; then SolveSLAM() with the constraint (:550): the growing-window solve again, HITL
        # residuals in every pass
        pg, poses = posegraph.solve_growing_window(xy, nrm, off, poses.copy(), max(1, window - 1), window, iterations=iterations, kind=kind,
                                                   device=device, verbose=verbose, backend=backend, initial=poses,
                                                   hitl=[con] if con.blocks else [], loop_closures=lc)
        out["t_hitl_solve_s"] = time.perf_counter() - t0
        out["err_hitl_m"] = posegraph.trajectory_error(poses, bag.truth)
        out["hitl_chosen_line_pose"] = [float(v) for v in con.chosen_line_pose]
    out["t_total_s"] = sum(v for k, v in out.items() if k.startswith("t_") and k.endswith("_s"))
    # By owner: the hot path of this repo (every call into the backend: correspondence search, normal equations,
    # gating, scan matching -- "gpu_path_s" with the product's backend, "cpu_path_s" with the oracle's), the sparse
    # solves (the reference's Ceres: out of scope, identical host code under either backend), and the remaining host
    # bookkeeping (assembly of the sparse system in numpy, HITL point selection, the synthetic message).
    key = "gpu_path_s" if backend.name == "hip" else "cpu_path_s"
    out[key] = posegraph.CLOCK["path"]
    out["host_solver_s"] = posegraph.CLOCK["host_solver"]
    out["host_other_s"] = out["t_total_s"] - out[key] - out["host_solver_s"]
    # ... host_other_s by owner (round 5): marshalling at the path's boundary (block lists, the HITL blocks' input arrays,
    # records -> transforms, loop-closure factors); the sparse system's assembly in numpy (the solver stand-in's
    # bookkeeping: Ceres does it inside Solve); GetRelevantPosesForHITL's point selection (host-side HITL curation);
    # and what is left: the harness (ground-truth comparisons, the synthetic HITL message).
    out["marshal_s"] = posegraph.CLOCK["marshal"]
    out["path_setup_s"] = posegraph.CLOCK["path_setup"]  # (of the path seconds: uploads + device allocations of the per-pass batches)
    out["host_assembly_s"] = posegraph.CLOCK["assemble"]
    out["hitl_select_s"] = posegraph.CLOCK["hitl_select"]
    out["harness_s"] = out["host_other_s"] - out["marshal_s"] - out["host_assembly_s"] - out["hitl_select_s"]
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--scans", type=int, default=320)
    ap.add_argument("--window", type=int, default=10)
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
