#!/usr/bin/env python3
"""Mints the committed golden vectors.

The reference itself cannot run here (third_party/csm is an empty submodule; the functors need
Eigen/Ceres/glog/ROS headers the image lacks), so these vectors come from the CPU oracle AFTER it
has been pinned by tests/test_oracle_kat.py (reference KATs, finite differences, sympy, ground
truth).  They freeze today's answers: any later change of oracle or HIP path that moves a bit
fails tests/test_golden.py.  Regenerate with: python tests/golden/make_golden.py
"""
import hashlib
import json
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from nautilus_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    out = {}
    arrays = {}
    # (i) the reference's own KATs, test/solver_test.cc:12-64 (segment (0,0)-(2,2), float)
    out["dist_kat"] = [{"point": p, "expect": e} for p, e in
                       [((1, 1), 0.0), ((0, 2), 2.0 * math.sin(math.pi / 4)), ((2, 0), 2.0 * math.sin(math.pi / 4)),
                        ((4, 4), math.sqrt(8)), ((-2, -2), math.sqrt(8)), ((2, 2), 0.0)]]
    # (ii) functor blocks, N = 1, 2, 5, incl. theta near +-pi
    rng = np.random.default_rng(20201114)
    blocks = []
    for bi, (n, ps, pt) in enumerate([(1, [0.3, -0.2, 0.4], [1.0, 2.0, -1.1]), (2, [5.0, 1.0, math.pi - 1e-6], [-3.0, 0.5, -math.pi + 1e-6]),
                                      (5, [-0.15, 0.0, 0.2], [0.0, 0.0, 0.0])]):
        c = rng.normal(0, 3, (n, 8)).astype(np.float32)
        arrays["blk%d_corr" % bi] = c
        entry = {"n": n, "source_pose": ps, "target_pose": pt}
        for kind, name in ((0, "normal"), (1, "point")):
            r, j0, j1 = O.lidar_block(kind, c[:, 0:2], c[:, 2:4], c[:, 4:6], c[:, 6:8], np.array(ps), np.array(pt))
            arrays["blk%d_%s_r" % (bi, name)] = r
            arrays["blk%d_%s_j0" % (bi, name)] = j0
            arrays["blk%d_%s_j1" % (bi, name)] = j1
        blocks.append(entry)
    out["lidar_blocks"] = blocks
    # L-corner scene of test/feature_extractor_test.cc:37-57 as a HITL PointToLine block
    xs = np.float32(0.5) - np.float32(0.02) * np.arange(25, dtype=np.float32)
    corner = np.concatenate([np.stack([xs, np.zeros(25, np.float32)], 1),
                             np.stack([np.zeros(24, np.float32), np.float32(0.02) * np.arange(1, 25, dtype=np.float32)], 1)])
    seg = np.array([0.0, 0.0, 0.5, 0.0], np.float32)
    pose, line = np.array([-0.15, 0.0, 0.2]), np.array([0.0, 0.0, 0.0])
    r, j0, j1 = O.point_to_line_block(seg, corner, pose, line)
    arrays.update(p2l_points=corner, p2l_seg=seg, p2l_r=r, p2l_j0=j0, p2l_j1=j1)
    out["p2l"] = {"pose": pose.tolist(), "line_pose": line.tolist()}
    r, j0, j1 = O.odometry_block(np.array([0.25, -0.01], np.float32), np.float32(0.05), 1.0, 1.0,
                                 np.array([1.0, 2.0, 3.1]), np.array([1.2, 2.1, -3.1]))
    arrays.update(odo_r=r, odo_j0=j0, odo_j1=j1)
    # (iii) CSM pairs with known ground truth + (iv) determinism vector (the same pair twice)
    bag = synth.SynthBag(40)
    ids = [8, 21, 33]
    scans = {i: bag.scans[i] for i in ids}
    src, tgt, th0 = bag.sample_pairs(per_target=2, targets=ids, max_dist=1.5, min_sep=2)
    keep = sorted(set(src.tolist()) | set(ids))
    for i in keep:
        arrays["scan%d" % i] = bag.scans[i]
    gs = O.grid_spec(30.0, 0.05, 2.0, 1e-10, 8)
    pairs = []
    for cfg in ({"n_theta": 61, "nx": 81, "ny": 81, "step_deg": 1.0}, {"n_theta": 9, "nx": 21, "ny": 13, "step_deg": 2.0}):
        ss = O.search_spec(cfg["n_theta"], cfg["nx"], cfg["ny"], math.radians(cfg["step_deg"]))
        for s, t, a in list(zip(src, tgt, th0)) + [(src[0], tgt[0], th0[0])]:
            g = O.grid_build(bag.scans[t], gs)
            m = O.csm_match(bag.scans[s], g, gs, a, ss)
            gx, gy, gth = bag.true_relative(s, t)
            pairs.append({"cfg": cfg, "src": int(s), "tgt": int(t), "theta0": float(a), "itheta": m.itheta, "ix": m.ix,
                          "iy": m.iy, "sum": m.sum, "score": m.score, "truth": [gx, gy, gth]})
    out["csm_pairs"] = pairs
    out["grid_sha256"] = {str(t): hashlib.sha256(O.grid_build(bag.scans[t], gs).tobytes()).hexdigest() for t in ids}
    out["grid_spec"] = {"range": 30.0, "res": 0.05, "sigma": 2.0, "floor_p": 1e-10, "cell_bits": 8}
    np.savez_compressed(os.path.join(HERE, "golden_arrays.npz"), **arrays)
    json.dump(out, open(os.path.join(HERE, "golden.json"), "w"), indent=1)
    print("wrote golden.json (%d csm pairs) and golden_arrays.npz (%d arrays)" % (len(pairs), len(arrays)))


if __name__ == "__main__":
    main()
