"""Committed golden vectors (tests/golden/, minted by tests/golden/make_golden.py).
CPU part: the oracle still reproduces them bit for bit.  GPU part: the HIP path reproduces them
through the C ABI (indices / sums / grids bit-exact, residuals and Jacobians within 1e-9)."""
import hashlib
import json
import math
import os

import numpy as np
import pytest

from oracle import oracle as O

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
G = json.load(open(os.path.join(HERE, "golden.json")))
A = np.load(os.path.join(HERE, "golden_arrays.npz"))
GS = O.grid_spec(**{"range_m": G["grid_spec"]["range"], "res": G["grid_spec"]["res"],
                    "sigma": G["grid_spec"]["sigma"], "floor_p": G["grid_spec"]["floor_p"],
                    "cell_bits": G["grid_spec"].get("cell_bits", 8)})  # (the frozen vectors are 8-bit grids)


def test_oracle_reproduces_dist_kats():
    for k in G["dist_kat"]:
        got = O.dist_to_segment_f(k["point"], (0, 0), (2, 2))
        assert abs(got - k["expect"]) <= 4 * np.spacing(np.float32(max(k["expect"], 1e-30)))


def test_oracle_reproduces_functor_goldens():
    for bi, b in enumerate(G["lidar_blocks"]):
        c = A["blk%d_corr" % bi]
        for kind, name in ((0, "normal"), (1, "point")):
            r, j0, j1 = O.lidar_block(kind, c[:, 0:2], c[:, 2:4], c[:, 4:6], c[:, 6:8],
                                      np.array(b["source_pose"]), np.array(b["target_pose"]))
            assert np.array_equal(r, A["blk%d_%s_r" % (bi, name)])
            assert np.array_equal(j0, A["blk%d_%s_j0" % (bi, name)])
            assert np.array_equal(j1, A["blk%d_%s_j1" % (bi, name)])
    r, j0, j1 = O.point_to_line_block(A["p2l_seg"], A["p2l_points"], np.array(G["p2l"]["pose"]),
                                      np.array(G["p2l"]["line_pose"]))
    assert np.array_equal(r, A["p2l_r"]) and np.array_equal(j0, A["p2l_j0"]) and np.array_equal(j1, A["p2l_j1"])


def test_oracle_reproduces_csm_goldens():
    grids = {}
    for p in G["csm_pairs"][:8]:  # the 61x81x81 cases cost ~0.1 s each on one core
        t = p["tgt"]
        if t not in grids:
            grids[t] = O.grid_build(A["scan%d" % t], GS)
            assert hashlib.sha256(grids[t].tobytes()).hexdigest() == G["grid_sha256"][str(t)]
        c = p["cfg"]
        m = O.csm_match(A["scan%d" % p["src"]], grids[t], GS, p["theta0"],
                        O.search_spec(c["n_theta"], c["nx"], c["ny"], math.radians(c["step_deg"])))
        assert (m.itheta, m.ix, m.iy, m.sum) == (p["itheta"], p["ix"], p["iy"], p["sum"])
        assert m.score == p["score"]


def test_csm_goldens_agree_with_ground_truth():
    """Sanity of the minted answers themselves: within one cell / one angular step of the truth."""
    for p in G["csm_pairs"]:
        c = p["cfg"]
        if c["n_theta"] != 61:
            continue
        tx, ty = (p["ix"] - 40) * 0.05, (p["iy"] - 40) * 0.05
        th = p["theta0"] + math.radians(p["itheta"] - 30)
        gx, gy, gth = p["truth"]
        assert abs(tx - gx) <= 0.11 and abs(ty - gy) <= 0.11 and abs(th - gth) <= math.radians(1.6)
    # determinism vector: the repeated pair is byte-identical
    a, b = G["csm_pairs"][0], G["csm_pairs"][6]
    assert {k: a[k] for k in a if k != "truth"} == {k: b[k] for k in b if k != "truth"}


@pytest.mark.gpu
def test_hip_path_reproduces_csm_goldens(gpu):
    from nautilus_amd import csm
    ids = sorted({p["tgt"] for p in G["csm_pairs"]})
    srcs = sorted({p["src"] for p in G["csm_pairs"]} | set(ids))
    index = {s: i for i, s in enumerate(srcs)}
    st = csm.ScanTable.from_list([A["scan%d" % s] for s in srcs])
    spec = csm.grid_spec(30.0, 0.05, 2.0, 1e-10, 40, cell_bits=8)
    grids = csm.LikelihoodGrids(st, [index[t] for t in ids], spec)
    for slot, t in enumerate(ids):
        assert hashlib.sha256(np.ascontiguousarray(grids.interior(slot)).tobytes()).hexdigest() == G["grid_sha256"][str(t)]
    for cfgkey in {json.dumps(p["cfg"], sort_keys=True) for p in G["csm_pairs"]}:
        c = json.loads(cfgkey)
        ps = [p for p in G["csm_pairs"] if p["cfg"] == c]
        search = csm.search_spec(c["n_theta"], c["nx"], c["ny"], math.radians(c["step_deg"]))
        got, sums = csm.match_pairs(st, grids, [index[p["src"]] for p in ps], [ids.index(p["tgt"]) for p in ps],
                                    [p["theta0"] for p in ps], search)
        for m, s, p in zip(got, sums, ps):
            assert (int(m["itheta"]), int(m["ix"]), int(m["iy"]), int(s)) == (p["itheta"], p["ix"], p["iy"], p["sum"])
            assert m["score"] == np.float32(p["score"])


@pytest.mark.gpu
def test_hip_path_reproduces_functor_goldens(gpu):
    from nautilus_amd import residuals as R
    for bi, b in enumerate(G["lidar_blocks"]):
        c = A["blk%d_corr" % bi]
        poses = np.array([b["source_pose"], b["target_pose"]])
        for kind, name in ((0, "normal"), (1, "point")):
            batch = R.LidarResidualBatch(kind, [c], [0], [1], 2)
            r, j0, j1 = batch.evaluate(poses)
            for got, key in ((r, "r"), (j0, "j0"), (j1, "j1")):
                want = A["blk%d_%s_%s" % (bi, name, key)]
                assert np.max(np.abs(got - want)) <= 1e-9 * max(1.0, np.max(np.abs(want)))
            batch.close()
