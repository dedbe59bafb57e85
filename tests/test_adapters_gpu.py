"""The C++ drop-ins (nautilus_amd/adapters/): CorrelativeScanMatcher used as solver.cc:633-644
uses it, and the slam_residuals replacement driven like Ceres' evaluation loop.  The binary is
built by __graft_entry__.build() (g++, links libnautilus_hip.so) and needs a GPU to run."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "nautilus_amd", "adapters", "adapter_test")


def test_adapter_headers_mirror_reference_signatures():
    h = open(os.path.join(ROOT, "nautilus_amd", "adapters", "CorrelativeScanMatcher.h")).read()
    assert "class CorrelativeScanMatcher" in h and "GetTransformation(" in h
    assert "double scanner_range, double trans_range, double low_res, double high_res" in h
    r = open(os.path.join(ROOT, "nautilus_amd", "adapters", "slam_residuals_hip.h")).read()
    for s in ("struct OdometryResidual", "struct LIDARNormalResidual", "struct LIDARPointResidual",
              "struct PointToLineResidual", "namespace nautilus {", "static nautilus_hip::BatchedCost *create(",
              "double translation_weight, double rotation_weight", "line_segment, const std::vector<"):
        assert s in r


@pytest.mark.gpu
def test_adapter_binary_on_gpu(gpu, tmp_path):
    """Runs the C++ drop-ins as solver.cc would, then compares what the C++ CorrelativeScanMatcher and
    CorrelativeScanMatcherBatch returned with the oracle's independent restatement of the same searches on the same
    point clouds: scores, translations and rotations equal as floats (the C++ host ships the implementation the
    oracle has seen)."""
    import math
    import numpy as np
    from oracle import oracle as O
    if not os.path.exists(BIN):
        subprocess.check_call(["make", "-C", os.path.dirname(BIN)])
    p = subprocess.run([BIN, str(tmp_path)], capture_output=True, text=True, timeout=300)
    print(p.stdout, p.stderr)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "ADAPTER_OK" in p.stdout
    pc = {n: np.fromfile(os.path.join(tmp_path, "pc_%s.f32" % n), dtype=np.float32).reshape(-1, 2) for n in "ab"}
    assert len(pc["a"]) > 500 and len(pc["b"]) == len(pc["a"])
    rows = [l.split() for l in open(os.path.join(tmp_path, "results.txt"))]
    hexf = lambda x: float.fromhex(x)
    seen = set()
    for r in rows:
        if r[0].startswith("get_transformation"):
            rot_a, rot_b, restrict, score, tx, ty, th = map(hexf, r[1:])
            a, b = (pc["a"], pc["b"]) if r[0] == "get_transformation" else (pc["b"], pc["a"])
            want = O.two_level_match(a, b, rot_a, rot_b, restrict, 30.0, 2.0, 0.3, 0.01, cell_bits=16)
            # (the score: the fine optimum's on the unquantised table, device / host logarithms: 2e-7 relative)
            assert abs(score - want[0]) <= 2e-7 * abs(want[0]) and np.float32(tx) == want[1][0][0] and np.float32(ty) == want[1][0][1]
            assert np.float32(th) == want[1][1], (r[0], th, want)
            seen.add(r[0])
        elif r[0] == "batch":
            i = int(r[1])
            score, tx, ty, th = map(hexf, r[2:])
            # CorrelativeScanMatcherBatch(30, 0.05, {61, 81, 81, 1 deg}): pairs (0 -> 1) and (1 -> 0), headings (ath - 0.03, 0)
            ath = 0.21
            src, tgt = (pc["a"], pc["b"]) if i == 0 else (pc["b"], pc["a"])
            d = (ath - 0.03) if i == 0 else -(ath - 0.03)
            theta0 = d - 2 * math.pi * round(d / (2 * math.pi))
            gs, ss = O.grid_spec(30.0, 0.05, 2.0, 1e-10, 16), O.search_spec(61, 81, 81, math.pi / 180.0)
            m = O.csm_match(src, O.grid_build(tgt, gs), gs, theta0, ss)
            assert score == float(np.float32(m.score))
            assert np.float32(tx) == np.float32((m.ix - 40) * 0.05) and np.float32(ty) == np.float32((m.iy - 40) * 0.05)
            assert np.float32(th) == np.float32(theta0 + (m.itheta - 30) * (math.pi / 180.0))
            seen.add("batch%d" % i)
    assert seen == {"get_transformation", "get_transformation_swapped", "batch0", "batch1"}
    # the residual half: what the C++ cost functions' Evaluate() returned for the three LIDAR blocks (batched pass,
    # compact target Jacobian rebuilt on the host) against the oracle's Jet<6> restatement of slam_residuals.h:65-89
    # (LIDARNormal) and :124-145 (LIDARPoint) on the same correspondences and parameter blocks
    for q in range(3):
        raw = open(os.path.join(tmp_path, "lidar_block_%d.bin" % q), "rb").read()
        kind, n = np.frombuffer(raw, dtype=np.int32, count=2)
        assert n == 300 and kind == (1 if q == 1 else 0)
        off = 8
        vecs = []
        for _ in range(4):
            vecs.append(np.frombuffer(raw, dtype=np.float32, count=2 * n, offset=off).reshape(n, 2))
            off += 8 * n
        pa = np.frombuffer(raw, dtype=np.float64, count=3, offset=off)
        pb = np.frombuffer(raw, dtype=np.float64, count=3, offset=off + 24)
        off += 48
        r = np.frombuffer(raw, dtype=np.float64, count=2 * n, offset=off)
        js = np.frombuffer(raw, dtype=np.float64, count=6 * n, offset=off + 16 * n).reshape(2 * n, 3)
        jt = np.frombuffer(raw, dtype=np.float64, count=6 * n, offset=off + 64 * n).reshape(2 * n, 3)
        assert len(raw) == off + 112 * n
        wr, w0, w1 = O.lidar_block(int(kind), vecs[0], vecs[1], vecs[2], vecs[3], pa, pb)
        assert np.abs(wr).max() > 0.01 and np.abs(w0).max() > 0.5
        assert np.allclose(r, wr, rtol=1e-9, atol=1e-12)      # bar of tests/test_resid_gpu.py
        assert np.allclose(js, w0, rtol=1e-9, atol=1e-9) and np.allclose(jt, w1, rtol=1e-9, atol=1e-9)
    assert "20 problem builds" in p.stdout and "two problems coexist" in p.stdout
