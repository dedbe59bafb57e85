"""GPU parity of the batched Ceres-functor evaluation (K4) against the CPU oracle (which
differentiates the restated functors with Jet<6> duals exactly as ceres::AutoDiffCostFunction
does).  Tolerance: 1e-9 relative to the block scale -- far inside BASELINE's 1e-5 -- because the
GPU uses closed-form Jacobians and a different sin/cos implementation."""
import ctypes as C
import math

import numpy as np
import pytest

from nautilus_amd import _lib, residuals as R, synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu

RTOL = 1e-9


def close(a, b, scale=1.0):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape
    err = np.max(np.abs(a - b)) if a.size else 0.0
    assert err <= RTOL * max(scale, float(np.max(np.abs(b))) if b.size else 1.0), err


def oracle_eval(kind, blocks, src, tgt, poses):
    r, j0, j1 = [], [], []
    for c, s, t in zip(blocks, src, tgt):
        a, b, d = O.lidar_block(kind, c[:, 0:2], c[:, 2:4], c[:, 4:6], c[:, 6:8], poses[s], poses[t])
        r.append(a), j0.append(b), j1.append(d)
    return np.concatenate(r), np.concatenate(j0), np.concatenate(j1)


@pytest.mark.parametrize("kind", [_lib.NHIP_LIDAR_NORMAL, _lib.NHIP_LIDAR_POINT])
def test_lidar_blocks_match_autodiff_oracle(gpu, small_bag, kind):
    rng = np.random.default_rng(7)
    poses = small_bag.odom + rng.normal(0, [0.01, 0.01, math.radians(0.2)], small_bag.odom.shape)
    blocks, src, tgt = [], [], []
    for i in range(4, 40, 3):
        for j in (i - 1, i - 4):
            c = small_bag.correspondences(i, j, poses)
            if len(c):
                blocks.append(c), src.append(i), tgt.append(j)
    blocks.append(blocks[0][:1]), src.append(src[0]), tgt.append(tgt[0])      # N = 1
    blocks.append(blocks[1][:257]), src.append(src[1]), tgt.append(tgt[1])    # crosses a 256-lane tile
    assert sum(len(b) for b in blocks) > 5000
    batch = R.LidarResidualBatch(kind, blocks, src, tgt, len(poses))
    res, js, jt = batch.evaluate(poses)
    wr, wj0, wj1 = oracle_eval(kind, blocks, src, tgt, poses)
    close(res, wr)
    close(js, wj0, 30.0)
    close(jt, wj1, 30.0)
    # residual-only and single-Jacobian requests (NULL jacobian pointers, solver.cc:384-386)
    r2, a2, b2 = batch.evaluate(poses, False, False)
    assert a2 is None and b2 is None and np.array_equal(r2, res)
    r3, a3, b3 = batch.evaluate(poses, False, True)
    assert a3 is None and np.array_equal(b3, jt) and np.array_equal(r3, res)
    r4, a4, b4 = batch.evaluate(poses, True, False)
    assert b4 is None and np.array_equal(a4, js)
    # the smallest form over PCIe (nhip_resid_batch_eval_q): residuals + q + 8 constants per block come down, both Jacobians
    # are rebuilt on the host from them and the correspondences (nhip_resid_jacobians_from_q): same residuals bit for bit,
    # Jacobians equal to the device's to rounding (u is recovered as q - t) and to the autodiff oracle's like the device's
    rq, jsq, jtq = batch.evaluate_q(poses)
    assert np.array_equal(rq, res)
    assert np.allclose(jsq, js, rtol=1e-12, atol=1e-12) and np.allclose(jtq, jt, rtol=1e-12, atol=1e-12)
    close(jsq, wj0, 30.0)
    close(jtq, wj1, 30.0)
    r5, q5, c5 = batch.evaluate_q(poses, rebuild=False)
    assert q5.shape == (batch.n_corr, 2) and c5.shape == (batch.n_blocks, 8) and np.array_equal(r5, res)
    # batched oracle driver agrees with the per-block one (used by bench's cpu_baseline)
    br, bj0, bj1 = O.lidar_batch(kind, batch.corr, batch.block_offsets, batch.block_src, batch.block_tgt, poses)
    assert np.array_equal(br, wr) and np.array_equal(bj0, wj0) and np.array_equal(bj1, wj1)
    batch.close()


def test_lidar_angles_near_pi_and_large_translation(gpu):
    rng = np.random.default_rng(3)
    n = 300
    c = rng.normal(0, 5, (n, 8)).astype(np.float32)
    poses = np.array([[100.0, -250.0, math.pi - 1e-9], [-80.0, 40.0, -math.pi + 1e-9], [0, 0, 0],
                      [1e-3, -1e-3, 12.5]])
    blocks = [c[:100], c[100:200], c[200:]]
    src, tgt = [0, 1, 3], [1, 2, 0]
    for kind in (_lib.NHIP_LIDAR_NORMAL, _lib.NHIP_LIDAR_POINT):
        batch = R.LidarResidualBatch(kind, blocks, src, tgt, 4)
        res, js, jt = batch.evaluate(poses)
        wr, wj0, wj1 = oracle_eval(kind, blocks, src, tgt, poses)
        close(res, wr), close(js, wj0), close(jt, wj1)
        batch.close()


def test_factory_checks_mirror_reference(gpu):
    p = np.zeros((3, 2), np.float32)
    with pytest.raises(ValueError):
        R.LIDARNormalResidual.create(p, p[:2], p, p)      # CHECK_EQ sizes, slam_residuals.h:99-101
    with pytest.raises(ValueError):
        R.LIDARPointResidual.create(p[:0], p[:0], p[:0], p[:0])  # CHECK_GT(size, 0), :165
    with pytest.raises(_lib.NhipError):
        R.LidarResidualBatch(0, [np.zeros((2, 8), np.float32)], [5], [0], 3)  # pose index out of range
    prob = R.ResidualProblem(4)
    blk = R.LIDARNormalResidual.create(p + 1, p + 2, p + 3, p + 4)
    assert blk.num_residuals == 6
    prob.AddResidualBlock(blk, 1, 0)
    prob.AddResidualBlock(R.LIDARPointResidual.create(p + 1, p + 2, p + 3, p + 4), 2, 1)
    out = prob.Evaluate(np.arange(12, dtype=np.float64).reshape(4, 3) * 0.1)
    assert set(out) == {0, 1} and out[0][0].shape == (6,) and out[1][1].shape == (6, 3)


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def test_point_to_line_matches_autodiff_oracle(gpu):
    """HITL PointToLineResidual, every branch of DistanceToLineSegment (slam_util.h:92-110)."""
    import torch
    rng = np.random.default_rng(11)
    segs = np.array([[0, 0, 2, 2], [-1, 3, 4, 3], [2, -2, 2, 5], [0.5, 0.5, -3, 1]], dtype=np.float32)
    poses = np.array([[0.2, -0.1, 0.3], [1.0, 2.0, -2.9], [-3, 0.5, 1.57]])
    lines = np.array([[0, 0, 0], [0.1, -0.2, 0.05], [-0.5, 0.3, -0.4]])  # chosen_line_pose, data_structures.h:46
    pts, pblock, bpose, bline = [], [], [], []
    for b in range(8):
        n = int(rng.integers(10, 90))
        pts.append(rng.uniform(-6, 6, (n, 2)).astype(np.float32))
        pblock += [b] * n
        bpose.append(b % 3), bline.append((b * 2) % 3)
    bseg = segs[np.arange(8) % 4]
    P = np.concatenate(pts)
    n = len(P)
    d_res = torch.empty(n, dtype=torch.float64, device="cuda:0")
    d_j0 = torch.empty(3 * n, dtype=torch.float64, device="cuda:0")
    d_j1 = torch.empty(3 * n, dtype=torch.float64, device="cuda:0")
    args = [_dev(bseg), _dev(P), _dev(np.asarray(pblock, np.int32)), _dev(np.asarray(bpose, np.int32)),
            _dev(np.asarray(bline, np.int32)), _dev(poses), _dev(lines)]
    _lib.check(_lib.load().nhip_resid_point_to_line_dev(
        args[0].data_ptr(), args[1].data_ptr(), args[2].data_ptr(), n, args[3].data_ptr(), args[4].data_ptr(), 8,
        args[5].data_ptr(), 3, args[6].data_ptr(), 3, d_res.data_ptr(), d_j0.data_ptr(), d_j1.data_ptr(), None))
    torch.cuda.synchronize()
    res, j0, j1 = d_res.cpu().numpy(), d_j0.cpu().numpy().reshape(n, 3), d_j1.cpu().numpy().reshape(n, 3)
    o = 0
    seen_inside = seen_end = 0
    for b in range(8):
        wr, w0, w1 = O.point_to_line_block(bseg[b], pts[b], poses[bpose[b]], lines[bline[b]])
        k = len(pts[b])
        close(res[o:o + k], wr), close(j0[o:o + k], w0, 10.0), close(j1[o:o + k], w1, 10.0)
        o += k
    assert np.all(res >= 0)


def test_reference_kats_through_the_hip_path(gpu):
    """The reference's own six known-answer tests (test/solver_test.cc:12-64: DistanceToLineSegment on
    the segment (0,0)-(2,2)) evaluated by the HIP PointToLineResidual kernel with identity poses."""
    import torch
    pts = np.array([[1, 1], [0, 2], [2, 0], [4, 4], [-2, -2], [2, 2]], dtype=np.float32)
    want = np.array([0.0, 2.0 * math.sin(math.pi / 4), 2.0 * math.sin(math.pi / 4), math.sqrt(8), math.sqrt(8), 0.0])
    seg = np.array([[0, 0, 2, 2]], dtype=np.float32)
    zero = np.zeros((1, 3))
    d = [_dev(seg), _dev(pts), _dev(np.zeros(6, np.int32)), _dev(np.zeros(1, np.int32)), _dev(np.zeros(1, np.int32)),
         _dev(zero), _dev(zero)]
    d_res = torch.empty(6, dtype=torch.float64, device="cuda:0")
    _lib.check(_lib.load().nhip_resid_point_to_line_dev(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), 6,
                                                        d[3].data_ptr(), d[4].data_ptr(), 1, d[5].data_ptr(), 1,
                                                        d[6].data_ptr(), 1, d_res.data_ptr(), None, None, None))
    torch.cuda.synchronize()
    got = d_res.cpu().numpy()
    assert got[0] == 0.0 and got[5] == 0.0                       # EXPECT_EQ(dist, 0) / EXPECT_FLOAT_EQ(dist, 0)
    assert np.all(np.abs(got - want) <= 4 * np.spacing(np.float32(np.maximum(want, 1e-30))))  # EXPECT_FLOAT_EQ


def test_odometry_matches_autodiff_oracle(gpu):
    import torch
    rng = np.random.default_rng(5)
    n = 200
    poses = rng.normal(0, 3, (n + 1, 3))
    poses[:, 2] = rng.uniform(-7, 7, n + 1)
    t_odom = rng.normal(0, 0.3, (n, 2)).astype(np.float32)
    r_odom = rng.uniform(-3.2, 3.2, n).astype(np.float32)
    pi, pj = np.arange(n, dtype=np.int32), np.arange(1, n + 1, dtype=np.int32)
    d = [_dev(t_odom), _dev(r_odom), _dev(pi), _dev(pj), _dev(poses)]
    d_res = torch.empty(3 * n, dtype=torch.float64, device="cuda:0")
    d_ji = torch.empty(9 * n, dtype=torch.float64, device="cuda:0")
    d_jj = torch.empty(9 * n, dtype=torch.float64, device="cuda:0")
    _lib.check(_lib.load().nhip_resid_odometry_dev(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(),
                                                   d[3].data_ptr(), n, 1.0, 2.5, d[4].data_ptr(), n + 1,
                                                   d_res.data_ptr(), d_ji.data_ptr(), d_jj.data_ptr(), None))
    torch.cuda.synchronize()
    res = d_res.cpu().numpy().reshape(n, 3)
    ji, jj = d_ji.cpu().numpy().reshape(n, 3, 3), d_jj.cpu().numpy().reshape(n, 3, 3)
    for f in range(n):
        wr, w0, w1 = O.odometry_block(t_odom[f], r_odom[f], 1.0, 2.5, poses[f], poses[f + 1])
        close(res[f], wr), close(ji[f], w0), close(jj[f], w1)


def test_host_pointer_forms_of_odometry_and_point_to_line(gpu):
    """nhip_resid_odometry / nhip_resid_point_to_line (what the C++ OdometryResidual::create /
    PointToLineResidual::create drop-ins call): host buffers in, host buffers out, against the oracle;
    out-of-range pose indices are an argument error, not a device fault."""
    L = _lib.load()
    rng = np.random.default_rng(23)
    n = 64
    poses = rng.normal(0, 2, (n + 1, 3))
    t_odom = rng.normal(0, 0.3, (n, 2)).astype(np.float32)
    r_odom = rng.uniform(-3.2, 3.2, n).astype(np.float32)
    pi, pj = np.arange(n, dtype=np.int32), np.arange(1, n + 1, dtype=np.int32)
    res, ji, jj = np.zeros((n, 3)), np.zeros((n, 3, 3)), np.zeros((n, 3, 3))
    _lib.check(L.nhip_resid_odometry(_lib.ptr(t_odom), _lib.ptr(r_odom), _lib.ptr(pi), _lib.ptr(pj), n, 0.75, 3.0,
                                     _lib.ptr(poses), n + 1, _lib.ptr(res), _lib.ptr(ji), _lib.ptr(jj)))
    for f in range(n):
        wr, w0, w1 = O.odometry_block(t_odom[f], r_odom[f], 0.75, 3.0, poses[f], poses[f + 1])
        close(res[f], wr), close(ji[f], w0), close(jj[f], w1)
    res_only = np.zeros((n, 3))
    _lib.check(L.nhip_resid_odometry(_lib.ptr(t_odom), _lib.ptr(r_odom), _lib.ptr(pi), _lib.ptr(pj), n, 0.75, 3.0,
                                     _lib.ptr(poses), n + 1, _lib.ptr(res_only), None, None))
    assert np.array_equal(res_only, res)
    bad = pj.copy()
    bad[5] = n + 1
    assert L.nhip_resid_odometry(_lib.ptr(t_odom), _lib.ptr(r_odom), _lib.ptr(pi), _lib.ptr(bad), n, 1.0, 1.0,
                                 _lib.ptr(poses), n + 1, _lib.ptr(res), None, None) == _lib.NHIP_ERR_ARG

    segs = np.array([[0, 0, 2, 2], [-1, 3, 4, 3], [2, -2, 2, 5]], dtype=np.float32)
    lines = np.array([[0, 0, 0], [0.1, -0.2, 0.05]])
    pts = [rng.uniform(-6, 6, (k, 2)).astype(np.float32) for k in (17, 1, 130)]
    pblock = np.concatenate([np.full(len(p), b, np.int32) for b, p in enumerate(pts)])
    bpose, bline = np.array([3, 0, 7], np.int32), np.array([1, 0, 1], np.int32)
    P = np.concatenate(pts)
    m = len(P)
    r, j0, j1 = np.zeros(m), np.zeros((m, 3)), np.zeros((m, 3))
    _lib.check(L.nhip_resid_point_to_line(_lib.ptr(segs), _lib.ptr(P), _lib.ptr(pblock), m, _lib.ptr(bpose),
                                          _lib.ptr(bline), 3, _lib.ptr(poses), n + 1, _lib.ptr(lines), 2,
                                          _lib.ptr(r), _lib.ptr(j0), _lib.ptr(j1)))
    o = 0
    for b in range(3):
        wr, w0, w1 = O.point_to_line_block(segs[b], pts[b], poses[bpose[b]], lines[bline[b]])
        k = len(pts[b])
        close(r[o:o + k], wr), close(j0[o:o + k], w0, 10.0), close(j1[o:o + k], w1, 10.0)
        o += k
    bline[2] = 2
    assert L.nhip_resid_point_to_line(_lib.ptr(segs), _lib.ptr(P), _lib.ptr(pblock), m, _lib.ptr(bpose),
                                      _lib.ptr(bline), 3, _lib.ptr(poses), n + 1, _lib.ptr(lines), 2,
                                      _lib.ptr(r), None, None) == _lib.NHIP_ERR_ARG


def test_full_size_block_properties(gpu):
    """Config #3 scale (1081-point blocks) through size-independent properties:
    LIDARPoint residuals are linear in the target point; Jacobian columns match central differences."""
    bag = synth.SynthBag(12, dense=True)
    poses = bag.odom.copy()
    blocks, src, tgt = bag.window_blocks(window=3, poses=poses)
    assert max(len(b) for b in blocks) > 900
    batch = R.LidarResidualBatch(_lib.NHIP_LIDAR_NORMAL, blocks, src, tgt, len(poses))
    res, js, jt = batch.evaluate(poses)
    eps = 1e-6
    for col in range(3):
        dp = np.zeros_like(poses)
        dp[:, col] = eps
        rp, _, _ = batch.evaluate(poses + dp, False, False)
        rm, _, _ = batch.evaluate(poses - dp, False, False)
        fd = (rp - rm) / (2 * eps)       # every pose moved: d r / d src[col] + d r / d tgt[col]
        assert np.max(np.abs(fd - (js[:, col] + jt[:, col]))) < 1e-5
    batch.close()


def test_feature_mode_blocks_match_oracle(gpu):
    """The production block shape: Solver::SolveSLAM only ever builds OptimizationType::FEATURE blocks (solver.cc:363):
    <= 20 planar rows (LIDARNormalResidual) and <= 10 edge rows (LIDARPointResidual) per (i, j) of a window-10 graph,
    9,945 blocks of each at 1000 poses (slam_types.h:66-67).  Through the host-buffer API (compact target Jacobian) and
    through the per-block normal equations, against the Jet<6> oracle on every block."""
    import ctypes as C
    import os
    import sys
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from nautilus_amd import _lib
    lib = _lib.load()
    blocks, poses = bench.feature_blocks()
    dev = torch.device("cuda:0")
    for kind, (corr, off, bs, bt) in blocks.items():
        assert len(bs) == 9945 and np.diff(off).max() <= (20, 10)[kind] and np.diff(off).min() >= 1
        n = len(corr)
        h = C.c_void_p()
        _lib.check(lib.nhip_resid_batch_create(kind, _lib.ptr(corr), _lib.ptr(off), _lib.ptr(bs), _lib.ptr(bt), len(bs),
                                               len(poses), C.byref(h)))
        r, js, jtt = np.empty(2 * n), np.empty((2 * n, 3)), np.empty(2 * n)
        _lib.check(lib.nhip_resid_batch_eval_compact(h, _lib.ptr(poses), _lib.ptr(r), _lib.ptr(js), _lib.ptr(jtt)))
        lib.nhip_resid_batch_free(h)
        wr, w0, w1 = O.lidar_batch(kind, corr, off, bs, bt, poses)
        assert np.allclose(r, wr, rtol=1e-9, atol=1e-12) and np.allclose(js, w0, rtol=1e-9, atol=1e-9)
        assert np.allclose(jtt, w1[:, 2], rtol=1e-9, atol=1e-9) and np.allclose(-js[:, :2], w1[:, :2], rtol=1e-9, atol=1e-9)
        # normal equations of every block: J^T J (upper triangle), J^T r, r^T r
        t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
        d_corr, d_off, d_bs, d_bt, d_poses = t(corr), t(off), t(bs), t(bt), t(poses)
        consts = torch.empty(8 * len(bs), dtype=torch.float64, device=dev)
        outb = torch.empty(28 * len(bs), dtype=torch.float64, device=dev)
        _lib.check(lib.nhip_resid_lidar_normal_eq_dev(kind, d_corr.data_ptr(), d_off.data_ptr(), d_bs.data_ptr(), d_bt.data_ptr(),
                                                      len(bs), d_poses.data_ptr(), len(poses), consts.data_ptr(), outb.data_ptr(), None))
        torch.cuda.synchronize()
        got = outb.cpu().numpy().reshape(-1, 28)
        J = np.concatenate([w0, w1], axis=1)  # (2n, 6)
        iu = np.triu_indices(6)
        for b in np.r_[0:5, np.random.default_rng(0).choice(len(bs), 60, replace=False)]:
            rows = slice(2 * off[b], 2 * off[b + 1])
            Jb, rb = J[rows], wr[rows]
            assert np.allclose(got[b, :21], (Jb.T @ Jb)[iu], rtol=1e-9, atol=1e-9)
            assert np.allclose(got[b, 21:27], Jb.T @ rb, rtol=1e-9, atol=1e-9) and np.isclose(got[b, 27], rb @ rb, rtol=1e-9)


def test_indices_in_device_memory_are_checked_in_every_family(gpu):
    """include/nautilus_hip.h, "Ids in device memory": every index a "_dev" entry point reads from a device array is
    checked by its kernel against the count passed beside the array -- an index outside it costs an error from
    nhip_dev_status(), the entry's outputs are zero (or NaN / no match for the gates), every other entry is untouched, and
    nothing is dereferenced out of bounds.  Residual families (block ids, pose indices), correspondence search (scan ids),
    candidate gates (node indices).  The reference CHECKs such input (slam_residuals.h:99-101,109)."""
    import ctypes as C
    import torch
    lib = _lib.load()
    dev = torch.device("cuda:0")
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    info = (C.c_int32 * 4)()
    assert lib.nhip_dev_status(sp, info) == _lib.NHIP_OK
    rng = np.random.default_rng(3)
    # ---- LIDAR residuals: a correspondence with a block id outside the batch, a block with a pose outside the table
    n_blocks, n_poses, per = 6, 5, 40
    corr = rng.normal(0, 1, (n_blocks * per, 8)).astype(np.float32)
    cblock = np.repeat(np.arange(n_blocks, dtype=np.int32), per)
    bs, bt = np.array([0, 1, 2, 3, 4, 1], np.int32), np.array([1, 2, 3, 4, 0, 3], np.int32)
    poses = rng.normal(0, 1, (n_poses, 3))
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    n = len(corr)

    def lidar(cb, bs_, bt_):
        d_res = torch.full((2 * n,), 7.0, dtype=torch.float64, device=dev)
        d_js = torch.full((6 * n,), 7.0, dtype=torch.float64, device=dev)
        d_jt = torch.full((6 * n,), 7.0, dtype=torch.float64, device=dev)
        d_c = torch.empty(8 * n_blocks, dtype=torch.float64, device=dev)
        a = [d(corr), d(cb), d(bs_), d(bt_), d(poses)]
        _lib.check(lib.nhip_resid_lidar_dev(_lib.NHIP_LIDAR_NORMAL, a[0].data_ptr(), a[1].data_ptr(), n, a[2].data_ptr(), a[3].data_ptr(),
                                            n_blocks, a[4].data_ptr(), n_poses, d_c.data_ptr(), d_res.data_ptr(), d_js.data_ptr(),
                                            d_jt.data_ptr(), sp))
        rc = lib.nhip_dev_status(sp, info)
        return rc, d_res.cpu().numpy().reshape(n, 2), d_js.cpu().numpy().reshape(n, 6)
    rc, good_r, good_j = lidar(cblock, bs, bt)
    assert rc == _lib.NHIP_OK
    cb_bad = cblock.copy()
    cb_bad[17] = 99
    rc, r, j = lidar(cb_bad, bs, bt)
    assert rc == _lib.NHIP_ERR_ARG and list(info)[1:] == [8, 99, 17] and "block id" in lib.nhip_last_error().decode()
    keep = np.ones(n, bool)
    keep[17] = False
    assert np.array_equal(r[keep], good_r[keep]) and np.array_equal(j[keep], good_j[keep]) and not r[17].any() and not j[17].any()
    bs_bad = bs.copy()
    bs_bad[2] = -4
    rc, r, j = lidar(cblock, bs_bad, bt)
    assert rc == _lib.NHIP_ERR_ARG and list(info)[1:] == [16, -4, 2]
    other = cblock != 2
    assert np.array_equal(r[other], good_r[other]) and np.isfinite(r).all()
    # ---- odometry factors and point-to-line blocks
    pi, pj = np.array([0, 1, 2, 9], np.int32), np.array([1, 2, 3, 0], np.int32)
    a = [d(rng.normal(0, 1, (4, 2)).astype(np.float32)), d(rng.normal(0, 1, 4).astype(np.float32)), d(pi), d(pj), d(poses)]
    d_res = torch.full((12,), 7.0, dtype=torch.float64, device=dev)
    _lib.check(lib.nhip_resid_odometry_dev(a[0].data_ptr(), a[1].data_ptr(), a[2].data_ptr(), a[3].data_ptr(), 4, 1.0, 1.0,
                                           a[4].data_ptr(), n_poses, d_res.data_ptr(), None, None, sp))
    assert lib.nhip_dev_status(sp, info) == _lib.NHIP_ERR_ARG and list(info)[1:] == [16, 9, 3]
    o = d_res.cpu().numpy().reshape(4, 3)
    assert not o[3].any() and o[:3].any(axis=1).all()
    seg = np.array([[0, 0, 2, 2]], np.float32)
    pts = rng.normal(0, 1, (10, 2)).astype(np.float32)
    pb = np.zeros(10, np.int32)
    pb[4] = 3
    a = [d(seg), d(pts), d(pb), d(np.zeros(1, np.int32)), d(np.zeros(1, np.int32)), d(poses), d(np.zeros((1, 3)))]
    d_res = torch.full((10,), 7.0, dtype=torch.float64, device=dev)
    _lib.check(lib.nhip_resid_point_to_line_dev(a[0].data_ptr(), a[1].data_ptr(), a[2].data_ptr(), 10, a[3].data_ptr(), a[4].data_ptr(), 1,
                                                a[5].data_ptr(), n_poses, a[6].data_ptr(), 1, d_res.data_ptr(), None, None, sp))
    assert lib.nhip_dev_status(sp, info) == _lib.NHIP_ERR_ARG and list(info)[1:] == [8, 3, 4]
    o = d_res.cpu().numpy()
    assert o[4] == 0.0 and (np.delete(o, 4) != 7.0).all()
    # ---- correspondence search: a block whose source scan does not exist finds nothing
    from nautilus_amd import synth
    from nautilus_amd.correspondence import IcpBatch
    bag = synth.SynthBag(6)
    xy = np.concatenate(bag.scans).astype(np.float32)
    off = np.concatenate([[0], np.cumsum([len(s) for s in bag.scans])]).astype(np.int32)
    nrm = np.concatenate(bag.normals).astype(np.float32)
    b = IcpBatch(xy, nrm, off, [1, 2, 3], [0, 1, 2])
    b.set_poses(bag.odom)
    n_good = b.search()
    counts_good = b.d_counts.cpu().numpy().copy()
    assert lib.nhip_dev_status(sp, info) == _lib.NHIP_OK and n_good > 0
    b.d_bsrc[1] = 77
    b.search()
    assert lib.nhip_dev_status(sp, info) == _lib.NHIP_ERR_ARG and list(info)[1:] == [32, 77, 1]
    counts = b.d_counts.cpu().numpy()
    assert counts[1] == 0 and counts[0] == counts_good[0] and counts[2] == counts_good[2]
    # ---- candidate gates
    cand = np.array([0, 2, 11, 4], np.int32)
    a = [d(poses), d(cand)]
    d_f = torch.full((16,), 7, dtype=torch.uint8, device=dev)
    _lib.check(lib.nhip_lc_pair_gate_dev(a[0].data_ptr(), n_poses, a[1].data_ptr(), 4, 100.0, 0, d_f.data_ptr(), sp))
    assert lib.nhip_dev_status(sp, info) == _lib.NHIP_ERR_ARG and info[1] == 16 and info[2] == 11
    f = d_f.cpu().numpy().reshape(4, 4)
    assert not f[2].any() and not f[:, 2].any() and f[0, 1] == 1
    a = [d(poses), d(np.array([0, 1], np.int32)), d(np.array([1, 8], np.int32)), d(np.tile(np.eye(2, dtype=np.float32).reshape(1, 4), (2, 1)))]
    d_s, d_f = torch.empty(2, dtype=torch.float64, device=dev), torch.empty(2, dtype=torch.uint8, device=dev)
    _lib.check(lib.nhip_lc_chi_square_gate_dev(a[0].data_ptr(), n_poses, a[1].data_ptr(), a[2].data_ptr(), a[3].data_ptr(), 2, 1e9,
                                               d_s.data_ptr(), d_f.data_ptr(), sp))
    assert lib.nhip_dev_status(sp, info) == _lib.NHIP_ERR_ARG and list(info)[1:] == [16, 8, 1]
    s_ = d_s.cpu().numpy()
    assert np.isfinite(s_[0]) and np.isnan(s_[1]) and list(d_f.cpu().numpy()) == [1, 0]
    assert lib.nhip_dev_status(sp, info) == _lib.NHIP_OK
