"""Pins the CPU oracle (test infrastructure) before it is trusted as the checker.

 * DistanceToLineSegment: the six known-answer tests of the reference itself,
   /root/reference/test/solver_test.cc:12-64 (the only golden values the reference holds on the
   hot path), with gtest's EXPECT_FLOAT_EQ tolerance (4 ULP).
 * The four cost functors: no golden values exist in the reference, so the
   Jet<6> autodiff restatement is cross-checked against (a) central finite differences of its own
   residuals, (b) independently derived closed forms (numpy, fp64), (c) sympy where installed.
 * CSM: parity unpinned (third_party/csm absent); the oracle is checked for internal consistency
   (argmax == argmax of the brute-force score volume, tie-break order, ground-truth recovery)."""
import math

import numpy as np
import pytest

from oracle import oracle as O


def ulp_diff_f32(a, b):
    ia = np.array([a], dtype=np.float32).view(np.int32)[0]
    ib = np.array([b], dtype=np.float32).view(np.int32)[0]
    return abs(int(ia) - int(ib))


# ---------------------------------------------------------------- reference KATs
KAT = [  # (point, expected) with segment (0,0)-(2,2); test/solver_test.cc:12-64
    ((1, 1), 0.0, "trivial_on_line", True),
    ((0, 2), 2.0 * math.sin(math.pi / 4), "trivial_off_line", False),
    ((2, 0), 2.0 * math.sin(math.pi / 4), "negative_off_line", False),
    ((4, 4), math.sqrt(8), "from_endpoint", False),
    ((-2, -2), math.sqrt(8), "from_start", False),
    ((2, 2), 0.0, "line_is_endpoint", False),
]


@pytest.mark.parametrize("pt,expect,name,exact", KAT, ids=[k[2] for k in KAT])
def test_distance_to_line_segment_reference_kats(pt, expect, name, exact):
    got = O.dist_to_segment_f(pt, (0, 0), (2, 2))
    if exact:
        assert got == 0.0  # EXPECT_EQ(dist, 0)
    else:
        assert ulp_diff_f32(got, np.float32(expect)) <= 4  # EXPECT_FLOAT_EQ
    assert abs(O.dist_to_segment_d(pt, (0, 0), (2, 2)) - expect) < 1e-15 * max(1.0, expect) * 4


def test_distance_to_line_segment_branches():
    # interior projection vs endpoint distance; degenerate axis-aligned segments (IsBetween closed)
    assert O.dist_to_segment_d((1, 5), (0, 0), (2, 0)) == 5.0
    assert O.dist_to_segment_d((3, 4), (0, 0), (2, 0)) == pytest.approx(math.hypot(1, 4), abs=1e-15)
    assert O.dist_to_segment_d((-3, 4), (0, 0), (2, 0)) == 5.0
    assert O.dist_to_segment_d((0, 1), (0, 0), (0, 3)) == 0.0
    assert O.dist_to_segment_d((2, 3), (0, 0), (0, 3)) == 2.0  # projection lands on the closed end


# ---------------------------------------------------------------- functors
def fd_jac(f, p0, p1, eps=1e-6):
    r0 = f(p0, p1)
    J0, J1 = np.zeros((len(r0), 3)), np.zeros((len(r0), 3))
    for k in range(3):
        d = np.zeros(3)
        d[k] = eps
        J0[:, k] = (f(p0 + d, p1) - f(p0 - d, p1)) / (2 * eps)
        J1[:, k] = (f(p0, p1 + d) - f(p0, p1 - d)) / (2 * eps)
    return J0, J1


def rot(t):
    return np.array([[math.cos(t), -math.sin(t)], [math.sin(t), math.cos(t)]])


def closed_form_lidar(kind, sp, tp, sn, tn, ps, pt):
    """Independent fp64 derivation: q = R_t^T (R_s p + t_s - t_t)."""
    Rs, Rt = rot(ps[2]), rot(pt[2])
    Jm = np.array([[0.0, -1.0], [1.0, 0.0]])
    n = len(sp)
    r, J0, J1 = np.zeros(2 * n), np.zeros((2 * n, 3)), np.zeros((2 * n, 3))
    for i in range(n):
        p, t = sp[i].astype(np.float64), tp[i].astype(np.float64)
        q = Rt.T @ (Rs @ p + ps[:2] - pt[:2])
        u = Rt.T @ Rs @ p
        dq_s = np.column_stack([Rt.T, Jm @ u])          # d q / d (x_s, y_s, th_s)
        dq_t = np.column_stack([-Rt.T, -(Jm @ q)])       # d q / d (x_t, y_t, th_t)
        if kind == 0:
            ns, nt = sn[i].astype(np.float64), tn[i].astype(np.float64)
            r[2 * i], r[2 * i + 1] = nt @ (q - t), ns @ (t - q)
            J0[2 * i], J0[2 * i + 1] = nt @ dq_s, -(ns @ dq_s)
            J1[2 * i], J1[2 * i + 1] = nt @ dq_t, -(ns @ dq_t)
        else:
            r[2 * i:2 * i + 2] = t - q
            J0[2 * i:2 * i + 2] = -dq_s
            J1[2 * i:2 * i + 2] = -dq_t
    return r, J0, J1


@pytest.mark.parametrize("kind", [0, 1])
@pytest.mark.parametrize("n", [1, 2, 5, 33])
def test_lidar_functor_jacobians(kind, n):
    rng = np.random.default_rng(100 * kind + n)
    sp, tp = rng.normal(0, 4, (n, 2)).astype(np.float32), rng.normal(0, 4, (n, 2)).astype(np.float32)
    sn, tn = rng.normal(0, 1, (n, 2)).astype(np.float32), rng.normal(0, 1, (n, 2)).astype(np.float32)
    for ps, pt in [(np.array([0.3, -0.2, 0.4]), np.array([1.0, 2.0, -1.1])),
                   (np.array([5.0, 1.0, math.pi - 1e-6]), np.array([-3.0, 0.5, -math.pi + 1e-6]))]:
        r, j0, j1 = O.lidar_block(kind, sp, tp, sn, tn, ps, pt)
        cr, c0, c1 = closed_form_lidar(kind, sp, tp, sn, tn, ps, pt)
        assert np.allclose(r, cr, rtol=0, atol=1e-12)
        assert np.allclose(j0, c0, rtol=0, atol=1e-11) and np.allclose(j1, c1, rtol=0, atol=1e-11)
        f = lambda a, b: O.lidar_block(kind, sp, tp, sn, tn, a, b, jac=(False, False))[0]
        f0, f1 = fd_jac(f, ps, pt)
        assert np.allclose(j0, f0, atol=2e-7) and np.allclose(j1, f1, atol=2e-7)
        # NULL jacobian pointers (constant parameter block, solver.cc:384-386)
        r2, a2, b2 = O.lidar_block(kind, sp, tp, sn, tn, ps, pt, jac=(False, True))
        assert a2 is None and np.array_equal(b2, j1) and np.array_equal(r2, r)
        # residual-only path runs the functor on plain doubles: same values
        assert np.allclose(f(ps, pt), r, rtol=0, atol=1e-13)


def test_lidar_normal_quirk_source_normal_in_source_frame():
    """slam_residuals.h:83-84 dots the SOURCE-frame normal with a target-frame vector; preserved."""
    sp, tp = np.array([[1.0, 0.0]], np.float32), np.array([[0.0, 0.0]], np.float32)
    sn, tn = np.array([[1.0, 0.0]], np.float32), np.array([[0.0, 1.0]], np.float32)
    ps, pt = np.array([0.0, 0.0, math.pi / 2]), np.zeros(3)
    r, _, _ = O.lidar_block(0, sp, tp, sn, tn, ps, pt)
    # q = (0, 1): r0 = nt.(q - t) = 1; r1 = ns.(t - q) = (1,0).(0,-1) = 0  (not rotated into target)
    assert np.allclose(r, [1.0, 0.0], atol=1e-15)


def test_odometry_functor():
    rng = np.random.default_rng(9)
    for _ in range(20):
        pi, pj = rng.normal(0, 3, 3), rng.normal(0, 3, 3)
        t, ro = rng.normal(0, 1, 2).astype(np.float32), np.float32(rng.uniform(-3, 3))
        r, j0, j1 = O.odometry_block(t, ro, 1.5, 0.7, pi, pj)
        d = pi[2] + float(ro) - pj[2]
        want = np.array([1.5 * (pi[0] + float(t[0]) - pj[0]), 1.5 * (pi[1] + float(t[1]) - pj[1]),
                         0.7 * math.atan2(math.sin(d), math.cos(d))])  # world-frame error, :29,36-38
        assert np.allclose(r, want, atol=1e-14)
        assert np.allclose(j0, np.diag([1.5, 1.5, 0.7]), atol=1e-14)
        assert np.allclose(j1, -np.diag([1.5, 1.5, 0.7]), atol=1e-14)


def test_point_to_line_functor_jacobians():
    rng = np.random.default_rng(4)
    seg = np.array([0.0, 0.0, 2.0, 2.0], np.float32)
    pts = rng.uniform(-4, 6, (60, 2)).astype(np.float32)
    pose, line = np.array([0.3, -0.4, 0.5]), np.array([0.1, 0.2, -0.3])
    r, j0, j1 = O.point_to_line_block(seg, pts, pose, line)
    f = lambda a, b: O.point_to_line_block(seg, pts, a, b, jac=(False, False))[0]
    f0, f1 = fd_jac(f, pose, line, 1e-7)
    assert np.all(r >= 0)
    assert np.allclose(j0, f0, atol=5e-6) and np.allclose(j1, f1, atol=5e-6)
    # with identity poses the residual is the plain distance of the KATs
    r0, _, _ = O.point_to_line_block(seg, np.array([[0, 2], [4, 4]], np.float32), np.zeros(3), np.zeros(3))
    assert np.allclose(r0, [2 * math.sin(math.pi / 4), math.sqrt(8)], atol=1e-15)


def test_jacobians_against_sympy():
    sympy = pytest.importorskip("sympy")
    xs, ys, ths, xt, yt, tht, px, py, tx, ty, nsx, nsy, ntx, nty = sympy.symbols(
        "xs ys ths xt yt tht px py tx ty nsx nsy ntx nty", real=True)
    Rs = sympy.Matrix([[sympy.cos(ths), -sympy.sin(ths)], [sympy.sin(ths), sympy.cos(ths)]])
    Rt = sympy.Matrix([[sympy.cos(tht), -sympy.sin(tht)], [sympy.sin(tht), sympy.cos(tht)]])
    q = Rt.T * (Rs * sympy.Matrix([px, py]) + sympy.Matrix([xs - xt, ys - yt]))
    e = q - sympy.Matrix([tx, ty])
    res = sympy.Matrix([ntx * e[0] + nty * e[1], -(nsx * e[0] + nsy * e[1])])
    J = res.jacobian([xs, ys, ths, xt, yt, tht])
    vals = {xs: 0.3, ys: -0.2, ths: 0.4, xt: 1.0, yt: 2.0, tht: -1.1, px: 1.5, py: -2.25, tx: 0.5, ty: 0.75,
            nsx: 0.6, nsy: -0.8, ntx: 0.28, nty: 0.96}
    Jn = np.array(J.subs(vals).evalf(30), dtype=np.float64)
    rn = np.array(res.subs(vals).evalf(30), dtype=np.float64).ravel()
    r, j0, j1 = O.lidar_block(0, np.array([[1.5, -2.25]], np.float32), np.array([[0.5, 0.75]], np.float32),
                              np.array([[0.6, -0.8]], np.float32), np.array([[0.28, 0.96]], np.float32),
                              np.array([0.3, -0.2, 0.4]), np.array([1.0, 2.0, -1.1]))
    f32 = lambda v: float(np.float32(v))
    # the oracle casts the float32 data to double: compare at float32-rounded inputs
    vals32 = dict(vals)
    for s, v in [(nsx, 0.6), (nsy, -0.8), (ntx, 0.28), (nty, 0.96)]:
        vals32[s] = f32(v)
    Jn = np.array(J.subs(vals32).evalf(30), dtype=np.float64)
    rn = np.array(res.subs(vals32).evalf(30), dtype=np.float64).ravel()
    assert np.allclose(r, rn, atol=1e-14)
    assert np.allclose(np.hstack([j0, j1]), Jn, atol=1e-13)


# ---------------------------------------------------------------- CSM oracle consistency
def test_csm_argmax_is_first_maximum_of_volume(small_bag):
    gs = O.grid_spec(30.0, 0.05, 2.0, 1e-10, 8)
    ss = O.search_spec(5, 9, 11, math.radians(2))
    g = O.grid_build(small_bag.scans[10], gs)
    vol = O.csm_scores(small_bag.scans[12], g, gs, 0.05, ss)
    m = O.csm_match(small_bag.scans[12], g, gs, 0.05, ss)
    lin = int(np.argmax(vol.ravel()))  # numpy argmax = first maximum in (k, ix, iy) order
    assert (m.itheta, m.ix, m.iy) == np.unravel_index(lin, vol.shape)
    assert m.sum == vol.max()
    n = len(small_bag.scans[12])
    Lf = math.log(1e-10)
    assert m.score == Lf + ((-Lf / 255.0) * m.sum) / n


def test_csm_tie_break_smallest_linear_index():
    """A single-cell target: several shifts tie; the first in (theta, x, y) order must win."""
    gs = O.grid_spec(2.0, 0.05, 0.5, 1e-10)
    tgt = np.array([[0.01, 0.01], [0.51, 0.01]], np.float32)
    g = O.grid_build(tgt, gs)
    src = np.array([[0.01, 0.01]], np.float32)
    ss = O.search_spec(1, 21, 21, 0.1)
    vol = O.csm_scores(src, g, gs, 0.0, ss)[0]
    m = O.csm_match(src, g, gs, 0.0, ss)
    peaks = np.argwhere(vol == vol.max())
    assert len(peaks) >= 2  # the two hit cells give equal sums
    assert (m.ix, m.iy) == tuple(peaks[0])  # row-major over (ix, iy): smallest ix, then iy


def test_csm_empty_and_outside():
    gs = O.grid_spec(2.0, 0.05, 1.0, 1e-10)
    g = O.grid_build(np.zeros((0, 2), np.float32), gs)
    assert not g.any()
    ss = O.search_spec(3, 5, 5, 0.1)
    m = O.csm_match(np.zeros((0, 2), np.float32), g, gs, 0.0, ss)
    assert (m.itheta, m.ix, m.iy, m.sum) == (0, 0, 0, 0) and m.score == math.log(1e-10)
    g2 = O.grid_build(np.array([[5.0, 5.0], [0.0, 0.0]], np.float32), gs)  # first point is off-grid: dropped
    assert g2[40, 40] > 0 and (g2 > 0).sum() <= 49
    m2 = O.csm_match(np.array([[100.0, 100.0]], np.float32), g2, gs, 0.0, ss)
    assert m2.sum == 0


def test_csm_recovers_known_offset(small_bag):
    """Goldens (iii) of SURVEY 8c: synthetic pair with a known offset, answer within one cell / step."""
    gs = O.grid_spec(30.0, 0.05, 2.0, 1e-10, 8)
    base = small_bag.scans[20]
    for (dx, dy, dth) in [(0.35, -0.20, math.radians(7)), (-0.6, 0.45, math.radians(-12))]:
        c, s = math.cos(-dth), math.sin(-dth)
        p = base - np.array([dx, dy], np.float32)
        src = np.stack([c * p[:, 0] - s * p[:, 1], s * p[:, 0] + c * p[:, 1]], 1).astype(np.float32)
        ss = O.search_spec(31, 41, 41, math.radians(1))
        m = O.csm_match(src, O.grid_build(base, gs), gs, 0.0, ss)
        assert abs((m.ix - 20) * 0.05 - dx) <= 0.051 and abs((m.iy - 20) * 0.05 - dy) <= 0.051
        assert abs(math.radians(m.itheta - 15) - dth) <= math.radians(1.01)


def test_exact_pose_score_equals_the_double_tables_volume(small_bag):
    """orc_csm_pose_score_exact (the oracle's statement of NHIP_SEARCH_EXACT_SCORE: blur sums evaluated around the cells a
    pose reads) against orc_csm_match_f64's score volume on the whole table of doubles: the same doubles, bit for bit, at
    every pose probed -- lattice corners, the optimum, poses whose lookups leave the grid -- and with a shifted centre."""
    gs, ss = O.grid_spec(12.0, 0.05, 2.0, 1e-10, 16), O.search_spec(5, 21, 21, math.radians(2.0))
    src, tgt = small_bag.scans[9].copy(), small_bag.scans[7]
    src[:3] = [[11.99, 0.0], [np.nan, 1.0], [-40.0, 2.0]]  # near the grid's edge, non-finite, outside
    g = O.grid_build_f64(tgt, gs)
    best, vol = O.csm_match_f64(src, g, gs, 0.03, ss, want_scores=True)
    poses = [(0, 0, 0), (4, 20, 20), (2, 10, 10), (int(best["itheta"]), int(best["ix"]), int(best["iy"])), (1, 3, 17)]
    for k, ix, iy in poses:
        assert O.pose_score_exact(src, tgt, gs, 0.03, ss, k, ix, iy) == vol[k, ix, iy]
    # a centre shifted by (3, -2) cells: pose (k, ix, iy) there is pose (k, ix + 3, iy - 2) here
    small = O.search_spec(5, 11, 11, math.radians(2.0))
    for k, ix, iy in [(0, 0, 0), (3, 5, 5), (4, 10, 10)]:
        assert O.pose_score_exact(src, tgt, gs, 0.03, small, k, ix, iy, origin=(3, -2)) == vol[k, ix + 5 + 3, iy + 5 - 2]
    assert O.pose_score_exact(np.zeros((0, 2), np.float32), tgt, gs, 0.0, ss, 0, 0, 0) == math.log(1e-10)
