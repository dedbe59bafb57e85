"""The C oracle of the scan matcher (oracle/csm_oracle.c) against an independent numpy restatement of
the build-defined spec, written from the spec text and vectorised differently
(convolutions, whole-volume correlation with a padded grid, numpy's first-maximum argmax) so that
a slip in either implementation shows up as a difference.  Small geometries: the numpy version is
O(volume x points) in Python loops over rotations only.

The reference's own matcher (third_party/csm) is not in the tree, so this does not pin the oracle
to the reference -- it pins the oracle to the written spec."""
import math

import numpy as np
import pytest

from oracle import oracle as O


def np_grid(points, range_m, res, sigma, floor_p, levels=255):
    """hit raster -> exact integer separable blur (taps round(16384 g / sum g), R = ceil(3 sigma))
    -> floor, ln, q = round((L - Lf) / step) in `levels` steps (255: 8-bit cells, 65535: 16-bit cells)."""
    S = int(math.floor((range_m * 2.0) / res))
    H = np.zeros((S, S), dtype=np.int64)
    for x, y in np.asarray(points, dtype=np.float32):
        if not (abs(x) < 1e9 and abs(y) < 1e9):
            continue
        c = S // 2 + math.floor(float(x) / res)      # float promoted to double, cimg_debug.h:31-37
        r = S // 2 + math.floor(float(y) / res)
        if 0 <= c < S and 0 <= r < S:
            H[r, c] = 1
    R = int(math.ceil(3.0 * sigma))
    g = np.array([math.exp(-(i * i) / (2.0 * sigma * sigma)) for i in range(-R, R + 1)])
    taps = np.floor(16384.0 * g / g.sum() + 0.5).astype(np.int64)
    K = int(taps.sum())
    V = np.apply_along_axis(lambda row: np.convolve(row, taps, mode="same"), 1, H)   # taps are symmetric
    V = np.apply_along_axis(lambda col: np.convolve(col, taps, mode="same"), 0, V)
    v = np.maximum(V.astype(np.float64) / (float(K) * float(K)), floor_p)
    Lf = math.log(floor_p)
    step = -Lf / float(levels)
    q = np.floor((np.log(v) - Lf) / step + 0.5)
    return np.clip(q, 0, levels).astype(np.uint8 if levels == 255 else np.uint16), K


def np_volume(points, grid, res, theta0, n_theta, nx, ny, theta_step, origin=(0, 0)):
    """sum[k, ix, iy] = sum_p G[row_p + iy - hy, col_p + ix - hx], out-of-grid = 0; rotation composed
    in double with individually rounded products, applied in float32 with individually rounded products."""
    S = grid.shape[0]
    hx, hy = (nx - 1) // 2, (ny - 1) // 2
    pad = max(nx, ny) + 2
    P = np.zeros((S + 2 * pad, S + 2 * pad), dtype=np.int64)
    P[pad:pad + S, pad:pad + S] = grid
    pts = np.asarray(points, dtype=np.float32).reshape(-1, 2)
    vol = np.zeros((n_theta, nx, ny), dtype=np.int64)
    c0, s0 = math.cos(theta0), math.sin(theta0)
    for k in range(n_theta):
        d = float(k - (n_theta - 1) // 2) * theta_step
        cd, sd = math.cos(d), math.sin(d)
        cf = np.float32(c0 * cd - s0 * sd)            # Python floats: each product and the difference round once
        sf = np.float32(s0 * cd + c0 * sd)
        xr = (cf * pts[:, 0]) - (sf * pts[:, 1])      # float32 arrays: individually rounded, no FMA in numpy
        yr = (sf * pts[:, 0]) + (cf * pts[:, 1])
        for x, y in zip(xr, yr):
            if not (abs(x) < 1e9 and abs(y) < 1e9):
                continue
            c = S // 2 + math.floor(float(x) / res) + origin[0] - hx
            r = S // 2 + math.floor(float(y) / res) + origin[1] - hy
            if c + nx <= 0 or c >= S or r + ny <= 0 or r >= S:
                continue
            # a window that overlaps the grid at all lies inside the padded copy (pad > lattice)
            win = P[r + pad:r + pad + ny, c + pad:c + pad + nx]    # [iy, ix]
            vol[k] += win.T
    return vol


CASES = [
    # range, res, sigma, floor, n_theta, nx, ny, theta_step_deg, theta0, origin
    (3.0, 0.05, 2.0, 1e-10, 5, 9, 9, 1.0, 0.3, (0, 0)),
    (2.0, 0.03, 1.0, 1e-10, 3, 7, 11, 2.0, -2.9, (0, 0)),
    (4.0, 0.1, 1.5, 1e-6, 7, 5, 5, 0.5, 3.1, (2, -1)),
]


@pytest.mark.parametrize("cell_bits", [16, 8])
@pytest.mark.parametrize("case", CASES)
def test_c_oracle_equals_numpy_restatement(case, cell_bits):
    range_m, res, sigma, floor_p, n_theta, nx, ny, step_deg, theta0, origin = case
    levels = 65535 if cell_bits == 16 else 255
    rng = np.random.default_rng(int(range_m * 100 + nx))
    # an L-shaped wall + clutter + a few points outside the grid and one non-finite
    t = rng.uniform(0, 1, 60)
    wall = np.concatenate([np.stack([t[:30] * range_m * 0.8 - 0.5, np.full(30, 0.7)], 1),
                           np.stack([np.full(30, -0.6), t[30:] * range_m * 0.7 - 0.4], 1)])
    tgt = np.concatenate([wall + rng.normal(0, 0.01, wall.shape), rng.uniform(-range_m, range_m, (10, 2)),
                          [[range_m * 1.5, 0.0], [0.0, -range_m * 1.01]]]).astype(np.float32)
    th_true, shift = theta0 + math.radians(step_deg), np.array([2 * res, -res])
    c, s = math.cos(-th_true), math.sin(-th_true)
    src = ((wall - shift) @ np.array([[c, -s], [s, c]]).T).astype(np.float32)
    src = np.concatenate([src, np.array([[np.nan, 0.0], [1e12, 1.0]], np.float32)])

    gs = O.grid_spec(range_m, res, sigma, floor_p, cell_bits)
    ss = O.search_spec(n_theta, nx, ny, math.radians(step_deg))
    want_grid, K = np_grid(tgt, range_m, res, sigma, floor_p, levels)
    got_grid = np.asarray(O.grid_build(tgt, gs)).reshape(want_grid.shape)
    assert got_grid.dtype == want_grid.dtype and np.array_equal(got_grid, want_grid)
    assert want_grid.max() > 0.78 * levels and (want_grid == 0).mean() > 0.5   # walls near the top level, mostly floor

    want_vol = np_volume(src, want_grid, res, theta0, n_theta, nx, ny, math.radians(step_deg), origin)
    got_vol = np.asarray(O.csm_scores(src, got_grid, gs, theta0, ss, origin)).reshape(n_theta, nx, ny)
    assert np.array_equal(got_vol, want_vol)
    assert want_vol.max() > 0

    m = O.csm_match(src, got_grid, gs, theta0, ss, origin)
    k, ix, iy = np.unravel_index(int(np.argmax(want_vol)), want_vol.shape)   # numpy: first maximum in C order
    assert (int(m.itheta), int(m.ix), int(m.iy)) == (k, ix, iy)
    assert int(m.sum) == int(want_vol.max())
    Lf = math.log(floor_p)
    assert float(m.score) == Lf + (-Lf / float(levels) * float(want_vol.max())) / float(len(src))
