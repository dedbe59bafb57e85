"""Seeded synthetic 2-D lidar data for tests and bench.

Beam model: 1081 beams, angle_min = -135 deg, increment 0.25 deg, range [0.1, 30] m; point i =
Rot(angle_min + i*inc) * (r_i, 0) in float32, as
/root/reference/src/input/pointcloud_helpers.cc:28-48 (no 55-beam truncation: that is ROS
front-end behaviour, slam_type_builder.cc:56-65).  World: outer rectangle + random wall
segments on a central island; trajectory: laps of a rounded rectangle around the island.
"""
import math

import numpy as np

N_BEAMS = 1081
ANGLE_MIN = math.radians(-135.0)
ANGLE_INC = math.radians(0.25)
RANGE_MIN, RANGE_MAX = 0.1, 30.0
SEED = 20201114


def make_world(width=40.0, height=25.0, n_walls=12, margin=4.0, seed=SEED):
    """Segments (M, 4) = x0 y0 x1 y1, centred on the origin."""
    rng = np.random.default_rng(seed)
    hw, hh = width / 2, height / 2
    segs = [(-hw, -hh, hw, -hh), (hw, -hh, hw, hh), (hw, hh, -hw, hh), (-hw, hh, -hw, -hh)]
    iw, ih = hw - margin - 1.5, hh - margin - 1.5  # island half-extent (inside the loop)
    for _ in range(n_walls):
        cx, cy = rng.uniform(-iw, iw), rng.uniform(-ih, ih)
        L, a = rng.uniform(1.5, 0.6 * min(iw, ih) + 1.5), rng.uniform(0, math.pi)
        dx, dy = 0.5 * L * math.cos(a), 0.5 * L * math.sin(a)
        x0, y0, x1, y1 = cx - dx, cy - dy, cx + dx, cy + dy
        segs.append((float(np.clip(x0, -iw, iw)), float(np.clip(y0, -ih, ih)),
                     float(np.clip(x1, -iw, iw)), float(np.clip(y1, -ih, ih))))
    return np.asarray(segs, dtype=np.float64)


def loop_trajectory(n_scans, width=40.0, height=25.0, margin=4.0, spacing=0.25, heading_sigma_deg=2.0,
                    seed=SEED):
    """Ground-truth poses (n, 3): laps of a rounded rectangle, heading = tangent + N(0, sigma)."""
    rng = np.random.default_rng(seed + 10)
    hw, hh, r = width / 2 - margin, height / 2 - margin, 1.5
    # piecewise path: 4 straights + 4 quarter circles, counter-clockwise
    sx, sy = hw - r, hh - r
    pieces = [("l", (hw, -sy), (hw, sy)), ("a", (sx, sy), 0.0), ("l", (sx, hh), (-sx, hh)),
              ("a", (-sx, sy), 0.5 * math.pi), ("l", (-hw, sy), (-hw, -sy)), ("a", (-sx, -sy), math.pi),
              ("l", (-sx, -hh), (sx, -hh)), ("a", (sx, -sy), 1.5 * math.pi)]
    lens = [math.hypot(p[2][0] - p[1][0], p[2][1] - p[1][1]) if p[0] == "l" else 0.5 * math.pi * r
            for p in pieces]
    per = sum(lens)
    poses = np.zeros((n_scans, 3))
    for i in range(n_scans):
        s = (i * spacing) % per
        for p, L in zip(pieces, lens):
            if s <= L:
                break
            s -= L
        if p[0] == "l":
            t = s / L
            x, y = p[1][0] + t * (p[2][0] - p[1][0]), p[1][1] + t * (p[2][1] - p[1][1])
            th = math.atan2(p[2][1] - p[1][1], p[2][0] - p[1][0])
        else:
            a = p[2] + s / r
            x, y, th = p[1][0] + r * math.cos(a), p[1][1] + r * math.sin(a), a + 0.5 * math.pi
        poses[i] = (x, y, th)
    poses[:, 2] += rng.normal(0.0, math.radians(heading_sigma_deg), n_scans)
    poses[:, :2] += rng.normal(0.0, 0.02, (n_scans, 2))
    return poses


def odometry_from_truth(poses, sigma_t=0.01, sigma_th_deg=0.1, seed=SEED):
    """Ground truth + random-walk drift."""
    rng = np.random.default_rng(seed + 1)
    n = len(poses)
    drift = np.cumsum(np.column_stack([rng.normal(0, sigma_t, (n, 2)),
                                       rng.normal(0, math.radians(sigma_th_deg), n)]), axis=0)
    return poses + drift


def raycast(poses, segs, noise=0.01, seed=SEED, chunk=256):
    """ranges (n, 1081) float64 (inf = no return) and hit segment ids (n, 1081)."""
    rng = np.random.default_rng(seed + 2)
    n = len(poses)
    ranges = np.full((n, N_BEAMS), np.inf)
    hit = np.full((n, N_BEAMS), -1, dtype=np.int32)
    beam = ANGLE_MIN + ANGLE_INC * np.arange(N_BEAMS)
    ax, ay = segs[:, 0], segs[:, 1]
    ex, ey = segs[:, 2] - segs[:, 0], segs[:, 3] - segs[:, 1]
    for c0 in range(0, n, chunk):
        P = poses[c0:c0 + chunk]
        ang = P[:, 2:3] + beam[None, :]                      # (c, B)
        dx, dy = np.cos(ang)[..., None], np.sin(ang)[..., None]  # (c, B, 1)
        ox, oy = P[:, 0, None, None], P[:, 1, None, None]
        den = dx * ey - dy * ex                               # (c, B, M)
        with np.errstate(divide="ignore", invalid="ignore"):
            t = ((ax - ox) * ey - (ay - oy) * ex) / den       # along the ray
            u = ((ax - ox) * dy - (ay - oy) * dx) / den       # along the segment
        ok = (np.abs(den) > 1e-12) & (t > 0) & (u >= 0) & (u <= 1)
        t = np.where(ok, t, np.inf)
        k = np.argmin(t, axis=2)
        r = np.take_along_axis(t, k[..., None], axis=2)[..., 0]
        ranges[c0:c0 + chunk] = r
        hit[c0:c0 + chunk] = np.where(np.isfinite(r), k, -1)
    ranges = ranges + rng.normal(0.0, noise, ranges.shape)
    return ranges, hit


def ranges_to_cloud(r_row):
    """pointcloud_helpers.cc:28-48 in float32; returns (points (N, 2) float32, kept beam indices)."""
    r = r_row.astype(np.float32)
    keep = np.nonzero((r >= np.float32(RANGE_MIN)) & (r <= np.float32(RANGE_MAX)))[0]
    ang = np.float32(ANGLE_MIN) + np.float32(ANGLE_INC) * keep.astype(np.float32)
    pts = np.stack([np.cos(ang).astype(np.float32) * r[keep], np.sin(ang).astype(np.float32) * r[keep]], axis=1)
    return pts.astype(np.float32), keep


class SynthBag:
    """n_scans scans + truth/odometry poses + per-point wall normals (scan frame)."""

    def __init__(self, n_scans, dense=False, seed=SEED, n_walls=12, spacing=0.25, poses_only=False):
        """poses_only: trajectory, odometry and pair sampling without the ray casting (the scans stay empty) -- what a
        shard plan needs; the 10,000-scan bag's poses take milliseconds, its scans 12 s."""
        # dense: 24 m x 16 m room (diagonal 28.8 m < 30 m) so every one of the 1081 beams returns
        self.width, self.height = (24.0, 16.0) if dense else (40.0, 25.0)
        self.margin = 3.0 if dense else 4.0
        self.segs = make_world(self.width, self.height, n_walls, self.margin, seed)
        self.truth = loop_trajectory(n_scans, self.width, self.height, self.margin, spacing=spacing, seed=seed)
        self.odom = odometry_from_truth(self.truth, seed=seed)
        self.scans, self.normals = [], []
        self.n_scans = n_scans
        if poses_only:
            return
        ranges, hit = raycast(self.truth, self.segs, seed=seed)
        d = self.segs[:, 2:4] - self.segs[:, 0:2]
        nrm = np.stack([-d[:, 1], d[:, 0]], axis=1)
        nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-12)
        rng = np.random.default_rng(seed + 4)
        for i in range(n_scans):
            pts, keep = ranges_to_cloud(ranges[i])
            self.scans.append(pts)
            nw = nrm[hit[i, keep]]
            c, s = math.cos(-self.truth[i, 2]), math.sin(-self.truth[i, 2])
            nl = np.stack([c * nw[:, 0] - s * nw[:, 1], s * nw[:, 0] + c * nw[:, 1]], axis=1)
            flip = np.sum(nl * pts, axis=1) > 0  # face the sensor
            nl[flip] *= -1
            nl += rng.normal(0, 0.02, nl.shape)
            nl /= np.maximum(np.linalg.norm(nl, axis=1, keepdims=True), 1e-12)
            self.normals.append(nl.astype(np.float32))
        self.n_scans = n_scans

    def sample_pairs(self, per_target=10, max_dist=1.5, min_sep=20, targets=None, seed=SEED):
        """(pair_src, pair_tgt_scan, theta0): sources within max_dist of each target's true pose.
        theta0 = AngleMod(odom_theta[src] - odom_theta[tgt]) (solver.cc:636-637 passes both headings)."""
        rng = np.random.default_rng(seed + 3)
        targets = np.arange(self.n_scans) if targets is None else np.asarray(targets)
        xy = self.truth[:, :2]
        src, tgt = [], []
        for t in targets:
            d = np.linalg.norm(xy - xy[t], axis=1)
            idx = np.arange(self.n_scans)
            cand = idx[(d < max_dist) & (np.abs(idx - t) > min_sep)]
            if len(cand) == 0:
                cand = idx[(d < max_dist) & (idx != t)]
            if len(cand) == 0:
                cand = idx[idx != t]
            src.extend(rng.choice(cand, per_target, replace=len(cand) < per_target))
            tgt.extend([t] * per_target)
        src, tgt = np.asarray(src, dtype=np.int32), np.asarray(tgt, dtype=np.int32)
        a = self.odom[src, 2] - self.odom[tgt, 2]
        theta0 = a - 2 * math.pi * np.rint(a / (2 * math.pi))
        return src, tgt, theta0

    def true_relative(self, src, tgt):
        """Ground-truth (tx, ty, theta) of scan src in scan tgt's frame."""
        ps, pt = self.truth[src], self.truth[tgt]
        d = ps[:2] - pt[:2]
        c, s = math.cos(pt[2]), math.sin(pt[2])
        a = ps[2] - pt[2]
        return c * d[0] + s * d[1], -s * d[0] + c * d[1], a - 2 * math.pi * round(a / (2 * math.pi))

    def correspondences(self, i, j, poses, outlier_threshold=0.25, max_points=None):
        """Nearest-neighbour correspondences of scan i (source) against scan j (target) under
        `poses`, gated at outlier_threshold (default_config.lua:66), like
        Solver::GetPointToPointMatching (solver.cc:132-172).  Returns an (N, 8) float32 block
        (source point, target point, source normal, target normal)."""
        from scipy.spatial import cKDTree
        ps, pt = self.scans[i], self.scans[j]
        if len(ps) == 0 or len(pt) == 0:
            return np.zeros((0, 8), dtype=np.float32)
        ci, si = math.cos(poses[i, 2]), math.sin(poses[i, 2])
        cj, sj = math.cos(poses[j, 2]), math.sin(poses[j, 2])
        w = np.stack([ci * ps[:, 0] - si * ps[:, 1] + poses[i, 0], si * ps[:, 0] + ci * ps[:, 1] + poses[i, 1]], 1)
        w -= poses[j, :2]
        q = np.stack([cj * w[:, 0] + sj * w[:, 1], -sj * w[:, 0] + cj * w[:, 1]], 1)
        dist, k = cKDTree(pt).query(q)
        m = np.nonzero(dist < outlier_threshold)[0]
        if max_points is not None:
            m = m[:max_points]
        return np.concatenate([ps[m], pt[k[m]], self.normals[i][m], self.normals[j][k[m]]], axis=1).astype(np.float32)

    def window_blocks(self, window=10, poses=None, **kw):
        """All (i, j) blocks, j in [i-window, i) (solver.cc:321-333)."""
        poses = self.odom if poses is None else poses
        blocks, src, tgt = [], [], []
        for i in range(self.n_scans):
            for j in range(max(i - window, 0), i):
                c = self.correspondences(i, j, poses, **kw)
                if len(c):
                    blocks.append(c)
                    src.append(i)
                    tgt.append(j)
        return blocks, np.asarray(src, dtype=np.int32), np.asarray(tgt, dtype=np.int32)
