"""Host-side callers and file formats either side of the hot path.
These stay on the host in the reference as well; they are restated here so the examples and the
end-to-end test can be driven the way nautilus drives them.  Nothing here is on the measured path.

  LCCandidateFilter::GetLCCandidates      /root/reference/src/loop_closure/lc_candidate_filter.cc:35-81
  LCMatcher::ChiSquareScore / GetPossibleMatches                 src/loop_closure/lc_matcher.cc:48-74
  pose file  "timestamp x y theta"        written solver.cc:565-579, read back main.cc:131-157
  map file   "x1,y1,x2,y2" per line       solver.cc:608-618
  HitlSlamInputMsg (two line segments)    msg/HitlSlamInputMsg.msg:1-4, solver.cc:467-478
"""
import numpy as np


def scatter_matrix_score(points):
    """min/max eigenvalue of the scan's scatter matrix, float32 like the reference (:35-51)."""
    p = np.asarray(points, dtype=np.float32).reshape(-1, 2)
    if len(p) == 0:
        return float("nan")
    mean = (np.float32(1.0 / len(p)) * p.sum(axis=0, dtype=np.float32)).astype(np.float32)
    d = p - mean
    s = (d[:, :, None] * d[:, None, :]).sum(axis=0, dtype=np.float32)
    ev = np.linalg.eigvals(s.astype(np.float32)).real
    return float(min(ev) / max(ev))


def lc_candidates(poses, scans, min_distance=5.0, min_score=0.70):
    """GetLCCandidates (:64-81): walk the nodes in order, skip nodes closer than 5 m to the last
    accepted scan, accept a node if its scatter score is >= 0.70."""
    out = []
    for i in range(len(scans)):
        if out:
            last = np.asarray(poses[out[-1]][:2], dtype=np.float32)
            here = np.asarray(poses[i][:2], dtype=np.float32)
            if float(np.linalg.norm(here - last)) < min_distance:
                continue
        if scatter_matrix_score(scans[i]) >= min_score:
            out.append(i)
    return out


def scatter_scores(backend, xy, offsets):
    """ComputeScatterMatrixScore of EVERY scan in one pass of the backend (product: nhip_lc_scatter_scores)."""
    return backend.scatter_scores(xy, offsets)


def lc_candidates_from_scores(poses, scores, min_distance=5.0, min_score=0.70):
    """GetLCCandidates (lc_candidate_filter.cc:64-81) on precomputed scores: walk the nodes in order, skip nodes
    closer than 5 m (float norm of the Vector2f translations, :53-62) to the last accepted scan, accept a node whose
    score is >= 0.70.  The walk is sequential (each accept depends on the previous one) and stays on the host."""
    out = []
    P = np.asarray(poses, dtype=np.float64)
    for i in range(len(scores)):
        if out:
            d = P[i, :2].astype(np.float32) - P[out[-1], :2].astype(np.float32)
            if np.float32(np.sqrt(np.float32(d[0] * d[0]) + np.float32(d[1] * d[1]))) < min_distance:
                continue
        if scores[i] >= min_score:
            out.append(i)
    return out


def geometric_pair_gate(poses, candidates, max_range=3.5, min_separation=20, backend=None):
    """All candidate pairs (later node = source, earlier = target) that pass the geometric gate: closer than
    max_range (lc_base_max_range, default_config.lua:122) and more than min_separation nodes apart -- in place of
    LCMatcher::GetPossibleMatches' per-pair ceres::Covariance (lc_matcher.cc:28-74).  With a backend the n x n
    flags come from it (product: nhip_lc_pair_gate); the unordered pairs are then read off on the host."""
    cand = np.asarray(candidates, dtype=np.int32)
    if len(cand) == 0:
        return np.zeros(0, np.int32), np.zeros(0, np.int32)
    if backend is None:
        from .posegraph import HipBackend
        backend = HipBackend()
    flags = backend.pair_gate(poses, cand, max_range, min_separation)
    i, j = np.nonzero(flags)
    keep = cand[i] > cand[j]
    return cand[i[keep]].astype(np.int32), cand[j[keep]].astype(np.int32)


def chi_square_score(cov2x2, source_xy, target_xy):
    """ChiSquareScore (lc_matcher.cc:50-57): d^T cov^-1 d with d = target - source translation, float32
    matrix as the reference casts it; cov is the cross-covariance block of the two poses."""
    cov = np.asarray(cov2x2, dtype=np.float32).reshape(2, 2)
    d = (np.asarray(target_xy, dtype=np.float32) - np.asarray(source_xy, dtype=np.float32)).astype(np.float32)
    return float(d @ np.linalg.inv(cov) @ d)


def lc_possible_matches(source, candidates, poses, covariance_fn, max_score=5000.0, backend=None):
    """GetPossibleMatches (lc_matcher.cc:59-74): every other candidate whose chi-square score against
    `source` is below 5000.  covariance_fn(pairs) -> (n, 2, 2) float32 (PoseGraph.cross_covariances).
    With a backend the scores and flags of all pairs come from one call of its chi_square_gate (product:
    nhip_lc_chi_square_gate); without one they are taken pair by pair with chi_square_score, the numpy statement
    of the same test that the backend is checked against in tests/."""
    others = [c for c in candidates if c != source]
    if not others:
        return []
    cov = covariance_fn([(source, c) for c in others])
    if backend is not None:
        _, flags = backend.chi_square_gate(poses, [source] * len(others), others, cov, max_score)
        return [c for c, f in zip(others, flags) if f]
    out = []
    for c, m in zip(others, cov):
        if chi_square_score(m, poses[source][:2], poses[c][:2]) < max_score:
            out.append(c)
    return out


def distance_to_line_segment_f32(points, seg):
    """DistanceToLineSegment<float> (slam_util.h:92-110) for an (n, 2) float32 array: Hyperplane::Through normal,
    projection inside the endpoints' closed x and y intervals -> |signed distance|, else nearest endpoint."""
    f = np.float32
    p = np.asarray(points, dtype=f).reshape(-1, 2)
    x0, y0, x1, y1 = (f(v) for v in seg)
    dx, dy = f(x1 - x0), f(y1 - y0)
    nx, ny = f(-dy), dx
    ln = f(np.sqrt(f(f(nx * nx) + f(ny * ny))))
    nx, ny = f(nx / ln), f(ny / ln)
    off = f(-f(f(x0 * nx) + f(y0 * ny)))
    sd = (p[:, 0] * nx + p[:, 1] * ny + off).astype(f)
    prx, pry = (p[:, 0] - sd * nx).astype(f), (p[:, 1] - sd * ny).astype(f)
    between = lambda v, a, b: ((v >= a) & (v <= b)) | ((v >= b) & (v <= a))
    inside = between(prx, x0, x1) & between(pry, y0, y1)
    d0 = np.sqrt((p[:, 0] - x0) ** 2 + (p[:, 1] - y0) ** 2).astype(f)
    d1 = np.sqrt((p[:, 0] - x1) ** 2 + (p[:, 1] - y1) ** 2).astype(f)
    return np.where(inside, np.abs(sd), np.minimum(d0, d1)).astype(f)


def hitl_relevant_poses(poses, scans, line_a, line_b, line_width=0.05, point_threshold=10):
    """GetRelevantPosesForHITL (solver.cc:479-513): per node, the points (scan frame) whose world position under the
    node's pose (Affine2f, float) lies within hitl_line_width of line a -- else of line b; a node with at least
    hitl_pose_point_threshold points on a joins line_a_poses, else with that many on b joins line_b_poses.
    Returns (a_poses, b_poses): lists of (node index, points (k, 2) float32)."""
    a_poses, b_poses = [], []
    for i, pts in enumerate(scans):
        pts = np.asarray(pts, dtype=np.float32).reshape(-1, 2)
        if len(pts) == 0:
            continue
        c, s = np.float32(np.cos(poses[i][2])), np.float32(np.sin(poses[i][2]))
        tx, ty = np.float32(poses[i][0]), np.float32(poses[i][1])
        w = np.stack([c * pts[:, 0] - s * pts[:, 1] + tx, s * pts[:, 0] + c * pts[:, 1] + ty], axis=1).astype(np.float32)
        on_a = distance_to_line_segment_f32(w, line_a) <= line_width
        on_b = ~on_a & (distance_to_line_segment_f32(w, line_b) <= line_width)
        if on_a.sum() >= point_threshold:
            a_poses.append((i, pts[on_a]))
        elif on_b.sum() >= point_threshold:
            b_poses.append((i, pts[on_b]))
    return a_poses, b_poses


def write_poses(path, timestamps, poses):
    """solver.cc:573-577: std::fixed (6 decimals) 'timestamp x y theta' per node."""
    with open(path, "w") as f:
        for t, p in zip(timestamps, poses):
            f.write("%.6f %.6f %.6f %.6f\n" % (t, p[0], p[1], p[2]))


def read_poses(path):
    """main.cc:139-146: whitespace-separated double timestamp + three floats; returns {timestamp: pose}."""
    out = {}
    with open(path) as f:
        for line in f:
            parts = line.split()
            if len(parts) < 4:
                break
            out[float(parts[0])] = np.array([np.float32(parts[1]), np.float32(parts[2]), np.float32(parts[3])], dtype=np.float32)
    return out


def load_solution(path, timestamps, poses):
    """LoadSolutionFromFile (main.cc:131-157): overwrite poses whose (6-decimal) timestamp is in the file."""
    table = read_poses(path)
    poses = np.array(poses, dtype=np.float64)
    missing = []
    for i, t in enumerate(timestamps):
        key = float("%.6f" % t)
        if key in table:
            poses[i] = table[key]
        else:
            missing.append(i)
    return poses, missing


def write_map_lines(path, lines):
    """solver.cc:612-616: 'x1,y1,x2,y2' per vectorised map line (default ostream precision: 6 significant)."""
    with open(path, "w") as f:
        for l in lines:
            f.write("%g,%g,%g,%g\n" % (l[0], l[1], l[2], l[3]))


def read_map_lines(path):
    return np.array([[float(v) for v in line.split(",")] for line in open(path) if line.strip()], dtype=np.float64).reshape(-1, 4)


def hitl_segments(msg):
    """LineSegmentsFromHitlMsg (solver.cc:467-478): msg = dict with line_a_start, line_a_end, line_b_start,
    line_b_end, each (x, y[, z]) -> two LineSegment<float> as (x0, y0, x1, y1) float32 rows."""
    row = lambda a, b: [msg[a][0], msg[a][1], msg[b][0], msg[b][1]]
    return np.array([row("line_a_start", "line_a_end"), row("line_b_start", "line_b_end")], dtype=np.float32)
