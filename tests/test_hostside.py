"""Host-side callers / formats around the path: CPU only."""
import numpy as np

from nautilus_amd import hostside, synth


def test_scatter_score_and_candidates():
    line = np.stack([np.linspace(0, 5, 200), np.zeros(200)], 1)          # degenerate: a single wall
    box = np.random.default_rng(0).uniform(-3, 3, (500, 2))               # spread in both axes
    assert hostside.scatter_matrix_score(line) < 1e-6
    assert hostside.scatter_matrix_score(box) > 0.7
    poses = np.array([[0, 0, 0], [1, 0, 0], [6, 0, 0], [7, 0, 0], [12, 1, 0]], dtype=float)
    scans = [box, box, line, box, box]
    # node 1 is < 5 m from node 0; node 2 is far enough but a bad scan; node 3 is accepted; node 4: 5.1 m on
    assert hostside.lc_candidates(poses, scans) == [0, 3, 4]
    bag = synth.SynthBag(60)
    c = hostside.lc_candidates(bag.truth, bag.scans)
    assert all(np.linalg.norm(bag.truth[b, :2] - bag.truth[a, :2]) >= 5.0 for a, b in zip(c, c[1:]))


def test_pose_and_map_files_roundtrip(tmp_path):
    ts = [1583000000.123456, 1583000000.223456, 1583000001.0]
    poses = np.array([[0.5, -1.25, 0.1], [1.5, 2.0, -3.0], [100.0, 0.0, 3.14159]])
    p = tmp_path / "poses.txt"
    hostside.write_poses(p, ts, poses)
    first = open(p).readline().split()
    assert first == ["1583000000.123456", "0.500000", "-1.250000", "0.100000"]   # std::fixed, solver.cc:573-577
    got, missing = hostside.load_solution(p, ts + [5.0], np.zeros((4, 3)))
    assert missing == [3] and np.allclose(got[:3], poses, atol=1e-6)
    m = tmp_path / "map.txt"
    hostside.write_map_lines(m, [[0, 0, 1.5, 2], [-3.25, 4, 5, 6]])
    assert open(m).readline().strip() == "0,0,1.5,2"
    assert np.allclose(hostside.read_map_lines(m), [[0, 0, 1.5, 2], [-3.25, 4, 5, 6]])
    seg = hostside.hitl_segments({"line_a_start": (0, 0, 0), "line_a_end": (2, 2, 0), "line_b_start": (1, 0), "line_b_end": (1, 5)})
    assert seg.dtype == np.float32 and seg.tolist() == [[0, 0, 2, 2], [1, 0, 1, 5]]


def test_chi_square_gate_mirrors_lc_matcher():
    """ChiSquareScore / GetPossibleMatches (lc_matcher.cc:48-74) with a stand-in covariance provider:
    d^T cov^-1 d in float32, a scan never matches itself, threshold 5000."""
    import numpy as np
    from nautilus_amd import hostside
    poses = np.array([[0.0, 0.0, 0.0], [1.0, 0.0, 0.1], [0.0, 30.0, 0.2], [0.5, 0.5, 0.3]])
    cov = {(0, 1): np.eye(2) * 1e-3, (0, 2): np.eye(2) * 1e-3, (0, 3): np.array([[2e-3, 1e-3], [1e-3, 2e-3]])}
    calls = []

    def provider(pairs):
        calls.append(list(pairs))
        return np.stack([cov[p] for p in pairs]).astype(np.float32)
    assert abs(hostside.chi_square_score(cov[(0, 1)], poses[0, :2], poses[1, :2]) - 1000.0) < 0.5
    got = hostside.lc_possible_matches(0, [0, 1, 2, 3], poses, provider)
    assert calls == [[(0, 1), (0, 2), (0, 3)]]          # one batched covariance request, self excluded
    assert got == [1, 3]                                  # 1000 and 166.7 pass, 900000 does not
    assert hostside.lc_possible_matches(2, [2], poses, provider) == []


def test_chi_square_gate_oracle_against_the_numpy_statement(chi_square_cases):
    """oracle.chi_square_gate (ChiSquareScore + the acceptance of GetPossibleMatches, lc_matcher.cc:50-74, with
    Matrix2f::inverse() restated in closed form) against hostside.chi_square_score, which inverts with LAPACK: the same
    scores to float rounding of a 2 x 2 inverse, the same accept decisions away from the threshold."""
    import numpy as np
    from nautilus_amd import hostside
    from oracle import oracle as O
    poses, src, tgt, cov = chi_square_cases
    scores, flags = O.chi_square_gate(poses, src, tgt, cov, 5000.0)
    assert not flags[:8].any()
    assert not np.isfinite(scores[40:48]).all()
    assert not flags[48:56].any()                                  # 0 * inf = NaN never passes `score < 5000`
    ok = 0
    for i in range(56, len(src), 7):
        want = hostside.chi_square_score(cov[i], poses[src[i], :2], poses[tgt[i], :2])
        cond = np.linalg.cond(cov[i].astype(np.float64))
        assert abs(scores[i] - want) <= 4e-6 * cond * abs(want) + 1e-6, (i, scores[i], want, cond)
        if abs(want - 5000.0) > 1e-3 * cond * 5000.0:
            assert bool(flags[i]) == (want < 5000.0 and src[i] != tgt[i])
            ok += 1
    assert ok > 300 and 0 < flags.sum() < len(flags)
    # hand-checked values: identity-like covariance, d = (1, 2): (1 + 4) / 0.01
    s, f = O.chi_square_gate([[0, 0, 0], [1, 2, 0]], [0, 0], [1, 0], np.stack([np.eye(2) * 0.01] * 2), 5000.0)
    assert abs(s[0] - 500.0) < 1e-3 and f.tolist() == [1, 0]
    # lc_possible_matches through a backend gives what the pair-by-pair numpy walk gives
    from oracle.cpu_backend import OracleBackend
    P = np.array([[0.0, 0.0, 0.0], [1.0, 0.0, 0.1], [0.0, 30.0, 0.2], [0.5, 0.5, 0.3]])
    table = {(0, 1): np.eye(2) * 1e-3, (0, 2): np.eye(2) * 1e-3, (0, 3): np.array([[2e-3, 1e-3], [1e-3, 2e-3]])}
    provider = lambda pairs: np.stack([table[p] for p in pairs]).astype(np.float32)
    assert hostside.lc_possible_matches(0, [0, 1, 2, 3], P, provider, backend=OracleBackend()) == [1, 3]


def test_hitl_relevant_poses_mirror_get_relevant_poses_for_hitl():
    """GetRelevantPosesForHITL (solver.cc:479-513): float world points, DistanceToLineSegment<float> <= 0.05, a point
    on line a is not tested against line b, a pose needs 10 points and joins a before b."""
    from oracle import oracle as O
    rng = np.random.default_rng(1)
    line_a, line_b = np.array([0, 0, 10, 0], np.float32), np.array([0, 5, 10, 5], np.float32)
    on_a = np.stack([np.linspace(1, 9, 30), rng.uniform(-0.04, 0.04, 30)], 1)
    on_b = np.stack([np.linspace(1, 9, 12), 5 + rng.uniform(-0.04, 0.04, 12)], 1)
    few_a = on_a[:9]
    off = rng.uniform(1, 4, (50, 2))
    poses = np.array([[0, 0, 0], [1.0, 0.5, 0.3], [0, 0, 0], [0, 0, 0]], dtype=float)

    def to_scan(world, pose):  # inverse pose: world -> scan frame
        c, s = np.cos(-pose[2]), np.sin(-pose[2])
        d = world - pose[:2]
        return np.stack([c * d[:, 0] - s * d[:, 1], s * d[:, 0] + c * d[:, 1]], 1).astype(np.float32)
    scans = [to_scan(np.concatenate([on_a, on_b, off]), poses[0]), to_scan(np.concatenate([on_b, off]), poses[1]),
             to_scan(np.concatenate([few_a, off]), poses[2]), np.zeros((0, 2), np.float32)]
    a_poses, b_poses = hostside.hitl_relevant_poses(poses, scans, line_a, line_b)
    assert [i for i, _ in a_poses] == [0] and [i for i, _ in b_poses] == [1]   # pose 0 has both: it joins a only
    assert len(a_poses[0][1]) == 30 and len(b_poses[0][1]) == 12
    # the float distance used for the selection is the oracle's DistanceToLineSegment<float> (the six KATs pin it)
    w = np.array([[1.0, 1.0], [0.0, 2.0], [4.0, 4.0], [-2.0, -2.0], [2.0, 2.0], [2.5, 0.1]], np.float32)
    seg = np.array([0, 0, 2, 2], np.float32)
    got = hostside.distance_to_line_segment_f32(w, seg)
    want = np.array([O.dist_to_segment_f(p, seg[:2], seg[2:]) for p in w], np.float32)
    assert np.array_equal(got, want)
