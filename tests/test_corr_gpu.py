"""correspondence search (K5) and on-device normal equations.
K5 parity: matched rows and counts bit-exact against the oracle's restatement of
Solver::GetPointToPointMatching (solver.cc:132-172) + KDTree::FindNearestPoint (kdtree.cc:253-305);
the oracle's linear scan is itself checked against scipy's kd-tree (CPU test below)."""
import math

import numpy as np
import pytest

from nautilus_amd import _lib, csm, synth
from oracle import oracle as O


def _normals_table(bag):
    return np.concatenate(bag.normals).astype(np.float32)


def test_oracle_linear_scan_equals_kdtree(small_bag):
    """The restated nearest-neighbour rule agrees with an independent kd-tree (scipy, fp64): same
    neighbour distances to float rounding, same keep/drop decisions away from the threshold."""
    from scipy.spatial import cKDTree
    poses = small_bag.odom
    aff = O.pose_affines(poses)
    i, j = 14, 11
    rows, idx = O.corr_search_block(small_bag.scans[i], small_bag.normals[i], small_bag.scans[j],
                                    small_bag.normals[j], aff[i], aff[j], 0.25)
    assert 50 < len(rows) <= len(small_bag.scans[i])
    ci, si, cj, sj = math.cos(poses[i, 2]), math.sin(poses[i, 2]), math.cos(poses[j, 2]), math.sin(poses[j, 2])
    p = small_bag.scans[i].astype(np.float64)
    w = np.stack([ci * p[:, 0] - si * p[:, 1] + poses[i, 0] - poses[j, 0],
                  si * p[:, 0] + ci * p[:, 1] + poses[i, 1] - poses[j, 1]], 1)
    q = np.stack([cj * w[:, 0] + sj * w[:, 1], -sj * w[:, 0] + cj * w[:, 1]], 1)
    dist, k = cKDTree(small_bag.scans[j].astype(np.float64)).query(q)
    keep = dist < 0.25
    safe = np.abs(dist - 0.25) > 1e-4
    kept_src = {tuple(r[:2]) for r in rows}
    for n in np.nonzero(safe)[0]:
        assert (tuple(small_bag.scans[i][n]) in kept_src) == bool(keep[n])
    # matched target of every kept row is a true nearest neighbour (distance equal to float rounding)
    tgt = small_bag.scans[j].astype(np.float64)
    src_index = {tuple(pt): n for n, pt in enumerate(small_bag.scans[i])}
    for r, t in zip(rows, idx):
        n = src_index[tuple(r[:2])]
        assert abs(np.linalg.norm(tgt[t] - q[n]) - dist[n]) < 2e-5
    # rows are in source order, carry the inputs' normals, and tie/empty edge cases behave
    order = [src_index[tuple(r[:2])] for r in rows]
    assert order == sorted(order)
    assert np.array_equal(rows[:, 4:6], small_bag.normals[i][order])
    assert np.array_equal(rows[:, 6:8], small_bag.normals[j][idx])
    dup = np.array([[1.0, 1.0], [1.0, 1.0], [1.2, 1.0]], np.float32)  # duplicate target: lowest index wins
    rr, ii = O.corr_search_block(np.array([[1.0, 1.0]], np.float32), dup[:1], dup, dup, [1, 0, 0, 0], [1, 0, 0, 0])
    assert list(ii) == [0]
    rr, ii = O.corr_search_block(np.zeros((0, 2), np.float32), np.zeros((0, 2), np.float32), dup, dup,
                                 [1, 0, 0, 0], [1, 0, 0, 0])
    assert len(rr) == 0


@pytest.mark.gpu
def test_corr_search_bit_exact(gpu, small_bag):
    from nautilus_amd.correspondence import IcpBatch, window_pairs
    rng = np.random.default_rng(2)
    poses = small_bag.odom + rng.normal(0, [0.02, 0.02, math.radians(0.3)], small_bag.odom.shape)
    xy, off = csm.pack_scans(small_bag.scans)
    nrm = _normals_table(small_bag)
    bs, bt = window_pairs(small_bag.n_scans, 5)
    batch = IcpBatch(xy, nrm, off, bs, bt)
    batch.set_poses(poses)
    n = batch.search()
    rows, boff = batch.correspondences()
    want, counts, cap = O.corr_search_batch(xy, nrm, off, bs, bt, O.pose_affines(poses), 0.25)
    assert np.array_equal(np.diff(boff), counts) and n == counts.sum() and n > 50000
    for b in range(len(bs)):
        assert np.array_equal(rows[boff[b]:boff[b + 1]], want[cap[b]:cap[b] + counts[b]]), b
    assert np.array_equal(batch.d_cblock[:n].cpu().numpy(), np.repeat(np.arange(len(bs)), counts))
    # determinism
    batch.search()
    rows2, boff2 = batch.correspondences()
    assert rows2.tobytes() == rows.tobytes() and boff2.tobytes() == boff.tobytes()


@pytest.mark.gpu
def test_corr_search_edge_cases(gpu):
    """Empty scans, blocks with no match, more than 2048 points (several LDS passes), far-apart poses."""
    from nautilus_amd.correspondence import IcpBatch
    rng = np.random.default_rng(5)
    big = rng.uniform(-5, 5, (4500, 2)).astype(np.float32)
    scans = [np.zeros((0, 2), np.float32), big, (big + rng.normal(0, 0.05, big.shape)).astype(np.float32)[:3000],
             np.array([[0.0, 0.0]], np.float32), big[:7] + np.float32(100.0)]
    normals = [rng.normal(0, 1, s.shape).astype(np.float32) for s in scans]
    xy, off = csm.pack_scans(scans)
    nrm = np.concatenate(normals)
    bs = np.array([0, 1, 2, 1, 3, 4, 1, 2], np.int32)
    bt = np.array([1, 0, 1, 2, 1, 1, 4, 2], np.int32)
    poses = np.array([[0, 0, 0], [0.01, -0.02, 0.003], [0, 0, 0.001], [0.5, 0.5, 1.0], [0, 0, 0]], dtype=np.float64)
    batch = IcpBatch(xy, nrm, off, bs, bt)
    batch.set_poses(poses)
    n = batch.search()
    rows, boff = batch.correspondences()
    want, counts, cap = O.corr_search_batch(xy, nrm, off, bs, bt, O.pose_affines(poses), 0.25)
    assert np.array_equal(np.diff(boff), counts)
    assert counts[0] == 0 and counts[1] == 0 and counts[5] == 0 and counts[2] > 2000 and counts[7] == 3000
    for b in range(len(bs)):
        assert np.array_equal(rows[boff[b]:boff[b + 1]], want[cap[b]:cap[b] + counts[b]]), b


@pytest.mark.gpu
@pytest.mark.parametrize("thr", [0.25, 0.03, 1e-30, 7.5, 1e30])
def test_corr_search_hash_grid_corner_cases(gpu, thr):
    """The bucketed search must return what the exhaustive scan returns: exact ties between duplicate
    targets (lowest index), points on cell boundaries and at negative coordinates, non-finite points,
    coordinates too large for exact cell indices (-> exhaustive path), thresholds from 0 to 'everything'."""
    from nautilus_amd.correspondence import IcpBatch
    rng = np.random.default_rng(17)
    cell = np.float32(0.25 * 1.001)
    lattice = (np.stack(np.meshgrid(np.arange(-12, 12), np.arange(-12, 12)), -1).reshape(-1, 2) * cell).astype(np.float32)
    dup = np.repeat(rng.uniform(-3, 3, (150, 2)).astype(np.float32), 4, axis=0)      # four copies of each target
    wall = np.stack([np.linspace(-20, 20, 1081), np.full(1081, 2.0)], 1).astype(np.float32)
    weird = rng.uniform(-4, 4, (400, 2)).astype(np.float32)
    weird[5] = [np.nan, 1.0]
    weird[6] = [np.inf, 0.0]
    weird[7] = [-np.inf, np.nan]
    weird[8] = [3e7, 1.0]                                                              # huge but finite
    far = (rng.uniform(-1, 1, (300, 2)) + [2.0e7, -3.0e7]).astype(np.float32)
    scans = [lattice, (lattice + rng.normal(0, 0.02, lattice.shape)).astype(np.float32), dup,
             (dup[::4] + rng.normal(0, 0.01, (150, 2))).astype(np.float32), wall,
             (wall + rng.normal(0, 0.05, wall.shape)).astype(np.float32), weird,
             (weird + np.float32(0.01)).astype(np.float32), far, (far + np.float32(1.0)).astype(np.float32)]
    normals = [rng.normal(0, 1, s_.shape).astype(np.float32) for s_ in scans]
    xy, off = csm.pack_scans(scans)
    nrm = np.concatenate(normals)
    bs = np.array([1, 0, 3, 2, 5, 4, 7, 6, 6, 9, 8, 3], np.int32)
    bt = np.array([0, 1, 2, 3, 4, 5, 6, 7, 6, 8, 9, 0], np.int32)
    poses = rng.normal(0, [0.03, 0.03, 0.01], (len(scans), 3))
    poses[8:] = 0.0
    batch = IcpBatch(xy, nrm, off, bs, bt, outlier_threshold=thr)
    batch.set_poses(poses)
    n = batch.search()
    rows, boff = batch.correspondences()
    want, counts, cap = O.corr_search_batch(xy, nrm, off, bs, bt, O.pose_affines(poses), thr)
    assert np.array_equal(np.diff(boff), counts) and n == counts.sum()
    for b in range(len(bs)):
        assert rows[boff[b]:boff[b + 1]].tobytes() == want[cap[b]:cap[b] + counts[b]].tobytes(), (b, thr)
    if thr == 0.25:
        assert counts[2] == 150 and counts[3] == 600 and counts[4] > 300 and counts[8] > 380  # (the test is not vacuous)
    if thr == 1e-30:
        assert counts[8] >= 390 and counts[:8].sum() == 0  # only exactly coincident points survive (2e7 + 1 == 2e7 in float)


@pytest.mark.gpu
@pytest.mark.parametrize("offset_m", [800.0, 1030.0, 5000.0])
def test_corr_search_large_coordinates(gpu, small_bag, offset_m):
    """Clouds far from the origin: float cell coordinates lose the margin the hash grid's 3 x 3 visit relies on at
    ~4166 cells (1 km at the 0.25 m threshold), so beyond 4096 cells the kernel takes the exhaustive scan.  800 m
    stays on the hash path, 1030 m and 5 km do not; all return the exhaustive scan's rows, bit for bit."""
    from nautilus_amd.correspondence import IcpBatch
    rng = np.random.default_rng(5)
    shift = np.array([offset_m, -0.7 * offset_m], dtype=np.float32)
    scans = [(small_bag.scans[i] + shift).astype(np.float32) for i in (3, 4, 5, 6)]
    # dense rows of targets a hair apart around the threshold distance from the sources
    row = np.stack([np.linspace(0, 30, 1500), np.zeros(1500)], 1).astype(np.float32) + shift
    scans += [row, (row + np.float32([0.2499, 0.0])).astype(np.float32), (row + rng.normal(0, 0.15, row.shape)).astype(np.float32)]
    normals = [rng.normal(0, 1, s_.shape).astype(np.float32) for s_ in scans]
    xy, off = csm.pack_scans(scans)
    nrm = np.concatenate(normals)
    bs = np.array([1, 2, 3, 0, 5, 6, 4], np.int32)
    bt = np.array([0, 1, 2, 3, 4, 4, 6], np.int32)
    poses = np.zeros((len(scans), 3))
    poses[:4] = small_bag.odom[3:7] - small_bag.odom[3]
    batch = IcpBatch(xy, nrm, off, bs, bt, outlier_threshold=0.25)
    batch.set_poses(poses)
    n = batch.search()
    rows, boff = batch.correspondences()
    want, counts, cap = O.corr_search_batch(xy, nrm, off, bs, bt, O.pose_affines(poses), 0.25)
    assert np.array_equal(np.diff(boff), counts) and n == counts.sum() and counts.sum() > 1000
    for b in range(len(bs)):
        assert rows[boff[b]:boff[b + 1]].tobytes() == want[cap[b]:cap[b] + counts[b]].tobytes(), (b, offset_m)


@pytest.mark.gpu
@pytest.mark.parametrize("thr,min_cos", [(0.25, math.cos(math.radians(20.0))), (0.6, 0.5), (0.25, 0.0), (0.05, 0.999)])
def test_corr_search_with_normal_gate(gpu, small_bag, thr, min_cos):
    """Solver::GetPointToNormalMatching (solver.cc:177-260): nearest target within the threshold whose
    normal satisfies |n_t . n_s| > min_cos -- bit-exact rows against the oracle's linear scan, on real
    scan windows (hash path), oversized and far-away clouds (exhaustive path) and degenerate normals."""
    from nautilus_amd.correspondence import IcpBatch, window_pairs
    rng = np.random.default_rng(31)
    scans = list(small_bag.scans[:12])
    normals = [n.copy() for n in small_bag.normals[:12]]
    big = rng.uniform(-4, 4, (2600, 2)).astype(np.float32)                       # > 2048 targets: exhaustive path
    scans += [big, (big[:900] + rng.normal(0, 0.03, (900, 2))).astype(np.float32),
              (big[:50] + np.float32(3e7)).astype(np.float32)]                   # huge coordinates: exhaustive path
    nb = rng.normal(0, 1, big.shape).astype(np.float32)
    nb /= np.linalg.norm(nb, axis=1, keepdims=True)
    normals += [nb, nb[:900].copy(), nb[:50].copy()]
    normals[13][5] = (0.0, 0.0)                                                   # degenerate normal: never similar
    normals[13][6] = (np.nan, 1.0)
    xy, off = csm.pack_scans(scans)
    nrm = np.concatenate(normals).astype(np.float32)
    bs, bt = window_pairs(12, 3)
    bs = np.concatenate([bs, [13, 12, 14, 13]]).astype(np.int32)
    bt = np.concatenate([bt, [12, 13, 14, 13]]).astype(np.int32)
    poses = np.zeros((len(scans), 3))
    poses[:12] = small_bag.odom[:12] + rng.normal(0, [0.02, 0.02, 0.004], (12, 3))
    batch = IcpBatch(xy, nrm, off, bs, bt, outlier_threshold=thr, min_abs_cosine=min_cos)
    batch.set_poses(poses)
    n = batch.search()
    rows, boff = batch.correspondences()
    want, counts, cap = O.corr_search_gated_batch(xy, nrm, off, bs, bt, O.pose_affines(poses), thr, min_cos)
    assert np.array_equal(np.diff(boff), counts) and n == counts.sum()
    for b in range(len(bs)):
        assert rows[boff[b]:boff[b + 1]].tobytes() == want[cap[b]:cap[b] + counts[b]].tobytes(), (b, thr, min_cos)
    if min_cos == 0.0 and thr == 0.25:
        # |dot| > 0 keeps every finite, non-orthogonal pair: nearly the ungated result
        plain = IcpBatch(xy, nrm, off, bs[:20], bt[:20], outlier_threshold=thr)
        plain.set_poses(poses)
        assert abs(plain.search() - int(counts[:20].sum())) <= 0.01 * counts[:20].sum()
    if thr == 0.25 and min_cos > 0.9:
        assert 1000 < counts[:len(bs) - 4].sum() < n + 1   # the gate keeps a real subset on scan windows


@pytest.mark.gpu
def test_search_feeds_residuals_and_normal_equations(gpu, small_bag):
    """K5 -> K4 without leaving HBM; normal equations == J^T J, J^T r, r^T r of the oracle's
    autodiff Jacobians on the same correspondences."""
    from nautilus_amd.correspondence import IcpBatch, window_pairs
    poses = small_bag.odom.copy()
    xy, off = csm.pack_scans(small_bag.scans)
    nrm = _normals_table(small_bag)
    bs, bt = window_pairs(small_bag.n_scans, 3)
    batch = IcpBatch(xy, nrm, off, bs, bt)
    batch.set_poses(poses)
    n = batch.search()
    rows, boff = batch.correspondences()
    for kind in (_lib.NHIP_LIDAR_NORMAL, _lib.NHIP_LIDAR_POINT):
        res, js, jt = batch.residuals(kind)
        wr, w0, w1 = O.lidar_batch(kind, rows, boff, bs, bt, poses)
        assert np.max(np.abs(res.cpu().numpy() - wr)) < 1e-9
        assert np.max(np.abs(js.cpu().numpy().reshape(-1, 3) - w0)) < 1e-8
        neq = batch.normal_equations(kind).cpu().numpy()
        iu = np.triu_indices(6)
        for b in range(len(bs)):
            sl = slice(2 * boff[b], 2 * boff[b + 1])
            J = np.hstack([w0[sl], w1[sl]])
            H, g, rr = J.T @ J, J.T @ wr[sl], wr[sl] @ wr[sl]
            scale = max(1.0, np.abs(H).max())
            assert np.max(np.abs(neq[b, :21] - H[iu])) <= 1e-9 * scale, b
            assert np.max(np.abs(neq[b, 21:27] - g)) <= 1e-9 * scale and abs(neq[b, 27] - rr) <= 1e-9 * scale
    # a Gauss-Newton step assembled from the block systems reduces the cost (sanity of the reduction)
    neq = batch.normal_equations(_lib.NHIP_LIDAR_POINT).cpu().numpy()
    cost0 = neq[:, 27].sum()
    N = small_bag.n_scans
    H, g = np.zeros((3 * N, 3 * N)), np.zeros(3 * N)
    iu = np.triu_indices(6)
    for b in range(len(bs)):
        Hb = np.zeros((6, 6))
        Hb[iu] = neq[b, :21]
        Hb = Hb + Hb.T - np.diag(np.diag(Hb))
        ix = np.r_[3 * bs[b]:3 * bs[b] + 3, 3 * bt[b]:3 * bt[b] + 3]
        H[np.ix_(ix, ix)] += Hb
        g[ix] += neq[b, 21:27]
    H[:3, :3] += np.eye(3) * 1e9  # pose 0 held constant (solver.cc:384-386)
    step = np.linalg.solve(H + 1e-6 * np.eye(3 * N), -g)
    batch.set_poses(poses + step.reshape(N, 3))
    cost1 = batch.normal_equations(_lib.NHIP_LIDAR_POINT).cpu().numpy()[:, 27].sum()
    assert cost1 < cost0


@pytest.mark.gpu
def test_batches_of_one_backend_share_a_device_arena(gpu, small_bag):
    """A solver that rebuilds its problem per window pass (OptimizeOverGrowingWindow, solver.cc:339-355) builds thirty
    IcpBatch objects in a run; with one DeviceArena they upload the clouds once and share their work buffers.  Same
    numbers as private buffers, the clouds' device copies are the same tensors, the buffers do not grow for a smaller
    problem -- and a batch whose arena a NEWER batch has bound takes it back when it is used again (its correspondences
    searched anew at its poses) instead of reading the other batch's rows."""
    from nautilus_amd.correspondence import DeviceArena, IcpBatch, window_pairs
    import torch
    poses = small_bag.odom.copy()
    xy, off = csm.pack_scans(small_bag.scans)
    nrm = _normals_table(small_bag)
    arena = DeviceArena()
    arena.reserve(torch, torch.device("cuda:0"), int((np.diff(off) * np.minimum(np.arange(small_bag.n_scans), 4)).sum()),
                  int(np.minimum(np.arange(small_bag.n_scans), 4).sum()))
    ptr_padded = arena.buf["padded"].data_ptr()
    want = {}
    for w in (4, 2):  # (the larger problem first, as a reserved arena sees it)
        bs, bt = window_pairs(small_bag.n_scans, w)
        plain = IcpBatch(xy, nrm, off, bs, bt)
        plain.set_poses(poses)
        plain.search()
        want[w] = (plain.correspondences()[0].copy(), plain.normal_equations(_lib.NHIP_LIDAR_NORMAL).cpu().numpy().copy())
    batches = {}
    for w in (4, 2):
        bs, bt = window_pairs(small_bag.n_scans, w)
        b = batches[w] = IcpBatch(xy, nrm, off, bs, bt, arena=arena)
        b.set_poses(poses)
        b.search()
        assert np.array_equal(b.correspondences()[0], want[w][0])
        assert np.array_equal(b.normal_equations(_lib.NHIP_LIDAR_NORMAL).cpu().numpy(), want[w][1])
    assert batches[4].d_xy.data_ptr() == batches[2].d_xy.data_ptr(), "the clouds are uploaded once per arena"
    assert arena.buf["padded"].data_ptr() == ptr_padded, "reserved for the largest problem: no buffer grew"
    assert arena.owner is batches[2]
    # the older batch, used again: takes the arena back, uploads its block lists again, searches anew
    got = batches[4].normal_equations(_lib.NHIP_LIDAR_NORMAL).cpu().numpy()
    assert arena.owner is batches[4]
    assert np.array_equal(got, want[4][1])
    assert np.array_equal(batches[4].correspondences()[0], want[4][0])
    # ... and the newer one after that, likewise
    assert np.array_equal(batches[2].normal_equations(_lib.NHIP_LIDAR_NORMAL).cpu().numpy(), want[2][1])
    # The order PoseGraph._assemble_inner uses on an older graph (research=False): set_poses() FIRST -- which makes the batch
    # the arena's owner again without searching -- then the normal equations.  The correspondences must be the ones found
    # at the poses of the batch's last search(), evaluated at the new poses (round 5 read the newer batch's rows here).
    moved = poses + np.random.default_rng(3).normal(0.0, [0.01, 0.01, 0.003], poses.shape)
    bs, bt = window_pairs(small_bag.n_scans, 4)
    plain = IcpBatch(xy, nrm, off, bs, bt)
    plain.set_poses(poses)
    plain.search()
    plain.set_poses(moved)
    want_moved = plain.normal_equations(_lib.NHIP_LIDAR_NORMAL).cpu().numpy().copy()
    assert arena.owner is batches[2]
    batches[4].set_poses(moved)
    assert arena.owner is batches[4]
    assert np.array_equal(batches[4].normal_equations(_lib.NHIP_LIDAR_NORMAL).cpu().numpy(), want_moved)
    assert np.array_equal(batches[4].correspondences()[0], want[4][0]), "found at the poses of the last search()"
