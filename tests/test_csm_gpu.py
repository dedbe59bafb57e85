"""GPU parity of the correlative scan matcher (K1 grid build, K2 correlation, K3 argmax) against
the CPU oracle, through the C ABI.  Bar: grids, integer sums and best-pose indices bit-exact;
scores equal as floats (same double expression rounded once)."""
import ctypes as C
import math

import numpy as np
import pytest

from nautilus_amd import _lib, csm, synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu

DEG = math.radians(1.0)


def _same_call(got, want):
    """(score, ((tx, ty), theta)) of the reference-shaped call against the oracle's: translation and rotation equal as
    floats; the score -- the fine optimum's mean log-likelihood on the unquantised table, a double sum of device / host
    logarithms rounded to float -- within 2e-7 relative (tolerance of north_star: 1e-5)."""
    return (abs(got[0] - want[0]) <= 2e-7 * abs(want[0]) and got[1][0][0] == want[1][0][0] and got[1][0][1] == want[1][0][1] and
            got[1][1] == want[1][1])


def _specs(range_m=30.0, res=0.05, sigma=2.0, max_shift=40, cell_bits=8):
    return (csm.grid_spec(range_m, res, sigma, 1e-10, max_shift, cell_bits),
            O.grid_spec(range_m, res, sigma, 1e-10, cell_bits))


def _check_pairs(scans_list, target_ids, pair_src, pair_slot, theta0, spec, ospec, search, origin=None):
    xy, off = csm.pack_scans(scans_list)
    st = csm.ScanTable(xy, off)
    grids = csm.LikelihoodGrids(st, target_ids, spec)
    got, sums = csm.match_pairs(st, grids, pair_src, pair_slot, theta0, search, origin)
    # every form of the matcher returns the same records: in batches of < 1024 pairs a pair with many candidates hands
    # rotations to a second kernel (<= 64 pairs: every pair does); NHIP_BNB_KERNELS=1 keeps everything in the pair's
    # workgroup, =2 hands over in any batch; the single kernel works rotation by rotation with register-held origins
    # for scans of <= 1088 points and through the workgroup queue otherwise (NHIP_BNB_QUEUE=1: always);
    # NHIP_BNB_LEVELS=1 leaves out the sub-block bounds.  Batches of >= 1024 pairs take the split form (bounds + seeds,
    # then the candidates of the pairs ordered by how many are left, the heaviest shared by several workgroups);
    # NHIP_BNB_SPLIT=1 brings it to small batches: in one round, or in rounds of three pairs whose candidates run on the
    # helper stream beside the next round's bounds, every pair shared by up to five workgroups
    import os
    for env in ({"NHIP_BNB_KERNELS": "1"}, {"NHIP_BNB_KERNELS": "1", "NHIP_BNB_LEVELS": "1"},
                {"NHIP_BNB_KERNELS": "1", "NHIP_BNB_QUEUE": "1"}, {"NHIP_BNB_KERNELS": "2", "NHIP_BNB_LEVELS": "1"},
                {"NHIP_BNB_KERNELS": "2", "NHIP_BNB_HEAVY_MIN": "1", "NHIP_BNB_KEEP_RANKS": "0"},
                {"NHIP_BNB_KERNELS": "1", "NHIP_BNB_SPLIT": "1"},
                {"NHIP_BNB_KERNELS": "1", "NHIP_BNB_SPLIT": "1", "NHIP_BNB_SPLIT_BATCH": "3", "NHIP_BNB_SPLIT_MIN": "1",
                 "NHIP_BNB_SPLIT_MAX": "5"},
                {"NHIP_BNB_KERNELS": "1", "NHIP_BNB_SPLIT": "1", "NHIP_BNB_SPLIT_BATCH": "2", "NHIP_BNB_SPLIT_OVERLAP": "0",
                 "NHIP_BNB_LEVELS": "1"}):
        os.environ.update(env)
        try:
            got_v, sums_v = csm.match_pairs(st, grids, pair_src, pair_slot, theta0, search, origin)
        finally:
            for k_ in env:
                os.environ.pop(k_, None)
        assert got_v.tobytes() == got.tobytes() and np.array_equal(sums_v, sums), env
    # the kernel that performs every add (csm_correlate_kernel / csm_correlate16_kernel by cell width) and the
    # branch-and-bound matcher return the same records; so does that kernel with every zero strip added
    ex = csm.search_spec(search.n_theta, search.nx, search.ny, search.theta_step, exhaustive=True)
    got_ex, sums_ex = csm.match_pairs(st, grids, pair_src, pair_slot, theta0, ex, origin)
    assert got_ex.tobytes() == got.tobytes() and np.array_equal(sums_ex, sums)
    os.environ["NHIP_CSM_DENSE"] = "1"
    try:
        got_d, sums_d = csm.match_pairs(st, grids, pair_src, pair_slot, theta0, ex, origin)
    finally:
        os.environ.pop("NHIP_CSM_DENSE", None)
    assert got_d.tobytes() == got.tobytes() and np.array_equal(sums_d, sums)
    if search.nx * search.ny <= 256:
        # lattices of few translations take the kernel whose lanes are poses (nhip_csm_small.hip) when every add is asked
        # for; NHIP_CSM_SMALL=0 sends them through the strip kernels like any other lattice
        os.environ["NHIP_CSM_SMALL"] = "0"
        try:
            got_s, sums_s = csm.match_pairs(st, grids, pair_src, pair_slot, theta0, ex, origin)
        finally:
            os.environ.pop("NHIP_CSM_SMALL", None)
        assert got_s.tobytes() == got.tobytes() and np.array_equal(sums_s, sums)
    ogr = O.grid_build_batch(xy, off, target_ids, ospec)
    oss = O.search_spec(search.n_theta, search.nx, search.ny, search.theta_step)
    want = O.csm_match_batch(xy, off, ogr, ospec, pair_src, pair_slot, theta0, oss, origin)
    for f in ("itheta", "ix", "iy"):
        assert np.array_equal(got[f], want[f]), (f, got[f], want[f])
    assert np.array_equal(sums, want["sum"])
    assert np.array_equal(got["score"], want["score"].astype(np.float32))
    grids.close()
    st.close()
    return got, want


def test_grid_bit_exact_and_border_zero(gpu, small_bag):
    spec, ospec = _specs()
    st = csm.ScanTable.from_list(small_bag.scans)
    ids = [0, 7, 23, 47]
    grids = csm.LikelihoodGrids(st, ids, spec)
    L = grids.layout
    for slot, sid in enumerate(ids):
        stored = grids.download(slot)
        want = O.grid_build(small_bag.scans[sid], ospec)
        inner = stored[L.pad:L.pad + L.side, L.pad:L.pad + L.side]
        assert np.array_equal(inner, want)
        assert want.max() > 200 and (want > 0).sum() > 1000
        # the matcher's tiled copies of the cells (two, the second shifted by 8 columns) hold the same bytes
        for cp in (0, 1):
            assert np.array_equal(grids.hi_plane(slot, copy=cp)[:, :L.rows], stored[:, :L.rows])
        border = stored.copy()
        border[L.pad:L.pad + L.side, L.pad:L.pad + L.side] = 0
        assert not border.any(), "zero border violated"
    grids.close()
    st.close()


def test_grid_16bit_cells_bit_exact(gpu, small_bag):
    """16-bit cells (65535 quantisation steps, 65536-entry threshold table on the device): the stored uint16 image
    equals the oracle's direct log() quantiser cell for cell; the border stays zero."""
    for args in ((30.0, 0.05, 2.0, 40), (10.0, 0.03, 1.0, 10), (30.0, 0.3, 2.0, 6)):
        spec, ospec = _specs(*args, cell_bits=16)
        st = csm.ScanTable.from_list(small_bag.scans[:10])
        grids = csm.LikelihoodGrids(st, [2, 7], spec)
        L = grids.layout
        assert L.cell_bytes == 2 and L.pitch % 16 == 0 and L.pitch >= 2 * L.rows
        for slot, sid in enumerate([2, 7]):
            stored = grids.download(slot)
            assert stored.dtype == np.uint16
            want = O.grid_build(small_bag.scans[sid], ospec)
            assert want.dtype == np.uint16 and want.max() > 50000
            assert np.array_equal(stored[L.pad:L.pad + L.side, L.pad:L.pad + L.side], want)
            border = stored[:, :L.rows].copy()
            border[L.pad:L.pad + L.side, L.pad:L.pad + L.side] = 0
            assert not border.any(), "zero border violated"
            # the plane of high bytes the matcher takes its block sums on: cell >> 8, zero wherever the image is
            hi = grids.hi_plane(slot)
            assert hi.shape == (L.rows, L.hi_pitch) and L.hi_pitch % 16 == 0 and L.hi_pitch >= L.rows
            assert np.array_equal(hi[:, :L.rows], (stored[:, :L.rows] >> 8).astype(np.uint8))
            assert not hi[:, L.rows:].any()
            # (on the device: two tiled copies, the second shifted by 8 columns; both hold the same plane)
            assert np.array_equal(grids.hi_plane(slot, copy=1), hi) and L.hi_bytes > 2 * L.rows * L.hi_pitch
            # ... and the tiled copy of the 16-bit cells that the exact pose sums read
            assert np.array_equal(grids.tiled16(slot)[:, :L.rows], stored[:, :L.rows])
        grids.close()
        st.close()


@pytest.mark.parametrize("range_m,res,sigma,max_shift", [(30.0, 0.3, 2.0, 6), (10.0, 0.03, 1.0, 10),
                                                         (30.0, 0.05, 0.7, 8), (4.0, 0.05, 5.0, 3)])
def test_grid_other_geometries(gpu, small_bag, range_m, res, sigma, max_shift):
    """side not a multiple of 4/64 (666), points outside the grid dropped (cimg_debug.h:48-50)."""
    spec, ospec = _specs(range_m, res, sigma, max_shift)
    st = csm.ScanTable.from_list(small_bag.scans[:6])
    grids = csm.LikelihoodGrids(st, [1, 4], spec)
    for slot, sid in enumerate([1, 4]):
        assert np.array_equal(grids.interior(slot), O.grid_build(small_bag.scans[sid], ospec))
    grids.close()
    st.close()


def test_match_small_lattice(gpu, small_bag):
    spec, ospec = _specs(max_shift=10)
    src, tgt, th0 = small_bag.sample_pairs(per_target=3, targets=[5, 20, 40], min_sep=2)
    ids = np.unique(tgt)
    slot = np.searchsorted(ids, tgt)
    _check_pairs(small_bag.scans, ids, src, slot, th0, spec, ospec, csm.search_spec(7, 21, 21, 2 * DEG))


def test_match_full_lattice_recovers_ground_truth(gpu):
    """BASELINE config #2 lattice (61 x 81 x 81, 1 deg / 5 cm) on a handful of pairs."""
    bag = synth.SynthBag(120)
    spec, ospec = _specs()
    src, tgt, th0 = bag.sample_pairs(per_target=2, targets=[10, 60, 110], max_dist=1.5, min_sep=3)
    ids = np.unique(tgt)
    slot = np.searchsorted(ids, tgt)
    search = csm.search_spec(61, 81, 81, DEG)
    got, _ = _check_pairs(bag.scans, ids, src, slot, th0, spec, ospec, search)
    for m, s, t, t0 in zip(got, src, tgt, th0):
        tx, ty, th = csm.match_to_transform(m, spec, search, t0)
        gx, gy, gth = bag.true_relative(s, t)
        assert abs(tx - gx) <= 0.11 and abs(ty - gy) <= 0.11, (tx, ty, gx, gy)
        assert abs(th - gth) <= math.radians(1.6)
        # mean log-likelihood of a true match sits far above the floor ln(1e-10) = -23.03
        # (same sign/scale convention as csm_score_threshold = -5, default_config.lua:85)
        assert m["score"] > -8.0


def test_score_volume_bit_exact(gpu, small_bag):
    """Every one of the n_theta*nx*ny sums, not just the argmax; non-square plane."""
    for bits in (8, 16):
        spec, ospec = _specs(max_shift=12, cell_bits=bits)
        st = csm.ScanTable.from_list(small_bag.scans)
        grids = csm.LikelihoodGrids(st, [12], spec)
        search = csm.search_spec(5, 25, 9, 3 * DEG)
        got = csm.score_volume(st, grids, 14, 0, 0.03, search)
        want = O.csm_scores(small_bag.scans[14], O.grid_build(small_bag.scans[12], ospec), ospec, 0.03,
                            O.search_spec(5, 25, 9, 3 * DEG))
        assert np.array_equal(got, want)
        assert got.max() > 0
        grids.close()
        st.close()


def test_score_volume_16bit_full_plane_every_alignment_class(gpu, small_bag):
    """csm_correlate16_kernel keeps two parity sets of (raw, hi) accumulators and four alignment classes (window start
    column mod 4), with the cells left of a lane's 28 travelling to its neighbour: EVERY sum of full 81 x 81 planes
    (all three lanes of a row, the neighbour exchanges, rows up to 80 = four strips) against the oracle, for search
    centres that put the windows into each class, with the skip map and with every strip added."""
    import os
    spec, ospec = _specs(max_shift=44, cell_bits=16)
    st = csm.ScanTable.from_list(small_bag.scans)
    grids = csm.LikelihoodGrids(st, [12], spec)
    og = O.grid_build(small_bag.scans[12], ospec)
    search, oss = csm.search_spec(3, 81, 81, 2 * DEG), O.search_spec(3, 81, 81, 2 * DEG)
    for origin in ((0, 0), (1, -2), (2, 3), (3, 1)):
        want = O.csm_scores(small_bag.scans[14], og, ospec, -0.02, oss, origin)
        got = csm.score_volume(st, grids, 14, 0, -0.02, search, origin)
        assert np.array_equal(got, want), origin
        assert want.max() > 65535 * 50
    os.environ["NHIP_CSM_DENSE"] = "1"
    try:
        got = csm.score_volume(st, grids, 14, 0, -0.02, search, (3, 1))
    finally:
        os.environ.pop("NHIP_CSM_DENSE", None)
    assert np.array_equal(got, want)
    grids.close()
    st.close()


def test_lattices_beyond_the_branch_and_bound_envelope_16bit(gpu, small_bag):
    """solver.cc:633-638 accepts any doubles: at the default cell width (16 bits) lattices of more than 88 x 88
    translations (here 97 x 101) or more rotations than the matcher's bounds fit in LDS (361 = +-180 degrees) take
    csm_correlate16_kernel and return the oracle's records; a spec that asks for skip maps at build time and a handle
    that builds them late agree."""
    src, tgt, th0 = small_bag.sample_pairs(per_target=2, targets=[9, 30], min_sep=2)
    ids = np.unique(tgt)
    slot = np.searchsorted(ids, tgt)
    spec, ospec = _specs(30.0, 0.05, 2.0, 60, cell_bits=16)
    got, _ = _check_pairs(small_bag.scans, ids, src, slot, th0, spec, ospec, csm.search_spec(3, 97, 101, 2 * DEG))
    got361, _ = _check_pairs(small_bag.scans, ids, src, slot, th0, spec, ospec, csm.search_spec(361, 9, 9, DEG))
    st = csm.ScanTable.from_list(small_bag.scans)
    with_map = csm.grid_spec(30.0, 0.05, 2.0, 1e-10, 60, 16, skip_map=True)
    grids = csm.LikelihoodGrids(st, ids, with_map)
    again, _ = csm.match_pairs(st, grids, src, slot, th0, csm.search_spec(3, 97, 101, 2 * DEG))
    assert again.tobytes() == got.tobytes()
    # the map behind a 16-bit image against its definition: rows [r, r + 21) x dwords [c, c + 42)
    L = grids.layout
    stored = grids.download(0)
    got_map = grids.skip_map(0)
    bits = np.unpackbits(got_map, axis=1, bitorder="little")[:, :L.pitch // 4]
    assert np.array_equal(bits, skip_map_definition(stored.view(np.uint8).reshape(L.rows, L.pitch), width=42))
    grids.close()
    st.close()


def test_wide_plane_uses_several_plane_blocks(gpu, small_bag):
    """nx > 84 and ny > 85: more than one plane block per rotation."""
    spec, ospec = _specs(30.0, 0.05, 2.0, 60)
    src, tgt, th0 = small_bag.sample_pairs(per_target=2, targets=[9], min_sep=2)
    _check_pairs(small_bag.scans, [9], src, [0, 0], th0, spec, ospec, csm.search_spec(3, 101, 121, 2 * DEG))


def test_ragged_empty_and_out_of_grid(gpu, small_bag):
    """Empty source scan, empty target, 1-point scans, points far outside the grid."""
    far = np.array([[500.0, -700.0], [29.99, 29.99], [-30.0, -30.0], [30.0, 30.0]], dtype=np.float32)
    scans = [np.zeros((0, 2), np.float32), small_bag.scans[3], np.array([[1.0, 2.0]], np.float32), far,
             small_bag.scans[4][:65], small_bag.scans[5][:64], small_bag.scans[6][:63]]
    spec, ospec = _specs(max_shift=8)
    pair_src = [0, 1, 2, 3, 4, 5, 6, 1, 3]
    pair_slot = [1, 0, 1, 1, 1, 1, 1, 2, 3]  # targets: [0 (empty), 1, 2 (one point), 3 (far)]
    th0 = np.linspace(-0.2, 0.2, len(pair_src))
    got, want = _check_pairs(scans, [0, 1, 2, 3], pair_src, pair_slot, th0, spec, ospec,
                             csm.search_spec(5, 17, 17, 2 * DEG))
    assert got[0]["itheta"] == 0 and got[0]["ix"] == 0 and got[0]["iy"] == 0  # empty scan: first index
    assert got[0]["score"] == np.float32(math.log(1e-10))


def test_long_clouds_cross_staging_batches(gpu, small_bag):
    """Source clouds longer than one LDS batch of rotated cells (1152 points) and longer than the
    16-bit SWAR unpack interval many times over; targets with several thousand points."""
    long_a = np.concatenate([small_bag.scans[3], small_bag.scans[4] + np.float32(0.02), small_bag.scans[5]])
    long_b = np.concatenate([small_bag.scans[i] for i in (6, 7, 8, 9, 10)])
    assert len(long_a) > 2 * 1152 and len(long_b) > 4000
    spec, ospec = _specs(max_shift=12)
    got, want = _check_pairs([long_a, long_b, small_bag.scans[3]], [0, 1, 2], [0, 1, 0, 1, 2], [2, 2, 1, 0, 1],
                             [0.0, 0.05, -0.04, 0.1, 0.0], spec, ospec, csm.search_spec(7, 25, 25, DEG))
    assert want["sum"].max() > 255 * 1152  # sums beyond what a single unpack interval could hold


@pytest.mark.parametrize("cell_bits", [16, 8])
def test_small_plane_kernel(gpu, small_bag, cell_bits):
    """Every add on lattices of few translations (csm_small_plane_kernel: lanes are poses, up to 256 of them): the coarse
    lattice of the drop-in call (181 x 13 x 13), one pose, 255 poses, planes wider than tall and taller than wide; scans of
    1 to 3,243 points incl. an empty one and a non-finite point, search centres off the middle -- against the oracle, the
    branch-and-bound matcher and the strip kernels (_check_pairs runs all of them)."""
    long_a = np.concatenate([small_bag.scans[3], small_bag.scans[4] + np.float32(0.02), small_bag.scans[5]])
    odd = small_bag.scans[9].copy()
    odd[7] = (np.float32(np.nan), np.float32(1.0))
    odd[11] = (np.float32(np.inf), np.float32(-2.0))
    scans = [small_bag.scans[6], long_a, small_bag.scans[7][:1], np.zeros((0, 2), np.float32), odd, small_bag.scans[8]]
    src, slot = [0, 1, 2, 3, 4, 5, 0], [1, 0, 1, 0, 0, 0, 0]
    th0 = [0.02, -0.05, 0.3, 0.0, 0.07, -0.01, 3.1]
    spec, ospec = _specs(max_shift=14, cell_bits=cell_bits)
    for lat in ((181, 13, 13), (3, 1, 1), (5, 15, 17), (7, 3, 25), (9, 25, 5), (1, 13, 19)):
        _check_pairs(scans, [0, 1], src, slot, th0, spec, ospec, csm.search_spec(lat[0], lat[1], lat[2], DEG))
    # search centres off the middle (the fine level of a two-level search)
    org = [[3, -2], [0, 0], [-5, 4], [1, 1], [0, -6], [2, 2], [-1, 0]]
    _check_pairs(scans, [0, 1], src, slot, th0, spec, ospec, csm.search_spec(21, 13, 13, DEG / 10), origin=org)


def test_scan_lengths_around_the_held_origins(gpu, small_bag):
    """The matcher keeps a rotation's window origins for scans of up to 17 * 64 = 1088 points (by-rotation kernel);
    longer scans take the general kernel.  Lengths on both sides of that limit, of a 64-point chunk and of the
    16-lane runs of the compressed bounds phase, 8- and 16-bit cells, in ONE batch (both kernels run)."""
    pool = np.concatenate([small_bag.scans[i] for i in (3, 4, 5, 6)])
    lengths = [1, 15, 16, 17, 63, 64, 65, 1023, 1024, 1025, 1087, 1088, 1089, 1151, 1152, 1153]
    scans = [np.ascontiguousarray(pool[7 * i:7 * i + n]) for i, n in enumerate(lengths)] + [small_bag.scans[8]]
    tgt = len(scans) - 1
    src = list(range(len(lengths)))
    th0 = [0.01 * (i - 8) for i in src]
    for bits in (8, 16):
        spec, ospec = _specs(max_shift=12, cell_bits=bits)
        _check_pairs(scans, [tgt], src, [0] * len(src), th0, spec, ospec, csm.search_spec(5, 25, 25, DEG))


def test_run_lists_at_their_extremes(gpu, small_bag):
    """The bounds phase compresses consecutive points that share a pooled entry into runs (an entry carries its first
    point's index, the gather takes a run's length from the next entry, a sentinel ends the last run).  Its extremes:
    every point of the scan in ONE cell (runs of 64, the longest a list entry can say, and a lane's 16-bit fields at
    their limit), every point in a block of its own (64 entries per 64 points: the ring of 128 fills), both mixed, at
    lengths that end on and off a 64-point chunk -- against the oracle and every other form of the matcher."""
    rng = np.random.default_rng(11)
    base = small_bag.scans[8]
    wall = base[np.argsort(np.hypot(base[:, 0], base[:, 1]))[:40]]          # points that do score against the target

    def one_cell(n):
        return np.repeat(wall[:1], n, axis=0).astype(np.float32)

    def own_block(n):  # consecutive points 1 m apart, back and forth: no two neighbours share a pooled entry
        t = np.arange(n)
        return np.stack([wall[0, 0] + (t % 2) * 1.0 + (t // 2 % 5) * 0.45, wall[0, 1] + (t % 3) * 0.9], 1).astype(np.float32)

    def mixed(n):
        a = np.concatenate([one_cell(70), own_block(70), np.repeat(wall[5:6], 200, axis=0), wall, one_cell(700)])
        return np.ascontiguousarray(a[:n]).astype(np.float32)

    scans = [one_cell(64), one_cell(128), one_cell(1088), one_cell(1081), one_cell(65),
             own_block(64), own_block(128), own_block(1088), own_block(1081), own_block(193),
             mixed(1081), mixed(1088), mixed(1089), one_cell(1300), own_block(1300),
             # entry 0 = 64 points, 62 single-point entries, then 64 points of another cell: list entries 0 and 64 fall to the
             # same lane, whose 16-bit fields would overflow -- the wave reduces between the two gather passes
             np.concatenate([one_cell(64), own_block(62), np.repeat(wall[5:6], 64, axis=0)]).astype(np.float32), base]
    tgt = len(scans) - 1
    src = list(range(len(scans) - 1))
    th0 = [0.02 * (i - 7) for i in src]
    for bits in (16, 8):
        spec, ospec = _specs(max_shift=12, cell_bits=bits)
        got, want = _check_pairs(scans, [tgt], src, [0] * len(src), th0, spec, ospec, csm.search_spec(5, 25, 25, DEG))
        assert want["sum"][0] > 0  # (the one-cell scans do score: 64 x one cell's value)


def test_hand_over_policies_agree_on_a_large_batch(gpu, small_bag):
    """1,500 pairs (past the 1,024 from which pairs keep their rotations): every hand-over policy -- none, pairs with
    >= 8 candidates from their third rotation on, every pair everything -- returns the records of the kernel that
    performs every add, byte for byte; so does a 1,000-pair batch (below the limit: the default policy hands over)."""
    import os
    n = len(small_bag.scans)
    rng = np.random.default_rng(5)
    src = rng.integers(0, n, 1500).astype(np.int32)
    tgt = rng.integers(0, n, 1500).astype(np.int32)
    th0 = rng.uniform(-0.3, 0.3, 1500)
    ids = np.unique(tgt)
    slot = np.searchsorted(ids, tgt).astype(np.int32)
    spec, _ = _specs(max_shift=20)
    search = csm.search_spec(21, 41, 41, DEG)
    st = csm.ScanTable.from_list(small_bag.scans)
    grids = csm.LikelihoodGrids(st, ids, spec)
    ex = csm.search_spec(21, 41, 41, DEG, exhaustive=True)
    want, want_sums = csm.match_pairs(st, grids, src, slot, th0, ex)
    for count in (1500, 1000):
        for env in ({}, {"NHIP_BNB_KERNELS": "1"},
                    {"NHIP_BNB_KERNELS": "2", "NHIP_BNB_HEAVY_MIN": "8", "NHIP_BNB_KEEP_RANKS": "2"},
                    {"NHIP_BNB_KERNELS": "2", "NHIP_BNB_HEAVY_MIN": "1", "NHIP_BNB_KEEP_RANKS": "0"}):
            os.environ.update(env)
            try:
                got, sums = csm.match_pairs(st, grids, src[:count], slot[:count], th0[:count], search)
            finally:
                for k_ in env:
                    os.environ.pop(k_, None)
            assert got.tobytes() == want[:count].tobytes() and np.array_equal(sums, want_sums[:count]), (count, env)
    grids.close()
    st.close()


def test_swar_fields_do_not_overflow_in_one_alignment_class(gpu):
    """Worst case for the 16-bit SWAR fields: thousands of source points whose windows all start in the
    same alignment class (x = multiples of 4 cells) and sit on cells of the maximum value 255 (the
    target is the same set of points, so every hit cell is a blur centre), visited with theta = 0 so
    nothing spreads them.  The per-class unpack rule must keep every field <= 255 * 255."""
    rng = np.random.default_rng(99)
    res = 0.05
    n = 3000
    cx = rng.integers(-40, 40, n) * 4            # column = 600 + cx: all in class (600 + cx - 40 + pad) & 3 = const
    cy = rng.integers(-150, 150, n)
    pts = np.stack([(cx + 0.5) * res, (cy + 0.5) * res], 1).astype(np.float32)
    dup = np.concatenate([pts, pts[:1500]])      # exact duplicates too (same cell: counted n times)
    spec, ospec = _specs(max_shift=40)
    for th in (0.0, math.pi / 2):                # pi/2: rows and columns swap -> another single class
        got, want = _check_pairs([dup, pts], [1], [0, 0], [0, 0], [th, th + 1e-3], spec, ospec,
                                 csm.search_spec(3, 81, 81, DEG))
        assert want["sum"].max() > 255 * 2000


@pytest.mark.parametrize("cell_bits", [16, 8])
def test_beams_piled_into_few_cells(gpu, cell_bits):
    """The candidates' phase keeps consecutive beams that fall into one stored cell as ONE list entry with a count
    (16-bit grids) and its packed sums hold 16-bit fields over groups of 8 lanes: at most 257 points per group.  Scans
    whose beams pile up -- all 1081 in one cell; runs of 60-64 beams per cell (entries of count 60+ in neighbouring
    lanes: the wave must fall back to one point per entry); alternating long and short runs; the same with the walls
    of a real scan behind them -- give the oracle's records in every form of the matcher."""
    rng = np.random.default_rng(5)
    res = 0.05
    one_cell = np.tile(np.array([[3.0 + 0.01, -2.0 + 0.01]], np.float32), (1081, 1))
    runs = np.repeat(np.stack([(np.arange(18) * 3 + 0.5) * res + 1.0, np.full(18, 2.0125)], 1), 60, axis=0)[:1081].astype(np.float32)
    mixed = []
    for i in range(40):
        k = 64 if i % 2 == 0 else 3
        mixed.append(np.tile(np.array([[(i * 2 + 0.5) * res - 2.0, (i % 7 + 0.5) * res + 1.0]]), (k, 1)))
    mixed = np.concatenate(mixed).astype(np.float32)[:1081]
    bag = synth.SynthBag(3, dense=True)
    piled = np.concatenate([np.tile(bag.scans[0][:1], (500, 1)), bag.scans[0][500:]]).astype(np.float32)
    jitter = (one_cell + rng.uniform(0, 0.002, one_cell.shape)).astype(np.float32)   # same cell, different floats
    scans = [one_cell, runs, mixed, piled, jitter, bag.scans[1], bag.scans[0]]
    spec, ospec = _specs(cell_bits=cell_bits)
    src = [0, 1, 2, 3, 4, 0, 1, 2, 3, 4, 6, 5]
    slot = [0, 1, 2, 3, 0, 5, 5, 6, 6, 3, 3, 6]
    th = [0.0, 0.01, -0.02, 0.03, 0.0, 0.3, -0.3, 0.1, 0.02, -0.01, 0.0, 0.05]
    got, want = _check_pairs(scans, list(range(7)), src, slot, th, spec, ospec, csm.search_spec(61, 81, 81, DEG))
    assert want["sum"][0] > 1000 * (200 if cell_bits == 8 else 50000)   # 1081 points on the peak of their own blur


def test_non_finite_points_are_off_grid(gpu, small_bag):
    """NaN / inf / absurd coordinates (a broken range reading) never fault and never score: dropped
    from a target raster, floor-only as source points -- same answer as the oracle."""
    bad = small_bag.scans[7].copy()
    bad[5] = (np.nan, 1.0)
    bad[6] = (2.0, np.inf)
    bad[7] = (-np.inf, np.nan)
    bad[8] = (1e12, -1e12)
    bad[9] = (3e38, 0.0)
    scans = [bad, small_bag.scans[9], np.full((3, 2), np.nan, np.float32)]
    spec, ospec = _specs(max_shift=8)
    got, want = _check_pairs(scans, [0, 1, 2], [0, 1, 2, 1, 0], [1, 0, 1, 2, 0], np.zeros(5), spec, ospec,
                             csm.search_spec(5, 17, 17, 2 * DEG))
    assert want["sum"][2] == 0 and want["sum"][3] == 0  # all-NaN source / all-NaN target score nothing


def test_rotation_wraparound(gpu, small_bag):
    """theta0 near +-pi, where AngleMod (math_util.h:81-84) wraps."""
    spec, ospec = _specs(max_shift=6)
    rot = []
    for a in (math.pi - 1e-3, -math.pi + 1e-3, 3.0, -3.1):
        c, s = math.cos(a), math.sin(a)
        p = small_bag.scans[8]
        rot.append(np.stack([c * p[:, 0] - s * p[:, 1], s * p[:, 0] + c * p[:, 1]], 1).astype(np.float32))
    scans = [small_bag.scans[8]] + rot
    th0 = [-(math.pi - 1e-3), -(-math.pi + 1e-3), -3.0, 3.1]
    got, _ = _check_pairs(scans, [0], [1, 2, 3, 4], [0, 0, 0, 0], th0, spec, ospec,
                          csm.search_spec(5, 13, 13, DEG))
    assert np.all(np.abs(got["itheta"] - 2) <= 1)  # the rotated copies realign at the lattice centre


def test_search_origin(gpu, small_bag):
    spec, ospec = _specs(max_shift=30)
    src, tgt, th0 = small_bag.sample_pairs(per_target=3, targets=[30], min_sep=2)
    origin = np.array([[5, -7], [-20, 20], [0, 11]], dtype=np.int32)
    _check_pairs(small_bag.scans, [30], src, [0, 0, 0], th0, spec, ospec, csm.search_spec(3, 21, 19, DEG), origin)
    st = csm.ScanTable.from_list(small_bag.scans)
    grids = csm.LikelihoodGrids(st, [30], spec)
    with pytest.raises(_lib.NhipError):  # centre + half-width beyond the stored border
        csm.match_pairs(st, grids, [1], [0], [0.0], csm.search_spec(3, 21, 19, DEG), [[25, 0]])


def test_argument_errors(gpu, small_bag):
    spec, _ = _specs(max_shift=4)
    st = csm.ScanTable.from_list(small_bag.scans[:4])
    grids = csm.LikelihoodGrids(st, [0, 1], spec)
    ok = csm.search_spec(3, 9, 9, DEG)
    for bad in [csm.search_spec(4, 9, 9, DEG), csm.search_spec(3, 8, 9, DEG), csm.search_spec(3, 11, 9, DEG)]:
        with pytest.raises(_lib.NhipError):
            csm.match_pairs(st, grids, [0], [0], [0.0], bad)
    with pytest.raises(_lib.NhipError):
        csm.match_pairs(st, grids, [9], [0], [0.0], ok)  # source out of range
    with pytest.raises(_lib.NhipError):
        csm.match_pairs(st, grids, [0], [2], [0.0], ok)  # grid slot out of range
    with pytest.raises(_lib.NhipError):
        csm.LikelihoodGrids(st, [7], spec)


def test_deterministic_bytes(gpu, small_bag):
    spec, _ = _specs(max_shift=10)
    st = csm.ScanTable.from_list(small_bag.scans)
    grids = csm.LikelihoodGrids(st, [3, 9], spec)
    src, slot, th0 = [1, 2, 5, 7] * 8, [0, 1, 1, 0] * 8, np.linspace(-0.3, 0.3, 32)
    s = csm.search_spec(9, 21, 21, DEG)
    a, sa = csm.match_pairs(st, grids, src, slot, th0, s)
    b, sb = csm.match_pairs(st, grids, src, slot, th0, s)
    assert a.tobytes() == b.tobytes() and sa.tobytes() == sb.tobytes()


def test_full_size_properties(gpu):
    """BASELINE config #2 sizes (1081-beam scans, 1200^2 grid, 61x81x81 lattice) checked through
    size-independent properties, on more pairs than the oracle is asked to redo:
      (a) self-match: a scan against its own grid peaks at the lattice centre with the scan's own
          grid sum; (b) integer cell shift equivariance: searching with centre (cx, cy) equals the
          zero-centre volume shifted; (c) a batch equals the concatenation of its halves."""
    bag = synth.SynthBag(64, dense=True)
    assert all(len(s) == synth.N_BEAMS for s in bag.scans)
    spec, ospec = _specs()
    st = csm.ScanTable.from_list(bag.scans)
    ids = np.arange(0, 64, 4)
    grids = csm.LikelihoodGrids(st, ids, spec)
    search = csm.search_spec(61, 81, 81, DEG)
    # (a)
    got, sums = csm.match_pairs(st, grids, ids, np.arange(len(ids)), np.zeros(len(ids)), search)
    assert np.all(got["itheta"] == 30) and np.all(got["ix"] == 40) and np.all(got["iy"] == 40)
    for slot, sid in enumerate(ids[:3]):
        g = grids.interior(slot)
        p = bag.scans[sid].astype(np.float64)
        c = 600 + np.floor(p[:, 0] / 0.05).astype(int)
        r = 600 + np.floor(p[:, 1] / 0.05).astype(int)
        assert sums[slot] == int(g[r, c].astype(np.int64).sum())
    # (b)
    s_small = csm.search_spec(3, 21, 21, DEG)
    v0 = csm.score_volume(st, grids, 5, 1, 0.02, csm.search_spec(3, 41, 41, DEG))
    v1 = csm.score_volume(st, grids, 5, 1, 0.02, s_small, origin=(7, -9))
    assert np.array_equal(v1, v0[:, 10 + 7:31 + 7, 10 - 9:31 - 9])
    # (c)
    src, tgt, th0 = bag.sample_pairs(per_target=4, targets=ids, max_dist=2.0, min_sep=1)
    slot = np.searchsorted(ids, tgt)
    whole, ws = csm.match_pairs(st, grids, src, slot, th0, search)
    h = len(src) // 2
    a, sa = csm.match_pairs(st, grids, src[:h], slot[:h], th0[:h], search)
    b, sb = csm.match_pairs(st, grids, src[h:], slot[h:], th0[h:], search)
    assert whole.tobytes() == np.concatenate([a, b]).tobytes()
    # and a sample of the batch against the oracle
    pick = np.arange(0, len(src), 13)
    ogr = O.grid_build_batch(st.xy, st.offsets, ids, ospec)
    want = O.csm_match_batch(st.xy, st.offsets, ogr, ospec, src[pick], slot[pick], th0[pick],
                             O.search_spec(61, 81, 81, DEG))
    for f in ("itheta", "ix", "iy"):
        assert np.array_equal(whole[f][pick], want[f])
    assert np.array_equal(ws[pick], want["sum"])
    grids.close()
    st.close()


def test_random_configuration_sweep(gpu, small_bag):
    """Seeded sweep over grid geometry, blur, lattice shape, search centre and ragged clouds:
    every case bit-exact against the oracle (indices, sums, scores)."""
    rng = np.random.default_rng(20201114)
    for case in range(24):
        res = float(rng.choice([0.03, 0.05, 0.1, 0.3]))
        range_m = float(rng.choice([3.0, 10.0, 30.0]))
        sigma = float(rng.choice([0.6, 1.0, 2.0, 3.5]))
        nx = int(rng.choice([1, 3, 21, 81, 85, 97]))
        ny = int(rng.choice([1, 5, 33, 81, 87, 101]))
        nth = int(rng.choice([1, 3, 7]))
        max_shift = max(nx, ny) // 2 + int(rng.integers(0, 12))
        spec, ospec = _specs(range_m, res, sigma, max_shift)
        k = int(rng.integers(3, 7))
        pick = rng.choice(len(small_bag.scans), k, replace=False)
        scans = []
        for i in pick:
            s = small_bag.scans[i]
            m = int(rng.choice([len(s), 1, 63, 64, 65, 200]))
            scans.append(s[:m] * np.float32(rng.choice([1.0, 0.3])))  # some clouds shrink into small grids
        n_pairs = int(rng.integers(1, 9))
        src = rng.integers(0, k, n_pairs)
        tgt_ids = np.unique(rng.integers(0, k, max(1, k // 2)))
        slot = rng.integers(0, len(tgt_ids), n_pairs)
        th0 = rng.uniform(-math.pi, math.pi, n_pairs)
        origin = None
        if rng.random() < 0.5:
            room_x, room_y = max_shift - nx // 2, max_shift - ny // 2
            origin = np.stack([rng.integers(-room_x, room_x + 1, n_pairs),
                               rng.integers(-room_y, room_y + 1, n_pairs)], 1).astype(np.int32)
        search = csm.search_spec(nth, nx, ny, float(rng.choice([0.5, 1.0, 3.0])) * DEG)
        _check_pairs(scans, tgt_ids, src, slot, th0, spec, ospec, search, origin)


@pytest.mark.parametrize("cell_bits", [16, 8])
def test_drop_in_class_matches_oracle_two_level_search(gpu, small_bag, cell_bits):
    """CorrelativeScanMatcher(30, 2, 0.3, 0.01).GetTransformation(...) as solver.cc:633-638 calls it (the Python
    mirror and the C++ header both call nhip_csm_get_transformation): the coarse (0.3 m) then fine (0.01 m,
    6000 x 6000 grid) searches against the oracle's independent restatement, float for float."""
    a, b = small_bag.scans[17][::3], small_bag.scans[15][::3]   # thinned: the 0.01 m oracle grid is 36 / 72 MB
    rot_a, rot_b = small_bag.odom[17, 2], small_bag.odom[15, 2]
    m = csm.CorrelativeScanMatcher(30, 2, 0.3, 0.01, cell_bits=cell_bits)
    score, ((tx, ty), th) = m.GetTransformation(a, b, rot_a, rot_b, math.radians(90))
    want = O.two_level_match(a, b, rot_a, rot_b, math.radians(90), 30.0, 2.0, 0.3, 0.01, cell_bits=cell_bits)
    assert _same_call((score, ((tx, ty), th)), want)
    gx, gy, gth = small_bag.true_relative(17, 15)
    assert abs(tx - gx) < 0.06 and abs(ty - gy) < 0.06 and abs(th - gth) < 0.02
    # other constructor arguments: a coarser fine level and a narrower rotation range
    m2 = csm.CorrelativeScanMatcher(20, 1.5, 0.25, 0.05, cell_bits=cell_bits)
    got = m2.GetTransformation(a, b, rot_a, rot_b, math.radians(20))
    want = O.two_level_match(a, b, rot_a, rot_b, math.radians(20), 20.0, 1.5, 0.25, 0.05, cell_bits=cell_bits)
    assert _same_call(got, want)


def test_drop_in_cache_serves_repeated_targets(gpu, small_bag):
    """SolveAutoLC -> GetRelativeTransform (solver.cc:630-649, 676-700) matches many sources against one target in a row:
    the target's two tables are built by the first of those calls and kept (least recently used out first under a byte
    cap).  Results do not depend on whether a call hit the cache, on the order of the calls, or on the cap; the floats
    are the oracle's."""
    thin = lambda i: small_bag.scans[i][::3]
    m = csm.CorrelativeScanMatcher(30, 2, 0.3, 0.01)
    call = lambda s_, t_: m.GetTransformation(thin(s_), thin(t_), small_bag.odom[s_, 2], small_bag.odom[t_, 2], math.radians(90))
    csm.drop_in_cache_clear()
    csm.drop_in_cache_configure(3 << 30)
    base = csm.drop_in_cache_stats()
    assert base["entries"] == 0 and base["bytes"] == 0
    targets, sources = [15, 22, 30], [17, 19, 24, 27]
    first = {(s_, t_): call(s_, t_) for t_ in targets for s_ in sources}           # per target: 1 miss, 3 hits
    st = csm.drop_in_cache_stats()
    assert st["entries"] == 3 and st["misses"] - base["misses"] == 3 and st["hits"] - base["hits"] == 9
    assert 3 * 200e6 < st["bytes"] < 3 * 400e6                                      # ~0.3 GB per target at (30, 2, 0.3, 0.01)
    again = {(s_, t_): call(s_, t_) for s_ in sources for t_ in targets}           # other order: all hits
    assert again == first and csm.drop_in_cache_stats()["hits"] - st["hits"] == 12
    # a target whose cloud differs in ONE float is another target
    other = thin(15).copy()
    other[5, 0] = np.nextafter(other[5, 0], np.float32(10))
    m.GetTransformation(thin(17), other, small_bag.odom[17, 2], small_bag.odom[15, 2], math.radians(90))
    assert csm.drop_in_cache_stats()["entries"] == 4
    # other constructor arguments: other tables
    m2 = csm.CorrelativeScanMatcher(20, 1.5, 0.25, 0.05)
    g2 = m2.GetTransformation(thin(17), thin(15), small_bag.odom[17, 2], small_bag.odom[15, 2], math.radians(20))
    assert csm.drop_in_cache_stats()["entries"] == 5
    # a cap below one target: nothing is kept, every call builds; same floats
    csm.drop_in_cache_configure(1 << 20)
    assert csm.drop_in_cache_stats()["entries"] == 0
    nocache = {k: call(*k) for k in list(first)[:3]}
    assert all(nocache[k] == first[k] for k in nocache) and csm.drop_in_cache_stats()["entries"] == 0
    assert m2.GetTransformation(thin(17), thin(15), small_bag.odom[17, 2], small_bag.odom[15, 2], math.radians(20)) == g2
    csm.drop_in_cache_configure(3 << 30)
    # the oracle's independent restatement, float for float (one pair per target)
    for t_ in targets:
        want = O.two_level_match(thin(17), thin(t_), small_bag.odom[17, 2], small_bag.odom[t_, 2], math.radians(90), 30.0, 2.0,
                                 0.3, 0.01, cell_bits=16)
        got = first[(17, t_)]
        assert _same_call(got, want)
    # several host threads on one target (ctypes releases the GIL): one of them builds, or two do and one entry stays
    csm.drop_in_cache_clear()
    import concurrent.futures as cf
    with cf.ThreadPoolExecutor(4) as ex:
        par = list(ex.map(lambda s_: call(s_, 22), sources * 2))
    assert par == [first[(s_, 22)] for s_ in sources * 2]
    csm.drop_in_cache_clear()
    assert csm.drop_in_cache_stats()["entries"] == 0


def test_drop_in_flat_landscape_first_pose_wins_across_the_parts(gpu, small_bag):
    """A source that scores nothing anywhere (every point beyond the table): every pose of both levels sums to 0 and the
    FIRST pose of each lattice is the answer -- also across the three workgroups the fine level's rotations are dealt over
    (their records tie; the host takes the smaller index), and through the small-plane kernel of the coarse level."""
    far = (small_bag.scans[17][::3] + np.float32(500.0)).astype(np.float32)
    b = small_bag.scans[15][::3]
    m = csm.CorrelativeScanMatcher(30, 2, 0.3, 0.01)
    got = m.GetTransformation(far, b, 0.3, 0.1, math.radians(90))
    want = O.two_level_match(far, b, 0.3, 0.1, math.radians(90), 30.0, 2.0, 0.3, 0.01, cell_bits=16)
    assert _same_call(got, want)
    assert got[0] == pytest.approx(math.log(1e-10))          # the floor: nothing scored
    # an empty source: the same
    e = m.GetTransformation(np.zeros((0, 2), np.float32), b, 0.3, 0.1, math.radians(90))
    we = O.two_level_match(np.zeros((0, 2), np.float32), b, 0.3, 0.1, math.radians(90), 30.0, 2.0, 0.3, 0.01, cell_bits=16)
    assert _same_call(e, we)


def _dropin_info():
    out = (C.c_double * 4)()
    _lib.check(_lib.load().nhip_csm_get_transformation_info(out))
    return {"coarse_score": out[0], "fine_form": int(out[1]), "chained": out[2] == 1.0, "coarse_itheta": int(out[3])}


def test_drop_in_fine_level_forms_return_the_same_floats(gpu, small_bag):
    """Round 6: the fine level of GetTransformation performs every add in the kernel whose lanes are poses (NHIP_SEARCH_LATENCY:
    tiles of four rows of the 61 x 61 plane; ~45 us whatever the clouds), both levels chained on the device.  The forms it
    replaced stay selectable (NHIP_DROPIN_FINE=bnb: the branch-and-bound matcher, which a flat landscape costs milliseconds
    on the 6000 x 6000 table; =strips: the strip kernels): on a matching pair and on a pair 6 m apart all three return the
    SAME floats, which are the oracle's -- chained or with the host between the levels (NHIP_DROPIN_CHAIN=0)."""
    import os
    thin = lambda i: small_bag.scans[i][::3]
    m = csm.CorrelativeScanMatcher(30, 2, 0.3, 0.01)
    near = (thin(17), thin(15), small_bag.odom[17, 2], small_bag.odom[15, 2], math.radians(90))
    j = int(np.argmax(np.hypot(*(small_bag.truth[:, :2] - small_bag.truth[15, :2]).T)))          # the scan farthest from 15
    far = (thin(j), thin(15), small_bag.odom[j, 2], small_bag.odom[15, 2], math.radians(90))
    got = {}
    try:
        for mode, form in (("bnb", 0), ("strips", 1), (None, 2)):
            if mode is None:
                os.environ.pop("NHIP_DROPIN_FINE", None)
            else:
                os.environ["NHIP_DROPIN_FINE"] = mode
            for name, args in (("near", near), ("far", far)):
                got[(mode, name)] = m.GetTransformation(*args)
                info = _dropin_info()
                assert info["chained"] and info["fine_form"] == form, (mode, info)
        os.environ["NHIP_DROPIN_CHAIN"] = "0"
        for name, args in (("near", near), ("far", far)):
            got[("unchained", name)] = m.GetTransformation(*args)
            assert not _dropin_info()["chained"]
    finally:
        os.environ.pop("NHIP_DROPIN_FINE", None)
        os.environ.pop("NHIP_DROPIN_CHAIN", None)
    for name in ("near", "far"):
        assert got[("bnb", name)] == got[("strips", name)] == got[(None, name)] == got[("unchained", name)], name
    for name, args in (("near", near), ("far", far)):
        want = O.two_level_match(args[0], args[1], args[2], args[3], args[4], 30.0, 2.0, 0.3, 0.01, cell_bits=16)
        assert _same_call(got[(None, name)], want), name


@pytest.mark.parametrize("cell_bits", [16, 8])
def test_small_plane_kernel_in_tiles_of_rows(gpu, small_bag, cell_bits):
    """NHIP_SEARCH_LATENCY: planes of more than 256 translations through csm_small_plane_kernel in tiles of whole rows (one
    workgroup per pair, rotation and tile) -- lattices whose rows do and do not divide into the tiles, a search centre off
    the origin, a short and an empty scan: records and sums equal the oracle's and the branch-and-bound matcher's."""
    spec, ospec = _specs(10.0, 0.05, 2.0, 40, cell_bits)
    scans = [small_bag.scans[i][::2] for i in (3, 4, 12, 13)] + [small_bag.scans[5][:7], np.zeros((0, 2), np.float32)]
    xy, off = csm.pack_scans(scans)
    st = csm.ScanTable(xy, off)
    ids = np.array([0, 2], dtype=np.int32)
    grids = csm.LikelihoodGrids(st, ids, spec)
    ogr = O.grid_build_batch(xy, off, ids, ospec)
    src = np.array([1, 3, 4, 5], dtype=np.int32)
    slot = np.array([0, 1, 1, 0], dtype=np.int32)
    th0 = np.array([0.02, -0.03, 0.4, 0.0])
    org = np.array([[3, -2], [0, 0], [-5, 4], [1, 1]], dtype=np.int32)
    for nth, nx, ny in ((5, 61, 61), (3, 81, 33), (7, 17, 71), (3, 81, 5)):
        search = csm.search_spec(nth, nx, ny, 0.5 * DEG, exhaustive=True, latency=True)
        got, sums = csm.match_pairs(st, grids, src, slot, th0, search, org if max(nx, ny) <= 70 else None)
        want = O.csm_match_batch(xy, off, ogr, ospec, src, slot, th0, O.search_spec(nth, nx, ny, 0.5 * DEG), org if max(nx, ny) <= 70 else None)
        for f in ("itheta", "ix", "iy"):
            assert np.array_equal(got[f], want[f]), (nth, nx, ny, f)
        assert np.array_equal(sums, want["sum"]), (nth, nx, ny)
        if nx <= 88 and ny <= 88:
            bnb, bsums = csm.match_pairs(st, grids, src, slot, th0, csm.search_spec(nth, nx, ny, 0.5 * DEG),
                                         org if max(nx, ny) <= 70 else None)
            assert bnb.tobytes() == got.tobytes() and np.array_equal(bsums, sums)
    grids.close()
    st.close()


def test_device_pointer_api_on_torch_stream(gpu, small_bag):
    """The *_dev entry points: caller-owned HBM (torch tensors), launched on torch's stream."""
    import torch
    dev = torch.device("cuda:0")
    spec, ospec = _specs(max_shift=10)
    L = csm.grid_layout(spec)
    xy, off = csm.pack_scans(small_bag.scans)
    ids = np.array([2, 11], dtype=np.int32)
    src = np.array([3, 4, 12, 13], dtype=np.int32)
    slot = np.array([0, 0, 1, 1], dtype=np.int32)
    th0 = np.array([0.01, -0.02, 0.03, 0.0])
    search = csm.search_spec(9, 21, 21, DEG)
    lib = _lib.load()
    t = lambda a: torch.from_numpy(a).to(dev)
    d_xy, d_off, d_ids, d_src, d_slot = t(xy), t(off), t(ids), t(src), t(slot)
    d_rot0, d_delta = t(csm.rot0_table(th0)), t(csm.delta_table(search))
    d_grids = torch.empty(lib.nhip_grids_bytes(C.byref(spec), 2), dtype=torch.uint8, device=dev)
    ws_bytes = lib.nhip_grid_workspace_bytes(C.byref(spec), 2)
    d_ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    d_keys = torch.empty(4, dtype=torch.int64, device=dev)
    d_out = torch.empty(4 * 4, dtype=torch.int32, device=dev)
    d_sums = torch.empty(4, dtype=torch.int32, device=dev)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        sp = C.c_void_p(stream.cuda_stream)
        n_sc = d_off.numel() - 1
        _lib.check(lib.nhip_grid_build_dev(d_xy.data_ptr(), d_off.data_ptr(), n_sc, d_ids.data_ptr(), 2, C.byref(spec),
                                           d_grids.data_ptr(), d_ws.data_ptr(), ws_bytes, sp))
        # without a workspace (each pair's workgroup evaluates its own candidates) ...
        _lib.check(lib.nhip_csm_match_dev(d_xy.data_ptr(), d_off.data_ptr(), n_sc, d_grids.data_ptr(), 2, C.byref(spec),
                                          d_src.data_ptr(), d_slot.data_ptr(), d_rot0.data_ptr(),
                                          d_delta.data_ptr(), None, 4, C.byref(search), d_keys.data_ptr(),
                                          d_out.data_ptr(), d_sums.data_ptr(), None, 0, sp))
        stream.synchronize()
        first = (d_out.cpu().numpy().copy(), d_sums.cpu().numpy().copy())
        # ... and with the grid-wide candidate list (a tiny one too: what does not fit stays with the workgroup)
        for nbytes in (lib.nhip_csm_workspace_bytes(4), 256 + 8 * 16 * 8):
            d_cws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            d_out.zero_()
            _lib.check(lib.nhip_csm_match_dev(d_xy.data_ptr(), d_off.data_ptr(), n_sc, d_grids.data_ptr(), 2, C.byref(spec),
                                              d_src.data_ptr(), d_slot.data_ptr(), d_rot0.data_ptr(),
                                              d_delta.data_ptr(), None, 4, C.byref(search), d_keys.data_ptr(),
                                              d_out.data_ptr(), d_sums.data_ptr(), d_cws.data_ptr(), nbytes, sp))
            stream.synchronize()
            assert np.array_equal(d_out.cpu().numpy(), first[0]) and np.array_equal(d_sums.cpu().numpy(), first[1])
    stream.synchronize()
    got = d_out.cpu().numpy().view(csm.MATCH_DTYPE)
    ogr = O.grid_build_batch(xy, off, ids, ospec)
    want = O.csm_match_batch(xy, off, ogr, ospec, src, slot, th0, O.search_spec(9, 21, 21, DEG))
    for f in ("itheta", "ix", "iy"):
        assert np.array_equal(got[f], want[f])
    assert np.array_equal(d_sums.cpu().numpy(), want["sum"])
    slots = d_grids[:2 * L.slot_bytes].cpu().numpy().reshape(2, L.slot_bytes)
    g = slots[:, :L.grid_bytes].reshape(2, L.rows, L.pitch)
    assert np.array_equal(g[1, L.pad:L.pad + L.side, L.pad:L.pad + L.side], ogr[1])
    # the skip map behind each image, against its definition on the stored image
    mp = 8 * ((L.pitch // 4 + 63) // 64)
    for t_ in range(2):
        got_map = slots[t_, L.grid_bytes:L.grid_bytes + L.rows * mp].reshape(L.rows, mp)
        bits = np.unpackbits(got_map, axis=1, bitorder="little")[:, :L.pitch // 4]
        assert np.array_equal(bits, skip_map_definition(g[t_]))
        assert not np.unpackbits(got_map, axis=1, bitorder="little")[:, L.pitch // 4:].any()


def skip_map_definition(stored, width=21):
    """include/nautilus_hip.h (nhip_grid_layout_t.skip_bytes): bit (r, c) = any non-zero cell in stored rows
    [r, r + 21) x aligned dwords [c, c + 21 * cell_bytes), clipped to the image.  stored: (rows, pitch) bytes."""
    rows, pitch = stored.shape
    nz = stored.reshape(rows, pitch // 4, 4).any(axis=2)
    big = np.zeros((rows + 21, pitch // 4 + width), dtype=np.int64)
    big[:rows, :pitch // 4] = nz
    I = np.zeros((big.shape[0] + 1, big.shape[1] + 1), dtype=np.int64)
    I[1:, 1:] = big.cumsum(0).cumsum(1)
    r = np.arange(rows)[:, None]
    c = np.arange(pitch // 4)[None, :]
    cnt = I[r + 21, c + width] - I[r, c + width] - I[r + 21, c] + I[r, c]
    return (cnt > 0).astype(np.uint8)


@pytest.mark.parametrize("cell_bits", [16, 8])
def test_zero_strip_skipping_changes_nothing(gpu, small_bag, monkeypatch, cell_bits):
    """The kernels that perform every add (NHIP_SEARCH_EXHAUSTIVE): NHIP_CSM_DENSE=1 adds every strip, zero or not;
    the default leaves the all-zero ones out through the skip map (16-bit grids: built late by the handle).  Records
    and integer sums must be identical (and equal the oracle's)."""
    xy, off = csm.pack_scans(small_bag.scans)
    ids = np.array([3, 11], dtype=np.int32)
    src = np.array([5, 9, 14, 2, 30, 31], dtype=np.int32)
    slot = np.array([0, 1, 1, 0, 1, 0], dtype=np.int32)
    th0 = np.array([0.1, -0.3, 0.0, 2.0, 0.7, -1.2])
    spec = csm.grid_spec(30.0, 0.05, 2.0, 1e-10, 40, cell_bits)
    search = csm.search_spec(7, 81, 81, DEG, exhaustive=True)
    st = csm.ScanTable(xy, off)
    grids = csm.LikelihoodGrids(st, ids, spec)
    m1, s1 = csm.match_pairs(st, grids, src, slot, th0, search)
    if cell_bits == 16:
        assert grids.skip_map(0).any(), "the first exhaustive search on 16-bit grids builds their skip maps"
    monkeypatch.setenv("NHIP_CSM_DENSE", "1")
    m2, s2 = csm.match_pairs(st, grids, src, slot, th0, search)
    monkeypatch.delenv("NHIP_CSM_DENSE")
    assert np.array_equal(s1, s2) and m1.tobytes() == m2.tobytes()
    ospec = O.grid_spec(30.0, 0.05, 2.0, 1e-10, cell_bits)
    ogr = O.grid_build_batch(xy, off, ids, ospec)
    want = O.csm_match_batch(xy, off, ogr, ospec, src, slot, th0, O.search_spec(7, 81, 81, DEG))
    assert np.array_equal(s1, want["sum"])
    for f in ("itheta", "ix", "iy"):
        assert np.array_equal(m1[f], want[f])
    grids.close(), st.close()


# ---------------------------------------------------------------------------- branch and bound, 16-bit cells
def _pool_numpy(stored, cell_bits, stride=8):
    """pool[i][j] = max of stored[8i : 8i + 15, 8j : 8j + 15] (clipped; stride 4: [4i : 4i + 7, 4j : 4j + 7]);
    16-bit cells scaled by ceil(max / 257)."""
    rows = stored.shape[0]
    n = (rows + stride - 1) // stride
    win = 2 * stride - 1
    out = np.zeros((n, n), dtype=np.int64)
    for i in range(n):
        band = stored[stride * i:stride * i + win, :rows].max(axis=0).astype(np.int64)
        for j in range(n):
            out[i, j] = band[stride * j:stride * j + win].max()
    return out if cell_bits == 8 else (out + 256) // 257


@pytest.mark.parametrize("cell_bits", [8, 16])
def test_pooled_table_matches_its_definition(gpu, small_bag, cell_bits):
    spec, _ = _specs(10.0, 0.05, 2.0, 12, cell_bits)
    st = csm.ScanTable.from_list(small_bag.scans[:4])
    grids = csm.LikelihoodGrids(st, [1, 3], spec)
    L = grids.layout
    for slot in (0, 1):
        stored = grids.download(slot)[:, :L.rows]
        pool = grids.pooled(slot)
        want = _pool_numpy(stored, cell_bits)
        n = want.shape[0]
        assert np.array_equal(pool[:n, :n], want) and want.max() >= 200
        rest = pool.copy()
        rest[:n, :n] = 0
        assert not rest.any(), "rows / columns beyond the image must stay zero"
        if cell_bits == 16:
            assert np.all(257 * pool[:n, :n].astype(np.int64) >= _pool_numpy(stored, 8))  # 257 * ceil(m / 257) >= m
        # second level: 7 x 7 cells at stride 4
        # second level: 7 x 7 cells at stride 4, stored as byte pairs {P4[i][j], P4[i + 1][j]}
        pairs = grids.pooled(slot, level=2)
        want4 = _pool_numpy(stored, cell_bits, stride=4)
        n4 = want4.shape[0]
        assert pairs.shape == (L.pool4_rows, L.pool4_pitch)
        assert np.array_equal(pairs[:n4, 0:2 * n4:2], want4)
        assert np.array_equal(pairs[:n4 - 1, 1:2 * n4:2], want4[1:]) and not pairs[n4 - 1, 1::2].any()
        rest = pairs.copy()
        rest[:n4, :2 * n4] = 0
        assert not rest.any()
        # a level-1 entry covers its four level-2 entries
        assert np.all(pool[:n, :n][:n4 // 2, :n4 // 2] >= want4[:2 * (n4 // 2):2, :2 * (n4 // 2):2])
    grids.close()
    st.close()


def test_match_16bit_cells_against_oracle(gpu, small_bag):
    """The 16-bit path end to end (grid, pooled bounds, exact block sums, argmax, score) on several lattices,
    including search centres and a lattice with partial blocks (21 = 2 * 8 + 5)."""
    spec, ospec = _specs(max_shift=12, cell_bits=16)
    src, tgt, th0 = small_bag.sample_pairs(per_target=3, targets=[5, 20, 40], min_sep=2)
    ids = np.unique(tgt)
    slot = np.searchsorted(ids, tgt)
    _check_pairs(small_bag.scans, ids, src, slot, th0, spec, ospec, csm.search_spec(7, 21, 21, 2 * DEG))
    _check_pairs(small_bag.scans, ids, src, slot, th0, spec, ospec, csm.search_spec(3, 9, 17, 1 * DEG),
                 origin=np.array([[3, -2]] * len(src), dtype=np.int32))
    _check_pairs(small_bag.scans, ids, src, slot, th0, spec, ospec, csm.search_spec(1, 1, 1, 1 * DEG))


def test_full_lattice_16bit_and_8bit_agree_with_oracle_and_each_other(gpu):
    """BASELINE config #2 lattice on dense 1081-beam scans, both cell widths bit-exact against their oracles;
    the branch-and-bound matcher evaluates only a small fraction of the blocks."""
    import os
    bag = synth.SynthBag(200, dense=True)
    src, tgt, th0 = bag.sample_pairs(per_target=3, targets=[30, 90, 150, 199], max_dist=3.5, min_sep=20)
    ids = np.unique(tgt)
    slot = np.searchsorted(ids, tgt)
    search = csm.search_spec(61, 81, 81, DEG)
    os.environ["NHIP_BNB_STATS"] = os.environ["NHIP_BNB_INSTRUMENT"] = "1"  # (the instrumented build of the kernels)
    try:
        csm.bnb_stats()
        for bits in (8, 16):
            spec, ospec = _specs(cell_bits=bits)
            _check_pairs(bag.scans, ids, src, slot, th0, spec, ospec, search)
            csm.bnb_stats()  # (reset: _check_pairs ran the matcher in several forms)
            st = csm.ScanTable.from_list(bag.scans)
            grids = csm.LikelihoodGrids(st, ids, spec)
            csm.match_pairs(st, grids, src, slot, th0, search)
            grids.close(), st.close()
            lv = csm.bnb_stats_levels()
            tot, ev = lv["blocks_total"], lv["blocks_whole"] + lv["sub_blocks"] / 4
            assert tot == len(src) * 61 * 11 * 11
            assert 0 < ev < 0.1 * tot, "branch and bound evaluated %d of %d blocks" % (ev, tot)
            # the second level: most candidate blocks are settled by their four sub-block bounds
            assert lv["candidates_refined"] > 0 and lv["sub_blocks"] < 2 * lv["candidates_refined"]
    finally:
        os.environ.pop("NHIP_BNB_STATS", None)
        os.environ.pop("NHIP_BNB_INSTRUMENT", None)


def test_cell_width_against_unquantised_table(gpu):
    """How far are the 8- and 16-bit tables from an ideal table of double log-likelihoods (the in-tree evidence for
    the reference's table is CImg<double>, cimg_debug.h:19)?  On config #2 pairs: 16-bit cells keep every reported
    score within 1e-5 relative of the unquantised score at the same pose and find the same best pose; 8-bit cells do
    not meet 1e-5 (documented in DESIGN.md section 3 with these numbers)."""
    bag = synth.SynthBag(300, dense=True)
    src, tgt, th0 = bag.sample_pairs(per_target=4, targets=[40, 100, 160, 220, 280], max_dist=3.5, min_sep=20)
    xy, off = csm.pack_scans(bag.scans)
    ids = np.unique(tgt)
    slot = np.searchsorted(ids, tgt)
    search = csm.search_spec(61, 81, 81, DEG)
    oss = O.search_spec(61, 81, 81, DEG)
    st = csm.ScanTable(xy, off)
    report = {}
    for bits in (8, 16):
        spec, ospec = _specs(cell_bits=bits)
        grids = csm.LikelihoodGrids(st, ids, spec)
        got, _ = csm.match_pairs(st, grids, src, slot, th0, search)
        grids.close()
        probe = np.stack([got["itheta"], got["ix"], got["iy"]], axis=1)
        ideal, at_probe = O.csm_match_f64_batch(xy, off, src, tgt, th0, ospec, oss, probe=probe)
        rel = np.abs((got["score"].astype(np.float64) - at_probe) / at_probe)
        same = (ideal["itheta"] == got["itheta"]) & (ideal["ix"] == got["ix"]) & (ideal["iy"] == got["iy"])
        # where the quantised argmax differs, the unquantised scores of the two poses are a near-tie
        gap = np.abs((ideal["score"] - at_probe) / ideal["score"])
        report[bits] = (float(rel.max()), float(same.mean()), float(gap.max()))
    st.close()
    print("cell width vs unquantised table: (max rel score dev, argmax agreement, max rel gap at disagreement)", report)
    assert report[16][0] < 1e-5 and report[16][1] == 1.0
    assert report[16][0] < 1e-5           # measured 4e-6 (float32 record: 6e-8)
    assert report[8][0] < 1e-3 and report[8][2] < 1e-3   # 8-bit: ~1e-4, near-ties only
    assert report[8][0] > report[16][0]


@pytest.mark.parametrize("cell_bits", [16, 8])
def test_config4_per_gpu_share(gpu, cell_bits):
    """BASELINE configs[3] (10k scans, 1M pairs over 8 GPUs) at one GPU's share: 1250 scans, 125,000 candidate pairs
    (100 per target) through bench.py's own sharded step at world size 1.  Checks: the branch-and-bound matcher and
    the kernel that performs every add return the same 125,000 records and sums; an oracle sample is bit-exact;
    properties that hold at any size: pairs of the same (source, target, theta0) get the same record, a pair matched
    against itself peaks at the lattice centre, scores are the formula of their sums."""
    import os
    import sys
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from nautilus_amd import sharding
    wl = bench.Workload("config4", 8, scans=1250, per_target=100)
    assert wl.n_pairs == 125000 and wl.n_scans == 1250
    # self pairs and duplicates replace the first pairs of the list (same targets, so the partition is unchanged)
    wl.src[:50] = wl.tgt[:50]
    wl.th0[:50] = 0.0
    wl.src[50:100], wl.th0[50:100] = wl.src[100:150], wl.th0[100:150]
    wl.tgt[50:100] = wl.tgt[100:150]
    plan = sharding.ShardPlan(wl.src, wl.tgt, wl.th0, 1)
    dev = torch.device("cuda", 0)
    m = bench.HipMatcher(wl, plan.shard(0), dev, cell_bits)
    elapsed, full = bench.run_sharded(plan, 0, 1, dev, m, steps=1, warmup=0)
    rec = full.cpu().numpy().view(csm.MATCH_DTYPE).reshape(-1)
    sums = np.empty(wl.n_pairs, np.int32)
    sums[plan.order] = m.records()[1].cpu().numpy()
    m.free_grids()
    mx = bench.HipMatcher(wl, plan.shard(0), dev, cell_bits, exhaustive=True)
    _, full_x = bench.run_sharded(plan, 0, 1, dev, mx, steps=1, warmup=0)
    assert torch.equal(full, full_x), "branch and bound differs from the exhaustive kernel"
    mx.free_grids()
    os.environ["NHIP_BNB_KERNELS"] = "2"  # candidates of all pairs through the grid-wide lists (default: small batches only)
    try:
        m1 = bench.HipMatcher(wl, plan.shard(0), dev, cell_bits)
        _, full_1 = bench.run_sharded(plan, 0, 1, dev, m1, steps=1, warmup=0)
        assert torch.equal(full, full_1), "one-kernel and two-kernel forms differ"
        m1.free_grids()
    finally:
        os.environ.pop("NHIP_BNB_KERNELS", None)
    # self pairs: rotation 0 (k = 30), no shift (ix = iy = 40)
    assert np.all(rec["itheta"][:50] == 30) and np.all(rec["ix"][:50] == 40) and np.all(rec["iy"][:50] == 40)
    assert rec[50:100].tobytes() == rec[100:150].tobytes()
    Lf, step = math.log(1e-10), -math.log(1e-10) / (255.0 if cell_bits == 8 else 65535.0)
    assert np.array_equal(rec["score"], (Lf + step * sums.astype(np.float64) / 1081.0).astype(np.float32))
    assert np.all((rec["itheta"] >= 0) & (rec["itheta"] < 61) & (rec["ix"] >= 0) & (rec["ix"] < 81) & (rec["iy"] < 81))
    # oracle sample
    sel = np.r_[0:4, 60:64, np.random.default_rng(0).choice(wl.n_pairs, 40, replace=False)]
    ospec, oss = O.grid_spec(cell_bits=cell_bits), O.search_spec(61, 81, 81, DEG)
    ids = np.unique(wl.tgt[sel])
    og = O.grid_build_batch(wl.xy, wl.off, ids, ospec)
    want = O.csm_match_batch(wl.xy, wl.off, og, ospec, wl.src[sel], np.searchsorted(ids, wl.tgt[sel]), wl.th0[sel], oss)
    for f in ("itheta", "ix", "iy"):
        assert np.array_equal(rec[f][sel], want[f])
    assert np.array_equal(sums[sel], want["sum"])


def test_config2_full_size_16bit_branch_and_bound_equals_every_add(gpu):
    """BASELINE configs[1] at full size and at the cell width of the headline (16 bits): 1,000 dense 1081-beam scans,
    10,000 candidate pairs, 61 x 81 x 81 lattice, 1200 x 1200 tables -- the branch-and-bound matcher (the product path
    bench.py times) and csm_correlate16_kernel (every add of the exhaustive definition) return the same 10,000 records
    and integer sums, byte for byte; a sample agrees with the oracle."""
    import os
    import sys
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from nautilus_amd import sharding
    wl = bench.Workload("weak", 1)
    assert wl.n_pairs == 10000 and wl.n_scans == 1000
    # (as bench.py runs it: the pairs launched heaviest first by the cost estimate from the odometry poses; the kernel
    #  that performs every add takes them in by-target order -- the records must agree in shard order)
    w = sharding.predicted_pair_cost(wl.bag.odom, wl.src, wl.tgt)
    plan = sharding.ShardPlan(wl.src, wl.tgt, wl.th0, 1, w)
    dev = torch.device("cuda", 0)
    m = bench.HipMatcher(wl, plan.shard(0), dev, 16, weights=plan.shard_weights(0))
    assert m.d_unperm is not None and not np.array_equal(m.src, plan.shard(0)[1])
    _, full = bench.run_sharded(plan, 0, 1, dev, m, steps=1, warmup=0)
    sums = m.records()[1].clone()
    m.free_grids()
    mx = bench.HipMatcher(wl, plan.shard(0), dev, 16, exhaustive=True)
    _, full_x = bench.run_sharded(plan, 0, 1, dev, mx, steps=1, warmup=0)
    assert torch.equal(full, full_x), "branch and bound differs from the kernel that performs every add (16-bit cells)"
    assert torch.equal(sums, mx.records()[1])
    mx.free_grids()
    # 10,000 pairs take the split form (bounds + seeds; the pairs ordered by candidates left; the candidates); the fused
    # form (one workgroup per pair from start to end), and the split form in rounds of 3,000 pairs with the candidates on
    # the helper stream, return the same records
    for env in ({"NHIP_BNB_SPLIT": "0"}, {"NHIP_BNB_SPLIT_BATCH": "3000", "NHIP_BNB_SPLIT_MIN": "100"}):
        os.environ.update(env)
        try:
            mf = bench.HipMatcher(wl, plan.shard(0), dev, 16, weights=plan.shard_weights(0))
            _, full_f = bench.run_sharded(plan, 0, 1, dev, mf, steps=1, warmup=0)
            assert torch.equal(full, full_f) and torch.equal(sums, mf.records()[1]), env
            mf.free_grids()
        finally:
            for k_ in env:
                os.environ.pop(k_, None)
    rec = full.cpu().numpy().view(csm.MATCH_DTYPE).reshape(-1)
    hsums = np.empty(wl.n_pairs, np.int32)
    hsums[plan.order] = sums.cpu().numpy()
    sel = np.random.default_rng(1).choice(wl.n_pairs, 24, replace=False)
    ospec, oss = O.grid_spec(cell_bits=16), O.search_spec(61, 81, 81, DEG)
    ids = np.unique(wl.tgt[sel])
    og = O.grid_build_batch(wl.xy, wl.off, ids, ospec)
    want = O.csm_match_batch(wl.xy, wl.off, og, ospec, wl.src[sel], np.searchsorted(ids, wl.tgt[sel]), wl.th0[sel], oss)
    for f in ("itheta", "ix", "iy"):
        assert np.array_equal(rec[f][sel], want[f])
    assert np.array_equal(hsums[sel], want["sum"])
    assert np.array_equal(rec["score"][sel], want["score"].astype(np.float32))


def test_drop_in_call_with_rotation_restriction_pi(gpu, small_bag):
    """/root/reference/src/optimization/solver.cc:633-638 passes DegToRad(90); the class accepts any restriction.  At
    the drop-in's default cell width a restriction of pi (361 coarse rotations: beyond what the branch-and-bound
    matcher's bounds fit in LDS) and a constructor with trans_range / low_res > 44 (97 x 97 coarse translations) go
    through csm_correlate16_kernel and agree with the oracle's two-level restatement, float for float."""
    a, b = small_bag.scans[17][::3], small_bag.scans[15][::3]
    rot_a, rot_b = small_bag.odom[17, 2], small_bag.odom[15, 2]
    m = csm.CorrelativeScanMatcher(30, 2, 0.3, 0.01)
    got = m.GetTransformation(a, b, rot_a, rot_b, math.pi)
    want = O.two_level_match(a, b, rot_a, rot_b, math.pi, 30.0, 2.0, 0.3, 0.01, cell_bits=16)
    assert _same_call(got, want)
    m2 = csm.CorrelativeScanMatcher(30, 4.8, 0.1, 0.05)   # coarse level: +-48 cells
    got = m2.GetTransformation(a, b, rot_a, rot_b, math.radians(30))
    want = O.two_level_match(a, b, rot_a, rot_b, math.radians(30), 30.0, 4.8, 0.1, 0.05, cell_bits=16)
    assert _same_call(got, want)


@pytest.mark.parametrize("cell_bits,geometry,skip_map", [(16, (30.0, 0.05, 2.0, 40), True), (8, (30.0, 0.05, 2.0, 40), True),
                                                         (16, (10.0, 0.03, 1.0, 10), True), (16, (12.0, 0.05, 2.0, 6), True),
                                                         (16, (30.0, 0.05, 2.0, 40), False), (16, (10.0, 0.03, 1.0, 10), False)])
def test_grid_rebuild_clears_what_the_last_build_wrote(gpu, small_bag, cell_bits, geometry, skip_map):
    """nhip_grid_rebuild_dev clears only the tiles the previous build listed in the workspace -- and must leave the
    buffer exactly as a build into zeroed memory would: other targets than before (their tiles lie elsewhere), a
    workspace whose header holds garbage, and a buffer the workspace has never seen (full of 0xFF) all give the
    slots of a fresh build, byte for byte (image, skip map, pooled tables, plane of high bytes)."""
    import torch
    dev = torch.device("cuda:0")
    lib = _lib.load()
    # (the second and third geometry have borders of 36 and 28 cells: their tiles do not start on the 16-byte
    #  boundaries of the tiled planes, and the clearing kernel takes its dword path)
    range_m, res, sigma, max_shift = geometry
    # (without a skip map the rebuild leaves the map's space and the first-level table's out of its clearing)
    spec = csm.grid_spec(range_m, res, sigma, 1e-10, max_shift, cell_bits, skip_map=skip_map)
    L = csm.grid_layout(spec)
    xy, off = csm.pack_scans(small_bag.scans)
    t = lambda a: torch.from_numpy(a).to(dev)
    d_xy, d_off = t(xy), t(off)
    n = 3
    nbytes = lib.nhip_grids_bytes(C.byref(spec), n)
    ws_bytes = lib.nhip_grid_workspace_bytes(C.byref(spec), n)
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def call(fn, ids, d_grids, d_ws):
        d_ids = t(np.asarray(ids, dtype=np.int32))
        _lib.check(fn(d_xy.data_ptr(), d_off.data_ptr(), d_off.numel() - 1, d_ids.data_ptr(), n, C.byref(spec), d_grids.data_ptr(),
                      d_ws.data_ptr(), ws_bytes, sp))
        torch.cuda.synchronize()
        return d_grids[:n * L.slot_bytes].cpu().numpy().copy()

    ids_a, ids_b = [3, 17, 40], [25, 8, 3]
    fresh_b = call(lib.nhip_grid_build_dev, ids_b, torch.zeros(nbytes, dtype=torch.uint8, device=dev),
                   torch.zeros(ws_bytes, dtype=torch.uint8, device=dev))
    assert fresh_b.any()
    G, W = torch.empty(nbytes, dtype=torch.uint8, device=dev), torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    a = call(lib.nhip_grid_build_dev, ids_a, G, W)
    assert not np.array_equal(a, fresh_b)
    assert np.array_equal(call(lib.nhip_grid_rebuild_dev, ids_b, G, W), fresh_b), "rebuild over another set of targets"
    assert np.array_equal(call(lib.nhip_grid_rebuild_dev, ids_b, G, W), fresh_b), "rebuild over the same targets"
    W2 = torch.randint(0, 256, (ws_bytes,), dtype=torch.uint8, device=dev)
    assert np.array_equal(call(lib.nhip_grid_rebuild_dev, ids_b, G, W2), fresh_b), "garbage workspace header"
    G3 = torch.full((nbytes,), 255, dtype=torch.uint8, device=dev)
    assert np.array_equal(call(lib.nhip_grid_rebuild_dev, ids_b, G3, W), fresh_b), "a buffer the workspace never saw"


@pytest.mark.parametrize("cell_bits", [16, 8])
def test_pooled_tables_from_the_tile_list_equal_the_band_kernels(gpu, small_bag, cell_bits):
    """Both pooled tables are built per listed 64 x 64 tile and cleared around the previous build's tiles
    (grid_pool4_tiles_kernel, grid_pool8_tiles_kernel); NHIP_GRID_POOL=bands keeps the kernels that walk every band of
    every slot.  Whole slots equal, byte for byte: a first build, a rebuild over other targets, a rebuild over the same."""
    import os
    import torch
    dev = torch.device("cuda:0")
    lib = _lib.load()
    spec = csm.grid_spec(30.0, 0.05, 2.0, 1e-10, 40, cell_bits)
    L = csm.grid_layout(spec)
    xy, off = csm.pack_scans(small_bag.scans)
    t = lambda a: torch.from_numpy(a).to(dev)
    d_xy, d_off = t(xy), t(off)
    n = 4
    nbytes = lib.nhip_grids_bytes(C.byref(spec), n)
    ws_bytes = lib.nhip_grid_workspace_bytes(C.byref(spec), n)
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run(form):
        if form:
            os.environ["NHIP_GRID_POOL"] = form
        try:
            G = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            W = torch.zeros(ws_bytes, dtype=torch.uint8, device=dev)
            out = []
            for fn, ids in ((lib.nhip_grid_build_dev, [3, 17, 40, 5]), (lib.nhip_grid_rebuild_dev, [25, 8, 3, 44]),
                            (lib.nhip_grid_rebuild_dev, [25, 8, 3, 44])):
                d_ids = t(np.asarray(ids, dtype=np.int32))
                _lib.check(fn(d_xy.data_ptr(), d_off.data_ptr(), d_off.numel() - 1, d_ids.data_ptr(), n, C.byref(spec), G.data_ptr(),
                              W.data_ptr(), ws_bytes, sp))
                torch.cuda.synchronize()
                out.append(G[:n * L.slot_bytes].cpu().numpy().copy())
            return out
        finally:
            os.environ.pop("NHIP_GRID_POOL", None)

    tiles, bands = run(None), run("bands")
    for a, b, what in zip(tiles, bands, ("first build", "rebuild over other targets", "rebuild over the same targets")):
        assert a.any() and np.array_equal(a, b), what
    assert np.array_equal(tiles[1], tiles[2])


def test_environment_switches_need_nhip_tunables(gpu):
    """A shipped process reads no behaviour switch: NHIP_BNB_SPLIT=0 changes the form of a 300-pair list only in a
    process started with NHIP_TUNABLES=1 (the library looks once, at its first call)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import numpy as np\n"
            "from nautilus_amd import csm, synth\n"
            "bag = synth.SynthBag(40, dense=True)\n"
            "st = csm.ScanTable.from_list(bag.scans)\n"
            "ids = np.arange(30, dtype=np.int32)\n"
            "grids = csm.LikelihoodGrids(st, ids, csm.grid_spec(max_shift=10))\n"
            "src, tgt, th0 = bag.sample_pairs(per_target=10, targets=ids, max_dist=3.5, min_sep=2)\n"
            "got, sums = csm.match_pairs(st, grids, src, np.searchsorted(ids, tgt), th0, csm.search_spec(9, 21, 21))\n"
            "import zlib; print('FORM', csm.last_launch()['form_id'], len(src), zlib.crc32(got.tobytes()))\n" % root)
    out = {}
    for tun in ("0", "1"):
        env = dict(os.environ, NHIP_BNB_SPLIT="0")
        env.pop("NHIP_TUNABLES", None)
        if tun == "1":
            env["NHIP_TUNABLES"] = "1"
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stdout + p.stderr
        out[tun] = [l for l in p.stdout.splitlines() if l.startswith("FORM")][0].split()
    assert out["0"][2] == out["1"][2] == "300"
    assert out["0"][1] == "1", "without NHIP_TUNABLES the switch must be ignored: 300 pairs take the split form"
    assert out["1"][1] == "0", "with NHIP_TUNABLES=1 NHIP_BNB_SPLIT=0 selects the one-kernel form"
    assert out["0"][3] == out["1"][3], "both forms return the same records"


@pytest.mark.parametrize("cell_bits", [16, 8])
def test_ids_in_device_memory_out_of_range_cost_an_error_not_the_process(gpu, small_bag, cell_bits):
    """The ids a "_dev" entry point reads from device memory are checked by the kernels against the counts passed beside
    them (include/nautilus_hip.h, "Ids in device memory"; the reference CHECKs such input, slam_residuals.h:99-101,109).
    Round 4 lost a process to exactly the ids below: scan 60 of a 48-scan bag handed to nhip_grid_rebuild_dev
    (gpurun_out/diag_csm.log).  Now: the bad target's grid is all floor, the others are what a build of
    the good ids alone gives, nhip_dev_status() returns NHIP_ERR_ARG naming the id, once; a pair list with one bad source
    and one bad slot leaves those two records at (0, 0, 0, floor score) and every other record untouched -- in every form
    of the matcher and in the kernels that perform every add."""
    import os
    import torch
    dev = torch.device("cuda:0")
    lib = _lib.load()
    spec = csm.grid_spec(30.0, 0.05, 2.0, 1e-10, 40, cell_bits)
    L = csm.grid_layout(spec)
    xy, off = csm.pack_scans(small_bag.scans)
    n_scans = len(off) - 1
    assert n_scans == 48
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_xy, d_off = t(xy), t(off)
    n = 4
    nbytes = lib.nhip_grids_bytes(C.byref(spec), n)
    ws_bytes = lib.nhip_grid_workspace_bytes(C.byref(spec), n)
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    info = (C.c_int32 * 4)()
    assert lib.nhip_dev_status(sp, info) == _lib.NHIP_OK  # nothing pending

    def build(fn, ids, G, W):
        d_ids = t(np.asarray(ids, dtype=np.int32))
        _lib.check(fn(d_xy.data_ptr(), d_off.data_ptr(), n_scans, d_ids.data_ptr(), n, C.byref(spec), G.data_ptr(), W.data_ptr(),
                      ws_bytes, sp))
        return lib.nhip_dev_status(sp, info)

    G = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    W = torch.zeros(ws_bytes, dtype=torch.uint8, device=dev)
    assert build(lib.nhip_grid_build_dev, [3, 17, 40, 5], G, W) == _lib.NHIP_OK
    # the ids of the failing log, through the rebuild (the call that aborted)
    rc = build(lib.nhip_grid_rebuild_dev, [25, 8, 3, 60], G, W)
    assert rc == _lib.NHIP_ERR_ARG and list(info) == [1, 1, 60, 3], (rc, list(info))
    msg = lib.nhip_last_error().decode()
    assert "d_target_ids" in msg and "60" in msg, msg
    assert lib.nhip_dev_status(sp, info) == _lib.NHIP_OK and list(info) == [0, 0, 0, 0]  # reported once
    got = G[:n * L.slot_bytes].cpu().numpy().reshape(n, L.slot_bytes)
    G2 = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    assert build(lib.nhip_grid_build_dev, [25, 8, 3, 3], G2, torch.zeros(ws_bytes, dtype=torch.uint8, device=dev)) == _lib.NHIP_OK
    want = G2[:n * L.slot_bytes].cpu().numpy().reshape(n, L.slot_bytes)
    assert np.array_equal(got[:3], want[:3]) and not got[3].any(), "good targets as built alone, the bad one all floor"
    # negative and huge ids, first build
    rc = build(lib.nhip_grid_build_dev, [-1, 8, 2 ** 31 - 1, 3], G, W)
    assert rc == _lib.NHIP_ERR_ARG and info[0] == 1 and info[2] in (-1, 2 ** 31 - 1)
    assert build(lib.nhip_grid_build_dev, [25, 8, 3, 44], G, W) == _lib.NHIP_OK

    # ---- pair lists
    search = csm.search_spec(61, 81, 81, DEG)
    rng = np.random.default_rng(5)
    n_pairs = 40
    src = rng.integers(0, n_scans, n_pairs).astype(np.int32)
    slot = rng.integers(0, n, n_pairs).astype(np.int32)
    th0 = rng.uniform(-0.2, 0.2, n_pairs)
    rot0 = np.empty((n_pairs, 2))
    _lib.check(lib.nhip_csm_rot0(_lib.ptr(th0), None, n_pairs, _lib.ptr(rot0)))
    d_rot0 = t(rot0)
    d_keys = torch.empty(n_pairs, dtype=torch.int64, device=dev)
    d_out = torch.empty((n_pairs, 4), dtype=torch.int32, device=dev)
    d_sums = torch.empty(n_pairs, dtype=torch.int32, device=dev)
    ws = lib.nhip_csm_workspace_bytes(n_pairs)
    d_ws = torch.empty(ws, dtype=torch.uint8, device=dev)

    def match(src_, slot_, srch):
        d_src, d_slot = t(src_), t(slot_)
        d_delta = t(csm.delta_table(srch))
        d_out.fill_(-7)
        _lib.check(lib.nhip_csm_match_dev(d_xy.data_ptr(), d_off.data_ptr(), n_scans, G.data_ptr(), n, C.byref(spec),
                                          d_src.data_ptr(), d_slot.data_ptr(), d_rot0.data_ptr(), d_delta.data_ptr(), None,
                                          n_pairs, C.byref(srch), d_keys.data_ptr(), d_out.data_ptr(), d_sums.data_ptr(),
                                          d_ws.data_ptr(), ws, sp))
        rc_ = lib.nhip_dev_status(sp, info)
        return rc_, d_out.cpu().numpy().copy(), d_sums.cpu().numpy().copy()

    rc, good, good_sums = match(src, slot, search)
    assert rc == _lib.NHIP_OK
    bad_src, bad_slot = src.copy(), slot.copy()
    bad_src[7], bad_slot[19] = 60, n + 2
    floor = np.float32(math.log(1e-10))
    ex = csm.search_spec(61, 81, 81, DEG, exhaustive=True)
    small = csm.search_spec(9, 13, 13, DEG, exhaustive=True)
    forms = [({}, search), ({"NHIP_BNB_KERNELS": "1"}, search), ({"NHIP_BNB_KERNELS": "1", "NHIP_BNB_QUEUE": "1"}, search),
             ({"NHIP_BNB_KERNELS": "1", "NHIP_BNB_SPLIT": "1"}, search),
             ({"NHIP_BNB_KERNELS": "1", "NHIP_BNB_SPLIT": "1", "NHIP_BNB_SPLIT_BATCH": "3"}, search), ({}, ex), ({}, small)]
    for env, srch in forms:
        os.environ.update(env)
        try:
            rc0, ref, ref_sums = match(src, slot, srch)
            rc, rec, sums = match(bad_src, bad_slot, srch)
        finally:
            for k_ in env:
                os.environ.pop(k_, None)
        assert rc0 == _lib.NHIP_OK
        if srch is search or srch is ex:
            assert np.array_equal(ref, good) and np.array_equal(ref_sums, good_sums), env
        assert rc == _lib.NHIP_ERR_ARG and info[0] == (2 | 4), (env, rc, list(info))
        assert (info[1], info[2], info[3]) in ((2, 60, 7), (4, n + 2, 19)), list(info)
        keep = np.ones(n_pairs, bool)
        keep[[7, 19]] = False
        assert np.array_equal(rec[keep], ref[keep]) and np.array_equal(sums[keep], ref_sums[keep]), env
        for i in (7, 19):
            assert list(rec[i, :3]) == [0, 0, 0] and rec[i, 3:4].view(np.float32)[0] == floor and sums[i] == 0, (env, rec[i])
    assert lib.nhip_dev_status(sp, info) == _lib.NHIP_OK
    # the score volume's scan and slot are host arguments: checked on the host
    d_vol = torch.empty(61 * 81 * 81, dtype=torch.int32, device=dev)
    d_delta = t(csm.delta_table(search))
    rc = lib.nhip_csm_scores_dev(d_xy.data_ptr(), d_off.data_ptr(), n_scans, G.data_ptr(), n, C.byref(spec), 60, 0,
                                 d_rot0.data_ptr(), d_delta.data_ptr(), 0, 0, C.byref(search), d_vol.data_ptr(), sp)
    assert rc == _lib.NHIP_ERR_ARG
    torch.cuda.synchronize()


@pytest.mark.parametrize("n_pairs", [192, 200, 300, 400, 487, 600])
def test_lists_from_192_pairs_take_the_split_form_at_the_production_lattice(gpu, small_bag, n_pairs):
    """include/nautilus_hip.h: "Lists of 192 pairs and more run as two kernels".  Round 4's launcher tested for the state of
    512 pairs before it sized the rounds, and nhip_csm_workspace_bytes(n) of 192 .. 487 pairs at 61 rotations is less than
    that: those lists ran as one kernel per pair with hand-over lists (same records, slower; the only form assertion
    used 9 rotations, where the threshold was 2.4 MB).  At the production lattice, through the handle API (whose
    workspace is exactly nhip_csm_workspace_bytes(n)): split form, one round, no hand-over kernel; one pair fewer than
    192: one kernel per pair."""
    spec = csm.grid_spec(30.0, 0.05, 2.0, 1e-10, 40, 16)
    search = csm.search_spec(61, 81, 81, DEG)
    st = csm.ScanTable.from_list(small_bag.scans)
    ids = np.arange(0, 48, 4, dtype=np.int32)
    grids = csm.LikelihoodGrids(st, ids, spec)
    rng = np.random.default_rng(n_pairs)
    src = rng.integers(0, 48, n_pairs).astype(np.int32)
    slot = rng.integers(0, len(ids), n_pairs).astype(np.int32)
    th0 = rng.uniform(-0.1, 0.1, n_pairs)
    got, sums = csm.match_pairs(st, grids, src, slot, th0, search)
    info = csm.last_launch()
    assert info["form_id"] == 1 and info["pairs_per_round"] == n_pairs and info["rounds"] == 1 and not info["hand_over_kernel"], info
    got1, sums1 = csm.match_pairs(st, grids, src[:191], slot[:191], th0[:191], search)
    info1 = csm.last_launch()
    assert info1["form_id"] == 0 and info1["hand_over_kernel"], info1
    assert got1.tobytes() == got[:191].tobytes() and np.array_equal(sums1, sums[:191])
    grids.close()
    st.close()


def test_parity_against_the_double_table_at_scale(gpu):
    """north_star: best-pose indices bit-exact and scores within 1e-5 relative of the CPU reference, whose table is a
    CImg<double> (/root/reference/src/visualization/cimg_debug.h:19).  The kernels are bit-exact against the QUANTISED oracle;
    this test puts a number on the distance to the reference's table type at scale: 1,000 pairs of configs[1] and 300 pairs
    in the style of configs[3] (100 per target, sources up to 3.5 m away), 16-bit cells, against the exhaustive search on an
    unquantised double table (bench.parity_vs_f64, the leg the bench line's `parity_vs_f64` comes from).  Asserted: every
    score reported with NHIP_SEARCH_EXACT_SCORE within 2e-7 relative of the double table's score at the same pose (north_star
    asks for 1e-5); wherever the quantised argmax is
    another pose than the double table's, the two poses' double-table scores differ by less than one quantisation step
    (3.5e-4 nat: the bound that holds by construction) -- and by how much less is printed and recorded in DESIGN.md section 3."""
    import bench
    import json
    wl = bench.Workload("weak", 1)
    r = bench.parity_vs_f64(wl, 1000, 300, 16, n_threads=bench._omp_threads())
    print("parity vs double table:", json.dumps(r))
    assert r["configs[1]"]["pairs"] >= 1000 and r["configs[3]-style"]["pairs"] >= 300
    # scores with NHIP_SEARCH_EXACT_SCORE: the double table's score at the winning pose (the record is a float: 6e-8).
    # The quantised formula Lf + step * sum / N is within 2.3e-5 at the worst of these pairs (median 9e-7): the 20-pair
    # test's 4e-6 was luck of the sample, and 1e-5 is NOT met by it at scale -- which is why the flag exists.
    assert r["max_rel_score"] < 2e-7
    assert 1e-6 < r["max_rel_score_quantised_formula"] < 5e-5
    assert r["max_gap_nat"] <= r["guaranteed_max_gap_nat"]
    for name in ("configs[1]", "configs[3]-style"):
        assert r[name]["index_agreement"] >= 0.97, (name, r[name])


@pytest.mark.parametrize("cell_bits", [16, 8])
def test_exact_score_is_the_double_tables_score_at_the_winning_pose(gpu, small_bag, cell_bits):
    """NHIP_SEARCH_EXACT_SCORE: indices and integer sums are those of the quantised search, bit for bit; the score is the
    mean log-likelihood of the winning pose on the UNQUANTISED table -- the reference's table type, CImg<double>
    (/root/reference/src/visualization/cimg_debug.h:19) -- recomputed from the hit raster: equal to the oracle's double
    table at that pose (orc_csm_match_f64's probe) to 2e-7 relative (the record is a float: 6e-8; the device's log and the
    order of its double sum differ from the host's in the last bits).  Tolerance of north_star: 1e-5.  Also: the hit
    raster equals the oracle's rasterisation; every kernel family reports the same exact score; a shifted search centre
    reports the same score for the same pose; points outside the grid and non-finite points contribute the floor."""
    spec, ospec = _specs(cell_bits=cell_bits)
    scans = [s.copy() for s in small_bag.scans]
    scans[5] = np.concatenate([scans[5], np.array([[40.0, 1.0], [np.nan, 0.0], [3e10, -2.0], [-31.0, 29.99]], np.float32)])
    xy, off = csm.pack_scans(scans)
    st = csm.ScanTable(xy, off)
    ids = np.array([2, 11, 20, 29, 38, 45], dtype=np.int32)
    grids = csm.LikelihoodGrids(st, ids, spec)
    for slot, sid in enumerate(ids[:3]):
        got = grids.hits(slot)
        S = grids.layout.side
        cells = np.zeros((S, S), np.uint8)
        p = scans[sid]
        c = S // 2 + np.floor(p[:, 0].astype(np.float64) / 0.05).astype(np.int64)
        r = S // 2 + np.floor(p[:, 1].astype(np.float64) / 0.05).astype(np.int64)
        ok = (c >= 0) & (c < S) & (r >= 0) & (r < S) & np.isfinite(p[:, 0]) & np.isfinite(p[:, 1])
        cells[r[ok], c[ok]] = 1
        assert np.array_equal(got, cells), "hit raster"
    src, tgt, th0 = small_bag.sample_pairs(per_target=8, targets=ids, max_dist=3.5, min_sep=2)
    src[0], src[9] = 5, 5  # (the scan with points outside the grid and non-finite points)
    slot = np.searchsorted(ids, tgt)
    oss = O.search_spec(61, 81, 81, DEG)
    plain, sums = csm.match_pairs(st, grids, src, slot, th0, csm.search_spec(61, 81, 81, DEG))
    exact, sums_e = csm.match_pairs(st, grids, src, slot, th0, csm.search_spec(61, 81, 81, DEG, exact_score=True))
    for f in ("itheta", "ix", "iy"):
        assert np.array_equal(plain[f], exact[f])
    assert np.array_equal(sums, sums_e)
    probe = np.stack([exact["itheta"], exact["ix"], exact["iy"]], axis=1)
    ideal, at_probe = O.csm_match_f64_batch(xy, off, src, tgt, th0, ospec, oss, probe=probe)
    rel = np.abs((exact["score"].astype(np.float64) - at_probe) / at_probe)
    rel_plain = np.abs((plain["score"].astype(np.float64) - at_probe) / at_probe)
    print("exact score vs double table: max rel %.3g (quantised formula: %.3g)" % (rel.max(), rel_plain.max()))
    assert rel.max() < 2e-7, rel.max()
    assert rel_plain.max() > rel.max()
    # the kernels that perform every add report the same records with the flag
    ex, _ = csm.match_pairs(st, grids, src, slot, th0, csm.search_spec(61, 81, 81, DEG, exhaustive=True, exact_score=True))
    assert ex.tobytes() == exact.tobytes()
    # a shifted search centre: the same pose, the same score
    org = np.tile(np.array([[3, -2]], np.int32), (len(src), 1))
    sh, _ = csm.match_pairs(st, grids, src, slot, th0, csm.search_spec(61, 69, 69, DEG, exact_score=True), org)
    same = (sh["itheta"] == exact["itheta"]) & (sh["ix"] + 3 - 34 == exact["ix"] - 40) & (sh["iy"] - 2 - 34 == exact["iy"] - 40)
    assert same.sum() >= len(src) // 2, "most optima lie inside the smaller window too"
    assert np.array_equal(sh["score"][same], exact["score"][same])
    # small lattices (the kernel whose lanes are poses), exhaustive and branch and bound
    small = csm.search_spec(9, 13, 13, DEG, exact_score=True)
    a, _ = csm.match_pairs(st, grids, src, slot, th0, small)
    b, _ = csm.match_pairs(st, grids, src, slot, th0, csm.search_spec(9, 13, 13, DEG, exhaustive=True, exact_score=True))
    assert a.tobytes() == b.tobytes()
    _, at_small = O.csm_match_f64_batch(xy, off, src, tgt, th0, ospec, O.search_spec(9, 13, 13, DEG),
                                        probe=np.stack([a["itheta"], a["ix"], a["iy"]], axis=1))
    assert np.abs((a["score"].astype(np.float64) - at_small) / at_small).max() < 2e-7
    grids.close()
    st.close()


@pytest.mark.parametrize("cell_bits", [16, 8])
def test_grids_without_the_row_major_image(gpu, small_bag, cell_bits):
    """NHIP_GRID_NO_IMAGE: the slots hold what the branch-and-bound matcher reads -- pooled tables, tiled planes, hit raster
    -- and no row-major image: a third smaller.  Everything derived is byte for byte what a build WITH the image derives
    (the pooled tables then come from the tiled copy of the cells); the matcher returns the same records in every form
    that takes such slots, plain and exact scores; a rebuild equals a fresh build; what needs the image is refused with
    an error code: the kernels that perform every add, score volumes, image downloads, lists with a scan of more than
    1088 points, the skip-map flag."""
    import os
    import torch
    lib = _lib.load()
    full_spec = csm.grid_spec(30.0, 0.05, 2.0, 1e-10, 40, cell_bits)
    spec = csm.grid_spec(30.0, 0.05, 2.0, 1e-10, 40, cell_bits, no_image=True)
    Lf_, L = csm.grid_layout(full_spec), csm.grid_layout(spec)
    # (up to the padding that keeps every slot's tiled planes on cache-line boundaries)
    assert L.grid_bytes == 0 and L.skip_bytes == 0 and abs(L.slot_bytes - (Lf_.slot_bytes - Lf_.grid_bytes - Lf_.skip_bytes)) < 256
    assert L.slot_bytes < 0.70 * Lf_.slot_bytes
    bad = csm.grid_spec(30.0, 0.05, 2.0, 1e-10, 40, 16, skip_map=True, no_image=True)
    assert lib.nhip_grid_layout(C.byref(bad), C.byref(_lib.GridLayout())) == _lib.NHIP_ERR_ARG
    scans = [s.copy() for s in small_bag.scans]
    scans.append(np.concatenate([scans[3], scans[4] + np.float32(0.01)]))  # 2162 points: beyond the by-rotation form
    xy, off = csm.pack_scans(scans)
    st = csm.ScanTable(xy, off)
    ids = np.array([1, 9, 17, 25, 33, 41], dtype=np.int32)
    gf, gl = csm.LikelihoodGrids(st, ids, full_spec), csm.LikelihoodGrids(st, ids, spec)
    for slot in range(len(ids)):
        for lv in (1, 2):
            assert np.array_equal(gf.pooled(slot, lv), gl.pooled(slot, lv)), "pooled table, level %d" % lv
        for cp in (0, 1):
            assert np.array_equal(gf.hi_plane(slot, cp), gl.hi_plane(slot, cp))
        if cell_bits == 16:
            assert np.array_equal(gf.tiled16(slot), gl.tiled16(slot))
        assert np.array_equal(gf.hits(slot), gl.hits(slot))
    with pytest.raises(_lib.NhipError):
        gl.download(0)
    src, tgt, th0 = small_bag.sample_pairs(per_target=7, targets=ids, max_dist=3.5, min_sep=2)
    slot = np.searchsorted(ids, tgt)
    for search in (csm.search_spec(61, 81, 81, DEG), csm.search_spec(61, 81, 81, DEG, exact_score=True), csm.search_spec(9, 21, 21, DEG)):
        want, want_sums = csm.match_pairs(st, gf, src, slot, th0, search)
        for env in ({}, {"NHIP_BNB_KERNELS": "1"}, {"NHIP_BNB_KERNELS": "1", "NHIP_BNB_LEVELS": "1"}, {"NHIP_BNB_KERNELS": "2", "NHIP_BNB_HEAVY_MIN": "1", "NHIP_BNB_KEEP_RANKS": "0"},
                    {"NHIP_BNB_KERNELS": "1", "NHIP_BNB_SPLIT": "1"}, {"NHIP_BNB_KERNELS": "1", "NHIP_BNB_SPLIT": "1", "NHIP_BNB_SPLIT_BATCH": "3", "NHIP_BNB_SPLIT_MIN": "1"}):
            os.environ.update(env)
            try:
                got, sums = csm.match_pairs(st, gl, src, slot, th0, search)
            finally:
                for k_ in env:
                    os.environ.pop(k_, None)
            assert got.tobytes() == want.tobytes() and np.array_equal(sums, want_sums), env
    # refused: every add, a lattice beyond the matcher's envelope, the general kernel (a long scan; the queue form), volumes
    with pytest.raises(_lib.NhipError, match="NHIP_GRID_NO_IMAGE"):
        csm.match_pairs(st, gl, src, slot, th0, csm.search_spec(61, 81, 81, DEG, exhaustive=True))
    with pytest.raises(_lib.NhipError, match="NHIP_GRID_NO_IMAGE"):
        csm.match_pairs(st, gl, src[:2], slot[:2], th0[:2], csm.search_spec(361, 9, 9, DEG))
    with pytest.raises(_lib.NhipError, match="NHIP_GRID_NO_IMAGE"):
        csm.match_pairs(st, gl, [len(scans) - 1], [0], [0.0], csm.search_spec(61, 81, 81, DEG))
    os.environ["NHIP_BNB_QUEUE"] = "1"
    try:
        with pytest.raises(_lib.NhipError, match="NHIP_GRID_NO_IMAGE"):
            csm.match_pairs(st, gl, src, slot, th0, csm.search_spec(61, 81, 81, DEG))
    finally:
        os.environ.pop("NHIP_BNB_QUEUE", None)
    with pytest.raises(_lib.NhipError, match="NHIP_GRID_NO_IMAGE"):
        csm.score_volume(st, gl, 0, 0, 0.0, csm.search_spec(3, 21, 21, DEG))
    gf.close()
    gl.close()
    st.close()
    # rebuild into image-less slots == fresh build, byte for byte (device-pointer API)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_xy, d_off = t(xy), t(off)
    n = 3
    nbytes = lib.nhip_grids_bytes(C.byref(spec), n)
    ws_bytes = lib.nhip_grid_workspace_bytes(C.byref(spec), n)
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def call(fn, ids_, G, W):
        d_ids = t(np.asarray(ids_, dtype=np.int32))
        _lib.check(fn(d_xy.data_ptr(), d_off.data_ptr(), len(off) - 1, d_ids.data_ptr(), n, C.byref(spec), G.data_ptr(), W.data_ptr(), ws_bytes, sp))
        torch.cuda.synchronize()
        return G[:n * L.slot_bytes].cpu().numpy().copy()

    fresh = call(lib.nhip_grid_build_dev, [25, 8, 3], torch.zeros(nbytes, dtype=torch.uint8, device=dev), torch.zeros(ws_bytes, dtype=torch.uint8, device=dev))
    G, W = torch.empty(nbytes, dtype=torch.uint8, device=dev), torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    a = call(lib.nhip_grid_build_dev, [3, 17, 40], G, W)
    assert not np.array_equal(a, fresh)
    assert np.array_equal(call(lib.nhip_grid_rebuild_dev, [25, 8, 3], G, W), fresh), "rebuild over other targets"
    assert np.array_equal(call(lib.nhip_grid_rebuild_dev, [25, 8, 3], G, W), fresh), "rebuild over the same targets"


@pytest.mark.gpu
@pytest.mark.parametrize("cell_bits,no_image", [(16, True), (16, False), (8, False)])
def test_handle_builds_rebuild_into_the_buffers_a_released_handle_left(gpu, small_bag, cell_bits, no_image):
    """nhip_grids_free hands the table buffer and its build workspace back to the device pool TOGETHER, contents known; the
    next nhip_grids_build of the same spec and target count takes the pair and clears what the previous build wrote instead
    of zero-filling every slot (1.3 ms per 1000 targets at 1200 x 1200).  The tables it builds over OTHER targets equal a
    fresh build's, plane by plane; the pair is not taken after anybody else took one of the two buffers (the pool hands
    them to any allocation they fit when it has nothing else), nor after a late skip-map build wrote into the tables, nor with the pool switched off."""
    lib = _lib.load()
    spec = csm.grid_spec(30.0, 0.05, 2.0, 1e-10, 40, cell_bits, no_image=no_image)
    xy, off = csm.pack_scans(small_bag.scans)
    st = csm.ScanTable(xy, off)
    ids_a = np.array([2, 11, 20, 29, 38], dtype=np.int32)
    ids_b = np.array([40, 5, 33, 14, 23], dtype=np.int32)   # other scans in the same slots

    def planes(g):
        out = []
        for slot in range(5):
            out += [g.pooled(slot, 1), g.pooled(slot, 2), g.hi_plane(slot, 0), g.hi_plane(slot, 1), g.hits(slot)]
            if cell_bits == 16:
                out.append(g.tiled16(slot))
            if not no_image:
                out.append(g.download(slot))
        return out

    _lib.check(lib.nhip_device_pool_release())
    _lib.check(lib.nhip_device_pool_configure(0))        # pool off: fresh, zero-filled buffers
    fresh = csm.LikelihoodGrids(st, ids_b, spec)
    assert not fresh.was_rebuilt()
    want = planes(fresh)
    fresh.close()
    _lib.check(lib.nhip_device_pool_configure(32 << 30))
    a = csm.LikelihoodGrids(st, ids_a, spec)
    assert not a.was_rebuilt()
    a.close()
    b = csm.LikelihoodGrids(st, ids_b, spec)
    assert b.was_rebuilt(), "same spec, same count, both buffers still in the pool"
    got = planes(b)
    assert all(np.array_equal(x, y) for x, y in zip(got, want)), "rebuilt tables == fresh tables"
    src, tgt, th0 = small_bag.sample_pairs(per_target=5, targets=np.sort(ids_b), max_dist=3.5, min_sep=2)
    slot = np.array([int(np.nonzero(ids_b == t)[0][0]) for t in tgt], dtype=np.int32)
    search = csm.search_spec(61, 81, 81, DEG, exact_score=True)
    m_b, s_b = csm.match_pairs(st, b, src, slot, th0, search)
    if cell_bits == 16 and not no_image:
        # a search that takes the every-add STRIP kernels builds skip maps late: the handle's tables are no longer what its
        # tile list describes, and its buffers go back with contents unknown  (a plane of at most 256 translations goes to
        # the kernel whose lanes are poses, which reads no map: 11 x 11 builds none -- round 6)
        csm.match_pairs(st, b, src[:3], slot[:3], th0[:3], csm.search_spec(5, 11, 11, DEG, exhaustive=True))
        csm.match_pairs(st, b, src[:3], slot[:3], th0[:3], csm.search_spec(5, 41, 41, DEG, exhaustive=True))
        b.close()
        c = csm.LikelihoodGrids(st, ids_b, spec)
        assert not c.was_rebuilt()
    else:
        b.close()
        # somebody else takes the table buffer in between (any allocation it fits): the workspace alone vouches for nothing
        n_big = int(0.8 * 5 * csm.grid_layout(spec).slot_bytes) // 8
        thief = csm.ScanTable(np.zeros((n_big, 2), dtype=np.float32), np.array([0, n_big], dtype=np.int32))
        c = csm.LikelihoodGrids(st, ids_b, spec)
        assert not c.was_rebuilt()
        thief.close()
    m_c, s_c = csm.match_pairs(st, c, src, slot, th0, search)
    assert m_c.tobytes() == m_b.tobytes() and np.array_equal(s_c, s_b)
    assert all(np.array_equal(x, y) for x, y in zip(planes(c), want))
    c.close()
    st.close()
    _lib.check(lib.nhip_device_pool_configure(4 << 30))   # the library's default cap (per device)
