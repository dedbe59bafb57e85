"""Extended randomized parity sweep of the scan matcher, HIP path vs oracle (not part of the suite):
    gpurun -- python tools/parity_sweep.py 400
Random grid geometry / blur / lattice / search centre / dense, sparse and clustered clouds; grids,
indices and integer sums must be bit-exact, for both cell widths, through the branch-and-bound matcher (lattices up
to 88 x 88) AND the kernel that performs every add (csm_correlate_kernel / csm_correlate16_kernel).  Round 1: 400
configurations, all equal; round 2: 3000 configurations x {8, 16}-bit, all equal (210 s).  `--quick`: 100 configurations
(tests/test_parity_sweep_gpu.py runs that form under -m gpu).  `--no-image` (round 5): slots without the row-major image
(NHIP_GRID_NO_IMAGE) -- the matcher against the oracle only (the every-add kernels and the grid download need the image)."""
import math, sys, time
import numpy as np
import os
os.environ.setdefault("NHIP_TUNABLES", "1")  # (the library reads its switches only then)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nautilus_amd import csm, synth
from oracle import oracle as O
DEG = math.radians(1.0)
bag = synth.SynthBag(64, dense=True)
sparse = synth.SynthBag(64, dense=False, seed=7)
n_ok = 0
t0 = time.time()
N_CONF = 100 if "--quick" in sys.argv else next((int(a) for a in sys.argv[1:] if a.isdigit()), 60)
NO_IMAGE = "--no-image" in sys.argv
for seed in range(N_CONF):
    rng = np.random.default_rng(1000 + seed)
    B = bag if rng.random() < 0.6 else sparse
    range_m = float(rng.choice([8.0, 12.0, 20.0, 30.0])); res = float(rng.choice([0.05, 0.08, 0.1, 0.025]))
    sigma = float(rng.choice([0.7, 1.0, 2.0, 3.0, 5.0]))
    hx, hy = int(rng.integers(0, 50)), int(rng.integers(0, 50))
    if NO_IMAGE: hx, hy = min(hx, 43), min(hy, 43)  # (lattices of more than 88 translations per axis take the every-add kernel)
    if range_m / res > 900: res = range_m / 600
    n_theta = 2 * int(rng.integers(0, 6)) + 1
    step = float(rng.choice([0.25, 1.0, 3.0])) * DEG
    bits = 16 if seed % 2 else 8
    spec = csm.grid_spec(range_m, res, sigma, 1e-10, max(hx, hy) + 12, bits, no_image=NO_IMAGE); ospec = O.grid_spec(range_m, res, sigma, 1e-10, bits)
    search = csm.search_spec(n_theta, 2 * hx + 1, 2 * hy + 1, step)
    n_pairs = int(rng.integers(1, 20))
    ids = np.unique(rng.integers(0, 64, int(rng.integers(1, 6)))).astype(np.int32)
    src = rng.integers(0, 64, n_pairs).astype(np.int32); slot = rng.integers(0, len(ids), n_pairs).astype(np.int32)
    th0 = rng.uniform(-math.pi, math.pi, n_pairs)
    origin = None
    if rng.random() < 0.3:
        origin = rng.integers(-10, 11, (n_pairs, 2)).astype(np.int32)
    scans = [s.copy() for s in B.scans]
    if rng.random() < 0.3:   # thin / duplicate / clustered points
        k = int(rng.integers(0, 64)); scans[k] = np.repeat(scans[k][::7], 3, axis=0)
    xy, off = csm.pack_scans(scans)
    st = csm.ScanTable(xy, off); grids = csm.LikelihoodGrids(st, ids, spec)
    got, sums = csm.match_pairs(st, grids, src, slot, th0, search, origin)
    if not NO_IMAGE:
        ex = csm.search_spec(n_theta, 2 * hx + 1, 2 * hy + 1, step, exhaustive=True)
        got_x, sums_x = csm.match_pairs(st, grids, src, slot, th0, ex, origin)
        assert got_x.tobytes() == got.tobytes() and np.array_equal(sums_x, sums), ("bnb vs exhaustive", seed)
    ogr = O.grid_build_batch(xy, off, ids, ospec)
    for s_, i_ in enumerate(ids):
        if not NO_IMAGE:
            assert np.array_equal(grids.interior(s_), ogr[s_]), ("grid", seed, s_)
    want = O.csm_match_batch(xy, off, ogr, ospec, src, slot, th0, O.search_spec(n_theta, 2 * hx + 1, 2 * hy + 1, step), origin)
    for f in ("itheta", "ix", "iy"):
        assert np.array_equal(got[f], want[f]), (f, seed)
    assert np.array_equal(sums, want["sum"]), ("sum", seed)
    grids.close(); st.close(); n_ok += 1
print("sweep ok:", n_ok, "configurations in %.1f s" % (time.time() - t0))
