"""Loop-closure candidate gating on the GPU against the oracle's restatement of
LCCandidateFilter::ComputeScatterMatrixScore (/root/reference/src/loop_closure/lc_candidate_filter.cc:22-51) and of the
geometric pair gate.  Bar: bit-exact (the kernel sums in float in point order like the reference)."""
import numpy as np
import pytest

from nautilus_amd import _lib, csm, hostside, posegraph, synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def test_scatter_scores_bit_exact(gpu, small_bag):
    rng = np.random.default_rng(3)
    scans = list(small_bag.scans[:20]) + [np.zeros((0, 2), np.float32), np.array([[1.5, -2.0]], np.float32),
                                          rng.normal(0, 3, (5000, 2)).astype(np.float32),
                                          np.stack([np.linspace(0, 9, 300), np.zeros(300)], 1).astype(np.float32),
                                          (rng.normal(0, 1, (64, 2)) + 1e4).astype(np.float32)]
    xy, off = csm.pack_scans(scans)
    got = posegraph.HipBackend().scatter_scores(xy, off)
    want = O.scatter_matrix_scores(xy, off)
    assert got.tobytes() == want.tobytes()          # NaN for the empty and the one-point scan included
    assert np.isnan(got[20]) and np.isnan(got[21]) and got[22] > 0.9 and got[23] < 1e-6
    # agreement with an independent numpy evaluation (float sums in numpy's order, LAPACK eigenvalues): float rounding
    for i in (0, 5, 22):
        assert abs(got[i] - hostside.scatter_matrix_score(scans[i])) < 5e-6


def test_scatter_scores_device_pointer_api(gpu, small_bag):
    import ctypes as C
    import torch
    xy, off = csm.pack_scans(small_bag.scans)
    dev = torch.device("cuda:0")
    d_xy, d_off = torch.from_numpy(xy).to(dev), torch.from_numpy(off).to(dev)
    d_out = torch.empty(len(off) - 1, dtype=torch.float64, device=dev)
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(_lib.load().nhip_lc_scatter_scores_dev(d_xy.data_ptr(), d_off.data_ptr(), len(off) - 1, d_out.data_ptr(), sp))
    assert d_out.cpu().numpy().tobytes() == O.scatter_matrix_scores(xy, off).tobytes()


def test_pair_gate_and_candidate_walk(gpu):
    bag = synth.SynthBag(600, dense=True)
    xy, off = csm.pack_scans(bag.scans)
    be = posegraph.HipBackend()
    scores = hostside.scatter_scores(be, xy, off)
    cand = hostside.lc_candidates_from_scores(bag.odom, scores, min_score=0.3)
    assert len(cand) > 20
    d = np.linalg.norm(np.diff(bag.odom[cand, :2].astype(np.float32), axis=0), axis=1)
    assert np.all(d >= 5.0 - 1e-5)                   # DistantFromLastScan, lc_candidate_filter.cc:53-62
    flags = be.pair_gate(bag.odom, cand, 3.5, 20)
    assert flags.tobytes() == O.pair_gate(bag.odom, cand, 3.5, 20).tobytes()
    src, tgt = hostside.geometric_pair_gate(bag.odom, cand, 3.5, 20, backend=be)
    assert len(src) > 0 and np.all(src > tgt) and np.all(src - tgt > 20)
    assert np.all(np.linalg.norm(bag.odom[src, :2] - bag.odom[tgt, :2], axis=1) < 3.5 + 1e-4)
    # the reference's own threshold (0.70) keeps fewer scans: most scans of this 24 m x 16 m room score ~0.4
    strict = hostside.lc_candidates_from_scores(bag.odom, scores)
    assert len(strict) < len(cand) and all(scores[i] >= 0.70 for i in strict)


def test_chi_square_gate_bit_exact(gpu, chi_square_cases):
    """nhip_lc_chi_square_gate (host-pointer and device-pointer forms) against the oracle's restatement of
    ChiSquareScore / GetPossibleMatches (lc_matcher.cc:50-74) on 4,000 pairs with covariance blocks of every
    conditioning, singular and zero blocks included: scores equal as bit patterns (NaN and inf too), flags equal."""
    import ctypes as C
    import torch
    poses, src, tgt, cov = chi_square_cases
    be = posegraph.HipBackend()
    want_s, want_f = O.chi_square_gate(poses, src, tgt, cov, 5000.0)
    got_s, got_f = be.chi_square_gate(poses, src, tgt, cov, 5000.0)
    assert got_s.tobytes() == want_s.tobytes() and got_f.tobytes() == want_f.tobytes()
    assert 0 < got_f.sum() < len(got_f) and np.isnan(got_s).any()
    dev = torch.device("cuda:0")
    d = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (poses, src, tgt, cov.reshape(-1, 4))]
    d_s, d_f = torch.empty(len(src), dtype=torch.float64, device=dev), torch.empty(len(src), dtype=torch.uint8, device=dev)
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(_lib.load().nhip_lc_chi_square_gate_dev(d[0].data_ptr(), len(poses), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(),
                                                      len(src), 100.0, d_s.data_ptr(), d_f.data_ptr(), sp))
    w_s, w_f = O.chi_square_gate(poses, src, tgt, cov, 100.0)
    assert d_s.cpu().numpy().tobytes() == w_s.tobytes() and d_f.cpu().numpy().tobytes() == w_f.tobytes()
    # argument checks: a pair outside the pose table is refused before anything is launched
    with pytest.raises(_lib.NhipError):
        be.chi_square_gate(poses, [0], [300], cov[:1])
    assert be.chi_square_gate(poses, [], [], np.zeros((0, 2, 2), np.float32))[0].size == 0
