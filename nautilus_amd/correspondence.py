"""Device-resident ICP front half: correspondence search -> residual / normal-equation evaluation.

Mirrors the reference's per-window problem build, batched:
  Solver::BuildOptimizationOverWindow      /root/reference/src/optimization/solver.cc:321-333
    -> AddLidarResiduals(i, j)             solver.cc:297-318
       -> GetPointToPointMatching(i, j)    solver.cc:132-172   (K5, nhip_corr_search_dev)
       -> LIDARPointResidual / LIDARNormalResidual blocks      (K4, nhip_resid_lidar_dev)
Everything stays in HBM (torch tensors are only the allocator): correspondences never visit the
host, and with `normal_equations()` neither do Jacobians.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check


def window_pairs(n_scans, window):
    """All (i, j), j in [max(i - window, 0), i): solver.cc:324-330."""
    i = np.repeat(np.arange(n_scans, dtype=np.int64), window)
    j = i - window + np.tile(np.arange(window, dtype=np.int64), n_scans)  # i - window .. i - 1, ascending, as the loop does
    keep = j >= 0
    return i[keep].astype(np.int32), j[keep].astype(np.int32)


class DeviceArena:
    """Device buffers that outlive one IcpBatch: a solver that rebuilds its problem per window pass (OptimizeOverGrowingWindow,
    solver.cc:339-355: thirty problem builds in the 10,000-scan loop) uploads the clouds ONCE and keeps its work buffers --
    gigabytes at window 10 -- instead of a hipMalloc / upload per build (measured: 0.2-1.5 s of a run's 0.6-1.9 s of path
    time, depending on the driver's state).  One batch uses the arena at a time: the newest owns it; an older batch that is
    used again takes it back and searches its correspondences anew."""

    def __init__(self):
        self.buf = {}
        self.clouds = None  # (key, d_xy, d_nrm, d_off, the host arrays: their ids stay theirs while the key is kept)
        self.owner = None

    def take(self, torch, dev, name, n, dtype):
        n = max(int(n), 1)
        t = self.buf.get(name)
        if t is None or t.dtype != dtype or t.numel() < n:
            self.buf[name] = None  # (release before the larger allocation)
            t = self.buf[name] = torch.empty(n, dtype=dtype, device=dev)
        return t

    def reserve(self, torch, dev, capacity, n_blocks):
        """Sizes of the largest batch to come (the last window of a growing-window solve): one allocation instead of ten."""
        for name, n, dt in (("padded", 8 * capacity, torch.float32), ("corr", 8 * capacity, torch.float32),
                            ("cblock", capacity, torch.int32), ("counts", n_blocks, torch.int32),
                            ("boff", n_blocks + 1, torch.int32), ("consts", 8 * n_blocks, torch.float64),
                            ("neq", 28 * n_blocks, torch.float64), ("cap", n_blocks + 1, torch.int64),
                            ("bsrc", n_blocks, torch.int32), ("btgt", n_blocks, torch.int32)):
            self.take(torch, dev, name, n, dt)


class IcpBatch:
    def __init__(self, xy, normals, offsets, block_src, block_tgt, device="cuda:0", outlier_threshold=0.25,
                 min_abs_cosine=None, arena=None):
        import torch
        self.torch = torch
        self.dev = torch.device(device)
        self.lib = _lib.load()
        self.arena = arena
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(self.dev)
        self.offsets_h = np.ascontiguousarray(offsets, dtype=np.int32)
        self.n_scans = len(self.offsets_h) - 1
        # the clouds: uploaded once per arena (the same host arrays, by identity and size, every window pass)
        key = (id(xy), id(normals), id(offsets), np.shape(xy), np.shape(normals), len(self.offsets_h))
        if arena is not None and arena.clouds is not None and arena.clouds[0] == key:
            _, self.d_xy, self.d_nrm, self.d_off, _ = arena.clouds
        else:
            self.d_xy, self.d_nrm, self.d_off = t(xy, np.float32), t(normals, np.float32), t(offsets, np.int32)
            if arena is not None:
                arena.clouds = (key, self.d_xy, self.d_nrm, self.d_off, (xy, normals, offsets))
        self.block_src = np.ascontiguousarray(block_src, dtype=np.int32)
        self.block_tgt = np.ascontiguousarray(block_tgt, dtype=np.int32)
        self.n_blocks = len(self.block_src)
        cap = np.zeros(self.n_blocks + 1, dtype=np.int64)
        cap[1:] = np.cumsum(self.offsets_h[self.block_src + 1] - self.offsets_h[self.block_src])
        self.cap_h = cap
        self.capacity = int(cap[-1])
        self.thr = float(outlier_threshold)
        # None: GetPointToPointMatching; a value (the reference uses cos(20 deg)): the normal gate of
        # GetPointToNormalMatching / FindClosestPointWithSimilarNormal (solver.cc:177-260)
        self.min_cos = None if min_abs_cosine is None else float(min_abs_cosine)
        e = lambda n, dt: torch.empty(max(int(n), 1), dtype=dt, device=self.dev)
        self.d_aff = e(4 * self.n_scans, torch.float32)
        self.d_poses = e(3 * self.n_scans, torch.float64)
        self.d_res = self.d_js = self.d_jt = self.d_neq = None
        self.n_corr = 0
        self._searched = False
        self._stale = False        # the arena was taken back after a search: d_corr / d_boff / d_cblock are another batch's
        self.d_aff_search = None   # the affines of the last search(): what a re-search after a take-back runs at
        self._bind()

    def _bind(self):
        """The batch's work buffers: its own, or the arena's (then this batch owns the arena until a newer one binds)."""
        torch, a = self.torch, self.arena
        if a is None:
            e = lambda name, n, dt: torch.empty(max(int(n), 1), dtype=dt, device=self.dev)
        else:
            e = lambda name, n, dt: a.take(torch, self.dev, name, n, dt)
            a.owner = self
            # a batch that re-takes an arena it had searched in holds stale correspondences, WHOEVER owns the arena when
            # the next call that needs them arrives (set_poses() first, as PoseGraph._assemble_inner does, makes this
            # batch the owner again without searching)
            self._stale = self._searched
        up = lambda name, h, dt: e(name, len(h), dt)[:len(h)].copy_(torch.from_numpy(h)) if len(h) else e(name, 1, dt)[:0]
        self.d_bsrc, self.d_btgt = up("bsrc", self.block_src, torch.int32), up("btgt", self.block_tgt, torch.int32)
        self.d_cap = up("cap", self.cap_h, torch.int64)
        self.d_padded = e("padded", 8 * self.capacity, torch.float32)
        self.d_counts = e("counts", self.n_blocks, torch.int32)
        self.d_boff = e("boff", self.n_blocks + 1, torch.int32)
        self.d_corr = e("corr", 8 * self.capacity, torch.float32)
        self.d_cblock = e("cblock", self.capacity, torch.int32)
        self.d_consts = e("consts", 8 * self.n_blocks, torch.float64)
        if a is not None:
            self.d_neq = None

    def _own(self, need_corr):
        """Before every use: a batch whose arena a newer batch has bound takes it back (block lists uploaded again) and, if
        the call needs correspondences, searches them anew at its current poses."""
        if self.arena is not None and self.arena.owner is not self:
            self._bind()
        if need_corr and self._stale:
            # at the poses the correspondences were FOUND at (a caller who keeps them across set_poses() calls --
            # research=False -- evaluates new poses on the old matches), not at the current ones
            self.search(_aff=self.d_aff_search)

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream().cuda_stream)

    def set_poses(self, poses):
        self._own(False)
        poses = np.ascontiguousarray(poses, dtype=np.float64).reshape(self.n_scans, 3)
        aff = np.empty((self.n_scans, 4), dtype=np.float32)
        check(self.lib.nhip_pose_affines(_lib.ptr(poses), self.n_scans, _lib.ptr(aff)))
        self.d_aff.copy_(self.torch.from_numpy(aff.reshape(-1)))
        self.d_poses.copy_(self.torch.from_numpy(poses.reshape(-1)))

    def search(self, sync=True, _aff=None):
        """K5 + compaction.  Returns the number of correspondences (needs one sync to size outputs)."""
        self._own(False)
        self._searched = True
        self._stale = False
        if _aff is None:  # (a fresh search: remember where it ran; a re-search after a take-back runs there again)
            if self.d_aff_search is None:
                self.d_aff_search = self.torch.empty_like(self.d_aff)
            self.d_aff_search.copy_(self.d_aff)
            _aff = self.d_aff
        sp = self._stream()
        if self.min_cos is None:
            check(self.lib.nhip_corr_search_dev(self.d_xy.data_ptr(), self.d_nrm.data_ptr(), self.d_off.data_ptr(), self.n_scans,
                                                self.d_bsrc.data_ptr(), self.d_btgt.data_ptr(), self.n_blocks,
                                                _aff.data_ptr(), self.thr, self.d_cap.data_ptr(),
                                                self.d_padded.data_ptr(), self.d_counts.data_ptr(), sp))
        else:
            check(self.lib.nhip_corr_search_normals_dev(
                self.d_xy.data_ptr(), self.d_nrm.data_ptr(), self.d_off.data_ptr(), self.n_scans, self.d_bsrc.data_ptr(),
                self.d_btgt.data_ptr(), self.n_blocks, _aff.data_ptr(), self.thr, self.min_cos,
                self.d_cap.data_ptr(), self.d_padded.data_ptr(), self.d_counts.data_ptr(), sp))
        check(self.lib.nhip_corr_compact_dev(self.d_padded.data_ptr(), self.d_cap.data_ptr(),
                                             self.d_counts.data_ptr(), self.n_blocks, self.d_boff.data_ptr(),
                                             self.d_corr.data_ptr(), self.d_cblock.data_ptr(), sp))
        if sync:
            self.n_corr = int(self.d_boff[self.n_blocks].item())
        return self.n_corr

    def correspondences(self):
        """Host copy: (rows (n_corr, 8), block_offsets (n_blocks + 1))."""
        self._own(True)
        n = self.n_corr
        return (self.d_corr[:8 * n].cpu().numpy().reshape(n, 8), self.d_boff.cpu().numpy())

    def residuals(self, kind, jacobians=True):
        self._own(True)
        torch, n = self.torch, self.n_corr
        if self.d_res is None or self.d_res.numel() < 2 * max(n, 1):
            self.d_res = torch.empty(2 * max(n, 1), dtype=torch.float64, device=self.dev)
            self.d_js = torch.empty(6 * max(n, 1), dtype=torch.float64, device=self.dev)
            self.d_jt = torch.empty(6 * max(n, 1), dtype=torch.float64, device=self.dev)
        check(self.lib.nhip_resid_lidar_dev(kind, self.d_corr.data_ptr(), self.d_cblock.data_ptr(), n,
                                            self.d_bsrc.data_ptr(), self.d_btgt.data_ptr(), self.n_blocks,
                                            self.d_poses.data_ptr(), self.n_scans, self.d_consts.data_ptr(),
                                            self.d_res.data_ptr(), self.d_js.data_ptr() if jacobians else None,
                                            self.d_jt.data_ptr() if jacobians else None, self._stream()))
        return self.d_res[:2 * n], (self.d_js[:6 * n] if jacobians else None), (self.d_jt[:6 * n] if jacobians else None)

    def normal_equations(self, kind):
        """Per block: 21 upper-triangle entries of J^T J, 6 of J^T r, r^T r (28 doubles)."""
        self._own(True)
        torch = self.torch
        if self.d_neq is None:
            self.d_neq = (torch.empty(28 * max(self.n_blocks, 1), dtype=torch.float64, device=self.dev) if self.arena is None
                          else self.arena.take(torch, self.dev, "neq", 28 * self.n_blocks, torch.float64))
        check(self.lib.nhip_resid_lidar_normal_eq_dev(kind, self.d_corr.data_ptr(), self.d_boff.data_ptr(),
                                                      self.d_bsrc.data_ptr(), self.d_btgt.data_ptr(), self.n_blocks,
                                                      self.d_poses.data_ptr(), self.n_scans,
                                                      self.d_consts.data_ptr(), self.d_neq.data_ptr(), self._stream()))
        return self.d_neq[:28 * self.n_blocks].view(self.n_blocks, 28)
