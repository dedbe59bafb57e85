"""Host-side pose-graph driver over the GPU hot path.

The reference hands its residual blocks to Ceres (SPARSE_SCHUR, solver.cc:266-275, 335-356), which
is not installable here; this module is the small caller that lets the whole loop run on the
batched API instead:
  * ICP blocks of the sliding window (solver.cc:321-333): correspondence search (K5) and per-block
    6x6 normal equations on the GPU (nhip_corr_search_dev, nhip_resid_lidar_normal_eq_dev);
  * odometry factors (OdometryResidual, slam_residuals.h:18-40; AddOdomFactors solver.cc:370-387)
    and loop-closure constraints from the scan matcher ("Add Odometry residual using the returned
    relative transform", the TODO at solver.cc:651-660), both evaluated by nhip_resid_odometry_dev;
  * HITL constraints (solver.cc:479-559): PointToLineResidual blocks of the points the user's two
    segments select, every block against line_a and one shared `chosen_line_pose` parameter block
    (AddHITLResiduals, solver.cc:515-532), evaluated by nhip_resid_point_to_line;
  * Gauss-Newton with Levenberg damping on the assembled sparse system (scipy.sparse on the
    host: N poses x 3 + 3 per HITL constraint), first pose held constant (solver.cc:384-386).
Only the linear solve and the bookkeeping are host work; every residual, Jacobian and nearest
neighbour comes from the backend -- HipBackend (the product: libnautilus_hip) unless a test or the
bench's cpu_baseline leg injects another one (oracle/cpu_backend.py times the same loop on the CPU
restatement; the product never imports it).
"""
import contextlib
import ctypes as C
import math
import time

import numpy as np

from . import _lib
from ._lib import check
from .correspondence import IcpBatch, window_pairs


# Where the loop's wall-clock goes, by owner (seconds since the last reset): "path" = calls into the backend -- the hot
# path of this repo: correspondence search, residuals / normal equations, gating, scan matching (every one returns
# host data, so the device is drained when the clock stops); "host_solver" = the sparse linear solves (Ceres' job in
# the reference: out of scope, the same code whatever the backend); the rest of a run is host bookkeeping.
# "marshal" = host work at the path's boundary: building the block lists and input arrays the backend calls take and
# turning their outputs into what the caller asked for; "assemble" = the sparse system's assembly from the per-block
# normal equations and factor Jacobians (numpy: what Ceres does inside its solve, like "host_solver"); "hitl_select" =
# GetRelevantPosesForHITL's point selection (host-side HITL curation, solver.cc:479-513).
CLOCK = {"path": 0.0, "host_solver": 0.0, "marshal": 0.0, "assemble": 0.0, "hitl_select": 0.0, "path_setup": 0.0}


def clock_reset():
    for k in CLOCK:
        CLOCK[k] = 0.0


@contextlib.contextmanager
def clocked(key):
    t0 = time.perf_counter()
    try:
        yield
    finally:
        CLOCK[key] += time.perf_counter() - t0


def compose(pose, rel):
    """pose (x, y, th) o rel (tx, ty, th): the pose of A given B and A-in-B (solver.cc:640-648)."""
    c, s = math.cos(pose[2]), math.sin(pose[2])
    return np.array([pose[0] + c * rel[0] - s * rel[1], pose[1] + s * rel[0] + c * rel[1], pose[2] + rel[2]])


class HipBackend:
    """Every evaluation on the MI355X through the C ABI."""
    name = "hip"

    def __init__(self, device="cuda:0"):
        import torch
        self.torch, self.dev, self.lib = torch, torch.device(device), _lib.load()
        from .correspondence import DeviceArena
        self.arena = DeviceArena()  # the clouds and the work buffers of the ICP batches, kept across problem builds

    def icp(self, xy, normals, offsets, block_src, block_tgt, outlier_threshold):
        return _HipIcp(IcpBatch(xy, normals, offsets, block_src, block_tgt, str(self.dev), outlier_threshold, arena=self.arena))

    def reserve_icp(self, offsets, window):
        """Work buffers for the largest problem of a growing-window solve (all (i, j), j in [i - window, i)): allocated once."""
        off = np.asarray(offsets, dtype=np.int64)
        per = np.minimum(np.arange(len(off) - 1), int(window))  # blocks whose source is scan i
        self.arena.reserve(self.torch, self.dev, int((np.diff(off) * per).sum()), int(per.sum()))

    def odometry(self, pose_i, pose_j, t_odom, r_odom, tw, rw, poses):
        """OdometryResidual blocks at `poses` (n, 3): residuals (F, 3), Jacobians (F, 3, 3) x 2."""
        n = len(pose_i)
        r, ji, jj = np.empty((n, 3)), np.empty((n, 3, 3)), np.empty((n, 3, 3))
        P = np.ascontiguousarray(poses, dtype=np.float64)
        check(self.lib.nhip_resid_odometry(_lib.ptr(t_odom), _lib.ptr(r_odom), _lib.ptr(pose_i), _lib.ptr(pose_j), n,
                                           float(tw), float(rw), _lib.ptr(P), len(P), _lib.ptr(r), _lib.ptr(ji), _lib.ptr(jj)))
        return r, ji, jj

    def point_to_line(self, segments, points, point_block, block_pose, block_line, poses, line_poses):
        """PointToLineResidual blocks: residuals (n,), d/d pose (n, 3), d/d line_pose (n, 3)."""
        n = len(points)
        r, j0, j1 = np.empty(n), np.empty((n, 3)), np.empty((n, 3))
        P, Lp = np.ascontiguousarray(poses, dtype=np.float64), np.ascontiguousarray(line_poses, dtype=np.float64)
        check(self.lib.nhip_resid_point_to_line(_lib.ptr(segments), _lib.ptr(points), _lib.ptr(point_block), n,
                                                _lib.ptr(block_pose), _lib.ptr(block_line), len(block_pose), _lib.ptr(P), len(P),
                                                _lib.ptr(Lp), len(Lp), _lib.ptr(r), _lib.ptr(j0), _lib.ptr(j1)))
        return r, j0, j1

    def scatter_scores(self, xy, offsets):
        """LCCandidateFilter's scatter-matrix score of every scan (nhip_lc_scatter_scores)."""
        from . import csm
        st = csm.ScanTable(xy, offsets)
        out = np.empty(st.n_scans)
        check(self.lib.nhip_lc_scatter_scores(st._h, _lib.ptr(out)))
        st.close()
        return out

    def pair_gate(self, poses, candidates, max_range, min_separation):
        P = np.ascontiguousarray(poses, dtype=np.float64)
        cand = np.ascontiguousarray(candidates, dtype=np.int32)
        flags = np.zeros((len(cand), len(cand)), dtype=np.uint8)
        check(self.lib.nhip_lc_pair_gate(_lib.ptr(P), len(P), _lib.ptr(cand), len(cand), float(max_range),
                                         int(min_separation), _lib.ptr(flags)))
        return flags

    def chi_square_gate(self, poses, pair_src, pair_tgt, cov, max_score=5000.0):
        """LCMatcher's chi-square test of n (source, candidate) pairs given their cross-covariance blocks
        (nhip_lc_chi_square_gate; lc_matcher.cc:50-74): (scores float64, flags uint8)."""
        P = np.ascontiguousarray(poses, dtype=np.float64)
        src = np.ascontiguousarray(pair_src, dtype=np.int32)
        tgt = np.ascontiguousarray(pair_tgt, dtype=np.int32)
        cov = np.ascontiguousarray(cov, dtype=np.float32).reshape(-1, 4)
        if not len(cov) == len(src) == len(tgt):
            raise ValueError("chi_square_gate: one covariance block per pair")
        scores, flags = np.zeros(len(src)), np.zeros(len(src), dtype=np.uint8)
        check(self.lib.nhip_lc_chi_square_gate(_lib.ptr(P), len(P), _lib.ptr(src), _lib.ptr(tgt), _lib.ptr(cov), len(src),
                                               float(max_score), _lib.ptr(scores), _lib.ptr(flags)))
        return scores, flags

    def match(self, xy, offsets, pair_src, pair_tgt, theta0, cell_bits=16):
        """Batched loop-closure scan matching (BASELINE config #2 lattice): (records, spec, search).
        Only the scans the list names go to the device (the candidate scans of a 10,000-scan bag are ~150: 1.3 MB instead
        of the bag's 86 MB, whose upload cost more than matching the 3,275 pairs)."""
        from . import csm
        spec = csm.grid_spec(30.0, 0.05, 2.0, 1e-10, 40, cell_bits)
        search = csm.search_spec(61, 81, 81, math.radians(1.0))
        pair_src, pair_tgt = np.asarray(pair_src), np.asarray(pair_tgt)
        used = np.unique(np.concatenate([pair_src, pair_tgt]))
        off = np.asarray(offsets, dtype=np.int64)
        xy2 = np.asarray(xy, dtype=np.float32).reshape(-1, 2)
        sub_xy = np.concatenate([xy2[off[i]:off[i + 1]] for i in used]) if len(used) else np.zeros((0, 2), np.float32)
        sub_off = np.concatenate([[0], np.cumsum(off[used + 1] - off[used])]).astype(np.int32)
        src, tgt = np.searchsorted(used, pair_src).astype(np.int32), np.searchsorted(used, pair_tgt).astype(np.int32)
        ids = np.unique(tgt).astype(np.int32)
        slot = np.searchsorted(ids, tgt).astype(np.int32)
        n, lib, torch, dev = len(src), self.lib, self.torch, self.dev
        if n == 0:
            return np.zeros(0, dtype=csm.MATCH_DTYPE), spec, search
        if len(sub_off) > 1 and int(np.diff(sub_off).max()) <= _lib.NHIP_SHORT_SCAN_POINTS:
            search.flags |= _lib.NHIP_SEARCH_SHORT_SCANS
        # Device buffers from torch's caching allocator (the tables of 150 targets are 1.8 GB: a hipMalloc / hipFree pair of
        # that size per call cost several times the match), the device-pointer entry points on torch's current stream.
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        d_xy, d_off, d_ids, d_src, d_slot = t(sub_xy), t(sub_off), t(ids), t(src), t(slot)
        rot0 = np.empty((n, 2))
        check(lib.nhip_csm_rot0(_lib.ptr(np.ascontiguousarray(theta0, dtype=np.float64)), None, n, _lib.ptr(rot0)))
        d_rot0, d_delta = t(rot0), t(csm.delta_table(search))
        # (nhip_grid_build_dev zero-fills the slots itself; only the 256 bytes of read slack behind them are this caller's)
        d_grids = torch.empty(lib.nhip_grids_bytes(C.byref(spec), len(ids)), dtype=torch.uint8, device=dev)
        d_grids[-256:].zero_()
        ws_g = lib.nhip_grid_workspace_bytes(C.byref(spec), len(ids))
        d_ws_g = torch.empty(ws_g, dtype=torch.uint8, device=dev)
        ws_m = lib.nhip_csm_workspace_bytes(n)
        d_ws_m = torch.empty(ws_m, dtype=torch.uint8, device=dev)
        d_keys = torch.empty(n, dtype=torch.int64, device=dev)
        d_out = torch.empty((n, 4), dtype=torch.int32, device=dev)
        sp = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        n_sub = len(sub_off) - 1
        check(lib.nhip_grid_build_dev(d_xy.data_ptr(), d_off.data_ptr(), n_sub, d_ids.data_ptr(), len(ids), C.byref(spec),
                                      d_grids.data_ptr(), d_ws_g.data_ptr(), ws_g, sp))
        check(lib.nhip_csm_match_dev(d_xy.data_ptr(), d_off.data_ptr(), n_sub, d_grids.data_ptr(), len(ids), C.byref(spec), d_src.data_ptr(),
                                     d_slot.data_ptr(), d_rot0.data_ptr(), d_delta.data_ptr(), None, n, C.byref(search),
                                     d_keys.data_ptr(), d_out.data_ptr(), None, d_ws_m.data_ptr(), ws_m, sp))
        m = d_out.cpu().numpy().view(csm.MATCH_DTYPE).reshape(-1).copy()
        return m, spec, search


class _HipIcp:
    def __init__(self, batch):
        self.b = batch
        self.block_src, self.block_tgt = batch.block_src, batch.block_tgt

    def set_poses(self, poses):
        self.b.set_poses(poses)

    def search(self):
        return self.b.search()

    def normal_equations(self, kind):
        return self.b.normal_equations(kind).cpu().numpy()

    @property
    def n_corr(self):
        return self.b.n_corr


class OdometryFactors:
    """Batched OdometryResidual blocks: r = (w_t (T_i + T_odom - T_j), w_r wrap(th_i + R_odom - th_j))."""

    def __init__(self, pose_i, pose_j, t_odom, r_odom, tw=1.0, rw=1.0):
        self.n = len(pose_i)
        self.pose_i = np.ascontiguousarray(pose_i, dtype=np.int32)
        self.pose_j = np.ascontiguousarray(pose_j, dtype=np.int32)
        self.t_odom = np.ascontiguousarray(np.reshape(t_odom, (-1, 2)), dtype=np.float32)
        self.r_odom = np.ascontiguousarray(r_odom, dtype=np.float32)
        self.tw, self.rw = float(tw), float(rw)

    def evaluate(self, backend, poses):
        if self.n == 0:
            return np.zeros((0, 3)), np.zeros((0, 3, 3)), np.zeros((0, 3, 3))
        return backend.odometry(self.pose_i, self.pose_j, self.t_odom, self.r_odom, self.tw, self.rw, poses)


def odometry_factors_from_poses(odom, **kw):
    """Consecutive-pose factors in the functor's convention: world-frame translation delta and heading
    delta of the odometry track (what GetSolvedOdomFactors produces from poses, solver.cc:406-427)."""
    d = np.diff(odom, axis=0)
    n = len(odom)
    return OdometryFactors(np.arange(n - 1), np.arange(1, n), d[:, :2], d[:, 2], **kw)


def loop_closure_factors(poses, pairs_src, pairs_tgt, rel, **kw):
    """One odometry-style constraint per accepted loop closure: the matcher says where scan src sits
    in scan tgt's frame; expressed in the functor's world-frame convention around the current estimate
    of the TARGET pose: T_src - T_tgt = R(th_tgt) t_rel, th_src - th_tgt = th_rel."""
    t_odom, r_odom = [], []
    for s, t, r in zip(pairs_src, pairs_tgt, rel):
        pred = compose(poses[t], r)
        t_odom.append(pred[:2] - poses[t][:2])
        r_odom.append(r[2])
    return OdometryFactors(np.asarray(pairs_tgt), np.asarray(pairs_src), np.asarray(t_odom).reshape(-1, 2),
                           np.asarray(r_odom), **kw)


class HitlConstraint:
    """HitlLCConstraint (data_structures.h:41-51) + its residual blocks (AddHITLResiduals, solver.cc:515-532): the
    points GetRelevantPosesForHITL selected on line a and on line b, EVERY block a PointToLineResidual against
    line_a, all sharing one extra parameter block `chosen_line_pose` (initialised to zero)."""

    def __init__(self, line_a, line_b, a_poses, b_poses):
        self.line_a = np.ascontiguousarray(line_a, dtype=np.float32).reshape(4)
        self.line_b = np.ascontiguousarray(line_b, dtype=np.float32).reshape(4)
        self.blocks = [(int(i), np.ascontiguousarray(p, dtype=np.float32).reshape(-1, 2)) for i, p in list(a_poses) + list(b_poses)]
        self.n_a, self.n_b = len(a_poses), len(b_poses)
        self.chosen_line_pose = np.zeros(3)

    def arrays(self, line_index):
        """The input arrays of nhip_resid_point_to_line for this constraint's blocks.  The blocks never change after
        construction (like the functors' copied vectors, slam_residuals.h:214-215): built once, not per evaluation -- at
        10,000 scans they hold 1.3 M points, and a solve evaluates them dozens of times."""
        if getattr(self, "_arrays", None) is None or self._arrays[0] != line_index:
            nb = len(self.blocks)
            seg = np.tile(self.line_a, (nb, 1)).astype(np.float32)
            pts = np.concatenate([p for _, p in self.blocks]).astype(np.float32) if nb else np.zeros((0, 2), np.float32)
            pb = np.repeat(np.arange(nb, dtype=np.int32), [len(p) for _, p in self.blocks]) if nb else np.zeros(0, np.int32)
            bp = np.array([i for i, _ in self.blocks], dtype=np.int32)
            bl = np.full(nb, line_index, dtype=np.int32)
            self._arrays = (line_index, (seg, pts, pb, bp, bl))
        return self._arrays[1]


class PoseGraph:
    def __init__(self, xy, normals, offsets, odom, window=10, kind=_lib.NHIP_LIDAR_POINT, outlier_threshold=0.25,
                 odom_weights=(1.0, 1.0), device="cuda:0", initial=None, backend=None):
        self.n = len(odom)
        self.kind = kind
        self.backend = backend if backend is not None else HipBackend(device)
        with clocked("marshal"):
            bs, bt = window_pairs(self.n, window)
        with clocked("path"), clocked("path_setup"):  # (uploads of the clouds and block lists, device allocations: part of "path")
            self.icp = self.backend.icp(xy, normals, offsets, bs, bt, outlier_threshold)
        self.odo = odometry_factors_from_poses(odom, tw=odom_weights[0], rw=odom_weights[1])
        self.lc = None
        self.hitl = []
        # odometry factors always come from `odom`; the estimate may start elsewhere (previous window pass)
        self.poses = np.array(odom if initial is None else initial, dtype=np.float64)

    def add_loop_closures(self, pairs_src, pairs_tgt, rel, weights=(10.0, 10.0)):
        self.lc = loop_closure_factors(self.poses, pairs_src, pairs_tgt, rel, tw=weights[0], rw=weights[1])

    def add_hitl(self, constraint):
        self.hitl.append(constraint)

    @property
    def n_unknowns(self):
        return 3 * self.n + 3 * len(self.hitl)

    def _assemble(self, poses, lines, research):
        """The sparse normal equations at `poses`.  Clocks: the backend calls are "path", building their input arrays
        "marshal", everything else here "assemble" (numpy: the solver's own bookkeeping)."""
        t_in, path0, marsh0 = time.perf_counter(), CLOCK["path"], CLOCK["marshal"]
        try:
            return self._assemble_inner(poses, lines, research)
        finally:
            CLOCK["assemble"] += (time.perf_counter() - t_in) - (CLOCK["path"] - path0) - (CLOCK["marshal"] - marsh0)

    def _assemble_inner(self, poses, lines, research):
        import scipy.sparse as sp
        N, NU = self.n, self.n_unknowns
        with clocked("path"):
            self.icp.set_poses(poses)
            if research:
                self.icp.search()  # correspondences are rebuilt per solve, like each window pass of the reference
            neq = self.icp.normal_equations(self.kind)
        rows, cols, vals = [], [], []
        g = np.zeros(NU)
        cost = 0.5 * float(neq[:, 27].sum()) if len(neq) else 0.0
        iu = np.triu_indices(6)
        bs, bt = self.icp.block_src, self.icp.block_tgt
        H6 = np.zeros((len(neq), 6, 6))
        H6[:, iu[0], iu[1]] = neq[:, :21]
        H6 = H6 + np.transpose(H6, (0, 2, 1)) - np.einsum("bij,ij->bij", H6, np.eye(6))
        idx = np.concatenate([3 * bs[:, None] + np.arange(3), 3 * bt[:, None] + np.arange(3)], axis=1)  # (B, 6)
        rows.append(np.repeat(idx, 6, axis=1).ravel())
        cols.append(np.tile(idx, (1, 6)).ravel())
        vals.append(H6.ravel())
        np.add.at(g, idx.ravel(), neq[:, 21:27].ravel())
        for fac in (self.odo, self.lc):
            if fac is None or fac.n == 0:
                continue
            with clocked("path"):
                r, ji, jj = fac.evaluate(self.backend, poses)
            J = np.concatenate([ji, jj], axis=2)  # (F, 3, 6)
            Hf = np.einsum("fki,fkj->fij", J, J)
            gf = np.einsum("fki,fk->fi", J, r)
            idf = np.concatenate([3 * fac.pose_i[:, None] + np.arange(3), 3 * fac.pose_j[:, None] + np.arange(3)], axis=1)
            rows.append(np.repeat(idf, 6, axis=1).ravel())
            cols.append(np.tile(idf, (1, 6)).ravel())
            vals.append(Hf.ravel())
            np.add.at(g, idf.ravel(), gf.ravel())
            cost += 0.5 * float((r * r).sum())
        for c, con in enumerate(self.hitl):
            with clocked("marshal"):
                seg, pts, pb, bp, bl = con.arrays(c)
            if len(pts) == 0:
                continue
            with clocked("path"):
                r, j0, j1 = self.backend.point_to_line(seg, pts, pb, bp, bl, poses, lines)
            J = np.concatenate([j0, j1], axis=1)  # (n, 6): d/d pose of the point's node, d/d chosen_line_pose
            ids = np.concatenate([3 * bp[pb][:, None] + np.arange(3), np.full((len(pts), 1), 3 * N + 3 * c) + np.arange(3)], axis=1)
            Hp = np.einsum("ni,nj->nij", J, J)
            rows.append(np.repeat(ids, 6, axis=1).ravel())
            cols.append(np.tile(ids, (1, 6)).ravel())
            vals.append(Hp.ravel())
            np.add.at(g, ids.ravel(), (J * r[:, None]).ravel())
            cost += 0.5 * float((r * r).sum())
        H = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(NU, NU)).tocsc()
        return H, g, cost

    def _lines(self):
        return np.array([c.chosen_line_pose for c in self.hitl], dtype=np.float64).reshape(-1, 3)

    def cross_covariances(self, pairs):
        """Covariance blocks the way LCMatcher asks Ceres for them (GetCovarianceMatrix,
        lc_matcher.cc:28-46): the (source pose, target pose) cross block of (J^T J)^-1 of the current
        problem with the pose just before the earlier of the two held constant instead of pose 0;
        returns the top-left 2 x 2 (translation) of each 3 x 3 block, float32 like the reference.
        J^T J comes from the backend's per-block normal equations; the sparse solves are host work."""
        import scipy.sparse as sp
        from scipy.sparse.linalg import splu
        H, _, _ = self._assemble(self.poses, self._lines(), research=False)
        H = H.tocsc()[:3 * self.n][:, :3 * self.n]
        out = np.zeros((len(pairs), 2, 2), dtype=np.float32)
        by_gauge = {}
        for k, (s_, t_) in enumerate(pairs):
            by_gauge.setdefault(max(min(int(s_), int(t_)) - 1, 0), []).append(k)
        for gauge, ks in by_gauge.items():
            free = np.concatenate([np.arange(0, 3 * gauge), np.arange(3 * gauge + 3, 3 * self.n)])
            pos = -np.ones(3 * self.n, dtype=np.int64)
            pos[free] = np.arange(len(free))
            lu = splu(H[free][:, free].tocsc() + 1e-12 * sp.identity(len(free), format="csc"))
            for k in ks:
                s_, t_ = int(pairs[k][0]), int(pairs[k][1])
                if s_ == gauge or t_ == gauge:
                    continue  # a constant block has no covariance
                rhs = np.zeros((len(free), 2))
                rhs[pos[3 * t_], 0] = 1.0
                rhs[pos[3 * t_ + 1], 1] = 1.0
                x = lu.solve(rhs)
                out[k] = x[[pos[3 * s_], pos[3 * s_ + 1]], :].astype(np.float32)
        return out

    def solve(self, iterations=8, damping=1e-3, verbose=False):
        """Gauss-Newton with Levenberg damping; pose 0 constant (SetParameterBlockConstant, solver.cc:384-386)."""
        import scipy.sparse as sp
        from scipy.sparse.linalg import spsolve
        poses, lines = self.poses.copy(), self._lines()
        H, g, cost = self._assemble(poses, lines, research=True)
        history = [cost]
        NU = self.n_unknowns
        free = np.arange(3, NU)
        lam = damping
        for it in range(iterations):
            Hf = H[free][:, free]
            step = np.zeros(NU)
            with clocked("host_solver"):
                step[free] = spsolve(Hf + lam * sp.diags(Hf.diagonal() + 1e-9), -g[free])
            trial = poses + step[:3 * self.n].reshape(-1, 3)
            trial_lines = lines + step[3 * self.n:].reshape(-1, 3)
            H2, g2, cost2 = self._assemble(trial, trial_lines, research=False)
            if cost2 < cost:
                poses, lines, H, g, cost, lam = trial, trial_lines, H2, g2, cost2, max(lam * 0.3, 1e-9)
            else:
                lam *= 10.0
                H, g, cost = self._assemble(poses, lines, research=False)
            history.append(cost)
            if verbose:
                print("iter %d cost %.6g lambda %.2g" % (it, cost, lam))
        self.poses = poses
        for c, con in enumerate(self.hitl):
            con.chosen_line_pose = lines[c].copy()
        return poses, history


def solve_growing_window(xy, normals, offsets, odom, window_min=1, window_max=10, iterations=4,
                         kind=_lib.NHIP_LIDAR_POINT, outlier_threshold=0.25, odom_weights=(1.0, 1.0),
                         device="cuda:0", verbose=False, backend=None, initial=None, hitl=(), loop_closures=None):
    """Solver::OptimizeOverGrowingWindow (solver.cc:339-355): for every window size from
    lidar_constraint_amount_min to _max the problem is rebuilt -- odometry factors, HITL residuals
    (AddHITLResiduals) plus fresh correspondences for all (i, j) blocks of the window, searched at the current
    estimate -- and solved.  Returns (PoseGraph of the last pass, poses)."""
    poses = np.array(odom if initial is None else initial, dtype=np.float64)
    backend = backend if backend is not None else HipBackend(device)
    if hasattr(backend, "reserve_icp"):
        with clocked("path"), clocked("path_setup"):
            backend.reserve_icp(offsets, window_max)
    pg = None
    for w in range(window_min, window_max + 1):
        pg = PoseGraph(xy, normals, offsets, odom, window=w, kind=kind, outlier_threshold=outlier_threshold,
                       odom_weights=odom_weights, device=device, initial=poses, backend=backend)
        for con in hitl:
            pg.add_hitl(con)
        if loop_closures is not None:
            pg.add_loop_closures(*loop_closures)
        poses, hist = pg.solve(iterations=iterations, verbose=verbose)
        if verbose:
            print("window %d: cost %.6g -> %.6g" % (w, hist[0], hist[-1]))
    return pg, poses


def trajectory_error(poses, truth):
    """RMS translation error after aligning the first pose (both tracks start from the same anchor)."""
    def rel(p):
        c, s = math.cos(-p[0, 2]), math.sin(-p[0, 2])
        d = p[:, :2] - p[0, :2]
        return np.stack([c * d[:, 0] - s * d[:, 1], s * d[:, 0] + c * d[:, 1]], 1)
    return float(np.sqrt(np.mean(np.sum((rel(poses) - rel(truth)) ** 2, axis=1))))
