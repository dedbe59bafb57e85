"""Host-side pose-graph driver over the GPU hot path.

The reference hands its residual blocks to Ceres (SPARSE_SCHUR, solver.cc:266-275, 335-356), which
is not installable here; this module is the small caller that lets the whole loop run on the
batched API instead:
  * ICP blocks of the sliding window (solver.cc:321-333): correspondence search (K5) and per-block
    6x6 normal equations on the GPU (nhip_corr_search_dev, nhip_resid_lidar_normal_eq_dev);
  * odometry factors (OdometryResidual, slam_residuals.h:18-40; AddOdomFactors solver.cc:370-387)
    and loop-closure constraints from the scan matcher ("Add Odometry residual using the returned
    relative transform", the TODO at solver.cc:651-660), both evaluated by nhip_resid_odometry_dev;
  * Gauss-Newton with Levenberg damping on the assembled sparse 3N x 3N system (scipy.sparse on the
    host: N poses x 3, a few thousand unknowns), first pose held constant (solver.cc:384-386).
Only the linear solve and the bookkeeping are host work; every residual, Jacobian and nearest
neighbour comes from the HIP kernels.
"""
import ctypes as C
import math

import numpy as np

from . import _lib
from ._lib import check
from .correspondence import IcpBatch, window_pairs


def compose(pose, rel):
    """pose (x, y, th) o rel (tx, ty, th): the pose of A given B and A-in-B (solver.cc:640-648)."""
    c, s = math.cos(pose[2]), math.sin(pose[2])
    return np.array([pose[0] + c * rel[0] - s * rel[1], pose[1] + s * rel[0] + c * rel[1], pose[2] + rel[2]])


class OdometryFactors:
    """Batched OdometryResidual blocks: r = (w_t (T_i + T_odom - T_j), w_r wrap(th_i + R_odom - th_j))."""

    def __init__(self, pose_i, pose_j, t_odom, r_odom, tw=1.0, rw=1.0, device="cuda:0"):
        import torch
        self.torch, self.dev = torch, torch.device(device)
        self.n = len(pose_i)
        self.pose_i = np.ascontiguousarray(pose_i, dtype=np.int32)
        self.pose_j = np.ascontiguousarray(pose_j, dtype=np.int32)
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(self.dev)
        self.d_i, self.d_j = t(self.pose_i, np.int32), t(self.pose_j, np.int32)
        self.d_t, self.d_r = t(np.reshape(t_odom, (-1, 2)), np.float32), t(r_odom, np.float32)
        self.tw, self.rw = float(tw), float(rw)
        self.d_res = torch.empty(3 * max(self.n, 1), dtype=torch.float64, device=self.dev)
        self.d_ji = torch.empty(9 * max(self.n, 1), dtype=torch.float64, device=self.dev)
        self.d_jj = torch.empty(9 * max(self.n, 1), dtype=torch.float64, device=self.dev)

    def evaluate(self, d_poses):
        if self.n == 0:
            return np.zeros((0, 3)), np.zeros((0, 3, 3)), np.zeros((0, 3, 3))
        sp = C.c_void_p(self.torch.cuda.current_stream().cuda_stream)
        check(_lib.load().nhip_resid_odometry_dev(self.d_t.data_ptr(), self.d_r.data_ptr(), self.d_i.data_ptr(),
                                                  self.d_j.data_ptr(), self.n, self.tw, self.rw, d_poses.data_ptr(),
                                                  self.d_res.data_ptr(), self.d_ji.data_ptr(), self.d_jj.data_ptr(), sp))
        return (self.d_res.cpu().numpy().reshape(-1, 3), self.d_ji.cpu().numpy().reshape(-1, 3, 3),
                self.d_jj.cpu().numpy().reshape(-1, 3, 3))


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


class PoseGraph:
    def __init__(self, xy, normals, offsets, odom, window=10, kind=_lib.NHIP_LIDAR_POINT, outlier_threshold=0.25,
                 odom_weights=(1.0, 1.0), device="cuda:0", initial=None):
        self.n = len(odom)
        self.kind = kind
        bs, bt = window_pairs(self.n, window)
        self.icp = IcpBatch(xy, normals, offsets, bs, bt, device, outlier_threshold)
        self.odo = odometry_factors_from_poses(odom, tw=odom_weights[0], rw=odom_weights[1], device=device)
        self.lc = None
        # odometry factors always come from `odom`; the estimate may start elsewhere (previous window pass)
        self.poses = np.array(odom if initial is None else initial, dtype=np.float64)

    def add_loop_closures(self, pairs_src, pairs_tgt, rel, weights=(10.0, 10.0)):
        self.lc = loop_closure_factors(self.poses, pairs_src, pairs_tgt, rel, tw=weights[0], rw=weights[1],
                                       device=str(self.icp.dev))

    def _assemble(self, poses, research):
        import scipy.sparse as sp
        N = self.n
        self.icp.set_poses(poses)
        if research:
            self.icp.search()  # correspondences are rebuilt per solve, like each window pass of the reference
        neq = self.icp.normal_equations(self.kind).cpu().numpy()
        rows, cols, vals = [], [], []
        g = np.zeros(3 * N)
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
            r, ji, jj = fac.evaluate(self.icp.d_poses)
            J = np.concatenate([ji, jj], axis=2)  # (F, 3, 6)
            Hf = np.einsum("fki,fkj->fij", J, J)
            gf = np.einsum("fki,fk->fi", J, r)
            idf = np.concatenate([3 * fac.pose_i[:, None] + np.arange(3), 3 * fac.pose_j[:, None] + np.arange(3)], axis=1)
            rows.append(np.repeat(idf, 6, axis=1).ravel())
            cols.append(np.tile(idf, (1, 6)).ravel())
            vals.append(Hf.ravel())
            np.add.at(g, idf.ravel(), gf.ravel())
            cost += 0.5 * float((r * r).sum())
        H = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(3 * N, 3 * N)).tocsc()
        return H, g, cost

    def cross_covariances(self, pairs):
        """Covariance blocks the way LCMatcher asks Ceres for them (GetCovarianceMatrix,
        lc_matcher.cc:28-46): the (source pose, target pose) cross block of (J^T J)^-1 of the current
        problem with the pose just before the earlier of the two held constant instead of pose 0;
        returns the top-left 2 x 2 (translation) of each 3 x 3 block, float32 like the reference.
        J^T J comes from the GPU's per-block normal equations; the sparse solves are host work."""
        import scipy.sparse as sp
        from scipy.sparse.linalg import splu
        H, _, _ = self._assemble(self.poses, research=False)
        H = H.tocsc()
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
        poses = self.poses.copy()
        H, g, cost = self._assemble(poses, research=True)
        history = [cost]
        free = np.arange(3, 3 * self.n)
        lam = damping
        for it in range(iterations):
            Hf = H[free][:, free]
            step = np.zeros(3 * self.n)
            step[free] = spsolve(Hf + lam * sp.diags(Hf.diagonal() + 1e-9), -g[free])
            trial = poses + step.reshape(-1, 3)
            H2, g2, cost2 = self._assemble(trial, research=False)
            if cost2 < cost:
                poses, H, g, cost, lam = trial, H2, g2, cost2, max(lam * 0.3, 1e-9)
            else:
                lam *= 10.0
                H, g, cost = self._assemble(poses, research=False)
            history.append(cost)
            if verbose:
                print("iter %d cost %.6g lambda %.2g" % (it, cost, lam))
        self.poses = poses
        return poses, history


def solve_growing_window(xy, normals, offsets, odom, window_min=1, window_max=10, iterations=4,
                         kind=_lib.NHIP_LIDAR_POINT, outlier_threshold=0.25, odom_weights=(1.0, 1.0),
                         device="cuda:0", verbose=False):
    """Solver::OptimizeOverGrowingWindow (solver.cc:339-355): for every window size from
    lidar_constraint_amount_min to _max the problem is rebuilt -- odometry factors plus fresh
    correspondences for all (i, j) blocks of the window, searched at the current estimate -- and solved.
    Returns (PoseGraph of the last pass, poses, total correspondences of the last pass)."""
    poses = np.array(odom, dtype=np.float64)
    pg = None
    for w in range(window_min, window_max + 1):
        pg = PoseGraph(xy, normals, offsets, odom, window=w, kind=kind, outlier_threshold=outlier_threshold,
                       odom_weights=odom_weights, device=device, initial=poses)
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
