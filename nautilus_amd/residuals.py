"""Host-side mirror of nautilus's cost-functor factories, over the C ABI.

Mirrors /root/reference/src/optimization/slam_residuals.h:
  LIDARNormalResidual.create(source_points, target_points, source_normals, target_normals)  :104-115
  LIDARPointResidual.create(...)                                                            :160-171
  PointToLineResidual.create(line_segment, points)                                          :206-212
  OdometryResidual.create(factor, translation_weight, rotation_weight)                      :49-55
Each returned block keeps `num_residuals` and the parameter-block layout of the Ceres cost
function it stands for; evaluation is batched: a ResidualProblem collects the blocks the way
ceres::Problem::AddResidualBlock does (solver.cc:280-283, 291-293, 378, 521, 528) and
evaluates all of them in one pass on the GPU (no CPU fallback).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, ptr, NHIP_LIDAR_NORMAL, NHIP_LIDAR_POINT


def _pts(a):
    return np.ascontiguousarray(a, dtype=np.float32).reshape(-1, 2)


class _LidarBlock:
    kind = None

    def __init__(self, source_points, target_points, source_normals, target_normals):
        sp, tp, sn, tn = _pts(source_points), _pts(target_points), _pts(source_normals), _pts(target_normals)
        # CHECK_EQ sizes (slam_residuals.h:99-101) and CHECK_GT(size, 0) (:109)
        if not (len(sp) == len(tp) == len(sn) == len(tn)):
            raise ValueError("correspondence vectors differ in length")
        if len(sp) == 0:
            raise ValueError("empty correspondence set")
        self.corr = np.concatenate([sp, tp, sn, tn], axis=1)  # (N, 8)
        self.num_residuals = 2 * len(sp)

    @classmethod
    def create(cls, source_points, target_points, source_normals, target_normals):
        return cls(source_points, target_points, source_normals, target_normals)


class LIDARNormalResidual(_LidarBlock):
    kind = NHIP_LIDAR_NORMAL


class LIDARPointResidual(_LidarBlock):
    kind = NHIP_LIDAR_POINT


class PointToLineResidual:
    def __init__(self, line_segment, points):
        self.segment = np.ascontiguousarray(line_segment, dtype=np.float32).reshape(4)  # x0 y0 x1 y1
        self.points = _pts(points)
        self.num_residuals = len(self.points)

    @classmethod
    def create(cls, line_segment, points):
        return cls(line_segment, points)


class OdometryResidual:
    num_residuals = 3

    def __init__(self, translation, rotation, translation_weight, rotation_weight):
        self.t_odom = np.asarray(translation, dtype=np.float32).reshape(2)
        self.r_odom = np.float32(rotation)
        self.translation_weight = float(translation_weight)
        self.rotation_weight = float(rotation_weight)

    @classmethod
    def create(cls, factor, translation_weight, rotation_weight):
        """factor: (pose_i, pose_j, translation (2,), rotation) like OdometryFactor2D (slam_types.h:102-120)."""
        return cls(factor[2], factor[3], translation_weight, rotation_weight)


class LidarResidualBatch:
    """All LIDAR residual blocks of one problem build: device-resident, immutable."""

    def __init__(self, kind, blocks, block_src, block_tgt, n_poses):
        self.kind = kind
        self.n_blocks = len(blocks)
        self.block_offsets = np.zeros(self.n_blocks + 1, dtype=np.int32)
        for i, b in enumerate(blocks):
            self.block_offsets[i + 1] = self.block_offsets[i] + len(b)
        self.n_corr = int(self.block_offsets[-1])
        self.corr = np.ascontiguousarray(np.concatenate(blocks, axis=0) if blocks else np.zeros((0, 8)),
                                         dtype=np.float32)
        self.block_src = np.ascontiguousarray(block_src, dtype=np.int32)
        self.block_tgt = np.ascontiguousarray(block_tgt, dtype=np.int32)
        self.n_poses = int(n_poses)
        self._h = C.c_void_p()
        check(_lib.load().nhip_resid_batch_create(kind, ptr(self.corr), ptr(self.block_offsets),
                                                  ptr(self.block_src), ptr(self.block_tgt),
                                                  self.n_blocks, self.n_poses, C.byref(self._h)))

    def evaluate(self, poses, want_jac_src=True, want_jac_tgt=True):
        """poses (n_poses, 3) float64 -> residuals (2*n_corr), jac_src, jac_tgt ((2*n_corr, 3) or None)."""
        poses = np.ascontiguousarray(poses, dtype=np.float64).reshape(self.n_poses, 3)
        res = np.empty(2 * self.n_corr, dtype=np.float64)
        js = np.empty((2 * self.n_corr, 3), dtype=np.float64) if want_jac_src else None
        jt = np.empty((2 * self.n_corr, 3), dtype=np.float64) if want_jac_tgt else None
        check(_lib.load().nhip_resid_batch_eval(self._h, ptr(poses), ptr(res), ptr(js), ptr(jt)))
        return res, js, jt

    def evaluate_q(self, poses, rebuild=True):
        """The smallest form over PCIe (nhip_resid_batch_eval_q): residuals, q = S2T p_s per correspondence and 8 constants
        per block come down -- 32 bytes per correspondence -- and (rebuild) both Jacobians are rebuilt on the host, block by
        block, by the library's own nhip_resid_jacobians_from_q (what the C++ adapter does while it copies a block's slice).
        Returns (residuals, jac_src, jac_tgt) like evaluate(), or (residuals, q, block_consts) with rebuild=False."""
        lib = _lib.load()
        poses = np.ascontiguousarray(poses, dtype=np.float64).reshape(self.n_poses, 3)
        res = np.empty(2 * self.n_corr, dtype=np.float64)
        q = np.empty((self.n_corr, 2), dtype=np.float64)
        consts = np.empty((self.n_blocks, 8), dtype=np.float64)
        check(lib.nhip_resid_batch_eval_q(self._h, ptr(poses), ptr(res), ptr(q), ptr(consts)))
        if not rebuild:
            return res, q, consts
        js = np.empty((2 * self.n_corr, 3), dtype=np.float64)
        jt = np.empty((2 * self.n_corr, 3), dtype=np.float64)
        for b in range(self.n_blocks):
            o, e = int(self.block_offsets[b]), int(self.block_offsets[b + 1])
            check(lib.nhip_resid_jacobians_from_q(self.kind, ptr(self.corr[o:e]), ptr(q[o:e]), ptr(consts[b]), e - o,
                                                  ptr(js[2 * o:2 * e]), ptr(jt[2 * o:2 * e])))
        return res, js, jt

    def close(self):
        if self._h:
            _lib.load().nhip_resid_batch_free(self._h)
            self._h = C.c_void_p()

    __del__ = close


class ResidualProblem:
    """Collects residual blocks like ceres::Problem::AddResidualBlock(cost, NULL, pose_a, pose_b)."""

    def __init__(self, n_poses):
        self.n_poses = int(n_poses)
        self._lidar = {NHIP_LIDAR_NORMAL: ([], [], []), NHIP_LIDAR_POINT: ([], [], [])}
        self._batches = None

    def AddResidualBlock(self, cost, pose_a_index, pose_b_index):
        if not isinstance(cost, _LidarBlock):
            raise TypeError("ResidualProblem batches LIDAR blocks; use the *_dev entry points for others")
        blocks, src, tgt = self._lidar[cost.kind]
        blocks.append(cost.corr)
        src.append(int(pose_a_index))
        tgt.append(int(pose_b_index))
        self._batches = None

    def build(self):
        self._batches = {k: LidarResidualBatch(k, b, s, t, self.n_poses)
                         for k, (b, s, t) in self._lidar.items() if b}
        return self._batches

    def Evaluate(self, poses, jacobians=True):
        if self._batches is None:
            self.build()
        return {k: b.evaluate(poses, jacobians, jacobians) for k, b in self._batches.items()}
