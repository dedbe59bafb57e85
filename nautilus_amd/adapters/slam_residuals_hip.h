// slam_residuals_hip.h -- drop-in for /root/reference/src/optimization/slam_residuals.h.
//
// Same struct names in namespace nautilus and the same static create(...) argument lists
//   OdometryResidual::create(factor, translation_weight, rotation_weight)         slam_residuals.h:49-51
//   LIDARNormalResidual::create(src_pts, tgt_pts, src_normals, tgt_normals)       :104-108
//   LIDARPointResidual::create(src_pts, tgt_pts, src_normals, tgt_normals)        :160-164
//   PointToLineResidual::create(line_segment, points)                             :206-208
// so solver.cc:280-283, 291-293, 378, 521, 528 compile unchanged; what create() returns is still
// something ceres::Problem::AddResidualBlock accepts (a ceres::CostFunction subclass when Ceres is
// installed, a structurally identical local base class otherwise).  Instead of
// AutoDiffCostFunction running a templated functor on Jets per block per thread, every block
// registers its immutable data (the functors copy their vectors too, slam_residuals.h:117-120) with
// a process-wide nautilus_hip::ResidualBatcher, which evaluates ALL blocks of a family in one pass
// on the MI355X (nhip_resid_*) and from which each block's Evaluate() copies its slice.
//
// Ceres 1.14 hook (the lines solver.cc needs):
//   options.evaluation_callback = &nautilus_hip::ResidualBatcher::Instance();   // BuildOptions()
//   nautilus_hip::ResidualBatcher::Instance().Bind(cost_fn, pose_a, pose_b);    // next to AddResidualBlock
//   nautilus_hip::ResidualBatcher::Instance().Reset();                          // CeresInformation::ResetProblem
// PrepareForEvaluation() (called by Ceres once per evaluation point, with the user's parameter
// blocks up to date) gathers the bound double[3] blocks, runs the batches and downloads results.
// There is no CPU fallback: Evaluate() on a batch that has not been prepared returns false.
#ifndef NAUTILUS_HIP_SLAM_RESIDUALS_H_
#define NAUTILUS_HIP_SLAM_RESIDUALS_H_

#include <cstdint>
#include <cstring>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "CorrelativeScanMatcher.h"  // Vec2f, Check
#include "nautilus_hip.h"

#if __has_include(<ceres/ceres.h>)
#include <ceres/ceres.h>
namespace nautilus_hip {
using CostFunctionBase = ceres::CostFunction;
using EvaluationCallbackBase = ceres::EvaluationCallback;
}
#else
namespace nautilus_hip {
// Structurally identical to ceres::CostFunction (Ceres 1.14 cost_function.h) for images without Ceres.
class CostFunctionBase {
 public:
  virtual ~CostFunctionBase() {}
  virtual bool Evaluate(double const *const *parameters, double *residuals, double **jacobians) const = 0;
  const std::vector<int32_t> &parameter_block_sizes() const { return parameter_block_sizes_; }
  int num_residuals() const { return num_residuals_; }

 protected:
  std::vector<int32_t> *mutable_parameter_block_sizes() { return &parameter_block_sizes_; }
  void set_num_residuals(int n) { num_residuals_ = n; }

 private:
  std::vector<int32_t> parameter_block_sizes_;
  int num_residuals_ = 0;
};
class EvaluationCallbackBase {
 public:
  virtual ~EvaluationCallbackBase() {}
  virtual void PrepareForEvaluation(bool evaluate_jacobians, bool new_evaluation_point) = 0;
};
}  // namespace nautilus_hip
#endif

namespace nautilus_hip {

enum Family { kLidarNormal = 0, kLidarPoint = 1, kOdometry = 2, kPointToLine = 3 };

// Distinct parameter-block pointers -> dense table (first-use order).
struct PoseTable {
  std::map<const double *, int32_t> index;
  std::vector<const double *> ptrs;
  int32_t Id(const double *p) {
    if (!p) throw std::runtime_error("nautilus_hip: residual block without bound parameter blocks (call Bind)");
    auto it = index.find(p);
    if (it != index.end()) return it->second;
    index[p] = (int32_t)ptrs.size();
    ptrs.push_back(p);
    return (int32_t)ptrs.size() - 1;
  }
  std::vector<double> Gather() const {
    std::vector<double> v(3 * ptrs.size());
    for (size_t i = 0; i < ptrs.size(); i++) std::memcpy(&v[3 * i], ptrs[i], 3 * sizeof(double));
    return v;
  }
};

class BatchedCost;

// Collects the residual blocks of one ceres::Problem build and evaluates each family in one pass.
class ResidualBatcher : public EvaluationCallbackBase {
 public:
  static ResidualBatcher &Instance() {
    static ResidualBatcher b;
    return b;
  }

  // CeresInformation::ResetProblem() (data_structures.h:111-116) starts a new problem: drop everything.
  void Reset() {
    std::lock_guard<std::mutex> lk(mu_);
    for (int k = 0; k < 2; k++) {
      if (batch_[k]) nhip_resid_batch_free(batch_[k]);
      batch_[k] = nullptr;
      corr_[k].clear();
      offsets_[k].assign(1, 0);
      bsrc_[k].clear();
      btgt_[k].clear();
    }
    for (int f = 0; f < 4; f++) { pa_[f].clear(); pb_[f].clear(); res_[f].clear(); j0_[f].clear(); j1_[f].clear(); }
    t_odom_.clear(); r_odom_.clear(); tw_.clear(); rw_.clear();
    seg_.clear(); p2l_pts_.clear(); p2l_block_.clear(); p2l_off_.assign(1, 0);
    prepared_ = false;
  }

  int RegisterLidar(int kind, const std::vector<Vec2f> &sp, const std::vector<Vec2f> &tp,
                    const std::vector<Vec2f> &sn, const std::vector<Vec2f> &tn) {
    std::lock_guard<std::mutex> lk(mu_);
    for (size_t i = 0; i < sp.size(); i++) {
      const float row[8] = {sp[i](0), sp[i](1), tp[i](0), tp[i](1), sn[i](0), sn[i](1), tn[i](0), tn[i](1)};
      corr_[kind].insert(corr_[kind].end(), row, row + 8);
    }
    offsets_[kind].push_back(offsets_[kind].back() + (int32_t)sp.size());
    if (batch_[kind]) { nhip_resid_batch_free(batch_[kind]); batch_[kind] = nullptr; }
    return NewBlock(kind);
  }

  int RegisterOdometry(float tx, float ty, float rot, double tw, double rw) {
    std::lock_guard<std::mutex> lk(mu_);
    t_odom_.push_back(tx); t_odom_.push_back(ty); r_odom_.push_back(rot);
    tw_.push_back(tw); rw_.push_back(rw);
    return NewBlock(kOdometry);
  }

  int RegisterPointToLine(float x0, float y0, float x1, float y1, const std::vector<Vec2f> &points) {
    std::lock_guard<std::mutex> lk(mu_);
    const int b = (int)pa_[kPointToLine].size();
    seg_.insert(seg_.end(), {x0, y0, x1, y1});
    for (const Vec2f &p : points) { p2l_pts_.push_back(p(0)); p2l_pts_.push_back(p(1)); p2l_block_.push_back(b); }
    p2l_off_.push_back(p2l_off_.back() + (int32_t)points.size());
    return NewBlock(kPointToLine);
  }

  // The parameter blocks handed to AddResidualBlock(cost, NULL, pose_a, pose_b) (solver.cc:280-283).
  void Bind(const CostFunctionBase *cost, double *pose_a, double *pose_b);

  // ceres::EvaluationCallback: one GPU pass per family per evaluation point.
  void PrepareForEvaluation(bool evaluate_jacobians, bool /*new_evaluation_point*/) override {
    std::lock_guard<std::mutex> lk(mu_);
    const bool J = evaluate_jacobians;
    for (int k = 0; k < 2; k++) {  // LIDARNormal / LIDARPoint
      const int32_t nb = (int32_t)pa_[k].size();
      if (nb == 0) continue;
      PoseTable T;
      std::vector<int32_t> bs(nb), bt(nb);
      for (int32_t b = 0; b < nb; b++) { bs[b] = T.Id(pa_[k][b]); bt[b] = T.Id(pb_[k][b]); }
      if (!batch_[k] || bs != bsrc_[k] || bt != btgt_[k]) {
        if (batch_[k]) nhip_resid_batch_free(batch_[k]);
        batch_[k] = nullptr;
        Check(nhip_resid_batch_create(k, corr_[k].data(), offsets_[k].data(), bs.data(), bt.data(), nb,
                                      (int32_t)T.ptrs.size(), &batch_[k]), "nhip_resid_batch_create");
        bsrc_[k] = bs;
        btgt_[k] = bt;
      }
      const std::vector<double> poses = T.Gather();
      const size_t n = (size_t)offsets_[k].back();
      res_[k].resize(2 * n);
      if (J) { j0_[k].resize(6 * n); j1_[k].resize(6 * n); }
      Check(nhip_resid_batch_eval(batch_[k], poses.data(), res_[k].data(), J ? j0_[k].data() : nullptr,
                                  J ? j1_[k].data() : nullptr), "nhip_resid_batch_eval");
    }
    if (!pa_[kOdometry].empty()) {
      const int32_t n = (int32_t)pa_[kOdometry].size();
      PoseTable T;
      std::vector<int32_t> pi(n), pj(n);
      for (int32_t f = 0; f < n; f++) { pi[f] = T.Id(pa_[kOdometry][f]); pj[f] = T.Id(pb_[kOdometry][f]); }
      const std::vector<double> poses = T.Gather();
      std::vector<double> &r = res_[kOdometry], &ji = j0_[kOdometry], &jj = j1_[kOdometry];
      r.resize(3 * (size_t)n);
      if (J) { ji.resize(9 * (size_t)n); jj.resize(9 * (size_t)n); }
      // unit weights on the device, each factor's own weights applied here (w * x is one rounding either way)
      Check(nhip_resid_odometry(t_odom_.data(), r_odom_.data(), pi.data(), pj.data(), n, 1.0, 1.0, poses.data(),
                                (int32_t)T.ptrs.size(), r.data(), J ? ji.data() : nullptr, J ? jj.data() : nullptr),
            "nhip_resid_odometry");
      for (int32_t f = 0; f < n; f++) {
        const double w[3] = {tw_[f], tw_[f], rw_[f]};
        for (int row = 0; row < 3; row++) {
          r[3 * f + row] *= w[row];
          if (J) for (int c = 0; c < 3; c++) { ji[9 * f + 3 * row + c] *= w[row]; jj[9 * f + 3 * row + c] *= w[row]; }
        }
      }
    }
    if (!pa_[kPointToLine].empty()) {
      const int32_t nb = (int32_t)pa_[kPointToLine].size();
      PoseTable TP, TL;
      std::vector<int32_t> bp(nb), bl(nb);
      for (int32_t b = 0; b < nb; b++) { bp[b] = TP.Id(pa_[kPointToLine][b]); bl[b] = TL.Id(pb_[kPointToLine][b]); }
      const std::vector<double> poses = TP.Gather(), lines = TL.Gather();
      const size_t n = p2l_block_.size();
      res_[kPointToLine].resize(n);
      if (J) { j0_[kPointToLine].resize(3 * n); j1_[kPointToLine].resize(3 * n); }
      Check(nhip_resid_point_to_line(seg_.data(), p2l_pts_.data(), p2l_block_.data(), (int64_t)n, bp.data(), bl.data(),
                                     nb, poses.data(), (int32_t)TP.ptrs.size(), lines.data(), (int32_t)TL.ptrs.size(),
                                     res_[kPointToLine].data(), J ? j0_[kPointToLine].data() : nullptr,
                                     J ? j1_[kPointToLine].data() : nullptr), "nhip_resid_point_to_line");
    }
    prepared_ = true;
    have_jac_ = evaluate_jacobians;
  }

  // Copies block `b`'s slice; false if the batch was not prepared (no CPU fallback).
  bool Fetch(int family, int b, double *residuals, double **jacobians) const {
    std::lock_guard<std::mutex> lk(mu_);
    if (!prepared_) return false;
    size_t r_off, r_n;  // residual rows of this block
    if (family <= kLidarPoint) { r_off = 2 * (size_t)offsets_[family][b]; r_n = 2 * (size_t)(offsets_[family][b + 1] - offsets_[family][b]); }
    else if (family == kOdometry) { r_off = 3 * (size_t)b; r_n = 3; }
    else { r_off = (size_t)p2l_off_[b]; r_n = (size_t)(p2l_off_[b + 1] - p2l_off_[b]); }
    std::memcpy(residuals, &res_[family][r_off], sizeof(double) * r_n);
    if (jacobians) {
      if ((jacobians[0] || jacobians[1]) && !have_jac_) return false;
      if (jacobians[0]) std::memcpy(jacobians[0], &j0_[family][3 * r_off], sizeof(double) * 3 * r_n);
      if (jacobians[1]) std::memcpy(jacobians[1], &j1_[family][3 * r_off], sizeof(double) * 3 * r_n);
    }
    return true;
  }

 private:
  ResidualBatcher() { offsets_[0].assign(1, 0); offsets_[1].assign(1, 0); p2l_off_.assign(1, 0); }
  int NewBlock(int family) {
    pa_[family].push_back(nullptr);
    pb_[family].push_back(nullptr);
    prepared_ = false;
    return (int)pa_[family].size() - 1;
  }
  mutable std::mutex mu_;
  // LIDAR families
  std::vector<float> corr_[2];
  std::vector<int32_t> offsets_[2], bsrc_[2], btgt_[2];
  nhip_resid_batch_t *batch_[2] = {nullptr, nullptr};
  // odometry
  std::vector<float> t_odom_, r_odom_;
  std::vector<double> tw_, rw_;
  // point-to-line
  std::vector<float> seg_, p2l_pts_;
  std::vector<int32_t> p2l_block_, p2l_off_;
  // per family: bound parameter blocks and the last evaluation
  std::vector<const double *> pa_[4], pb_[4];
  std::vector<double> res_[4], j0_[4], j1_[4];
  bool prepared_ = false, have_jac_ = false;
  friend class BatchedCost;
};

// What create() returns: two 3-vectors as parameter blocks and `num_residuals` residuals, exactly
// the shape of AutoDiffCostFunction<F, DYNAMIC | 3, 3, 3>.
class BatchedCost : public CostFunctionBase {
 public:
  BatchedCost(int family, int block, int num_residuals) : family_(family), block_(block) {
    mutable_parameter_block_sizes()->push_back(3);
    mutable_parameter_block_sizes()->push_back(3);
    set_num_residuals(num_residuals);
  }
  bool Evaluate(double const *const * /*parameters*/, double *residuals, double **jacobians) const override {
    return ResidualBatcher::Instance().Fetch(family_, block_, residuals, jacobians);
  }
  int family() const { return family_; }
  int block() const { return block_; }

 private:
  int family_, block_;
};
using LidarCost = BatchedCost;

inline void ResidualBatcher::Bind(const CostFunctionBase *cost, double *pose_a, double *pose_b) {
  const BatchedCost *c = dynamic_cast<const BatchedCost *>(cost);
  if (!c) return;  // not one of ours
  std::lock_guard<std::mutex> lk(mu_);
  pa_[c->family()][c->block()] = pose_a;
  pb_[c->family()][c->block()] = pose_b;
  prepared_ = false;
}

inline void CheckSizes(const std::vector<Vec2f> &sp, const std::vector<Vec2f> &tp, const std::vector<Vec2f> &sn,
                       const std::vector<Vec2f> &tn) {
  // CHECK_EQ x3 (slam_residuals.h:99-101) and CHECK_GT(size, 0) (:109)
  if (sp.size() != tp.size() || tp.size() != tn.size() || sn.size() != tn.size())
    throw std::invalid_argument("correspondence vectors differ in length");
  if (sp.empty()) throw std::invalid_argument("empty correspondence set");
}

}  // namespace nautilus_hip

namespace nautilus {

struct OdometryResidual {
  // Factor: slam_types::OdometryFactor2D (slam_types.h:102-120) or anything with .translation(i), .rotation
  template <class Factor>
  static nautilus_hip::BatchedCost *create(const Factor &factor, double translation_weight, double rotation_weight) {
    const int b = nautilus_hip::ResidualBatcher::Instance().RegisterOdometry(
        (float)factor.translation(0), (float)factor.translation(1), (float)factor.rotation, translation_weight,
        rotation_weight);
    return new nautilus_hip::BatchedCost(nautilus_hip::kOdometry, b, 3);
  }
};

struct LIDARNormalResidual {
  static nautilus_hip::BatchedCost *create(const std::vector<nautilus_hip::Vec2f> &source_points,
                                           const std::vector<nautilus_hip::Vec2f> &target_points,
                                           const std::vector<nautilus_hip::Vec2f> &source_normals,
                                           const std::vector<nautilus_hip::Vec2f> &target_normals) {
    nautilus_hip::CheckSizes(source_points, target_points, source_normals, target_normals);
    const int b = nautilus_hip::ResidualBatcher::Instance().RegisterLidar(NHIP_LIDAR_NORMAL, source_points, target_points,
                                                                           source_normals, target_normals);
    return new nautilus_hip::BatchedCost(nautilus_hip::kLidarNormal, b, 2 * (int)source_points.size());
  }
};

struct LIDARPointResidual {
  static nautilus_hip::BatchedCost *create(const std::vector<nautilus_hip::Vec2f> &source_points,
                                           const std::vector<nautilus_hip::Vec2f> &target_points,
                                           const std::vector<nautilus_hip::Vec2f> &source_normals,
                                           const std::vector<nautilus_hip::Vec2f> &target_normals) {
    nautilus_hip::CheckSizes(source_points, target_points, source_normals, target_normals);
    const int b = nautilus_hip::ResidualBatcher::Instance().RegisterLidar(NHIP_LIDAR_POINT, source_points, target_points,
                                                                           source_normals, target_normals);
    return new nautilus_hip::BatchedCost(nautilus_hip::kLidarPoint, b, 2 * (int)source_points.size());
  }
};

struct PointToLineResidual {
  // Segment: LineSegment<float> (data_structures.h:13-32) or anything with .start(i), .end(i)
  template <class Segment>
  static nautilus_hip::BatchedCost *create(const Segment &line_segment, const std::vector<nautilus_hip::Vec2f> points) {
    const int b = nautilus_hip::ResidualBatcher::Instance().RegisterPointToLine(
        (float)line_segment.start(0), (float)line_segment.start(1), (float)line_segment.end(0),
        (float)line_segment.end(1), points);
    return new nautilus_hip::BatchedCost(nautilus_hip::kPointToLine, b, (int)points.size());
  }
};

}  // namespace nautilus

#endif  // NAUTILUS_HIP_SLAM_RESIDUALS_H_
