// slam_residuals_hip.h -- drop-in for /root/reference/src/optimization/slam_residuals.h.
//
// Same struct names in namespace nautilus and the same static create(...) argument lists
// (slam_residuals.h:49-51, 104-108, 160-164, 206-208); what create() returns is still something
// ceres::Problem::AddResidualBlock accepts (a ceres::CostFunction subclass when Ceres is installed,
// a structurally identical local base class otherwise).  Instead of AutoDiffCostFunction running a
// templated functor on Jets per block per thread, every block registers its immutable data
// (the functors copy their vectors too, slam_residuals.h:117-120) with a process-wide
// nautilus_hip::ResidualBatcher, which evaluates ALL blocks in one pass on the MI355X
// (nhip_resid_*_dev) and from which each block's Evaluate() copies its slice.
//
// Ceres 1.14 hook (the two lines solver.cc needs):
//   options.evaluation_callback = &nautilus_hip::ResidualBatcher::Instance();   // BuildOptions()
//   nautilus_hip::ResidualBatcher::Instance().Bind(cost_fn, pose_a, pose_b);    // next to AddResidualBlock
// PrepareForEvaluation() (called by Ceres once per evaluation point, with the user's parameter
// blocks up to date) gathers the bound double[3] poses, runs the batch and downloads results.
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

class LidarCost;

// Collects the LIDAR residual blocks of one ceres::Problem build and evaluates them together.
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
      src_ptr_[k].clear();
      tgt_ptr_[k].clear();
    }
    prepared_ = false;
  }

  // Called by the create() factories: returns the block's index inside its kind.
  int Register(int kind, const std::vector<Vec2f> &sp, const std::vector<Vec2f> &tp,
               const std::vector<Vec2f> &sn, const std::vector<Vec2f> &tn) {
    std::lock_guard<std::mutex> lk(mu_);
    if (offsets_[kind].empty()) offsets_[kind].assign(1, 0);
    for (size_t i = 0; i < sp.size(); i++) {
      const float row[8] = {sp[i](0), sp[i](1), tp[i](0), tp[i](1), sn[i](0), sn[i](1), tn[i](0), tn[i](1)};
      corr_[kind].insert(corr_[kind].end(), row, row + 8);
    }
    offsets_[kind].push_back(offsets_[kind].back() + (int32_t)sp.size());
    src_ptr_[kind].push_back(nullptr);
    tgt_ptr_[kind].push_back(nullptr);
    if (batch_[kind]) { nhip_resid_batch_free(batch_[kind]); batch_[kind] = nullptr; }
    prepared_ = false;
    return (int)src_ptr_[kind].size() - 1;
  }

  // The parameter blocks handed to AddResidualBlock(cost, NULL, pose_a, pose_b) (solver.cc:280-283).
  void Bind(const CostFunctionBase *cost, double *pose_a, double *pose_b);

  // ceres::EvaluationCallback: one GPU pass per evaluation point.
  void PrepareForEvaluation(bool evaluate_jacobians, bool /*new_evaluation_point*/) override {
    std::lock_guard<std::mutex> lk(mu_);
    for (int k = 0; k < 2; k++) {
      const int32_t nb = (int32_t)src_ptr_[k].size();
      if (nb == 0) continue;
      // pose table = distinct bound pointers, in first-use order
      std::map<const double *, int32_t> index;
      std::vector<int32_t> bs(nb), bt(nb);
      std::vector<const double *> table;
      auto id = [&](const double *p) {
        if (!p) throw std::runtime_error("nautilus_hip: residual block without bound parameter blocks (call Bind)");
        auto it = index.find(p);
        if (it != index.end()) return it->second;
        index[p] = (int32_t)table.size();
        table.push_back(p);
        return (int32_t)table.size() - 1;
      };
      for (int32_t b = 0; b < nb; b++) { bs[b] = id(src_ptr_[k][b]); bt[b] = id(tgt_ptr_[k][b]); }
      if (!batch_[k] || bs != bsrc_[k] || bt != btgt_[k]) {
        if (batch_[k]) nhip_resid_batch_free(batch_[k]);
        batch_[k] = nullptr;
        Check(nhip_resid_batch_create(k, corr_[k].data(), offsets_[k].data(), bs.data(), bt.data(), nb,
                                      (int32_t)table.size(), &batch_[k]), "nhip_resid_batch_create");
        bsrc_[k] = bs;
        btgt_[k] = bt;
      }
      std::vector<double> poses(3 * table.size());
      for (size_t i = 0; i < table.size(); i++) std::memcpy(&poses[3 * i], table[i], 3 * sizeof(double));
      const size_t n = (size_t)offsets_[k].back();
      res_[k].resize(2 * n);
      if (evaluate_jacobians) { jsrc_[k].resize(6 * n); jtgt_[k].resize(6 * n); }
      Check(nhip_resid_batch_eval(batch_[k], poses.data(), res_[k].data(),
                                  evaluate_jacobians ? jsrc_[k].data() : nullptr,
                                  evaluate_jacobians ? jtgt_[k].data() : nullptr), "nhip_resid_batch_eval");
    }
    prepared_ = true;
    have_jac_ = evaluate_jacobians;
  }

  // Copies block `b`'s slice; false if the batch was not prepared (no CPU fallback).
  bool Fetch(int kind, int b, double *residuals, double **jacobians) const {
    std::lock_guard<std::mutex> lk(mu_);
    if (!prepared_) return false;
    const int32_t o = offsets_[kind][b], n = offsets_[kind][b + 1] - o;
    std::memcpy(residuals, &res_[kind][2 * (size_t)o], sizeof(double) * 2 * n);
    if (jacobians) {
      if ((jacobians[0] || jacobians[1]) && !have_jac_) return false;
      if (jacobians[0]) std::memcpy(jacobians[0], &jsrc_[kind][6 * (size_t)o], sizeof(double) * 6 * n);
      if (jacobians[1]) std::memcpy(jacobians[1], &jtgt_[kind][6 * (size_t)o], sizeof(double) * 6 * n);
    }
    return true;
  }

 private:
  ResidualBatcher() { offsets_[0].assign(1, 0); offsets_[1].assign(1, 0); }
  mutable std::mutex mu_;
  std::vector<float> corr_[2];
  std::vector<int32_t> offsets_[2], bsrc_[2], btgt_[2];
  std::vector<const double *> src_ptr_[2], tgt_ptr_[2];
  nhip_resid_batch_t *batch_[2] = {nullptr, nullptr};
  std::vector<double> res_[2], jsrc_[2], jtgt_[2];
  bool prepared_ = false, have_jac_ = false;
  friend class LidarCost;
};

// What create() returns: a cost function with two 3-vectors as parameter blocks and 2N residuals,
// exactly the shape of AutoDiffCostFunction<F, DYNAMIC, 3, 3>(f, 2N).
class LidarCost : public CostFunctionBase {
 public:
  LidarCost(int kind, int block, int n) : kind_(kind), block_(block) {
    mutable_parameter_block_sizes()->push_back(3);
    mutable_parameter_block_sizes()->push_back(3);
    set_num_residuals(2 * n);
  }
  bool Evaluate(double const *const * /*parameters*/, double *residuals, double **jacobians) const override {
    return ResidualBatcher::Instance().Fetch(kind_, block_, residuals, jacobians);
  }
  int kind() const { return kind_; }
  int block() const { return block_; }

 private:
  int kind_, block_;
};

inline void ResidualBatcher::Bind(const CostFunctionBase *cost, double *pose_a, double *pose_b) {
  const LidarCost *c = dynamic_cast<const LidarCost *>(cost);
  if (!c) return;  // OdometryResidual / PointToLineResidual blocks are not batched here
  std::lock_guard<std::mutex> lk(mu_);
  src_ptr_[c->kind()][c->block()] = pose_a;
  tgt_ptr_[c->kind()][c->block()] = pose_b;
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

struct LIDARNormalResidual {
  static nautilus_hip::LidarCost *create(const std::vector<nautilus_hip::Vec2f> &source_points,
                                         const std::vector<nautilus_hip::Vec2f> &target_points,
                                         const std::vector<nautilus_hip::Vec2f> &source_normals,
                                         const std::vector<nautilus_hip::Vec2f> &target_normals) {
    nautilus_hip::CheckSizes(source_points, target_points, source_normals, target_normals);
    const int b = nautilus_hip::ResidualBatcher::Instance().Register(NHIP_LIDAR_NORMAL, source_points, target_points,
                                                                      source_normals, target_normals);
    return new nautilus_hip::LidarCost(NHIP_LIDAR_NORMAL, b, (int)source_points.size());
  }
};

struct LIDARPointResidual {
  static nautilus_hip::LidarCost *create(const std::vector<nautilus_hip::Vec2f> &source_points,
                                         const std::vector<nautilus_hip::Vec2f> &target_points,
                                         const std::vector<nautilus_hip::Vec2f> &source_normals,
                                         const std::vector<nautilus_hip::Vec2f> &target_normals) {
    nautilus_hip::CheckSizes(source_points, target_points, source_normals, target_normals);
    const int b = nautilus_hip::ResidualBatcher::Instance().Register(NHIP_LIDAR_POINT, source_points, target_points,
                                                                      source_normals, target_normals);
    return new nautilus_hip::LidarCost(NHIP_LIDAR_POINT, b, (int)source_points.size());
  }
};

}  // namespace nautilus

#endif  // NAUTILUS_HIP_SLAM_RESIDUALS_H_
