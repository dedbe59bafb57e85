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
// a process-wide nautilus_hip::ResidualBatcher, which evaluates ALL live blocks of a family in one pass
// on the MI355X (nhip_resid_*) and from which each block's Evaluate() copies its slice.
//
// Lifetime follows Ceres' ownership (data_structures.h:111-116): ceres::Problem owns the cost functions it was
// given and deletes them when it is destroyed -- CeresInformation::ResetProblem() does that ten times per SolveSLAM
// (solver.cc:335-356).  A BatchedCost registers its block when it is created and UNREGISTERS it in its destructor,
// so a rebuilt problem leaves nothing behind (no Reset() call in the reference's sources), and the blocks of two
// problems that exist at the same time simply coexist: an evaluation pass covers every live block, each at the
// current values of its own parameter blocks.
//
// Ceres 1.14 hook: the ONE line the batched evaluation needs is
//   options.evaluation_callback = &nautilus_hip::ResidualBatcher::Instance();   // BuildOptions(), solver.cc:266-275
// PrepareForEvaluation() (called by Ceres once per evaluation point, with the user's parameter blocks up to date)
// gathers the double[3] blocks, runs the batches and downloads the results into pinned memory; Evaluate() copies
// its slice without taking a lock.  Parameter blocks are learned from each block's first Evaluate() (Ceres
// evaluates at the user's own blocks when the callback is set) or, without that first single-block pass, from
//   nautilus_hip::AddResidualBlock(problem, cost, NULL, pose_a, pose_b);        // instead of problem.AddResidualBlock
// Evaluate() at parameter values other than the prepared ones -- ceres::Problem::Evaluate, Covariance::Compute
// (LCMatcher::GetCovarianceMatrix, lc_matcher.cc:28-46), GradientChecker, or no callback at all -- is detected by
// comparing the values and served by a single-block evaluation on the GPU at the parameters passed in.
// There is no CPU fallback.
#ifndef NAUTILUS_HIP_SLAM_RESIDUALS_H_
#define NAUTILUS_HIP_SLAM_RESIDUALS_H_

#include <atomic>
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

// Page-locked array of doubles (nhip_host_alloc): the results cross PCIe every evaluation.
struct PinnedDoubles {
  double *p = nullptr;
  size_t cap = 0;
  void reserve(size_t n) {
    if (n <= cap) return;
    release();
    Check(nhip_host_alloc(sizeof(double) * n, reinterpret_cast<void **>(&p)), "nhip_host_alloc");
    cap = n;
  }
  void release() {
    if (p) nhip_host_free(p);
    p = nullptr;
    cap = 0;
  }
  double *data() { return p; }
  const double *data() const { return p; }
};

class ResidualBatcher;

// One residual block: what its create() call copied, the parameter blocks it was bound to, and where its rows sit in
// the arrays of the last evaluation pass.  Owned by the BatchedCost that create() returned.
struct Block {
  int family = 0;
  int32_t live_index = -1;  // position in the batcher's list of live blocks of the family
  int32_t slot = -1;        // position in the family's compiled arrays; -1: registered after they were compiled
  int32_t rows = 0;         // residual rows: 2 N, 3, N
  int64_t row0 = 0;         // first row in the family's result arrays (valid while slot >= 0)
  const double *pa = nullptr, *pb = nullptr;
  std::vector<float> data;  // LIDAR: 8 floats per correspondence; odometry: tx ty rot; point-to-line: x0 y0 x1 y1, points
  double tw = 1.0, rw = 1.0;  // odometry weights
};

// Collects the live residual blocks and evaluates each family in one pass.
//
// Threading: blocks are created, bound and destroyed while a problem is built or torn down (one thread; never while
// Evaluate() calls are in flight -- Ceres does not).  PrepareForEvaluation runs on one thread with no Evaluate() in
// flight (the ceres::EvaluationCallback contract); afterwards Evaluate() may be called from any number of threads at
// once: the fast path only READS the prepared arrays and takes no lock.  An Evaluate() that arrives at parameter
// values other than the prepared ones (Problem::Evaluate, Covariance::Compute, GradientChecker, a solver without the
// callback), for a block whose parameter blocks are not known yet, or for a block created after the last pass is
// served by a single-block evaluation on the GPU at exactly the parameters passed in (slow path, serialised by a
// mutex) -- never by stale values, never by the CPU.
class ResidualBatcher : public EvaluationCallbackBase {
 public:
  static ResidualBatcher &Instance() {
    static ResidualBatcher *b = new ResidualBatcher();  // never destroyed: pinned memory must not outlive the HIP runtime
    return *b;
  }

  // Releases the device batches and pinned result buffers (they are rebuilt by the next pass).  Live blocks stay:
  // they belong to their cost functions.  Never needed for correctness -- kept for hosts that want the memory back.
  void Reset() {
    std::lock_guard<std::mutex> lk(mu_);
    prepared_.store(false, std::memory_order_release);
    for (int k = 0; k < 2; k++) {
      if (batch_[k]) nhip_resid_batch_free(batch_[k]);
      batch_[k] = nullptr;
      jtt_[k].release();
    }
    for (int f = 0; f < 4; f++) {
      dirty_[f] = true;
      for (Block *b : live_[f]) b->slot = -1;
      res_[f].release(); j0_[f].release(); j1_[f].release();
    }
  }

  Block *RegisterLidar(int kind, const std::vector<Vec2f> &sp, const std::vector<Vec2f> &tp,
                       const std::vector<Vec2f> &sn, const std::vector<Vec2f> &tn) {
    Block *b = new Block();
    b->family = kind;
    b->rows = 2 * (int32_t)sp.size();
    b->data.reserve(8 * sp.size());
    for (size_t i = 0; i < sp.size(); i++) {
      const float row[8] = {sp[i](0), sp[i](1), tp[i](0), tp[i](1), sn[i](0), sn[i](1), tn[i](0), tn[i](1)};
      b->data.insert(b->data.end(), row, row + 8);
    }
    return Add(b);
  }

  Block *RegisterOdometry(float tx, float ty, float rot, double tw, double rw) {
    Block *b = new Block();
    b->family = kOdometry;
    b->rows = 3;
    b->data = {tx, ty, rot};
    b->tw = tw;
    b->rw = rw;
    return Add(b);
  }

  Block *RegisterPointToLine(float x0, float y0, float x1, float y1, const std::vector<Vec2f> &points) {
    Block *b = new Block();
    b->family = kPointToLine;
    b->rows = (int32_t)points.size();
    b->data = {x0, y0, x1, y1};
    for (const Vec2f &p : points) { b->data.push_back(p(0)); b->data.push_back(p(1)); }
    return Add(b);
  }

  // ~BatchedCost: the block leaves the live set (O(1): the last live block takes its place) and is freed.
  void Unregister(Block *b) {
    std::lock_guard<std::mutex> lk(mu_);
    std::vector<Block *> &L = live_[b->family];
    L[b->live_index] = L.back();
    L[b->live_index]->live_index = b->live_index;
    L.pop_back();
    dirty_[b->family] = true;
    prepared_.store(false, std::memory_order_release);
    delete b;
  }

  // The parameter blocks handed to AddResidualBlock(cost, NULL, pose_a, pose_b) (solver.cc:280-283).  Optional:
  // an unbound block learns them from its first Evaluate() (with ceres::EvaluationCallback set, Ceres evaluates at
  // the user's own parameter blocks), at the price of one single-block GPU evaluation per block on that first pass.
  // nautilus_hip::AddResidualBlock() below binds and adds in one line.
  void Bind(const CostFunctionBase *cost, double *pose_a, double *pose_b);

  // ceres::EvaluationCallback: one GPU pass per family per evaluation point, over every live block.
  void PrepareForEvaluation(bool evaluate_jacobians, bool /*new_evaluation_point*/) override {
    std::lock_guard<std::mutex> lk(mu_);
    prepared_.store(false, std::memory_order_release);
    const bool J = evaluate_jacobians;
    for (int k = 0; k < 2; k++) {  // LIDARNormal / LIDARPoint
      if (live_[k].empty()) continue;
      PoseTable T;
      CompileLidar(k, &T);
      const std::vector<double> poses = T.Gather();
      poses_a_[k] = poses;
      const size_t n = (size_t)rows_total_[k] / 2;
      res_[k].reserve(2 * n);
      if (J) {
        // the smallest form over PCIe: residuals, q = S2T p_s per correspondence and 8 constants per block (32 bytes per
        // correspondence instead of 80); a block's Jacobian rows are rebuilt from them and the block's own points and
        // normals while its slice is copied (CopyPrepared -> nhip_resid_jacobians_from_q)
        jtt_[k].reserve(2 * n);
        consts_[k].resize(8 * live_[k].size());
        Check(nhip_resid_batch_eval_q(batch_[k], poses.data(), res_[k].data(), jtt_[k].data(), consts_[k].data()),
              "nhip_resid_batch_eval_q");
      } else {
        Check(nhip_resid_batch_eval(batch_[k], poses.data(), res_[k].data(), nullptr, nullptr), "nhip_resid_batch_eval");
      }
    }
    if (!live_[kOdometry].empty()) {
      const std::vector<Block *> &L = live_[kOdometry];
      const int32_t n = (int32_t)L.size();
      PoseTable T;
      AssignSlots(kOdometry);
      BuildIndices(kOdometry, &T, nullptr);
      t_odom_.resize(2 * (size_t)n); r_odom_.resize(n);
      for (int32_t f = 0; f < n; f++) { t_odom_[2 * f] = L[f]->data[0]; t_odom_[2 * f + 1] = L[f]->data[1]; r_odom_[f] = L[f]->data[2]; }
      const std::vector<double> poses = T.Gather();
      poses_a_[kOdometry] = poses;
      res_[kOdometry].reserve(3 * (size_t)n);
      if (J) { j0_[kOdometry].reserve(9 * (size_t)n); j1_[kOdometry].reserve(9 * (size_t)n); }
      double *r = res_[kOdometry].data(), *ji = j0_[kOdometry].data(), *jj = j1_[kOdometry].data();
      // unit weights on the device, each factor's own weights applied here (w * x is one rounding either way)
      Check(nhip_resid_odometry(t_odom_.data(), r_odom_.data(), ia_[kOdometry].data(), ib_[kOdometry].data(), n, 1.0, 1.0,
                                poses.data(), (int32_t)T.ptrs.size(), r, J ? ji : nullptr, J ? jj : nullptr),
            "nhip_resid_odometry");
      for (int32_t f = 0; f < n; f++) ApplyOdometryWeights(*L[f], r + 3 * f, J ? ji + 9 * f : nullptr, J ? jj + 9 * f : nullptr);
    }
    if (!live_[kPointToLine].empty()) {
      const std::vector<Block *> &L = live_[kPointToLine];
      const int32_t nb = (int32_t)L.size();
      PoseTable TP, TL;
      AssignSlots(kPointToLine);
      BuildIndices(kPointToLine, &TP, &TL);
      seg_.clear(); p2l_pts_.clear(); p2l_block_.clear();
      for (int32_t b = 0; b < nb; b++) {
        seg_.insert(seg_.end(), L[b]->data.begin(), L[b]->data.begin() + 4);
        p2l_pts_.insert(p2l_pts_.end(), L[b]->data.begin() + 4, L[b]->data.end());
        p2l_block_.insert(p2l_block_.end(), (size_t)L[b]->rows, b);
      }
      const std::vector<double> poses = TP.Gather(), lines = TL.Gather();
      poses_a_[kPointToLine] = poses;
      poses_b_[kPointToLine] = lines;
      const size_t n = p2l_block_.size();
      res_[kPointToLine].reserve(n);
      if (J) { j0_[kPointToLine].reserve(3 * n); j1_[kPointToLine].reserve(3 * n); }
      Check(nhip_resid_point_to_line(seg_.data(), p2l_pts_.data(), p2l_block_.data(), (int64_t)n, ia_[kPointToLine].data(),
                                     ib_[kPointToLine].data(), nb, poses.data(), (int32_t)TP.ptrs.size(), lines.data(),
                                     (int32_t)TL.ptrs.size(), res_[kPointToLine].data(),
                                     J ? j0_[kPointToLine].data() : nullptr, J ? j1_[kPointToLine].data() : nullptr),
            "nhip_resid_point_to_line");
    }
    have_jac_ = evaluate_jacobians;
    prepared_.store(true, std::memory_order_release);
  }

  // The block's residuals (and Jacobians) at `parameters`.  Fast path: the prepared arrays, no lock.
  bool Fetch(Block *b, double const *const *parameters, double *residuals, double **jacobians) {
    const bool want_j = jacobians && (jacobians[0] || jacobians[1]);
    if (prepared_.load(std::memory_order_acquire) && b->slot >= 0 && b->pa && (!want_j || have_jac_) &&
        SameParameters(*b, parameters)) {
      CopyPrepared(*b, residuals, jacobians);
      return true;
    }
    return EvaluateOne(b, parameters, residuals, jacobians);
  }

  long slow_path_calls() const { return slow_calls_; }
  // (tests) live blocks of a family, rows of its last compiled batch, device batches built so far
  size_t live_blocks(int family) const { return live_[family].size(); }
  int64_t compiled_rows(int family) const { return rows_total_[family]; }
  long batches_built() const { return batches_built_; }

 private:
  ResidualBatcher() {}
  Block *Add(Block *b) {
    std::lock_guard<std::mutex> lk(mu_);
    b->live_index = (int32_t)live_[b->family].size();
    live_[b->family].push_back(b);
    dirty_[b->family] = true;
    return b;  // (the prepared arrays of the OTHER blocks stay valid: this one has slot -1 until the next pass)
  }
  // unbound blocks read a dummy pose until their first Evaluate() binds them (their prepared values are never used)
  static const double *Bound(const double *p) { static const double zero[3] = {0, 0, 0}; return p ? p : zero; }
  // slot = position in the live list; rows laid out in that order
  void AssignSlots(int f) {
    int64_t row = 0;
    for (size_t i = 0; i < live_[f].size(); i++) {
      live_[f][i]->slot = (int32_t)i;
      live_[f][i]->row0 = row;
      row += live_[f][i]->rows;
    }
    rows_total_[f] = row;
    dirty_[f] = false;
  }
  // indices of the blocks' parameter blocks into dense pose tables (TB null: both blocks in TA)
  void BuildIndices(int f, PoseTable *TA, PoseTable *TB) {
    const size_t nb = live_[f].size();
    ia_[f].resize(nb); ib_[f].resize(nb);
    for (size_t b = 0; b < nb; b++) {
      ia_[f][b] = TA->Id(Bound(live_[f][b]->pa));
      ib_[f][b] = (TB ? TB : TA)->Id(Bound(live_[f][b]->pb));
    }
  }
  // The LIDAR family's device batch: rebuilt when the set of live blocks or their parameter-block pattern changed.
  void CompileLidar(int k, PoseTable *T) {
    const bool was_dirty = dirty_[k];
    if (was_dirty) AssignSlots(k);
    BuildIndices(k, T, nullptr);
    if (!batch_[k] || was_dirty || ia_[k] != bsrc_[k] || ib_[k] != btgt_[k]) {
      if (batch_[k]) nhip_resid_batch_free(batch_[k]);
      batch_[k] = nullptr;
      const std::vector<Block *> &L = live_[k];
      std::vector<float> corr;
      corr.reserve(4 * (size_t)rows_total_[k]);
      std::vector<int32_t> offsets(1, 0);
      for (const Block *b : L) {
        corr.insert(corr.end(), b->data.begin(), b->data.end());
        offsets.push_back(offsets.back() + b->rows / 2);
      }
      Check(nhip_resid_batch_create(k, corr.data(), offsets.data(), ia_[k].data(), ib_[k].data(), (int32_t)L.size(),
                                    (int32_t)T->ptrs.size(), &batch_[k]), "nhip_resid_batch_create");
      bsrc_[k] = ia_[k];
      btgt_[k] = ib_[k];
      batches_built_++;
    }
  }
  static void ApplyOdometryWeights(const Block &b, double *r, double *ji, double *jj) {
    const double w[3] = {b.tw, b.tw, b.rw};
    for (int row = 0; row < 3; row++) {
      r[row] *= w[row];
      if (ji) for (int c = 0; c < 3; c++) ji[3 * row + c] *= w[row];
      if (jj) for (int c = 0; c < 3; c++) jj[3 * row + c] *= w[row];
    }
  }
  // Were the prepared values computed at exactly these parameter values?
  bool SameParameters(const Block &b, double const *const *parameters) const {
    const int f = b.family;
    const double *a = &poses_a_[f][3 * (size_t)ia_[f][b.slot]];
    const double *c = (f == kPointToLine) ? &poses_b_[f][3 * (size_t)ib_[f][b.slot]] : &poses_a_[f][3 * (size_t)ib_[f][b.slot]];
    return std::memcmp(a, parameters[0], 3 * sizeof(double)) == 0 && std::memcmp(c, parameters[1], 3 * sizeof(double)) == 0;
  }
  void CopyPrepared(const Block &b, double *residuals, double **jacobians) const {
    const int family = b.family;
    const size_t r_off = (size_t)b.row0, r_n = (size_t)b.rows;
    std::memcpy(residuals, res_[family].data() + r_off, sizeof(double) * r_n);
    if (!jacobians) return;
    if (family <= kLidarPoint) {
      // both Jacobians of the block from its q values (2 doubles per correspondence = per two rows), the block's 8
      // constants and its own points / normals: the closed forms of the device kernel, evaluated here
      Check(nhip_resid_jacobians_from_q(family, b.data.data(), jtt_[family].data() + r_off, consts_[family].data() + 8 * (size_t)b.slot,
                                        (int64_t)(r_n / 2), jacobians[0], jacobians[1]), "nhip_resid_jacobians_from_q");
      return;
    }
    const double *js = j0_[family].data() + 3 * r_off;
    if (jacobians[0]) std::memcpy(jacobians[0], js, sizeof(double) * 3 * r_n);
    if (jacobians[1]) std::memcpy(jacobians[1], j1_[family].data() + 3 * r_off, sizeof(double) * 3 * r_n);
  }
  // Slow path: this block alone, on the GPU, at the parameters passed in.
  bool EvaluateOne(Block *b, double const *const *parameters, double *residuals, double **jacobians) {
    std::lock_guard<std::mutex> lk(mu_);
    slow_calls_++;
    if (!b->pa) {  // first sight of this block's parameter blocks
      b->pa = parameters[0];
      b->pb = parameters[1];
      prepared_.store(false, std::memory_order_release);  // the next PrepareForEvaluation batches it
    }
    double *j0 = jacobians ? jacobians[0] : nullptr, *j1 = jacobians ? jacobians[1] : nullptr;
    const int family = b->family;
    if (family <= kLidarPoint) {
      if (!batch_[family] || dirty_[family] || b->slot < 0) {  // (a block newer than the device batch: rebuild it)
        PoseTable T;
        dirty_[family] = true;
        CompileLidar(family, &T);
        prepared_.store(false, std::memory_order_release);  // rows moved
      }
      return nhip_resid_batch_eval_block(batch_[family], b->slot, parameters[0], parameters[1], residuals, j0, j1) == NHIP_OK;
    }
    if (family == kOdometry) {
      double two[6];
      std::memcpy(two, parameters[0], 24); std::memcpy(two + 3, parameters[1], 24);
      const int32_t i0 = 0, i1 = 1;
      if (nhip_resid_odometry(&b->data[0], &b->data[2], &i0, &i1, 1, 1.0, 1.0, two, 2, residuals, j0, j1) != NHIP_OK) return false;
      ApplyOdometryWeights(*b, residuals, j0, j1);
      return true;
    }
    const int32_t zero = 0;
    std::vector<int32_t> blk((size_t)b->rows, 0);
    return nhip_resid_point_to_line(&b->data[0], b->data.data() + 4, blk.data(), b->rows, &zero, &zero, 1,
                                    parameters[0], 1, parameters[1], 1, residuals, j0, j1) == NHIP_OK;
  }

  std::mutex mu_;
  std::atomic<bool> prepared_{false};
  bool have_jac_ = false;
  long slow_calls_ = 0, batches_built_ = 0;
  std::vector<Block *> live_[4];
  bool dirty_[4] = {false, false, false, false};
  int64_t rows_total_[4] = {0, 0, 0, 0};
  // LIDAR families: the device batch and the parameter-block pattern it was built for
  std::vector<int32_t> bsrc_[2], btgt_[2];
  nhip_resid_batch_t *batch_[2] = {nullptr, nullptr};
  PinnedDoubles jtt_[2];  // q = S2T p_s, 2 doubles per correspondence (nhip_resid_batch_eval_q)
  std::vector<double> consts_[2];  // per block of the LIDAR families: S2T's linear part and translation, Linv's first row
  // staging of the two small families (rebuilt per pass: a few KB)
  std::vector<float> t_odom_, r_odom_, seg_, p2l_pts_;
  std::vector<int32_t> p2l_block_;
  // per family: indices of the blocks' parameter blocks into the last gathered pose table(s), the last evaluation
  std::vector<int32_t> ia_[4], ib_[4];
  std::vector<double> poses_a_[4], poses_b_[4];
  PinnedDoubles res_[4], j0_[4], j1_[4];
  friend class BatchedCost;
};

// What create() returns: two 3-vectors as parameter blocks and `num_residuals` residuals, exactly
// the shape of AutoDiffCostFunction<F, DYNAMIC | 3, 3, 3>.  Owns its block: ceres::Problem deletes the cost
// function (CeresInformation::ResetProblem, data_structures.h:111-116) and the block goes with it.
class BatchedCost : public CostFunctionBase {
 public:
  explicit BatchedCost(Block *block) : block_(block) {
    mutable_parameter_block_sizes()->push_back(3);
    mutable_parameter_block_sizes()->push_back(3);
    set_num_residuals(block->rows);
  }
  ~BatchedCost() override { ResidualBatcher::Instance().Unregister(block_); }
  BatchedCost(const BatchedCost &) = delete;
  BatchedCost &operator=(const BatchedCost &) = delete;
  bool Evaluate(double const *const *parameters, double *residuals, double **jacobians) const override {
    return ResidualBatcher::Instance().Fetch(block_, parameters, residuals, jacobians);
  }
  int family() const { return block_->family; }
  Block *block() const { return block_; }

 private:
  Block *block_;
};
using LidarCost = BatchedCost;

inline void ResidualBatcher::Bind(const CostFunctionBase *cost, double *pose_a, double *pose_b) {
  const BatchedCost *c = dynamic_cast<const BatchedCost *>(cost);
  if (!c) return;  // not one of ours
  std::lock_guard<std::mutex> lk(mu_);
  c->block()->pa = pose_a;
  c->block()->pb = pose_b;
  prepared_.store(false, std::memory_order_release);
}

// problem.AddResidualBlock(cost, loss, pose_a, pose_b) and the binding in one call: the one-line form of
// solver.cc:280-283, 291-293, 378, 521, 528 that avoids the single-block first pass of unbound blocks.
template <class Problem, class Loss>
inline auto AddResidualBlock(Problem &problem, CostFunctionBase *cost, Loss *loss, double *pose_a, double *pose_b)
    -> decltype(problem.AddResidualBlock(cost, loss, pose_a, pose_b)) {
  ResidualBatcher::Instance().Bind(cost, pose_a, pose_b);
  return problem.AddResidualBlock(cost, loss, pose_a, pose_b);
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
    return new nautilus_hip::BatchedCost(nautilus_hip::ResidualBatcher::Instance().RegisterOdometry(
        (float)factor.translation(0), (float)factor.translation(1), (float)factor.rotation, translation_weight,
        rotation_weight));
  }
};

struct LIDARNormalResidual {
  static nautilus_hip::BatchedCost *create(const std::vector<nautilus_hip::Vec2f> &source_points,
                                           const std::vector<nautilus_hip::Vec2f> &target_points,
                                           const std::vector<nautilus_hip::Vec2f> &source_normals,
                                           const std::vector<nautilus_hip::Vec2f> &target_normals) {
    nautilus_hip::CheckSizes(source_points, target_points, source_normals, target_normals);
    return new nautilus_hip::BatchedCost(nautilus_hip::ResidualBatcher::Instance().RegisterLidar(
        NHIP_LIDAR_NORMAL, source_points, target_points, source_normals, target_normals));
  }
};

struct LIDARPointResidual {
  static nautilus_hip::BatchedCost *create(const std::vector<nautilus_hip::Vec2f> &source_points,
                                           const std::vector<nautilus_hip::Vec2f> &target_points,
                                           const std::vector<nautilus_hip::Vec2f> &source_normals,
                                           const std::vector<nautilus_hip::Vec2f> &target_normals) {
    nautilus_hip::CheckSizes(source_points, target_points, source_normals, target_normals);
    return new nautilus_hip::BatchedCost(nautilus_hip::ResidualBatcher::Instance().RegisterLidar(
        NHIP_LIDAR_POINT, source_points, target_points, source_normals, target_normals));
  }
};

struct PointToLineResidual {
  // Segment: LineSegment<float> (data_structures.h:13-32) or anything with .start(i), .end(i)
  template <class Segment>
  static nautilus_hip::BatchedCost *create(const Segment &line_segment, const std::vector<nautilus_hip::Vec2f> points) {
    return new nautilus_hip::BatchedCost(nautilus_hip::ResidualBatcher::Instance().RegisterPointToLine(
        (float)line_segment.start(0), (float)line_segment.start(1), (float)line_segment.end(0),
        (float)line_segment.end(1), points));
  }
};

}  // namespace nautilus

#endif  // NAUTILUS_HIP_SLAM_RESIDUALS_H_
