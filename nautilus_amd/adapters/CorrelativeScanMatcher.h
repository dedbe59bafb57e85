// CorrelativeScanMatcher.h -- drop-in for third_party/csm/src/CorrelativeScanMatcher.h as nautilus
// uses it: global-namespace class, ctor (scanner_range, trans_range, low_res, high_res), and
//   std::pair<double, std::pair<Eigen::Vector2f, float>>
//   GetTransformation(pc_a, pc_b, rot_a, rot_b, rot_restriction)
// (/root/reference/src/optimization/solver.h:18,126; solver.cc:56,633-644).  Header-only; link
// with -lnautilus_hip.  All arithmetic runs on the MI355X through the C ABI (nautilus_hip.h);
// there is no CPU path: a failure throws std::runtime_error carrying nhip_last_error()
// (the reference would glog-CHECK and abort).
//
// Search (build-defined): exhaustive on the low_res grid over
// +-trans_range and +-rot_restriction (1 degree steps), then exhaustive on the high_res grid over
// +-low_res around the coarse optimum with 0.1 degree steps -- the coarse-to-fine the ctor's
// (low_res, high_res) pair implies; implemented once, in nhip_csm_get_transformation.
// CorrelativeScanMatcherBatch below is the batched form the loop-closure driver should call with the
// whole candidate-pair list.
#ifndef NAUTILUS_HIP_CORRELATIVE_SCAN_MATCHER_H_
#define NAUTILUS_HIP_CORRELATIVE_SCAN_MATCHER_H_

#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "nautilus_hip.h"

#if __has_include(<Eigen/Dense>)
#include <Eigen/Dense>
namespace nautilus_hip { using Vec2f = Eigen::Vector2f; }
#else
namespace nautilus_hip {
// Layout-compatible stand-in used only when Eigen is not installed (two packed floats, like
// Eigen::Vector2f); lets the adapter be compiled and tested in an image without Eigen.
struct Vec2f {
  float v[2];
  Vec2f() : v{0.f, 0.f} {}
  Vec2f(float x, float y) : v{x, y} {}
  float operator()(int i) const { return v[i]; }
  float x() const { return v[0]; }
  float y() const { return v[1]; }
};
}  // namespace nautilus_hip
#endif

// Cell width of the likelihood tables the drop-ins build: 16 keeps scores within 1e-5 of an unquantised table
// (DESIGN.md section 3); define NAUTILUS_HIP_CELL_BITS=8 for the faster 8-bit tables.
#ifndef NAUTILUS_HIP_CELL_BITS
#define NAUTILUS_HIP_CELL_BITS 16
#endif

namespace nautilus_hip {

static_assert(sizeof(Vec2f) == 2 * sizeof(float), "point clouds are passed as packed float pairs");

inline void Check(int rc, const char *what) {
  if (rc != NHIP_OK) throw std::runtime_error(std::string(what) + ": " + nhip_last_error());
}

struct ScansHandle {
  nhip_scans_t *h = nullptr;
  ScansHandle(const std::vector<const std::vector<Vec2f> *> &clouds) {
    std::vector<int32_t> off(clouds.size() + 1, 0);
    for (size_t i = 0; i < clouds.size(); i++) off[i + 1] = off[i] + (int32_t)clouds[i]->size();
    std::vector<float> xy(2 * (size_t)off.back());
    for (size_t i = 0; i < clouds.size(); i++)
      for (size_t p = 0; p < clouds[i]->size(); p++) {
        xy[2 * (off[i] + p)] = (*clouds[i])[p](0);
        xy[2 * (off[i] + p) + 1] = (*clouds[i])[p](1);
      }
    Check(nhip_scans_upload(xy.data(), off.data(), (int32_t)clouds.size(), &h), "nhip_scans_upload");
  }
  ~ScansHandle() { nhip_scans_free(h); }
  ScansHandle(const ScansHandle &) = delete;
  ScansHandle &operator=(const ScansHandle &) = delete;
};

struct GridsHandle {
  nhip_grids_t *h = nullptr;
  GridsHandle(const ScansHandle &s, const std::vector<int32_t> &targets, const nhip_grid_spec_t &spec) {
    Check(nhip_grids_build(s.h, targets.data(), (int32_t)targets.size(), &spec, &h), "nhip_grids_build");
  }
  ~GridsHandle() { nhip_grids_free(h); }
  GridsHandle(const GridsHandle &) = delete;
  GridsHandle &operator=(const GridsHandle &) = delete;
};

inline double AngleMod(double a) { return a - 2.0 * M_PI * std::rint(a / (2.0 * M_PI)); }  // math_util.h:81-84

}  // namespace nautilus_hip

class CorrelativeScanMatcher {
 public:
  using Vector2f = nautilus_hip::Vec2f;

  CorrelativeScanMatcher(double scanner_range, double trans_range, double low_res, double high_res)
      : params_{scanner_range, trans_range, low_res, high_res, 2.0, 1e-10, NAUTILUS_HIP_CELL_BITS, 0} {}

  // The search is nhip_csm_get_transformation's (the one implementation, behind the C ABI).
  std::pair<double, std::pair<Vector2f, float>> GetTransformation(
      const std::vector<Vector2f> &pointcloud_a, const std::vector<Vector2f> &pointcloud_b,
      double rotation_a, double rotation_b, double rotation_restriction) const {
    double score;
    float tx, ty, th;
    nautilus_hip::Check(
        nhip_csm_get_transformation(&params_, reinterpret_cast<const float *>(pointcloud_a.data()),
                                    (int32_t)pointcloud_a.size(), reinterpret_cast<const float *>(pointcloud_b.data()),
                                    (int32_t)pointcloud_b.size(), rotation_a, rotation_b, rotation_restriction, &score,
                                    &tx, &ty, &th),
        "nhip_csm_get_transformation");
    return {score, {Vector2f(tx, ty), th}};
  }

 private:
  nhip_csm_params_t params_;
};

// Batched form: all scans once, all candidate pairs in one call (what SolveAutoLC's pair list,
// solver.cc:676-700, should be handed to).  One grid per distinct target; single-level search
// `search` on a grid of resolution `res` (BASELINE config #2: res 0.05, 61 x 81 x 81).
class CorrelativeScanMatcherBatch {
 public:
  using Vector2f = nautilus_hip::Vec2f;
  struct Result {
    double score;
    Vector2f translation;
    float rotation;
  };

  CorrelativeScanMatcherBatch(double scanner_range, double res, const nhip_search_t &search)
      : search_(search) {
    spec_ = {scanner_range, res, 2.0, 1e-10, std::max(search.nx, search.ny) / 2, NAUTILUS_HIP_CELL_BITS, 0, 0};
  }

  // pairs: (source index, target index) into `clouds`; rotations: world heading of every cloud.
  std::vector<Result> Match(const std::vector<std::vector<Vector2f>> &clouds,
                            const std::vector<std::pair<size_t, size_t>> &pairs,
                            const std::vector<double> &rotations) const {
    using namespace nautilus_hip;
    std::vector<const std::vector<Vector2f> *> ptrs;
    for (const auto &c : clouds) ptrs.push_back(&c);
    ScansHandle scans(ptrs);
    std::vector<int32_t> slot_of(clouds.size(), -1), targets, src(pairs.size()), slot(pairs.size());
    std::vector<double> theta0(pairs.size());
    for (size_t i = 0; i < pairs.size(); i++) {
      const size_t s = pairs[i].first, t = pairs[i].second;
      if (s >= clouds.size() || t >= clouds.size()) throw std::runtime_error("pair index out of range");
      if (slot_of[t] < 0) { slot_of[t] = (int32_t)targets.size(); targets.push_back((int32_t)t); }
      src[i] = (int32_t)s;
      slot[i] = slot_of[t];
      theta0[i] = AngleMod(rotations[s] - rotations[t]);
    }
    GridsHandle grids(scans, targets, spec_);
    std::vector<nhip_match_t> m(pairs.size());
    Check(nhip_csm_match(scans.h, grids.h, src.data(), slot.data(), theta0.data(), nullptr, (int32_t)pairs.size(),
                         &search_, m.data(), nullptr), "nhip_csm_match");
    std::vector<Result> out(pairs.size());
    for (size_t i = 0; i < pairs.size(); i++) {
      float tx, ty, th;
      Check(nhip_match_to_transform(&m[i], &spec_, &search_, theta0[i], 0, 0, &tx, &ty, &th), "nhip_match_to_transform");
      out[i] = {(double)m[i].score, Vector2f(tx, ty), th};
    }
    return out;
  }

 private:
  nhip_grid_spec_t spec_;
  nhip_search_t search_;
};

#endif  // NAUTILUS_HIP_CORRELATIVE_SCAN_MATCHER_H_
