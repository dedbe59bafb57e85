/*
 * oracle/residuals_oracle.cc -- CPU restatement of nautilus's Ceres cost functors.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this.  The product path never links or calls it.
 *
 * Pinning: DistanceToLineSegment is pinned by the six known-answer tests of
 * /root/reference/test/solver_test.cc:12-64 (tests/test_oracle_kat.py).  The four functors
 * have no golden values in the reference; they are restated line by
 * line from source that IS in the tree, and their Jacobians are produced the way
 * ceres::AutoDiffCostFunction produces them (forward-mode duals, Jet<double,6>), then
 * cross-checked against central differences and sympy in tests/.  The reference itself
 * cannot be compiled here (Eigen 3.3.7 / Ceres 1.14 / glog / ROS headers absent).
 *
 * Follows:
 *   src/optimization/slam_residuals.h:17-61    OdometryResidual
 *   src/optimization/slam_residuals.h:64-121   LIDARNormalResidual
 *   src/optimization/slam_residuals.h:123-177  LIDARPointResidual
 *   src/optimization/slam_residuals.h:179-216  PointToLineResidual
 *   src/util/slam_util.h:20-28                 PoseArrayToAffine
 *   src/util/slam_util.h:87-110                IsBetween, DistanceToLineSegment
 *   Eigen 3.3.7 semantics restated: Translation * Rotation2D -> Affine; Affine-mode
 *   Transform::inverse() (general 2x2 inverse of the linear part, translation = -Linv*t);
 *   Hyperplane::Through / signedDistance / absDistance / projection;
 *   MatrixBase::unitOrthogonal (2-D) and normalized().
 *   Ceres 1.14 semantics restated: Jet<double,6> arithmetic, sin/cos/atan2/sqrt/abs,
 *   comparisons on the scalar part; AutoDiffCostFunction::Evaluate writes row-major
 *   num_residuals x 3 Jacobians per parameter block, skipping null pointers.
 */
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// ---------------------------------------------------------------- Jet<double, 6>
struct Jet6 {
  double a;
  double v[6];
  Jet6() : a(0.0) { for (double &x : v) x = 0.0; }
  Jet6(double s) : a(s) { for (double &x : v) x = 0.0; }  // NOLINT: T(scalar) as in Ceres
  Jet6(double s, int k) : a(s) { for (double &x : v) x = 0.0; v[k] = 1.0; }
};
inline Jet6 operator+(const Jet6 &f, const Jet6 &g) {
  Jet6 h; h.a = f.a + g.a; for (int i = 0; i < 6; i++) h.v[i] = f.v[i] + g.v[i]; return h;
}
inline Jet6 operator-(const Jet6 &f, const Jet6 &g) {
  Jet6 h; h.a = f.a - g.a; for (int i = 0; i < 6; i++) h.v[i] = f.v[i] - g.v[i]; return h;
}
inline Jet6 operator-(const Jet6 &f) {
  Jet6 h; h.a = -f.a; for (int i = 0; i < 6; i++) h.v[i] = -f.v[i]; return h;
}
inline Jet6 operator*(const Jet6 &f, const Jet6 &g) {
  Jet6 h; h.a = f.a * g.a;
  for (int i = 0; i < 6; i++) h.v[i] = f.a * g.v[i] + f.v[i] * g.a;
  return h;
}
inline Jet6 operator/(const Jet6 &f, const Jet6 &g) {
  // Ceres jet.h: g_a_inverse = 1/g.a; f_a_by_g_a = f.a * g_a_inverse;
  // h.v = (f.v - f_a_by_g_a * g.v) * g_a_inverse
  Jet6 h; const double gi = 1.0 / g.a; const double fg = f.a * gi; h.a = fg;
  for (int i = 0; i < 6; i++) h.v[i] = (f.v[i] - fg * g.v[i]) * gi;
  return h;
}
inline bool operator<(const Jet6 &f, const Jet6 &g) { return f.a < g.a; }
inline bool operator<=(const Jet6 &f, const Jet6 &g) { return f.a <= g.a; }
inline bool operator>=(const Jet6 &f, const Jet6 &g) { return f.a >= g.a; }
inline Jet6 sin(const Jet6 &f) {
  Jet6 h; h.a = std::sin(f.a); const double c = std::cos(f.a);
  for (int i = 0; i < 6; i++) h.v[i] = c * f.v[i];
  return h;
}
inline Jet6 cos(const Jet6 &f) {
  Jet6 h; h.a = std::cos(f.a); const double s = -std::sin(f.a);
  for (int i = 0; i < 6; i++) h.v[i] = s * f.v[i];
  return h;
}
inline Jet6 atan2(const Jet6 &g, const Jet6 &f) {
  // Ceres jet.h: atan2(g, f): tmp = 1/(f.a^2 + g.a^2); v = tmp * (-g.a * f.v + f.a * g.v)
  Jet6 h; h.a = std::atan2(g.a, f.a);
  const double tmp = 1.0 / (f.a * f.a + g.a * g.a);
  for (int i = 0; i < 6; i++) h.v[i] = tmp * (-g.a * f.v[i] + f.a * g.v[i]);
  return h;
}
inline Jet6 sqrt(const Jet6 &f) {
  Jet6 h; h.a = std::sqrt(f.a); const double t = 1.0 / (2.0 * h.a);
  for (int i = 0; i < 6; i++) h.v[i] = t * f.v[i];
  return h;
}
inline Jet6 abs(const Jet6 &f) { return f.a < 0.0 ? -f : f; }

inline float sin(float x) { return std::sin(x); }
inline float cos(float x) { return std::cos(x); }
inline float sqrt(float x) { return std::sqrt(x); }
inline float abs(float x) { return std::fabs(x); }
inline double sin(double x) { return std::sin(x); }
inline double cos(double x) { return std::cos(x); }
inline double sqrt(double x) { return std::sqrt(x); }
inline double abs(double x) { return std::fabs(x); }
inline double atan2(double y, double x) { return std::atan2(y, x); }

// ---------------------------------------------------------------- mini Eigen
template <typename T> struct Vec2 { T x, y; };
template <typename T> inline Vec2<T> operator-(const Vec2<T> &a, const Vec2<T> &b) {
  return {a.x - b.x, a.y - b.y};
}
template <typename T> inline Vec2<T> operator+(const Vec2<T> &a, const Vec2<T> &b) {
  return {a.x + b.x, a.y + b.y};
}
template <typename T> inline T dot(const Vec2<T> &a, const Vec2<T> &b) {
  return a.x * b.x + a.y * b.y;
}
template <typename T> inline T norm(const Vec2<T> &a) { return sqrt(dot(a, a)); }

// Affine 2-D transform [m00 m01 tx; m10 m11 ty]
template <typename T> struct Affine2 { T m00, m01, m10, m11, tx, ty; };

// slam_util.h:20-28: Translation2T(x, y) * Rotation2DT(theta).toRotationMatrix()
template <typename T> inline Affine2<T> PoseArrayToAffine(const T *rotation, const T *translation) {
  const T c = cos(rotation[0]), s = sin(rotation[0]);
  return {c, -s, s, c, translation[0], translation[1]};
}
// Eigen Transform<T,2,Affine>::inverse(): linear part inverted as a general 2x2 matrix
// (adjugate * 1/det), translation = -(Linv * t).
template <typename T> inline Affine2<T> Inverse(const Affine2<T> &A) {
  const T det = A.m00 * A.m11 - A.m10 * A.m01;
  const T invdet = T(1.0) / det;
  Affine2<T> R;
  R.m00 = A.m11 * invdet;
  R.m10 = -A.m10 * invdet;
  R.m01 = -A.m01 * invdet;
  R.m11 = A.m00 * invdet;
  R.tx = -(R.m00 * A.tx + R.m01 * A.ty);
  R.ty = -(R.m10 * A.tx + R.m11 * A.ty);
  return R;
}
template <typename T> inline Affine2<T> operator*(const Affine2<T> &A, const Affine2<T> &B) {
  Affine2<T> C;
  C.m00 = A.m00 * B.m00 + A.m01 * B.m10;
  C.m01 = A.m00 * B.m01 + A.m01 * B.m11;
  C.m10 = A.m10 * B.m00 + A.m11 * B.m10;
  C.m11 = A.m10 * B.m01 + A.m11 * B.m11;
  C.tx = A.m00 * B.tx + A.m01 * B.ty + A.tx;
  C.ty = A.m10 * B.tx + A.m11 * B.ty + A.ty;
  return C;
}
template <typename T> inline Vec2<T> operator*(const Affine2<T> &A, const Vec2<T> &p) {
  return {A.m00 * p.x + A.m01 * p.y + A.tx, A.m10 * p.x + A.m11 * p.y + A.ty};
}

template <typename T> struct LineSegment { Vec2<T> start, end; };

// slam_util.h:87-89
template <typename T> inline bool IsBetween(const T &val, const T &a, const T &b) {
  return (val >= a && val <= b) || (val >= b && val <= a);
}

// slam_util.h:92-110
template <typename T> inline T DistanceToLineSegment(const Vec2<T> &point, const LineSegment<T> &seg) {
  // Hyperplane::Through(p0, p1): normal = (p1 - p0).unitOrthogonal(), offset = -p0.dot(normal)
  const Vec2<T> d = seg.end - seg.start;
  Vec2<T> n = {-d.y, d.x};
  const T z = dot(n, n);
  const T len = sqrt(z);
  n = {n.x / len, n.y / len};
  const T offset = -dot(seg.start, n);
  const T signed_dist = dot(n, point) + offset;          // signedDistance
  const Vec2<T> proj = {point.x - signed_dist * n.x,     // projection
                        point.y - signed_dist * n.y};
  if (IsBetween(proj.x, seg.start.x, seg.end.x) && IsBetween(proj.y, seg.start.y, seg.end.y)) {
    return abs(signed_dist);                               // absDistance
  }
  const T dist_to_start = norm(point - seg.start);
  const T dist_to_endpoint = norm(point - seg.end);
  return (dist_to_endpoint < dist_to_start) ? dist_to_endpoint : dist_to_start;  // std::min
}

template <typename T> inline Vec2<T> cast2(const float *p) { return {T((double)p[0]), T((double)p[1])}; }

// slam_residuals.h:65-89
template <typename T>
void LIDARNormal(const float *sp, const float *tp, const float *sn, const float *tn, int n,
                 const T *source_pose, const T *target_pose, T *residuals) {
  const Affine2<T> source_to_world = PoseArrayToAffine(&source_pose[2], &source_pose[0]);
  const Affine2<T> world_to_target = Inverse(PoseArrayToAffine(&target_pose[2], &target_pose[0]));
  const Affine2<T> source_to_target = world_to_target * source_to_world;
  for (int i = 0; i < n; i++) {
    Vec2<T> s = cast2<T>(sp + 2 * i);
    const Vec2<T> t = cast2<T>(tp + 2 * i);
    s = source_to_target * s;
    residuals[2 * i] = dot(cast2<T>(tn + 2 * i), s - t);
    residuals[2 * i + 1] = dot(cast2<T>(sn + 2 * i), t - s);
  }
}

// slam_residuals.h:124-145
template <typename T>
void LIDARPoint(const float *sp, const float *tp, int n, const T *source_pose,
                const T *target_pose, T *residuals) {
  const Affine2<T> source_to_world = PoseArrayToAffine(&source_pose[2], &source_pose[0]);
  const Affine2<T> world_to_target = Inverse(PoseArrayToAffine(&target_pose[2], &target_pose[0]));
  const Affine2<T> source_to_target = world_to_target * source_to_world;
  for (int i = 0; i < n; i++) {
    Vec2<T> s = cast2<T>(sp + 2 * i);
    const Vec2<T> t = cast2<T>(tp + 2 * i);
    s = source_to_target * s;
    const Vec2<T> diff = t - s;
    residuals[2 * i] = diff.x;
    residuals[2 * i + 1] = diff.y;
  }
}

// slam_residuals.h:180-200
template <typename T>
void PointToLine(const float *seg /*x0 y0 x1 y1*/, const float *pts, int n, const T *pose,
                 const T *line_pose, T *residuals) {
  const Affine2<T> pose_to_world = PoseArrayToAffine(&pose[2], &pose[0]);
  const Affine2<T> line_to_world = PoseArrayToAffine(&line_pose[2], &line_pose[0]);
  const Vec2<T> line_start = line_to_world * cast2<T>(seg);
  const Vec2<T> line_end = line_to_world * cast2<T>(seg + 2);
  const LineSegment<T> transformed = {line_start, line_end};
  for (int i = 0; i < n; i++) {
    Vec2<T> p = cast2<T>(pts + 2 * i);
    p = pose_to_world * p;
    residuals[i] = DistanceToLineSegment(p, transformed);
  }
}

// slam_residuals.h:18-40 (T_odom is Vector2f, R_odom is float, weights are double)
template <typename T>
void Odometry(const float *t_odom, float r_odom, double tw, double rw, const T *pose_i,
              const T *pose_j, T *residual) {
  const Vec2<T> Ti = {pose_i[0], pose_i[1]};
  const Vec2<T> Tj = {pose_j[0], pose_j[1]};
  const Vec2<T> err = Ti + cast2<T>(t_odom) - Tj;
  const T rotation_diff = pose_i[2] + T((double)r_odom) - pose_j[2];
  const T error_rotation = atan2(sin(rotation_diff), cos(rotation_diff));
  residual[0] = T(tw) * err.x;
  residual[1] = T(tw) * err.y;
  residual[2] = T(rw) * error_rotation;
}

// AutoDiffCostFunction::Evaluate: seed parameter block 0 on partials 0..2, block 1 on 3..5.
inline void SeedJets(const double *p0, const double *p1, Jet6 *j0, Jet6 *j1) {
  for (int k = 0; k < 3; k++) { j0[k] = Jet6(p0[k], k); j1[k] = Jet6(p1[k], 3 + k); }
}
inline void Scatter(const Jet6 *r, int nres, double *residuals, double *jac0, double *jac1) {
  for (int i = 0; i < nres; i++) {
    residuals[i] = r[i].a;
    if (jac0) for (int k = 0; k < 3; k++) jac0[3 * i + k] = r[i].v[k];
    if (jac1) for (int k = 0; k < 3; k++) jac1[3 * i + k] = r[i].v[3 + k];
  }
}

}  // namespace

extern "C" {

// test/solver_test.cc:12-64 instantiates DistanceToLineSegment<float>.
float orc_dist_to_segment_f(float px, float py, float x0, float y0, float x1, float y1) {
  LineSegment<float> s = {{x0, y0}, {x1, y1}};
  return DistanceToLineSegment<float>({px, py}, s);
}
double orc_dist_to_segment_d(double px, double py, double x0, double y0, double x1, double y1) {
  LineSegment<double> s = {{x0, y0}, {x1, y1}};
  return DistanceToLineSegment<double>({px, py}, s);
}

/* kind: 0 = LIDARNormalResidual, 1 = LIDARPointResidual.  jac0/jac1 may be NULL
 * (a NULL jacobian pointer for a constant parameter block, solver.cc:384-386); when both
 * are NULL the functor runs on plain doubles exactly as AutoDiffCostFunction does. */
int orc_lidar_block(int kind, const float *sp, const float *tp, const float *sn,
                    const float *tn, int n, const double *source_pose,
                    const double *target_pose, double *residuals, double *jac0, double *jac1) {
  if (n <= 0) return -1;
  if (!jac0 && !jac1) {
    if (kind == 0) LIDARNormal<double>(sp, tp, sn, tn, n, source_pose, target_pose, residuals);
    else LIDARPoint<double>(sp, tp, n, source_pose, target_pose, residuals);
    return 0;
  }
  Jet6 a[3], b[3];
  SeedJets(source_pose, target_pose, a, b);
  std::vector<Jet6> r(2 * (size_t)n);
  if (kind == 0) LIDARNormal<Jet6>(sp, tp, sn, tn, n, a, b, r.data());
  else LIDARPoint<Jet6>(sp, tp, n, a, b, r.data());
  Scatter(r.data(), 2 * n, residuals, jac0, jac1);
  return 0;
}

int orc_point_to_line_block(const float *seg, const float *pts, int n, const double *pose,
                            const double *line_pose, double *residuals, double *jac0,
                            double *jac1) {
  if (n <= 0) return -1;
  if (!jac0 && !jac1) {
    PointToLine<double>(seg, pts, n, pose, line_pose, residuals);
    return 0;
  }
  Jet6 a[3], b[3];
  SeedJets(pose, line_pose, a, b);
  std::vector<Jet6> r((size_t)n);
  PointToLine<Jet6>(seg, pts, n, a, b, r.data());
  Scatter(r.data(), n, residuals, jac0, jac1);
  return 0;
}

int orc_odometry_block(const float *t_odom, float r_odom, double tw, double rw,
                       const double *pose_i, const double *pose_j, double *residuals,
                       double *jac0, double *jac1) {
  if (!jac0 && !jac1) {
    Odometry<double>(t_odom, r_odom, tw, rw, pose_i, pose_j, residuals);
    return 0;
  }
  Jet6 a[3], b[3];
  SeedJets(pose_i, pose_j, a, b);
  Jet6 r[3];
  Odometry<Jet6>(t_odom, r_odom, tw, rw, a, b, r);
  Scatter(r, 3, residuals, jac0, jac1);
  return 0;
}

/*
 * Batched LIDAR evaluation with the product's batch layout (for parity tests and the
 * cpu_baseline leg): correspondences are 8 floats each (sp, tp, sn, tn), blocks are
 * [block_offsets[b], block_offsets[b+1]) with pose indices block_src/block_tgt into
 * poses[n_poses][3].  residuals: 2 doubles per correspondence; jac_src / jac_tgt: 6 doubles
 * per correspondence (row-major 2x3), either may be NULL.
 * Parallelism mirrors the reference: blocks across threads (Ceres num_threads,
 * solver.cc:271); the reference's inner per-point omp loop (slam_residuals.h:75) is not
 * nested here.
 */
int orc_lidar_batch(int kind, const float *corr, const int32_t *block_offsets,
                    const int32_t *block_src, const int32_t *block_tgt, int32_t n_blocks,
                    const double *poses, double *residuals, double *jac_src, double *jac_tgt,
                    int32_t n_threads) {
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel
#endif
  {
    std::vector<float> sp, tp, sn, tn;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 4)
#endif
    for (int32_t b = 0; b < n_blocks; b++) {
      const int32_t o = block_offsets[b], n = block_offsets[b + 1] - o;
      if (n <= 0) continue;
      sp.resize(2 * n); tp.resize(2 * n); sn.resize(2 * n); tn.resize(2 * n);
      for (int i = 0; i < n; i++) {
        const float *c = corr + 8 * (size_t)(o + i);
        sp[2 * i] = c[0]; sp[2 * i + 1] = c[1]; tp[2 * i] = c[2]; tp[2 * i + 1] = c[3];
        sn[2 * i] = c[4]; sn[2 * i + 1] = c[5]; tn[2 * i] = c[6]; tn[2 * i + 1] = c[7];
      }
      orc_lidar_block(kind, sp.data(), tp.data(), sn.data(), tn.data(), n,
                      poses + 3 * (size_t)block_src[b], poses + 3 * (size_t)block_tgt[b],
                      residuals + 2 * (size_t)o, jac_src ? jac_src + 6 * (size_t)o : nullptr,
                      jac_tgt ? jac_tgt + 6 * (size_t)o : nullptr);
    }
  }
  (void)n_threads;
  return 0;
}

}  // extern "C"

// Closed-form Jacobians of the two LIDAR functors, for the "residuals analytic (CPU)"
// baseline of BASELINE.md section 3 and as a second opinion on the autodiff restatement.  With
// w = R(th_s) p + t_s - t_t, q = R(th_t)^T w, u = R(th_s - th_t) p:
//   dq/dt_s = R(th_t)^T, dq/dth_s = (-u_y, u_x), dq/dt_t = -R(th_t)^T, dq/dth_t = (q_y, -q_x).
static void lidar_block_analytic(int kind, const float *c, int n, const double *ps, const double *pt,
                                 double *res, double *j0, double *j1) {
  const double cs = std::cos(ps[2]), ss = std::sin(ps[2]), ct = std::cos(pt[2]), st = std::sin(pt[2]);
  const double cd = cs * ct + ss * st, sd = ss * ct - cs * st;  // cos, sin (th_s - th_t)
  const double dx = ps[0] - pt[0], dy = ps[1] - pt[1];
  const double ox = ct * dx + st * dy, oy = -st * dx + ct * dy;  // R(th_t)^T (t_s - t_t)
  for (int i = 0; i < n; i++) {
    const float *r = c + 8 * (size_t)i;
    const double px = r[0], py = r[1], tx = r[2], ty = r[3];
    const double ux = cd * px - sd * py, uy = sd * px + cd * py;
    const double qx = ux + ox, qy = uy + oy;
    // rows of dq/d(source pose) and dq/d(target pose): 2 x 3 each
    const double qs[2][3] = {{ct, st, -uy}, {-st, ct, ux}};
    const double qt[2][3] = {{-ct, -st, qy}, {st, -ct, -qx}};
    if (kind == 1) {
      res[2 * i] = tx - qx;
      res[2 * i + 1] = ty - qy;
      for (int k = 0; k < 3; k++) {
        if (j0) { j0[6 * i + k] = -qs[0][k]; j0[6 * i + 3 + k] = -qs[1][k]; }
        if (j1) { j1[6 * i + k] = -qt[0][k]; j1[6 * i + 3 + k] = -qt[1][k]; }
      }
    } else {
      const double snx = r[4], sny = r[5], tnx = r[6], tny = r[7];
      res[2 * i] = tnx * (qx - tx) + tny * (qy - ty);
      res[2 * i + 1] = snx * (tx - qx) + sny * (ty - qy);
      for (int k = 0; k < 3; k++) {
        if (j0) { j0[6 * i + k] = tnx * qs[0][k] + tny * qs[1][k]; j0[6 * i + 3 + k] = -(snx * qs[0][k] + sny * qs[1][k]); }
        if (j1) { j1[6 * i + k] = tnx * qt[0][k] + tny * qt[1][k]; j1[6 * i + 3 + k] = -(snx * qt[0][k] + sny * qt[1][k]); }
      }
    }
  }
}

extern "C" int orc_lidar_batch_analytic(int kind, const float *corr, const int32_t *block_offsets,
                                        const int32_t *block_src, const int32_t *block_tgt, int32_t n_blocks,
                                        const double *poses, double *residuals, double *jac_src, double *jac_tgt,
                                        int32_t n_threads) {
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel for schedule(dynamic, 4)
#endif
  for (int32_t b = 0; b < n_blocks; b++) {
    const int32_t o = block_offsets[b], n = block_offsets[b + 1] - o;
    if (n <= 0) continue;
    lidar_block_analytic(kind, corr + 8 * (size_t)o, n, poses + 3 * (size_t)block_src[b],
                         poses + 3 * (size_t)block_tgt[b], residuals + 2 * (size_t)o,
                         jac_src ? jac_src + 6 * (size_t)o : nullptr, jac_tgt ? jac_tgt + 6 * (size_t)o : nullptr);
  }
  (void)n_threads;
  return 0;
}

// =====================================================================================
// Correspondence search -- CPU restatement of
//   Solver::GetPointToPointMatching            src/optimization/solver.cc:132-172
//   FindClosestPoint                           src/optimization/solver.cc:80-90
//   KDTree<float,2>::FindNearestPoint          src/util/kdtree.cc:253-305
//   GetPoseAsAffine<float>                     src/util/slam_util.h:37-40
// For every source point: transform into the target frame with
// target_to_world.inverse() * source_to_world (Affine2f, float arithmetic), take the nearest
// target point (float Euclidean norm), keep it if dist < outlier_threshold, and append
// (source point, source normal, match, target normal) in source order.
// The kd-tree descent returns the exact nearest neighbour whenever one lies within the
// threshold (it only prunes branches farther than min(best, threshold)), so a linear scan is
// the same function.  Build-defined detail: exact distance ties go to the lowest target index
// (the tree's own tie order depends on its build and is not specified by the reference).
// Normals are inputs (the reference reads them from the trees, solver.cc:67-78; its Hough normal
// estimation is non-deterministic, normal_computation.cc:82).
namespace {
struct Aff2f { float m00, m01, m10, m11, tx, ty; };
inline Aff2f PoseAffineF(const float *a /* cos sin x y, already cast from double */) {
  return {a[0], -a[1], a[1], a[0], a[2], a[3]};
}
inline Aff2f InverseF(const Aff2f &A) {
  const float det = A.m00 * A.m11 - A.m10 * A.m01;
  const float invdet = 1.0f / det;
  Aff2f R;
  R.m00 = A.m11 * invdet; R.m10 = -A.m10 * invdet; R.m01 = -A.m01 * invdet; R.m11 = A.m00 * invdet;
  R.tx = -(R.m00 * A.tx + R.m01 * A.ty);
  R.ty = -(R.m10 * A.tx + R.m11 * A.ty);
  return R;
}
inline Aff2f MulF(const Aff2f &A, const Aff2f &B) {
  Aff2f C;
  C.m00 = A.m00 * B.m00 + A.m01 * B.m10; C.m01 = A.m00 * B.m01 + A.m01 * B.m11;
  C.m10 = A.m10 * B.m00 + A.m11 * B.m10; C.m11 = A.m10 * B.m01 + A.m11 * B.m11;
  C.tx = A.m00 * B.tx + A.m01 * B.ty + A.tx;
  C.ty = A.m10 * B.tx + A.m11 * B.ty + A.ty;
  return C;
}
}  // namespace

extern "C" {

/* (cos, sin, x, y) of a double[3] pose, each cast to float: the entries of
 * PoseArrayToAffine<double>(pose).cast<float>() (slam_util.h:37-40). */
void orc_pose_affines(const double *poses, int32_t n, float *out /* 4n */) {
  for (int32_t i = 0; i < n; i++) {
    out[4 * i + 0] = (float)std::cos(poses[3 * i + 2]);
    out[4 * i + 1] = (float)std::sin(poses[3 * i + 2]);
    out[4 * i + 2] = (float)poses[3 * i + 0];
    out[4 * i + 3] = (float)poses[3 * i + 1];
  }
}

/* One (source, target) block.  corr_out: up to n_src rows of 8 floats (source point, target
 * point, source normal, target normal -- the layout the residual batch takes); match_idx_out
 * (optional): target index per kept row.  Returns the number of correspondences. */
int32_t orc_corr_search_block(const float *src_xy, const float *src_nrm, int32_t n_src,
                              const float *tgt_xy, const float *tgt_nrm, int32_t n_tgt,
                              const float *src_aff /*cos sin x y*/, const float *tgt_aff,
                              float outlier_threshold, float *corr_out, int32_t *match_idx_out) {
  const Aff2f C = MulF(InverseF(PoseAffineF(tgt_aff)), PoseAffineF(src_aff));
  int32_t n = 0;
  for (int32_t p = 0; p < n_src; p++) {
    const float px = src_xy[2 * p], py = src_xy[2 * p + 1];
    const float qx = C.m00 * px + C.m01 * py + C.tx;
    const float qy = C.m10 * px + C.m11 * py + C.ty;
    float best = 0.f;
    int32_t bi = -1;
    for (int32_t t = 0; t < n_tgt; t++) {
      const float dx = tgt_xy[2 * t] - qx, dy = tgt_xy[2 * t + 1] - qy;
      const float d2 = dx * dx + dy * dy;  // squaredNorm, individually rounded (no FMA)
      if (bi < 0 || d2 < best) { best = d2; bi = t; }
    }
    if (bi < 0) continue;
    if (!(std::sqrt(best) < outlier_threshold)) continue;  // dist < CONFIG_outlier_threshold
    float *c = corr_out + 8 * (size_t)n;
    c[0] = px; c[1] = py; c[2] = tgt_xy[2 * bi]; c[3] = tgt_xy[2 * bi + 1];
    c[4] = src_nrm[2 * p]; c[5] = src_nrm[2 * p + 1]; c[6] = tgt_nrm[2 * bi]; c[7] = tgt_nrm[2 * bi + 1];
    if (match_idx_out) match_idx_out[n] = bi;
    n++;
  }
  return n;
}

/* Solver::GetPointToNormalMatching + FindClosestPointWithSimilarNormal (solver.cc:177-260; defined,
 * not called at this commit): FindNeighborPoints collects every target with norm(diff) < threshold
 * (kdtree.cc:234-251 -- the growing-threshold wrapper always passes CONFIG_outlier_threshold to it,
 * solver.cc:182-183), they are sorted by distance and the first whose normal is "similar" --
 * fabs(n_target . n_source) > max_cosine_value, math_util.h:46-49, n_source in the SOURCE frame --
 * wins.  std::sort is not stable and compares float norms; build-defined here: squared distance,
 * then the lowest target index. */
int32_t orc_corr_search_gated_block(const float *src_xy, const float *src_nrm, int32_t n_src,
                                    const float *tgt_xy, const float *tgt_nrm, int32_t n_tgt,
                                    const float *src_aff, const float *tgt_aff, float outlier_threshold,
                                    float min_abs_cosine, float *corr_out, int32_t *match_idx_out) {
  const Aff2f C = MulF(InverseF(PoseAffineF(tgt_aff)), PoseAffineF(src_aff));
  int32_t n = 0;
  for (int32_t p = 0; p < n_src; p++) {
    const float px = src_xy[2 * p], py = src_xy[2 * p + 1];
    const float nx = src_nrm[2 * p], ny = src_nrm[2 * p + 1];
    const float qx = C.m00 * px + C.m01 * py + C.tx;
    const float qy = C.m10 * px + C.m11 * py + C.ty;
    float best = 0.f;
    int32_t bi = -1;
    for (int32_t t = 0; t < n_tgt; t++) {
      const float dx = tgt_xy[2 * t] - qx, dy = tgt_xy[2 * t + 1] - qy;
      const float d2 = dx * dx + dy * dy;
      if (!(std::sqrt(d2) < outlier_threshold)) continue;
      const float a = tgt_nrm[2 * t] * nx, b = tgt_nrm[2 * t + 1] * ny;
      const float dot = a + b;
      if (!(std::fabs(dot) > min_abs_cosine)) continue;
      if (bi < 0 || d2 < best) { best = d2; bi = t; }
    }
    if (bi < 0) continue;
    float *c = corr_out + 8 * (size_t)n;
    c[0] = px; c[1] = py; c[2] = tgt_xy[2 * bi]; c[3] = tgt_xy[2 * bi + 1];
    c[4] = nx; c[5] = ny; c[6] = tgt_nrm[2 * bi]; c[7] = tgt_nrm[2 * bi + 1];
    if (match_idx_out) match_idx_out[n] = bi;
    n++;
  }
  return n;
}

int orc_corr_search_gated_batch(const float *xy, const float *normals, const int32_t *offsets,
                                const int32_t *block_src, const int32_t *block_tgt, int32_t n_blocks,
                                const float *pose_aff, float outlier_threshold, float min_abs_cosine,
                                const int64_t *cap_offsets, float *corr, int32_t *counts, int32_t n_threads) {
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel for schedule(dynamic, 8)
#endif
  for (int32_t b = 0; b < n_blocks; b++) {
    const int32_t s = block_src[b], t = block_tgt[b];
    counts[b] = orc_corr_search_gated_block(xy + 2 * (size_t)offsets[s], normals + 2 * (size_t)offsets[s],
                                            offsets[s + 1] - offsets[s], xy + 2 * (size_t)offsets[t],
                                            normals + 2 * (size_t)offsets[t], offsets[t + 1] - offsets[t],
                                            pose_aff + 4 * (size_t)s, pose_aff + 4 * (size_t)t, outlier_threshold,
                                            min_abs_cosine, corr + 8 * (size_t)cap_offsets[b], nullptr);
  }
  (void)n_threads;
  return 0;
}

/* Batched driver with the product's layout: scans table + per-block (source scan, target scan). */
int orc_corr_search_batch(const float *xy, const float *normals, const int32_t *offsets,
                          const int32_t *block_src, const int32_t *block_tgt, int32_t n_blocks,
                          const float *pose_aff, float outlier_threshold, const int64_t *cap_offsets,
                          float *corr, int32_t *counts, int32_t n_threads) {
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel for schedule(dynamic, 8)
#endif
  for (int32_t b = 0; b < n_blocks; b++) {
    const int32_t s = block_src[b], t = block_tgt[b];
    counts[b] = orc_corr_search_block(xy + 2 * (size_t)offsets[s], normals + 2 * (size_t)offsets[s],
                                      offsets[s + 1] - offsets[s], xy + 2 * (size_t)offsets[t],
                                      normals + 2 * (size_t)offsets[t], offsets[t + 1] - offsets[t],
                                      pose_aff + 4 * (size_t)s, pose_aff + 4 * (size_t)t, outlier_threshold,
                                      corr + 8 * (size_t)cap_offsets[b], nullptr);
  }
  (void)n_threads;
  return 0;
}

}  // extern "C"


// =====================================================================================
// Loop-closure candidate gating -- CPU restatement of
//   src/loop_closure/lc_candidate_filter.cc:22-51  ComputeMean, ComputeScatterMatrixScore
// (float accumulators, point order; built with -ffp-contract=off so products and sums round individually, as Eigen's
// Vector2f / Matrix2f expressions do on baseline x86-64).  The reference takes the eigenvalues with Eigen's iterative
// EigenSolver<Matrix2f>; here, as in the product, they are the closed form of the symmetric 2 x 2 matrix in double
// (agreement to float rounding; Eigen is not in the image).  The pair gate is the build's geometric stand-in for
// LCMatcher::GetPossibleMatches (lc_matcher.cc:59-74).
extern "C" {

double orc_scatter_matrix_score(const float *xy, int32_t n) {
  float mx = 0.f, my = 0.f;
  for (int32_t i = 0; i < n; i++) { mx += xy[2 * i]; my += xy[2 * i + 1]; }
  const float inv = (float)(1.0 / (double)n);
  mx = inv * mx; my = inv * my;
  float a = 0.f, b = 0.f, c = 0.f, d = 0.f;
  for (int32_t i = 0; i < n; i++) {
    const float dx = xy[2 * i] - mx, dy = xy[2 * i + 1] - my;
    const float xx = dx * dx, xy_ = dx * dy, yx = dy * dx, yy = dy * dy;
    a += xx; b += xy_; c += yx; d += yy;
  }
  const double A = a, B = b, C = c, D = d;
  const double half_tr = 0.5 * (A + D), half_df = 0.5 * (A - D);
  const double p1 = half_df * half_df, p2 = B * C;
  const double disc = p1 + p2;
  const double root = std::sqrt(disc < 0.0 ? 0.0 : disc);
  const double e1 = half_tr + root, e2 = half_tr - root;
  const double lo = e1 < e2 ? e1 : e2, hi = e1 < e2 ? e2 : e1;
  return lo / hi;
}

void orc_scatter_matrix_scores(const float *xy, const int32_t *offsets, int32_t n_scans, double *scores) {
  for (int32_t s = 0; s < n_scans; s++)
    scores[s] = orc_scatter_matrix_score(xy + 2 * (size_t)offsets[s], offsets[s + 1] - offsets[s]);
}

void orc_pair_gate(const double *poses, const int32_t *cand, int32_t n, double max_range, int32_t min_sep, uint8_t *flags) {
  const float mr = (float)max_range;
  for (int32_t i = 0; i < n; i++)
    for (int32_t j = 0; j < n; j++) {
      const int32_t a = cand[i], b = cand[j];
      const float dx = (float)poses[3 * b] - (float)poses[3 * a], dy = (float)poses[3 * b + 1] - (float)poses[3 * a + 1];
      const float xx = dx * dx, yy = dy * dy;
      const float dist = std::sqrt(xx + yy);
      const int32_t sep = a > b ? a - b : b - a;
      flags[(size_t)i * n + j] = (a != b && sep > min_sep && dist < mr) ? 1 : 0;
    }
}

// LCMatcher::ChiSquareScore + the acceptance of GetPossibleMatches (src/loop_closure/lc_matcher.cc:50-74) for n
// (source, candidate) pairs: cov[i] is the Matrix2f GetCovarianceMatrix returned (row major), the translations are
// Vector2f (slam_util.h:48-53).  Eigen is not in the image: Matrix2f::inverse() is restated as its published
// fixed-size closed form (Eigen/src/LU/InverseImpl.h, compute_inverse<_, _, 2>: adjugate times 1 / (m00 m11 - m10 m01)),
// and d^T * inv * d as the row vector (d^T inv) dotted with d -- float operations rounded one by one
// (-ffp-contract=off).
void orc_chi_square_gate(const double *poses, const int32_t *src, const int32_t *tgt, const float *cov, int32_t n,
                         double max_score, double *scores, uint8_t *flags) {
  for (int32_t t = 0; t < n; t++) {
    const int32_t a = src[t], b = tgt[t];
    const float *m = cov + 4 * (size_t)t;
    const float d0 = (float)poses[3 * b] - (float)poses[3 * a], d1 = (float)poses[3 * b + 1] - (float)poses[3 * a + 1];
    const float p0 = m[0] * m[3], p1 = m[2] * m[1];
    const float det = p0 - p1;
    const float invdet = 1.0f / det;
    const float i00 = m[3] * invdet, i10 = -m[2] * invdet, i01 = -m[1] * invdet, i11 = m[0] * invdet;
    const float a0 = d0 * i00, a1 = d1 * i10, b0 = d0 * i01, b1 = d1 * i11;
    const float r0 = a0 + a1, r1 = b0 + b1;
    const float c0 = r0 * d0, c1 = r1 * d1;
    const double score = (double)(c0 + c1);
    scores[t] = score;
    flags[t] = (a != b && score < max_score) ? 1 : 0;
  }
}

}  // extern "C"
