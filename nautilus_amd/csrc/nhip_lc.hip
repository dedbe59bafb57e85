// nhip_lc.hip -- the step before the scan matcher: which scans are loop-closure
// candidates and which pairs of them go to the matcher.
//
//   lc_scatter_score_kernel   LCCandidateFilter's ComputeScatterMatrixScore for EVERY scan at once
//                             (src/loop_closure/lc_candidate_filter.cc:22-51): float mean, float scatter matrix,
//                             min / max eigenvalue.
//   lc_pair_gate_kernel       a geometric gate over all ordered candidate pairs, in place of the per-pair
//                             ceres::Covariance + chi-square test of LCMatcher (src/loop_closure/lc_matcher.cc:28-74),
//                             whose cost is a sparse factorisation per pair.
//   lc_chi_square_kernel      LCMatcher's own test given the cross-covariance blocks: ChiSquareScore and the
//                             `score < 5000` acceptance of GetPossibleMatches (lc_matcher.cc:50-74) for every
//                             (source, candidate) pair at once.  The covariance blocks themselves come from the
//                             host's sparse solve (ceres::Covariance in the reference, :28-46).
//
// The reference sums in float, in point order (Eigen::Vector2f / Matrix2f accumulators).  One lane per scan walks
// its points in that order with individually rounded float operations, so the sums are the reference's bit for bit
// (a tree reduction would differ in the last bits: a score near the 0.70 threshold could flip).  The eigenvalues of
// the symmetric 2 x 2 matrix are taken in closed form in double (the reference runs Eigen's iterative real
// EigenSolver in float; the two agree to float rounding).  1,000 scans x 1081 points are 8.6 MB: the pass is short
// whichever way it is parallelised.
#include "nhip_common.h"

namespace nhip {

namespace {

__global__ __launch_bounds__(64) void lc_scatter_score_kernel(const float2 *__restrict__ xy, const int32_t *__restrict__ offsets,
                                                               int32_t n_scans, double *__restrict__ scores) {
  const int32_t s = blockIdx.x * 64 + threadIdx.x;
  if (s >= n_scans) return;
  const int32_t beg = offsets[s], n = offsets[s + 1] - beg;
  // ComputeMean: mean_vector += p, then (1.0 / size) * mean_vector with the double factor converted to float
  float mx = 0.f, my = 0.f;
  for (int32_t i = 0; i < n; i++) {
    const float2 p = xy[beg + i];
    mx = __fadd_rn(mx, p.x);
    my = __fadd_rn(my, p.y);
  }
  const float inv = (float)(1.0 / (double)n);
  mx = __fmul_rn(inv, mx);
  my = __fmul_rn(inv, my);
  // scatter_matrix += (p - mean) * (p - mean)^T
  float a = 0.f, b = 0.f, c = 0.f, d = 0.f;
  for (int32_t i = 0; i < n; i++) {
    const float2 p = xy[beg + i];
    const float dx = __fsub_rn(p.x, mx), dy = __fsub_rn(p.y, my);
    a = __fadd_rn(a, __fmul_rn(dx, dx));
    b = __fadd_rn(b, __fmul_rn(dx, dy));
    c = __fadd_rn(c, __fmul_rn(dy, dx));
    d = __fadd_rn(d, __fmul_rn(dy, dy));
  }
  // eigenvalues of [[a, b], [c, d]] (b == c): (a + d) / 2 +- sqrt(((a - d) / 2)^2 + b c), in double
  const double A = a, B = b, Cc = c, D = d;
  const double half_tr = __dmul_rn(0.5, __dadd_rn(A, D)), half_df = __dmul_rn(0.5, __dsub_rn(A, D));
  const double disc = __dadd_rn(__dmul_rn(half_df, half_df), __dmul_rn(B, Cc));
  const double root = __dsqrt_rn(disc < 0.0 ? 0.0 : disc);
  const double e1 = __dadd_rn(half_tr, root), e2 = __dsub_rn(half_tr, root);
  const double lo = e1 < e2 ? e1 : e2, hi = e1 < e2 ? e2 : e1;
  scores[s] = __ddiv_rn(lo, hi);  // std::min(ev_1, ev_2) / std::max(ev_1, ev_2); 0 / 0 = NaN for an empty scan, as there
}

// flags[i * n + j] = 1: candidate i (source) and candidate j (target) go to the matcher.
__global__ __launch_bounds__(256) void lc_pair_gate_kernel(const double *__restrict__ poses, const int32_t *__restrict__ cand,
                                                           int32_t n, float max_range, int32_t min_sep,
                                                           uint8_t *__restrict__ flags, int32_t n_poses, uint32_t *__restrict__ status) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (int64_t)n * n) return;
  const int32_t i = (int32_t)(t / n), j = (int32_t)(t % n);
  const int32_t a = cand[i], b = cand[j];
  if (!id_in(a, n_poses) || !id_in(b, n_poses)) {  // (a candidate that is no node: reported, paired with nothing)
    flag_bad_id(status, BAD_POSE_ID, id_in(a, n_poses) ? b : a, id_in(a, n_poses) ? j : i);
    flags[t] = 0;
    return;
  }
  // GetPoseTranslation returns a Vector2f (slam_util.h:48-53): the distance is a float norm
  const float dx = __fsub_rn((float)poses[3 * b], (float)poses[3 * a]), dy = __fsub_rn((float)poses[3 * b + 1], (float)poses[3 * a + 1]);
  const float dist = __fsqrt_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
  const int32_t sep = a > b ? a - b : b - a;
  flags[t] = (a != b && sep > min_sep && dist < max_range) ? 1 : 0;
}

// ChiSquareScore (lc_matcher.cc:50-57): d = Vector2f(target) - Vector2f(source), score = d^T * cov.inverse() * d in
// float, widened to double.  Eigen's fixed-size 2 x 2 inverse is the closed form (adjugate times 1 / determinant,
// determinant m00 m11 - m10 m01); the product is (d^T inv) first, then the dot with d -- each float operation rounded
// on its own, as the expression templates evaluate them on baseline x86-64.
__global__ __launch_bounds__(256) void lc_chi_square_kernel(const double *__restrict__ poses, const int32_t *__restrict__ src,
                                                            const int32_t *__restrict__ tgt, const float *__restrict__ cov,
                                                            int32_t n, double max_score, double *__restrict__ scores,
                                                            uint8_t *__restrict__ flags, int32_t n_poses, uint32_t *__restrict__ status) {
  const int32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  const int32_t a = src[t], b = tgt[t];
  if (!id_in(a, n_poses) || !id_in(b, n_poses)) {  // (a node index outside the poses: reported, score NaN, no match)
    flag_bad_id(status, BAD_POSE_ID, id_in(a, n_poses) ? b : a, t);
    scores[t] = __longlong_as_double(0x7ff8000000000000ll);
    flags[t] = 0;
    return;
  }
  const float4 m = reinterpret_cast<const float4 *>(cov)[t];  // row major: m00 m01 m10 m11
  const float d0 = __fsub_rn((float)poses[3 * b], (float)poses[3 * a]), d1 = __fsub_rn((float)poses[3 * b + 1], (float)poses[3 * a + 1]);
  const float det = __fsub_rn(__fmul_rn(m.x, m.w), __fmul_rn(m.z, m.y));
  const float invdet = __fdiv_rn(1.0f, det);
  const float i00 = __fmul_rn(m.w, invdet), i10 = __fmul_rn(-m.z, invdet), i01 = __fmul_rn(-m.y, invdet), i11 = __fmul_rn(m.x, invdet);
  const float r0 = __fadd_rn(__fmul_rn(d0, i00), __fmul_rn(d1, i10)), r1 = __fadd_rn(__fmul_rn(d0, i01), __fmul_rn(d1, i11));
  const double score = (double)__fadd_rn(__fmul_rn(r0, d0), __fmul_rn(r1, d1));
  scores[t] = score;
  flags[t] = (a != b && score < max_score) ? 1 : 0;  // `match.node_idx == source.node_idx` is skipped (:64-66); NaN fails
}

}  // namespace

int launch_lc_chi_square(const double *d_poses, int32_t n_poses, const int32_t *d_src, const int32_t *d_tgt, const float *d_cov,
                         int32_t n, double max_score, double *d_scores, uint8_t *d_flags, hipStream_t s) {
  if (n == 0) return NHIP_OK;
  hipLaunchKernelGGL(lc_chi_square_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d_poses, d_src, d_tgt, d_cov, n, max_score,
                     d_scores, d_flags, n_poses, dev_status());
  NHIP_TRY_HIP(hipGetLastError());
  return NHIP_OK;
}

int launch_lc_scatter_scores(const float *d_xy, const int32_t *d_offsets, int32_t n_scans, double *d_scores, hipStream_t s) {
  if (n_scans == 0) return NHIP_OK;
  hipLaunchKernelGGL(lc_scatter_score_kernel, dim3((n_scans + 63) / 64), dim3(64), 0, s,
                     reinterpret_cast<const float2 *>(d_xy), d_offsets, n_scans, d_scores);
  NHIP_TRY_HIP(hipGetLastError());
  return NHIP_OK;
}

int launch_lc_pair_gate(const double *d_poses, int32_t n_poses, const int32_t *d_cand, int32_t n, double max_range,
                        int32_t min_sep, uint8_t *d_flags, hipStream_t s) {
  if (n == 0) return NHIP_OK;
  const int64_t total = (int64_t)n * n;
  NHIP_REQUIRE(total < (int64_t)0x7fffffff * 256, "lc_pair_gate: too many candidates");
  hipLaunchKernelGGL(lc_pair_gate_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, s, d_poses, d_cand, n,
                     (float)max_range, min_sep, d_flags, n_poses, dev_status());
  NHIP_TRY_HIP(hipGetLastError());
  return NHIP_OK;
}

}  // namespace nhip
