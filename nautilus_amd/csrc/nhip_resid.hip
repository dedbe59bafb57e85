// nhip_resid.hip -- K4: batched evaluation of nautilus's Ceres cost functors on gfx950.
//
// Replaces the bodies of (src/optimization/slam_residuals.h)
//   LIDARNormalResidual::operator()  :65-89     LIDARPointResidual::operator()  :124-145
//   PointToLineResidual::operator()  :180-200   OdometryResidual::operator()    :18-40
// and the Jacobians ceres::AutoDiffCostFunction<..., 3, 3> derives from them
// (row-major num_residuals x 3 per parameter block).
//
// LIDAR kernels are pure streaming (fp64 math on fp32 inputs, ~60 flop per 144 B): one lane
// per correspondence, 16-byte loads, Jacobian rows staged through LDS so every store
// instruction writes 1 KiB of consecutive bytes.  Closed-form Jacobians:
//   q = S2T p,  u = L p (L = linear part of S2T)
//   dq/dt_s = Linv,  dq/dtheta_s = (-u_y, u_x),  dq/dt_t = -Linv,  dq/dtheta_t = (q_y, -q_x)
#include "nhip_common.h"

namespace nhip {

namespace {

constexpr int RT = 256;

// Per-block constants: S2T = inverse(A(target_pose)) * A(source_pose), slam_residuals.h:70-74,
// with Eigen's Affine-mode inverse (general 2x2 inverse of the linear part).
// consts[8] = { l00, l01, l10, l11, tx, ty, i00 (= Linv(0,0)), i01 (= Linv(0,1)) }
__global__ void resid_block_consts_kernel(const int32_t *__restrict__ block_src,
                                          const int32_t *__restrict__ block_tgt, int32_t n_blocks,
                                          const double *__restrict__ poses, int32_t n_poses,
                                          double *__restrict__ consts, uint32_t *__restrict__ status) {
  const int32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n_blocks) return;
  // (pose indices from device memory: one outside [0, n_poses) is reported and read as pose 0 -- never dereferenced)
  int32_t is = block_src[b], it = block_tgt[b];
  if (!id_in(is, n_poses) || !id_in(it, n_poses)) {
    flag_bad_id(status, BAD_POSE_ID, id_in(is, n_poses) ? it : is, b);
    is = it = 0;
  }
  const double *ps = poses + 3 * (size_t)is;
  const double *pt = poses + 3 * (size_t)it;
  const double cs = cos(ps[2]), ss = sin(ps[2]);
  const double ct = cos(pt[2]), st = sin(pt[2]);
  // A_t = [ct -st x_t; st ct y_t]; inverse: adj / det
  const double det = ct * ct - st * (-st);
  const double invdet = 1.0 / det;
  const double i00 = ct * invdet, i10 = -st * invdet, i01 = st * invdet, i11 = ct * invdet;
  const double itx = -(i00 * pt[0] + i01 * pt[1]);
  const double ity = -(i10 * pt[0] + i11 * pt[1]);
  double *c = consts + 8 * (size_t)b;
  c[0] = i00 * cs + i01 * ss;
  c[1] = i00 * (-ss) + i01 * cs;
  c[2] = i10 * cs + i11 * ss;
  c[3] = i10 * (-ss) + i11 * cs;
  c[4] = i00 * ps[0] + i01 * ps[1] + itx;
  c[5] = i10 * ps[0] + i11 * ps[1] + ity;
  c[6] = i00;
  c[7] = i01;
}

// One lane per correspondence.  WANT_J selects residual-only vs residual + both Jacobians;
// jac_src / jac_tgt may individually be null (Ceres passes NULL for constant blocks).
// Outputs are written once and read by another kernel or the host: streaming (nontemporal) stores.
typedef double dbl2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_stream(double2 *p, const double2 &v) {
#ifdef NHIP_PLAIN_STORES
  *p = v;
#else
  dbl2_t t = {v.x, v.y};
  __builtin_nontemporal_store(t, reinterpret_cast<dbl2_t *>(p));
#endif
}

template <int KIND, bool WANT_J>
__global__ __launch_bounds__(RT) void resid_lidar_kernel(
    const float4 *__restrict__ corr, const int32_t *__restrict__ corr_block, int64_t n_corr,
    const double *__restrict__ consts, double2 *__restrict__ residuals,
    double2 *__restrict__ jac_src, double2 *__restrict__ jac_tgt, double2 *__restrict__ jac_tgt_theta,
    int32_t block_base, int32_t n_blocks, uint32_t *__restrict__ status, double2 *__restrict__ q_out) {
  // q_out (optional): the transformed source point q = S2T p_s per correspondence.  With the block's constants (Linv, t) it
  // determines every Jacobian entry (u = q - t; the closed forms below), so a host that holds the correspondences rebuilds
  // both Jacobians while it copies its slice: 32 instead of 80 bytes per correspondence cross PCIe (nhip_resid_batch_eval_q)
  __shared__ double2 s_j[WANT_J ? 2 * 3 * RT : 1];
  const int64_t i0 = (int64_t)blockIdx.x * RT;
  const int64_t i = i0 + threadIdx.x;
  const bool in_range = i < n_corr;
  bool live = in_range;
  // (a block id from device memory outside the batch's blocks: reported, the row's residuals and Jacobians are zero)
  int32_t blk = 0;
  if (live) {
    blk = corr_block[i] - block_base;
    if (!id_in(blk, n_blocks)) {
      flag_bad_id(status, BAD_BLOCK_ID, blk + block_base, (int32_t)(i < 0x7fffffff ? i : 0x7fffffff));
      live = false;
    }
  }
  double r0 = 0, r1 = 0, q_x = 0, q_y = 0;
  double js[6] = {0, 0, 0, 0, 0, 0}, jt[6] = {0, 0, 0, 0, 0, 0};
  if (live) {
    const float4 a = corr[2 * i];      // source point, target point
    const float4 n = corr[2 * i + 1];  // source normal, target normal
    const double2 *c = reinterpret_cast<const double2 *>(consts + 8 * (size_t)blk);
    const double2 c01 = c[0], c23 = c[1], c45 = c[2], c67 = c[3];
    const double l00 = c01.x, l01 = c01.y, l10 = c23.x, l11 = c23.y;
    const double i00 = c67.x, i01 = c67.y, i10 = -i01, i11 = i00;
    const double px = a.x, py = a.y, tx = a.z, ty = a.w;
    const double ux = l00 * px + l01 * py, uy = l10 * px + l11 * py;
    const double qx = ux + c45.x, qy = uy + c45.y;
    q_x = qx;
    q_y = qy;
    if (KIND == NHIP_LIDAR_NORMAL) {
      const double nsx = n.x, nsy = n.y, ntx = n.z, nty = n.w;
      const double ex = qx - tx, ey = qy - ty;
      r0 = ntx * ex + nty * ey;        // target_normal . (S2T p_s - p_t)   :80-82
      r1 = nsx * (-ex) + nsy * (-ey);  // source_normal . (p_t - S2T p_s)   :83-84
      if (WANT_J) {
        // dq/dx_s = (i00, i10), dq/dy_s = (i01, i11), dq/dth_s = (-uy, ux)
        js[0] = ntx * i00 + nty * i10;
        js[1] = ntx * i01 + nty * i11;
        js[2] = ntx * (-uy) + nty * ux;
        js[3] = -(nsx * i00 + nsy * i10);
        js[4] = -(nsx * i01 + nsy * i11);
        js[5] = -(nsx * (-uy) + nsy * ux);
        // dq/dt_t = -Linv, dq/dth_t = (qy, -qx)
        jt[0] = -js[0];
        jt[1] = -js[1];
        jt[2] = ntx * qy - nty * qx;
        jt[3] = -js[3];
        jt[4] = -js[4];
        jt[5] = -(nsx * qy - nsy * qx);
      }
    } else {
      r0 = tx - qx;  // target - S2T source   :139-142
      r1 = ty - qy;
      if (WANT_J) {
        js[0] = -i00; js[1] = -i01; js[2] = uy;
        js[3] = -i10; js[4] = -i11; js[5] = -ux;
        jt[0] = i00;  jt[1] = i01;  jt[2] = -qy;
        jt[3] = i10;  jt[4] = i11;  jt[5] = qx;
      }
    }
  }
  if (in_range) {
    store_stream(&residuals[i], make_double2(r0, r1));
    if (q_out) store_stream(&q_out[i], make_double2(q_x, q_y));
    // the theta column of the target Jacobian on its own: its x, y columns are the negated x, y columns of the
    // source Jacobian (dq/dt_t = -dq/dt_s), so a host that rebuilds them needs only these two values
    if (WANT_J && jac_tgt_theta) store_stream(&jac_tgt_theta[i], make_double2(jt[2], jt[5]));
  }
  if (WANT_J) {
    // transpose through LDS: lane t holds 3 double2 per Jacobian; the block's 3*RT double2
    // are stored as consecutive 16-byte pieces, one per lane per instruction.
    double2 *ss = s_j, *st = s_j + 3 * RT;
    const int t = threadIdx.x;
    ss[3 * t + 0] = make_double2(js[0], js[1]);
    ss[3 * t + 1] = make_double2(js[2], js[3]);
    ss[3 * t + 2] = make_double2(js[4], js[5]);
    st[3 * t + 0] = make_double2(jt[0], jt[1]);
    st[3 * t + 1] = make_double2(jt[2], jt[3]);
    st[3 * t + 2] = make_double2(jt[4], jt[5]);
    __syncthreads();
    const int64_t lim = (n_corr - i0 < RT ? n_corr - i0 : RT) * 3;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const int64_t e = (int64_t)k * RT + t;
      if (e < lim) {
        if (jac_src) store_stream(&jac_src[3 * i0 + e], ss[e]);
        if (jac_tgt) store_stream(&jac_tgt[3 * i0 + e], st[e]);
      }
    }
  }
}

// Per-block normal equations: one workgroup per block, every lane runs the
// same per-correspondence math as resid_lidar_kernel and keeps 28 fp64 partial sums (upper
// triangle of J^T J, J^T r, r^T r); wave64 shuffles, then LDS across the 4 waves.  Output is
// 224 B per block instead of 112 B per correspondence.
template <int KIND>
__global__ __launch_bounds__(RT) void resid_normal_eq_kernel(const float4 *__restrict__ corr,
                                                             const int32_t *__restrict__ block_offsets,
                                                             const double *__restrict__ consts,
                                                             double *__restrict__ out) {
  __shared__ double s_part[RT / 64][28];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int32_t o = block_offsets[b], n = block_offsets[b + 1] - o;
  const double *c = consts + 8 * (size_t)b;
  const double l00 = c[0], l01 = c[1], l10 = c[2], l11 = c[3], ctx = c[4], cty = c[5];
  const double i00 = c[6], i01 = c[7], i10 = -i01, i11 = i00;
  double acc[28];
#pragma unroll
  for (int k = 0; k < 28; k++) acc[k] = 0.0;
  // (the block's rows are loaded LOADS at a time ahead of the fp64 math: 1081 rows are only ~4 per lane)
  constexpr int LOADS = 5;
  for (int32_t i0 = tid; i0 < n; i0 += RT * LOADS) {
    float4 av[LOADS], nv[LOADS];
#pragma unroll
    for (int u = 0; u < LOADS; u++) {
      const int32_t i = i0 + u * RT;
      if (i < n) {
        av[u] = corr[2 * (size_t)(o + i)];
        nv[u] = corr[2 * (size_t)(o + i) + 1];
      }
    }
#pragma unroll
    for (int u = 0; u < LOADS; u++) {
    if (i0 + u * RT >= n) break;
    const float4 a = av[u];
    const float4 nn = nv[u];
    const double px = a.x, py = a.y, tx = a.z, ty = a.w;
    const double ux = l00 * px + l01 * py, uy = l10 * px + l11 * py;
    const double qx = ux + ctx, qy = uy + cty;
    double r[2], J[2][6];
    if (KIND == NHIP_LIDAR_NORMAL) {
      const double nsx = nn.x, nsy = nn.y, ntx = nn.z, nty = nn.w;
      const double ex = qx - tx, ey = qy - ty;
      r[0] = ntx * ex + nty * ey;
      r[1] = -(nsx * ex + nsy * ey);
      J[0][0] = ntx * i00 + nty * i10;
      J[0][1] = ntx * i01 + nty * i11;
      J[0][2] = ntx * (-uy) + nty * ux;
      J[1][0] = -(nsx * i00 + nsy * i10);
      J[1][1] = -(nsx * i01 + nsy * i11);
      J[1][2] = -(nsx * (-uy) + nsy * ux);
      J[0][3] = -J[0][0];
      J[0][4] = -J[0][1];
      J[0][5] = ntx * qy - nty * qx;
      J[1][3] = -J[1][0];
      J[1][4] = -J[1][1];
      J[1][5] = -(nsx * qy - nsy * qx);
    } else {
      r[0] = tx - qx;
      r[1] = ty - qy;
      J[0][0] = -i00; J[0][1] = -i01; J[0][2] = uy;
      J[1][0] = -i10; J[1][1] = -i11; J[1][2] = -ux;
      J[0][3] = i00;  J[0][4] = i01;  J[0][5] = -qy;
      J[1][3] = i10;  J[1][4] = i11;  J[1][5] = qx;
    }
    int k = 0;
#pragma unroll
    for (int p = 0; p < 6; p++)
#pragma unroll
      for (int q = p; q < 6; q++) acc[k++] += J[0][p] * J[0][q] + J[1][p] * J[1][q];
#pragma unroll
    for (int p = 0; p < 6; p++) acc[21 + p] += J[0][p] * r[0] + J[1][p] * r[1];
    acc[27] += r[0] * r[0] + r[1] * r[1];
    }
  }
  // wave reduction as a reduce-scatter: at each butterfly step a lane hands half of its values to
  // its partner and keeps the other half, so 32 -> 16 -> 8 -> 4 -> 2 -> 1 values per lane and 32
  // fp64 shuffles in all (a plain all-reduce of 28 values takes 168); lane l ends with value l >> 1.
  {
    double v[32];
#pragma unroll
    for (int k = 0; k < 32; k++) v[k] = k < 28 ? acc[k] : 0.0;
    const int lane = tid & 63;
#define NHIP_RS_STEP(M, N)                                          \
  {                                                                 \
    const bool lo = (lane & (M)) == 0;                              \
    _Pragma("unroll") for (int j = 0; j < (N) / 2; j++) {           \
      const double send = lo ? v[j + (N) / 2] : v[j];               \
      const double recv = __shfl_xor(send, (M), 64);                \
      v[j] = (lo ? v[j] : v[j + (N) / 2]) + recv;                   \
    }                                                               \
  }
    NHIP_RS_STEP(32, 32)
    NHIP_RS_STEP(16, 16)
    NHIP_RS_STEP(8, 8)
    NHIP_RS_STEP(4, 4)
    NHIP_RS_STEP(2, 2)
#undef NHIP_RS_STEP
    const double total = v[0] + __shfl_xor(v[0], 1, 64);
    if ((lane & 1) == 0 && (lane >> 1) < 28) s_part[tid >> 6][lane >> 1] = total;
  }
  __syncthreads();
  if (tid < 28) {
    double v = 0.0;
#pragma unroll
    for (int w = 0; w < RT / 64; w++) v += s_part[w][tid];
    out[28 * (size_t)b + tid] = v;
  }
}

// ---------------------------------------------------------------- forward-mode duals
// PointToLineResidual is branchy (DistanceToLineSegment, slam_util.h:92-110) and its
// segment moves with line_pose, so it is differentiated the way Ceres does it: 6 partials
// carried alongside the value.
struct Dual6 {
  double a;
  double v[6];
};
__device__ __forceinline__ Dual6 dconst(double s) {
  Dual6 r; r.a = s;
#pragma unroll
  for (int i = 0; i < 6; i++) r.v[i] = 0.0;
  return r;
}
__device__ __forceinline__ Dual6 dvar(double s, int k) { Dual6 r = dconst(s); r.v[k] = 1.0; return r; }
__device__ __forceinline__ Dual6 operator+(const Dual6 &f, const Dual6 &g) {
  Dual6 r; r.a = f.a + g.a;
#pragma unroll
  for (int i = 0; i < 6; i++) r.v[i] = f.v[i] + g.v[i];
  return r;
}
__device__ __forceinline__ Dual6 operator-(const Dual6 &f, const Dual6 &g) {
  Dual6 r; r.a = f.a - g.a;
#pragma unroll
  for (int i = 0; i < 6; i++) r.v[i] = f.v[i] - g.v[i];
  return r;
}
__device__ __forceinline__ Dual6 operator-(const Dual6 &f) {
  Dual6 r; r.a = -f.a;
#pragma unroll
  for (int i = 0; i < 6; i++) r.v[i] = -f.v[i];
  return r;
}
__device__ __forceinline__ Dual6 operator*(const Dual6 &f, const Dual6 &g) {
  Dual6 r; r.a = f.a * g.a;
#pragma unroll
  for (int i = 0; i < 6; i++) r.v[i] = f.a * g.v[i] + f.v[i] * g.a;
  return r;
}
__device__ __forceinline__ Dual6 operator/(const Dual6 &f, const Dual6 &g) {
  Dual6 r; const double gi = 1.0 / g.a; const double fg = f.a * gi; r.a = fg;
#pragma unroll
  for (int i = 0; i < 6; i++) r.v[i] = (f.v[i] - fg * g.v[i]) * gi;
  return r;
}
__device__ __forceinline__ Dual6 dsqrt(const Dual6 &f) {
  Dual6 r; r.a = sqrt(f.a); const double t = 1.0 / (2.0 * r.a);
#pragma unroll
  for (int i = 0; i < 6; i++) r.v[i] = t * f.v[i];
  return r;
}
__device__ __forceinline__ bool between(double v, double a, double b) {  // slam_util.h:87-89
  return (v >= a && v <= b) || (v >= b && v <= a);
}

__global__ __launch_bounds__(RT) void resid_point_to_line_kernel(
    const float4 *__restrict__ segments, const float2 *__restrict__ points,
    const int32_t *__restrict__ point_block, int64_t n_points,
    const int32_t *__restrict__ block_pose, const int32_t *__restrict__ block_line,
    const double *__restrict__ poses, const double *__restrict__ line_poses,
    double *__restrict__ residuals, double *__restrict__ jac_pose, double *__restrict__ jac_line, int32_t n_blocks,
    int32_t n_poses, int32_t n_line_poses, uint32_t *__restrict__ status) {
  const int64_t i = (int64_t)blockIdx.x * RT + threadIdx.x;
  if (i >= n_points) return;
  // (ids from device memory: a block id outside the blocks, or a block whose pose / line-pose index is outside the arrays,
  //  is reported and its point's residual and Jacobians are zero)
  const int32_t b = point_block[i];
  const bool b_ok = id_in(b, n_blocks);
  const int32_t ip = b_ok ? block_pose[b] : -1, il = b_ok ? block_line[b] : -1;
  if (!b_ok || !id_in(ip, n_poses) || !id_in(il, n_line_poses)) {
    flag_bad_id(status, b_ok ? BAD_POSE_ID : BAD_BLOCK_ID, b_ok ? (id_in(ip, n_poses) ? il : ip) : b, (int32_t)(i < 0x7fffffff ? i : 0x7fffffff));
    residuals[i] = 0.0;
    if (jac_pose) { jac_pose[3 * i] = 0.0; jac_pose[3 * i + 1] = 0.0; jac_pose[3 * i + 2] = 0.0; }
    if (jac_line) { jac_line[3 * i] = 0.0; jac_line[3 * i + 1] = 0.0; jac_line[3 * i + 2] = 0.0; }
    return;
  }
  const double *pp = poses + 3 * (size_t)ip;
  const double *lp = line_poses + 3 * (size_t)il;
  const float4 sg = segments[b];
  const float2 pt = points[i];
  // pose_to_world = T(x, y) * R(theta), slam_util.h:20-28; partials 0..2 = pose, 3..5 = line_pose
  const Dual6 x = dvar(pp[0], 0), y = dvar(pp[1], 1);
  Dual6 c = dconst(cos(pp[2])), s = dconst(sin(pp[2]));
  c.v[2] = -s.a; s.v[2] = c.a;
  const Dual6 lx = dvar(lp[0], 3), ly = dvar(lp[1], 4);
  Dual6 lc = dconst(cos(lp[2])), ls = dconst(sin(lp[2]));
  lc.v[5] = -ls.a; ls.v[5] = lc.a;
  const Dual6 sx0 = lc * dconst(sg.x) - ls * dconst(sg.y) + lx;  // line_start, :186-187
  const Dual6 sy0 = ls * dconst(sg.x) + lc * dconst(sg.y) + ly;
  const Dual6 sx1 = lc * dconst(sg.z) - ls * dconst(sg.w) + lx;  // line_end, :188
  const Dual6 sy1 = ls * dconst(sg.z) + lc * dconst(sg.w) + ly;
  const Dual6 px = c * dconst(pt.x) - s * dconst(pt.y) + x;      // pointT, :194
  const Dual6 py = s * dconst(pt.x) + c * dconst(pt.y) + y;
  // Hyperplane::Through(start, end): n = unitOrthogonal(end - start), offset = -n . start
  const Dual6 dx = sx1 - sx0, dy = sy1 - sy0;
  Dual6 nx = -dy, ny = dx;
  const Dual6 len = dsqrt(nx * nx + ny * ny);
  nx = nx / len; ny = ny / len;
  const Dual6 off = -(sx0 * nx + sy0 * ny);
  const Dual6 sd = nx * px + ny * py + off;  // signedDistance
  const Dual6 prx = px - sd * nx, pry = py - sd * ny;  // projection
  Dual6 d;
  if (between(prx.a, sx0.a, sx1.a) && between(pry.a, sy0.a, sy1.a)) {
    d = sd.a < 0.0 ? -sd : sd;  // absDistance
  } else {
    const Dual6 ax = px - sx0, ay = py - sy0, bx = px - sx1, by = py - sy1;
    const Dual6 ds = dsqrt(ax * ax + ay * ay), de = dsqrt(bx * bx + by * by);
    d = (de.a < ds.a) ? de : ds;  // std::min<T>(dist_to_start, dist_to_endpoint)
  }
  residuals[i] = d.a;
  if (jac_pose) { jac_pose[3 * i] = d.v[0]; jac_pose[3 * i + 1] = d.v[1]; jac_pose[3 * i + 2] = d.v[2]; }
  if (jac_line) { jac_line[3 * i] = d.v[3]; jac_line[3 * i + 1] = d.v[4]; jac_line[3 * i + 2] = d.v[5]; }
}

// OdometryResidual: r = (tw (Ti + T_odom - Tj), rw atan2(sin d, cos d)), d = th_i + R_odom - th_j.
// d/dd atan2(sin d, cos d) = (cos^2 + sin^2) / (cos^2 + sin^2) evaluated as Ceres does:
// tmp = 1 / (cos^2 + sin^2); v = tmp * (-sin * (-sin) + cos * cos).
__global__ void resid_odometry_kernel(const float2 *__restrict__ t_odom,
                                      const float *__restrict__ r_odom,
                                      const int32_t *__restrict__ pose_i,
                                      const int32_t *__restrict__ pose_j, int32_t n, double tw,
                                      double rw, const double *__restrict__ poses,
                                      double *__restrict__ residuals, double *__restrict__ ji,
                                      double *__restrict__ jj, int32_t n_poses, uint32_t *__restrict__ status) {
  const int32_t f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n) return;
  // (pose indices from device memory: a factor with one outside [0, n_poses) is reported; residuals and Jacobians zero)
  const int32_t ii = pose_i[f], ij = pose_j[f];
  if (!id_in(ii, n_poses) || !id_in(ij, n_poses)) {
    flag_bad_id(status, BAD_POSE_ID, id_in(ii, n_poses) ? ij : ii, f);
    residuals[3 * f + 0] = residuals[3 * f + 1] = residuals[3 * f + 2] = 0.0;
    if (ji) for (int q = 0; q < 9; q++) ji[9 * (size_t)f + q] = 0.0;
    if (jj) for (int q = 0; q < 9; q++) jj[9 * (size_t)f + q] = 0.0;
    return;
  }
  const double *pi = poses + 3 * (size_t)ii;
  const double *pj = poses + 3 * (size_t)ij;
  const float2 t = t_odom[f];
  const double ex = pi[0] + (double)t.x - pj[0];
  const double ey = pi[1] + (double)t.y - pj[1];
  const double d = pi[2] + (double)r_odom[f] - pj[2];
  const double sd = sin(d), cd = cos(d);
  residuals[3 * f + 0] = tw * ex;
  residuals[3 * f + 1] = tw * ey;
  residuals[3 * f + 2] = rw * atan2(sd, cd);
  const double g = rw * ((sd * sd + cd * cd) / (cd * cd + sd * sd));
  if (ji) {
    double *J = ji + 9 * (size_t)f;
    J[0] = tw; J[1] = 0; J[2] = 0; J[3] = 0; J[4] = tw; J[5] = 0; J[6] = 0; J[7] = 0; J[8] = g;
  }
  if (jj) {
    double *J = jj + 9 * (size_t)f;
    J[0] = -tw; J[1] = 0; J[2] = 0; J[3] = 0; J[4] = -tw; J[5] = 0; J[6] = 0; J[7] = 0; J[8] = -g;
  }
}

}  // namespace

int launch_resid_lidar(int kind, const float *d_corr, const int32_t *d_corr_block, int64_t n_corr,
                       const int32_t *d_block_src, const int32_t *d_block_tgt, int32_t n_blocks,
                       const double *d_poses, int32_t n_poses, double *d_block_consts,
                       double *d_res, double *d_jsrc, double *d_jtgt, hipStream_t s, double *d_jtgt_theta,
                       int32_t block_base, double *d_q) {
  NHIP_REQUIRE(kind == NHIP_LIDAR_NORMAL || kind == NHIP_LIDAR_POINT, "resid_lidar: bad kind %d",
               kind);
  NHIP_REQUIRE(n_corr >= 0 && n_blocks >= 0 && n_poses >= 0, "resid_lidar: negative size");
  if (n_corr == 0 || n_blocks == 0) return NHIP_OK;
  hipLaunchKernelGGL(resid_block_consts_kernel, dim3((n_blocks + 255) / 256), dim3(256), 0, s,
                     d_block_src, d_block_tgt, n_blocks, d_poses, n_poses, d_block_consts, dev_status());
  const dim3 grid((uint32_t)((n_corr + RT - 1) / RT)), block(RT);
  const float4 *corr = reinterpret_cast<const float4 *>(d_corr);
  double2 *res = reinterpret_cast<double2 *>(d_res);
  double2 *js = reinterpret_cast<double2 *>(d_jsrc), *jt = reinterpret_cast<double2 *>(d_jtgt);
  const bool want_j = d_jsrc || d_jtgt || d_jtgt_theta;
  double2 *jtt = reinterpret_cast<double2 *>(d_jtgt_theta);
  double2 *qo = reinterpret_cast<double2 *>(d_q);
  timer_begin(NHIP_TIMER_RESID, s);
  if (kind == NHIP_LIDAR_NORMAL) {
    if (want_j)
      hipLaunchKernelGGL((resid_lidar_kernel<NHIP_LIDAR_NORMAL, true>), grid, block, 0, s, corr,
                         d_corr_block, n_corr, d_block_consts, res, js, jt, jtt, block_base, n_blocks, dev_status(), qo);
    else
      hipLaunchKernelGGL((resid_lidar_kernel<NHIP_LIDAR_NORMAL, false>), grid, block, 0, s, corr,
                         d_corr_block, n_corr, d_block_consts, res, js, jt, jtt, block_base, n_blocks, dev_status(), qo);
  } else {
    if (want_j)
      hipLaunchKernelGGL((resid_lidar_kernel<NHIP_LIDAR_POINT, true>), grid, block, 0, s, corr,
                         d_corr_block, n_corr, d_block_consts, res, js, jt, jtt, block_base, n_blocks, dev_status(), qo);
    else
      hipLaunchKernelGGL((resid_lidar_kernel<NHIP_LIDAR_POINT, false>), grid, block, 0, s, corr,
                         d_corr_block, n_corr, d_block_consts, res, js, jt, jtt, block_base, n_blocks, dev_status(), qo);
  }
  timer_end(NHIP_TIMER_RESID, s);
  NHIP_TRY_HIP(hipGetLastError());
  return NHIP_OK;
}

int launch_resid_normal_eq(int kind, const float *d_corr, const int32_t *d_block_offsets,
                           const int32_t *d_block_src, const int32_t *d_block_tgt, int32_t n_blocks,
                           const double *d_poses, int32_t n_poses, double *d_block_consts, double *d_out,
                           hipStream_t s) {
  NHIP_REQUIRE(kind == NHIP_LIDAR_NORMAL || kind == NHIP_LIDAR_POINT, "resid_normal_eq: bad kind %d", kind);
  NHIP_REQUIRE(n_blocks >= 0, "resid_normal_eq: negative size");
  if (n_blocks == 0) return NHIP_OK;
  hipLaunchKernelGGL(resid_block_consts_kernel, dim3((n_blocks + 255) / 256), dim3(256), 0, s,
                     d_block_src, d_block_tgt, n_blocks, d_poses, n_poses, d_block_consts, dev_status());
  const float4 *corr = reinterpret_cast<const float4 *>(d_corr);
  timer_begin(NHIP_TIMER_NORMEQ, s);
  if (kind == NHIP_LIDAR_NORMAL)
    hipLaunchKernelGGL((resid_normal_eq_kernel<NHIP_LIDAR_NORMAL>), dim3(n_blocks), dim3(RT), 0, s, corr,
                       d_block_offsets, d_block_consts, d_out);
  else
    hipLaunchKernelGGL((resid_normal_eq_kernel<NHIP_LIDAR_POINT>), dim3(n_blocks), dim3(RT), 0, s, corr,
                       d_block_offsets, d_block_consts, d_out);
  timer_end(NHIP_TIMER_NORMEQ, s);
  NHIP_TRY_HIP(hipGetLastError());
  return NHIP_OK;
}

int launch_resid_point_to_line(const float *d_segments, const float *d_points,
                               const int32_t *d_point_block, int64_t n_points,
                               const int32_t *d_block_pose, const int32_t *d_block_line,
                               int32_t n_blocks, const double *d_poses, int32_t n_poses, const double *d_line_poses,
                               int32_t n_line_poses, double *d_res, double *d_jpose, double *d_jline, hipStream_t s) {
  NHIP_REQUIRE(n_points >= 0 && n_blocks >= 0 && n_poses >= 0 && n_line_poses >= 0, "resid_point_to_line: negative size");
  if (n_points == 0) return NHIP_OK;
  hipLaunchKernelGGL(resid_point_to_line_kernel, dim3((uint32_t)((n_points + RT - 1) / RT)),
                     dim3(RT), 0, s, reinterpret_cast<const float4 *>(d_segments),
                     reinterpret_cast<const float2 *>(d_points), d_point_block, n_points,
                     d_block_pose, d_block_line, d_poses, d_line_poses, d_res, d_jpose, d_jline, n_blocks, n_poses,
                     n_line_poses, dev_status());
  NHIP_TRY_HIP(hipGetLastError());
  return NHIP_OK;
}

int launch_resid_odometry(const float *d_t_odom, const float *d_r_odom, const int32_t *d_pose_i,
                          const int32_t *d_pose_j, int32_t n_factors, double tw, double rw,
                          const double *d_poses, int32_t n_poses, double *d_res, double *d_ji, double *d_jj,
                          hipStream_t s) {
  NHIP_REQUIRE(n_factors >= 0 && n_poses >= 0, "resid_odometry: negative size");
  if (n_factors == 0) return NHIP_OK;
  hipLaunchKernelGGL(resid_odometry_kernel, dim3((n_factors + 255) / 256), dim3(256), 0, s,
                     reinterpret_cast<const float2 *>(d_t_odom), d_r_odom, d_pose_i, d_pose_j,
                     n_factors, tw, rw, d_poses, d_res, d_ji, d_jj, n_poses, dev_status());
  NHIP_TRY_HIP(hipGetLastError());
  return NHIP_OK;
}

}  // namespace nhip
