// nhip_csm_shared.h -- what the two kernels that perform every add of the (theta, x, y) correlation share:
// csm_correlate_kernel (8-bit cells, nhip_csm.hip) and csm_correlate16_kernel (16-bit cells, nhip_csm16.hip).
#pragma once
#include "nhip_common.h"

namespace nhip {
namespace csm {

struct CsmParams {
  const float2 *xy;
  const int32_t *offsets;
  const uint8_t *grids;
  const int32_t *pair_src;
  const int32_t *pair_slot;
  const double *rot0_cs;
  const double *delta_cs;
  const int32_t *pair_origin;  // optional (x, y) cell offset of each pair's search centre
  unsigned long long *keys;
  int32_t *volume;  // full score volume (scores kernel only)
  int32_t n_pairs, n_theta, nx, ny, hx, hy, npbx, npby;
  int32_t S, pad, pitch, rows, max_shift;
  int32_t single_src, single_slot;  // scores kernel: the one pair
  int32_t single_ox, single_oy;
  int32_t dense;  // 1: ignore the skip maps (every strip is added, zero or not)
  int32_t tile_rows, n_tiles;  // csm_small_plane_kernel: rows of the plane of translations per workgroup, workgroups per rotation
  IdBounds ids;   // counts the pairs' scan ids and grid slots are checked against (nhip_common.h)
  int64_t grid_bytes, slot_bytes;
  double res, inv_res;
};

// Stored-grid coordinates (row, col) of the top-left cell of point q's window under rotation
// (cf, sf), packed (row << 16) | col.  Spec: rotate in float with individually rounded
// products (Eigen Affine2f * Vector2f on baseline x86-64: no FMA), cell = S/2 +
// floor(double(v) / res) (cimg_debug.h:31-37).  Cells are clamped to [-h-1, S+h]: beyond that
// range every lookup of the window falls on the zero border, and so does the clamped window.
__device__ __forceinline__ uint32_t window_cell(float2 q, float cf, float sf, const CsmParams &P,
                                                int32_t ox, int32_t oy, int32_t cx, int32_t cy) {
  const float xr = __fsub_rn(__fmul_rn(cf, q.x), __fmul_rn(sf, q.y));
  const float yr = __fadd_rn(__fmul_rn(sf, q.x), __fmul_rn(cf, q.y));
  long col = -P.hx - 1, row = -P.hy - 1;  // non-finite points score nothing
  if ((fabsf(xr) < 1e9f) && (fabsf(yr) < 1e9f)) {
    const long half = P.S / 2;
    col = half + (long)floor_quotient((double)xr, P.res, P.inv_res) + cx;
    row = half + (long)floor_quotient((double)yr, P.res, P.inv_res) + cy;
    col = col < -P.hx - 1 ? -P.hx - 1 : (col > P.S + P.hx ? P.S + P.hx : col);
    row = row < -P.hy - 1 ? -P.hy - 1 : (row > P.S + P.hy ? P.S + P.hy : row);
  }
  const uint32_t pcol = (uint32_t)(col - P.hx + ox + P.pad);
  const uint32_t prow = (uint32_t)(row - P.hy + oy + P.pad);
  return (prow << 16) | pcol;
}

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int m) {
  uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
  lo = __shfl_xor(lo, m, 64);
  hi = __shfl_xor(hi, m, 64);
  return ((unsigned long long)hi << 32) | lo;
}

// Where to put point j inside a fresh tile: ahead of the direction the beam sweep is moving.
__device__ __forceinline__ int32_t place(int32_t here, int32_t ahead, int32_t span) {
  const int32_t d = ahead - here;
  const int32_t off = d > 2 ? span / 8 : (d < -2 ? span - span / 8 : span / 2);
  const int32_t a = here - off;
  return a < 0 ? 0 : a;
}

}  // namespace csm
}  // namespace nhip
