// nhip_csm.hip -- K2 + K3: exhaustive (theta, x, y) correlation and argmax on gfx950.
//
// Replaces CorrelativeScanMatcher::GetTransformation (call site
// src/optimization/solver.cc:633-638), batched over candidate pairs.
//
// Formulation (accumulator-stationary): one workgroup owns one rotation k of one pair and
// keeps an (nx x ny) plane of integer score accumulators in registers -- lane t owns 28
// consecutive x-shifts of y-shift t/3.  Points are visited one at a time by the whole
// workgroup (the point's window origin is wave-uniform, read with v_readlane), every lane
// reads its 28 window bytes as two 16-byte loads and adds them.  A point's contribution to
// the plane is the (nx x ny) window of the target grid anchored at its rotated cell, so
// consecutive lanes read consecutive bytes of grid rows (coalesced), and all arithmetic is
// integer: sums are order-independent, hence bit-exact against the CPU oracle.
//
// No bounds checks in the inner loop: grids are stored with a zero border of
// pad = 2*max_shift+4 cells, and points whose whole window misses the grid are redirected
// to the all-zero corner window (offset 0).
#include "nhip_common.h"

namespace nhip {

namespace {

constexpr int CSM_THREADS = 256;
constexpr int SEG_DW = 7;              // dwords of accumulated columns per lane
constexpr int SEG_COLS = 4 * SEG_DW;   // 28 x-shifts per lane
constexpr int SEGS = 3;                // lanes per plane row
constexpr int PB_NX = SEGS * SEG_COLS; // 84 x-shifts per plane block
constexpr int PB_NY = CSM_THREADS / SEGS;  // 85 y-shifts per plane block
constexpr int LDS_POINTS = 2048;       // window origins staged per pass

typedef uint32_t u32x4 __attribute__((ext_vector_type(4), aligned(4)));

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
  int32_t S, pad, pitch, max_shift;
  int32_t single_src, single_slot;  // scores kernel: the one pair
  int32_t single_ox, single_oy;
  int64_t grid_bytes;
  double res;
};

// Window origin (byte offset into the stored grid) of point q under rotation (cf, sf).
// Spec: rotate in float with individually rounded products (Eigen Affine2f * Vector2f on
// baseline x86-64: no FMA), cell = S/2 + floor(double(v) / res) (cimg_debug.h:31-37).
__device__ __forceinline__ int32_t window_origin(float2 q, float cf, float sf, const CsmParams &P,
                                                 int32_t ox, int32_t oy, int32_t cx, int32_t cy) {
  const float xr = __fsub_rn(__fmul_rn(cf, q.x), __fmul_rn(sf, q.y));
  const float yr = __fadd_rn(__fmul_rn(sf, q.x), __fmul_rn(cf, q.y));
  if (!(fabsf(xr) < 1e9f) || !(fabsf(yr) < 1e9f)) return 0;
  const long half = P.S / 2;
  const long col = half + (long)floor(__ddiv_rn((double)xr, P.res)) + cx;
  const long row = half + (long)floor(__ddiv_rn((double)yr, P.res)) + cy;
  // whole window outside the grid -> contributes only floor cells (0): use the zero corner
  if (col + P.hx < 0 || col - P.hx >= P.S || row + P.hy < 0 || row - P.hy >= P.S) return 0;
  return (int32_t)((row - P.hy + oy + P.pad) * P.pitch + (col - P.hx + ox + P.pad));
}

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int m) {
  uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
  lo = __shfl_xor(lo, m, 64);
  hi = __shfl_xor(hi, m, 64);
  return ((unsigned long long)hi << 32) | lo;
}

template <bool VOLUME>
__global__ __launch_bounds__(CSM_THREADS) void csm_correlate_kernel(CsmParams P) {
  __shared__ int32_t s_origin[LDS_POINTS];
  __shared__ unsigned long long s_best[CSM_THREADS / 64];

  // ---- block -> (pair, rotation, plane block); all rotations of a pair share an XCD
  const int32_t npb = P.npbx * P.npby;
  const int32_t per_pair = P.n_theta * npb;
  int32_t pair, w;
  if (VOLUME) {
    pair = 0;
    w = blockIdx.x;
  } else {
    const uint32_t bid = blockIdx.x;
    const uint32_t xcd = bid & 7u, j = bid >> 3;
    pair = (int32_t)((j / per_pair) * 8u + xcd);
    w = (int32_t)(j % per_pair);
    if (pair >= P.n_pairs) return;
  }
  const int32_t k = w / npb;
  const int32_t pb = w % npb;
  const int32_t ox = (pb % P.npbx) * PB_NX, oy = (pb / P.npbx) * PB_NY;

  const int32_t src = VOLUME ? P.single_src : P.pair_src[pair];
  const int32_t slot = VOLUME ? P.single_slot : P.pair_slot[pair];
  const int32_t beg = P.offsets[src], n_pts = P.offsets[src + 1] - beg;
  const uint8_t *grid = P.grids + (size_t)slot * P.grid_bytes;
  // search centre in cells; a centre the stored border cannot cover scores nothing (never faults)
  int32_t cx = VOLUME ? P.single_ox : (P.pair_origin ? P.pair_origin[2 * pair] : 0);
  int32_t cy = VOLUME ? P.single_oy : (P.pair_origin ? P.pair_origin[2 * pair + 1] : 0);
  const bool centre_ok = (abs(cx) + P.hx <= P.max_shift) && (abs(cy) + P.hy <= P.max_shift);
  if (!centre_ok) { cx = 0; cy = 0; }

  // rotation k: R(theta0) * R(delta_k), composed in double with individually rounded ops
  const double c0 = P.rot0_cs[2 * pair], s0 = P.rot0_cs[2 * pair + 1];
  const double cd = P.delta_cs[2 * k], sd = P.delta_cs[2 * k + 1];
  const float cf = __double2float_rn(__dsub_rn(__dmul_rn(c0, cd), __dmul_rn(s0, sd)));
  const float sf = __double2float_rn(__dadd_rn(__dmul_rn(s0, cd), __dmul_rn(c0, sd)));

  const int tid = threadIdx.x, lane = tid & 63;
  const int dy = tid / SEGS, seg = tid % SEGS;
  // lanes past the plane block's rows re-read row 0 (their sums are never used)
  const int dyc = (oy + dy < P.ny && dy < PB_NY) ? dy : 0;
  const uint32_t lane_off = (uint32_t)(dyc * P.pitch + seg * SEG_COLS);

  uint32_t acc[SEG_COLS];
#pragma unroll
  for (int i = 0; i < SEG_COLS; i++) acc[i] = 0;

  for (int32_t base = 0; base < n_pts; base += LDS_POINTS) {
    const int32_t cnt = min(n_pts - base, LDS_POINTS);
    const int32_t cnt64 = (cnt + 63) & ~63;
    __syncthreads();
    for (int32_t i = tid; i < cnt64; i += CSM_THREADS)
      s_origin[i] = (i < cnt && centre_ok)
                        ? window_origin(P.xy[beg + base + i], cf, sf, P, ox, oy, cx, cy)
                        : 0;
    __syncthreads();
    for (int32_t pb64 = 0; pb64 < cnt64; pb64 += 64) {
      const int32_t vorg = s_origin[pb64 + lane];
#pragma unroll 4
      for (int j = 0; j < 64; j++) {
        const uint32_t org = (uint32_t)__builtin_amdgcn_readlane(vorg, j);
        const uint32_t sh = org & 3u;
        const uint8_t *p = grid + (org & ~3u) + lane_off;
        const u32x4 a = *reinterpret_cast<const u32x4 *>(p);
        const u32x4 b = *reinterpret_cast<const u32x4 *>(p + 16);
        const uint32_t d[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int i = 0; i < SEG_DW; i++) {
          const uint32_t wv = __builtin_amdgcn_alignbyte(d[i + 1], d[i], sh);
          acc[4 * i + 0] += wv & 0xffu;
          acc[4 * i + 1] += (wv >> 8) & 0xffu;
          acc[4 * i + 2] += (wv >> 16) & 0xffu;
          acc[4 * i + 3] += wv >> 24;
        }
      }
    }
  }

  const int32_t iy = oy + dy;
  const bool row_ok = (dy < PB_NY) && (iy < P.ny);
  if (VOLUME) {
    if (row_ok) {
#pragma unroll
      for (int i = 0; i < SEG_COLS; i++) {
        const int32_t ix = ox + seg * SEG_COLS + i;
        if (ix < P.nx) P.volume[((size_t)k * P.nx + ix) * P.ny + iy] = (int32_t)acc[i];
      }
    }
    return;
  }

  // ---- K3: argmax with deterministic tie-break (smallest linear index wins)
  unsigned long long best = 0ull;
  if (row_ok) {
#pragma unroll
    for (int i = 0; i < SEG_COLS; i++) {
      const int32_t ix = ox + seg * SEG_COLS + i;
      if (ix < P.nx) {
        const uint32_t lin = (uint32_t)((k * P.nx + ix) * P.ny + iy);
        const unsigned long long key = ((unsigned long long)acc[i] << 32) | (0xffffffffu - lin);
        best = key > best ? key : best;
      }
    }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const unsigned long long o = shfl_xor_u64(best, m);
    best = o > best ? o : best;
  }
  if (lane == 0) s_best[tid >> 6] = best;
  __syncthreads();
  if (tid == 0) {
#pragma unroll
    for (int i = 1; i < CSM_THREADS / 64; i++) best = s_best[i] > best ? s_best[i] : best;
    atomicMax(&P.keys[pair], best);
  }
}

__global__ void csm_finalize_kernel(const unsigned long long *__restrict__ keys,
                                    const int32_t *__restrict__ pair_src,
                                    const int32_t *__restrict__ offsets, int32_t n_pairs,
                                    int32_t nx, int32_t ny, double Lf, double step,
                                    nhip_match_t *__restrict__ out, int32_t *__restrict__ sums) {
  const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pairs) return;
  const unsigned long long key = keys[i];
  const uint32_t sum = (uint32_t)(key >> 32);
  const uint32_t lin = 0xffffffffu - (uint32_t)key;
  const int32_t src = pair_src[i];
  const int32_t n = offsets[src + 1] - offsets[src];
  nhip_match_t m;
  m.iy = (int32_t)(lin % (uint32_t)ny);
  m.ix = (int32_t)((lin / (uint32_t)ny) % (uint32_t)nx);
  m.itheta = (int32_t)(lin / ((uint32_t)ny * (uint32_t)nx));
  double sc = Lf;
  if (n > 0) sc = __dadd_rn(Lf, __ddiv_rn(__dmul_rn(step, (double)sum), (double)n));
  m.score = __double2float_rn(sc);
  out[i] = m;
  if (sums) sums[i] = (int32_t)sum;
}

int check_search(const nhip_grid_spec_t *spec, const nhip_search_t *search) {
  NHIP_REQUIRE(search->n_theta >= 1 && (search->n_theta & 1), "search: n_theta must be odd >= 1");
  NHIP_REQUIRE(search->nx >= 1 && (search->nx & 1), "search: nx must be odd >= 1");
  NHIP_REQUIRE(search->ny >= 1 && (search->ny & 1), "search: ny must be odd >= 1");
  NHIP_REQUIRE((search->nx - 1) / 2 <= spec->max_shift && (search->ny - 1) / 2 <= spec->max_shift,
               "search: shifts +-%d/+-%d exceed the grids' max_shift %d", (search->nx - 1) / 2,
               (search->ny - 1) / 2, spec->max_shift);
  NHIP_REQUIRE((int64_t)search->n_theta * search->nx * search->ny < 0x7fffffffll,
               "search: lattice too large for 32-bit linear index");
  return NHIP_OK;
}

void fill_params(CsmParams &P, const nhip_grid_spec_t *spec, const GridLayout &L,
                 const nhip_search_t *search) {
  memset(&P, 0, sizeof(P));
  P.n_theta = search->n_theta;
  P.nx = search->nx;
  P.ny = search->ny;
  P.hx = (search->nx - 1) / 2;
  P.hy = (search->ny - 1) / 2;
  P.npbx = (search->nx + PB_NX - 1) / PB_NX;
  P.npby = (search->ny + PB_NY - 1) / PB_NY;
  P.S = L.S;
  P.pad = L.pad;
  P.pitch = L.pitch;
  P.max_shift = spec->max_shift;
  P.grid_bytes = L.grid_bytes;
  P.res = spec->res;
}

}  // namespace

int launch_csm_match(const float *d_xy, const int32_t *d_offsets, const uint8_t *d_grids,
                     const nhip_grid_spec_t *spec, const GridLayout &L, const int32_t *d_pair_src,
                     const int32_t *d_pair_slot, const double *d_rot0_cs, const double *d_delta_cs,
                     const int32_t *d_pair_origin, int32_t n_pairs, const nhip_search_t *search,
                     uint64_t *d_keys,
                     nhip_match_t *d_out, int32_t *d_sums, hipStream_t s) {
  int rc = check_search(spec, search);
  if (rc) return rc;
  if (n_pairs == 0) return NHIP_OK;
  CsmParams P;
  fill_params(P, spec, L, search);
  P.xy = reinterpret_cast<const float2 *>(d_xy);
  P.offsets = d_offsets;
  P.grids = d_grids;
  P.pair_src = d_pair_src;
  P.pair_slot = d_pair_slot;
  P.rot0_cs = d_rot0_cs;
  P.delta_cs = d_delta_cs;
  P.pair_origin = d_pair_origin;
  P.keys = reinterpret_cast<unsigned long long *>(d_keys);
  P.n_pairs = n_pairs;
  const int64_t per_pair = (int64_t)P.n_theta * P.npbx * P.npby;
  const int64_t blocks = ((int64_t)(n_pairs + 7) / 8) * 8 * per_pair;
  NHIP_REQUIRE(blocks < 0x7fffffffll, "csm_match: %lld workgroups exceed one launch; split the batch",
               (long long)blocks);
  NHIP_TRY_HIP(hipMemsetAsync(d_keys, 0, sizeof(uint64_t) * (size_t)n_pairs, s));
  timer_begin(NHIP_TIMER_CSM, s);
  hipLaunchKernelGGL(csm_correlate_kernel<false>, dim3((uint32_t)blocks), dim3(CSM_THREADS), 0, s,
                     P);
  timer_end(NHIP_TIMER_CSM, s);
  hipLaunchKernelGGL(csm_finalize_kernel, dim3((n_pairs + 255) / 256), dim3(256), 0, s, P.keys,
                     d_pair_src, d_offsets, n_pairs, P.nx, P.ny, L.Lf, L.step, d_out, d_sums);
  NHIP_TRY_HIP(hipGetLastError());
  return NHIP_OK;
}

int launch_csm_scores(const float *d_xy, const int32_t *d_offsets, const uint8_t *d_grids,
                      const nhip_grid_spec_t *spec, const GridLayout &L, int32_t src, int32_t slot,
                      const double *d_rot0_cs, const double *d_delta_cs, int32_t origin_x,
                      int32_t origin_y, const nhip_search_t *search, int32_t *d_sums,
                      hipStream_t s) {
  int rc = check_search(spec, search);
  if (rc) return rc;
  CsmParams P;
  fill_params(P, spec, L, search);
  P.xy = reinterpret_cast<const float2 *>(d_xy);
  P.offsets = d_offsets;
  P.grids = d_grids;
  P.rot0_cs = d_rot0_cs;
  P.delta_cs = d_delta_cs;
  P.volume = d_sums;
  P.n_pairs = 1;
  P.single_src = src;
  P.single_slot = slot;
  P.single_ox = origin_x;
  P.single_oy = origin_y;
  const int64_t blocks = (int64_t)P.n_theta * P.npbx * P.npby;
  hipLaunchKernelGGL(csm_correlate_kernel<true>, dim3((uint32_t)blocks), dim3(CSM_THREADS), 0, s,
                     P);
  NHIP_TRY_HIP(hipGetLastError());
  return NHIP_OK;
}

}  // namespace nhip
