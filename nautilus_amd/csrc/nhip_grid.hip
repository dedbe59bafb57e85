// nhip_grid.hip -- K1: likelihood-grid construction on gfx950.
//
// Replaces the lookup table that CorrelativeScanMatcher rasterises from the target cloud
// (call site src/optimization/solver.cc:633-638; geometry src/visualization/cimg_debug.h:20-64).
// Spec: hit raster -> exact integer separable Gaussian blur ->
// floor, natural log, 8-bit quantisation (by an integer threshold table, so the grid is
// bit-identical to the CPU formulation).  Stored with a zero border of `pad` cells so the
// correlation kernel never bounds-checks.
//
// HBM-bound byte work: per target, S*S hit bytes written+read and pitch*pitch grid bytes
// written; almost all 64x64 tiles are empty and exit after the LDS any-hit vote.
#include "nhip_common.h"

namespace nhip {

namespace {

constexpr int TILE = 64;
constexpr int MAX_R = 16;
constexpr int TH_MAX = TILE + 2 * MAX_R;  // 96

struct GridKernelTables {
  int32_t taps[2 * MAX_R + 1];
  uint32_t thr[256];
};

// One block per target scan: mark the cells that contain at least one point.
__global__ __launch_bounds__(256) void grid_raster_kernel(
    const float2 *__restrict__ xy, const int32_t *__restrict__ offsets,
    const int32_t *__restrict__ target_ids, int32_t t0, uint8_t *__restrict__ H,
    uint8_t *__restrict__ occ, int32_t S, int32_t tiles, int32_t R, double res) {
  const int32_t t = blockIdx.x;
  const int32_t scan = target_ids[t0 + t];
  const int32_t beg = offsets[scan], end = offsets[scan + 1];
  uint8_t *h = H + (size_t)t * S * S;
  uint8_t *o = occ + (size_t)t * tiles * tiles;
  const long half = S / 2;
  for (int32_t p = beg + threadIdx.x; p < end; p += blockDim.x) {
    const float2 q = xy[p];
    if (!(fabsf(q.x) < 1e9f) || !(fabsf(q.y) < 1e9f)) continue;
    // cimg_debug.h:31-37: side/2 + floor(x / resolution), float promoted to double
    const long c = half + (long)floor((double)q.x / res);
    const long r = half + (long)floor((double)q.y / res);
    if (c < 0 || c >= S || r < 0 || r >= S) continue;  // cimg_debug.h:48-50
    h[(size_t)r * S + c] = 1;
    // tiles whose (tile + blur halo) contains this cell: at most 2 x 2 (R <= 16 < TILE)
    const int tx0 = (int)max(c - R, 0l) / TILE, tx1 = (int)min(c + R, (long)S - 1) / TILE;
    const int ty0 = (int)max(r - R, 0l) / TILE, ty1 = (int)min(r + R, (long)S - 1) / TILE;
    for (int ty = ty0; ty <= ty1; ty++)
      for (int tx = tx0; tx <= tx1; tx++) o[ty * tiles + tx] = 1;
  }
}

// 64x64 output tile per block: H tile (+halo) -> LDS, horizontal pass -> LDS, vertical
// pass + quantise -> padded grid.  Grid memory is pre-zeroed; empty tiles return early.
__global__ __launch_bounds__(256) void grid_blur_kernel(const uint8_t *__restrict__ H,
                                                        const uint8_t *__restrict__ occ,
                                                        uint8_t *__restrict__ grids, int32_t S,
                                                        int32_t pad, int32_t pitch, int64_t slot_bytes,
                                                        int32_t R, GridKernelTables tab) {
  __shared__ uint8_t sH[TH_MAX][TH_MAX + 4];
  __shared__ uint32_t sV[TH_MAX][TILE + 1];
  __shared__ uint32_t sThr[256];
  const int32_t t = blockIdx.z;
  // ~95 % of the tiles of a scan's grid see no hit within their halo: leave before touching H
  if (!occ[((size_t)t * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x]) return;
  const int32_t r0 = blockIdx.y * TILE, c0 = blockIdx.x * TILE;
  const uint8_t *h = H + (size_t)t * S * S;
  uint8_t *g = grids + (size_t)t * slot_bytes;
  const int TH = TILE + 2 * R;
  int any = 0;
  for (int i = threadIdx.x; i < TH * TH; i += 256) {
    const int rr = i / TH, cc = i % TH;
    const int r = r0 + rr - R, c = c0 + cc - R;
    uint8_t v = 0;
    if (r >= 0 && r < S && c >= 0 && c < S) v = h[(size_t)r * S + c];
    sH[rr][cc] = v;
    any |= v;
  }
  sThr[threadIdx.x] = tab.thr[threadIdx.x];
  if (!__syncthreads_or(any)) return;
  // horizontal pass: V1[rr][c] = sum_j taps[j] * H[rr][c + j]
  for (int i = threadIdx.x; i < TH * TILE; i += 256) {
    const int rr = i / TILE, c = i % TILE;
    uint32_t a = 0;
    for (int j = 0; j <= 2 * R; j++) a += (uint32_t)tab.taps[j] * sH[rr][c + j];
    sV[rr][c] = a;
  }
  __syncthreads();
  // vertical pass + quantise; each thread produces 4 consecutive columns of one row
  for (int i = threadIdx.x; i < TILE * (TILE / 4); i += 256) {
    const int r = i / (TILE / 4), c4 = (i % (TILE / 4)) * 4;
    if (r0 + r >= S) continue;
    uint32_t packed = 0;
    for (int b = 0; b < 4; b++) {
      uint32_t a = 0;
      for (int k = 0; k <= 2 * R; k++) a += (uint32_t)tab.taps[k] * sV[r + k][c4 + b];
      uint32_t q = 0;
      if (a) {
        // q = #{k in 1..255 : thr[k] <= a}; thr is non-decreasing
        for (int step = 128; step >= 1; step >>= 1) {
          const uint32_t n = q + step;
          if (n <= 255 && sThr[n] <= a) q = n;
        }
      }
      packed |= q << (8 * b);
    }
    uint8_t *dst = g + (size_t)(r0 + r + pad) * pitch + (c0 + c4 + pad);
    if (c0 + c4 + 3 < S) {
      *reinterpret_cast<uint32_t *>(dst) = packed;  // pad, c0, c4 are multiples of 4
    } else {
      for (int b = 0; b < 4; b++)
        if (c0 + c4 + b < S) dst[b] = (uint8_t)(packed >> (8 * b));
    }
  }
}

// ---- skip map ---------------------------------------------------------------------------
// A likelihood grid is zero except within the blur radius of a wall.  A wave of
// csm_correlate_kernel adds, per point, the CSM_WAVE_ROWS x CSM_ROW_DW-dword strip of the grid that
// starts at (window row, window column & ~3); on the 1081-beam scans ~45 % of those strips hold
// nothing but zeros.  The map stores one BIT per stored row r and aligned dword column c: "rows
// [r, r + 21) x dwords [c, c + 21) contain a non-zero cell" (bit c & 7 of byte c >> 3 of map row r,
// SKIP_PITCH(pitch) bytes per row), so the kernel can leave those strips out -- the sums are
// unchanged, bit for bit -- for one byte load per point.  One block per 64-row x 64-dword map
// tile; tiles whose footprint touches no occupied blur tile stay on the memset's zeros.
constexpr int MT = 64;
constexpr int SK_ROWS = MT + CSM_WAVE_ROWS - 1;  // grid rows feeding one map tile (84)
static_assert(CSM_ROW_DW - 1 <= 20 && MT == 64, "row mask is built from one 64-lane and one 20-lane ballot");

__global__ __launch_bounds__(256) void grid_skipmap_kernel(const uint8_t *__restrict__ occ,
                                                           uint8_t *__restrict__ grids, int32_t S,
                                                           int32_t tiles, int32_t pad, int32_t pitch,
                                                           int32_t rows, int64_t grid_bytes,
                                                           int64_t slot_bytes) {
  __shared__ unsigned long long sH[SK_ROWS];  // per grid row: bit c = a non-zero dword in [c0 + c, c0 + c + 21)
  const int32_t t = blockIdx.z, tid = threadIdx.x;
  const int32_t r0 = blockIdx.y * MT, c0 = blockIdx.x * MT;  // first map row / dword column
  // footprint in raster coordinates -> blur tiles that could have written into it
  const int32_t fr0 = r0 - pad, fr1 = r0 + SK_ROWS - 1 - pad;
  const int32_t fc0 = 4 * c0 - pad, fc1 = 4 * (c0 + MT + CSM_ROW_DW - 1) - 1 - pad;
  int any = 0;
  if (fr1 >= 0 && fr0 < S && fc1 >= 0 && fc0 < S) {
    const int32_t ty0 = max(fr0, 0) / TILE, ty1 = min(fr1, S - 1) / TILE;
    const int32_t tx0 = max(fc0, 0) / TILE, tx1 = min(fc1, S - 1) / TILE;
    const int32_t ntx = tx1 - tx0 + 1, nt = (ty1 - ty0 + 1) * ntx;
    for (int32_t i = tid; i < nt; i += 256)
      any |= occ[((size_t)t * tiles + ty0 + i / ntx) * tiles + tx0 + i % ntx];
  }
  if (!__syncthreads_or(any)) return;
  uint8_t *g = grids + (size_t)t * slot_bytes;
  uint8_t *M = g + grid_bytes;
  const int32_t mpitch = pitch / 4;
  const int wave = tid >> 6, lane = tid & 63;
  // horizontal: 84-bit non-zero mask of a row (two ballots), then OR over windows of 21 bits
  constexpr int UNR = 4;
  for (int32_t rb = wave * UNR; rb < SK_ROWS; rb += 4 * UNR) {
    uint32_t v0[UNR], v1[UNR];
#pragma unroll
    for (int u = 0; u < UNR; u++) {
      const int32_t r = r0 + rb + u;
      v0[u] = v1[u] = 0;
      if (rb + u < SK_ROWS && r < rows) {
        const uint32_t *row = reinterpret_cast<const uint32_t *>(g + (size_t)r * pitch);
        if (c0 + lane < mpitch) v0[u] = row[c0 + lane];
        if (lane < CSM_ROW_DW - 1 && c0 + 64 + lane < mpitch) v1[u] = row[c0 + 64 + lane];
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; u++) {
      const unsigned long long b0 = __ballot(v0[u] != 0u), b1 = __ballot(v1[u] != 0u);
      unsigned __int128 m = ((unsigned __int128)b1 << 64) | b0;
      m |= m >> 1;
      m |= m >> 2;
      m |= m >> 4;                                             // windows of 8
      const unsigned __int128 m21 = m | (m >> 8) | (m >> 13);  // [c, c+16) U [c+13, c+21)
      if (lane == 0 && rb + u < SK_ROWS) sH[rb + u] = (unsigned long long)m21;
    }
  }
  __syncthreads();
  // vertical: map row r = OR of the 21 row masks from r on; 64 bits = 8 map bytes
  if (tid < MT && r0 + tid < rows) {
    unsigned long long v = 0;
    for (int j = 0; j < CSM_WAVE_ROWS; j++) v |= sH[tid + j];
    *reinterpret_cast<unsigned long long *>(M + (size_t)(r0 + tid) * skip_pitch(pitch) + c0 / 8) = v;
  }
}

}  // namespace

int launch_grid_build(const float *d_xy, const int32_t *d_offsets, const int32_t *d_target_ids,
                      int32_t n_targets, const nhip_grid_spec_t *spec, const GridLayout &L,
                      uint8_t *d_grids, void *d_ws, int64_t ws_bytes, hipStream_t s) {
  NHIP_REQUIRE(L.R <= MAX_R, "grid_build: blur radius %d > %d (sigma too large)", L.R, MAX_R);
  NHIP_REQUIRE(L.K * L.K < (1ll << 32), "grid_build: tap sum overflows 32-bit accumulation");
  NHIP_REQUIRE(L.pitch % 4 == 0, "grid_build: pitch must be a multiple of 4");
  const int tiles = (L.S + TILE - 1) / TILE;
  const int64_t per = (int64_t)L.S * L.S + (int64_t)tiles * tiles;  // hit raster + tile occupancy
  const int64_t chunk = ws_bytes / per;
  NHIP_REQUIRE(chunk >= 1, "grid_build: workspace %lld B < one hit raster (%lld B)",
               (long long)ws_bytes, (long long)per);
  GridTables T;
  int rc = make_tables(spec, L, &T);
  if (rc) return rc;
  GridKernelTables kt;
  memset(&kt, 0, sizeof(kt));
  for (int i = 0; i <= 2 * L.R; i++) kt.taps[i] = T.taps[i];
  for (int i = 0; i < 256; i++) kt.thr[i] = T.thr[i];
  for (int64_t t0 = 0; t0 < n_targets; t0 += chunk) {
    const int32_t n = (int32_t)((n_targets - t0 < chunk) ? (n_targets - t0) : chunk);
    uint8_t *H = static_cast<uint8_t *>(d_ws);
    uint8_t *occ = H + (size_t)n * L.S * L.S;
    uint8_t *g = d_grids + (size_t)t0 * L.slot_bytes;
    NHIP_TRY_HIP(hipMemsetAsync(H, 0, (size_t)n * per, s));  // raster and occupancy in one fill
    NHIP_TRY_HIP(hipMemsetAsync(g, 0, (size_t)n * L.slot_bytes, s));  // images and skip maps
    hipLaunchKernelGGL(grid_raster_kernel, dim3(n), dim3(256), 0, s,
                       reinterpret_cast<const float2 *>(d_xy), d_offsets, d_target_ids,
                       (int32_t)t0, H, occ, L.S, tiles, L.R, spec->res);
    timer_begin(NHIP_TIMER_GRID, s);
    hipLaunchKernelGGL(grid_blur_kernel, dim3(tiles, tiles, n), dim3(256), 0, s, H, occ, g, L.S, L.pad,
                       L.pitch, L.slot_bytes, L.R, kt);
    const int32_t rows = L.S + 2 * L.pad, mpitch = L.pitch / 4;
    hipLaunchKernelGGL(grid_skipmap_kernel, dim3((mpitch + MT - 1) / MT, (rows + MT - 1) / MT, n), dim3(256), 0, s,
                       occ, g, L.S, tiles, L.pad, L.pitch, rows, L.grid_bytes, L.slot_bytes);
    timer_end(NHIP_TIMER_GRID, s);
  }
  NHIP_TRY_HIP(hipGetLastError());
  return NHIP_OK;
}

}  // namespace nhip
