// nhip_grid.hip -- K1: likelihood-grid construction on gfx950.
//
// Replaces the lookup table that CorrelativeScanMatcher rasterises from the target cloud
// (call site src/optimization/solver.cc:633-638; geometry src/visualization/cimg_debug.h:20-64).
// Spec: hit raster -> exact integer separable Gaussian blur ->
// floor, natural log, 8- or 16-bit quantisation (by an integer threshold table, so the grid is
// bit-identical to the CPU formulation).  Stored with a zero border of `pad` cells so the
// correlation kernel never bounds-checks.
//
// HBM-bound byte work: per target the image and its skip map are zero-filled and the ~20 % of
// 64x64 tiles within blur reach of a hit are computed (kernels below); no intermediate raster.
#include "nhip_common.h"

namespace nhip {

namespace {

constexpr int TILE = 64;
constexpr int MAX_R = 16;
constexpr int TH_MAX = TILE + 2 * MAX_R;  // 96

struct GridKernelTables {
  int32_t taps[2 * MAX_R + 1];
  uint32_t thr[256];
  float q16_a, q16_b;  // 16-bit cells: q ~ q16_a * ln(sum) + q16_b, the first guess of the table search
};

// Cell of a point (cimg_debug.h:31-37: side/2 + floor(x / resolution), float promoted to double);
// false for non-finite points and cells outside the grid (dropped, cimg_debug.h:48-50).
__device__ __forceinline__ bool hit_cell(float2 q, int32_t S, double res, double inv_res, int32_t *c, int32_t *r) {
  if (!(fabsf(q.x) < 1e9f) || !(fabsf(q.y) < 1e9f)) return false;
  const double fc = floor_quotient((double)q.x, res, inv_res), fr = floor_quotient((double)q.y, res, inv_res);
  const double half = (double)(S / 2);
  if (!(fc >= -half && fc < (double)S - half && fr >= -half && fr < (double)S - half)) return false;
  *c = S / 2 + (int32_t)fc;
  *r = S / 2 + (int32_t)fr;
  return true;
}

// The points of target scan `scan` (an id read from device memory): an id outside [0, n_scans) is an EMPTY scan -- its
// grid comes out all floor -- and is reported through the device's status words (nhip_dev_status), never dereferenced.
__device__ __forceinline__ void target_points(const int32_t *__restrict__ offsets, int32_t n_scans, int32_t scan, int32_t index,
                                              uint32_t *status, bool report, int32_t *beg, int32_t *end) {
  *beg = *end = 0;
  if (id_in(scan, n_scans)) {
    *beg = offsets[scan];
    *end = offsets[scan + 1];
  } else if (report) {
    flag_bad_id(status, BAD_TARGET_ID, scan, index);
  }
}

// One block per target scan: mark every 64x64 tile whose blur halo contains a hit.  (There is no
// hit raster: the blur kernel gathers a tile's hits straight from the point list.)
__global__ __launch_bounds__(256) void grid_occupancy_kernel(
    const float2 *__restrict__ xy, const int32_t *__restrict__ offsets,
    const int32_t *__restrict__ target_ids, int32_t t0, uint8_t *__restrict__ occ, int32_t S, int32_t tiles,
    int32_t R, double res, double inv_res, int32_t n_scans, uint32_t *__restrict__ status) {
  const int32_t t = blockIdx.x;
  int32_t beg, end;
  target_points(offsets, n_scans, target_ids[t0 + t], t0 + t, status, threadIdx.x == 0, &beg, &end);
  uint8_t *o = occ + (size_t)t * tiles * tiles;
  for (int32_t p = beg + threadIdx.x; p < end; p += blockDim.x) {
    int32_t c, r;
    if (!hit_cell(xy[p], S, res, inv_res, &c, &r)) continue;
    // tiles whose (tile + blur halo) contains this cell: at most 2 x 2 (R <= 16 < TILE)
    const int tx0 = max(c - R, 0) / TILE, tx1 = min(c + R, S - 1) / TILE;
    const int ty0 = max(r - R, 0) / TILE, ty1 = min(r + R, S - 1) / TILE;
    for (int ty = ty0; ty <= ty1; ty++)
      for (int tx = tx0; tx <= tx1; tx++) o[ty * tiles + tx] = 1;
  }
}

// The same with the list of occupied tiles in one launch (round 4): the target's tiles are marked in an LDS bitmap, every
// occupancy byte of the target is written (no memset of the array), and the block appends its occupied tiles to the list
// with ONE atomic on the counter -- instead of a memset, this kernel and grid_tile_list_kernel over all n x tiles^2 bytes.
// Entries of one target are consecutive and ascending; targets come in the order their blocks finish.
constexpr int OCC_WORDS_MAX = 2048;  // tiles^2 <= 65,536 bits (sides of up to 16,384 cells)
__global__ __launch_bounds__(256) void grid_occupancy_list_kernel(
    const float2 *__restrict__ xy, const int32_t *__restrict__ offsets,
    const int32_t *__restrict__ target_ids, int32_t t0, uint8_t *__restrict__ occ, int32_t S, int32_t tiles,
    int32_t R, double res, double inv_res, int32_t *__restrict__ count, int32_t *__restrict__ list, int32_t n_scans,
    uint32_t *__restrict__ status) {
  __shared__ uint32_t sBits[OCC_WORDS_MAX];
  __shared__ int32_t sBase, sN;
  const int32_t t = blockIdx.x, nt = tiles * tiles, nw = (nt + 31) / 32;
  int32_t beg, end;
  target_points(offsets, n_scans, target_ids[t0 + t], t0 + t, status, threadIdx.x == 0, &beg, &end);
  for (int i = threadIdx.x; i < nw; i += 256) sBits[i] = 0u;
  if (threadIdx.x == 0) sN = 0;
  __syncthreads();
  for (int32_t p = beg + threadIdx.x; p < end; p += 256) {
    int32_t c, r;
    if (!hit_cell(xy[p], S, res, inv_res, &c, &r)) continue;
    // tiles whose (tile + blur halo) contains this cell: at most 2 x 2 (R <= 16 < TILE)
    const int tx0 = max(c - R, 0) / TILE, tx1 = min(c + R, S - 1) / TILE;
    const int ty0 = max(r - R, 0) / TILE, ty1 = min(r + R, S - 1) / TILE;
    for (int ty = ty0; ty <= ty1; ty++)
      for (int tx = tx0; tx <= tx1; tx++) {
        const int k = ty * tiles + tx;
        atomicOr(&sBits[k >> 5], 1u << (k & 31));
      }
  }
  __syncthreads();
  // occupancy bytes (the skip map's and the band kernels' input), and this thread's words' share of the list
  uint8_t *o = occ + (size_t)t * nt;
  for (int i = threadIdx.x; i < nt; i += 256) o[i] = (uint8_t)((sBits[i >> 5] >> (i & 31)) & 1u);
  int32_t mine = 0;
  for (int w = threadIdx.x; w < nw; w += 256) mine += __builtin_popcount(sBits[w]);
  const int32_t at = mine ? atomicAdd(&sN, mine) : 0;  // (order inside the target's segment: by thread, then ascending)
  __syncthreads();
  if (threadIdx.x == 0) sBase = sN ? atomicAdd(count, sN) : 0;
  __syncthreads();
  int32_t k = sBase + at;
  for (int w = threadIdx.x; w < nw; w += 256) {
    uint32_t m = sBits[w];
    while (m) {
      const int b = __builtin_ctz(m);
      m &= m - 1u;
      list[k++] = t * nt + 32 * w + b;
    }
  }
}

// ~95 % of the 64x64 tiles of a scan's grid see no hit within their blur halo.  Launching a
// workgroup per tile just to read its occupancy byte and leave cost more than the blur itself
// (361k workgroups per 1000 targets), so the occupied (target, tile) pairs are compacted into a
// list first and the blur runs as a persistent grid over that list.
__global__ __launch_bounds__(256) void grid_tile_list_kernel(const uint8_t *__restrict__ occ, int32_t n_tiles_total,
                                                             int32_t *__restrict__ count, int32_t *__restrict__ list) {
  const int32_t i = blockIdx.x * 256 + threadIdx.x;
  const bool on = i < n_tiles_total && occ[i];
  // one atomic per wave: rank within the wave by ballot
  const unsigned long long m = __ballot(on);
  const int lane = threadIdx.x & 63;
  int32_t base = 0;
  if (lane == 0 && m) base = atomicAdd(count, (int32_t)__builtin_popcountll(m));
  base = __shfl(base, 0, 64);
  if (on) list[base + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = i;
}

// 64x64 output tile per list entry.  The hit raster is sparse (a few dozen hits per tile), so the
// separable blur is evaluated as a scatter: every hit of the tile's (64 + 2R)^2 neighbourhood adds
// taps[i] * taps[j] to the (2R+1)^2 outputs around it (LDS integer atomics -- the same integer sum as
// the two-pass form, in any order), ~170 adds per hit instead of 29 multiply-adds per OUTPUT; then
// the non-zero sums are quantised by binary search of the threshold table and stored as aligned
// dwords.  Grid memory is pre-zeroed.
constexpr int MAX_TILE_HITS = TH_MAX * TH_MAX;  // every cell of the neighbourhood a hit

// CB = bytes per cell.  8-bit cells: 256-entry threshold table passed by value (LDS copy); 16-bit cells: the
// 65536-entry table lives in the workspace (thr16, L2-resident) and is searched in 16 steps.
template <int CB>
__global__ __launch_bounds__(256) void grid_blur_kernel(const float2 *__restrict__ xy,
                                                        const int32_t *__restrict__ offsets,
                                                        const int32_t *__restrict__ target_ids, int32_t t0,
                                                        const int32_t *__restrict__ count,
                                                        const int32_t *__restrict__ list, int32_t tiles,
                                                        uint8_t *__restrict__ grids, int32_t S,
                                                        int32_t pad, int32_t pitch, int64_t slot_bytes,
                                                        int32_t R, double res, double inv_res, GridKernelTables tab,
                                                        const uint32_t *__restrict__ thr16, int64_t hi_offset,
                                                        int32_t hi_tpr, int64_t hi_copy_bytes, int32_t t16_tpr,
                                                        int32_t n_scans, int64_t hits_offset, int32_t hits_pitch,
                                                        int32_t has_image, uint32_t *__restrict__ masks) {
  // masks: null, or GRID_WS_MASK_WORDS words per list entry -- the lines of the tiled planes this build writes inside the
  // entry's tile (geometries whose tiles start on line boundaries: pad a multiple of 16), for the next rebuild's clear
  __shared__ uint32_t sMask[GRID_WS_MASK_WORDS];
  __shared__ unsigned long long sBal[2][TILE / 4];  // per four rows of the tile: lanes = (row & 3) * 16 + group of four columns
  __shared__ uint32_t sA[TILE][TILE + 1];
  __shared__ uint16_t sHits[MAX_TILE_HITS];
  __shared__ uint32_t sSeen[(TH_MAX * TH_MAX + 31) / 32];  // one bit per neighbourhood cell: a cell is a hit once
  __shared__ uint32_t sThr[256];
  __shared__ int32_t sTaps[2 * MAX_R + 1];
  __shared__ int32_t sNH;
  if (CB == 1) sThr[threadIdx.x] = tab.thr[threadIdx.x];
  if (threadIdx.x <= 2 * R) sTaps[threadIdx.x] = tab.taps[threadIdx.x];
  const int32_t n_entries = *count;
  const int TH = TILE + 2 * R, NT = 2 * R + 1;
  for (int32_t e = blockIdx.x; e < n_entries; e += gridDim.x) {
    const int32_t entry = list[e];
    const int32_t t = entry / (tiles * tiles), tile = entry % (tiles * tiles);
    const int32_t r0 = (tile / tiles) * TILE, c0 = (tile % tiles) * TILE;
    uint8_t *g = grids + (size_t)t * slot_bytes;
    __syncthreads();  // the previous entry is done with the LDS arrays
    for (int i = threadIdx.x; i < TILE * (TILE + 1); i += 256) (&sA[0][0])[i] = 0u;
    for (int i = threadIdx.x; i < (TH_MAX * TH_MAX + 31) / 32; i += 256) sSeen[i] = 0u;
    if (threadIdx.x == 0) sNH = 0;
    if (threadIdx.x < GRID_WS_MASK_WORDS) {
      sMask[threadIdx.x] = 0u;
      if (masks) masks[(size_t)e * GRID_WS_MASK_WORDS + threadIdx.x] = 0u;  // (an entry without hits writes nothing)
    }
    if (threadIdx.x < 2 * (TILE / 4)) (&sBal[0][0])[threadIdx.x] = 0ull;
    __syncthreads();
    // hits of the neighbourhood -> list (row, column packed; TH <= 96): every point of the target scan
    // whose cell falls inside, each cell once
    {
      // (a target whose id is out of range has no tiles on the list: the occupancy kernel reported it)
      int32_t beg, end;
      target_points(offsets, n_scans, target_ids[t0 + t], t0 + t, nullptr, false, &beg, &end);
      // (the neighbourhood in metres, a cell wider on every side: nineteen points in twenty lie outside it and are
      //  dropped by four single-precision compares instead of two double-precision quotients; the exact test follows)
      const float resf = (float)res;
      const float x_lo = (float)(c0 - R - S / 2 - 1) * resf, x_hi = (float)(c0 + TILE + R - S / 2 + 1) * resf;
      const float y_lo = (float)(r0 - R - S / 2 - 1) * resf, y_hi = (float)(r0 + TILE + R - S / 2 + 1) * resf;
      for (int32_t p = beg + threadIdx.x; p < end; p += 256) {
        int32_t c, r;
        const float2 q = xy[p];
        if (!(q.x >= x_lo && q.x < x_hi && q.y >= y_lo && q.y < y_hi)) continue;
        if (!hit_cell(q, S, res, inv_res, &c, &r)) continue;
        const int32_t rr = r - (r0 - R), cc = c - (c0 - R);
        if (rr < 0 || rr >= TH || cc < 0 || cc >= TH) continue;
        const uint32_t idx = (uint32_t)(rr * TH + cc), bit = 1u << (idx & 31u);
        if (!(atomicOr(&sSeen[idx >> 5], bit) & bit)) sHits[atomicAdd(&sNH, 1)] = (uint16_t)((rr << 8) | cc);
      }
    }
    __syncthreads();
    const int32_t nh = sNH;
    if (nh == 0) continue;
    // the tile's own hits into the hit raster (64 rows x 64 bits = two dwords per row; pre-zeroed, a tile owns its dwords:
    // tiles start at multiples of 64 cells and the raster's border is 32): what the exact-score pass reads
    if (threadIdx.x < 2 * TILE) {
      const int r = threadIdx.x >> 1, h = threadIdx.x & 1;
      if (r0 + r < S && c0 + 32 * h < S) {
        const uint32_t b0 = (uint32_t)((R + r) * TH + R + 32 * h);  // the row's first bit of this half in sSeen
        const uint32_t w0 = sSeen[b0 >> 5], w1 = sSeen[(b0 >> 5) + 1], sh = b0 & 31u;
        const uint32_t bits = sh ? (w0 >> sh) | (w1 << (32u - sh)) : w0;
        if (bits)
          *reinterpret_cast<uint32_t *>(g + hits_offset + (size_t)(r0 + r + HIT_PAD) * hits_pitch +
                                        4 * (size_t)(((c0 + HIT_PAD) >> 5) + h)) = bits;
      }
    }
    // one work item per (hit, output row): up to 2R+1 atomic adds
    for (int32_t wi = threadIdx.x; wi < nh * NT; wi += 256) {
      const int32_t hit = sHits[wi / NT], di = wi % NT;
      const int32_t ro = (hit >> 8) - di, cc = hit & 0xff;
      if (ro < 0 || ro >= TILE) continue;
      const uint32_t tr = (uint32_t)sTaps[di];
      for (int dj = 0; dj < NT; dj++) {
        const int32_t co = cc - dj;
        if (co >= 0 && co < TILE) atomicAdd(&sA[ro][co], tr * (uint32_t)sTaps[dj]);
      }
    }
    __syncthreads();
    // quantise; each thread produces 4 consecutive columns of one row, in four turns
    // 16-bit cells.  The thresholds grow exponentially: a first guess from ln(a) is the answer except next to a threshold, and
    // the table settles it exactly.  The guesses of the eight cells of TWO turns first, then their table entries (thr[g],
    // thr[g + 1]), all in flight at once: two trips to the L2-resident table per thread and tile (round 4 made one per turn;
    // the cell-by-cell form before it two to four dependent ones per cell; all four turns at once need 171 registers).
    constexpr int TURNS = TILE * (TILE / 4) / 256, GROUP = 2;
    static_assert(TURNS % GROUP == 0, "whole groups of turns");
#pragma unroll 1
    for (int t0 = 0; t0 < TURNS; t0 += GROUP) {
    uint32_t avT[GROUP][4], gvT[GROUP][4];
    uint2 tvT[GROUP][4];
    if (CB == 2) {
#pragma unroll
      for (int t = 0; t < GROUP; t++) {
        const int i = threadIdx.x + 256 * (t0 + t), r = i / (TILE / 4), c4 = (i % (TILE / 4)) * 4;
#pragma unroll
        for (int b = 0; b < 4; b++) {
          avT[t][b] = r0 + r < S ? sA[r][c4 + b] : 0u;
          const float gf = tab.q16_a * __logf((float)(avT[t][b] ? avT[t][b] : 1u)) + tab.q16_b;
          gvT[t][b] = gf <= 0.f ? 0u : (gf >= 65534.f ? 65534u : (uint32_t)gf);
        }
      }
#pragma unroll
      for (int t = 0; t < GROUP; t++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
          // (cells without a sum load nothing)
          tvT[t][b] = make_uint2(0u, 0u);
          if (avT[t][b]) tvT[t][b] = make_uint2(thr16[gvT[t][b]], thr16[gvT[t][b] + 1u]);
        }
    }
#pragma unroll
    for (int t = 0; t < GROUP; t++) {
      const int i = threadIdx.x + 256 * (t0 + t);
      const int r = i / (TILE / 4), c4 = (i % (TILE / 4)) * 4;
      if (r0 + r >= S) continue;
      uint32_t qv[4];
      uint32_t any = 0;
      if (CB == 2) {
        uint32_t av[4], gv[4];
        uint2 tv[4];
#pragma unroll
        for (int b = 0; b < 4; b++) {
          av[b] = avT[t][b];
          gv[b] = gvT[t][b];
          tv[b] = tvT[t][b];
        }
#pragma unroll
        for (int b = 0; b < 4; b++) {
          const uint32_t a = av[b];
          uint32_t q = 0;
          if (a) {
            uint32_t gq = gv[b];
            if ((gq == 0u || tv[b].x <= a) && tv[b].y > a) {
              q = gq;
            } else {
              // next to a threshold, or a poor guess: a few steps either way, else the binary search
              int it = 0;
              while (gq < 65535u && it < 6 && thr16[gq + 1] <= a) {
                gq++;
                it++;
              }
              while (gq > 0u && it < 12 && thr16[gq] > a) {
                gq--;
                it++;
              }
              const bool settled = (gq == 0u || thr16[gq] <= a) && (gq == 65535u || thr16[gq + 1] > a);
              if (settled) {
                q = gq;
              } else {
                for (int step = 32768; step >= 1; step >>= 1) {
                  const uint32_t n = q + step;
                  if (n <= 65535 && thr16[n] <= a) q = n;
                }
              }
            }
          }
          qv[b] = q;
          any |= q;
        }
      } else {
        for (int b = 0; b < 4; b++) {
          const uint32_t a = sA[r][c4 + b];
          uint32_t q = 0;
          if (a) {
            // q = #{k in 1..levels : thr[k] <= a}; thr is non-decreasing
            for (int step = 128; step >= 1; step >>= 1) {
              const uint32_t n = q + step;
              if (n <= 255 && sThr[n] <= a) q = n;
            }
          }
          qv[b] = q;
          any |= q;
        }
      }
      if (any == 0u) continue;  // the grid is pre-zeroed
      uint8_t *dst = g + (size_t)(r0 + r + pad) * pitch + (size_t)(c0 + c4 + pad) * CB;
      if (!has_image) {  // (NHIP_GRID_NO_IMAGE: the cells live in the matcher's tiled copies only)
      } else if (c0 + c4 + 3 < S) {  // pad, c0, c4 are multiples of 4: a whole aligned dword / qword
        if (CB == 1) *reinterpret_cast<uint32_t *>(dst) = qv[0] | (qv[1] << 8) | (qv[2] << 16) | (qv[3] << 24);
        else *reinterpret_cast<uint2 *>(dst) = make_uint2(qv[0] | (qv[1] << 16), qv[2] | (qv[3] << 16));
      } else {
        for (int b = 0; b < 4; b++)
          if (c0 + c4 + b < S) {
            if (CB == 1) dst[b] = (uint8_t)qv[b];
            else reinterpret_cast<uint16_t *>(dst)[b] = (uint16_t)qv[b];
          }
      }
      if (masks) {
        // which of the wave's groups (four rows x 16 groups of four columns) store anything, and which store high bytes:
        // two ballots, kept per (wave, turn) = per four rows; the entry's line masks are formed from them after the loop
        bool hi_any = false;
        for (int b = 0; b < 4; b++)
          if (c0 + c4 + b < S && (CB == 2 ? qv[b] >> 8 : qv[b])) hi_any = true;
        const unsigned long long act = __ballot(1), hib = __ballot(hi_any);  // (lanes still here: a non-zero group inside the raster)
        if ((int)(threadIdx.x & 63u) == __ffsll((long long)act) - 1) {
          sBal[0][r >> 2] = act;
          sBal[1][r >> 2] = hib;
        }
      }
      if (CB == 2) {  // the matcher's tiled copy of the 16-bit cells (a 4-aligned group of four lies in one tile row)
        uint8_t *td = g + hi_offset + 2 * hi_copy_bytes + t16_tiled((uint32_t)(r0 + r + pad), (uint32_t)(c0 + c4 + pad), (uint32_t)t16_tpr);
        if (c0 + c4 + 3 < S) {
          *reinterpret_cast<uint2 *>(td) = make_uint2(qv[0] | (qv[1] << 16), qv[2] | (qv[3] << 16));
        } else {
          for (int b = 0; b < 4; b++)
            if (c0 + c4 + b < S) reinterpret_cast<uint16_t *>(td)[b] = (uint16_t)qv[b];
        }
      }
      {  // the matcher's 8-bit plane: the high bytes of 16-bit cells, the cells themselves of 8-bit ones (columns past
         // the raster stay zero, like the image's border)
        uint32_t h = 0u;
        for (int b = 0; b < 4; b++)
          if (c0 + c4 + b < S) h |= (CB == 2 ? qv[b] >> 8 : qv[b]) << (8 * b);
        if (h) {  // (both tiled copies: a 4-aligned group of four cells never straddles a tile of either)
          const uint32_t hr = (uint32_t)(r0 + r + pad), hc = (uint32_t)(c0 + c4 + pad);
          *reinterpret_cast<uint32_t *>(g + hi_offset + hi_tiled(hr, hc, 0u, (uint32_t)hi_tpr, (uint32_t)hi_copy_bytes)) = h;
          *reinterpret_cast<uint32_t *>(g + hi_offset + hi_tiled(hr, hc, 1u, (uint32_t)hi_tpr, (uint32_t)hi_copy_bytes)) = h;
        }
      }
    }
    }  // (groups of turns)
    if (masks) {
      // one thread per line of the tile: 32 of the first copy of the 8-bit plane (row of eight lr, 16-byte column k: groups
      // 4k .. 4k + 3 of the rows' ballots), 40 of the copy shifted by 8 columns (groups 4k - 2 .. 4k + 1), 64 of the
      // 16-bit copy (8-cell column k: groups 2k, 2k + 1)
      __syncthreads();
      if (threadIdx.x < 136) {
        const uint32_t t = threadIdx.x;
        uint32_t lr, word, bit;
        unsigned long long pat;  // the groups of one row; the ballot holds four rows, 16 lanes apart
        bool hi = true;
        if (t < 32) {
          lr = t >> 2;
          pat = 0xFull << (4u * (t & 3u));
          word = 0u; bit = t;
        } else if (t < 72) {
          const uint32_t b1 = t - 32u, k = b1 % 5u;
          lr = b1 / 5u;
          pat = k == 0u ? 0x3ull : (k == 4u ? 0xC000ull : 0xFull << (4u * k - 2u));
          word = 1u + (b1 >> 5); bit = b1 & 31u;
        } else {
          const uint32_t b2 = t - 72u;
          lr = b2 >> 3;
          pat = 0x3ull << (2u * (b2 & 7u));
          word = 3u + (b2 >> 5); bit = b2 & 31u;
          hi = false;
        }
        pat |= pat << 16;
        pat |= pat << 32;
        const unsigned long long any = (sBal[hi ? 1 : 0][2u * lr] | sBal[hi ? 1 : 0][2u * lr + 1u]) & pat;
        if (any && (hi || CB == 2)) atomicOr(&sMask[word], 1u << bit);
      }
      __syncthreads();
      if (threadIdx.x < GRID_WS_MASK_WORDS) masks[(size_t)e * GRID_WS_MASK_WORDS + threadIdx.x] = sMask[threadIdx.x];
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
static_assert(2 * CSM_ROW_DW - 1 <= 64 && MT == 64, "row mask is built from two 64-lane ballots");

// CB = bytes per cell: a strip row spans ROW_DW = CB * CSM_ROW_DW aligned dwords (21 / 42; csm_correlate16_kernel
// starts its strips at 8-byte-aligned columns and looks up the even dword).  z = target index within the launch
// (t_base + blockIdx.z: launches are chunked at 65,535 targets).
template <int CB>
__global__ __launch_bounds__(256) void grid_skipmap_kernel(const uint8_t *__restrict__ occ,
                                                           uint8_t *__restrict__ grids, int32_t S,
                                                           int32_t tiles, int32_t pad, int32_t pitch,
                                                           int32_t rows, int64_t grid_bytes,
                                                           int64_t slot_bytes, int32_t t_base) {
  constexpr int ROW_DW = CB * CSM_ROW_DW;
  constexpr int CPD = 4 / CB;  // cells per dword
  __shared__ unsigned long long sH[SK_ROWS];  // per grid row: bit c = a non-zero dword in [c0 + c, c0 + c + ROW_DW)
  const int32_t t = t_base + blockIdx.z, tid = threadIdx.x;
  const int32_t r0 = blockIdx.y * MT, c0 = blockIdx.x * MT;  // first map row / dword column
  // footprint in raster coordinates -> blur tiles that could have written into it
  const int32_t fr0 = r0 - pad, fr1 = r0 + SK_ROWS - 1 - pad;
  const int32_t fc0 = CPD * c0 - pad, fc1 = CPD * (c0 + MT + ROW_DW - 1) - 1 - pad;
  int any = occ ? 0 : 1;  // (no occupancy bytes: the late build of the handle API computes every tile)
  if (occ && fr1 >= 0 && fr0 < S && fc1 >= 0 && fc0 < S) {
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
  // horizontal: (64 + ROW_DW - 1)-bit non-zero mask of a row (two ballots), then OR over windows of ROW_DW bits
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
        if (lane < ROW_DW - 1 && c0 + 64 + lane < mpitch) v1[u] = row[c0 + 64 + lane];
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; u++) {
      const unsigned long long b0 = __ballot(v0[u] != 0u), b1 = __ballot(v1[u] != 0u);
      unsigned __int128 m = ((unsigned __int128)b1 << 64) | b0;
      m |= m >> 1;
      m |= m >> 2;
      m |= m >> 4;                                             // windows of 8
      unsigned __int128 mw = m | (m >> 8) | (m >> 13);         // [c, c+16) U [c+13, c+21): windows of 21
      if (CB == 2) mw |= mw >> 21;                             // windows of 42
      if (lane == 0 && rb + u < SK_ROWS) sH[rb + u] = (unsigned long long)mw;
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

// ---- max-pooled tables (bounds of the branch-and-bound matcher, nhip_bnb.hip) --------------------------
// Level 1 (ST = 8): pool[i][j] = max of the stored cells [8i, 8i + 15) x [8j, 8j + 15) (clipped to the image), one
// byte: the largest value an 8 x 8 block of translations can read for a point whose window origin has
// (row >> 3, col >> 3) = (i - Y, j - X).  Level 2 (ST = 4): [4i, 4i + 7) x [4j, 4j + 7), the same for a 4 x 4
// sub-block, stored as byte pairs {P4[i][j], P4[i + 1][j]} (the two sub-block rows of a block in one read).
// 16-bit cells are scaled to a byte by ceil(max / 257), so 257 * pool >= max.
// One block per (band of 8 pooled rows, segment of 512 stored dwords, target): every thread walks the band's
// 8 ST + ST - 1 stored rows down its dword columns keeping eight running maxima (a stored row feeds at most two
// pooled rows), the column maxima go to LDS and a (2 ST - 1)-cell horizontal max finishes the entries.  Columns whose
// 64 x 64 blur tiles are all unoccupied hold only zeros and are not read (~80 % of a scan's image).
constexpr int POOL_BAND = 8;       // pooled rows per block
constexpr int POOL_SEG_DW = 512;   // stored dwords per column segment
constexpr int POOL_HALO_DW = 4;    // >= 7 cells * 2 bytes / 4

typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b) {
  const us2 r = __builtin_elementwise_max(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b));
  return __builtin_bit_cast(uint32_t, r);
}

template <int CB, int ST>
__global__ __launch_bounds__(256) void grid_pool_kernel(const uint8_t *__restrict__ occ, uint8_t *__restrict__ grids,
                                                        int32_t S, int32_t tiles, int32_t pad, int32_t rows,
                                                        int32_t pitch, int64_t table_offset, int64_t slot_bytes,
                                                        int32_t pool_pitch, int32_t t_base) {
  constexpr int WIN = 2 * ST - 1;                    // cells a pooled entry spans per axis
  constexpr int ROWS_IN = POOL_BAND * ST + ST - 1;   // stored rows a band of pooled rows reads
  // column maxima per pooled row: 8-bit cells as two half-word planes (even bytes, odd bytes), 16-bit cells as is
  __shared__ uint32_t sM[POOL_BAND][CB == 1 ? 2 : 1][POOL_SEG_DW + POOL_HALO_DW];
  constexpr int CPD = 4 / CB;  // cells per dword
  const int32_t t = t_base + blockIdx.z, band = blockIdx.x, seg = blockIdx.y, tid = threadIdx.x;
  const uint8_t *g = grids + (size_t)t * slot_bytes;
  uint8_t *pool = grids + (size_t)t * slot_bytes + table_offset;
  const int32_t ndw = pitch / 4;
  const int32_t dw0 = seg * POOL_SEG_DW, dw1 = min(dw0 + POOL_SEG_DW + POOL_HALO_DW, ndw);
  const int32_t r0 = band * POOL_BAND * ST;
  // blur tiles the band's rows can touch
  const int32_t ty0 = max(r0 - pad, 0) / TILE, ty1 = min(r0 + ROWS_IN - 1 - pad, S - 1) / TILE;
  const bool rows_in = r0 + ROWS_IN - 1 - pad >= 0 && r0 - pad < S;
  int any = 0;
  for (int32_t c = dw0 + tid; c < dw1; c += 256) {
    uint32_t me[POOL_BAND], mo[POOL_BAND];
#pragma unroll
    for (int i = 0; i < POOL_BAND; i++) me[i] = mo[i] = 0u;
    const int32_t rc = c * CPD - pad;  // raster column of the dword's first cell (a dword never straddles tiles)
    bool live = rows_in && rc >= 0 && rc < S;
    if (live) {
      int o = 0;
      for (int32_t ty = ty0; ty <= ty1; ty++) o |= occ[((size_t)t * tiles + ty) * tiles + rc / TILE];
      live = o != 0;
    }
    if (live) {
#pragma unroll
      for (int rr = 0; rr < ROWS_IN; rr++) {
        // (rows past the image re-read its last row: zero border)
        const uint32_t w = reinterpret_cast<const uint32_t *>(g + (size_t)min(r0 + rr, rows - 1) * pitch)[c];
        const uint32_t we = CB == 1 ? (w & 0x00ff00ffu) : w, wo = CB == 1 ? ((w >> 8) & 0x00ff00ffu) : 0u;
        // pooled rows i with ST i <= rr < ST i + WIN
#pragma unroll
        for (int i = 0; i < POOL_BAND; i++) {
          if (ST * i <= rr && rr < ST * i + WIN) {
            me[i] = pk_max_u16(me[i], we);
            if (CB == 1) mo[i] = pk_max_u16(mo[i], wo);
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < POOL_BAND; i++) {
      sM[i][0][c - dw0] = me[i];
      if (CB == 1) sM[i][1][c - dw0] = mo[i];
      any |= (me[i] | mo[i]) != 0u;
    }
  }
  if (!__syncthreads_or(any)) return;  // the memset's zeros stand
  const int32_t cells = rows;          // stored columns = stored rows (square image)
  const int32_t nj = (cells + ST - 1) / ST;
  constexpr int JSEG = POOL_SEG_DW * CPD / ST;  // pooled entries per segment
  const int32_t j0 = seg * JSEG;
  for (int32_t e = tid; e < POOL_BAND * JSEG; e += 256) {
    const int32_t i = e / JSEG, j = j0 + e % JSEG;
    if (j >= nj || band * POOL_BAND + i >= (rows + ST - 1) / ST) continue;
    uint32_t m = 0;
    const int32_t c1 = min(j * ST + WIN, cells);
    for (int32_t c = j * ST; c < c1; c++) {
      const int32_t d = c / CPD - dw0;
      uint32_t v;
      if (CB == 1) v = (sM[i][c & 1][d] >> (8 * (c & 2))) & 0xffu;  // byte c&3 of the dword: plane c&1, half-word (c>>1)&1
      else v = (sM[i][0][d] >> (16 * (c & 1))) & 0xffffu;
      m = max(m, v);
    }
    if (m) {
      const uint8_t v = (uint8_t)(CB == 1 ? m : (m + 256u) / 257u);
      const int32_t pi = band * POOL_BAND + i;
      if (ST == BNB_B) {
        pool[(size_t)pi * pool_pitch + j] = v;
      } else {  // pairs: (i, 2j) = P4[i][j], (i, 2j + 1) = P4[i + 1][j]
        pool[(size_t)pi * pool_pitch + 2 * j] = v;
        if (pi > 0) pool[(size_t)(pi - 1) * pool_pitch + 2 * j + 1] = v;
      }
    }
  }
}

// Level 2 (ST = 4) driven by the blur's TILE LIST (round 4): the image is non-zero only inside the listed 64 x 64 tiles,
// so only entries whose 7 x 7 window meets a listed tile can be non-zero -- per tile the 17 x 17 entries that start
// between 4 cells before it and its last cell (the table is pre-zeroed; an entry two tiles share is written twice with
// the same byte).  One workgroup per list entry: the 71 x 71 stored cells under those windows to LDS, 7-cell maxima down
// the columns, then along the rows.  grid_pool_kernel<CB, 4> launched a block per (band, segment, target) -- 88,000 per
// 1000 targets, four fifths of which found their columns unoccupied and left: 0.22 ms against 0.08 here.
constexpr int P4_NE = TILE / BNB_B4 + 1;            // 17 entries per axis
constexpr int P4_REG = BNB_B4 * (P4_NE - 1) + 2 * BNB_B4 - 1;  // 71 cells per axis
template <int CB>
__global__ __launch_bounds__(256) void grid_pool4_tiles_kernel(const int32_t *__restrict__ count, const int32_t *__restrict__ list,
                                                               int32_t tiles, uint8_t *__restrict__ grids, int32_t pad,
                                                               int32_t rows, int32_t pitch, int64_t table_offset,
                                                               int64_t slot_bytes, int32_t pool_pitch, int32_t has_image,
                                                               int64_t hi_offset, int32_t hi_tpr, int64_t hi_copy_bytes,
                                                               int32_t t16_tpr) {
  static_assert(BNB_B4 == 4, "windows of seven cells at stride four");
  // two cells per LDS word (16-bit cells as stored; 8-bit cells widened), 36 words per row of 71 (+ 1) cells
  constexpr int RW = (P4_REG + 1) / 2;
  __shared__ uint32_t sC[P4_REG][RW + 1];
  __shared__ uint32_t sV[P4_NE][RW + 1];
  const int32_t n_entries = *count, tid = threadIdx.x;
  const int32_t n4 = (rows + BNB_B4 - 1) / BNB_B4;  // pooled rows = pooled columns (square image)
  for (int32_t e = blockIdx.x; e < n_entries; e += gridDim.x) {
    const int32_t entry = list[e];
    const int32_t t = entry / (tiles * tiles), tile = entry % (tiles * tiles);
    // stored row / column of the region's first cell (pad is a multiple of 4 and >= 16: never negative, 4-aligned)
    const int32_t r0 = (tile / tiles) * TILE + pad - BNB_B4, c0 = (tile % tiles) * TILE + pad - BNB_B4;
    const uint8_t *g = grids + (size_t)t * slot_bytes;
    uint8_t *pool = grids + (size_t)t * slot_bytes + table_offset;
    __syncthreads();  // the previous entry is done with the LDS arrays
    for (int32_t i = tid; i < P4_REG * RW; i += 256) {
      const int32_t rr = i / RW, w = i - rr * RW;
      const int32_t sr = r0 + rr, sc = c0 + 2 * w;  // (even: the pitch covers whole pairs of cells)
      uint32_t v = 0u;                              // (windows are clipped to the image)
      if (sr < rows && sc < rows) {
        // (two cells at an even column: one dword of the image, or -- NHIP_GRID_NO_IMAGE -- of the matcher's tiled copy of
        //  the cells, where an even pair never straddles a tile row)
        if (CB == 2) {
          v = has_image ? *reinterpret_cast<const uint32_t *>(g + (size_t)sr * pitch + 2 * sc)
                        : *reinterpret_cast<const uint32_t *>(g + hi_offset + 2 * hi_copy_bytes + t16_tiled((uint32_t)sr, (uint32_t)sc, (uint32_t)t16_tpr));
        } else {
          const uint32_t h = has_image ? *reinterpret_cast<const uint16_t *>(g + (size_t)sr * pitch + sc)
                                       : *reinterpret_cast<const uint16_t *>(g + hi_offset + hi_tiled((uint32_t)sr, (uint32_t)sc, 0u, (uint32_t)hi_tpr, (uint32_t)hi_copy_bytes));
          v = (h & 0xffu) | ((h & 0xff00u) << 8);
        }
      }
      sC[rr][w] = v;
    }
    __syncthreads();
    for (int32_t i = tid; i < P4_NE * RW; i += 256) {
      const int32_t pi = i / RW, w = i - pi * RW;
      uint32_t m = 0u;
#pragma unroll
      for (int k = 0; k < 2 * BNB_B4 - 1; k++) m = pk_max_u16(m, sC[BNB_B4 * pi + k][w]);
      sV[pi][w] = m;
    }
    __syncthreads();
    for (int32_t i = tid; i < P4_NE * P4_NE; i += 256) {
      const int32_t a = i / P4_NE, b = i - a * P4_NE;
      // cells 4b .. 4b + 6: words 2b, 2b + 1, 2b + 2 whole and the low half of word 2b + 3
      const uint32_t m3 = pk_max_u16(pk_max_u16(sV[a][2 * b], sV[a][2 * b + 1]), sV[a][2 * b + 2]);
      uint32_t m = max(m3 & 0xffffu, m3 >> 16);
      m = max(m, sV[a][2 * b + 3] & 0xffffu);
      const int32_t pi = r0 / BNB_B4 + a, pj = c0 / BNB_B4 + b;
      if (m == 0u || pi >= n4 || pj >= n4) continue;
      const uint8_t v = (uint8_t)(CB == 1 ? m : (m + 256u) / 257u);
      // pairs: (i, 2j) = P4[i][j], (i, 2j + 1) = P4[i + 1][j]
      pool[(size_t)pi * pool_pitch + 2 * pj] = v;
      if (pi > 0) pool[(size_t)(pi - 1) * pool_pitch + 2 * pj + 1] = v;
    }
  }
}

// Level 1 from level 2: the window [8i, 8i + 15) x [8j, 8j + 15) of a level-1 entry is exactly the union of the nine
// level-2 windows [4a, 4a + 7) x [4b, 4b + 7), a = 2i .. 2i + 2, b = 2j .. 2j + 2, and a maximum of maxima is the
// maximum (ceil(. / 257) is monotone, so the scaled bytes of 16-bit cells commute with it too): the 36 KB table is
// derived from the 286 KB one instead of from a second pass over the image (0.37 -> 0.03 ms per 1000 targets).
__global__ __launch_bounds__(256) void grid_pool8_from_pool4_kernel(uint8_t *__restrict__ grids, int32_t rows, int64_t pool_offset,
                                                                   int64_t pool4_offset, int64_t slot_bytes, int32_t pool_pitch,
                                                                   int32_t pool4_pitch, int32_t t_base) {
  // one thread per four entries (i, 4q .. 4q + 3): five dwords of each of three level-2 rows in, one dword out
  const int32_t t = t_base + blockIdx.z, i = 4 * blockIdx.y + (threadIdx.x >> 6), q = blockIdx.x * 64 + (threadIdx.x & 63);
  const int32_t n8 = (rows + BNB_B - 1) / BNB_B;  // pooled rows = pooled columns (square image)
  if (i >= n8 || 4 * q >= n8) return;
  uint8_t *g = grids + (size_t)t * slot_bytes;
  const uint8_t *p4 = g + pool4_offset;
  uint32_t m[4] = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int da = 0; da < 3; da++) {
    // byte (a, 2 b) = P4[a][b]: for entry j, b = 2j, 2j + 1, 2j + 2 are bytes 4j, 4j + 2, 4j + 4 of row a
    const uint32_t *row = reinterpret_cast<const uint32_t *>(p4 + (size_t)(2 * i + da) * pool4_pitch) + 4 * q;
    uint32_t w[5];
#pragma unroll
    for (int d = 0; d < 5; d++) w[d] = row[d];
#pragma unroll
    for (int e = 0; e < 4; e++) m[e] = max(m[e], max(max(w[e] & 0xffu, (w[e] >> 16) & 0xffu), w[e + 1] & 0xffu));
  }
  // (entries past the table's n8 columns come out 0: their level-2 bytes are)
  *reinterpret_cast<uint32_t *>(g + pool_offset + (size_t)i * pool_pitch + 4 * q) = m[0] | (m[1] << 8) | (m[2] << 16) | (m[3] << 24);
}

// Level 1 from level 2 for the listed tiles only: the level-1 entries that read a level-2 entry a listed tile can have
// written (rows pi0 .. pi0 + 16 of level 2 feed rows (pi0 - 1) >> 1 .. (pi0 + 16) >> 1 of level 1: ten), each the full
// maximum of its nine level-2 entries, whoever wrote those.  Everything else in the table is zero (cleared the same way).
__device__ __forceinline__ void p8_range(int32_t p0, int32_t *lo, int32_t *n) {
  *lo = (p0 - 1) >> 1;
  *n = ((p0 + P4_NE - 1) >> 1) - *lo + 1;
}
__global__ __launch_bounds__(128) void grid_pool8_tiles_kernel(const int32_t *__restrict__ count, const int32_t *__restrict__ list,
                                                               int32_t tiles, uint8_t *__restrict__ grids, int32_t pad, int32_t rows,
                                                               int64_t pool_offset, int64_t pool4_offset, int64_t slot_bytes,
                                                               int32_t pool_pitch, int32_t pool4_pitch) {
  const int32_t n_entries = *count;
  const int32_t n8 = (rows + BNB_B - 1) / BNB_B;
  for (int32_t e = blockIdx.x; e < n_entries; e += gridDim.x) {
    const int32_t entry = list[e];
    const int32_t t = entry / (tiles * tiles), tile = entry % (tiles * tiles);
    const int32_t pi0 = ((tile / tiles) * TILE + pad) / BNB_B4 - 1, pj0 = ((tile % tiles) * TILE + pad) / BNB_B4 - 1;
    int32_t i0, ni, j0, nj;
    p8_range(pi0, &i0, &ni);
    p8_range(pj0, &j0, &nj);
    uint8_t *g = grids + (size_t)t * slot_bytes;
    const uint8_t *p4 = g + pool4_offset;
    for (int32_t k = threadIdx.x; k < ni * nj; k += 128) {
      const int32_t i = i0 + k / nj, j = j0 + k % nj;
      if (i >= n8 || j >= n8) continue;
      uint32_t m = 0u;
#pragma unroll
      for (int da = 0; da < 3; da++) {
        const uint8_t *row = p4 + (size_t)(2 * i + da) * pool4_pitch + 4 * j;  // byte (a, 2 b) = P4[a][b]
        m = max(m, max(max((uint32_t)row[0], (uint32_t)row[2]), (uint32_t)row[4]));
      }
      g[pool_offset + (size_t)i * pool_pitch + j] = (uint8_t)m;
    }
  }
}

// ---- incremental rebuild ---------------------------------------------------------------------------------
// A build leaves in the workspace the list of (target slot, 64 x 64 tile) entries it wrote and, in the header, a tag
// of the buffer it wrote them to.  nhip_grid_rebuild_dev clears exactly those tiles (image, and the plane of high
// bytes of 16-bit grids) instead of zero-filling gigabytes: ~20 % of a dense scan's tiles hold anything.  The tag is
// checked ON THE DEVICE (no host round trip): a header that does not describe this very buffer -- fresh or recycled
// workspace memory, another buffer, another geometry -- makes the same kernel clear everything instead.
constexpr uint64_t GRID_TAG_SEED = 0x9e3779b97f4a7c15ull;
inline uint64_t grid_tag(const void *d_grids, int64_t n_targets, const GridLayout &L, int32_t flags) {
  uint64_t h = GRID_TAG_SEED;
  const uint64_t v[6] = {(uint64_t)(uintptr_t)d_grids, (uint64_t)n_targets, (uint64_t)L.slot_bytes, (uint64_t)L.S,
                         (uint64_t)L.cb, (uint64_t)flags};
  for (uint64_t x : v) {
    h ^= x + GRID_TAG_SEED + (h << 6) + (h >> 2);
    h *= 0xff51afd7ed558ccdull;
  }
  return h | 1ull;  // (never 0: a zeroed header is never valid)
}

// W = bytes per store the tiles' row starts allow ((pad * cell bytes) mod 16; tile columns are multiples of 64 cells)
template <int W>
__device__ __forceinline__ void zero_store(uint8_t *p) {
  if (W == 16) *reinterpret_cast<uint4 *>(p) = make_uint4(0, 0, 0, 0);
  else if (W == 8) *reinterpret_cast<uint2 *>(p) = make_uint2(0, 0);
  else *reinterpret_cast<uint32_t *>(p) = 0u;
}

// rows [r0, r0 + 64) x bytes [col_byte, col_byte + row_bytes) of a plane of pitch `pitch`, clipped to the raster's rows
template <int W>
__device__ __forceinline__ void zero_tile(uint8_t *plane, int32_t pitch, int32_t r0, int32_t pad, int32_t S, int32_t col_byte,
                                          int32_t row_bytes) {
  const int per_row = row_bytes / W;
  for (int i = threadIdx.x; i < TILE * per_row; i += 256) {
    const int r = i / per_row, d = i % per_row;
    if (r0 + r < S) zero_store<W>(plane + (size_t)(r0 + r + pad) * pitch + col_byte + W * d);
  }
}

template <int W, int WH>
__global__ __launch_bounds__(256) void grid_clear_kernel(const int32_t *__restrict__ header, uint64_t expect,
                                                         const int32_t *__restrict__ list, uint8_t *__restrict__ grids,
                                                         int32_t n_targets, int32_t S, int32_t tiles, int32_t pad,
                                                         int32_t pitch, int32_t cb, int64_t slot_bytes, int64_t table_offset,
                                                         int64_t table_bytes, int64_t hi_offset, int32_t hi_tpr,
                                                         int64_t hi_copy_bytes, int32_t t16_tpr, int64_t p4_offset,
                                                         int32_t p4_pitch, int64_t p8_offset, int32_t p8_pitch,
                                                         int64_t hits_offset, int64_t hits_bytes, int32_t has_image,
                                                         const uint32_t *__restrict__ masks) {
  // masks: null, or per list entry the lines of the tiled planes the previous build wrote inside the tile (grid_blur_kernel)
  const uint64_t tag = *reinterpret_cast<const uint64_t *>(header + 2);
  if (tag != expect) {  // unknown contents: everything goes (16-byte stores, grid-stride)
    uint4 *p = reinterpret_cast<uint4 *>(grids);
    const int64_t n16 = (int64_t)n_targets * slot_bytes / 16;  // (slot_bytes is a multiple of 16)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) p[i] = make_uint4(0, 0, 0, 0);
    return;
  }
  // (1) the tiles the previous build wrote: 64 rows x 64 cells of the image and, for 16-bit cells, of the plane of high
  // bytes (a tile's last columns may lie in the raster's zero border: clearing them again is harmless)
  const int32_t n_entries = header[0];
  for (int32_t e = blockIdx.x; e < n_entries; e += gridDim.x) {
    const int32_t entry = list[e];
    const int32_t t = entry / (tiles * tiles), tile = entry % (tiles * tiles);
    const int32_t r0 = (tile / tiles) * TILE, c0 = (tile % tiles) * TILE;
    uint8_t *g = grids + (size_t)t * slot_bytes;
    if (has_image) zero_tile<W>(g, pitch, r0, pad, S, (c0 + pad) * cb, TILE * cb);
    if (p4_pitch > 0) {
      // the second-level entries this tile's cells can have reached (grid_pool4_tiles_kernel: 17 x 17 entries from four
      // cells before the tile, each also the second byte of the pair one row up): 18 rows x 17 byte pairs.  Entries
      // elsewhere are zero already -- the image is non-zero only inside listed tiles -- so the table as a whole
      // (286 KB per slot at 1200 x 1200) is not rewritten.
      const int32_t pi0 = (r0 + pad) / BNB_B4 - 1, pj0 = (c0 + pad) / BNB_B4 - 1;
      for (int i = threadIdx.x; i < (P4_NE + 1) * P4_NE; i += 256) {
        const int32_t pi = pi0 - 1 + i / P4_NE, pj = pj0 + i % P4_NE;
        if (pi >= 0) *reinterpret_cast<uint16_t *>(g + p4_offset + (size_t)pi * p4_pitch + 2 * pj) = 0;
      }
      // ... and the first-level entries grid_pool8_tiles_kernel wrote for it
      int32_t i0, ni, j0, nj;
      p8_range(pi0, &i0, &ni);
      p8_range(pj0, &j0, &nj);
      for (int i = threadIdx.x; i < ni * nj; i += 256) g[p8_offset + (size_t)(i0 + i / nj) * p8_pitch + j0 + i % nj] = 0;
    }
    {
      // the tile's 64 x 64 cells in the matcher's tiled planes, 16 bytes (one tile row) a store where the tiles allow:
      // pad and c0 are multiples of 16 columns here (W == 16), so per row the first copy of the high bytes takes four
      // whole tile rows, the shifted copy half a tile row + three whole + half, the 16-bit copy eight whole ones
      uint8_t *hp = g + hi_offset;
      const uint32_t tpr = (uint32_t)hi_tpr, cpb = (uint32_t)hi_copy_bytes, cc = (uint32_t)(c0 + pad);
      if (((c0 + pad) & 15) == 0) {
        const int per = cb == 2 ? 17 : 9;  // (8-bit cells: no tiled 16-bit copy)
        // the lines the previous build wrote (all of them without masks): bits 0..31 first copy (row of eight x 4 + column),
        // 32..71 shifted copy (x 5), 96..159 16-bit copy (x 8)
        uint32_t mk[GRID_WS_MASK_WORDS];
#pragma unroll
        for (int k = 0; k < GRID_WS_MASK_WORDS; k++) mk[k] = masks ? masks[(size_t)e * GRID_WS_MASK_WORDS + k] : 0xffffffffu;
        for (int i = threadIdx.x; i < TILE * per; i += 256) {
          // (eight consecutive threads take the eight rows of one tile = the eight 16-byte pieces of one 128-byte line:
          //  row-by-row order sent every line to memory as eight partial writes)
          const int r = 8 * (i / (8 * per)) + (i & 7), d = (i >> 3) % per;
          if (r0 + r >= S) continue;
          const uint32_t row = (uint32_t)(r0 + r + pad), lr = (uint32_t)r >> 3;
          if (d < 4) {
            if (!((mk[0] >> (lr * 4u + (uint32_t)d)) & 1u)) continue;
            *reinterpret_cast<uint4 *>(hp + hi_tiled(row, cc + 16u * (uint32_t)d, 0u, tpr, cpb)) = make_uint4(0, 0, 0, 0);
          } else if (d < 9) {
            const int e = d - 4;  // columns cc + 16 e - 8 ... of the plain plane = a tile row of the shifted copy
            const uint32_t b1 = lr * 5u + (uint32_t)e;
            if (!(((b1 < 32u ? mk[1] : mk[2]) >> (b1 & 31u)) & 1u)) continue;
            uint8_t *q = hp + hi_tiled(row, cc + 16u * (uint32_t)e, 1u, tpr, cpb) - 8;
            if (e == 0) *reinterpret_cast<uint2 *>(q + 8) = make_uint2(0, 0);
            else if (e == 4) *reinterpret_cast<uint2 *>(q) = make_uint2(0, 0);
            else *reinterpret_cast<uint4 *>(q) = make_uint4(0, 0, 0, 0);
          } else {
            const uint32_t b2 = lr * 8u + (uint32_t)(d - 9);
            if (!(((b2 < 32u ? mk[3] : mk[4]) >> (b2 & 31u)) & 1u)) continue;
            *reinterpret_cast<uint4 *>(hp + 2 * hi_copy_bytes + t16_tiled(row, cc + 8u * (uint32_t)(d - 9), (uint32_t)t16_tpr)) = make_uint4(0, 0, 0, 0);
          }
        }
      } else {  // (odd geometries: a dword / four cells at a time)
        for (int i = threadIdx.x; i < TILE * (TILE / 4) * 3; i += 256) {
          const int k = i % 3, d = (i / 3) % (TILE / 4), r = (i / 3) / (TILE / 4);
          if (r0 + r >= S) continue;
          const uint32_t row = (uint32_t)(r0 + r + pad), col = cc + 4u * (uint32_t)d;
          if (k < 2) *reinterpret_cast<uint32_t *>(hp + hi_tiled(row, col, (uint32_t)k, tpr, cpb)) = 0u;
          else if (cb == 2) *reinterpret_cast<uint2 *>(hp + 2 * hi_copy_bytes + t16_tiled(row, col, (uint32_t)t16_tpr)) = make_uint2(0u, 0u);
        }
      }
    }
  }
  // (2) the derived tables of every slot (skip map, both pooled tables: between the image and the plane of high bytes)
  const int64_t per_slot = table_bytes / 16;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per_slot * n_targets; i += (int64_t)gridDim.x * 256)
    *reinterpret_cast<uint4 *>(grids + (i / per_slot) * slot_bytes + table_offset + 16 * (i % per_slot)) = make_uint4(0, 0, 0, 0);
  // (3) the hit rasters, whole: 200 KB per slot in 16-byte stores.  (Tile by tile -- two dwords per row and tile, every
  // row another 128-byte line -- the same bits cost 0.10 ms per 1000 targets as partial line writes; this way 0.04.)
  const int64_t hits16 = hits_bytes / 16;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < hits16 * n_targets; i += (int64_t)gridDim.x * 256)
    *reinterpret_cast<uint4 *>(grids + (i / hits16) * slot_bytes + hits_offset + 16 * (i % hits16)) = make_uint4(0, 0, 0, 0);
}

__global__ void grid_tag_kernel(int32_t *header, uint64_t tag) { *reinterpret_cast<uint64_t *>(header + 2) = tag; }

template <int CB, int ST>
void launch_pool(const uint8_t *occ, uint8_t *g, const GridLayout &L, int32_t tiles, int32_t n, hipStream_t s) {
  const int32_t rows = L.S + 2 * L.pad, mpitch = L.pitch / 4;
  const int32_t pooled_rows = (rows + ST - 1) / ST;
  const int64_t off = L.grid_bytes + L.skip_bytes + (ST == BNB_B ? 0 : L.pool_bytes);
  const int32_t pp = ST == BNB_B ? L.pool_pitch : L.pool4_pitch;
  for (int32_t z0 = 0; z0 < n; z0 += 65535) {  // gridDim.z is limited to 65,535
    const int32_t nz = n - z0 < 65535 ? n - z0 : 65535;
    const dim3 pg((pooled_rows + POOL_BAND - 1) / POOL_BAND, (mpitch + POOL_SEG_DW - 1) / POOL_SEG_DW, nz);
    hipLaunchKernelGGL((grid_pool_kernel<CB, ST>), pg, dim3(256), 0, s, occ, g, L.S, tiles, L.pad, rows, L.pitch, off,
                       L.slot_bytes, pp, z0);
  }
}

void launch_pool8_from_pool4(uint8_t *g, const GridLayout &L, int32_t n, hipStream_t s) {
  const int32_t rows = L.S + 2 * L.pad, n8 = (rows + BNB_B - 1) / BNB_B;
  const int64_t off8 = L.grid_bytes + L.skip_bytes, off4 = off8 + L.pool_bytes;
  for (int32_t z0 = 0; z0 < n; z0 += 65535) {  // gridDim.z is limited to 65,535
    const int32_t nz = n - z0 < 65535 ? n - z0 : 65535;
    hipLaunchKernelGGL(grid_pool8_from_pool4_kernel, dim3((n8 + 255) / 256, (n8 + 3) / 4, nz), dim3(256), 0, s, g, rows, off8, off4,
                       L.slot_bytes, L.pool_pitch, L.pool4_pitch, z0);
  }
}

}  // namespace

int launch_grid_build(const float *d_xy, const int32_t *d_offsets, int32_t n_scans, const int32_t *d_target_ids,
                      int32_t n_targets, const nhip_grid_spec_t *spec, const GridLayout &L,
                      uint8_t *d_grids, void *d_ws, int64_t ws_bytes, hipStream_t s, bool incremental) {
  uint32_t *const status = dev_status();
  NHIP_REQUIRE(L.R <= MAX_R, "grid_build: blur radius %d > %d (sigma too large)", L.R, MAX_R);
  NHIP_REQUIRE(L.K * L.K < (1ll << 32), "grid_build: tap sum overflows 32-bit accumulation");
  NHIP_REQUIRE(L.pitch % 4 == 0, "grid_build: pitch must be a multiple of 4");
  const int tiles = (L.S + TILE - 1) / TILE;
  // workspace: 256-byte header (list counter) | [16-bit cells: threshold table] | tile occupancy bytes | tile list
  const int64_t per = grid_ws_per_target(L.S);
  const int64_t fixed = GRID_WS_HEADER + (L.cb == 2 ? GRID_WS_THR16 : 0);
  const int64_t chunk = (ws_bytes - fixed - 4) / per;
  NHIP_REQUIRE(chunk >= 1, "grid_build: workspace %lld B < one target (%lld B)",
               (long long)ws_bytes, (long long)(per + fixed + 4));
  GridTables T;
  int rc = make_tables(spec, L, &T);
  if (rc) return rc;
  GridKernelTables kt;
  memset(&kt, 0, sizeof(kt));
  for (int i = 0; i <= 2 * L.R; i++) kt.taps[i] = T.taps[i];
  for (int i = 0; i < 256; i++) kt.thr[i] = T.thr[i];
  if (L.cb == 2) {
    // fit of the guess through two entries at the top of the table, where the integer thresholds are large and their
    // rounding does not matter (through thr16[16384] = 9 the guess was 75 steps off); a hint only: the table decides
    const double t1 = (double)T.thr16[57344], t2 = (double)T.thr16[65535];
    if (t1 >= 1.0 && t2 > t1) {
      const double a = (65535.0 - 57344.0) / (log(t2) - log(t1));
      kt.q16_a = (float)a;
      kt.q16_b = (float)(57344.0 - a * log(t1));
    }
  }
  uint32_t *d_thr16 = nullptr;
  if (L.cb == 2) {  // the table travels with the launch (the caller owns the workspace; nothing is allocated here)
    d_thr16 = reinterpret_cast<uint32_t *>(static_cast<uint8_t *>(d_ws) + GRID_WS_HEADER);
    NHIP_TRY_HIP(hipMemcpyAsync(d_thr16, T.thr16, GRID_WS_THR16, hipMemcpyHostToDevice, s));
  }
  // one pass over all targets leaves a complete tile list behind: only then can the next build be incremental
  const bool one_pass = chunk >= n_targets;
  const uint64_t tag = grid_tag(d_grids, n_targets, L, spec->flags);
  timer_begin(NHIP_TIMER_GRID, s);
  for (int64_t t0 = 0; t0 < n_targets; t0 += chunk) {
    const int32_t n = (int32_t)((n_targets - t0 < chunk) ? (n_targets - t0) : chunk);
    uint8_t *base = static_cast<uint8_t *>(d_ws);
    int32_t *count = reinterpret_cast<int32_t *>(base);
    uint8_t *occ = base + fixed;
    const size_t occ_bytes = (size_t)n * tiles * tiles;
    int32_t *list = reinterpret_cast<int32_t *>(occ + ((occ_bytes + 3) & ~(size_t)3));
    // line masks behind the list, for geometries whose tiles start on line boundaries of the tiled planes
    uint32_t *masks = L.pad % 16 == 0 ? reinterpret_cast<uint32_t *>(list + (size_t)n * tiles * tiles) : nullptr;
    uint8_t *g = d_grids + (size_t)t0 * L.slot_bytes;
    timer_begin(NHIP_TIMER_GRID_CLEAR, s);
    if (incremental && one_pass) {
      // the tiles the previous build wrote (or everything, if the header does not vouch for this buffer), then the
      // derived tables between the image and the plane of high bytes: skip map and the two pooled tables, every slot
      // (without a skip map -- 16-bit grids unless the spec asks for one -- only the second-level table: nothing reads
      //  the map's space, and the first-level table is rewritten entry by entry from the second)
      const bool with_map = L.has_image && (L.cb == 1 || (spec->flags & NHIP_GRID_SKIP_MAP));
      const int64_t hio = L.grid_bytes + L.skip_bytes + L.pool_bytes + L.pool4_bytes;
      // (without a map the second-level table is cleared tile by tile: p4p > 0)
      const int64_t tb = with_map ? L.skip_bytes + L.pool_bytes + L.pool4_bytes : 0;
      const int64_t to = with_map ? L.grid_bytes : L.grid_bytes + L.skip_bytes + L.pool_bytes;
      const int64_t p4o = L.grid_bytes + L.skip_bytes + L.pool_bytes;
      const int32_t p4p = with_map ? 0 : L.pool4_pitch;
      const int w = (L.pad * L.cb) % 16 == 0 ? 16 : ((L.pad * L.cb) % 8 == 0 ? 8 : 4), wh = L.pad % 16 == 0 ? 16 : (L.pad % 8 == 0 ? 8 : 4);
#define NHIP_CLEAR(W, WH)                                                                                             \
  hipLaunchKernelGGL((grid_clear_kernel<W, WH>), dim3(4096), dim3(256), 0, s, count, tag, list, g, n, L.S, tiles, L.pad, \
                     L.pitch, L.cb, L.slot_bytes, to, tb, hio, L.hi_tpr, L.hi_copy_bytes, L.t16_tpr, p4o, p4p, \
                     L.grid_bytes + L.skip_bytes, L.pool_pitch, hio + L.hi_bytes, L.hits_bytes, L.has_image ? 1 : 0, masks)
      if (w == 16 && wh == 16) NHIP_CLEAR(16, 16);
      else if (w == 16) NHIP_CLEAR(16, 4);
      else if (w == 8) NHIP_CLEAR(8, 4);
      else NHIP_CLEAR(4, 4);
#undef NHIP_CLEAR
    } else {
      NHIP_TRY_HIP(hipMemsetAsync(g, 0, (size_t)n * L.slot_bytes, s));
    }
    // counter (and tag: the buffer is in flux until this build is through) and occupancy
    NHIP_TRY_HIP(hipMemsetAsync(base, 0, GRID_WS_HEADER, s));
    const double inv_res = 1.0 / spec->res;
    const int32_t n_tiles_total = n * tiles * tiles;
    if (tiles * tiles <= 32 * OCC_WORDS_MAX) {
      timer_end(NHIP_TIMER_GRID_CLEAR, s);
      hipLaunchKernelGGL(grid_occupancy_list_kernel, dim3(n), dim3(256), 0, s,
                         reinterpret_cast<const float2 *>(d_xy), d_offsets, d_target_ids,
                         (int32_t)t0, occ, L.S, tiles, L.R, spec->res, inv_res, count, list, n_scans, status);
    } else {
      NHIP_TRY_HIP(hipMemsetAsync(occ, 0, occ_bytes, s));
      timer_end(NHIP_TIMER_GRID_CLEAR, s);
      hipLaunchKernelGGL(grid_occupancy_kernel, dim3(n), dim3(256), 0, s,
                         reinterpret_cast<const float2 *>(d_xy), d_offsets, d_target_ids,
                         (int32_t)t0, occ, L.S, tiles, L.R, spec->res, inv_res, n_scans, status);
      hipLaunchKernelGGL(grid_tile_list_kernel, dim3((n_tiles_total + 255) / 256), dim3(256), 0, s, occ,
                         n_tiles_total, count, list);
    }
    const int32_t blur_blocks = n_tiles_total < 8192 ? n_tiles_total : 8192;  // persistent over the list
    const int32_t rows = L.S + 2 * L.pad, mpitch = L.pitch / 4;
    if (L.cb == 1)
      hipLaunchKernelGGL(grid_blur_kernel<1>, dim3(blur_blocks), dim3(256), 0, s,
                         reinterpret_cast<const float2 *>(d_xy), d_offsets, d_target_ids, (int32_t)t0, count, list,
                         tiles, g, L.S, L.pad, L.pitch, L.slot_bytes, L.R, spec->res, inv_res, kt, d_thr16,
                         L.grid_bytes + L.skip_bytes + L.pool_bytes + L.pool4_bytes, L.hi_tpr, L.hi_copy_bytes, 0, n_scans,
                         L.grid_bytes + L.skip_bytes + L.pool_bytes + L.pool4_bytes + L.hi_bytes, L.hits_pitch, L.has_image ? 1 : 0, masks);
    else
      hipLaunchKernelGGL(grid_blur_kernel<2>, dim3(blur_blocks), dim3(256), 0, s,
                         reinterpret_cast<const float2 *>(d_xy), d_offsets, d_target_ids, (int32_t)t0, count, list,
                         tiles, g, L.S, L.pad, L.pitch, L.slot_bytes, L.R, spec->res, inv_res, kt, d_thr16,
                         L.grid_bytes + L.skip_bytes + L.pool_bytes + L.pool4_bytes, L.hi_tpr, L.hi_copy_bytes, L.t16_tpr, n_scans,
                         L.grid_bytes + L.skip_bytes + L.pool_bytes + L.pool4_bytes + L.hi_bytes, L.hits_pitch, L.has_image ? 1 : 0, masks);
    // gridDim.z is limited to 65,535: the targets of a chunk go in slices.  (The skip map serves the kernels that
    // perform every add; the branch-and-bound matcher never reads it, so 16-bit grids -- its product path -- carry
    // one only when the spec asks.)
    const bool want_map = L.has_image && (L.cb == 1 || (spec->flags & NHIP_GRID_SKIP_MAP));
    for (int32_t z0 = 0; z0 < n && want_map; z0 += 65535) {
      const int32_t nz = n - z0 < 65535 ? n - z0 : 65535;
      const dim3 mg((mpitch + MT - 1) / MT, (rows + MT - 1) / MT, nz);
      if (L.cb == 1)
        hipLaunchKernelGGL(grid_skipmap_kernel<1>, mg, dim3(256), 0, s, occ, g, L.S, tiles, L.pad, L.pitch, rows,
                           L.grid_bytes, L.slot_bytes, z0);
      else
        hipLaunchKernelGGL(grid_skipmap_kernel<2>, mg, dim3(256), 0, s, occ, g, L.S, tiles, L.pad, L.pitch, rows,
                           L.grid_bytes, L.slot_bytes, z0);
    }
    {
      // second-level table from the listed tiles (NHIP_GRID_POOL=bands: the band kernel, measurement), first from second
      const char *pk = tunable("NHIP_GRID_POOL");
      const int64_t off4 = L.grid_bytes + L.skip_bytes + L.pool_bytes;
      if (pk && pk[0] == 'b' && L.has_image) {  // (the band kernels walk the image)
        if (L.cb == 1) launch_pool<1, BNB_B4>(occ, g, L, tiles, n, s);
        else launch_pool<2, BNB_B4>(occ, g, L, tiles, n, s);
      } else if (L.cb == 1) {
        hipLaunchKernelGGL(grid_pool4_tiles_kernel<1>, dim3(blur_blocks), dim3(256), 0, s, count, list, tiles, g, L.pad, rows,
                           L.pitch, off4, L.slot_bytes, L.pool4_pitch, L.has_image ? 1 : 0, off4 + L.pool4_bytes, L.hi_tpr,
                           L.hi_copy_bytes, L.t16_tpr);
      } else {
        hipLaunchKernelGGL(grid_pool4_tiles_kernel<2>, dim3(blur_blocks), dim3(256), 0, s, count, list, tiles, g, L.pad, rows,
                           L.pitch, off4, L.slot_bytes, L.pool4_pitch, L.has_image ? 1 : 0, off4 + L.pool4_bytes, L.hi_tpr,
                           L.hi_copy_bytes, L.t16_tpr);
      }
      const bool want_map8 = L.has_image && (L.cb == 1 || (spec->flags & NHIP_GRID_SKIP_MAP));
      if ((pk && pk[0] == 'b' && L.has_image) || want_map8 || !(incremental && one_pass)) {
        // (the whole table from the whole second-level table: first builds -- whose memset covers it anyway, but the
        //  handle API's late builds have no list -- and grids with a map, whose clear zeroes every derived table)
        launch_pool8_from_pool4(g, L, n, s);
      } else {
        hipLaunchKernelGGL(grid_pool8_tiles_kernel, dim3(blur_blocks), dim3(128), 0, s, count, list, tiles, g, L.pad, rows,
                           L.grid_bytes + L.skip_bytes, off4, L.slot_bytes, L.pool_pitch, L.pool4_pitch);
      }
    }
    if (one_pass) hipLaunchKernelGGL(grid_tag_kernel, dim3(1), dim3(1), 0, s, count, tag);
  }
  timer_end(NHIP_TIMER_GRID, s);
  NHIP_TRY_HIP(hipGetLastError());
  return NHIP_OK;
}

int launch_skipmap_build(uint8_t *d_grids, int32_t n_grids, const GridLayout &L, hipStream_t s) {
  const int32_t rows = L.S + 2 * L.pad, mpitch = L.pitch / 4;
  const int tiles = (L.S + TILE - 1) / TILE;
  for (int32_t z0 = 0; z0 < n_grids; z0 += 65535) {
    const int32_t nz = n_grids - z0 < 65535 ? n_grids - z0 : 65535;
    const dim3 mg((mpitch + MT - 1) / MT, (rows + MT - 1) / MT, nz);
    if (L.cb == 1)
      hipLaunchKernelGGL(grid_skipmap_kernel<1>, mg, dim3(256), 0, s, static_cast<const uint8_t *>(nullptr), d_grids, L.S,
                         tiles, L.pad, L.pitch, rows, L.grid_bytes, L.slot_bytes, z0);
    else
      hipLaunchKernelGGL(grid_skipmap_kernel<2>, mg, dim3(256), 0, s, static_cast<const uint8_t *>(nullptr), d_grids, L.S,
                         tiles, L.pad, L.pitch, rows, L.grid_bytes, L.slot_bytes, z0);
  }
  NHIP_TRY_HIP(hipGetLastError());
  return NHIP_OK;
}

}  // namespace nhip
