// nhip_csm.hip -- K2 + K3: exhaustive (theta, x, y) correlation and argmax on gfx950.
//
// Replaces CorrelativeScanMatcher::GetTransformation (call site
// src/optimization/solver.cc:633-638), batched over candidate pairs.
//
// Formulation (accumulator-stationary, LDS-tiled, one wave per workgroup): a 64-thread workgroup
// owns one 21-row strip of the (nx x ny) score plane of one rotation k of one pair and keeps its
// integer accumulators in registers -- three lanes per y-shift, 28 consecutive x-shifts each.
// A point's contribution to the strip is the (81 x 21) window of the target grid anchored at its
// rotated cell.  Per lane-chunk of 64 points the wave computes the rotated cells (one point per
// lane) and looks up the target's skip map (nhip_grid.hip): points whose window strip holds only
// zeros are dropped -- they would add nothing, so the sums are unchanged.  The others are visited
// in beam order; consecutive beams hit neighbouring cells, so a run of points shares one grid
// tile: the wave stages a 48-row x 212-byte tile of the grid in LDS (16-byte reads of HBM/L2,
// once per run), then every point of the run is a wave-uniform LDS offset (v_readlane) from
// which each lane reads its 7 aligned dwords and accumulates them SWAR-style (below).
// LDS pitch 53 dwords makes the 32-lane read groups conflict-free (bank = 7 * lane mod 32).
// All arithmetic is integer: sums are order-independent, hence bit-exact against the oracle.
// One-wave workgroups never wait for each other (a shared tile made a 4-wave workgroup as slow
// as its busiest strip), and the dispatcher balances the unequal strips across the SIMDs.
//
// No bounds checks: grids carry a zero border of pad = 2*max_shift+16 cells and rotated cells
// are clamped to one cell outside the window-overlap range (a clamped point only sees border).
#include "nhip_csm_shared.h"

namespace nhip {

namespace {

using namespace csm;

#ifndef NHIP_WG_WAVES
#define NHIP_WG_WAVES 1
#endif
constexpr int WG_WAVES = NHIP_WG_WAVES;    // strips (waves) that share one tile; 1: waves never wait for each other
constexpr int CSM_THREADS = 64 * WG_WAVES;
constexpr int SEG_DW = 7;                  // aligned dwords a lane reads and accumulates per point
constexpr int SEG_COLS = 4 * SEG_DW;       // 28 x-shifts per lane
constexpr int SEGS = 3;                    // lanes per plane row: 84 aligned bytes >= 81 + 3
constexpr int WAVE_ROWS = 63 / SEGS;       // 21 plane rows per wave (lane 63 idles: rows never straddle waves)
static_assert(WAVE_ROWS == CSM_WAVE_ROWS && SEGS * SEG_DW == CSM_ROW_DW,
              "the skip map (nhip_grid.hip) is built for this wave footprint");
constexpr int PB_NX = SEGS * SEG_COLS - 3; // 81 x-shifts per plane block (84 bytes minus alignment slack)
constexpr int PB_NY = WG_WAVES * WAVE_ROWS; // y-shifts per plane block (21 per wave)
constexpr int LP_DW = 53;                  // LDS tile pitch in dwords (conflict-free: 53 = 21 mod 32)
constexpr int LP = 4 * LP_DW;              // 212 bytes
#ifndef NHIP_TILE_ROWS
#define NHIP_TILE_ROWS (48 * NHIP_WG_WAVES)
#endif
#ifndef NHIP_FILL_INFLIGHT
#define NHIP_FILL_INFLIGHT 4
#endif
constexpr int TILE_ROWS = NHIP_TILE_ROWS;
constexpr int FILL_INFLIGHT = NHIP_FILL_INFLIGHT;  // 16-byte tile-fill loads a lane keeps in flight
constexpr int ROW_BYTES = SEGS * SEG_COLS; // bytes of a tile row one point touches from its aligned start (84)
constexpr int COL_SPAN = LP - ROW_BYTES;   // max (pcol - tile_col0) of a covered point (128)

// ---- SWAR byte accumulation -------------------------------------------------------------
// gfx950 issues plain VOP2 integer ops (v_add_u32, v_and_b32, v_lshrrev_b32) at 2 clk per
// wave64 but every SDWA / VOP3 form (byte-select adds, v_alignbyte, v_bfe, v_perm, v_add3) at
// ~4.2 clk (tools/ubench_valu.hip).  So a lane never extracts bytes per point.  The three lanes
// of a plane row read the 21 ALIGNED dwords that start at (window start & ~3) -- 84 bytes, enough
// for the 81 window columns under any alignment -- 7 dwords each, and do per dword w:
//     even += w & 0x00FF00FF      (two 16-bit fields: sum b0 | sum b2)
//     odd  += w >> 8              (= sum b1 + 256 sum b2 + 65536 sum b3, no overflow <= 255 pts)
// i.e. 4 full-rate ops per 4 lookups.  The byte alignment s = (window start) & 3 is
// wave-uniform, so points are accumulated into one of four register sets (one class-filtered
// sub-loop each) and the sets are unpacked (before any class reaches 255 points) into the lane's 28 window-relative
// 32-bit sums: byte p of a class-s lane is window column 28*seg + p - s, so bytes p < s belong
// to the left neighbour lane and travel there with one wave shuffle each (6 per unpack).
constexpr int FLUSH_POINTS = 255;  // a 16-bit field holds 255 byte values
// Every alignment class has its own register set, so what must stay <= 255 is the number of points
// ADDED PER CLASS since the last unpack (skipped points do not count): a lane-chunk of 64 points is
// started only while no class has more than this many, so a field never exceeds 191 + 64 values.
// With ~600 points added per wave and four classes, most waves unpack once, at the end.
constexpr int FLUSH_START_MAX = FLUSH_POINTS - 64;

struct Swar {
  uint32_t e[4][SEG_DW], o[4][SEG_DW];
};

// Add one point's pre-split dwords into register set SH (14 full-rate adds).
template <int SH>
__device__ __forceinline__ void swar_add(Swar &A, const uint32_t (&te)[SEG_DW], const uint32_t (&to)[SEG_DW]) {
#pragma unroll
  for (int i = 0; i < SEG_DW; i++) {
    A.e[SH][i] += te[i];
    A.o[SH][i] += to[i];
  }
}

// n members of the group share class SH: the 14 plain (full-rate) adds, n times.  The empty
// volatile asm keeps the loop a loop: without it hipcc rewrites it as acc += n * t with
// v_mad_u32_u24, a half-rate VOP3 that costs twice as much in the common n == 1 case
// (and an explicit n == 1 / n > 1 branch pair blows the register allocation: 44 B of scratch).
template <int SH>
__device__ __forceinline__ void swar_add_n(Swar &A, const uint32_t (&te)[SEG_DW], const uint32_t (&to)[SEG_DW], int n) {
#pragma nounroll
  for (int r = 0; r < n; r++) {
    asm volatile("" ::: "memory");
    swar_add<SH>(A, te, to);
  }
}

// All points of the current run segment (lanes in seg_mask), grouped by ALIGNED BASE.
// Points whose window starts fall in the same aligned dword of the same tile row read the very
// same 7 dwords and differ only in their alignment class -- and consecutive beams land in
// neighbouring cells, so on the 1081-beam scans only ~47 % of the points have a base of their
// own.  One ballot finds every lane of the segment that shares the lowest live lane's base; the
// dwords are read and split (w & 0x00FF00FF, w >> 8) ONCE per group and then added into each
// member's class set.  Measured alternatives: one point per iteration
// (no grouping) 79k pairs/s; reading 2-4 points together or a two-buffer software pipeline were
// 0-12 % slower than that (registers cost occupancy; the 4 resident waves already overlap).
__device__ __forceinline__ void swar_segment(Swar &A, const uint8_t *tile_bytes, uint32_t lane_off,
                                             uint32_t vorg, unsigned long long seg_mask) {
  const uint32_t vbase = vorg & ~3u, vcls = vorg & 3u;
  const unsigned long long cm0 = __ballot(vcls == 0u), cm1 = __ballot(vcls == 1u), cm2 = __ballot(vcls == 2u);
  unsigned long long m = seg_mask;
#pragma nounroll
  while (m) {
    const int jj = (int)__builtin_ctzll(m);
    const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int32_t)vbase, jj);
    const unsigned long long same = __ballot(vbase == base) & m;  // includes lane jj
    m &= ~same;
    const uint32_t *p = reinterpret_cast<const uint32_t *>(tile_bytes + base + lane_off);
    uint32_t te[SEG_DW], to[SEG_DW];
#pragma unroll
    for (int i = 0; i < SEG_DW; i++) {
      const uint32_t w = p[i];
      te[i] = w & 0x00ff00ffu;
      to[i] = w >> 8;
    }
    const int n0 = __builtin_popcountll(same & cm0), n1 = __builtin_popcountll(same & cm1);
    const int n2 = __builtin_popcountll(same & cm2);
    const int n3 = __builtin_popcountll(same) - n0 - n1 - n2;
    swar_add_n<0>(A, te, to, n0);
    swar_add_n<1>(A, te, to, n1);
    swar_add_n<2>(A, te, to, n2);
    swar_add_n<3>(A, te, to, n3);
  }
}

__device__ __forceinline__ void swar_clear(Swar &A) {
#pragma unroll
  for (int s = 0; s < 4; s++)
#pragma unroll
    for (int i = 0; i < SEG_DW; i++) A.e[s][i] = A.o[s][i] = 0;
}

// acc[j] += byte sums.  Byte p = 4*i + k of a class-s lane is window column 28*seg + p - s:
// p >= s lands in this lane's acc[p - s]; p < s (at most 3 bytes per class) is column
// 28*seg - (s - p), i.e. acc[28 - (s - p)] of lane - 1, which reads it with a shuffle from
// lane + 1 (same plane row: rows never straddle waves; segment-2 lanes have no right neighbour).
__device__ __forceinline__ void swar_flush(Swar &A, uint32_t (&acc)[SEG_COLS], bool has_right) {
#pragma unroll
  for (int s = 0; s < 4; s++) {
#pragma unroll
    for (int i = 0; i < SEG_DW; i++) {
      const uint32_t ev = A.e[s][i];
      const uint32_t b0 = ev & 0xffffu, b2 = ev >> 16;
      const uint32_t od = A.o[s][i] - (b2 << 8);
      const uint32_t b1 = od & 0xffffu, b3 = od >> 16;
      const uint32_t b[4] = {b0, b1, b2, b3};
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int p = 4 * i + k;
        if (p >= s) {
          acc[p - s] += b[k];
        } else {
          const uint32_t from_right = (uint32_t)__shfl_down((int)b[k], 1, 64);
          acc[SEG_COLS - (s - p)] += has_right ? from_right : 0u;
        }
      }
    }
  }
  swar_clear(A);
}

#ifndef NHIP_WAVES_PER_SIMD
#define NHIP_WAVES_PER_SIMD 4
#endif
// DENSE: ignore the skip maps (NHIP_CSM_DENSE=1) -- a separate instantiation, so profiles list it apart
template <bool VOLUME, bool DENSE>
__global__ __launch_bounds__(CSM_THREADS, NHIP_WAVES_PER_SIMD) void csm_correlate_kernel(CsmParams P) {
  __shared__ uint32_t s_tile[TILE_ROWS * LP_DW];

  // ---- block -> (pair, rotation, plane block); everything of a pair shares an XCD
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
  const int32_t nyb = min(P.ny - oy, PB_NY);  // plane rows of this block (1..21)
  const int32_t row_span = TILE_ROWS - nyb;   // max (prow - tile_row0) of a covered point

  int32_t src = VOLUME ? P.single_src : P.pair_src[pair];
  int32_t slot = VOLUME ? P.single_slot : P.pair_slot[pair];
  // (ids from device memory: a pair whose scan or slot lies outside the caller's counts scores nothing and is reported)
  const bool ids_ok = VOLUME || pair_ids_ok(P.ids, src, slot, pair, threadIdx.x == 0 && w == 0);
  if (!ids_ok) src = slot = 0;
  const int32_t beg = ids_ok ? P.offsets[src] : 0, n_pts = ids_ok ? P.offsets[src + 1] - beg : 0;
  const uint8_t *grid = P.grids + (size_t)slot * P.slot_bytes;
  const uint8_t *skip_map = grid + P.grid_bytes;
  const int32_t mpitch = skip_pitch(P.pitch);
  // search centre in cells; a centre the stored border cannot cover scores nothing
  int32_t cx = VOLUME ? P.single_ox : (P.pair_origin ? P.pair_origin[2 * pair] : 0);
  int32_t cy = VOLUME ? P.single_oy : (P.pair_origin ? P.pair_origin[2 * pair + 1] : 0);
  const bool centre_ok = (abs(cx) + P.hx <= P.max_shift) && (abs(cy) + P.hy <= P.max_shift);

  // rotation k: R(theta0) * R(delta_k), composed in double with individually rounded ops
  const double c0 = P.rot0_cs[2 * pair], s0 = P.rot0_cs[2 * pair + 1];
  const double cd = P.delta_cs[2 * k], sd = P.delta_cs[2 * k + 1];
  const float cf = __double2float_rn(__dsub_rn(__dmul_rn(c0, cd), __dmul_rn(s0, sd)));
  const float sf = __double2float_rn(__dadd_rn(__dmul_rn(s0, cd), __dmul_rn(c0, sd)));

  // lane = 3 * (plane row) + segment; lane 63 idles
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lane_c = lane < 63 ? lane : 62;
  const int dy = wave * WAVE_ROWS + lane_c / SEGS, seg = lane_c % SEGS;
  const bool lane_live = lane < 63;
  const bool has_right = lane_live && seg < SEGS - 1;  // lane + 1 holds the next 28 bytes of the same row
  // lanes past the plane block's rows re-read row 0 (their sums are never used)
  const int dyc = (dy < nyb) ? dy : 0;
  const uint32_t lane_off = (uint32_t)(dyc * LP + seg * SEG_COLS);
  const uint8_t *tile_bytes = reinterpret_cast<const uint8_t *>(s_tile);
  // tile fill: lane -> (row within a 4-row step, 16-byte chunk of the row)
  constexpr int ROW_CH = (LP + 15) / 16;     // 14
  constexpr int FILL_ROWS = 64 / ROW_CH;     // 4 rows per step (lanes 56..63 idle)
  static_assert(TILE_ROWS % (FILL_ROWS * WG_WAVES) == 0, "tile rows must be a whole number of fill steps");

  uint32_t acc[SEG_COLS];
#pragma unroll
  for (int i = 0; i < SEG_COLS; i++) acc[i] = 0;
  Swar A;
  swar_clear(A);

  // current tile: stored-grid rows [t_row0, t_row0 + TILE_ROWS), byte columns [t_col0, t_col0 + LP)
  int32_t t_row0 = 0, t_col0 = 0;
  bool have_tile = false;

  // lane-chunks of 64 points; an unconditional unpack after every run of chunks that added
  // (at most) FLUSH_POINTS points
  int32_t c64 = 0;
  while (c64 < n_pts && centre_ok) {
    int32_t added0 = 0, added1 = 0, added2 = 0, added3 = 0;  // per alignment class
    for (; c64 < n_pts && max(max(added0, added1), max(added2, added3)) <= FLUSH_START_MAX; c64 += 64) {
      const int32_t n = min(n_pts - c64, 64);
      // one point per lane: rotated window cell, and whether this block's strip of its window
      // holds anything but zeros (skip map, nhip_grid.hip)
      uint32_t vcell = 0u, vwork = 0u;  // vwork bit u: strip u of this plane block has work for the point
      if (lane < n) {
        vcell = window_cell(P.xy[beg + c64 + lane], cf, sf, P, ox, oy, cx, cy);
#pragma unroll
        for (int u = 0; u < WG_WAVES; u++) {
          if (u * WAVE_ROWS >= nyb) break;
          const uint32_t bit = DENSE ? 1u : (((uint32_t)skip_map[(size_t)((vcell >> 16) + u * WAVE_ROWS) * mpitch + ((vcell & 0xffffu) >> 5)] >> ((vcell >> 2) & 7u)) & 1u);
          vwork |= bit << u;
        }
      }
      const int32_t vcol = (int32_t)(vcell & 0xffffu), vrow = (int32_t)(vcell >> 16);
      unsigned long long todo = __ballot(vwork != 0u);           // points some wave of the workgroup needs
      const unsigned long long mine = __ballot((vwork >> wave) & 1u);  // points this wave adds
      {
        // class of a point = (window start column) & 3: tile origins are multiples of 16, so it does
        // not depend on the tile the point will be read from
        const uint32_t vc = (uint32_t)vcol & 3u;
        const unsigned long long k0 = __ballot(vc == 0u) & mine, k1 = __ballot(vc == 1u) & mine;
        const unsigned long long k2 = __ballot(vc == 2u) & mine;
        added0 += __builtin_popcountll(k0);
        added1 += __builtin_popcountll(k1);
        added2 += __builtin_popcountll(k2);
        added3 += __builtin_popcountll(mine) - __builtin_popcountll(k0 | k1 | k2);
      }
      while (todo) {
        const int32_t j = (int32_t)__builtin_ctzll(todo);
        // remaining points inside the staged tile; e = first remaining point that is not
        bool cov = have_tile && (uint32_t)(vcol - t_col0) <= (uint32_t)COL_SPAN &&
                   (uint32_t)(vrow - t_row0) <= (uint32_t)row_span;
        unsigned long long miss = ~__ballot(cov) & todo;
        int32_t e = miss ? (int32_t)__builtin_ctzll(miss) : 64;
        if (e == j) {
          // point j is outside: stage a new tile around it, biased along the sweep direction.
          // (This branch never touches the accumulators.)
          const int32_t ja = min(j + 16, n - 1);
          const int32_t cj = __builtin_amdgcn_readlane(vcol, j), rj = __builtin_amdgcn_readlane(vrow, j);
          const int32_t ca = __builtin_amdgcn_readlane(vcol, ja), ra = __builtin_amdgcn_readlane(vrow, ja);
          t_col0 = place(cj, ca, COL_SPAN - 15) & ~15;
          t_row0 = place(rj, ra, row_span);
          have_tile = true;
          const uint8_t *gsrc = grid + (size_t)t_row0 * P.pitch + t_col0;
          __syncthreads();  // single wave: orders the LDS reads of the old tile before the stores
          // Fill: lanes 0..55 move four tile rows per step -- one 16-byte global load per lane (14 per
          // 224-byte row span; t_col0 and the pitch are multiples of 16), FILL_INFLIGHT steps at a
          // time -- then 4-byte LDS stores (the 212-byte LDS pitch that makes the reads conflict-free
          // is not a multiple of 16).  Tile rows past the stored grid re-read its last row; no
          // covered window reaches them.
          if (lane < FILL_ROWS * ROW_CH) {
            // (recomputed here: staging is rare, VGPRs are not); wave w takes steps w, w + WG_WAVES, ...
            const int fr = lane / ROW_CH + FILL_ROWS * wave, fk = lane % ROW_CH;
            const int fill_dw = fr * LP_DW + 4 * fk;
            const uint8_t *lsrc = gsrc + 16 * fk;
            const int32_t last_row = P.rows - 1 - t_row0;
            constexpr int STEP_ROWS = FILL_ROWS * WG_WAVES;  // rows one step of the whole workgroup covers
#pragma unroll
            for (int b = 0; b < TILE_ROWS / STEP_ROWS; b += FILL_INFLIGHT) {
              uint4 v[FILL_INFLIGHT];
#pragma unroll
              for (int u = 0; u < FILL_INFLIGHT; u++) {
                if (STEP_ROWS * (b + u) >= TILE_ROWS) continue;  // (compile time: the last batch may be short)
                const int32_t r = min(STEP_ROWS * (b + u) + fr, last_row);
                v[u] = *reinterpret_cast<const uint4 *>(lsrc + (uint32_t)(r * P.pitch));
              }
#pragma unroll
              for (int u = 0; u < FILL_INFLIGHT; u++)
                if (STEP_ROWS * (b + u) < TILE_ROWS) s_tile[STEP_ROWS * (b + u) * LP_DW + fill_dw] = v[u].x;
              if (fk < ROW_CH - 1) {
#pragma unroll
                for (int u = 0; u < FILL_INFLIGHT; u++) {
                  if (STEP_ROWS * (b + u) >= TILE_ROWS) continue;
                  uint32_t *dst = s_tile + STEP_ROWS * (b + u) * LP_DW + fill_dw;
                  dst[1] = v[u].y;
                  dst[2] = v[u].z;
                  dst[3] = v[u].w;
                }
              }
            }
          }
          __syncthreads();
          cov = (uint32_t)(vcol - t_col0) <= (uint32_t)COL_SPAN &&
                (uint32_t)(vrow - t_row0) <= (uint32_t)row_span;
          miss = ~__ballot(cov) & todo;
          e = miss ? (int32_t)__builtin_ctzll(miss) : 64;  // > j: the new tile covers point j
        }
        // remaining points before e are covered: LDS byte offset of each lane's window start,
        // then the grouped SWAR accumulation
        const uint32_t vorg = (uint32_t)(vrow - t_row0) * LP + (uint32_t)(vcol - t_col0);
        const unsigned long long seg_mask = todo & (e == 64 ? ~0ull : ((1ull << e) - 1ull));
        swar_segment(A, tile_bytes, lane_off, vorg, seg_mask & mine);
        todo &= ~seg_mask;
      }
    }
    swar_flush(A, acc, has_right);
  }

  const int32_t iy = oy + dy;
  const bool row_ok = lane_live && dy < nyb;
  if (VOLUME) {
    if (row_ok) {
#pragma unroll
      for (int i = 0; i < SEG_COLS; i++) {
        const int32_t ix = ox + seg * SEG_COLS + i;
        if (seg * SEG_COLS + i < PB_NX && ix < P.nx)
          P.volume[((size_t)k * P.nx + ix) * P.ny + iy] = (int32_t)acc[i];
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
      if (seg * SEG_COLS + i < PB_NX && ix < P.nx) {
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
  if (lane == 0) atomicMax(&P.keys[pair], best);
}

__global__ void csm_finalize_kernel(const unsigned long long *__restrict__ keys,
                                    const int32_t *__restrict__ pair_src,
                                    const int32_t *__restrict__ offsets, int32_t n_scans, int32_t n_pairs,
                                    int32_t nx, int32_t ny, double Lf, double step,
                                    nhip_match_t *__restrict__ out, int32_t *__restrict__ sums) {
  const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pairs) return;
  const unsigned long long key = keys[i];
  const uint32_t sum = (uint32_t)(key >> 32);
  const uint32_t lin = 0xffffffffu - (uint32_t)key;
  const int32_t src = pair_src[i];
  const int32_t n = id_in(src, n_scans) ? offsets[src + 1] - offsets[src] : 0;  // (an id out of range: the matcher reported it)
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

// ---- NHIP_SEARCH_EXACT_SCORE: the winning pose's score on the UNQUANTISED table -----------------------------------------
// The argmax is found on quantised cells (bit-exact against the oracle; on 1,300 pairs of the bench workload it is also the
// argmax of a double table every time).  The score reported with it, Lf + step * sum / N, carries the cells' rounding: up
// to 2.3e-5 relative at the best-matching pairs of that sample, where |score| is smallest -- outside the north star's
// 1e-5.  This pass recomputes the score of the ONE pose that won the way the reference's table type would give it
// (CImg<double>, cimg_debug.h:19): for each of the scan's points the exact integer blur sum V of the cell it reads --
// from the hit raster the table was blurred from (13 x 13 bits around the cell at sigma = 2) --, ln(max(V / K^2, floor))
// in double, the mean over the points in double: 1081 x 13 dword pairs of a 200 KB raster per pair.
struct ExactParams {
  const float2 *xy;
  const int32_t *offsets;
  const uint8_t *grids;
  const int32_t *pair_src, *pair_slot;
  const double *rot0_cs, *delta_cs;
  const int32_t *pair_origin;
  const int32_t *pair_kbase;  // optional: entry of delta_cs that is pair i's rotation 0 (nhip_bnb_params.h)
  nhip_match_t *out;
  const unsigned long long *keys;  // optional: the search's keys, decoded here (the record's indices, the sum) instead of by csm_finalize_kernel
  int32_t *sums;                   // with keys: where the integer sums go (may be null)
  IdBounds ids;
  int32_t n_pairs, pairs_per_xcd, nx, ny, hx, hy, S, R, hits_pitch, max_shift;
  int64_t slot_bytes, hits_offset;
  double res, inv_res, K2, floor_p, Lf;
  int32_t taps[2 * 16 + 1];
};

// One 256-thread workgroup per pair; a thread takes the points tid, tid + 256, ... (five at most on a 1081-beam scan).  The
// first version ran one wave per pair with a rolled loop over the window's rows: 17 points x 13 dependent round trips per
// lane, 0.44 ms per 10,000 pairs of pure latency.  Here the 2 NR dword loads of a point's window are issued together
// (NR = 2 R + 1 rows, a compile-time constant for the blur radii in use; the generic instantiation loops).
constexpr int EX_THREADS = 256;  // (512 -- three rounds of loads per thread instead of five -- measured slower: 0.140 against 0.131 ms)
template <int NR>
__global__ __launch_bounds__(EX_THREADS) void csm_exact_score_kernel(ExactParams P) {
  __shared__ double s_part[EX_THREADS / 64];
  __shared__ uint32_t s_taps[2 * 16 + 1];  // (indexed by a set bit's position: from LDS, not from the kernel's argument block)
  // Row sums by table: a window row of up to 14 bits is two 7-bit halves, s_lut[h][bits] = sum of the taps of the set
  // bits of half h -- two LDS reads per row, where the loop over set bits ran as long as the wave's fullest lane needed
  // (the pass was 420 vector instructions per point, most of them that loop: 0.145 -> 0.131 ms per 10,000 pairs).
  __shared__ uint32_t s_lut[2][128];
  if (threadIdx.x <= 2 * 16) s_taps[threadIdx.x] = (uint32_t)P.taps[threadIdx.x];
  __syncthreads();
  if (threadIdx.x < 256) {
    const int h = threadIdx.x >> 7, b = threadIdx.x & 127;
    uint32_t t = 0u;
#pragma unroll
    for (int i = 0; i < 7; i++)
      if ((b >> i) & 1) t += s_taps[7 * h + i];  // (entries past 2 R are zero)
    s_lut[h][b] = t;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // consecutive pairs (one target's, usually) on ONE XCD, as in the matcher's kernels: its L2 then fetches the target's hit
  // raster once -- with pair = blockIdx the eight XCDs each fetched it (536 MB of L2 misses per 10,000 pairs for 200 MB
  // of rasters: the pass was bound by them)
  const int32_t pair = (int32_t)(blockIdx.x & 7u) * P.pairs_per_xcd + (int32_t)(blockIdx.x >> 3);
  if (pair >= P.n_pairs) return;
  int32_t src = P.pair_src[pair], slot = P.pair_slot[pair];
  const bool ids_ok = pair_ids_ok(P.ids, src, slot, pair, false);  // (the matcher reported it)
  if (!ids_ok) src = slot = 0;
  const int32_t beg = ids_ok ? P.offsets[src] : 0, n_pts = ids_ok ? P.offsets[src + 1] - beg : 0;
  nhip_match_t m;
  if (P.keys) {  // (csm_finalize_kernel's decoding; the quantised-formula score it would store is what this pass replaces)
    const unsigned long long key = P.keys[pair];
    const uint32_t lin = 0xffffffffu - (uint32_t)key;
    m.iy = (int32_t)(lin % (uint32_t)P.ny);
    m.ix = (int32_t)((lin / (uint32_t)P.ny) % (uint32_t)P.nx);
    m.itheta = (int32_t)(lin / ((uint32_t)P.ny * (uint32_t)P.nx));
    m.score = (float)P.Lf;
    if (threadIdx.x == 0) {
      P.out[pair] = m;
      if (P.sums) P.sums[pair] = (int32_t)(uint32_t)(key >> 32);
    }
  } else {
    m = P.out[pair];
  }
  {
    // a search centre the stored border cannot cover "scores nothing" in every matcher kernel (sum 0, pose 0, score Lf):
    // the record keeps that score -- the real score at pose 0 would contradict the sum beside it
    const int32_t ox = P.pair_origin ? P.pair_origin[2 * pair] : 0, oy = P.pair_origin ? P.pair_origin[2 * pair + 1] : 0;
    if (abs(ox) + P.hx > P.max_shift || abs(oy) + P.hy > P.max_shift) return;
  }
  const int32_t cx = (P.pair_origin ? P.pair_origin[2 * pair] : 0) + m.ix - P.hx;
  const int32_t cy = (P.pair_origin ? P.pair_origin[2 * pair + 1] : 0) + m.iy - P.hy;
  // rotation itheta: R(theta0) * R(delta_k), composed in double with individually rounded ops, as every matcher kernel
  const double c0 = P.rot0_cs[2 * pair], s0 = P.rot0_cs[2 * pair + 1];
  const int32_t kd = m.itheta + (P.pair_kbase ? P.pair_kbase[pair] : 0);
  const double cd = P.delta_cs[2 * kd], sd = P.delta_cs[2 * kd + 1];
  const float cf = __double2float_rn(__dsub_rn(__dmul_rn(c0, cd), __dmul_rn(s0, sd)));
  const float sf = __double2float_rn(__dadd_rn(__dmul_rn(s0, cd), __dmul_rn(c0, sd)));
  const uint8_t *hits = P.grids + (size_t)slot * P.slot_bytes + P.hits_offset;
  const int nr = NR > 0 ? NR : 2 * P.R + 1;
  const uint32_t mask = (1u << nr) - 1u;  // (R <= 15: at most 31 bits)
  double acc = 0.0;
  for (int32_t p = (int32_t)threadIdx.x; p < n_pts; p += EX_THREADS) {
    const float2 q = P.xy[beg + p];
    const float xr = __fsub_rn(__fmul_rn(cf, q.x), __fmul_rn(sf, q.y));
    const float yr = __fadd_rn(__fmul_rn(sf, q.x), __fmul_rn(cf, q.y));
    double L = P.Lf;  // non-finite points and lookups outside the grid contribute the floor
    if ((fabsf(xr) < 1e9f) && (fabsf(yr) < 1e9f)) {
      const double fc = floor_quotient((double)xr, P.res, P.inv_res) + (double)(P.S / 2 + cx);
      const double fr = floor_quotient((double)yr, P.res, P.inv_res) + (double)(P.S / 2 + cy);
      if (fc >= 0.0 && fc < (double)P.S && fr >= 0.0 && fr < (double)P.S) {
        const int32_t col = (int32_t)fc, row = (int32_t)fr;
        const uint32_t bit0 = (uint32_t)(col - P.R + HIT_PAD), sh = bit0 & 31u;  // first bit of the row windows
        const uint8_t *w = hits + (size_t)(row - P.R + HIT_PAD) * P.hits_pitch + 4 * (size_t)(bit0 >> 5);
        uint32_t V = 0u;
        if (NR > 0) {
          // (a row's 64-bit window in ONE load from its 4-byte-aligned address: the pass is bound by its load instructions)
          struct __attribute__((packed, aligned(4))) Win { uint32_t lo, hi; };
          Win win[NR > 0 ? NR : 1];
#pragma unroll
          for (int i = 0; i < NR; i++) win[i] = *reinterpret_cast<const Win *>(w + (size_t)i * P.hits_pitch);
#pragma unroll
          for (int i = 0; i < NR; i++) {
            uint32_t bits = (uint32_t)((((unsigned long long)win[i].hi << 32) | win[i].lo) >> sh) & mask;
            uint32_t rowsum = 0u;
            if (NR <= 14) {
              rowsum = s_lut[0][bits & 127u] + s_lut[1][bits >> 7];
            } else {
              while (bits) {
                rowsum += s_taps[__builtin_ctz(bits)];
                bits &= bits - 1u;
              }
            }
            V += s_taps[i] * rowsum;
          }
        } else {
          for (int i = 0; i < nr; i++) {
            const uint32_t lo = *reinterpret_cast<const uint32_t *>(w + (size_t)i * P.hits_pitch);
            const uint32_t hi = *reinterpret_cast<const uint32_t *>(w + (size_t)i * P.hits_pitch + 4);
            uint32_t bits = (uint32_t)((((unsigned long long)hi << 32) | lo) >> sh) & mask;
            uint32_t rowsum = 0u;
            while (bits) {
              rowsum += s_taps[__builtin_ctz(bits)];
              bits &= bits - 1u;
            }
            V += s_taps[i] * rowsum;
          }
        }
        double v = __ddiv_rn((double)V, P.K2);
        if (v < P.floor_p) v = P.floor_p;
        L = log(v);
      }
    }
    acc += L;
  }
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) {
    const unsigned long long o = shfl_xor_u64(__double_as_longlong(acc), s);
    acc += __longlong_as_double((long long)o);
  }
  if (lane == 0) s_part[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = s_part[0];
#pragma unroll
    for (int w = 1; w < EX_THREADS / 64; w++) tot += s_part[w];
    P.out[pair].score = __double2float_rn(n_pts > 0 ? __ddiv_rn(tot, (double)n_pts) : P.Lf);
  }
}

int launch_csm_exact_score(const float *d_xy, const int32_t *d_offsets, const IdBounds &ids, const uint8_t *d_grids,
                           const nhip_grid_spec_t *spec, const GridLayout &L, const int32_t *d_pair_src,
                           const int32_t *d_pair_slot, const double *d_rot0_cs, const double *d_delta_cs,
                           const int32_t *d_pair_origin, const int32_t *d_pair_kbase, int32_t n_pairs,
                           const nhip_search_t *search, nhip_match_t *d_out, hipStream_t s,
                           const uint64_t *d_keys_to_decode = nullptr, int32_t *d_sums = nullptr) {
  NHIP_REQUIRE(L.R <= 15, "exact score: blur radius %d > 15", L.R);
  GridTables T;
  int rc = make_tables(spec, L, &T);
  if (rc) return rc;
  ExactParams P;
  memset(&P, 0, sizeof(P));
  P.xy = reinterpret_cast<const float2 *>(d_xy);
  P.offsets = d_offsets;
  P.grids = d_grids;
  P.pair_src = d_pair_src;
  P.pair_slot = d_pair_slot;
  P.rot0_cs = d_rot0_cs;
  P.delta_cs = d_delta_cs;
  P.pair_origin = d_pair_origin;
  P.pair_kbase = d_pair_kbase;
  P.out = d_out;
  P.keys = reinterpret_cast<const unsigned long long *>(d_keys_to_decode);
  P.sums = d_sums;
  P.ids = ids;
  P.n_pairs = n_pairs;
  P.pairs_per_xcd = (n_pairs + 7) / 8;
  P.nx = search->nx;
  P.ny = search->ny;
  P.hx = (search->nx - 1) / 2;
  P.hy = (search->ny - 1) / 2;
  P.S = L.S;
  P.R = L.R;
  P.max_shift = spec->max_shift;
  P.hits_pitch = L.hits_pitch;
  P.slot_bytes = L.slot_bytes;
  P.hits_offset = L.grid_bytes + L.skip_bytes + L.pool_bytes + L.pool4_bytes + L.hi_bytes;
  P.res = spec->res;
  P.inv_res = 1.0 / spec->res;
  P.K2 = (double)L.K * (double)L.K;
  P.floor_p = spec->floor_p;
  P.Lf = L.Lf;
  for (int i = 0; i <= 2 * L.R; i++) P.taps[i] = T.taps[i];
  timer_begin(NHIP_TIMER_EXACT_SCORE, s);
  if (L.R == 6) hipLaunchKernelGGL(csm_exact_score_kernel<13>, dim3(8u * (uint32_t)P.pairs_per_xcd), dim3(EX_THREADS), 0, s, P);  // sigma = 2
  else if (L.R == 3) hipLaunchKernelGGL(csm_exact_score_kernel<7>, dim3(8u * (uint32_t)P.pairs_per_xcd), dim3(EX_THREADS), 0, s, P);  // sigma = 1
  else hipLaunchKernelGGL(csm_exact_score_kernel<0>, dim3(8u * (uint32_t)P.pairs_per_xcd), dim3(EX_THREADS), 0, s, P);
  timer_end(NHIP_TIMER_EXACT_SCORE, s);
  NHIP_TRY_HIP(hipGetLastError());
  return NHIP_OK;
}

int check_search(const nhip_grid_spec_t *spec, const GridLayout &L, const nhip_search_t *search, bool exhaustive) {
  NHIP_REQUIRE(search->n_theta >= 1 && (search->n_theta & 1), "search: n_theta must be odd >= 1");
  NHIP_REQUIRE(search->nx >= 1 && (search->nx & 1), "search: nx must be odd >= 1");
  NHIP_REQUIRE(search->ny >= 1 && (search->ny & 1), "search: ny must be odd >= 1");
  NHIP_REQUIRE((search->nx - 1) / 2 <= spec->max_shift && (search->ny - 1) / 2 <= spec->max_shift,
               "search: shifts +-%d/+-%d exceed the grids' max_shift %d", (search->nx - 1) / 2,
               (search->ny - 1) / 2, spec->max_shift);
  NHIP_REQUIRE((int64_t)search->n_theta * search->nx * search->ny < 0x7fffffffll,
               "search: lattice too large for 32-bit linear index");
  NHIP_REQUIRE(L.S + 2 * L.pad < 65536, "search: stored grid side %d does not fit 16-bit cell packing",
               L.S + 2 * L.pad);
  NHIP_REQUIRE(L.pitch % 16 == 0, "search: grid pitch must be a multiple of 16");
  (void)exhaustive;  // (both cell widths have a kernel that performs every add: this file and nhip_csm16.hip)
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
  P.rows = L.S + 2 * L.pad;
  P.max_shift = spec->max_shift;
  P.grid_bytes = L.grid_bytes;
  P.slot_bytes = L.slot_bytes;
  // NHIP_CSM_DENSE=1 switches the zero-strip skipping off (measurement: the same kernel, every add done)
  const char *dense = tunable("NHIP_CSM_DENSE");
  P.dense = ((dense && dense[0] == '1') || (search->flags & NHIP_SEARCH_DENSE)) ? 1 : 0;
  P.res = spec->res;
  P.inv_res = 1.0 / spec->res;
}

}  // namespace

bool csm_takes_exhaustive(const GridLayout &L, const nhip_search_t *search) {
  const char *ex = tunable("NHIP_CSM_EXHAUSTIVE");
  return (search->flags & NHIP_SEARCH_EXHAUSTIVE) || (ex && ex[0] == '1') || !bnb_fits(L, search);
}

void launch_csm_finalize(const uint64_t *d_keys, const int32_t *d_pair_src, const int32_t *d_offsets, int32_t n_scans, int32_t n_pairs,
                         int32_t nx, int32_t ny, const GridLayout &L, nhip_match_t *d_out, int32_t *d_sums, hipStream_t s) {
  hipLaunchKernelGGL(csm_finalize_kernel, dim3((n_pairs + 255) / 256), dim3(256), 0, s,
                     reinterpret_cast<const unsigned long long *>(d_keys), d_pair_src, d_offsets, n_scans, n_pairs, nx, ny, L.Lf,
                     L.step, d_out, d_sums);
}

static int launch_csm_match_quantised(const float *d_xy, const int32_t *d_offsets, const IdBounds &ids, const uint8_t *d_grids,
                                      const nhip_grid_spec_t *spec, const GridLayout &L, const int32_t *d_pair_src,
                                      const int32_t *d_pair_slot, const double *d_rot0_cs, const double *d_delta_cs,
                                      const int32_t *d_pair_origin, int32_t n_pairs, const nhip_search_t *search,
                                      uint64_t *d_keys, nhip_match_t *d_out, int32_t *d_sums, hipStream_t s,
                                      void *d_workspace, int64_t workspace_bytes, const int32_t *d_pair_kbase);

int launch_csm_match(const float *d_xy, const int32_t *d_offsets, const IdBounds &ids, const uint8_t *d_grids,
                     const nhip_grid_spec_t *spec, const GridLayout &L, const int32_t *d_pair_src,
                     const int32_t *d_pair_slot, const double *d_rot0_cs, const double *d_delta_cs,
                     const int32_t *d_pair_origin, int32_t n_pairs, const nhip_search_t *search,
                     uint64_t *d_keys, nhip_match_t *d_out, int32_t *d_sums, hipStream_t s,
                     void *d_workspace, int64_t workspace_bytes, const int32_t *d_pair_kbase) {
  int rc = launch_csm_match_quantised(d_xy, d_offsets, ids, d_grids, spec, L, d_pair_src, d_pair_slot, d_rot0_cs, d_delta_cs,
                                      d_pair_origin, n_pairs, search, d_keys, d_out, d_sums, s, d_workspace, workspace_bytes,
                                      d_pair_kbase);
  if (rc || n_pairs == 0 || !(search->flags & NHIP_SEARCH_EXACT_SCORE)) return rc;
  // (the records are final -- indices and integer sums; the pass replaces their score field)
  // (SEARCH_I_NO_FINALIZE: the search left its keys undecoded -- the small-plane kernel of a chained call; this pass decodes them)
  const bool decode = (search->flags & SEARCH_I_NO_FINALIZE) != 0;
  return launch_csm_exact_score(d_xy, d_offsets, ids, d_grids, spec, L, d_pair_src, d_pair_slot, d_rot0_cs, d_delta_cs,
                                d_pair_origin, d_pair_kbase, n_pairs, search, d_out, s, decode ? d_keys : nullptr, decode ? d_sums : nullptr);
}

static int launch_csm_match_quantised(const float *d_xy, const int32_t *d_offsets, const IdBounds &ids, const uint8_t *d_grids,
                                      const nhip_grid_spec_t *spec, const GridLayout &L, const int32_t *d_pair_src,
                                      const int32_t *d_pair_slot, const double *d_rot0_cs, const double *d_delta_cs,
                                      const int32_t *d_pair_origin, int32_t n_pairs, const nhip_search_t *search,
                                      uint64_t *d_keys, nhip_match_t *d_out, int32_t *d_sums, hipStream_t s,
                                      void *d_workspace, int64_t workspace_bytes, const int32_t *d_pair_kbase) {
  const bool exhaustive = csm_takes_exhaustive(L, search);
  // (per-pair offsets into the rotation table are the branch-and-bound matcher's: an internal caller that passes them has
  //  made sure the lattice is one it takes)
  NHIP_REQUIRE(!d_pair_kbase || !exhaustive, "csm_match: rotation offsets per pair with a lattice the matcher does not take");
  int rc = check_search(spec, L, search, exhaustive);
  if (rc) return rc;
  NHIP_REQUIRE(L.has_image || !exhaustive, "csm_match: this search takes the kernel that performs every add (NHIP_SEARCH_EXHAUSTIVE, or "
               "a lattice beyond the branch-and-bound matcher's envelope), which reads the row-major image the grids were built "
               "without (NHIP_GRID_NO_IMAGE)");
  if (n_pairs == 0) return NHIP_OK;
  if (!exhaustive) {  // branch and bound: the same records, most adds never performed (nhip_bnb.hip)
    int handled = 0;
    rc = launch_csm_bnb(d_xy, d_offsets, ids, d_grids, spec, L, d_pair_src, d_pair_slot, d_rot0_cs, d_delta_cs,
                        d_pair_origin, n_pairs, search, d_keys, d_out, d_sums, s, &handled, d_workspace, workspace_bytes,
                        d_pair_kbase);
    if (rc || handled) return rc;
  }
  // planes of few translations (the coarse level of GetTransformation: 13 x 13): the kernel whose lanes are poses
  // (NHIP_CSM_SMALL=0, measurement / tests: the strip kernels below for these lattices too)
  const char *sm = tunable("NHIP_CSM_SMALL");
  // ... and, for lists of a few pairs, larger planes in tiles of whole rows (the fine level of GetTransformation)
  if ((csm_small_plane_fits(search) || ((search->flags & NHIP_SEARCH_LATENCY) && csm_small_tiled_fits(search, n_pairs, nullptr, nullptr))) &&
      !(sm && sm[0] == '0'))
    return launch_csm_small_match(d_xy, d_offsets, ids, d_grids, spec, L, d_pair_src, d_pair_slot, d_rot0_cs, d_delta_cs,
                                  d_pair_origin, n_pairs, search, d_keys, d_out, d_sums, s);
  if (L.cb == 2)
    return launch_csm16_match(d_xy, d_offsets, ids, d_grids, spec, L, d_pair_src, d_pair_slot, d_rot0_cs, d_delta_cs,
                              d_pair_origin, n_pairs, search, d_keys, d_out, d_sums, s);
  CsmParams P;
  fill_params(P, spec, L, search);
  P.xy = reinterpret_cast<const float2 *>(d_xy);
  P.offsets = d_offsets;
  P.grids = d_grids;
  P.pair_src = d_pair_src;
  P.pair_slot = d_pair_slot;
  P.ids = ids;
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
  if (P.dense)
    hipLaunchKernelGGL((csm_correlate_kernel<false, true>), dim3((uint32_t)blocks), dim3(CSM_THREADS), 0, s, P);
  else
    hipLaunchKernelGGL((csm_correlate_kernel<false, false>), dim3((uint32_t)blocks), dim3(CSM_THREADS), 0, s, P);
  timer_end(NHIP_TIMER_CSM, s);
  hipLaunchKernelGGL(csm_finalize_kernel, dim3((n_pairs + 255) / 256), dim3(256), 0, s, P.keys,
                     d_pair_src, d_offsets, ids.n_scans, n_pairs, P.nx, P.ny, L.Lf, L.step, d_out, d_sums);
  NHIP_TRY_HIP(hipGetLastError());
  return NHIP_OK;
}

int launch_csm_scores(const float *d_xy, const int32_t *d_offsets, const uint8_t *d_grids,
                      const nhip_grid_spec_t *spec, const GridLayout &L, int32_t src, int32_t slot,
                      const double *d_rot0_cs, const double *d_delta_cs, int32_t origin_x,
                      int32_t origin_y, const nhip_search_t *search, int32_t *d_sums,
                      hipStream_t s) {
  int rc = check_search(spec, L, search, true);
  if (rc) return rc;
  NHIP_REQUIRE(L.has_image, "csm_scores: the score volume comes from the kernel that performs every add, which reads the row-major "
               "image the grids were built without (NHIP_GRID_NO_IMAGE)");
  if (L.cb == 2)
    return launch_csm16_scores(d_xy, d_offsets, d_grids, spec, L, src, slot, d_rot0_cs, d_delta_cs, origin_x, origin_y,
                               search, d_sums, s);
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
  if (P.dense)
    hipLaunchKernelGGL((csm_correlate_kernel<true, true>), dim3((uint32_t)blocks), dim3(CSM_THREADS), 0, s, P);
  else
    hipLaunchKernelGGL((csm_correlate_kernel<true, false>), dim3((uint32_t)blocks), dim3(CSM_THREADS), 0, s, P);
  NHIP_TRY_HIP(hipGetLastError());
  return NHIP_OK;
}

}  // namespace nhip
