// nhip_csm16.hip -- K2 + K3 for 16-bit cells: every add of the exhaustive (theta, x, y) correlation.
//
// Replaces CorrelativeScanMatcher::GetTransformation (call site src/optimization/solver.cc:633-638), batched over
// candidate pairs, at the cell width that keeps reported scores within 1e-5 of an unquantised table (DESIGN.md
// section 3).  It is the cross-check of the branch-and-bound matcher at that width (nhip_bnb.hip: same records, bit
// for bit), the path of lattices that matcher does not hold (more than 88 x 88 translations, more rotations than its
// bounds fit in LDS) and the kernel that performs the work SURVEY.md section 8(d) prices.
//
// Formulation: that of csm_correlate_kernel (nhip_csm.hip) -- accumulator-stationary, LDS-tiled, one wave per
// 21-row strip of the (nx x ny) plane of one rotation of one pair, three lanes per plane row with 28 consecutive
// x-shifts each, points visited in beam order as runs that share one staged tile of the target grid, grouped by
// the aligned LDS address their windows start at -- with what two-byte cells change:
//  * a lane's 28 cells are 56 bytes: seven 8-BYTE reads (ds_read_b64, 256 B/clk/CU; the same bytes as ds_read_b32
//    or ds_read2_b64 would take twice the LDS cycles).  8-byte reads must be 8-byte aligned (an unaligned one is
//    replayed at 64 cycles), so windows are grouped by their 8-byte-aligned start and there are again FOUR
//    alignment classes: s = (window start column) & 3 cells;
//  * tile pitch 53 qwords (424 bytes): lane l = 3 * row + segment reads qword slot 53 * row + 7 * segment = 7 l
//    (mod 32) -- a permutation of the 32 slots for each 32-lane group, conflict-free;
//  * no packed fields: a dword w holds two cells; `raw += w; hi += w >> 16` -- three plain VOP2 operations per two
//    lookups -- and sum(lo) = raw - (sum(hi) << 16) modulo 2^32 at the end, exact while a sum stays below 2^32
//    (scans of up to 65,536 points; the API admits 32,768, whose sums fit the int32 it reports).  Nothing overflows in
//    between, so there is no periodic unpack;
//  * cells pair up inside a dword: for odd s a lane's x-shift pairs straddle dwords, so the accumulators come in two
//    PARITY sets of 15 (raw, hi) pairs -- 60 registers; classes 2 and 3 shift the pair index by one, and their
//    first dword (like the low cell of class 1's) belongs to the left neighbour lane's last x-shifts: it is added
//    into pair 14 of the lane's own set and travels there with a wave shuffle at the end.
// LDS is what limits the waves here: a tile of 48 rows x 424 bytes per one-wave workgroup (the 8-bit kernel's shape)
// is 20 KB -- 8 waves per CU.  Four waves (84 plane rows: one 81 x 81 plane) sharing ONE 96-row tile (40.7 KB) put
// 16 waves on a CU, and the shared fills more than pay for the barriers: measured on 3,000 config #2 pairs
// (profiles/r03_csm16_variants.jsonl, tools/c16_variants.sh) 49.6 ms with one wave per 48-row tile, 45.0 / 43.5 with two
// per 64 / 72 rows, 41.1 / 43.0 / 46.4 with four per 120 / 112 / 104 rows (12 waves per CU), 37.9 with four per 96 rows;
// letting hipcc fold the high-half shift into SDWA adds: 40.6.
#include "nhip_csm_shared.h"

namespace nhip {

namespace {

using namespace csm;

#ifndef NHIP_C16_WG_WAVES
#define NHIP_C16_WG_WAVES 4
#endif
#ifndef NHIP_C16_SDWA
#define NHIP_C16_SDWA 0  // 1: let hipcc fold the high-half shift into SDWA adds (measurement)
#endif
constexpr int WG_WAVES = NHIP_C16_WG_WAVES;  // strips (waves) that share one tile
constexpr int THREADS = 64 * WG_WAVES;
constexpr int SEG_QW = 7;                  // aligned qwords a lane reads per point
constexpr int SEG_DW = 2 * SEG_QW;         // 14 dwords = 28 cells
constexpr int SEG_COLS = 2 * SEG_DW;       // 28 x-shifts per lane
constexpr int SEGS = 3;                    // lanes per plane row: 84 aligned cells >= 81 + 3
constexpr int WAVE_ROWS = 63 / SEGS;       // 21 plane rows per wave (lane 63 idles)
static_assert(WAVE_ROWS == CSM_WAVE_ROWS && SEGS * SEG_DW == 2 * CSM_ROW_DW,
              "the skip map (nhip_grid.hip) is built for this wave footprint");
constexpr int PB_NX = SEGS * SEG_COLS - 3; // 81 x-shifts per plane block
constexpr int PB_NY = WG_WAVES * WAVE_ROWS;
constexpr int LP_QW = 53;                  // LDS tile pitch in qwords (conflict-free: 53 = 21 mod 32)
constexpr int LP = 8 * LP_QW;              // 424 bytes
#ifndef NHIP_C16_TILE_ROWS
#define NHIP_C16_TILE_ROWS (24 * NHIP_C16_WG_WAVES)
#endif
#ifndef NHIP_C16_FILL_INFLIGHT
#define NHIP_C16_FILL_INFLIGHT 4
#endif
constexpr int TILE_ROWS = NHIP_C16_TILE_ROWS;
constexpr int FILL_INFLIGHT = NHIP_C16_FILL_INFLIGHT;  // 16-byte tile-fill loads a lane keeps in flight
constexpr int ROW_BYTES = 2 * SEGS * SEG_COLS;  // bytes of a tile row one point touches from its aligned start (168)
constexpr int COL_SPAN = LP - ROW_BYTES;        // max byte offset (2 * pcol - tile_col0) of a covered point (256)
constexpr int PAIRS = SEG_DW + 1;               // (raw, hi) pairs per parity set: 14 + the left neighbour's

// Accumulators of one lane.  Parity 0 (classes 0, 2): pair j < 14 = x-shifts (2j, 2j + 1) of the lane's 28;
// pair 14 = x-shifts (26, 27) of the left neighbour.  Parity 1 (classes 1, 3): pair j < 14 = x-shifts
// (2j - 1, 2j) -- the low half of pair 0 is the left neighbour's x-shift 27; pair 14 = its x-shifts (25, 26).
struct Acc16 {
  uint32_t raw[2][PAIRS], hi[2][PAIRS];
};

__device__ __forceinline__ void acc_clear(Acc16 &A) {
#pragma unroll
  for (int p = 0; p < 2; p++)
#pragma unroll
    for (int j = 0; j < PAIRS; j++) A.raw[p][j] = A.hi[p][j] = 0u;
}

// One point of class S: dword i of the lane's 14 goes to pair i - (S >> 1) of parity set S & 1 (28 full-rate adds).
template <int S>
__device__ __forceinline__ void acc_add(Acc16 &A, const uint32_t (&w)[SEG_DW], const uint32_t (&h)[SEG_DW]) {
  constexpr int PAR = S & 1, SH = S >> 1;
#pragma unroll
  for (int i = 0; i < SEG_DW; i++) {
    constexpr int LEFT = PAIRS - 1;
    const int j = i - SH < 0 ? LEFT : i - SH;
    A.raw[PAR][j] += w[i];
    A.hi[PAR][j] += h[i];
  }
}

// n members of the group share class S.  (The empty asm keeps the loop a loop of plain adds: hipcc would rewrite it
// as acc += n * w with v_mad_u32_u24, a half-rate VOP3 -- twice the cost in the common n == 1 case.)
template <int S>
__device__ __forceinline__ void acc_add_n(Acc16 &A, const uint32_t (&w)[SEG_DW], const uint32_t (&h)[SEG_DW], int n) {
#pragma nounroll
  for (int r = 0; r < n; r++) {
    asm volatile("" ::: "memory");
    acc_add<S>(A, w, h);
  }
}

// The lane's seven qwords of one group.  One asm statement: hipcc would merge neighbouring 8-byte reads into
// ds_read2_b64 (half the bytes per LDS cycle, and banked like 4-byte reads: the pitch is chosen for ds_read_b64).
__device__ __forceinline__ void read_qwords(uint32_t addr, uint32_t (&w)[SEG_DW]) {
  unsigned long long q0, q1, q2, q3, q4, q5, q6;
  asm volatile(
      "ds_read_b64 %0, %7\n\t"
      "ds_read_b64 %1, %7 offset:8\n\t"
      "ds_read_b64 %2, %7 offset:16\n\t"
      "ds_read_b64 %3, %7 offset:24\n\t"
      "ds_read_b64 %4, %7 offset:32\n\t"
      "ds_read_b64 %5, %7 offset:40\n\t"
      "ds_read_b64 %6, %7 offset:48\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "=&v"(q4), "=&v"(q5), "=&v"(q6)
      : "v"(addr)
      : "memory");
  const unsigned long long q[SEG_QW] = {q0, q1, q2, q3, q4, q5, q6};
#pragma unroll
  for (int i = 0; i < SEG_QW; i++) {
    w[2 * i] = (uint32_t)q[i];
    w[2 * i + 1] = (uint32_t)(q[i] >> 32);
  }
}

// All points of the current run segment (lanes in seg_mask), grouped by the 8-byte-aligned LDS offset their windows
// start at: the members of a group read the very same seven qwords and differ only in their class.
__device__ __forceinline__ void acc_segment(Acc16 &A, uint32_t tile_addr, uint32_t lane_off, uint32_t vorg,
                                            unsigned long long seg_mask) {
  const uint32_t vbase = vorg & ~7u, vcls = (vorg >> 1) & 3u;
  const unsigned long long cm0 = __ballot(vcls == 0u), cm1 = __ballot(vcls == 1u), cm2 = __ballot(vcls == 2u);
  unsigned long long m = seg_mask;
#pragma nounroll
  while (m) {
    const int jj = (int)__builtin_ctzll(m);
    const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int32_t)vbase, jj);
    const unsigned long long same = __ballot(vbase == base) & m;  // includes lane jj
    m &= ~same;
    uint32_t w[SEG_DW], h[SEG_DW];
    read_qwords(tile_addr + base + lane_off, w);
#pragma unroll
    for (int i = 0; i < SEG_DW; i++) {
      h[i] = w[i] >> 16;
#if !NHIP_C16_SDWA
      // (opaque: hipcc would fold the shift into every add as an SDWA operand select -- v_add_u32_sdwa issues at
      //  ~4.2 clocks against ~2.3 for the plain add, tools/ubench_valu.hip -- 14 shifts per group are cheaper)
      asm volatile("" : "+v"(h[i]));
#endif
    }
    const int n0 = __builtin_popcountll(same & cm0), n1 = __builtin_popcountll(same & cm1);
    const int n2 = __builtin_popcountll(same & cm2);
    const int n3 = __builtin_popcountll(same) - n0 - n1 - n2;
    acc_add_n<0>(A, w, h, n0);
    acc_add_n<1>(A, w, h, n1);
    acc_add_n<2>(A, w, h, n2);
    acc_add_n<3>(A, w, h, n3);
  }
}

// The lane's 28 sums (x-shifts 28 * segment + i) from the two parity sets; what belongs to the left neighbour comes
// from lane + 1 (same plane row: rows never straddle waves; segment-2 lanes have no right neighbour).
__device__ __forceinline__ void acc_finish(const Acc16 &A, uint32_t (&acc)[SEG_COLS], bool has_right) {
  constexpr int LEFT = PAIRS - 1;
#pragma unroll
  for (int i = 0; i < SEG_COLS; i++) acc[i] = 0u;
#pragma unroll
  for (int j = 0; j < PAIRS; j++) {
    const uint32_t lo0 = A.raw[0][j] - (A.hi[0][j] << 16), hi0 = A.hi[0][j];
    const uint32_t lo1 = A.raw[1][j] - (A.hi[1][j] << 16), hi1 = A.hi[1][j];
    if (j < LEFT) {
      acc[2 * j] += lo0;
      acc[2 * j + 1] += hi0;
      acc[2 * j] += hi1;
      if (j > 0) {
        acc[2 * j - 1] += lo1;
      } else {
        const uint32_t r = (uint32_t)__shfl_down((int)lo1, 1, 64);  // the right neighbour's x-shift -1 = my 27
        acc[SEG_COLS - 1] += has_right ? r : 0u;
      }
    } else {
      const uint32_t r0 = (uint32_t)__shfl_down((int)lo0, 1, 64), r1 = (uint32_t)__shfl_down((int)hi0, 1, 64);
      const uint32_t r2 = (uint32_t)__shfl_down((int)lo1, 1, 64), r3 = (uint32_t)__shfl_down((int)hi1, 1, 64);
      acc[SEG_COLS - 2] += has_right ? r0 : 0u;  // parity 0: x-shifts (-2, -1) of the right neighbour
      acc[SEG_COLS - 1] += has_right ? r1 : 0u;
      acc[SEG_COLS - 3] += has_right ? r2 : 0u;  // parity 1: x-shifts (-3, -2)
      acc[SEG_COLS - 2] += has_right ? r3 : 0u;
    }
  }
}

#ifndef NHIP_C16_WAVES_PER_SIMD
#define NHIP_C16_WAVES_PER_SIMD 4
#endif
// DENSE: the grids carry no skip map (16-bit grids are built without one unless the spec asks: the matcher's product
// path never reads it), or NHIP_CSM_DENSE=1: every strip is added, zero or not
template <bool VOLUME, bool DENSE>
__global__ __launch_bounds__(THREADS, NHIP_C16_WAVES_PER_SIMD) void csm_correlate16_kernel(CsmParams P) {
  __shared__ __align__(16) unsigned long long s_tile[TILE_ROWS * LP_QW];

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
  const int32_t nyb = min(P.ny - oy, PB_NY);  // plane rows of this block
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
  const bool has_right = lane_live && seg < SEGS - 1;
  const int dyc = (dy < nyb) ? dy : 0;  // lanes past the plane block's rows re-read row 0 (their sums are never used)
  const uint32_t lane_off = (uint32_t)(dyc * LP + seg * 2 * SEG_COLS);
  const uint32_t tile_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)s_tile;
  // tile fill: lane -> (row within a 2-row step, 16-byte chunk of the row)
  constexpr int ROW_CH = (LP + 15) / 16;     // 27 (the last one half used)
  constexpr int FILL_ROWS = 64 / ROW_CH;     // 2 rows per step (lanes 54..63 idle)
  static_assert(TILE_ROWS % (FILL_ROWS * WG_WAVES) == 0, "tile rows must be a whole number of fill steps");

  Acc16 A;
  acc_clear(A);

  // current tile: stored-grid rows [t_row0, t_row0 + TILE_ROWS), BYTE columns [t_col0, t_col0 + LP)
  int32_t t_row0 = 0, t_col0 = 0;
  bool have_tile = false;

  for (int32_t c64 = 0; c64 < n_pts && centre_ok; c64 += 64) {
    const int32_t n = min(n_pts - c64, 64);
    // one point per lane: rotated window cell, and whether this block's strip of its window holds anything but
    // zeros (skip map: one bit per stored row and aligned dword; the window's 8-byte-aligned start is an even dword)
    uint32_t vcell = 0u, vwork = 0u;
    if (lane < n) {
      vcell = window_cell(P.xy[beg + c64 + lane], cf, sf, P, ox, oy, cx, cy);
#pragma unroll
      for (int u = 0; u < WG_WAVES; u++) {
        if (u * WAVE_ROWS >= nyb) break;
        const uint32_t pc = vcell & 0xffffu;
        const uint32_t bit = DENSE ? 1u : (((uint32_t)skip_map[(size_t)((vcell >> 16) + u * WAVE_ROWS) * mpitch + (pc >> 4)] >> (((pc >> 2) & 3u) << 1)) & 1u);
        vwork |= bit << u;
      }
    }
    const int32_t vcol = (int32_t)(2u * (vcell & 0xffffu)), vrow = (int32_t)(vcell >> 16);  // (column in bytes)
    unsigned long long todo = __ballot(vwork != 0u);                 // points some wave of the workgroup needs
    const unsigned long long mine = __ballot((vwork >> wave) & 1u);  // points this wave adds
    while (todo) {
      const int32_t j = (int32_t)__builtin_ctzll(todo);
      // remaining points inside the staged tile; e = first remaining point that is not
      bool cov = have_tile && (uint32_t)(vcol - t_col0) <= (uint32_t)COL_SPAN &&
                 (uint32_t)(vrow - t_row0) <= (uint32_t)row_span;
      unsigned long long miss = ~__ballot(cov) & todo;
      int32_t e = miss ? (int32_t)__builtin_ctzll(miss) : 64;
      if (e == j) {
        // point j is outside: stage a new tile around it, biased along the sweep direction
        const int32_t ja = min(j + 16, n - 1);
        const int32_t cj = __builtin_amdgcn_readlane(vcol, j), rj = __builtin_amdgcn_readlane(vrow, j);
        const int32_t ca = __builtin_amdgcn_readlane(vcol, ja), ra = __builtin_amdgcn_readlane(vrow, ja);
        t_col0 = place(cj, ca, COL_SPAN - 15) & ~15;
        t_row0 = place(rj, ra, row_span);
        have_tile = true;
        const uint8_t *gsrc = grid + (size_t)t_row0 * P.pitch + t_col0;
        __syncthreads();  // orders the LDS reads of the old tile before the stores
        // Fill: lanes 0..53 move two tile rows per step -- one 16-byte global load per lane (27 per 432-byte row
        // span; t_col0 and the pitch are multiples of 16), FILL_INFLIGHT steps at a time -- then 8-byte LDS stores
        // (rows of the 424-byte LDS pitch start on 8-byte boundaries).  Tile rows past the stored grid re-read its
        // last row; no covered window reaches them.
        if (lane < FILL_ROWS * ROW_CH) {
          const int fr = lane / ROW_CH + FILL_ROWS * wave, fk = lane % ROW_CH;
          const int fill_qw = fr * LP_QW + 2 * fk;
          const uint8_t *lsrc = gsrc + 16 * fk;
          const int32_t last_row = P.rows - 1 - t_row0;
          constexpr int STEP_ROWS = FILL_ROWS * WG_WAVES;
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
              if (STEP_ROWS * (b + u) < TILE_ROWS)
                s_tile[STEP_ROWS * (b + u) * LP_QW + fill_qw] = ((unsigned long long)v[u].y << 32) | v[u].x;
            if (fk < ROW_CH - 1) {
#pragma unroll
              for (int u = 0; u < FILL_INFLIGHT; u++)
                if (STEP_ROWS * (b + u) < TILE_ROWS)
                  s_tile[STEP_ROWS * (b + u) * LP_QW + fill_qw + 1] = ((unsigned long long)v[u].w << 32) | v[u].z;
            }
          }
        }
        __syncthreads();
        cov = (uint32_t)(vcol - t_col0) <= (uint32_t)COL_SPAN && (uint32_t)(vrow - t_row0) <= (uint32_t)row_span;
        miss = ~__ballot(cov) & todo;
        e = miss ? (int32_t)__builtin_ctzll(miss) : 64;  // > j: the new tile covers point j
      }
      // remaining points before e are covered: LDS byte offset of each lane's window start, then the grouped adds
      const uint32_t vorg = (uint32_t)(vrow - t_row0) * LP + (uint32_t)(vcol - t_col0);
      const unsigned long long seg_mask = todo & (e == 64 ? ~0ull : ((1ull << e) - 1ull));
      acc_segment(A, tile_addr, lane_off, vorg, seg_mask & mine);
      todo &= ~seg_mask;
    }
  }
  uint32_t acc[SEG_COLS];
  acc_finish(A, acc, has_right);

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

void fill_params16(CsmParams &P, const nhip_grid_spec_t *spec, const GridLayout &L, const nhip_search_t *search) {
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
  // without a skip map in the slots (spec->flags) every strip is added; NHIP_CSM_DENSE=1 asks for that too
  const char *dense = tunable("NHIP_CSM_DENSE");
  P.dense = ((dense && dense[0] == '1') || (search->flags & NHIP_SEARCH_DENSE) || !(spec->flags & NHIP_GRID_SKIP_MAP)) ? 1 : 0;
  P.res = spec->res;
  P.inv_res = 1.0 / spec->res;
}

}  // namespace

int launch_csm16_match(const float *d_xy, const int32_t *d_offsets, const IdBounds &ids, const uint8_t *d_grids,
                       const nhip_grid_spec_t *spec, const GridLayout &L, const int32_t *d_pair_src,
                       const int32_t *d_pair_slot, const double *d_rot0_cs, const double *d_delta_cs,
                       const int32_t *d_pair_origin, int32_t n_pairs, const nhip_search_t *search,
                       uint64_t *d_keys, nhip_match_t *d_out, int32_t *d_sums, hipStream_t s) {
  CsmParams P;
  fill_params16(P, spec, L, search);
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
    hipLaunchKernelGGL((csm_correlate16_kernel<false, true>), dim3((uint32_t)blocks), dim3(THREADS), 0, s, P);
  else
    hipLaunchKernelGGL((csm_correlate16_kernel<false, false>), dim3((uint32_t)blocks), dim3(THREADS), 0, s, P);
  timer_end(NHIP_TIMER_CSM, s);
  launch_csm_finalize(d_keys, d_pair_src, d_offsets, ids.n_scans, n_pairs, P.nx, P.ny, L, d_out, d_sums, s);
  NHIP_TRY_HIP(hipGetLastError());
  return NHIP_OK;
}

int launch_csm16_scores(const float *d_xy, const int32_t *d_offsets, const uint8_t *d_grids,
                        const nhip_grid_spec_t *spec, const GridLayout &L, int32_t src, int32_t slot,
                        const double *d_rot0_cs, const double *d_delta_cs, int32_t origin_x, int32_t origin_y,
                        const nhip_search_t *search, int32_t *d_sums, hipStream_t s) {
  CsmParams P;
  fill_params16(P, spec, L, search);
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
    hipLaunchKernelGGL((csm_correlate16_kernel<true, true>), dim3((uint32_t)blocks), dim3(THREADS), 0, s, P);
  else
    hipLaunchKernelGGL((csm_correlate16_kernel<true, false>), dim3((uint32_t)blocks), dim3(THREADS), 0, s, P);
  NHIP_TRY_HIP(hipGetLastError());
  return NHIP_OK;
}

}  // namespace nhip
