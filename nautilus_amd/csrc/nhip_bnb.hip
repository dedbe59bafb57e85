// nhip_bnb.hip -- K2 + K3 by branch and bound: the argmax (and its sum) of the exhaustive (theta, x, y)
// correlation, without performing most of its adds.
//
// Replaces CorrelativeScanMatcher::GetTransformation (call site src/optimization/solver.cc:633-638), batched
// over candidate pairs, with exactly the result of csm_correlate_kernel (nhip_csm.hip): maximum integer sum,
// ties to the smallest linear index (k * nx + ix) * ny + iy.
//
// Bound: the plane of translations of one rotation is cut into 8 x 8 blocks (Y, X).  A point whose window origin
// (top-left lookup cell) is (r, c) reads, for the 64 poses of block (Y, X), stored cells in rows
// [r + 8Y, r + 8Y + 8) and columns [c + 8X, c + 8X + 8), all inside the 15 x 15 cells that the pooled entry
// pool[(r >> 3) + Y][(c >> 3) + X] covers (nhip_grid.hip).  So U(k, Y, X) = sum over points of that entry is an
// upper bound of every sum in the block (16-bit cells: 257 * U, the pool holds ceil(max / 257)).
// Search: (1) U for all blocks of all rotations -- a gather of 11 x 11 bytes per RUN of points that share a pooled
// entry (consecutive beams do: run-length compression cuts the gathers ~5x) from the LDS-resident pooled table;
// (2) every wave evaluates its highest-bound block exactly, which gives a lower bound `best` of the optimum;
// (3) rotation by rotation, best first: every block with U >= best's sum is refined through the bounds of its four
// 4 x 4 sub-blocks (second table, pooled over 7 x 7 cells at stride 4: one 16-byte read per point serves a strip of
// three neighbouring blocks) and exact sums are gathered from the stored grid only for the sub-blocks whose bound
// still reaches `best`, which rises as it goes.  Blocks and sub-blocks with a bound below best's sum cannot hold the
// optimum, nor a tie with it, and are never touched.  On the 1081-beam workload 0.36 % of the poses are evaluated.
//
// Parallelisation is the transpose of the exhaustive kernel's: LANES ARE POINTS (one point per lane and
// 64-point chunk), REGISTERS ARE POSES -- byte sums SWAR-packed two per register -- and one transposing
// reduction (reduce-scatter over the 64 lanes) per rotation / block turns 64 partial sums per pose into one
// total per lane.  All sums are integers: order-independent, bit-exact against the oracle.
//
// One 512-thread workgroup per pair; waves take rotations k = wave, wave + 8, ...; LDS holds the target's
// pooled table (36 KB at 1200 x 1200) and the bounds (n_theta x 128 dwords).
//
// Kernels.  csm_bnb_kernel<CB, POOL_LDS, BY_ROT, SPLIT>: steps (1)-(3) for one pair (the fused form), or with SPLIT
// steps (1)-(2) only, leaving the pair's rows of bounds in the caller's workspace; csm_bnb_order_kernel +
// csm_bnb_cand_kernel<CB>: step (3) of all pairs of a list as a launch of its own (the split form: lists of 6,144 to
// 65,536 pairs -- the first part's workgroups all take the same time, the candidates' vary 100-fold and the heaviest
// pairs are shared by several workgroups); csm_bnb_rot_kernel<CB>: rotations handed over by pairs with flat landscapes
// in small batches.  Exact sums of 16-bit grids run on the plane of high bytes (256 sum(hi) + 255 n bounds a pose's sum;
// only the poses that bound admits read 16-bit cells), which -- like the copy of the cells those reads use -- is stored
// in tiles of one cache line, because what a gather costs here is the number of distinct lines per load.
#include <atomic>
#include <mutex>
#include <vector>

#include "nhip_bnb_params.h"

// This file is compiled twice.  NHIP_BNB_INSTR = 0 (nhip_bnb.hip itself): the product kernels -- no statistics, no
// timestamps, no debug switches in the code -- and the host side.  NHIP_BNB_INSTR = 1 (nhip_bnb_instr.hip includes this
// file): the same kernels with the instrumentation compiled in, launched only when NHIP_BNB_INSTRUMENT=1 is set in
// the environment (tools/bnb_*.py, the tests that count evaluated blocks, bench.py's algorithm.stats).
#ifndef NHIP_BNB_INSTR
#define NHIP_BNB_INSTR 0
#endif
// (The compile-time measurement switches of rounds 2-4 -- lane exchanges through ds_bpermute, runs cut at 8 / 16 lanes,
//  unaligned row loads, highest-bound-first order, unmerged origin lists, per-lane nesting and double-precision forms of
//  the window origins, the row-major 8-bit plane, unaligned LDS reads, the v_fract near test, run lengths at the heads --
//  are out of the source: each lost its A/B, the numbers are in profiles/r03_matcher_experiments.txt and
//  profiles/r04_bounds_variants.txt, the code in the commits those files name.  What is left is what ships.)
#if NHIP_BNB_INSTR
#define csm_bnb_kernel csm_bnb_kernel_instr          // (their own names in profiles)
#define csm_bnb_rot_kernel csm_bnb_rot_kernel_instr
#define BNB_STATS(P) ((P).stats)
#define BNB_TIMELINE(P) ((P).timeline)
#else
#define BNB_STATS(P) (static_cast<unsigned long long *>(nullptr))
#define BNB_TIMELINE(P) (static_cast<unsigned long long *>(nullptr))
#endif

namespace nhip {

namespace {

using namespace bnb;

constexpr int BNB_THREADS = 64 * BNB_WAVES;
// Waves per workgroup of the split form's first kernel (bounds + seeds).  What bounds that kernel is the latency of a
// wave's own instruction chain more than issue slots: with ONE workgroup per CU (two waves per SIMD) it takes 1.63x the
// time of two.  Five waves per SIMD would need workgroups of ten waves at 96 registers; built and twice as slow -- ten
// waves spread 3 + 3 + 2 + 2 over the SIMDs, a second workgroup's would make six on two of them, which 96 registers do
// not allow, so ONE workgroup was resident per CU.  Twelve waves need 80 registers, a third workgroup of eight also 53 KB
// of LDS (profiles/r04_bounds_variants.txt).
constexpr int SPLIT_WAVES = 8;
constexpr int SEG_CHUNKS = 32;        // 64-point chunks between reductions: 32 * 255 * 8 lanes < 65536 (16-bit fields); even
constexpr int EVAL_CHUNKS = 2;  // 64-point chunks whose row loads a block evaluation keeps in flight
constexpr uint32_t M8 = 0x00ff00ffu;

// Window origins from single-precision arithmetic.  The spec's cell is floor(double(v) / res) (cimg_debug.h:31-37: float
// promoted to double, double division).  m = RN(v * RN_f32(1 / res)) differs from the true quotient q by at most
// |q| * 2^-23 (one rounding of the reciprocal, one of the product), and the spec's RN_double(q) by 2^-53 |q| more: the
// floors can differ only if m lies within that distance of an integer.  Lanes within |m| * 2^-22 of one (twice the bound;
// about one coordinate in 2,000 on the 1200-cell grid) take the double-precision path, so the result is the spec's,
// always.  From |m| >= 2^22 on (no fraction bits left to test) the cell is far outside any grid (sides <= 16384) on
// either path and the clamp decides; v_cvt_i32_f32 saturates.
// Window origin (stored-grid row, column of the top-left lookup cell) of point q under rotation (cf, sf): the
// same arithmetic as window_cell of nhip_csm.hip (spec: DESIGN.md section 3, items 1 and 3).
// LEAN (the bounds phase): see below; the candidates' kernels keep round 3's form -- the lean one costs the candidates'
// kernel of 16-bit grids, at its 96 registers, seven spilled dwords and 0.07 ms.
template <bool LEAN = false>
__device__ __forceinline__ void window_origin(float2 q, float cf, float sf, const BnbParams &P, int32_t cx, int32_t cy,
                                              int32_t *prow, int32_t *pcol) {
  const float xr = __fsub_rn(__fmul_rn(cf, q.x), __fmul_rn(sf, q.y));
  const float yr = __fadd_rn(__fmul_rn(sf, q.x), __fmul_rn(cf, q.y));
  const int32_t half = P.S / 2;
  const bool finite __attribute__((unused)) = (fabsf(xr) < 1e9f) && (fabsf(yr) < 1e9f);
  int32_t ix, iy;
  // the floors of both quotients with ONE test for the rare path, taken by the wave only if some lane needs it
  // (about one chunk in 16): the straight-line code has no nested exec masks.  A lane is "near" when either quotient
  // lies within |m| * 2^-22 of an integer -- which includes every |m| >= 2^22 (no fraction bits left), whose floors
  // the double-precision path then takes like any other.
  const float mx = __fmul_rn(xr, P.inv_res_f), my = __fmul_rn(yr, P.inv_res_f);
  const float fx = floorf(mx), fy = floorf(my);
  if (LEAN) {
    // The same test written so that it also holds for what is not a number: !(min(r, 1 - r) > tol) is true for NaN (an
    // infinite quotient: inf - inf), and every |v| >= 1e9 has |m| >= 2^22 at any cell size below 238 m, i.e. r == 0.  So
    // the points the spec calls non-finite all take the rare path, which gives them the floor that clamps to the window
    // position of a point that scores nothing (column -hx - 1, row -hy - 1: the lower clamp bounds), and the straight-line
    // code needs neither the two magnitude compares nor the selects -- one constant per coordinate after the clamp.
    const float rx = __fsub_rn(mx, fx), ry = __fsub_rn(my, fy);  // exact
    const bool slow = !(fminf(rx, __fsub_rn(1.0f, rx)) > __fmul_rn(fabsf(mx), 0x1p-22f)) ||
                      !(fminf(ry, __fsub_rn(1.0f, ry)) > __fmul_rn(fabsf(my), 0x1p-22f));
    ix = (int32_t)fx;
    iy = (int32_t)fy;
    if (__builtin_amdgcn_ballot_w64(slow) != 0ull) {
      if (slow) {
        // (the magnitude test on the promoted values -- 1e9 is a float -- so that it stays on this path)
        const double xd = (double)xr, yd = (double)yr;
        const bool fin = (fabs(xd) < 1e9) && (fabs(yd) < 1e9);
        ix = (int32_t)fmin(fmax(floor_quotient(xd, P.res, P.inv_res), -2147483000.0), 2147483000.0);
        iy = (int32_t)fmin(fmax(floor_quotient(yd, P.res, P.inv_res), -2147483000.0), 2147483000.0);
        if (!fin) ix = iy = -2147483000;
      }
    }
    const int32_t lo_x = -P.hx - 1 - half - cx, lo_y = -P.hy - 1 - half - cy;
    ix = min(max(ix, lo_x), P.S + P.hx - half - cx);
    iy = min(max(iy, lo_y), P.S + P.hy - half - cy);
    *pcol = ix + (half + cx - P.hx + P.pad);
    *prow = iy + (half + cy - P.hy + P.pad);
    return;
  }
  const float rx = __fsub_rn(mx, fx), ry = __fsub_rn(my, fy);  // exact
  const bool near = fminf(rx, __fsub_rn(1.0f, rx)) <= __fmul_rn(fabsf(mx), 0x1p-22f) ||
                    fminf(ry, __fsub_rn(1.0f, ry)) <= __fmul_rn(fabsf(my), 0x1p-22f);
  ix = (int32_t)fx;
  iy = (int32_t)fy;
  if (__builtin_amdgcn_ballot_w64(near && finite) != 0ull) {
    if (near && finite) {
      ix = (int32_t)fmin(fmax(floor_quotient((double)xr, P.res, P.inv_res), -2147483000.0), 2147483000.0);
      iy = (int32_t)fmin(fmax(floor_quotient((double)yr, P.res, P.inv_res), -2147483000.0), 2147483000.0);
    }
  }
  ix = min(max(ix, -P.hx - 1 - half - cx), P.S + P.hx - half - cx);
  iy = min(max(iy, -P.hy - 1 - half - cy), P.S + P.hy - half - cy);
  // col = clamp(S / 2 + floor(xr / res) + cx, -hx - 1, S + hx), as window_cell of nhip_csm.hip -- with the clamp
  // applied to the quotient's floor, so that everything stays in 32-bit arithmetic; non-finite points score nothing
  const int32_t col = finite ? half + ix + cx : -P.hx - 1, row = finite ? half + iy + cy : -P.hy - 1;
  *pcol = col - P.hx + P.pad;
  *prow = row - P.hy + P.pad;
}

// Rows of the stored grid are read through a buffer descriptor of the pair's grid slot: 12 bytes at a 4-byte-aligned
// offset in ONE instruction (buffer_load_dwordx3; hipcc splits the same read through a flat pointer into two
// overlapping 8-byte loads).  Every load instruction costs the L1 one tag lookup per lane and line, and the block
// evaluation is bound by exactly that.
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// Buffer descriptor of a wave-uniform range.  The inputs pass through readfirstlane so that hipcc can PROVE the
// descriptor uniform and keeps it in SGPRs: a descriptor it parks in VGPRs costs a serialising "waterfall" loop of
// ~10 instructions around every single buffer load.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void *base, int64_t bytes) {
  const uint64_t a = reinterpret_cast<uint64_t>(base);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
  const int n = __builtin_amdgcn_readfirstlane((int)(bytes < 0x7fffffffll ? bytes : 0x7fffffffll));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((uint64_t)hi << 32) | lo), 0, n, 0x00020000);
}

__device__ __forceinline__ uint32_t idx_guard(bool live, uint32_t v) { return live ? v : 0u; }

__device__ __forceinline__ uint32_t shfl_xor_u32(uint32_t v, int m) { return (uint32_t)__shfl_xor((int)v, m, 64); }

__device__ __forceinline__ unsigned long long shfl_xor_u64b(unsigned long long v, int m) {
  uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
  lo = shfl_xor_u32(lo, m);
  hi = shfl_xor_u32(hi, m);
  return ((unsigned long long)hi << 32) | lo;
}

// ---- one lane acts for the wave ------------------------------------------------------------------------------
// `if (lane == 0) x = atomicAdd(...); x = readfirstlane(x);` at the head of a loop whose body ends in
// `if (lane == 0) atomicMax(...)` is two tests of ONE value, and hipcc threads the second into the first: lanes 1..63,
// for which both are false, get a loop of their own that bypasses both blocks, and lane 0 is parked until they leave
// it.  readfirstlane is a convergent operation: without lane 0 it returns lane 1's x = 0, the sub-wave takes entry 0
// again and again (lane 0's atomicMax, which would prune it, never runs) and the kernel does not return.  That was the
// hang of the general instantiation under NHIP_BNB_LEVELS=1 once its counters were compiled out (round 3; the
// counters' increments kept the two blocks apart): profiles/r04_general_kernel_hang_isa.txt shows the threaded loop.
// So the lane id of every such test passes through an empty asm: each test is then of a value the compiler knows
// nothing about, and no two of them can be related.
__device__ __forceinline__ bool wave_leader(int lane) {
  asm volatile("" : "+v"(lane));
  return lane == 0;
}
// atomicAdd by one lane, the old value in every lane (wave-uniform, in a scalar register)
__device__ __forceinline__ uint32_t wave_fetch_add(uint32_t *p, uint32_t v, int lane) {
  uint32_t r = 0u;
  if (wave_leader(lane)) r = atomicAdd(p, v);
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)r);
}
// atomicMax of a wave-uniform key by one lane (generic address: LDS or global)
__device__ __forceinline__ void wave_atomic_max(unsigned long long *p, unsigned long long key, int lane) {
  if (wave_leader(lane)) atomicMax(p, key);
}

// One step of a transposing reduction over the lanes: lanes pair up across MASK; of every two registers the
// lane keeps the one its own bit selects, adds the partner's copy of the same register, and gives the other away.
// N registers in, N / 2 out.  No step goes through LDS:
//   MASK 1, 2   partners inside a quad: DPP quad permutation fused into the add;
//   MASK 4, 8   partners inside a row of 16 lanes: two DPP adds with complementary BANK masks (a bank = 4 lanes) --
//               lanes whose bit is clear add register 2i of the lane MASK above (row_ror:16 - MASK), the others
//               register 2i + 1 of the lane MASK below (row_ror:MASK); no select instructions at all;
//   MASK 16, 32 partners in another row / the other half of the wave: v_permlane16_swap / v_permlane32_swap
//               exchange the odd rows (upper half) of register 2i with the even rows (lower half) of register
//               2i + 1, after which the two registers hold own and partner's copy lane by lane: one add.
template <int MASK>
__device__ __forceinline__ uint32_t shfl_xor_c(uint32_t v) {
  if (MASK == 1) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);  // quad_perm [1, 0, 3, 2]
  if (MASK == 2) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);  // quad_perm [2, 3, 0, 1]
  return shfl_xor_u32(v, MASK);
}

template <int MASK>
__device__ __forceinline__ uint32_t rs_pair(uint32_t x, uint32_t y, bool bit) {
  if (MASK == 4) {
    uint32_t r;
    // (s_nop 1: a DPP operand written by the previous vector instruction needs two wait states)
    asm volatile("s_nop 1\n\tv_add_u32_dpp %0, %1, %1 row_ror:12 row_mask:0xf bank_mask:0x5\n\t"
                 "v_add_u32_dpp %0, %2, %2 row_ror:4 row_mask:0xf bank_mask:0xa"
                 : "=&v"(r) : "v"(x), "v"(y));
    return r;
  }
  if (MASK == 8) {
    uint32_t r;
    asm volatile("s_nop 1\n\tv_add_u32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
                 "v_add_u32_dpp %0, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xc"
                 : "=&v"(r) : "v"(x), "v"(y));
    return r;
  }
  if (MASK == 16) {
    const auto sw = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    return sw[0] + sw[1];
  }
  if (MASK == 32) {
    const auto sw = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    return sw[0] + sw[1];
  }
  const uint32_t keep = bit ? y : x, send = bit ? x : y;
  return keep + shfl_xor_c<MASK>(send);
}

template <int N, int MASK, int CAP>
__device__ __forceinline__ void rs_step(uint32_t (&R)[CAP], bool bit) {
  static_assert(N <= CAP, "rs_step: more registers than the array holds");
#pragma unroll
  for (int i = 0; i < N / 2; i++) R[i] = rs_pair<MASK>(R[2 * i], R[2 * i + 1], bit);
}

// sum over the lane pairs MASK apart of one register (the tail of a reduction whose copies may coincide)
template <int MASK>
__device__ __forceinline__ uint32_t add_xor(uint32_t v) {
  if (MASK == 16) {
    const auto sw = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return sw[0] + sw[1];
  }
  if (MASK == 32) {
    const auto sw = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return sw[0] + sw[1];
  }
  return v + shfl_xor_u32(v, MASK);
}

// ---- bounds of one rotation ------------------------------------------------------------------------------
// Returns this lane's two totals of the 128-slot layout: slot v = lane + 64 * i (i = 0, 1) holds packed register
// r = 32 i + 16 b5 + 8 b4 + 4 b3 + 2 b1 + b0, field b2 (b = bits of the lane) -- see slot_block().
// POOL_LDS: the pooled table is staged in LDS (`pool`); otherwise it is read from the grid slot in global memory
// through the buffer descriptor `prs` (tables of large grids, e.g. the 6000 x 6000 grid of the two-level drop-in).
//
// Consecutive beams hit the same wall: on a 1081-beam scan 5 to 10 consecutive points share a pooled entry
// (8 x 8 cells = 40 cm), and every one of them would gather the same 11 x 11 bytes.  So the points are first
// run-length compressed: a lane whose pooled offset differs from its predecessor's (or that starts a 64-point chunk)
// is the head of a run and writes (offset, run length <= 64) to the wave's list in LDS; the gather then works on
// list entries, 64 at a time, and adds every byte `length` times (one multiply-add in place of the add): ~165
// entries for 1081 points, three passes.  (Runs cut at every 8th lane, the first form: 265 entries, five passes.)
// Field widths: the accumulators and the first two reduction steps (over 4 lanes) hold 16-bit fields, so a lane may
// gather a total run length of at most LANE_WEIGHT = 64 between two reductions (4 lanes * 64 * 255 = 65,280); the
// wave reduces early when a pass would take some lane past that, otherwise once per rotation.
constexpr uint32_t LANE_WEIGHT = 64u;
constexpr int LIST_ENTRIES = 128; // ring of pending entries per wave (a chunk appends <= 64, 64 are consumed at a time)

template <bool POOL_LDS>
__device__ __forceinline__ void coarse_rotation(const BnbParams &P, const uint8_t *pool, __amdgpu_buffer_rsrc_t prs,
                                                const float2 *pts, int32_t n_pts, float cf, float sf, int32_t cx,
                                                int32_t cy, int lane, uint32_t *list, uint32_t (&tot)[2]) {
  const int32_t DP = P.pool_pitch;
  const uint32_t zero_a = (uint32_t)(((P.rows + BNB_B - 1) / BNB_B) * DP);  // NB + 1 rows of zeros below the pooled image
  tot[0] = tot[1] = 0u;
  uint32_t E[NB][3], O[NB][3];  // per block row Y: 12 byte sums = dwords 0..2, even (b0 | b2) and odd (raw, see below)
#pragma unroll
  for (int y = 0; y < NB; y++)
#pragma unroll
    for (int d = 0; d < 3; d++) E[y][d] = O[y][d] = 0u;
  uint32_t head = 0u, tail = 0u;  // ring positions (wave-uniform)
  bool pending = false;           // passes gathered since the last reduction
  uint32_t weight = 0u;           // this lane's run lengths gathered since the last reduction

  // 64 list entries: every lane gathers the 11 x 12 bytes of its entry, weighted by the run length
  auto gather = [&](uint32_t a, uint32_t cnt) {
    const uint32_t sh = (a & 3u) * 8u;
    const uint32_t *q = reinterpret_cast<const uint32_t *>(pool + (a & ~3u));
#pragma unroll
    for (int y = 0; y < NB; y++) {
      uint32_t w0, w1, w2, w3;
      if (POOL_LDS) {
        const uint32_t *row = q + (y * DP) / 4;  // DP is a multiple of 16
        w0 = row[0]; w1 = row[1]; w2 = row[2]; w3 = row[3];
      } else {
        const u32x4 r4 = __builtin_amdgcn_raw_buffer_load_b128(prs, (int)((a & ~3u) + (uint32_t)(y * DP)), 0, 0);
        w0 = r4.x; w1 = r4.y; w2 = r4.z; w3 = r4.w;
      }
      const uint32_t n0 = __builtin_amdgcn_alignbit(w1, w0, sh), n1 = __builtin_amdgcn_alignbit(w2, w1, sh);
      const uint32_t n2 = __builtin_amdgcn_alignbit(w3, w2, sh);
      // even: b0 | b2 << 16; odd (raw): w >> 8 = b1 + 256 b2 + 65536 b3, repaired at the reduction
      E[y][0] += __umul24(n0 & M8, cnt); O[y][0] += __umul24(n0 >> 8, cnt);
      E[y][1] += __umul24(n1 & M8, cnt); O[y][1] += __umul24(n1 >> 8, cnt);
      E[y][2] += __umul24(n2 & M8, cnt); O[y][2] += __umul24(n2 >> 8, cnt);
    }
  };
  // the transposing reduction of the 128 packed sums (see the layout above); clears the accumulators
  auto reduce = [&]() {
    // 64 packed registers: R[6 y + d] (y < 10): d < 3 = E[y][d] (X = 4 d, 4 d + 2), d >= 3 = O[y][d - 3] (X = 4 (d - 3) + 1, + 3);
    // the hi field of O[y][2] is X = 11 (unused): rows 0..2 carry X = 5, 7, 9 of block row 10 there.
    // R[60..62] = E[10][0..2], R[63] = O[10][0].
    uint32_t R[64];
#pragma unroll
    for (int y = 0; y < NB; y++)
#pragma unroll
      for (int d = 0; d < 3; d++) O[y][d] -= (E[y][d] >> 16) << 8;  // now b1 | b3 << 16
#pragma unroll
    for (int y = 0; y < 10; y++)
#pragma unroll
      for (int d = 0; d < 3; d++) {
        R[6 * y + d] = E[y][d];
        R[6 * y + 3 + d] = O[y][d];
      }
    R[5] = (O[0][2] & 0xffffu) | (O[10][1] << 16);          // (10, 5)
    R[11] = (O[1][2] & 0xffffu) | (O[10][1] & 0xffff0000u);  // (10, 7)
    R[17] = (O[2][2] & 0xffffu) | (O[10][2] << 16);          // (10, 9)
    R[60] = E[10][0];
    R[61] = E[10][1];
    R[62] = E[10][2];
    R[63] = O[10][0];
    rs_step<64, 1>(R, lane & 1);
    rs_step<32, 2>(R, lane & 2);
    uint32_t V[32];
#pragma unroll
    for (int i = 0; i < 16; i++) {
      V[2 * i] = R[i] & 0xffffu;
      V[2 * i + 1] = R[i] >> 16;
    }
    rs_step<32, 4>(V, lane & 4);
    rs_step<16, 8>(V, lane & 8);
    rs_step<8, 16>(V, lane & 16);
    rs_step<4, 32>(V, lane & 32);
    tot[0] += V[0];
    tot[1] += V[1];
#pragma unroll
    for (int y = 0; y < NB; y++)
#pragma unroll
      for (int d = 0; d < 3; d++) E[y][d] = O[y][d] = 0u;
  };

  // One chunk's pooled offsets `a` (point c + lane; lanes past the scan's end carry the zero rows) into the run list,
  // then the gather passes that have become due.  more == false: no chunk, the list is drained and reduced.
  auto feed = [&](uint32_t a, int32_t c, bool more) {
    if (more) {
      const bool live = c + lane < n_pts;
      // runs of equal offsets inside the chunk: the predecessor's offset by a DPP shift across the wave, no LDS round
      // trip; the chunk's first lane is a head anyway
      const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp((int)a, (int)a, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
      const bool is_head = (lane & 63) == 0 || a != prev;
      // (the lane masks straight from the compares: a ballot of the bools goes through a 0 / 1 register and back)
      const unsigned long long H = __builtin_amdgcn_uicmp(a, prev, 33 /* ne */) | 1ull;
      // (the entry carries its first point's index modulo 128; the gather subtracts it from the next entry's -- one LDS
      //  read per 64 entries instead of two 64-bit shifts, a compare and two bit searches per 64 points.  The last run
      //  ends at the sentinel entry written after the last chunk.)
      uint32_t cnt = (uint32_t)(c + lane) & 127u;
      cnt += 1u;  // (stored as cnt - 1 below)
      const int32_t n_live = n_pts - c;  // (lanes past the scan's end emit nothing)
      const unsigned long long He = H & (n_live >= 64 ? ~0ull : (1ull << n_live) - 1ull);
      if (is_head && live) {
        const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(He >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)He, 0u));
        list[(tail + before) & (LIST_ENTRIES - 1)] = a | ((cnt - 1u) << RUN_SHIFT);
      }
      tail += (uint32_t)__builtin_popcountll(He);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    if (!more) {
      // the sentinel: where the last run ends.  (At most 64 entries are pending here -- the last chunk's turn drained
      // the list below 65 -- so the slot is free.)
      if (lane == 0) list[tail & (LIST_ENTRIES - 1)] = ((uint32_t)n_pts & 127u) << RUN_SHIFT;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    // gather passes: whenever 64 entries are pending (and the one after them, which ends the 64th's run),
    // and to the last entry once the scan is through; then one more turn for the rotation's (only, as a rule)
    // reduction -- one copy of that code
    for (;;) {
      const uint32_t avail = tail - head;
      if (avail < 65u && more) break;
      const bool last = avail == 0u;  // (!more)
      // (lanes past the list gather the zero rows with length 0)
      const bool mine = (uint32_t)lane < avail;
      const uint32_t entry = mine ? list[(head + (uint32_t)lane) & (LIST_ENTRIES - 1)] : zero_a;
      const uint32_t ea = entry & ((1u << RUN_SHIFT) - 1u);
      const uint32_t next = mine ? list[(head + (uint32_t)lane + 1u) & (LIST_ENTRIES - 1)] : 0u;
      const uint32_t cnt = mine ? ((next >> RUN_SHIFT) - (entry >> RUN_SHIFT)) & 127u : 0u;
      // (also before a pass that could overflow some lane's fields)
      if (pending && (last || __ballot(weight + cnt > LANE_WEIGHT) != 0ull)) {
        reduce();
        pending = false;
        weight = 0u;
      }
      if (last) break;
      gather(ea, cnt);
      weight += cnt;
      pending = true;
      head += avail < 64u ? avail : 64u;
      __builtin_amdgcn_wave_barrier();
    }
  };

  // (Tried: the window origins of TWO chunks per turn in one basic block, so that the scheduler interleaves the two
  //  chains -- bounds + seeds 3.18 -> 3.23 ms, profiles/r04_bounds_variants.txt: the chains' latency is hidden already.)
  // (the points of the next PD chunks are in flight while one chunk is worked: with two workgroups per CU gathering
  //  from their grids, a point load takes ~1,300 clocks, more than a chunk's work)
  constexpr int PD = 2;
  float2 qn[PD];
#pragma unroll
  for (int d = 0; d < PD; d++) qn[d] = 64 * d + lane < n_pts ? pts[64 * d + lane] : make_float2(0.f, 0.f);
  for (int32_t c = 0;; c += 64) {
    const bool more = c < n_pts;  // (one more turn after the last chunk drains the list)
    uint32_t a = zero_a;
    if (more) {
      const float2 pt = qn[0];
#pragma unroll
      for (int d = 0; d < PD - 1; d++) qn[d] = qn[d + 1];
      if (c + 64 * PD + lane < n_pts) qn[PD - 1] = pts[c + 64 * PD + lane];
      const bool live = c + lane < n_pts;
      if (live) {
        int32_t prow, pcol;
        window_origin<true>(pt, cf, sf, P, cx, cy, &prow, &pcol);
        // (both factors are below 2^12: rows and pitch of the pooled image; the padding keeps prow positive)
        a = __umul24((uint32_t)prow >> 3, (uint32_t)DP) + ((uint32_t)pcol >> 3);
      }
    }
    feed(a, c, more);
    if (!more) break;
  }
}

// (Tried, commit 9e18995: LANES = (list entry, block row) -- a lane holds 12 byte sums instead of 11 x 12, no transposing
//  reduction, 100 / 80 / 64 registers at 8 / 12 / 16 waves per workgroup.  Same records; bounds + seeds 3.16 -> 4.01 / 3.58 /
//  3.7 ms: decoding an entry per (entry, row) instead of per entry doubles the vector instructions per row, which eats what
//  the missing reduction saves, and six waves per SIMD do not make up for it.  profiles/r04_bounds_variants.txt.)
// (block row Y, block column X) of slot v of the 128-slot layout; false for the unused slots.
__device__ __forceinline__ bool slot_block(int v, int *Y, int *X) {
  const int lane = v & 63, i = v >> 6;
  const int r = 32 * i + 16 * ((lane >> 5) & 1) + 8 * ((lane >> 4) & 1) + 4 * ((lane >> 3) & 1) + 2 * ((lane >> 1) & 1) + (lane & 1);
  const int f = (lane >> 2) & 1;
  if (r >= 60) {
    *Y = 10;
    *X = r == 63 ? 1 + 2 * f : 4 * (r - 60) + 2 * f;
    return true;
  }
  const int y = r / 6, d = r % 6;
  if (d == 5 && f == 1) {  // the relocated values of block row 10
    *Y = 10;
    *X = 5 + 2 * y;
    return y < 3;
  }
  *Y = y;
  *X = d < 3 ? 4 * d + 2 * f : 4 * (d - 3) + 1 + 2 * f;
  return true;
}

// Byte offset (into the grid slot) of the aligned dword that holds the first cell of a point's 8 x 8 patch of block
// (Y, X), and the bit shift of that cell inside it (8-bit cells).  Lanes without a point read the zero border
// (row 0 of the stored image).
__device__ __forceinline__ void patch_origin(const BnbParams &P, bool live, float2 q, float cf, float sf, int32_t cx,
                                             int32_t cy, int32_t Y, int32_t X, uint32_t *g, uint32_t *sh) {
  *g = 0u;
  *sh = 0u;
  if (live) {
    int32_t prow, pcol;
    window_origin(q, cf, sf, P, cx, cy, &prow, &pcol);
    const int32_t col = pcol + BNB_B * X;
    *g = (uint32_t)((prow + BNB_B * Y) * P.pitch + (col & ~3));
    *sh = (uint32_t)(col & 3) * 8u;
  }
}

// ---- exact sums of one 8 x 8 block -----------------------------------------------------------------------
// Returns the block's best key (sum << 32 | ~linear index) over its valid poses, the same in every lane.
template <int CB>
__device__ __forceinline__ unsigned long long eval_block(const BnbParams &P, const uint8_t *grid, const float2 *pts,
                                                         int32_t n_pts, float cf, float sf, int32_t cx, int32_t cy,
                                                         int32_t k, int32_t Y, int32_t X, int lane) {
  uint32_t total = 0u;  // this lane's pose: (dy, dx) below
  int dy, dx;
  // (stored image + skip map: every offset the evaluation can form lies inside; see nhip_api.hip make_layout)
  const __amdgpu_buffer_rsrc_t rsrc = uniform_rsrc(grid, P.grid_bytes + P.skip_bytes);
  if (CB == 1) {
    const float2 none = make_float2(0.f, 0.f);
    for (int32_t c0 = 0; c0 < n_pts; c0 += 64 * SEG_CHUNKS) {
      uint32_t E[8][2], O[8][2];
#pragma unroll
      for (int y = 0; y < 8; y++) E[y][0] = E[y][1] = O[y][0] = O[y][1] = 0u;
      const int32_t c1 = min(n_pts, c0 + 64 * SEG_CHUNKS);
      // EVAL_CHUNKS 64-point chunks per iteration, their row loads issued together, and the points of the next
      // iteration fetched before this one's rows are consumed (the dependent chain is point -> window origin -> rows).
      float qx[EVAL_CHUNKS], qy[EVAL_CHUNKS];  // (plain floats: arrays of float2 end up in scratch)
#pragma unroll
      for (int u = 0; u < EVAL_CHUNKS; u++) {
        const float2 q = c0 + 64 * u + lane < c1 ? pts[c0 + 64 * u + lane] : none;
        qx[u] = q.x;
        qy[u] = q.y;
      }
      for (int32_t c = c0; c < c1; c += 64 * EVAL_CHUNKS) {
        float nx[EVAL_CHUNKS], ny[EVAL_CHUNKS];
#pragma unroll
        for (int u = 0; u < EVAL_CHUNKS; u++) {
          const int32_t idx = c + 64 * (EVAL_CHUNKS + u) + lane;
          const float2 q = idx < c1 ? pts[idx] : none;
          nx[u] = q.x;
          ny[u] = q.y;
        }
        uint32_t g[EVAL_CHUNKS], sh[EVAL_CHUNKS], w[EVAL_CHUNKS][8][3];
#pragma unroll
        for (int u = 0; u < EVAL_CHUNKS; u++) {
          const int32_t idx = c + 64 * u + lane;
          patch_origin(P, idx < c1, make_float2(qx[u], qy[u]), cf, sf, cx, cy, Y, X, &g[u], &sh[u]);
#pragma unroll
          for (int y = 0; y < 8; y++) {
            const u32x3 r = __builtin_amdgcn_raw_buffer_load_b96(rsrc, (int)(g[u] + (uint32_t)(y * P.pitch)), 0, 0);
            w[u][y][0] = r.x; w[u][y][1] = r.y; w[u][y][2] = r.z;
          }
        }
#pragma unroll
        for (int u = 0; u < EVAL_CHUNKS; u++) {
#pragma unroll
          for (int y = 0; y < 8; y++) {
            const uint32_t n0 = __builtin_amdgcn_alignbit(w[u][y][1], w[u][y][0], sh[u]);
            const uint32_t n1 = __builtin_amdgcn_alignbit(w[u][y][2], w[u][y][1], sh[u]);
            E[y][0] += n0 & M8; O[y][0] += n0 >> 8;
            E[y][1] += n1 & M8; O[y][1] += n1 >> 8;
          }
          qx[u] = nx[u];
          qy[u] = ny[u];
        }
      }
      uint32_t R[32];  // R[4 y + d]: d = 0: dx 0, 2; 1: dx 1, 3; 2: dx 4, 6; 3: dx 5, 7
#pragma unroll
      for (int y = 0; y < 8; y++) {
        R[4 * y + 0] = E[y][0];
        R[4 * y + 1] = O[y][0] - ((E[y][0] >> 16) << 8);
        R[4 * y + 2] = E[y][1];
        R[4 * y + 3] = O[y][1] - ((E[y][1] >> 16) << 8);
      }
      rs_step<32, 1>(R, lane & 1);
      rs_step<16, 2>(R, lane & 2);
      rs_step<8, 4>(R, lane & 4);
      uint32_t V[8];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        V[2 * i] = R[i] & 0xffffu;
        V[2 * i + 1] = R[i] >> 16;
      }
      rs_step<8, 8>(V, lane & 8);
      rs_step<4, 16>(V, lane & 16);
      rs_step<2, 32>(V, lane & 32);
      total += V[0];
    }
    const int r = 8 * (2 * ((lane >> 5) & 1) + ((lane >> 4) & 1)) + 4 * ((lane >> 2) & 1) + 2 * ((lane >> 1) & 1) + (lane & 1);
    const int f = (lane >> 3) & 1, d = r & 3;
    dy = r >> 2;
    dx = 4 * (d >> 1) + (d & 1) + 2 * f;
  } else {
    uint32_t A[64];  // A[8 y + x]: 32-bit sums (n_pts * 65535 < 2^32 for n_pts <= 65536)
#pragma unroll
    for (int i = 0; i < 64; i++) A[i] = 0u;
    float2 qn = lane < n_pts ? pts[lane] : make_float2(0.f, 0.f);
    for (int32_t c = 0; c < n_pts; c += 64) {
      const float2 q = qn;
      if (c + 64 + lane < n_pts) qn = pts[c + 64 + lane];  // next chunk's point: in flight while this one's rows load
      uint32_t g = 0u, sh = 0u;
      if (c + lane < n_pts) {
        int32_t prow, pcol;
        window_origin(q, cf, sf, P, cx, cy, &prow, &pcol);
        const int32_t col = pcol + BNB_B * X;
        g = (uint32_t)((prow + BNB_B * Y) * P.pitch + ((2 * col) & ~3));
        sh = (uint32_t)(col & 1) * 16u;
      }
#pragma unroll
      for (int y = 0; y < 8; y++) {
        const u32x4 r4 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(g + (uint32_t)(y * P.pitch)), 0, 0);
        const uint32_t w[5] = {r4.x, r4.y, r4.z, r4.w,
                               __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)(g + (uint32_t)(y * P.pitch) + 16u), 0, 0)};
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const uint32_t nj = __builtin_amdgcn_alignbit(w[j + 1], w[j], sh);
          A[8 * y + 2 * j] += nj & 0xffffu;
          A[8 * y + 2 * j + 1] += nj >> 16;
        }
      }
    }
    rs_step<64, 1>(A, lane & 1);
    rs_step<32, 2>(A, lane & 2);
    rs_step<16, 4>(A, lane & 4);
    rs_step<8, 8>(A, lane & 8);
    rs_step<4, 16>(A, lane & 16);
    rs_step<2, 32>(A, lane & 32);
    total = A[0];  // lane l holds A[l]: bit s of the register index was selected by bit s of the lane
    dy = lane >> 3;
    dx = lane & 7;
  }
  const int32_t ix = BNB_B * X + dx, iy = BNB_B * Y + dy;
  unsigned long long key = 0ull;
  if (ix < P.nx && iy < P.ny) {
    const uint32_t lin = (uint32_t)((k * P.nx + ix) * P.ny + iy);
    key = ((unsigned long long)total << 32) | (0xffffffffu - lin);
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const unsigned long long o = shfl_xor_u64b(key, m);
    key = o > key ? o : key;
  }
  return key;
}

// ---- second level: bounds of the four 4 x 4 sub-blocks of block (Y, X) ------------------------------------
// Sub-block (sy, sx) of a point with window origin (r, c) reads stored rows [r + 8Y + 4sy, + 4) and columns
// [c + 8X + 4sx, + 4): inside the 7 x 7 cells of P4[(r >> 2) + 2Y + sy][(c >> 2) + 2X + sx].  The table holds the
// byte pair {P4[i][j], P4[i + 1][j]} at (i, 2j): the four entries of a block are four consecutive bytes, ONE 8-byte
// load per point from a 4-byte-aligned offset -- against eight 12-byte loads for the block's exact sums.
// Returns the bounds (already scaled to the cell width) of sub-block q = 2 sy + sx in out[q], the same in every lane.
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void sub_bounds(const BnbParams &P, __amdgpu_buffer_rsrc_t p4, const float2 *pts, int32_t n_pts,
                                           float cf, float sf, int32_t cx, int32_t cy, int32_t Y, int32_t X, int lane,
                                           uint32_t scale, uint32_t (&out)[4]) {
  const int32_t DP = P.pool4_pitch;
  uint32_t A[4] = {0u, 0u, 0u, 0u};  // 32-bit sums, one per sub-block
  constexpr int U = 4;               // chunks whose loads are in flight together
  for (int32_t c = 0; c < n_pts; c += 64 * U) {
    uint32_t a[U];
    u32x2 w[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int32_t idx = c + 64 * u + lane;
      a[u] = 0u;  // (no point: the table's first bytes lie in the zero border)
      if (idx < n_pts) {
        int32_t prow, pcol;
        window_origin(pts[idx], cf, sf, P, cx, cy, &prow, &pcol);
        a[u] = (uint32_t)(((prow >> 2) + 2 * Y) * DP + 2 * ((pcol >> 2) + 2 * X));
      }
      w[u] = __builtin_amdgcn_raw_buffer_load_b64(p4, (int)(a[u] & ~3u), 0, 0);
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const uint32_t n = idx_guard(c + 64 * u + lane < n_pts, __builtin_amdgcn_alignbit(w[u].y, w[u].x, (a[u] & 2u) * 8u));
      A[0] += n & 0xffu;          // (sy 0, sx 0)
      A[2] += (n >> 8) & 0xffu;   // (sy 1, sx 0)
      A[1] += (n >> 16) & 0xffu;  // (sy 0, sx 1)
      A[3] += n >> 24;            // (sy 1, sx 1)
    }
  }
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) {
#pragma unroll
    for (int q = 0; q < 4; q++) A[q] += shfl_xor_u32(A[q], m);
  }
#pragma unroll
  for (int q = 0; q < 4; q++) out[q] = A[q] * scale;
}

// ---- exact sums of one 4 x 4 sub-block ----------------------------------------------------------------------
// As eval_block on rows [4 sy, 4 sy + 4) and columns [4 sx, 4 sx + 4) of block (Y, X): four loads per point.
template <int CB>
__device__ __forceinline__ unsigned long long eval_sub(const BnbParams &P, __amdgpu_buffer_rsrc_t rsrc, const float2 *pts,
                                                       int32_t n_pts, float cf, float sf, int32_t cx, int32_t cy,
                                                       int32_t k, int32_t Y, int32_t X, int32_t sy, int32_t sx, int lane) {
  uint32_t total = 0u;
  int dy, dx;
  const float2 none = make_float2(0.f, 0.f);
  constexpr int U = 4;  // chunks whose loads are in flight together
  if (CB == 1) {
    static_assert(SEG_CHUNKS % U == 0, "segments are whole iterations");
    for (int32_t c0 = 0; c0 < n_pts; c0 += 64 * SEG_CHUNKS) {
      uint32_t E[4], O[4];
#pragma unroll
      for (int y = 0; y < 4; y++) E[y] = O[y] = 0u;
      const int32_t c1 = min(n_pts, c0 + 64 * SEG_CHUNKS);
      for (int32_t c = c0; c < c1; c += 64 * U) {
        uint32_t sh[U];
        u32x2 w[U][4];
#pragma unroll
        for (int u = 0; u < U; u++) {
          const int32_t idx = c + 64 * u + lane;
          uint32_t g;
          patch_origin(P, idx < c1, idx < c1 ? pts[idx] : none, cf, sf, cx, cy, Y, X, &g, &sh[u]);
          if (idx < c1) g += (uint32_t)(BNB_B4 * sy * P.pitch + BNB_B4 * sx);
#pragma unroll
          for (int y = 0; y < 4; y++) w[u][y] = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)(g + (uint32_t)(y * P.pitch)), 0, 0);
        }
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
          for (int y = 0; y < 4; y++) {
            const uint32_t n = __builtin_amdgcn_alignbit(w[u][y].y, w[u][y].x, sh[u]);
            E[y] += n & M8;
            O[y] += n >> 8;
          }
      }
      uint32_t R[8];  // R[2 y + d]: d = 0: dx 0, 2; d = 1: dx 1, 3
#pragma unroll
      for (int y = 0; y < 4; y++) {
        R[2 * y] = E[y];
        R[2 * y + 1] = O[y] - ((E[y] >> 16) << 8);
      }
      rs_step<8, 1>(R, lane & 1);
      rs_step<4, 2>(R, lane & 2);
      rs_step<2, 4>(R, lane & 4);
      uint32_t V[2] = {R[0] & 0xffffu, R[0] >> 16};
      rs_step<2, 8>(V, lane & 8);
      V[0] = add_xor<16>(V[0]);
      V[0] = add_xor<32>(V[0]);
      total += V[0];
    }
    dy = ((lane >> 1) & 1) + 2 * ((lane >> 2) & 1);
    dx = (lane & 1) + 2 * ((lane >> 3) & 1);
  } else {
    uint32_t A[16];  // A[4 y + x]
#pragma unroll
    for (int i = 0; i < 16; i++) A[i] = 0u;
    for (int32_t c = 0; c < n_pts; c += 64 * U) {
      uint32_t sh[U];
      u32x3 w[U][4];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int32_t idx = c + 64 * u + lane;
        uint32_t g = 0u;
        sh[u] = 0u;
        if (idx < n_pts) {
          int32_t prow, pcol;
          window_origin(pts[idx], cf, sf, P, cx, cy, &prow, &pcol);
          const int32_t col = pcol + BNB_B * X + BNB_B4 * sx;
          g = (uint32_t)((prow + BNB_B * Y + BNB_B4 * sy) * P.pitch + ((2 * col) & ~3));
          sh[u] = (uint32_t)(col & 1) * 16u;
        }
#pragma unroll
        for (int y = 0; y < 4; y++) w[u][y] = __builtin_amdgcn_raw_buffer_load_b96(rsrc, (int)(g + (uint32_t)(y * P.pitch)), 0, 0);
      }
#pragma unroll
      for (int u = 0; u < U; u++)
#pragma unroll
        for (int y = 0; y < 4; y++) {
          const uint32_t n0 = __builtin_amdgcn_alignbit(w[u][y].y, w[u][y].x, sh[u]);
          const uint32_t n1 = __builtin_amdgcn_alignbit(w[u][y].z, w[u][y].y, sh[u]);
          A[4 * y + 0] += n0 & 0xffffu;
          A[4 * y + 1] += n0 >> 16;
          A[4 * y + 2] += n1 & 0xffffu;
          A[4 * y + 3] += n1 >> 16;
        }
    }
    rs_step<16, 1>(A, lane & 1);
    rs_step<8, 2>(A, lane & 2);
    rs_step<4, 4>(A, lane & 4);
    rs_step<2, 8>(A, lane & 8);
    A[0] = add_xor<16>(A[0]);
    A[0] = add_xor<32>(A[0]);
    total = A[0];  // lane l holds A[l & 15]
    dy = (lane >> 2) & 3;
    dx = lane & 3;
  }
  const int32_t ix = BNB_B * X + BNB_B4 * sx + dx, iy = BNB_B * Y + BNB_B4 * sy + dy;
  unsigned long long key = 0ull;
  if (ix < P.nx && iy < P.ny) {
    const uint32_t lin = (uint32_t)((k * P.nx + ix) * P.ny + iy);
    key = ((unsigned long long)total << 32) | (0xffffffffu - lin);
  }
#pragma unroll
  for (int m = 8; m >= 1; m >>= 1) {  // (lanes 16.. hold copies)
    const unsigned long long o = shfl_xor_u64b(key, m);
    key = o > key ? o : key;
  }
  return key;
}

// ==== the same three passes with the window origins of one rotation kept by the wave ========================
// All candidates of rotation k share the 1081 window origins; computing them (a point load, two double-precision
// floor quotients) per candidate made every pass a chain of dependent latencies.  A wave that owns rotation k keeps
// them packed (row << 16 | column; both < 65536) in LDS -- OCL chunks of 64, scans of up to 64 * OCL points, in the
// space of the pooled table, which the workgroup no longer needs once its bounds are done -- and a pass becomes:
// all loads of six to nine chunks issued back to back, then the adds.  (Held in 18 registers they were spilled:
// the register allocator kept the array in scratch memory and the kernel wrote 7 GB of it per launch.)  Lanes
// without a point hold origin (0, 0): every patch of theirs lies in the zero border (8 * NB + 7 < pad) and pooled
// entries there are zero.

// Entry format: row << 19 | column << 6 | (points - 1): consecutive beams that fall into the SAME stored cell (a third
// of a 1081-beam scan's) have the same window origin and read the same bytes in every bound and every exact sum of
// the rotation; they are kept as one entry with their number, and every sum adds the entry's bytes that many times.
// 749 entries instead of 1081 points on the bench workload: 12 chunks of loads instead of 17 in everything that
// follows.  Rows and columns < 8192 (grids up to 8000 cells + border; larger ones take the general kernel).
constexpr int ORG_COL_SHIFT = 6, ORG_ROW_SHIFT = 19;
__device__ __forceinline__ uint32_t org_row(uint32_t o) { return o >> ORG_ROW_SHIFT; }
__device__ __forceinline__ uint32_t org_col(uint32_t o) { return (o >> ORG_COL_SHIFT) & (ORG_LIMIT - 1u); }
__device__ __forceinline__ uint32_t org_cnt(uint32_t o) { return (o & 63u) + 1u; }

// (entry of chunk c for this lane; `org` points at the lane's word of chunk 0.  Past the list: row 0, column 0, whose
//  cells lie in the zero border)
__device__ __forceinline__ uint32_t origin_of(const uint32_t *org, int c) { return c < OCL ? org[64 * c] : 0u; }

// Returns the number of 64-entry chunks of the list (wave-uniform).  The packed sums of the bounds and exact sums hold
// 16-bit fields that are added over 8 lanes before they are unpacked: the points of every aligned group of 8 lanes,
// over all chunks, must not exceed 257 (257 * 255 = 65,535).  One point per entry keeps that by construction (<= 18
// chunks); with merged entries the wave checks it and, if a group would pass the limit (hundreds of beams in a few
// cells), builds the list again with one point per entry.
// `merged`: measured on 10,000 pairs with row-major planes 7.45 -> 7.38 ms for 16-bit grids and 6.81 -> 6.93 ms for
// 8-bit ones (the multiply-adds that replace the adds cost what the loads saved); with the tiled planes both widths
// merge (8-bit: 6.31 -> 6.25 ms).
__device__ __forceinline__ int32_t cache_origins(const BnbParams &P, const float2 *pts, int32_t n_pts, float cf, float sf,
                                                 int32_t cx, int32_t cy, int lane, uint32_t *org, bool merged) {
  uint32_t *base = org - lane;  // the wave's list
  for (int merge = merged ? 1 : 0; merge >= 0; merge--) {
    // (rolled: the origin arithmetic holds a division on its rare path.  The points of the next D chunks are in flight
    //  while one chunk's origins are computed; that array rotates so that its indices stay static.)
    constexpr int D = 6;
    float px[D], py[D];
#pragma unroll
    for (int d = 0; d < D; d++) {
      const float2 q = 64 * d + lane < n_pts ? pts[64 * d + lane] : make_float2(0.f, 0.f);
      px[d] = q.x;
      py[d] = q.y;
    }
    uint32_t tail = 0u;  // entries written (wave-uniform)
#pragma unroll 1
    for (int c = 0; c < OCL; c++) {
      const int32_t idx = 64 * c + lane;
      if (64 * c >= n_pts) break;
      const float2 pt = make_float2(px[0], py[0]);
      const float2 qn = idx + 64 * D < n_pts ? pts[idx + 64 * D] : make_float2(0.f, 0.f);
#pragma unroll
      for (int d = 0; d < D - 1; d++) {
        px[d] = px[d + 1];
        py[d] = py[d + 1];
      }
      px[D - 1] = qn.x;
      py[D - 1] = qn.y;
      const bool live = idx < n_pts;
      uint32_t o = 0u;
      if (live) {
        int32_t prow, pcol;
        window_origin(pt, cf, sf, P, cx, cy, &prow, &pcol);
        o = ((uint32_t)prow << ORG_ROW_SHIFT) | ((uint32_t)pcol << ORG_COL_SHIFT);
      }
      // runs of equal origins inside the chunk: the predecessor's by a DPP shift across the wave
      const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp((int)o, (int)o, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
      const bool head = live && (lane == 0 || o != prev || merge == 0);
      const unsigned long long H = __ballot(head), L = __ballot(live);
      // run length = distance to the next head, or to the end of the chunk's live lanes
      const unsigned long long rest = ((H | ~L) >> lane) >> 1;  // (a dead lane ends the run as a head would)
      const uint32_t cnt = rest ? (uint32_t)__builtin_ctzll(rest) + 1u : (uint32_t)(64 - lane);
      if (head) {
        const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(H >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)H, 0u));
        base[tail + before] = o | (cnt - 1u);
      }
      tail += (uint32_t)__builtin_popcountll(H);
    }
    const int32_t nch = (int32_t)((tail + 63u) >> 6);
    // (the rest of the list reads as row 0, column 0: the sums below unroll over groups of chunks and may read past nch)
    for (uint32_t e = tail + (uint32_t)lane; e < (uint32_t)ORG_WAVE; e += 64u) base[e] = 0u;
    // (the wave reads only its own words back: LDS operations of one wave are performed in order)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (merge == 0) return nch;
    // the points of this lane's entries, summed over the aligned group of 8 lanes
    uint32_t w = 0u;
    for (int c = 0; c < nch; c++) {
      const uint32_t e = org[64 * c];
      w += 64u * (uint32_t)c + (uint32_t)lane < tail ? org_cnt(e) : 0u;
    }
    w += shfl_xor_u32(w, 1);
    w += shfl_xor_u32(w, 2);
    w += shfl_xor_u32(w, 4);
    if (__ballot(w > 257u) == 0ull) return nch;
    __builtin_amdgcn_wave_barrier();
  }
  return 0;  // (not reached)
}

// Sub-block bounds of a strip of up to three blocks (Y, X0), (Y, X0 + 1), (Y, X0 + 2): their twelve table bytes are
// consecutive, ONE 16-byte load per point.  out[4 t + q]: block X0 + t, sub-block q = 2 sy + sx.
// `len` blocks are wanted (wave-uniform): their 4 len bytes start at a 2-byte-aligned offset, so 8 / 12 / 16 bytes are
// loaded -- the vector-memory address unit's time goes with the dwords a lane loads, and the candidates are bound by it.
__device__ __forceinline__ void strip_bounds_c(const BnbParams &P, __amdgpu_buffer_rsrc_t p4, const uint32_t *org,
                                               int32_t nch, int32_t Y, int32_t X0, int len, uint32_t scale,
                                               uint32_t (&out)[12]) {
  const uint32_t DP = (uint32_t)P.pool4_pitch;
  const uint32_t off = (uint32_t)(2 * Y) * DP + (uint32_t)(4 * X0);
  uint32_t E[3] = {0u, 0u, 0u}, O[3] = {0u, 0u, 0u};  // 16-bit fields: 18 chunks * 255 * 8 lanes < 65536
  constexpr int ROUNDS = 2;
  constexpr int H = OC / ROUNDS;  // chunks whose loads are in flight together
#pragma unroll
  for (int h = 0; h < ROUNDS; h++) {
    if (H * h >= nch) continue;
    u32x4 w[H];
    uint32_t sh[H], cn[H];
    uint32_t aa[H];
#pragma unroll
    for (int j = 0; j < H; j++) {
      const uint32_t o = origin_of(org, H * h + j);
      // (lanes without a point: origin (0, 0), whose entries lie in the zero border)
      const uint32_t a = (org_row(o) >> 2) * DP + 2u * (org_col(o) >> 2) + off;
      sh[j] = (a & 2u) * 8u;
      aa[j] = a & ~3u;
      cn[j] = org_cnt(o);
    }
    if (len == 1) {
#pragma unroll
      for (int j = 0; j < H; j++) {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(p4, (int)aa[j], 0, 0);
        w[j].x = v.x; w[j].y = v.y; w[j].z = 0u; w[j].w = 0u;
      }
    } else if (len == 2) {
#pragma unroll
      for (int j = 0; j < H; j++) {
        const u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(p4, (int)aa[j], 0, 0);
        w[j].x = v.x; w[j].y = v.y; w[j].z = v.z; w[j].w = 0u;
      }
    } else {
#pragma unroll
      for (int j = 0; j < H; j++) w[j] = __builtin_amdgcn_raw_buffer_load_b128(p4, (int)aa[j], 0, 0);
    }
#pragma unroll
    for (int j = 0; j < H; j++) {
      const uint32_t n0 = __builtin_amdgcn_alignbit(w[j].y, w[j].x, sh[j]);
      const uint32_t n1 = __builtin_amdgcn_alignbit(w[j].z, w[j].y, sh[j]);
      const uint32_t n2 = __builtin_amdgcn_alignbit(w[j].w, w[j].z, sh[j]);
      // bytes of n_t: (sy 0, sx 0), (sy 1, sx 0), (sy 0, sx 1), (sy 1, sx 1) of block X0 + t
      E[0] += __umul24(n0 & M8, cn[j]); O[0] += __umul24((n0 >> 8) & M8, cn[j]);
      E[1] += __umul24(n1 & M8, cn[j]); O[1] += __umul24((n1 >> 8) & M8, cn[j]);
      E[2] += __umul24(n2 & M8, cn[j]); O[2] += __umul24((n2 >> 8) & M8, cn[j]);
    }
  }
#pragma unroll
  for (int m = 1; m < 8; m <<= 1) {
#pragma unroll
    for (int t = 0; t < 3; t++) {
      E[t] += shfl_xor_u32(E[t], m);
      O[t] += shfl_xor_u32(O[t], m);
    }
  }
#pragma unroll
  for (int t = 0; t < 3; t++) {
    out[4 * t + 0] = E[t] & 0xffffu;  // (0, 0)
    out[4 * t + 1] = E[t] >> 16;      // (0, 1)
    out[4 * t + 2] = O[t] & 0xffffu;  // (1, 0)
    out[4 * t + 3] = O[t] >> 16;      // (1, 1)
  }
#pragma unroll
  for (int m = 8; m < 64; m <<= 1) {
#pragma unroll
    for (int q = 0; q < 12; q++) out[q] += shfl_xor_u32(out[q], m);
  }
#pragma unroll
  for (int q = 0; q < 12; q++) out[q] *= scale;
}

// ---- the pair's running best
template <bool GLOBAL>
__device__ __forceinline__ uint32_t best_sum(unsigned long long *best) {
  unsigned long long b;
  if (GLOBAL) b = __hip_atomic_load(best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else b = *(volatile unsigned long long *)best;
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32));  // one value for the whole wave
}
// The pair's own workgroup keeps its best in LDS (GLOBAL = false: csm_bnb_kernel); the takers of handed-over
// rotations share it through keys[pair] (GLOBAL = true: csm_bnb_rot_kernel).  Between two looks at a best that
// lives in global memory (a device-scope atomic load: microseconds under load) a taker works with its copy, raised
// by its own finds; a stale copy only costs pruning, never the result.
template <bool GLOBAL>
__device__ __forceinline__ uint32_t best_sum_cached(unsigned long long *best, uint32_t copy) {
  return GLOBAL ? copy : best_sum<false>(best);
}

// Exact sums on the matcher's 8-BIT plane (the cells of 8-bit grids; the high bytes of 16-bit cells), for the rotation
// whose origins the wave holds.  The plane is tiled, two copies (nhip_common.h hi_tiled; `pitch` = tiles per tile row,
// `copy_bytes` = bytes of a copy): a row's bytes come from the copy in which they start in a tile's first half, so no
// read crosses a tile, and the rows of a point's window are 16 bytes apart inside a tile and (tiles per row - 1) * 128
// + 16 further at its end.
// 4 x 4 sub-block (sy, sx) of block (Y, X): four 8-byte row loads per point.  Returns the sum of this lane's pose
// (*dy, *dx inside the sub-block; lanes 16.. hold copies of lanes 0..15).
__device__ __forceinline__ uint32_t sub_sums8(__amdgpu_buffer_rsrc_t rsrc, uint32_t pitch, uint32_t copy_bytes, const uint32_t *org,
                                              int32_t nch, int32_t Y, int32_t X, int32_t sy, int32_t sx, int lane, int *dy, int *dx) {
  const uint32_t roff = (uint32_t)(BNB_B * Y + BNB_B4 * sy), coff = (uint32_t)(BNB_B * X + BNB_B4 * sx);
  const uint32_t wrap = pitch * HI_TILE_BYTES - HI_TILE_BYTES;  // from a tile's last row to the next tile's first
  // chunks per round: 4 U row loads in flight.  (6 -> 2 together with block_sums8's rows in two groups of four: the same
  // speed within the run-to-run spread -- profiles/r03_bnb_ab_scratch_free.log -- and the by-rotation kernels fit their 128
  // registers: no scratch memory at all, where each launch used to write 225-370 MB of spills for 160 KB of records.)
  constexpr int U = 2;
  static_assert(OC % U == 0, "whole rounds");
  // (18 chunks * 255 * 8 lanes < 65536: the packed fields hold a whole scan)
  uint32_t E[4] = {0u, 0u, 0u, 0u}, O[4] = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int r = 0; r < OC / U; r++) {
    if (U * r >= nch) continue;
    u32x2 w[U][4];
    uint32_t sh[U], cn[U];
#pragma unroll
    for (int j = 0; j < U; j++) {
      const uint32_t o = origin_of(org, U * r + j);
      cn[j] = org_cnt(o);
      const uint32_t row0 = org_row(o) + roff, col0 = org_col(o) + coff, col4 = col0 & ~3u;
      const uint32_t cp = (col4 >> 3) & 1u, q = row0 & 7u;
      const uint32_t v0 = hi_tiled(row0, col4, cp, pitch, copy_bytes);
      sh[j] = (col0 & 3u) * 8u;
#pragma unroll
      for (int y = 0; y < 4; y++)
        w[j][y] = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)(v0 + (q + (uint32_t)y >= 8u ? wrap : 0u) + 16u * (uint32_t)y), 0, 0);
    }
#pragma unroll
    for (int j = 0; j < U; j++)
#pragma unroll
      for (int y = 0; y < 4; y++) {
        const uint32_t n = __builtin_amdgcn_alignbit(w[j][y].y, w[j][y].x, sh[j]);
        E[y] += __umul24(n & M8, cn[j]);
        O[y] += __umul24(n >> 8, cn[j]);
      }
  }
  uint32_t R[8];  // R[2 y + d]: d = 0: dx 0, 2; d = 1: dx 1, 3
#pragma unroll
  for (int y = 0; y < 4; y++) {
    R[2 * y] = E[y];
    R[2 * y + 1] = O[y] - ((E[y] >> 16) << 8);
  }
  rs_step<8, 1>(R, lane & 1);
  rs_step<4, 2>(R, lane & 2);
  rs_step<2, 4>(R, lane & 4);
  uint32_t V[2] = {R[0] & 0xffffu, R[0] >> 16};
  rs_step<2, 8>(V, lane & 8);
  V[0] = add_xor<16>(V[0]);
  V[0] = add_xor<32>(V[0]);
  *dy = ((lane >> 1) & 1) + 2 * ((lane >> 2) & 1);
  *dx = (lane & 1) + 2 * ((lane >> 3) & 1);
  return V[0];
}

// whole 8 x 8 block (Y, X): eight 12-byte row loads per point; lane l holds the sum of pose (*dy, *dx) of the block
__device__ __forceinline__ uint32_t block_sums8(__amdgpu_buffer_rsrc_t rsrc, uint32_t pitch, uint32_t copy_bytes, const uint32_t *org,
                                                int32_t nch, int32_t Y, int32_t X, int lane, int *dy, int *dx) {
  const uint32_t roff = (uint32_t)(BNB_B * Y), coff = (uint32_t)(BNB_B * X);
  const uint32_t wrap = pitch * HI_TILE_BYTES - HI_TILE_BYTES;  // from a tile's last row to the next tile's first
  // (one chunk's eight row loads in flight: with two the 32 accumulators + 48 row registers spill, measured 5 % slower)
  constexpr int U = 1;
  uint32_t E[8][2], O[8][2];
#pragma unroll
  for (int y = 0; y < 8; y++) E[y][0] = E[y][1] = O[y][0] = O[y][1] = 0u;
#pragma unroll
  for (int r = 0; r < OC / U; r++) {
    if (U * r >= nch) continue;
    // rows in groups of ROWS per chunk: 8 -> all eight 12-byte loads of a chunk in flight (24 registers), 4 -> two
    // groups of four (12 registers: with the 32 accumulators the kernel then stays inside its 128 registers)
    constexpr int ROWS = 4;
    uint32_t gg[U], sh[U], cn[U], qq[U];
#pragma unroll
    for (int j = 0; j < U; j++) {
      const uint32_t o = origin_of(org, U * r + j);
      cn[j] = org_cnt(o);
      // (12 bytes from a 4-aligned column: the copy in which they start in a tile's first half)
      const uint32_t row0 = org_row(o) + roff, col0 = org_col(o) + coff, col4 = col0 & ~3u;
      qq[j] = row0 & 7u;
      gg[j] = hi_tiled(row0, col4, (col4 >> 3) & 1u, pitch, copy_bytes);
      sh[j] = (col0 & 3u) * 8u;
    }
#pragma unroll
    for (int y0 = 0; y0 < 8; y0 += ROWS) {
      u32x3 w[U][ROWS];
#pragma unroll
      for (int j = 0; j < U; j++)
#pragma unroll
        for (int y = 0; y < ROWS; y++)
          w[j][y] = __builtin_amdgcn_raw_buffer_load_b96(rsrc, (int)(gg[j] + (qq[j] + (uint32_t)(y0 + y) >= 8u ? wrap : 0u) + 16u * (uint32_t)(y0 + y)), 0, 0);
#pragma unroll
      for (int j = 0; j < U; j++)
#pragma unroll
        for (int y = 0; y < ROWS; y++) {
          const uint32_t n0 = __builtin_amdgcn_alignbit(w[j][y].y, w[j][y].x, sh[j]);
          const uint32_t n1 = __builtin_amdgcn_alignbit(w[j][y].z, w[j][y].y, sh[j]);
          E[y0 + y][0] += __umul24(n0 & M8, cn[j]); O[y0 + y][0] += __umul24(n0 >> 8, cn[j]);
          E[y0 + y][1] += __umul24(n1 & M8, cn[j]); O[y0 + y][1] += __umul24(n1 >> 8, cn[j]);
        }
    }
  }
  uint32_t R[32];  // R[4 y + d]: d = 0: dx 0, 2; 1: dx 1, 3; 2: dx 4, 6; 3: dx 5, 7
#pragma unroll
  for (int y = 0; y < 8; y++) {
    R[4 * y + 0] = E[y][0];
    R[4 * y + 1] = O[y][0] - ((E[y][0] >> 16) << 8);
    R[4 * y + 2] = E[y][1];
    R[4 * y + 3] = O[y][1] - ((E[y][1] >> 16) << 8);
  }
  rs_step<32, 1>(R, lane & 1);
  rs_step<16, 2>(R, lane & 2);
  rs_step<8, 4>(R, lane & 4);
  uint32_t V[8];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    V[2 * i] = R[i] & 0xffffu;
    V[2 * i + 1] = R[i] >> 16;
  }
  rs_step<8, 8>(V, lane & 8);
  rs_step<4, 16>(V, lane & 16);
  rs_step<2, 32>(V, lane & 32);
  const int r = 8 * (2 * ((lane >> 5) & 1) + ((lane >> 4) & 1)) + 4 * ((lane >> 2) & 1) + 2 * ((lane >> 1) & 1) + (lane & 1);
  const int f = (lane >> 3) & 1, d = r & 3;
  *dy = r >> 2;
  *dx = 4 * (d >> 1) + (d & 1) + 2 * f;
  return V[0];
}

// the best key (sum << 32 | ~linear index) over the lanes' poses (ix, iy) of rotation k, the same in every lane
__device__ __forceinline__ unsigned long long best_key(const BnbParams &P, int32_t k, int32_t ix, int32_t iy, uint32_t total,
                                                       int top) {
  unsigned long long key = 0ull;
  if (ix < P.nx && iy < P.ny) {
    const uint32_t lin = (uint32_t)((k * P.nx + ix) * P.ny + iy);
    key = ((unsigned long long)total << 32) | (0xffffffffu - lin);
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    if (m > top) continue;
    const unsigned long long o = shfl_xor_u64b(key, m);
    key = o > key ? o : key;
  }
  return key;
}

// ---- 16-bit cells: a pose's exact sum from the stored image -------------------------------------------------
// sum and max over the 64 lanes, no LDS: DPP inside a row of 16, permlane swaps across rows
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);   // quad_perm [1, 0, 3, 2]
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);   // quad_perm [2, 3, 0, 1]
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xF, 0xF, false);  // row_ror:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, false);  // row_ror:8
  v = add_xor<16>(v);
  return add_xor<32>(v);
}
__device__ __forceinline__ uint32_t wave_max(uint32_t v) {
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xF, 0xF, false));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, false));
  v = max(v, shfl_xor_u32(v, 16));
  return max(v, shfl_xor_u32(v, 32));
}

// sum over the scan's points of the 16-bit cell that pose (ix, iy) of the rotation reads: one 2-byte load per point
__device__ __forceinline__ uint32_t pose_sum16(const BnbParams &P, __amdgpu_buffer_rsrc_t rsrc16, const uint32_t *org,
                                               int32_t nch, int32_t ix, int32_t iy) {
  // (rsrc16: the tiled copy of the 16-bit image -- one cell per point, neighbours along a wall in the same lines)
  uint32_t acc = 0u;  // (17 chunks * 65535 fits)
#pragma unroll
  for (int c = 0; c < OCL; c++) {
    if (c >= nch) continue;
    const uint32_t o = origin_of(org, c);
    acc += org_cnt(o) * (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(
                            rsrc16, (int)t16_tiled(org_row(o) + (uint32_t)iy, org_col(o) + (uint32_t)ix, (uint32_t)P.t16_tpr), 0, 0);
  }
  return wave_sum(acc);
}

// Two stages for 16-bit cells.  `hsum` is the lane's pose sum over the plane of HIGH bytes (what sub_sums8 /
// block_sums8 return on that plane, at the cost of 8-bit cells); a cell is 256 * high + low with low <= 255, so
//     256 * hsum + 255 * points  >=  the pose's 16-bit sum.
// Only poses whose bound reaches the best sum found so far can hold the optimum or a tie with it; their exact sums
// are read from the 16-bit image, highest bound first (it raises the best fastest), until no pose of the block is
// left above it.  Near the optimum that is a handful of poses; elsewhere none.  Returns the number evaluated.
template <bool GLOBAL>
__device__ __forceinline__ uint32_t refine16(const BnbParams &P, __amdgpu_buffer_rsrc_t rsrc16, const uint32_t *org,
                                             int32_t nch, int32_t n_pts, int32_t k, int32_t ix, int32_t iy, uint32_t hsum,
                                             bool mine, int lane, unsigned long long *best, uint32_t &bcopy) {
  const bool valid = mine && ix < P.nx && iy < P.ny;
  uint32_t ub = valid ? 256u * hsum + 255u * (uint32_t)n_pts : 0u;  // (points <= 1088: no overflow)
  uint32_t n_eval = 0u;
  for (;;) {
    const uint32_t top = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_max(ub));
    if (top == 0u || top < best_sum_cached<GLOBAL>(best, bcopy)) break;
    const int j = (int)__builtin_ctzll(__ballot(ub == top));  // (top != 0: a lane holds it)
    const int32_t jx = __builtin_amdgcn_readlane(ix, j), jy = __builtin_amdgcn_readlane(iy, j);
    const uint32_t sum = pose_sum16(P, rsrc16, org, nch, jx, jy);
    const uint32_t lin = (uint32_t)((k * P.nx + jx) * P.ny + jy);
    const unsigned long long key = ((unsigned long long)sum << 32) | (0xffffffffu - lin);
    wave_atomic_max(best, key, lane);  // (generic address: LDS or global)
    if (GLOBAL) bcopy = max(bcopy, sum);
    if (lane == j) ub = 0u;
    n_eval++;
  }
  return n_eval;
}

__device__ __forceinline__ void rotation_k(const BnbParams &P, int32_t pair, int32_t k, float *cf, float *sf) {
  // R(theta0) * R(delta_k), composed in double with individually rounded ops (as csm_correlate_kernel)
  const double c0 = P.rot0_cs[2 * pair], s0 = P.rot0_cs[2 * pair + 1];
  const int32_t kd = k + (P.pair_kbase ? P.pair_kbase[pair] : 0);  // (the pair's rotation k is entry kbase + k of the table)
  const double cd = P.delta_cs[2 * kd], sd = P.delta_cs[2 * kd + 1];
  *cf = __double2float_rn(__dsub_rn(__dmul_rn(c0, cd), __dmul_rn(s0, sd)));
  *sf = __double2float_rn(__dadd_rn(__dmul_rn(s0, cd), __dmul_rn(c0, sd)));
}

// ---- one candidate block: refine through the sub-block bounds, or evaluate whole --------------------------
// `best` is the pair's running best key (LDS of the pair's workgroup, or keys[pair] in global memory for the
// rotations handed to the second kernel).  A sub-block is skipped only if its bound is below the best SUM found so far: it cannot hold
// the optimum nor a tie with it.  n[0] whole blocks evaluated, n[1] candidates refined, n[2] sub-blocks evaluated.
struct PairCtx {
  const uint8_t *grid;
  const float2 *pts;
  int32_t n_pts, cx, cy, pair;
};

template <int CB>
__device__ __forceinline__ void process_candidate(const BnbParams &P, const PairCtx &C, int32_t k, int32_t v, int lane,
                                                  unsigned long long *best, uint32_t (&n)[4]) {
  const int Y = v / NB, X = v - NB * Y;  // v: block index NB * Y + X
  float cf, sf;
  rotation_k(P, C.pair, k, &cf, &sf);
  if (P.levels >= 2) {
    const __amdgpu_buffer_rsrc_t p4 = uniform_rsrc(C.grid + P.grid_bytes + P.skip_bytes + P.pool_bytes, P.pool4_bytes);
    uint32_t sb[4];
    sub_bounds(P, p4, C.pts, C.n_pts, cf, sf, C.cx, C.cy, Y, X, lane, CB == 1 ? 1u : 257u, sb);
    n[1]++;
    const uint32_t bsum = best_sum<false>(best);
    int alive = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) alive += (sb[q] != 0u && sb[q] >= bsum) ? 1 : 0;
    if (alive == 0) return;
    if (alive <= 2) {
      const __amdgpu_buffer_rsrc_t rsrc = uniform_rsrc(C.grid, P.grid_bytes + P.skip_bytes);
#pragma unroll 1
      for (int q = 0; q < 4; q++) {
        const uint32_t b = q == 0 ? sb[0] : (q == 1 ? sb[1] : (q == 2 ? sb[2] : sb[3]));  // (no indexed array: scratch)
        if (b == 0u || b < best_sum<false>(best)) continue;
        const unsigned long long key = eval_sub<CB>(P, rsrc, C.pts, C.n_pts, cf, sf, C.cx, C.cy, k, Y, X, q >> 1, q & 1, lane);
        wave_atomic_max(best, key, lane);
        n[2]++;
      }
      return;
    }
  }
  const unsigned long long key = eval_block<CB>(P, C.grid, C.pts, C.n_pts, cf, sf, C.cx, C.cy, k, Y, X, lane);
  wave_atomic_max(best, key, lane);
  n[0]++;
}

// ... with the rotation's origins held by the wave and the block's four sub-block bounds at hand.  `rsrc` is the
// 8-bit plane the exact block sums are taken on (8-bit grids: the image; 16-bit grids: the plane of high bytes) and
// `pitch8` its pitch; `rsrc16` the 16-bit image (CB == 2).
template <int CB, bool GLOBAL>
__device__ __forceinline__ void process_candidate_c(const BnbParams &P, __amdgpu_buffer_rsrc_t rsrc, uint32_t pitch8,  // (CB 2: tiles per row)
                                                    __amdgpu_buffer_rsrc_t rsrc16, const uint32_t *org, int32_t nch,
                                                    int32_t n_pts, int32_t k, int32_t Y, int32_t X,
                                                    uint32_t sb0, uint32_t sb1, uint32_t sb2, uint32_t sb3, int lane,
                                                    unsigned long long *best, uint32_t &bcopy, uint32_t (&n)[4]) {
  const uint32_t bsum = best_sum_cached<GLOBAL>(best, bcopy);
  const int alive = (sb0 != 0u && sb0 >= bsum) + (sb1 != 0u && sb1 >= bsum) + (sb2 != 0u && sb2 >= bsum) +
                    (sb3 != 0u && sb3 >= bsum);
  if (alive == 0) return;
  // a block with this many live sub-blocks is evaluated whole: one pass of 8 row loads per point instead of up to four
  // passes of 4, all 64 poses in one reduction (row-major planes: 2 beat 3, 7.70 -> 7.45 ms per 10,000 pairs; tiled planes:
  // loads are cheaper and 3 beats 2, 6.37 -> 6.22 ms)
  constexpr int WHOLE_MIN = 3;
  if (alive >= WHOLE_MIN) {
    int dy, dx;
    const uint32_t total = block_sums8(rsrc, pitch8, (uint32_t)P.hi_copy_bytes, org, nch, Y, X, lane, &dy, &dx);
    const int32_t ix = BNB_B * X + dx, iy = BNB_B * Y + dy;
    if (CB == 1) {
      const unsigned long long key = best_key(P, k, ix, iy, total, 32);
      wave_atomic_max(best, key, lane);  // (generic address: LDS or global)
      if (GLOBAL) bcopy = max(bcopy, (uint32_t)(key >> 32));
    } else {
      n[3] += refine16<GLOBAL>(P, rsrc16, org, nch, n_pts, k, ix, iy, total, true, lane, best, bcopy);
    }
    n[0]++;
    return;
  }
#pragma unroll 1
  for (int q = 0; q < 4; q++) {
    const uint32_t b = q == 0 ? sb0 : (q == 1 ? sb1 : (q == 2 ? sb2 : sb3));
    if (b == 0u || b < best_sum_cached<GLOBAL>(best, bcopy)) continue;
    int dy, dx;
    const uint32_t total = sub_sums8(rsrc, pitch8, (uint32_t)P.hi_copy_bytes, org, nch, Y, X, q >> 1, q & 1, lane, &dy, &dx);
    const int32_t ix = BNB_B * X + BNB_B4 * (q & 1) + dx, iy = BNB_B * Y + BNB_B4 * (q >> 1) + dy;
    if (CB == 1) {
      const unsigned long long key = best_key(P, k, ix, iy, total, 8);  // (lanes 16.. hold copies)
      wave_atomic_max(best, key, lane);
      if (GLOBAL) bcopy = max(bcopy, (uint32_t)(key >> 32));
    } else {
      n[3] += refine16<GLOBAL>(P, rsrc16, org, nch, n_pts, k, ix, iy, total, lane < 16, lane, best, bcopy);
    }
    n[2]++;
  }
}

// ---- one rotation of one pair: its candidate blocks (masks m0 | m1 over block index b = NB * Y + X, bounds u0 / u1
// lane-wise) through sub-block bounds and exact sums, origins held in registers.  `done`: the workgroup's bound row
// of this rotation, where finished blocks are zeroed (seeds), or null.
struct PhaseClocks {
  long long org, strip, eval;
};

template <int CB, bool GLOBAL>
__device__ __forceinline__ void rotation_pass(const BnbParams &P, const PairCtx &C, int32_t k, uint32_t u0, uint32_t u1,
                                              unsigned long long m0, unsigned long long m1, int lane,
                                              unsigned long long *best, uint32_t *done, uint32_t *org,
                                              uint32_t (&n_work)[4], PhaseClocks &clk) {
  long long t_mark = 0;
  float cf, sf;
  rotation_k(P, C.pair, k, &cf, &sf);
  if (BNB_STATS(P)) t_mark = clock64();
  const int32_t nch = cache_origins(P, C.pts, C.n_pts, cf, sf, C.cx, C.cy, lane, org, true);
  if (BNB_STATS(P)) clk.org += clock64() - t_mark;
  // (stored image + skip map: every offset an evaluation can form lies inside; see nhip_api.hip make_layout)
  // (8-bit grids: the image, on which the exact sums run; 16-bit grids: the tiled copy of the image, for pose_sum16)
  const __amdgpu_buffer_rsrc_t rsrc16 = CB == 1 ? uniform_rsrc(C.grid, P.grid_bytes + P.skip_bytes)
                                                : uniform_rsrc(C.grid + P.hi_offset + 2 * P.hi_copy_bytes, P.t16_bytes);
  // the 8-bit plane of the exact block sums, tiled, two copies: the cells of 8-bit grids, the high bytes of 16-bit ones;
  // pitch8 = its tiles per tile row
  const __amdgpu_buffer_rsrc_t rsrc = uniform_rsrc(C.grid + P.hi_offset, 2 * P.hi_copy_bytes);
  const uint32_t pitch8 = (uint32_t)P.hi_tpr;
  const __amdgpu_buffer_rsrc_t p4 = uniform_rsrc(C.grid + P.grid_bytes + P.skip_bytes + P.pool_bytes, P.pool4_bytes);
  // candidates in block order b = NB * Y + X; neighbours in X (up to three) share one pass over the table
  while ((m0 | m1) != 0ull) {
    const int b0 = m0 ? (int)__builtin_ctzll(m0) : 64 + (int)__builtin_ctzll(m1);
    const int Y = b0 / NB, X0 = b0 - NB * Y;
    int len = 1;
    while (len < 3 && X0 + len < NB) {
      const int b = b0 + len;
      if (!(((b < 64 ? m0 >> b : m1 >> (b - 64)) & 1ull))) break;
      len++;
    }
    uint32_t sb[12];
    uint32_t bcopy = GLOBAL ? best_sum<true>(best) : 0u;  // (one look per strip at a best in global memory)
    if (P.levels >= 2) {
      if (BNB_STATS(P)) t_mark = clock64();
      strip_bounds_c(P, p4, org, nch, Y, X0, len, CB == 1 ? 1u : 257u, sb);
      if (BNB_STATS(P)) clk.strip += clock64() - t_mark;
      n_work[1] += (uint32_t)len;
    } else {
#pragma unroll
      for (int q = 0; q < 12; q++) sb[q] = 0xffffffffu;
    }
#pragma unroll 1
    for (int t = 0; t < len; t++) {
      const int b = b0 + t;
      if (b < 64) m0 &= ~(1ull << b);
      else m1 &= ~(1ull << (b - 64));
      const uint32_t ub = (uint32_t)__builtin_amdgcn_readlane((int)(b < 64 ? u0 : u1), b & 63);
      if (ub < best_sum_cached<GLOBAL>(best, bcopy)) continue;  // the best has risen meanwhile
      // (selects, not an indexed array: that would live in scratch)
      const uint32_t s0 = t == 0 ? sb[0] : (t == 1 ? sb[4] : sb[8]), s1 = t == 0 ? sb[1] : (t == 1 ? sb[5] : sb[9]);
      const uint32_t s2 = t == 0 ? sb[2] : (t == 1 ? sb[6] : sb[10]), s3 = t == 0 ? sb[3] : (t == 1 ? sb[7] : sb[11]);
      if (BNB_STATS(P)) t_mark = clock64();
      process_candidate_c<CB, GLOBAL>(P, rsrc, pitch8, rsrc16, org, nch, C.n_pts, k, Y, X0 + t, s0, s1, s2, s3, lane, best,
                                      bcopy, n_work);
      if (BNB_STATS(P)) clk.eval += clock64() - t_mark;
      if (done && wave_leader(lane)) done[b] = 0u;
    }
  }
}

// ---- hand-over entries
__device__ __forceinline__ void read_entry(const RotEntry *ent, int32_t *pair, int32_t *k, unsigned long long *m0,
                                           unsigned long long *m1) {
  const unsigned long long w0 = ent->w[0], w1 = ent->w[1], w2 = ent->w[2], w3 = ent->w[3];
  *pair = (int32_t)(w0 >> 24);
  *k = (int32_t)(w0 & 0xffffffull);
  *m0 = w1 | (w2 << 41);          // blocks 0..63
  *m1 = (w2 >> 23) | (w3 << 18);  // blocks 64..120
}

__device__ __forceinline__ void pair_context(const BnbParams &P, int32_t pair, PairCtx *C) {
  const int32_t src = P.pair_src[pair], slot = P.pair_slot[pair];
  const int32_t beg = P.offsets[src];
  C->grid = P.grids + (size_t)slot * P.slot_bytes;
  C->pts = P.xy + beg;
  C->n_pts = P.offsets[src + 1] - beg;
  C->cx = P.pair_origin ? P.pair_origin[2 * pair] : 0;
  C->cy = P.pair_origin ? P.pair_origin[2 * pair + 1] : 0;
  C->pair = pair;
}

// BY_ROT: the pairs whose scan fits the register-held origins (n_pts <= 64 * OC), rotation by rotation; the other
// instantiation takes the longer scans -- or, with P.general_all (NHIP_BNB_QUEUE=1), every pair.
// Both are launched; a workgroup whose pair belongs to the other one returns at once.
// SPLIT (by-rotation form only): the workgroup ends after the seeds and leaves its state in P.ps_* for
// csm_bnb_cand_kernel.
// (words of 8 bytes behind the rows of bounds: the candidate queue -- or, in the split form, which has no queue, the
//  run lists of its ten waves and the rotations' order)
constexpr int QSPACE_SPLIT = (SPLIT_WAVES * LIST_ENTRIES * 4 + MAX_ROT * 12 + 15) / 16 * 2;
template <int CB, bool POOL_LDS, bool BY_ROT, bool SPLIT = false>
__global__ __launch_bounds__(SPLIT ? 64 * SPLIT_WAVES : BNB_THREADS, SPLIT ? SPLIT_WAVES / 2 : 4) void csm_bnb_kernel(BnbParams P) {
  constexpr int WAVES = SPLIT ? SPLIT_WAVES : BNB_WAVES, THREADS = 64 * WAVES;
  constexpr int QSPACE = SPLIT ? QSPACE_SPLIT : QCAP;
  extern __shared__ __align__(16) uint8_t smem[];
  // first region: the pooled table (POOL_LDS) while the bounds are computed, then the waves' window origins
  uint8_t *s_pool = smem;
  uint32_t *s_org = reinterpret_cast<uint32_t *>(smem);
  uint32_t *s_U = reinterpret_cast<uint32_t *>(smem + P.lds_first);  // n_theta * 128
  unsigned long long *s_queue = reinterpret_cast<unsigned long long *>(s_U + (size_t)P.n_theta * 128);  // QSPACE
  unsigned long long *s_best = s_queue + QSPACE;
  uint32_t *s_cnt = reinterpret_cast<uint32_t *>(s_best + 1);
  uint32_t *s_qn = s_cnt + 1, *s_qhead = s_cnt + 2;
  unsigned long long *s_slow = s_best + 4;  // (stats: the slowest wave's time in the candidate phase)
  // phase 1 only: per wave a ring of LIST_ENTRIES run-length entries, in the queue's space
  uint32_t *s_list = reinterpret_cast<uint32_t *>(s_queue);
  static_assert(BNB_WAVES * LIST_ENTRIES * 4 <= QCAP * 4, "the run lists fit the first half of the queue");
  // by-rotation kernel: per rotation its highest bound (<< 32 | k), and the rotations in descending order of it,
  // behind the run lists (n_theta <= MAX_ROT): the second half of the queue's space
  unsigned long long *s_kmax = s_queue + (SPLIT ? WAVES * LIST_ENTRIES / 2 : QCAP / 2);
  uint32_t *s_order = reinterpret_cast<uint32_t *>(s_kmax + MAX_ROT);
  static_assert(SPLIT_WAVES * LIST_ENTRIES * 4 + MAX_ROT * 12 <= QSPACE_SPLIT * 8 && MAX_ROT * 12 <= QCAP * 4, "lists, maxima and order fit");

  // block -> pair: the pairs of one target are consecutive; keep them on one XCD (blocks b and b + 8 share one)
  const uint32_t bid = blockIdx.x;
  const int32_t pair = (int32_t)((bid & 7u) * (uint32_t)P.pairs_per_xcd + (bid >> 3));
  if ((int32_t)(bid >> 3) >= P.pairs_per_xcd || pair >= P.n_pairs) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

  int32_t src = P.pair_src[pair], slot = P.pair_slot[pair];
  // (ids from device memory: a pair whose scan or slot lies outside the caller's counts is an empty scan -- it scores
  //  nothing, like a scan without points -- and is reported through the device's status words; both instantiations see it)
  const bool ids_ok = pair_ids_ok(P.ids, src, slot, pair + P.pair_base, threadIdx.x == 0);
  if (!ids_ok) src = slot = 0;
  const int32_t beg = ids_ok ? P.offsets[src] : 0, n_pts = ids_ok ? P.offsets[src + 1] - beg : 0;
  const float2 *pts = P.xy + beg;
  const uint8_t *grid = P.grids + (size_t)slot * P.slot_bytes;
  const int32_t cx = P.pair_origin ? P.pair_origin[2 * pair] : 0;
  const int32_t cy = P.pair_origin ? P.pair_origin[2 * pair + 1] : 0;
  const bool centre_ok = (abs(cx) + P.hx <= P.max_shift) && (abs(cy) + P.hy <= P.max_shift);
  if (BY_ROT != (n_pts <= 64 * OCL && (uint32_t)P.rows < ORG_LIMIT && !P.general_all)) return;  // the other instantiation's pair

  // pose 0 with sum 0 is a lower bound of the optimum (sums are >= 0; if all are 0, pose 0 is the answer)
  const unsigned long long key0 = 0xffffffffull;
  // (work counters: the instrumented build only.  Round 3 kept them in the general instantiation of the product build
  //  because it hung without them -- see wave_leader() for the cause, which was in the source, not in the counters.)
  unsigned long long *const stats_g = NHIP_BNB_INSTR ? P.stats : nullptr;
  const long long t_start = BNB_STATS(P) ? clock64() : 0;
  if (BNB_TIMELINE(P) && threadIdx.x == 0 && pair < BNB_STATS_PAIRS) BNB_TIMELINE(P)[4 * pair] = wall_clock64();
  if (threadIdx.x == 0) {
    *s_slow = 0ull;
    *s_best = key0;
    s_cnt[0] = s_cnt[3] = s_cnt[4] = s_cnt[5] = 0u;
    *s_qn = 0u;
    *s_qhead = 0u;
  }
  if (!centre_ok || n_pts <= 0) {  // (a centre the stored border cannot cover scores nothing)
    if (threadIdx.x == 0) P.keys[pair] = key0;
    return;
  }
  if (POOL_LDS) {  // the target's pooled table -> LDS
    const uint4 *gp = reinterpret_cast<const uint4 *>(grid + P.grid_bytes + P.skip_bytes);
    uint4 *sp = reinterpret_cast<uint4 *>(s_pool);
    // (four loads in flight per thread: 35 KB are 4.4 rounds of 512 threads, and a round trip each was 10 % of the pair)
    const int32_t n16 = (int32_t)(P.pool_bytes / 16);
    for (int32_t i = threadIdx.x; i < n16; i += 4 * THREADS) {
      uint4 v[4];
#pragma unroll
      for (int j = 0; j < 4; j++) v[j] = gp[min(i + j * THREADS, n16 - 1)];
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (i + j * THREADS < n16) sp[i + j * THREADS] = v[j];
    }
  }
  const __amdgpu_buffer_rsrc_t prs = uniform_rsrc(grid + P.grid_bytes + P.skip_bytes, P.pool_bytes);
  __syncthreads();

  // (1) bounds of every block of every rotation this wave owns; the wave's own best bound
  const uint32_t scale = CB == 1 ? 1u : 257u;
  unsigned long long wbest = 0ull;  // (U << 32) | (k << 8 | slot)
  for (int32_t k = wave; k < P.n_theta; k += WAVES) {
    float cf, sf;
    rotation_k(P, pair, k, &cf, &sf);
    uint32_t umax = 0u;
    uint32_t tot[2];
    coarse_rotation<POOL_LDS>(P, s_pool, prs, pts, n_pts, cf, sf, cx, cy, lane, s_list + wave * LIST_ENTRIES, tot);
    if (lane < 128 - NB * NB) s_U[k * 128 + NB * NB + lane] = 0u;  // (the row's unused tail)
#pragma unroll
    for (int i = 0; i < 2; i++) {
      // the bounds go to LDS in block order: entry b = NB * Y + X, so that neighbours in X are neighbours in b
      int Y, X;
      if (!slot_block(lane + 64 * i, &Y, &X)) continue;
      const uint32_t u = (Y < P.nby && X < P.nbx) ? tot[i] * scale : 0u;
      const int b = NB * Y + X;
      s_U[k * 128 + b] = u;
      umax = u > umax ? u : umax;
      const unsigned long long cand = ((unsigned long long)u << 32) | (uint32_t)((k << 8) | b);
      wbest = cand > wbest ? cand : wbest;
    }
    if (BY_ROT) {
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) {
        const uint32_t o = shfl_xor_u32(umax, m);
        umax = o > umax ? o : umax;
      }
      if (wave_leader(lane)) s_kmax[k] = ((unsigned long long)umax << 32) | (uint32_t)k;
    }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const unsigned long long o = shfl_xor_u64b(wbest, m);
    wbest = o > wbest ? o : wbest;
  }
  // (2) seed: the wave's highest-bound block, evaluated exactly
  uint32_t n_work[4] = {0u, 0u, 0u, 0u};
  PairCtx C;
  C.grid = grid;
  C.pts = pts;
  C.n_pts = n_pts;
  C.cx = cx;
  C.cy = cy;
  C.pair = pair;
  // Scans of up to 64 * OCL points: a wave owns a rotation at a time and keeps its window origins in LDS.
  // Longer scans take the general path below.
  if (BY_ROT) {
    // (the pooled table's space becomes the origins' once every wave is done with its bounds)
    // Seeds: every wave evaluates its highest-bound block.  (Fewer seeds -- only the waves with the highest bounds -- were
    // measured: 6 -> 8.1 ms, 4 -> 8.3, 1 -> 9.3 against 8.0 per 10,000 pairs; profiles/r03_matcher_experiments.txt.)
    __syncthreads();
    uint32_t *org = s_org + wave * ORG_WAVE + lane;
    // (NHIP_BNB_STATS=1: shader-clock sums -- wave time in phase 3 by part, and the workgroup's wall time)
    PhaseClocks clk = {0, 0, 0};
    long long t_busy = 0, t_wall = 0;
    const long long t_phase1 = BNB_STATS(P) ? clock64() : 0;
    if (BNB_TIMELINE(P) && threadIdx.x == 0 && pair < BNB_STATS_PAIRS) BNB_TIMELINE(P)[4 * pair + 1] = wall_clock64();
    const uint32_t xcd = bid & 7u;
    bool handed_over = false;
    // One loop, one copy of the candidate code (it is large; three inlined copies did not fit the instruction
    // cache and halved the speed of everything).  A wave's work items, in this order:
    //  SEED  the wave's own highest-bound block, alone: the seeds give `best` a good lower bound before anything is
    //        pruned against it; then the workgroup orders its rotations by their highest bound and counts the
    //        candidates the seeds have left;
    //  OWN   rotations of the pair handed out one at a time, best first: every block of the rotation whose bound
    //        reaches the best sum found so far goes through its sub-block bounds and, where those hold, exact sums.
    //        A pair with a flat landscape (>= heavy_min candidates; see launch_csm_bnb) works only its first
    //        keep_ranks rotations here and hands the others -- the rotation and the mask of its candidate blocks --
    //        to the list of this XCD for the second kernel.
    enum { SEED, OWN };
    int state = SEED;
    bool heavy = false;  // (the pair hands its rotations over)
    for (;;) {
      bool have = false;
      int32_t k = 0;
      uint32_t u0 = 0u, u1 = 0u;
      unsigned long long m0 = 0ull, m1 = 0ull;
      uint32_t *done = nullptr;
      if (state == SEED) {
        if ((uint32_t)(wbest >> 32) != 0u) {
          const int32_t v = (int32_t)(wbest & 0xffu);
          const uint32_t ub = (uint32_t)(wbest >> 32);
          k = (int32_t)((uint32_t)wbest >> 8);
          u0 = v == lane ? ub : 0u;
          u1 = v == lane + 64 ? ub : 0u;
          m0 = v < 64 ? 1ull << v : 0ull;
          m1 = v < 64 ? 0ull : 1ull << (v - 64);
          done = s_U + k * 128;
          have = true;
        }
      } else if (state == OWN) {
        const int32_t rank = (int32_t)wave_fetch_add(s_qhead, 1u, lane);
        if (rank >= P.n_theta) break;
        k = (int32_t)s_order[rank];
        u0 = s_U[k * 128 + lane];
        u1 = s_U[k * 128 + 64 + lane];
        const uint32_t bsum = best_sum<false>(s_best);
        m0 = __ballot(u0 != 0u && u0 >= bsum);
        m1 = __ballot(u1 != 0u && u1 >= bsum && lane + 64 < NB * NB);
        if ((m0 | m1) == 0ull) continue;
        if (heavy && (uint32_t)rank >= P.keep_ranks) {
          // (a pair's rotations go round the eight lists, starting at its home XCD's: pairs with flat landscapes are
          //  consecutive pairs -- one XCD's list held them all and the other seven XCDs' waves found theirs empty)
          const uint32_t lx = (xcd + (uint32_t)rank) & 7u;
          const uint32_t e = wave_fetch_add(P.rot_count + 8 * lx, 1u, lane);
          if (e < P.rot_cap) {
            if (lane < 4) {
              const unsigned long long M41 = (1ull << 41) - 1ull;
              const unsigned long long word = lane == 0   ? ((unsigned long long)(uint32_t)pair << 24) | (uint32_t)k
                                              : lane == 1 ? (m0 & M41)
                                              : lane == 2 ? ((m0 >> 41) | (m1 << 23)) & M41
                                                          : (m1 >> 18);
              P.rot_list[(size_t)lx * P.rot_cap + e].w[lane] = word;
            }
            handed_over = true;
            continue;
          }  // (list full: the rotation stays here)
        }
        have = true;
      }
      if (have) rotation_pass<CB, false>(P, C, k, u0, u1, m0, m1, lane, s_best, done, org, n_work, clk);
      if (state == SEED) {
        __syncthreads();
        if (BNB_TIMELINE(P) && threadIdx.x == 0 && pair < BNB_STATS_PAIRS) BNB_TIMELINE(P)[4 * pair + 2] = wall_clock64();
        if (BNB_STATS(P)) {
          t_busy = clock64();
          t_wall = wall_clock64();
        }
        // best first: the rotations in descending order of their highest bound (rank by counting: n_theta^2 compares)
        for (int32_t kk = threadIdx.x; kk < P.n_theta; kk += THREADS) {
          const unsigned long long mine = s_kmax[kk];
          int32_t rank = 0;
          for (int32_t j = 0; j < P.n_theta; j++) rank += s_kmax[j] > mine ? 1 : 0;
          s_order[rank] = (uint32_t)kk;
        }
        // ... and how many candidates the seeds have left: a flat landscape leaves thousands (the median pair: ~30)
        if (P.rot_list || SPLIT) {
          const uint32_t bsum = best_sum<false>(s_best);
          uint32_t mine = 0u;
          for (int32_t kk = wave; kk < P.n_theta; kk += WAVES) {
            const uint32_t c0 = s_U[kk * 128 + lane], c1 = s_U[kk * 128 + 64 + lane];
            mine += (uint32_t)__builtin_popcountll(__ballot(c0 != 0u && c0 >= bsum)) +
                    (uint32_t)__builtin_popcountll(__ballot(c1 != 0u && c1 >= bsum && lane + 64 < NB * NB));
          }
          if (wave_leader(lane)) atomicAdd(s_qn, mine);
        }
        __syncthreads();
        if (SPLIT) {
          // the state for csm_bnb_cand_kernel: the rows of the rotations that still hold a candidate, best first
          const uint32_t bsum = best_sum<false>(s_best);
          uint32_t *rows = P.ps_rows + (size_t)pair * (size_t)P.n_theta * 128u;
          uint32_t live = 0u;
          for (int32_t r = wave; r < P.n_theta; r += WAVES) {
            const uint32_t kk = s_order[r];
            const uint32_t kmax = (uint32_t)(s_kmax[kk] >> 32);
            if (kmax == 0u || kmax < bsum) break;  // (descending: no later rank holds one either)
            live = (uint32_t)r + 1u;
            const uint32_t w1 = lane == 62 ? kmax : (lane == 63 ? kk : s_U[kk * 128 + 64 + lane]);
            rows[(size_t)r * 128u + lane] = s_U[kk * 128 + lane];
            rows[(size_t)r * 128u + 64 + lane] = w1;
          }
          if (live && wave_leader(lane)) atomicMax(s_qhead, live);  // (the hand-out counter is not used in this form)
          __syncthreads();
          if (threadIdx.x == 0) {
            P.ps_count[pair] = *s_qn;
            P.ps_live[pair] = *s_qhead;
            P.ps_next[pair] = 0u;
          }
          break;
        }
        heavy = P.rot_list && *s_qn >= P.heavy_min;
        state = OWN;
      }
    }
    if (BNB_STATS(P) && lane == 0) {
      const long long now = clock64();
      atomicAdd(&BNB_STATS(P)[4], (unsigned long long)(now - t_busy));       // wave time in phase 3 (until out of work)
      atomicAdd(&BNB_STATS(P)[11], (unsigned long long)(wall_clock64() - t_wall));  // the same in 100 MHz ticks
      atomicAdd(&BNB_STATS(P)[5], (unsigned long long)clk.org);
      atomicAdd(&BNB_STATS(P)[6], (unsigned long long)clk.strip);
      atomicAdd(&BNB_STATS(P)[7], (unsigned long long)clk.eval);
      atomicMax(s_slow, (unsigned long long)(now - t_busy));  // slowest wave
      if (wave == 0) {
        atomicAdd(&BNB_STATS(P)[9], (unsigned long long)(t_busy - t_phase1));  // seeds (wave 0's view)
        atomicAdd(&BNB_STATS(P)[10], (unsigned long long)(t_phase1 - t_start)); // bounds
        if (handed_over) atomicAdd(&BNB_STATS(P)[12], 1ull);
      }
    }
  } else {
  if ((uint32_t)(wbest >> 32) != 0u) {
      const int32_t k = (int32_t)((uint32_t)wbest >> 8), v = (int32_t)(wbest & 0xffu);
      const int Y = v / NB, X = v - NB * Y;
      float cf, sf;
      rotation_k(P, pair, k, &cf, &sf);
      const unsigned long long key = eval_block<CB>(P, grid, pts, n_pts, cf, sf, cx, cy, k, Y, X, lane);
      if (wave_leader(lane)) {
        atomicMax(s_best, key);
        s_U[k * 128 + v] = 0u;  // done
      }
      n_work[0]++;
    }
    __syncthreads();
    // (3) every block whose bound reaches the best sum found so far.  The survivors cluster in a few rotations,
    // i.e. in a few waves: they go through one queue per workgroup that all eight waves drain.
    for (int32_t k = wave; k < P.n_theta; k += BNB_WAVES) {
  #pragma unroll
      for (int i = 0; i < 2; i++) {
        const uint32_t u = s_U[k * 128 + lane + 64 * i];
        const uint32_t bsum = (uint32_t)(*(volatile unsigned long long *)s_best >> 32);
        bool cand = u != 0u && u >= bsum;
        const unsigned long long m = __ballot(cand);
        if (m == 0ull) continue;
        const uint32_t base = wave_fetch_add(s_qn, (uint32_t)__builtin_popcountll(m), lane);
        const uint32_t pos = base + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
        const unsigned long long entry = ((unsigned long long)u << 32) | (uint32_t)((k << 8) | (lane + 64 * i));
        if (cand && pos < (uint32_t)QCAP) s_queue[pos] = entry;
        unsigned long long over = __ballot(cand && pos >= (uint32_t)QCAP);  // queue full: this wave takes them itself
        while (over) {
          const int j = (int)__builtin_ctzll(over);
          over &= over - 1ull;
          const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)entry, j);
          const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(entry >> 32), j);
          if (hi >= best_sum<false>(s_best))  // (the best may have risen meanwhile)
            process_candidate<CB>(P, C, (int32_t)(lo >> 8), (int32_t)(lo & 0xffu), lane, s_best, n_work);
        }
      }
    }
    __syncthreads();
    {
      const uint32_t qn = min(*s_qn, (uint32_t)QCAP);
      for (;;) {
        const uint32_t i = wave_fetch_add(s_qhead, 1u, lane);
        if (i >= qn) break;
        const unsigned long long entry = s_queue[i];
        if ((uint32_t)(entry >> 32) >= best_sum<false>(s_best))
          process_candidate<CB>(P, C, (int32_t)((uint32_t)entry >> 8), (int32_t)(entry & 0xffu), lane, s_best, n_work);
      }
    }
}
  if (stats_g && lane == 0) {
    atomicAdd(&s_cnt[0], n_work[0]);
    atomicAdd(&s_cnt[3], n_work[1]);
    atomicAdd(&s_cnt[4], n_work[2]);
    atomicAdd(&s_cnt[5], n_work[3]);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    P.keys[pair] = *s_best;  // (a pair that handed rotations over: the second kernel raises it from here)
    if (BNB_TIMELINE(P) && pair < BNB_STATS_PAIRS) {
      uint32_t hw;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      uint32_t xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      // 44 bits of time (100 MHz: two days), then the CU the workgroup ran on: HW_ID[15:8] = cu, sh, se; XCC id
      BNB_TIMELINE(P)[4 * pair + 3] = (wall_clock64() & 0xfffffffffffull) | ((unsigned long long)((hw >> 8) & 0xffu) << 44) |
                                      ((unsigned long long)(xcc & 0xfu) << 52);
    }
    if (BNB_STATS(P) && BY_ROT) atomicAdd(&BNB_STATS(P)[8], *s_slow);  // sum over pairs of the slowest wave's phase 3
    if (stats_g) {
      atomicAdd(&stats_g[0], (unsigned long long)s_cnt[0]);
      atomicAdd(&stats_g[1], (unsigned long long)(P.n_theta * P.nbx * P.nby));
      atomicAdd(&stats_g[2], (unsigned long long)s_cnt[3]);
      atomicAdd(&stats_g[3], (unsigned long long)s_cnt[4]);
      atomicAdd(&stats_g[14], (unsigned long long)s_cnt[5]);  // poses of 16-bit grids evaluated exactly (refine16)
      if (pair < BNB_STATS_PAIRS) atomicAdd(&stats_g[BNB_STATS_HEAD + pair], 4ull * s_cnt[0] + s_cnt[4]);
    }
  }
}

// Second kernel: the rotations of the pairs with many candidates, one wave per (pair, rotation), every wave of the
// chip.  The pair's running best is keys[pair] (left by its workgroup after the seeds, raised by every wave that
// works on the pair); a block is skipped when that has passed its bound.
template <int CB>
__global__ __launch_bounds__(256, 4) void csm_bnb_rot_kernel(BnbParams P) {
  const int lane = threadIdx.x & 63;
  // workgroup b works on the list of XCD b & 7 (blocks b and b + 8 share an XCD)
  const uint32_t xcd = blockIdx.x & 7u;
  const uint32_t filled = P.rot_count[8 * xcd];
  const uint32_t count = filled < P.rot_cap ? filled : P.rot_cap;
  const RotEntry *list = P.rot_list + (size_t)xcd * P.rot_cap;
  uint32_t n_work[4] = {0u, 0u, 0u, 0u};
  __shared__ uint32_t s_org2[4 * ORG_WAVE];
  uint32_t *org = s_org2 + (threadIdx.x >> 6) * ORG_WAVE + lane;
  PhaseClocks clk = {0, 0, 0};
  const long long t0 = BNB_STATS(P) ? clock64() : 0;
  for (;;) {
    const uint32_t i = wave_fetch_add(P.rot_count + 8 * xcd + 1, 1u, lane);
    if (i >= count) break;
    int32_t pair, k;
    unsigned long long m0, m1;
    read_entry(list + i, &pair, &k, &m0, &m1);
    PairCtx C;
    pair_context(P, pair, &C);
    uint32_t n[4] = {0u, 0u, 0u, 0u};
    // (no block bounds here: 0xffffffff lets every candidate through to its sub-block bounds, which are checked
    //  against the best as it stands in keys[pair])
    rotation_pass<CB, true>(P, C, k, 0xffffffffu, 0xffffffffu, m0, m1, lane, &P.keys[pair], nullptr, org, n, clk);
    if (BNB_STATS(P) && lane == 0 && pair < BNB_STATS_PAIRS) atomicAdd(&BNB_STATS(P)[BNB_STATS_HEAD + pair], 4ull * n[0] + n[2]);
    n_work[0] += n[0];
    n_work[1] += n[1];
    n_work[2] += n[2];
    n_work[3] += n[3];
  }
  if (BNB_STATS(P) && lane == 0) {
    if (n_work[0]) atomicAdd(&BNB_STATS(P)[0], (unsigned long long)n_work[0]);
    if (n_work[1]) atomicAdd(&BNB_STATS(P)[2], (unsigned long long)n_work[1]);
    if (n_work[2]) atomicAdd(&BNB_STATS(P)[3], (unsigned long long)n_work[2]);
    if (n_work[3]) atomicAdd(&BNB_STATS(P)[14], (unsigned long long)n_work[3]);
    atomicAdd(&BNB_STATS(P)[13], (unsigned long long)(clock64() - t0));  // wave time in the second kernel
    atomicAdd(&BNB_STATS(P)[5], (unsigned long long)clk.org);
    atomicAdd(&BNB_STATS(P)[6], (unsigned long long)clk.strip);
    atomicAdd(&BNB_STATS(P)[7], (unsigned long long)clk.eval);
  }
}

// ---- the split form's second and third launch ----------------------------------------------------------------
// The candidates' launch: per XCD (the pairs of a target stay where its grid is L2-resident) the list of its workgroups'
// pairs, IN PAIR ORDER -- the workgroups of a target's pairs then run together and share its lines in the XCD's L2.  The
// pairs with the most candidates left get one more workgroup per P.split_min of them, up to P.split_max (they share the
// pair's rotations through ps_next and its best through keys[pair]), granted by candidate count from the heaviest down
// while the XCD's list has room for a whole bucket's.  (Ordering the list by that count -- longest first -- was measured
// on the 10,000-pair workload and lost: 8.02 ms against 7.78 in target order; the kernel is bound by its memory pipeline,
// the shared heavy pairs leave no tail to hide, and the order scatters a target's pairs in time, L2 misses 4 % -> 30 %;
// shared pairs first: 7.9 ms.  profiles/r03_matcher_experiments.txt, commit 5048845 holds the code.)
// One workgroup per XCD segment.
constexpr int SORT_THREADS = 1024;
constexpr int SORT_BUCKETS = 16 * 21;
__device__ __forceinline__ int cand_bucket(uint32_t c) {  // sixteenths of an octave of the count; heaviest = bucket 0
  if (c == 0u) return SORT_BUCKETS - 1;
  const int e = 31 - __builtin_clz(c);                  // floor(log2 c)
  const int f = e >= 4 ? (int)((c >> (e - 4)) & 15u) : (int)((c << (4 - e)) & 15u);
  const int b = 16 * e + f;                              // ascending in c
  return b >= SORT_BUCKETS - 1 ? 0 : SORT_BUCKETS - 2 - b;
}
__device__ __forceinline__ uint32_t cand_shares(const BnbParams &P, uint32_t c) {
  const uint32_t w = 1u + c / P.split_min;
  return w < P.split_max ? w : P.split_max;
}

__global__ __launch_bounds__(SORT_THREADS) void csm_bnb_order_kernel(BnbParams P) {
  __shared__ uint32_t s_extra[SORT_BUCKETS];
  __shared__ uint32_t s_wave[SORT_THREADS / 64];
  __shared__ uint32_t s_base, s_idle;
  __shared__ int s_grant;
  const int32_t xcd = blockIdx.x;
  const int32_t lo = xcd * P.pairs_per_xcd, hi = min(lo + P.pairs_per_xcd, P.n_pairs);
  int32_t *work = P.ps_work + (size_t)xcd * P.ps_work_stride;
  for (int i = threadIdx.x; i < SORT_BUCKETS; i += SORT_THREADS) s_extra[i] = 0u;
  for (int i = threadIdx.x; i < P.ps_work_stride; i += SORT_THREADS) work[i] = -1;
  if (threadIdx.x == 0) s_base = s_idle = 0u;
  __syncthreads();
  // additional workgroups are granted by candidate count, from the heaviest pairs down, while the list has room for a
  // whole bucket's
  for (int32_t p = lo + (int32_t)threadIdx.x; p < hi; p += SORT_THREADS) {
    const uint32_t c = P.ps_count[p];
    atomicAdd(&s_extra[cand_bucket(c)], cand_shares(P, c) - 1u);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t room = (uint32_t)P.ps_work_stride - (uint32_t)(hi > lo ? hi - lo : 0);
    int grant = 0;
    while (grant < SORT_BUCKETS && s_extra[grant] <= room) room -= s_extra[grant++];
    s_grant = grant;
  }
  __syncthreads();
  // the pairs with candidates, in pair order EXACTLY (a prefix sum, not atomics whose order scrambles windows of 1024 pairs)
  // -- in two passes when P.front_min is set: first the pairs with at least that many candidates left (a few per cent of
  // the list, a third of its work: a workgroup of theirs runs for 0.3 - 2 ms, and one that the in-order dispatch starts in
  // the launch's last half millisecond IS the launch's tail), then the others, whose workgroups take 0.03 - 0.2 ms
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int pass = P.front_min ? 0 : 1; pass < 2; pass++)
  for (int32_t p0 = lo; p0 < hi; p0 += SORT_THREADS) {
    const int32_t p = p0 + (int32_t)threadIdx.x;
    const uint32_t c = p < hi ? P.ps_count[p] : 0u;
    const bool mine = P.front_min == 0u || ((c >= P.front_min) == (pass == 0));
    const uint32_t w = (c != 0u && mine) ? (cand_bucket(c) < s_grant ? cand_shares(P, c) : 1u) : 0u;
    if (p < hi && mine) P.ps_nw[p] = c != 0u ? w : 1u;
    uint32_t v = w;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t u = (uint32_t)__shfl_up((int)v, off, 64);
      if (lane >= off) v += u;
    }
    if (lane == 63) s_wave[wv] = v;
    __syncthreads();
    uint32_t before = s_base;
    for (int q = 0; q < wv; q++) before += s_wave[q];
    const uint32_t at = before + v - w;
    for (uint32_t j = 0; j < w; j++) work[at + j] = p;
    __syncthreads();
    if (threadIdx.x == SORT_THREADS - 1) s_base = before + v;
    __syncthreads();
  }
  // (behind them the pairs with no candidate left: their workgroups return at once; any order)
  const uint32_t tail0 = s_base;
  for (int32_t p = lo + (int32_t)threadIdx.x; p < hi; p += SORT_THREADS)
    if (P.ps_count[p] == 0u) work[tail0 + atomicAdd(&s_idle, 1u)] = p;
}

// The work lists, spread form (the default): every pair ONCE in its home XCD's list, in pair order (the pairs of a target
// run together where its tables are L2-resident), and IN FRONT of those the additional workgroups of the pairs with many
// candidates left -- one more per P.split_min of them, up to P.split_max per pair -- dealt round-robin over ALL EIGHT
// lists.  Round 3 kept a pair's additional workgroups in its home XCD's list, behind a budget per XCD: the pairs with
// flat landscapes are consecutive pairs (the sources far from one target), so one XCD held them all, granted the
// heaviest bucket its shares and left the next bucket with ONE workgroup per pair -- on 3,000 pairs the candidates'
// launch ran 3.6 ms of which the last 2 ms were ten such pairs on one eighth of the chip (profiles/r04_small_lists.txt).
// The additional workgroups come first in the launch so that a heavy pair's work starts with the launch, not after it.
constexpr int ORDER_THREADS = 256;
__global__ __launch_bounds__(ORDER_THREADS) void csm_bnb_order_spread_kernel(BnbParams P) {
  const int32_t p = (int32_t)(blockIdx.x * ORDER_THREADS + threadIdx.x);
  if (p >= P.n_pairs) return;
  const int32_t extra_slots = P.ps_work_stride - P.pairs_per_xcd;  // per list
  const int32_t home = p / P.pairs_per_xcd;
  P.ps_work[(size_t)home * P.ps_work_stride + extra_slots + (p - home * P.pairs_per_xcd)] = p;
  const uint32_t want = cand_shares(P, P.ps_count[p]) - 1u;
  uint32_t got = 0u;
  if (want) {
    const uint32_t t0 = atomicAdd(P.ps_ticket, want);
    for (uint32_t j = 0; j < want; j++) {
      const uint32_t e = t0 + j;
      if ((int32_t)(e >> 3) >= extra_slots) break;  // (the lists are full: the pair works with what it got)
      // (a pair's additional workgroups start on the XCD after its home's and go round from there)
      P.ps_work[(size_t)((e + (uint32_t)home + 1u) & 7u) * P.ps_work_stride + (e >> 3)] = p;
      got++;
    }
  }
  P.ps_nw[p] = 1u + got;
}

// The candidates: the OWN loop of csm_bnb_kernel on the state its split form left.  Four waves per workgroup, each takes
// the pair's live rotations one at a time, best first.  A pair with one workgroup keeps its hand-out counter and its
// best in LDS; a shared pair uses ps_next[pair] and keys[pair] -- the code is the same, through generic pointers, with
// the look-once-per-strip discipline of a best that may live in global memory.
constexpr int CAND_WAVES = 4;  // waves per workgroup of the candidates' kernel
constexpr int CAND_THREADS = 64 * CAND_WAVES;
// waves per SIMD the kernel is compiled for (register budget 512 / that): 4 for 8-bit grids, 5 for 16-bit grids -- with
// their planes tiled the working set fits five (6.78 -> 6.43 ms; 4 / 6: profiles/r04_bounds_variants.txt)
template <int CB>
__global__ __launch_bounds__(CAND_THREADS, CB == 2 ? 5 : 4) void csm_bnb_cand_kernel(BnbParams P) {
  __shared__ uint32_t s_org2[CAND_WAVES * ORG_WAVE];
  __shared__ unsigned long long s_best2;
  __shared__ uint32_t s_next2;
  const int lane = threadIdx.x & 63;
  const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
  if ((int32_t)slot >= P.ps_work_stride) return;
  const int32_t pair = P.ps_work[(size_t)xcd * P.ps_work_stride + slot];
  if (pair < 0) return;
  const uint32_t live = P.ps_live[pair];
  if (live == 0u) return;  // (nothing left after the seeds, or a pair of the general kernel)
  const bool shared = P.ps_nw[pair] > 1u;
  // (NHIP_BNB_TIMELINE=1, instrumented build: first start and last end over the pair's workgroups, 100 MHz ticks; the
  //  pair's index here is its index in the ROUND -- tools look at lists of one round)
  if (BNB_TIMELINE(P) && threadIdx.x == 0 && pair < BNB_STATS_PAIRS)
    atomicMin(&BNB_TIMELINE(P)[4 * (size_t)BNB_STATS_PAIRS + 2 + pair], wall_clock64());
  if (threadIdx.x == 0) {
    s_best2 = P.keys[pair];
    s_next2 = 0u;
  }
  __syncthreads();
  unsigned long long *best = shared ? &P.keys[pair] : &s_best2;
  uint32_t *next = shared ? &P.ps_next[pair] : &s_next2;
  const uint32_t *rows = P.ps_rows + (size_t)pair * (size_t)P.n_theta * 128u;
  uint32_t *org = s_org2 + (threadIdx.x >> 6) * ORG_WAVE + lane;
  PairCtx C;
  pair_context(P, pair, &C);
  uint32_t n_work[4] = {0u, 0u, 0u, 0u};
  PhaseClocks clk = {0, 0, 0};
  const long long t0 = BNB_STATS(P) ? clock64() : 0;
  for (;;) {
    const uint32_t rank = wave_fetch_add(next, 1u, lane);
    if (rank >= live) break;
    const uint32_t u0 = rows[(size_t)rank * 128u + lane], u1 = rows[(size_t)rank * 128u + 64 + lane];
    const uint32_t kmax = (uint32_t)__builtin_amdgcn_readlane((int)u1, 62);
    const int32_t k = __builtin_amdgcn_readlane((int)u1, 63);
    const uint32_t bsum = best_sum<true>(best);
    if (kmax < bsum) break;  // (best first: the later ranks hold nothing either)
    const unsigned long long m0 = __ballot(u0 != 0u && u0 >= bsum);
    const unsigned long long m1 = __ballot(u1 != 0u && u1 >= bsum && lane + 64 < NB * NB);
    if ((m0 | m1) == 0ull) continue;
    rotation_pass<CB, true>(P, C, k, u0, u1, m0, m1, lane, best, nullptr, org, n_work, clk);
  }
  if (BNB_STATS(P) && lane == 0) {
    if (n_work[0]) atomicAdd(&BNB_STATS(P)[0], (unsigned long long)n_work[0]);
    if (n_work[1]) atomicAdd(&BNB_STATS(P)[2], (unsigned long long)n_work[1]);
    if (n_work[2]) atomicAdd(&BNB_STATS(P)[3], (unsigned long long)n_work[2]);
    if (n_work[3]) atomicAdd(&BNB_STATS(P)[14], (unsigned long long)n_work[3]);
    atomicAdd(&BNB_STATS(P)[13], (unsigned long long)(clock64() - t0));
    atomicAdd(&BNB_STATS(P)[5], (unsigned long long)clk.org);
    atomicAdd(&BNB_STATS(P)[6], (unsigned long long)clk.strip);
    atomicAdd(&BNB_STATS(P)[7], (unsigned long long)clk.eval);
    if (pair < BNB_STATS_PAIRS) atomicAdd(&BNB_STATS(P)[BNB_STATS_HEAD + pair], 4ull * n_work[0] + n_work[2]);
  }
  if (BNB_TIMELINE(P) && lane == 0 && pair < BNB_STATS_PAIRS)
    atomicMax(&BNB_TIMELINE(P)[5 * (size_t)BNB_STATS_PAIRS + 2 + pair], wall_clock64());
  if (!shared) {
    __syncthreads();
    if (threadIdx.x == 0) P.keys[pair] = s_best2;
  }
}

}  // namespace

// ---- the kernel launches of one batch (compiled in both builds)
namespace bnb {

namespace {
// hipFuncSetAttribute once per instantiation and LDS size reached (not per launch)
template <int CB, bool PL, bool BR, bool SP = false>
int launch_main(const BnbParams &P, size_t lds, int64_t blocks, hipStream_t s) {
  // (the split form's first kernel: its own workgroup size, and the queue's space replaced by the lists' and the order's)
  constexpr int THREADS = SP ? 64 * SPLIT_WAVES : BNB_THREADS;
  if (SP) lds = lds - (size_t)QCAP * 8 + (size_t)QSPACE_SPLIT * 8;
  // (the attribute belongs to the function object of the CURRENT device: one high-water mark per device, so that a host
  //  with one thread per device raises it on each of them)
  constexpr int MAX_DEV = 64;
  static std::atomic<size_t> lds_set[MAX_DEV];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) dev = -1;
  if (dev < 0 || lds > lds_set[dev].load(std::memory_order_relaxed)) {
    NHIP_TRY_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(csm_bnb_kernel<CB, PL, BR, SP>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (dev >= 0) lds_set[dev].store(lds, std::memory_order_relaxed);
  }
  hipLaunchKernelGGL((csm_bnb_kernel<CB, PL, BR, SP>), dim3((uint32_t)blocks), dim3(THREADS), lds, s, P);
  return NHIP_OK;
}

// The split form's first part on one batch: bounds + seeds of the by-rotation pairs, the general kernel for the scans
// that form does not take, the order of the candidates' launch ...
template <int CB, bool PL>
int launch_split_a(const BnbParams &P, size_t lds, int64_t blocks, hipStream_t s) {
  int rc = launch_main<CB, PL, true, true>(P, lds, blocks, s);
  if (rc) return rc;
  // (the general instantiation only has work when some scan does not fit the by-rotation form: a caller that knows its
  //  scan lengths says so -- NHIP_SEARCH_SHORT_SCANS -- and saves the launch of n_pairs workgroups that return at once)
  if (!P.short_scans && (rc = launch_main<CB, PL, false>(P, lds, blocks, s))) return rc;
  if (P.ps_ticket)
    hipLaunchKernelGGL(csm_bnb_order_spread_kernel, dim3((uint32_t)((P.n_pairs + ORDER_THREADS - 1) / ORDER_THREADS)), dim3(ORDER_THREADS), 0, s, P);
  else
    hipLaunchKernelGGL(csm_bnb_order_kernel, dim3(8), dim3(SORT_THREADS), 0, s, P);
  return NHIP_OK;
}

template <int CB, bool PL>
int launch_both(const BnbParams &P, size_t lds, int64_t blocks, hipStream_t s) {
  // both instantiations are launched; a workgroup whose pair belongs to the other one returns at once
  if (!P.general_all) {
    int rc = launch_main<CB, PL, true>(P, lds, blocks, s);
    if (rc) return rc;
    if (P.short_scans) return NHIP_OK;
  }
  return launch_main<CB, PL, false>(P, lds, blocks, s);
}
}  // namespace

#if NHIP_BNB_INSTR
int launch_bnb_kernels_instr(const BnbParams &P, int cb, bool pool_lds, size_t lds, int64_t blocks, bool second_kernel,
                             hipStream_t s) {
#else
int launch_bnb_kernels(const BnbParams &P, int cb, bool pool_lds, size_t lds, int64_t blocks, bool second_kernel,
                       hipStream_t s) {
#endif
  int rc;
  if (cb == 1 && pool_lds) rc = launch_both<1, true>(P, lds, blocks, s);
  else if (cb == 1) rc = launch_both<1, false>(P, lds, blocks, s);
  else if (pool_lds) rc = launch_both<2, true>(P, lds, blocks, s);
  else rc = launch_both<2, false>(P, lds, blocks, s);
  if (rc) return rc;
  if (second_kernel) {
    const uint32_t rot_blocks = 256 * 4;  // four workgroups of four waves per CU; the waves take entries off the lists
    if (cb == 1) hipLaunchKernelGGL(csm_bnb_rot_kernel<1>, dim3(rot_blocks), dim3(256), 0, s, P);
    else hipLaunchKernelGGL(csm_bnb_rot_kernel<2>, dim3(rot_blocks), dim3(256), 0, s, P);
  }
  return NHIP_OK;
}

#if NHIP_BNB_INSTR
int launch_bnb_split_a_instr(const BnbParams &P, int cb, bool pool_lds, size_t lds, int64_t blocks, hipStream_t s) {
#else
int launch_bnb_split_a(const BnbParams &P, int cb, bool pool_lds, size_t lds, int64_t blocks, hipStream_t s) {
#endif
  if (cb == 1 && pool_lds) return launch_split_a<1, true>(P, lds, blocks, s);
  if (cb == 1) return launch_split_a<1, false>(P, lds, blocks, s);
  if (pool_lds) return launch_split_a<2, true>(P, lds, blocks, s);
  return launch_split_a<2, false>(P, lds, blocks, s);
}

// ... and its second: the candidates (on any stream ordered behind the first part of the same batch)
#if NHIP_BNB_INSTR
int launch_bnb_split_b_instr(const BnbParams &P, int cb, hipStream_t s) {
#else
int launch_bnb_split_b(const BnbParams &P, int cb, hipStream_t s) {
#endif
  if (cb == 1) hipLaunchKernelGGL(csm_bnb_cand_kernel<1>, dim3((uint32_t)(8 * P.ps_work_stride)), dim3(CAND_THREADS), 0, s, P);
  else hipLaunchKernelGGL(csm_bnb_cand_kernel<2>, dim3((uint32_t)(8 * P.ps_work_stride)), dim3(CAND_THREADS), 0, s, P);
  return NHIP_OK;
}

}  // namespace bnb


}  // namespace nhip
