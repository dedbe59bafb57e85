// nhip_corr.hip -- K5: correspondence search on gfx950.
//
// Replaces, batched over all (source, target) residual blocks of a problem build,
//   Solver::GetPointToPointMatching          src/optimization/solver.cc:132-172
//   FindClosestPoint                         src/optimization/solver.cc:80-90
//   KDTree<float,2>::FindNearestPoint        src/util/kdtree.cc:253-305
// which the reference runs from scratch for every (i, j) block and every window size
// (solver.cc:321-356): three tree descents per matched point.
//
// Per block one workgroup: the target cloud is staged in LDS (8 B per point) IN THE ORDER OF ITS BUCKETS: a uniform
// grid with cells a shade wider than the outlier threshold, counting sort on 1024 buckets in LDS; the four cells
// (4g .. 4g + 3, cy) share a hashed group of four consecutive buckets, so a query's three cells of a row are one or
// two runs of consecutive points.  Every lane owns a contiguous range of source
// points, transforms them into the target frame with the float affine inverse(T_target) * T_source
// (Eigen Affine2f semantics, individually rounded products) and visits the 3 x 3 cells around
// each: a target that passes sqrt(d2) < outlier_threshold lies in one of them, and so does the
// global nearest neighbour whenever any target passes, so the kept rows equal an exhaustive
// scan's (exact nearest neighbour, ties to the lowest index, kept if within the threshold).
// Target clouds larger than the LDS stage, or coordinates so large that float cell indices are
// no longer exact, take the exhaustive scan (LDS broadcast reads).  Kept rows are written in
// source order (block-wide exclusive scan of the per-lane counts) as the 8-float rows K4 consumes.
// ~25 candidates per source point instead of 1081.  Where the time goes (tools/corr_phase_probe.py, 9,945 blocks of
// 1081-point scans, 0.58 ms): staging, sort and the rows' output 0.06 ms; the walk 0.52 ms, and it is its per-candidate
// iterations under divergent trip counts (a wave runs a bucket run as long as its fullest lane needs: ~40 % of the lanes
// work) -- not the point reads (0.02 ms), not the index indirection (points in bucket order: -3 %), not the per-run
// setup (runs of three cells instead of nine single cells: -1.5 %); one loop per lane over all nine cells, which would
// iterate max-over-lanes of the totals, was measured slower (0.68 ms: its cell-advance test runs every iteration).
#include "nhip_common.h"

namespace nhip {

namespace {

constexpr int CT = 256;
constexpr int TGT_CHUNK = 2048;  // target points staged per pass (16 KB)
constexpr int MAX_PER_LANE = 8;  // source points a lane owns per pass (2048 per pass)

struct Aff2f {
  float m00, m01, m10, m11, tx, ty;
};

__device__ __forceinline__ Aff2f pose_affine(const float *a) {  // cos sin x y
  return {a[0], -a[1], a[1], a[0], a[2], a[3]};
}

__device__ __forceinline__ Aff2f inverse_f(const Aff2f &A) {
  const float det = __fsub_rn(__fmul_rn(A.m00, A.m11), __fmul_rn(A.m10, A.m01));
  const float invdet = __fdiv_rn(1.0f, det);
  Aff2f R;
  R.m00 = __fmul_rn(A.m11, invdet);
  R.m10 = __fmul_rn(-A.m10, invdet);
  R.m01 = __fmul_rn(-A.m01, invdet);
  R.m11 = __fmul_rn(A.m00, invdet);
  R.tx = -__fadd_rn(__fmul_rn(R.m00, A.tx), __fmul_rn(R.m01, A.ty));
  R.ty = -__fadd_rn(__fmul_rn(R.m10, A.tx), __fmul_rn(R.m11, A.ty));
  return R;
}

__device__ __forceinline__ float dot2(float a, float b, float c, float d) {
  return __fadd_rn(__fmul_rn(a, b), __fmul_rn(c, d));
}

__device__ __forceinline__ Aff2f mul_f(const Aff2f &A, const Aff2f &B) {
  Aff2f C;
  C.m00 = dot2(A.m00, B.m00, A.m01, B.m10);
  C.m01 = dot2(A.m00, B.m01, A.m01, B.m11);
  C.m10 = dot2(A.m10, B.m00, A.m11, B.m10);
  C.m11 = dot2(A.m10, B.m01, A.m11, B.m11);
  C.tx = __fadd_rn(dot2(A.m00, B.tx, A.m01, B.ty), A.tx);
  C.ty = __fadd_rn(dot2(A.m10, B.tx, A.m11, B.ty), A.ty);
  return C;
}

#ifndef NHIP_CORR_NB
#define NHIP_CORR_NB 1024
#endif
constexpr int NB = NHIP_CORR_NB;  // hash buckets of the target grid

// Buckets come in groups of four: the cells (4g .. 4g + 3, cy) of a row occupy four CONSECUTIVE buckets (the group is
// hashed, the cell's place in it is cx & 3), so the three cells cx - 1 .. cx + 1 a query visits in a row are one run of
// consecutive buckets -- or two, when they straddle a group -- instead of three separate ones: 4.5 bucket ranges per
// point instead of 9, each with its hash, its two dependent LDS reads and its short loop (the walk is bound by that
// per-range overhead: tools/corr_phase_probe.py).  Cells that share a bucket only add candidates the distance test drops.
__device__ __forceinline__ uint32_t group_hash(int32_t gx, int32_t cy) {
  return ((((uint32_t)gx * 73856093u) ^ ((uint32_t)cy * 19349663u)) & (uint32_t)(NB / 4 - 1)) << 2;
}
__device__ __forceinline__ uint32_t cell_hash(int32_t cx, int32_t cy) {
  return group_hash(cx >> 2, cy) | ((uint32_t)cx & 3u);
}

// block-wide inclusive scan of one int per thread: wave scans by shuffles, wave totals through LDS
// (s_scan: at least CT / 64 ints), two barriers
__device__ __forceinline__ int32_t block_scan_incl(int32_t v, int32_t *s_scan, int tid) {
  const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int32_t u = __shfl_up(v, off, 64);
    if (lane >= off) v += u;
  }
  __syncthreads();  // s_scan may still be read by the previous user
  if (lane == 63) s_scan[wv] = v;
  __syncthreads();
#pragma unroll
  for (int w = 0; w < CT / 64; w++) v += (w < wv) ? s_scan[w] : 0;
  return v;
}

// GATE: the normal gate of Solver::GetPointToNormalMatching / FindClosestPointWithSimilarNormal
// (solver.cc:177-260): candidates are the targets within the threshold whose normal satisfies
// |n_target . n_source| > min_cos (NormalsSimilar, math_util.h:46-49; n_source stays in the source
// frame, as in the reference); the nearest candidate wins, ties to the lowest index.
template <bool GATE>
__global__ __launch_bounds__(CT) void corr_search_kernel(
    const float2 *__restrict__ xy, const float2 *__restrict__ normals,
    const int32_t *__restrict__ offsets, const int32_t *__restrict__ block_src,
    const int32_t *__restrict__ block_tgt, const float *__restrict__ pose_aff, float thr, float min_cos,
    const int64_t *__restrict__ cap_offsets, float4 *__restrict__ corr,
    int32_t *__restrict__ counts, int32_t n_scans, uint32_t *__restrict__ status) {
  __shared__ float2 s_tgt[TGT_CHUNK];
  __shared__ float2 s_tgn[GATE ? TGT_CHUNK : 1];  // target normals (gate only)
  __shared__ uint16_t s_sorted[TGT_CHUNK];  // target indices grouped by bucket
  __shared__ uint32_t s_start[NB + 1];      // bucket h = s_sorted[s_start[h] .. s_start[h + 1])
  __shared__ uint32_t s_cur[NB];
  __shared__ int32_t s_scan[CT];
  const int b = blockIdx.x, tid = threadIdx.x;
  int32_t s = block_src[b], t = block_tgt[b];
  // (scan ids from device memory: a block with one outside [0, n_scans) is reported and matches nothing -- count 0)
  const bool ids_ok = id_in(s, n_scans) && id_in(t, n_scans);
  if (!ids_ok) {
    if (tid == 0) flag_bad_id(status, BAD_SCAN_ID, id_in(s, n_scans) ? t : s, b);
    s = t = 0;
  }
  const int32_t sb = ids_ok ? offsets[s] : 0, ns = ids_ok ? offsets[s + 1] - sb : 0;
  const int32_t tb = ids_ok ? offsets[t] : 0, nt = ids_ok ? offsets[t + 1] - tb : 0;
  const Aff2f C = mul_f(inverse_f(pose_affine(pose_aff + 4 * (size_t)t)), pose_affine(pose_aff + 4 * (size_t)s));
  float4 *out = corr + 2 * (size_t)cap_offsets[b];
  int32_t written = 0;  // rows already emitted by earlier source passes (uniform)

  // Cells 0.1 % wider than the threshold: two points closer than thr are less than 0.999 cells apart per axis.  A
  // float cell coordinate fx = g * inv_cell carries a relative error of ~1.2e-7 (inv_cell and the product are each
  // rounded), so the computed coordinates of such a pair differ by less than 0.999 + 2.4e-7 |fx|: below 1 -- the
  // pair lands in the same or in adjacent cells and the 3 x 3 visit finds it -- as long as |fx| < 4166.
  // Larger coordinates (beyond 4096 cells = 1 km at the 0.25 m threshold) take the exhaustive scan.
  const float inv_cell = __fdiv_rn(1.0f, __fmul_rn(thr, 1.001f));
  constexpr float CELL_LIMIT = 4096.0f;
  bool hashed = nt <= TGT_CHUNK && thr > 0.f && inv_cell < 3.0e38f;
  if (hashed) {
    // stage + bucket the whole target cloud once
    int big = 0;
    for (int32_t i = tid; i < NB; i += CT) s_start[i] = 0u;
    __syncthreads();
    // (all of a thread's target points -- up to TGT_CHUNK / CT = 8 -- are requested before the first is used)
    constexpr int TPL = TGT_CHUNK / CT;
    {
      float2 tg[TPL];
#pragma unroll
      for (int k = 0; k < TPL; k++) {
        const int32_t i = tid + k * CT;
        if (i < nt) tg[k] = xy[tb + i];
      }
#pragma unroll
      for (int k = 0; k < TPL; k++) {
        const int32_t i = tid + k * CT;
        if (i >= nt) continue;
        const float fx = __fmul_rn(tg[k].x, inv_cell), fy = __fmul_rn(tg[k].y, inv_cell);
        if (!(fabsf(fx) < CELL_LIMIT) || !(fabsf(fy) < CELL_LIMIT)) big = 1;
        else atomicAdd(&s_start[cell_hash((int32_t)floorf(fx), (int32_t)floorf(fy))], 1u);
      }
    }
    hashed = !__syncthreads_or(big);
  }
  if (hashed) {
    // exclusive scan of the NB bucket counts (8 per thread), then scatter
    uint32_t c[NB / CT], tot = 0;
#pragma unroll
    for (int k = 0; k < NB / CT; k++) {
      c[k] = s_start[tid * (NB / CT) + k];
      tot += c[k];
    }
    uint32_t run = (uint32_t)block_scan_incl((int32_t)tot, s_scan, tid) - tot;
#pragma unroll
    for (int k = 0; k < NB / CT; k++) {
      s_start[tid * (NB / CT) + k] = run;
      s_cur[tid * (NB / CT) + k] = run;
      run += c[k];
    }
    if (tid == CT - 1) s_start[NB] = run;
    __syncthreads();
    // The points themselves go to LDS IN BUCKET ORDER (s_tgt[slot], s_tgn[slot]; s_sorted[slot] = the point's index):
    // the walk then reads a bucket's points at consecutive addresses, with no index to fetch and follow first.  (Read
    // again from global memory -- the L2 has them -- rather than held in registers across the scan: 16 more registers
    // would cost the kernel its fifth wave per SIMD.)
    constexpr int TPL = TGT_CHUNK / CT;
    float2 tg[TPL], tn[GATE ? TPL : 1];
#pragma unroll
    for (int k = 0; k < TPL; k++) {
      const int32_t i = tid + k * CT;
      if (i < nt) {
        tg[k] = xy[tb + i];
        if (GATE) tn[GATE ? k : 0] = normals[tb + i];
      }
    }
#pragma unroll
    for (int k = 0; k < TPL; k++) {
      const int32_t i = tid + k * CT;
      if (i >= nt) continue;
      const float2 g = tg[k];
      const uint32_t h = cell_hash((int32_t)floorf(__fmul_rn(g.x, inv_cell)), (int32_t)floorf(__fmul_rn(g.y, inv_cell)));
      const uint32_t slot = atomicAdd(&s_cur[h], 1u);
      s_tgt[slot] = g;
      if (GATE) s_tgn[slot] = tn[GATE ? k : 0];
      s_sorted[slot] = (uint16_t)i;
    }
    __syncthreads();
  }

  for (int32_t s0 = 0; s0 < ns; s0 += CT * MAX_PER_LANE) {
    const int32_t n_pass = min(ns - s0, CT * MAX_PER_LANE);
    // lane t owns points s0 + t, s0 + t + CT, ...: consecutive lanes hold consecutive points, so loads
    // and the kept rows of one round are contiguous across the wave
    const int32_t hi = s0 + n_pass;
#define SRC_INDEX(k) (s0 + (k) * CT + tid)
    float qx[MAX_PER_LANE], qy[MAX_PER_LANE], best[MAX_PER_LANE];
    float snx[GATE ? MAX_PER_LANE : 1], sny[GATE ? MAX_PER_LANE : 1];
    int32_t bi[MAX_PER_LANE];
    int big = 0;
#pragma unroll
    for (int k = 0; k < MAX_PER_LANE; k++) {
      bi[k] = -1;
      best[k] = 3.0e38f;
      qx[k] = qy[k] = 0.f;
      if (SRC_INDEX(k) < hi) {
        const float2 p = xy[sb + SRC_INDEX(k)];
        if (GATE) {
          const float2 sn = normals[sb + SRC_INDEX(k)];
          snx[k] = sn.x;
          sny[k] = sn.y;
        }
        qx[k] = __fadd_rn(dot2(C.m00, p.x, C.m01, p.y), C.tx);
        qy[k] = __fadd_rn(dot2(C.m10, p.x, C.m11, p.y), C.ty);
        // (a non-finite query matches nothing on either path; only a huge finite one needs the scan)
        if (fabsf(__fmul_rn(qx[k], inv_cell)) >= CELL_LIMIT || fabsf(__fmul_rn(qy[k], inv_cell)) >= CELL_LIMIT) big = 1;
      }
    }
    const bool scan_all = !hashed || __syncthreads_or(hashed ? big : 0);
    if (!scan_all) {
#pragma unroll
      for (int k = 0; k < MAX_PER_LANE; k++) {
        if (SRC_INDEX(k) >= hi) continue;
        const float fx = __fmul_rn(qx[k], inv_cell), fy = __fmul_rn(qy[k], inv_cell);
        if (!(fabsf(fx) < CELL_LIMIT) || !(fabsf(fy) < CELL_LIMIT)) continue;  // NaN / inf: no match
        const int32_t icx = (int32_t)floorf(fx), icy = (int32_t)floorf(fy);
        const int32_t lo = icx - 1, hi2 = icx + 1;
        const bool two = (lo >> 2) != (hi2 >> 2);  // the three cells straddle a group of four
        for (int32_t oy = -1; oy <= 1; oy++)
          for (int32_t r = 0; r < 2; r++) {
            if (r == 1 && !two) break;
            // first run: from cell lo to cell hi2 or the end of lo's group; second run: the cells of hi2's group
            const uint32_t b0 = r == 0 ? cell_hash(lo, icy + oy) : group_hash(hi2 >> 2, icy + oy);
            const uint32_t b1 = r == 0 ? (two ? (b0 | 3u) : b0 + 2u) : b0 + ((uint32_t)hi2 & 3u);
            const uint32_t e = s_start[b1 + 1];
            for (uint32_t i = s_start[b0]; i < e; i++) {
              const float2 g = s_tgt[i];  // (bucket order: consecutive addresses)
              const float dx = __fsub_rn(g.x, qx[k]), dy = __fsub_rn(g.y, qy[k]);
              const float d2 = __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));
              bool ok = true;
              if (GATE) {
                const float2 gn = s_tgn[i];
                ok = __fsqrt_rn(d2) < thr && fabsf(dot2(gn.x, snx[k], gn.y, sny[k])) > min_cos;
              }
              // the order of visits is arbitrary: (d2, index) lexicographic = "lowest index wins ties".  bi[k] holds the
              // SLOT of the best so far; the indices are fetched only on an exact tie of the squared distances
              bool better = ok && d2 < best[k];
              if (ok && d2 == best[k] && bi[k] >= 0) better = s_sorted[i] < s_sorted[bi[k]];
              best[k] = better ? d2 : best[k];
              bi[k] = better ? (int32_t)i : bi[k];
            }
          }
        if (bi[k] >= 0) bi[k] = (int32_t)s_sorted[bi[k]];  // slot -> index of the target point
      }
    } else {
      for (int32_t t0 = 0; t0 < nt; t0 += TGT_CHUNK) {
        const int32_t nc = min(nt - t0, TGT_CHUNK);
        if (!hashed) {  // (a hashed block already holds the whole target cloud in s_tgt)
          __syncthreads();
          for (int32_t i = tid; i < nc; i += CT) {
            s_tgt[i] = xy[tb + t0 + i];
            if (GATE) s_tgn[i] = normals[tb + t0 + i];
          }
          __syncthreads();
        }
        for (int32_t i = 0; i < nc; i++) {
          const float2 g = s_tgt[i];  // same address in every lane: LDS broadcast
          // (a hashed block holds its cloud in bucket order: position i is point s_sorted[i], and ties go by index)
          const int32_t idx = hashed ? (int32_t)s_sorted[i] : t0 + i;
#pragma unroll
          for (int k = 0; k < MAX_PER_LANE; k++) {
            const float dx = __fsub_rn(g.x, qx[k]), dy = __fsub_rn(g.y, qy[k]);
            const float d2 = __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));
            bool ok = true;
            if (GATE) {
              const float2 gn = s_tgn[i];
              ok = __fsqrt_rn(d2) < thr && fabsf(dot2(gn.x, snx[GATE ? k : 0], gn.y, sny[GATE ? k : 0])) > min_cos;
            }
            const bool better = ok && (d2 < best[k] || (d2 == best[k] && (uint32_t)idx < (uint32_t)bi[k]));  // the lowest index wins ties
            best[k] = better ? d2 : best[k];
            bi[k] = better ? idx : bi[k];
          }
        }
      }
    }
    // keep = nearest within the threshold (kdtree.cc:253-305 + solver.cc:84-89).  Source order is
    // round-major (round k holds points s0 + k*CT ..): per round a wave ballot gives the in-wave rank,
    // one LDS table of (round, wave) counts and ONE barrier give the rest.
    bool keep[MAX_PER_LANE];
    int32_t rank[MAX_PER_LANE];
    const int lane = tid & 63, wv = tid >> 6;
    __syncthreads();  // s_scan is free again (previous pass / bucket scan)
#pragma unroll
    for (int k = 0; k < MAX_PER_LANE; k++) {
      keep[k] = (SRC_INDEX(k) < hi) && bi[k] >= 0 && (__fsqrt_rn(best[k]) < thr);
      const unsigned long long m = __ballot(keep[k]);
      rank[k] = __builtin_popcountll(m & ((1ull << lane) - 1ull));
      if (lane == 0) s_scan[k * (CT / 64) + wv] = __builtin_popcountll(m);
    }
    __syncthreads();
    int32_t pos = written;
#pragma unroll
    for (int k = 0; k < MAX_PER_LANE; k++) {
      int32_t before = 0, round_total = 0;
#pragma unroll
      for (int w = 0; w < CT / 64; w++) {
        const int32_t c = s_scan[k * (CT / 64) + w];
        before += (w < wv) ? c : 0;
        round_total += c;
      }
      if (keep[k]) {
        const int32_t at = pos + before + rank[k];
        const float2 p = xy[sb + SRC_INDEX(k)], ps = normals[sb + SRC_INDEX(k)];
        const float2 g = xy[tb + bi[k]], gn = normals[tb + bi[k]];
        out[2 * (size_t)at] = make_float4(p.x, p.y, g.x, g.y);
        out[2 * (size_t)at + 1] = make_float4(ps.x, ps.y, gn.x, gn.y);
      }
      pos += round_total;
    }
    written = pos;
#undef SRC_INDEX
  }
  if (tid == 0) counts[b] = written;
}

// ---- compaction into the contiguous layout of the residual batch --------------------------
// block_offsets = exclusive scan of counts (one workgroup; n_blocks is ~1e4)
__global__ __launch_bounds__(1024) void corr_scan_kernel(const int32_t *__restrict__ counts,
                                                         int32_t n_blocks,
                                                         int32_t *__restrict__ block_offsets) {
  __shared__ int32_t s[1024];
  __shared__ int32_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int32_t base = 0; base < n_blocks; base += 1024) {
    const int32_t i = base + threadIdx.x;
    const int32_t v = i < n_blocks ? counts[i] : 0;
    s[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      const int32_t u = threadIdx.x >= off ? s[threadIdx.x - off] : 0;
      __syncthreads();
      s[threadIdx.x] += u;
      __syncthreads();
    }
    if (i < n_blocks) block_offsets[i] = carry + s[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 0) carry += s[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) block_offsets[n_blocks] = carry;
}

__global__ __launch_bounds__(CT) void corr_compact_kernel(const float4 *__restrict__ padded,
                                                          const int64_t *__restrict__ cap_offsets,
                                                          const int32_t *__restrict__ block_offsets,
                                                          float4 *__restrict__ corr,
                                                          int32_t *__restrict__ corr_block) {
  const int b = blockIdx.x;
  const int32_t o = block_offsets[b], n = block_offsets[b + 1] - o;
  const float4 *src = padded + 2 * (size_t)cap_offsets[b];
  for (int32_t i = threadIdx.x; i < 2 * n; i += CT) corr[2 * (size_t)o + i] = src[i];
  for (int32_t i = threadIdx.x; i < n; i += CT) corr_block[o + i] = b;
}

}  // namespace

int launch_corr_search(const float *d_xy, const float *d_normals, const int32_t *d_offsets, int32_t n_scans,
                       const int32_t *d_block_src, const int32_t *d_block_tgt, int32_t n_blocks,
                       const float *d_pose_aff, float thr, float min_cos, bool gate,
                       const int64_t *d_cap_offsets, float *d_corr_padded, int32_t *d_counts, hipStream_t s) {
  if (n_blocks == 0) return NHIP_OK;
  timer_begin(NHIP_TIMER_CORR, s);
  if (gate)
    hipLaunchKernelGGL(corr_search_kernel<true>, dim3(n_blocks), dim3(CT), 0, s,
                       reinterpret_cast<const float2 *>(d_xy), reinterpret_cast<const float2 *>(d_normals),
                       d_offsets, d_block_src, d_block_tgt, d_pose_aff, thr, min_cos, d_cap_offsets,
                       reinterpret_cast<float4 *>(d_corr_padded), d_counts, n_scans, dev_status());
  else
    hipLaunchKernelGGL(corr_search_kernel<false>, dim3(n_blocks), dim3(CT), 0, s,
                       reinterpret_cast<const float2 *>(d_xy), reinterpret_cast<const float2 *>(d_normals),
                       d_offsets, d_block_src, d_block_tgt, d_pose_aff, thr, 0.f, d_cap_offsets,
                       reinterpret_cast<float4 *>(d_corr_padded), d_counts, n_scans, dev_status());
  timer_end(NHIP_TIMER_CORR, s);
  NHIP_TRY_HIP(hipGetLastError());
  return NHIP_OK;
}

int launch_corr_compact(const float *d_corr_padded, const int64_t *d_cap_offsets,
                        const int32_t *d_counts, int32_t n_blocks, int32_t *d_block_offsets,
                        float *d_corr, int32_t *d_corr_block, hipStream_t s) {
  hipLaunchKernelGGL(corr_scan_kernel, dim3(1), dim3(1024), 0, s, d_counts, n_blocks, d_block_offsets);
  if (n_blocks > 0)
    hipLaunchKernelGGL(corr_compact_kernel, dim3(n_blocks), dim3(CT), 0, s,
                       reinterpret_cast<const float4 *>(d_corr_padded), d_cap_offsets, d_block_offsets,
                       reinterpret_cast<float4 *>(d_corr), d_corr_block);
  NHIP_TRY_HIP(hipGetLastError());
  return NHIP_OK;
}

}  // namespace nhip
