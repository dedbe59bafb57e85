// nhip_bnb_params.h -- kernel parameters of the branch-and-bound matcher, shared by its two translation units:
// nhip_bnb.hip (the product kernels) and nhip_bnb_instr.hip (the same kernels with their instrumentation compiled in).
#pragma once
#include "nhip_common.h"

namespace nhip {
namespace bnb {

// One rotation of a pair with many candidates, handed to the second kernel: (pair, rotation) and the mask of its
// 121 candidate blocks.
struct RotEntry {
  unsigned long long w[4];  // w[0]: pair << 24 | rotation; w[1..3]: candidate blocks 0..40, 41..81, 82..120
};

struct BnbParams {
  const float2 *xy;
  const int32_t *offsets;
  const uint8_t *grids;
  const int32_t *pair_src;
  const int32_t *pair_slot;
  IdBounds ids;  // counts the pairs' scan ids and grid slots are checked against (nhip_common.h)
  const double *rot0_cs;
  const double *delta_cs;
  const int32_t *pair_origin;
  const int32_t *pair_kbase;  // optional: entry of delta_cs that is pair i's rotation 0 (a search's rotations dealt over
                              // several "pairs" = workgroups: nhip_csm_get_transformation's fine level); null = 0
  unsigned long long *keys;
  unsigned long long *timeline;  // optional (NHIP_BNB_TIMELINE=1): per pair 4 x 100 MHz ticks: start, bounds, seeds, end
  unsigned long long *stats;  // optional: [0] blocks evaluated whole, [1] blocks in all, [2] candidates refined,
                              //   [3] 4 x 4 sub-blocks evaluated; then one count of candidates per pair
  RotEntry *rot_list;          // optional lists of (pair, rotation) work items in the caller's workspace, one per XCD so
  uint32_t *rot_count;        //   that a pair's rotations are worked where its grid is L2-resident; per XCD 32 bytes of
  uint32_t rot_cap;           //   counters {filled, next}; entries per list
  uint32_t heavy_min;         // candidates (after the seeds) from which a PAIR hands its rotations over ...
  uint32_t keep_ranks;        // ... except its first keep_ranks rotations in best-first order (one per wave)
  // Split form (large batches): the workgroup of a pair ends after bounds and seeds and leaves its state -- per live
  // rotation, best first, the 128-word row of block bounds (word 126: the rotation's highest bound, word 127: the
  // rotation) -- in the caller's workspace; the candidates are worked by a second launch, heaviest pairs first.
  uint32_t *ps_rows;    // [pair][rank][128]; null = the fused form
  uint32_t *ps_count;   // [pair] candidate blocks the seeds have left
  uint32_t *ps_live;    // [pair] rotations with at least one of them
  uint32_t *ps_next;    // [pair] next rank to hand out (pairs worked by several workgroups)
  uint32_t *ps_nw;      // [pair] workgroups of the second launch that share the pair
  int32_t *ps_work;     // [8][ps_work_stride] pair of each workgroup of the second launch (-1: none), per XCD
  uint32_t *ps_ticket;  // spread form: counter of additional workgroups dealt over the eight lists (null: round 3's lists)
  int32_t ps_work_stride;
  uint32_t split_min;   // candidates per additional workgroup of a pair
  uint32_t split_max;   // workgroups per pair at most
  uint32_t sort_coarse; // log2 of the width of the ordering's buckets in sixteenths of an octave of the candidate count
  uint32_t seeds;       // waves of a pair's workgroup that evaluate a seed block (those with the highest bounds)
  int32_t n_pairs, n_theta, nx, ny, hx, hy, nbx, nby;
  int32_t S, pad, pitch, rows, max_shift;
  int32_t pool_pitch, pool_rows, pairs_per_xcd;
  int32_t pool4_pitch;
  int32_t lds_first;    // bytes of the kernel's first LDS region: max(pooled table if staged, origins)
  int32_t whole_min;    // sub-blocks alive from which an 8-bit block is evaluated whole (3; NHIP_BNB_WHOLE_MIN)
  int32_t general_all;  // the general instantiation takes every pair (NHIP_BNB_QUEUE=1)
  int32_t short_scans;  // the caller vouches that every scan fits the by-rotation form (NHIP_SEARCH_SHORT_SCANS)
  int32_t levels;  // 2: candidates are refined through the 4 x 4 sub-block bounds; 1: evaluated whole (NHIP_BNB_LEVELS)
  int32_t debug;   // NHIP_BNB_DEBUG (timing experiments only, results are wrong): 1 = no phase 3, 2 = bounds only,
                   // 4 = phase 3 without exact sums, 5 = phase 3 without sub-block bounds and exact sums,
                   // 26 / 27 = bounds only, without their reductions / gathers; 28 = without the run lists too,
                   // 29 = without the window origins too, 30 = no chunk loop, 31 = no rotations (staging + launch).
                   // With NHIP_BNB_STATS=1 the counters' atomics dominate the short forms: time those as product
                   // builds with -DNHIP_BNB_EXPERIMENT=<value> (tools/bnb_variants.sh)
  int64_t grid_bytes, skip_bytes, slot_bytes, pool_bytes, pool4_bytes;
  int64_t hi_offset, hi_bytes;  // 16-bit grids: the plane of high bytes inside a slot
  int32_t hi_pitch;
  int32_t hi_tpr;          // tiles per tile row and bytes of one copy of the tiled plane (nhip_common.h hi_tiled)
  int64_t hi_copy_bytes;
  int64_t t16_bytes;       // the tiled copy of the 16-bit image behind the two copies (nhip_common.h t16_tiled)
  int32_t t16_tpr;
  double res, inv_res;
  float inv_res_f;  // RN_f32(1 / res): the single-precision path of the window origins
};

// Launches the matcher's kernel(s) for one batch: the product build, or (nhip_bnb_instr.hip) the build that honours
// P.stats / P.timeline / P.debug.  lds: dynamic LDS bytes of csm_bnb_kernel; blocks: its grid.
int launch_bnb_kernels(const BnbParams &P, int cb, bool pool_lds, size_t lds, int64_t blocks, bool second_kernel,
                       hipStream_t s);
int launch_bnb_kernels_instr(const BnbParams &P, int cb, bool pool_lds, size_t lds, int64_t blocks, bool second_kernel,
                             hipStream_t s);
// The split form (P.ps_rows set) on one batch: bounds + seeds + the ordering of the pairs by candidates left (a),
// then the candidates (b; on any stream ordered behind a).
int launch_bnb_split_a(const BnbParams &P, int cb, bool pool_lds, size_t lds, int64_t blocks, hipStream_t s);
int launch_bnb_split_a_instr(const BnbParams &P, int cb, bool pool_lds, size_t lds, int64_t blocks, hipStream_t s);
int launch_bnb_split_b(const BnbParams &P, int cb, hipStream_t s);
int launch_bnb_split_b_instr(const BnbParams &P, int cb, hipStream_t s);

}  // namespace bnb
}  // namespace nhip
