// nhip_bnb_params.h -- kernel parameters of the branch-and-bound matcher, shared by its two translation units:
// nhip_bnb.hip (the product kernels) and nhip_bnb_instr.hip (the same kernels with their instrumentation compiled in).
#pragma once
#include "nhip_common.h"

namespace nhip {
namespace bnb {

// ---- sizes the kernels (nhip_bnb.hip) and their host side (nhip_bnb_host.hip) share
constexpr int BNB_WAVES = 8;          // waves per workgroup of csm_bnb_kernel
constexpr int NB = BNB_MAX_NB;        // blocks per axis held in registers (11: nx, ny <= 88)
constexpr int MAX_ROT = 340;          // rotations per search: QCAP * 4 bytes hold their 12 bytes of ordering data
constexpr int QCAP = 1024;            // candidate queue entries per workgroup (overflow is evaluated by the wave that found it)
constexpr int RUN_SHIFT = 25;         // run-list entry = pooled offset | (index of the run's first point mod 128) << RUN_SHIFT
constexpr int OC = 18;                // 64-entry chunks the by-rotation passes are unrolled for
constexpr int OCL = 17;               // chunks of window origins a wave holds: 1088 points (a 1081-beam scan)
constexpr int ORG_WAVE = OCL * 64;    // words of LDS per wave
constexpr int ORG_LDS = BNB_WAVES * ORG_WAVE * 4;  // bytes per workgroup (34,816: the 1200 x 1200 grid's pooled table is 35,712)
constexpr uint32_t ORG_LIMIT = 1u << 13;           // rows / columns of a stored grid the packed origins hold
constexpr int BNB_STATS_PAIRS = 1 << 20;           // per-pair counters kept by NHIP_BNB_STATS=1
constexpr int BNB_STATS_HEAD = 16;    // totals: 4 counts, then shader-clock sums of the by-rotation kernel (see nhip_bnb_stats_levels)

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
  uint32_t front_min;   // candidates from which a pair's workgroups go to the FRONT of its XCD's list (0: pair order throughout)
  int32_t pair_base;    // index of this launch's pair 0 in the caller's arrays (rounds of the split form): what a bad id is reported at
  int32_t n_pairs, n_theta, nx, ny, hx, hy, nbx, nby;
  int32_t S, pad, pitch, rows, max_shift;
  int32_t pool_pitch, pool_rows, pairs_per_xcd;
  int32_t pool4_pitch;
  int32_t lds_first;    // bytes of the kernel's first LDS region: max(pooled table if staged, origins)
  int32_t general_all;  // the general instantiation takes every pair (NHIP_BNB_QUEUE=1)
  int32_t short_scans;  // the caller vouches that every scan fits the by-rotation form (NHIP_SEARCH_SHORT_SCANS)
  int32_t levels;  // 2: candidates are refined through the 4 x 4 sub-block bounds; 1: evaluated whole (NHIP_BNB_LEVELS)
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
