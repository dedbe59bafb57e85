// nhip_common.h -- shared internals of libnautilus_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/nautilus_hip.h"

namespace nhip {

// ---- error plumbing (thread-local message, int codes across the ABI) ----
void set_error(const char *fmt, ...);
int hip_fail(hipError_t e, const char *what, const char *file, int line);

#define NHIP_TRY_HIP(expr)                                            \
  do {                                                                \
    hipError_t _e = (expr);                                           \
    if (_e != hipSuccess) return ::nhip::hip_fail(_e, #expr, __FILE__, __LINE__); \
  } while (0)

#define NHIP_REQUIRE(cond, ...)          \
  do {                                   \
    if (!(cond)) {                       \
      ::nhip::set_error(__VA_ARGS__);    \
      return NHIP_ERR_ARG;               \
    }                                    \
  } while (0)

int require_device();

// Environment switches that select a form of the matcher (tests and measurements: every form returns the same records)
// are honoured only by a process that had NHIP_TUNABLES=1 in its environment when the library first asked (looked up
// once, std::call_once).  Such a process reads them at every launch, so a test can change form between calls; any other
// process -- a shipped host -- never reads any of them: tunable() is a null pointer there and every policy is its default.
const char *tunable(const char *name);

// ---- ids that live in device memory ----
// The "_dev" entry points read scan ids, grid slots, block and pose indices from DEVICE arrays the host cannot look at
// without a round trip.  Every kernel checks such an id against the count its caller passed before it forms an address
// from it: an id outside [0, count) is treated as an empty scan / a pair that scores nothing / a row that is skipped, and
// is reported through the current device's status words, which nhip_dev_status() turns into NHIP_ERR_ARG after the
// stream has drained.  A stale id costs the caller an error code, never the process (an out-of-bounds read in a kernel
// is an HSA abort of the whole process; it happened once, in a test: DESIGN.md section 5, "Ids in device memory").
struct IdBounds {          // (plain data: it travels inside the kernels' parameter blocks)
  int32_t n_scans;         // scans behind d_offsets (n_scans + 1 entries)
  int32_t n_slots;         // grids behind d_grids
  uint32_t *status;        // the device's status words, or null (ids are still checked, nothing is reported)
};
constexpr uint32_t BAD_TARGET_ID = 1u, BAD_PAIR_SRC = 2u, BAD_PAIR_SLOT = 4u, BAD_BLOCK_ID = 8u, BAD_POSE_ID = 16u,
                   BAD_SCAN_ID = 32u;
constexpr int DEV_STATUS_WORDS = 4;  // {OR of the kinds seen, kind / value / index of the first report}
uint32_t *dev_status();              // of the current device (allocated on the device's first use; null if that failed)
#ifdef __HIPCC__
__device__ __forceinline__ void flag_bad_id(uint32_t *status, uint32_t kind, int32_t value, int32_t index) {
  if (!status) return;
  if (atomicOr(status, kind) != 0u) return;  // (the first report keeps the details)
  status[1] = kind;
  status[2] = (uint32_t)value;
  status[3] = (uint32_t)index;
}
__device__ __forceinline__ bool id_in(int32_t id, int32_t n) { return (uint32_t)id < (uint32_t)n; }
// a pair's source scan and grid slot; false: the pair scores nothing (and is reported)
__device__ __forceinline__ bool pair_ids_ok(const IdBounds &B, int32_t src, int32_t slot, int32_t pair, bool report) {
  const bool s_ok = id_in(src, B.n_scans), g_ok = id_in(slot, B.n_slots);
  if (s_ok && g_ok) return true;
  if (report) flag_bad_id(B.status, s_ok ? BAD_PAIR_SLOT : BAD_PAIR_SRC, s_ok ? slot : src, pair);
  return false;
}
#endif

// ---- grid layout (host) ----
struct GridLayout {
  int32_t S, pad, pitch, R;
  int32_t cb;      // bytes per cell: 1 or 2
  int32_t levels;  // quantisation steps: 255 or 65535
  int64_t K;  // tap sum
  int64_t plain_bytes; // pitch * rows: a grid in plain row-major form (the downloads' form)
  bool has_image;      // false with NHIP_GRID_NO_IMAGE: no row-major image (and no skip map) in the slot
  int64_t grid_bytes;  // the stored image: plain_bytes, or 0 without one
  int64_t skip_bytes;  // skip_pitch(pitch) * rows rounded up to 16: the skip map that follows the image
  int64_t pool_bytes;  // pool_rows * pool_pitch: the max-pooled table (branch-and-bound bounds) after the skip map
  int64_t pool4_bytes; // pool4_rows * pool4_pitch: the stride-4 pooled table (second bound level) after the first
  int64_t hi_bytes;    // 16-bit cells: stored bytes of the plane of high bytes after the second pooled table (two tiled
                       // copies: hi_tiled() below); else 0
  int32_t hi_pitch;    // bytes per row of the plane in its plain (downloaded) form: the 8-bit pitch of the same grid
  int32_t hi_tpr;      // tiles per tile row of a copy
  int64_t hi_copy_bytes;  // bytes of one copy
  int64_t t16_bytes;   // 16-bit cells: after the two copies, the 16-bit image once more, tiled 8 rows x 8 cells (t16_tiled())
  int32_t t16_tpr;     // its tiles per tile row
  int64_t hits_bytes;  // after the matcher's planes: the HIT RASTER, one bit per cell (rows S + 2 * HIT_PAD, hits_pitch bytes each):
  int32_t hits_pitch;  // what the table was blurred from.  The exact-score pass (NHIP_SEARCH_EXACT_SCORE) recomputes the
                       // integer blur sums of the 1081 cells the winning pose reads from it, and their logarithms in double
  int64_t slot_bytes;  // grid_bytes + skip_bytes + pool_bytes + pool4_bytes + hi_bytes + hits_bytes: stride between grids
  int32_t pool_pitch, pool_rows;
  int32_t pool4_pitch, pool4_rows;
  double Lf, step;
};
// The plane of high bytes is stored in tiles of 8 rows x 16 bytes = one 128-byte cache line, so that the points of a
// chunk -- neighbours along a wall whatever its direction -- read their rows from a dozen lines instead of forty
// (in a row-major plane every row of every point is a line of its own).  A row read of the matcher is 8 or 12 bytes
// from a 4-byte-aligned column and must not cross a tile: there are TWO copies, the second with its tiles shifted by
// 8 columns, and a read takes the copy in which it starts in a tile's first half.
// Byte offset of (row, col) in copy cp (0 / 1):
constexpr uint32_t HI_TILE_BYTES = 128u;
__host__ __device__ __forceinline__ uint32_t hi_tiled(uint32_t row, uint32_t col, uint32_t cp, uint32_t tpr, uint32_t copy_bytes) {
  const uint32_t c = col + 8u * cp;
  return cp * copy_bytes + ((row >> 3) * tpr + (c >> 4)) * HI_TILE_BYTES + (row & 7u) * 16u + (c & 15u);
}

// The matcher's exact 16-bit pose sums read ONE cell per point: from a copy of the 16-bit image tiled 8 rows x 8 cells
// (128 bytes), for the same reason.  Byte offset of cell (row, col):
__host__ __device__ __forceinline__ uint32_t t16_tiled(uint32_t row, uint32_t col, uint32_t tpr) {
  return ((row >> 3) * tpr + (col >> 3)) * HI_TILE_BYTES + (row & 7u) * 16u + (col & 7u) * 2u;
}

// The hit raster's zero border, in cells, on every side: >= the largest blur radius (16), and a multiple of 32 so that the
// 64 x 64 tiles of the build start on dword boundaries of the bit rows.  Bit (row, col) of the raster is bit
// (col + HIT_PAD) & 31 of dword (col + HIT_PAD) >> 5 of bit row row + HIT_PAD.
constexpr int HIT_PAD = 32;

// Branch and bound works on 8 x 8 blocks of translations; a pooled entry covers the 15 x 15 stored cells an 8 x 8
// block can reach from any window origin with the same (row >> 3, col >> 3).
constexpr int BNB_B = 8;
constexpr int BNB_POOL = 2 * BNB_B - 1;
constexpr int BNB_MAX_NB = 11;  // blocks per axis the kernel's register layout holds: nx, ny <= 88
// Second level: the four 4 x 4 sub-blocks of a block, bounded from a table pooled over 7 x 7 cells at stride 4.
constexpr int BNB_B4 = 4;
constexpr int BNB_POOL4 = 2 * BNB_B4 - 1;
// Geometry the skip map shares with the correlation kernel: a wave of csm_correlate_kernel owns
// CSM_WAVE_ROWS plane rows and reads CSM_ROW_DW aligned dwords of each (nhip_csm.hip).
constexpr int CSM_WAVE_ROWS = 21;
constexpr int CSM_ROW_DW = 21;  // per cell byte: a 16-bit-cell strip spans 2 * CSM_ROW_DW dwords
#ifdef __HIPCC__
// floor(RN(v / res)) without the division on the common path.  m = RN(v * RN(1 / res)) differs
// from the correctly rounded quotient by less than |m| * 2^-51, so the two floors can differ only
// if m lies within that distance of an integer; those lanes (one point in ~10^12) take the division.
__device__ __forceinline__ double floor_quotient(double v, double res, double inv_res) {
  const double m = __dmul_rn(v, inv_res);
  double f = floor(m);
  const double frac = __dsub_rn(m, f);  // exact
  const double tol = __dmul_rn(fabs(m), 0x1p-50);
  if (frac <= tol || __dsub_rn(1.0, frac) <= tol) f = floor(__ddiv_rn(v, res));
  return f;
}
#endif

// grid-build workspace: 256-byte header, [16-bit cells: the 65536-entry quantiser threshold table,] then per
// target one occupancy byte per 64x64 tile, one 4-byte list slot per tile and GRID_WS_MASK_WORDS words of LINE MASKS per
// list slot: which 128-byte lines of the matcher's tiled planes the blur wrote inside the tile (the next rebuild clears
// those and no others -- two thirds of a listed tile's lines hold nothing)
constexpr int64_t GRID_WS_HEADER = 256;
constexpr int64_t GRID_WS_THR16 = 65536 * 4;
constexpr int GRID_WS_MASK_WORDS = 5;  // 32 lines of the first copy of the 8-bit plane, 40 of the shifted copy, 64 of the 16-bit copy
inline int64_t grid_ws_per_target(int32_t S) {
  const int64_t tiles = (S + 63) / 64;
  return (5 + 4 * GRID_WS_MASK_WORDS) * tiles * tiles;
}
// bytes per skip-map row: one bit per aligned dword column, whole 8-byte words
__host__ __device__ constexpr int32_t skip_pitch(int32_t pitch) { return ((pitch / 4 + 63) / 64) * 8; }
int make_layout(const nhip_grid_spec_t *spec, GridLayout *L);

// Device-side constant tables of one grid spec (taps + quantiser thresholds), cached.
struct GridTables {
  int32_t taps[129];
  uint32_t thr[256];               // 8-bit cells
  const uint32_t *thr16 = nullptr;  // 16-bit cells: 65536 entries, owned by a process-wide cache
};
int make_tables(const nhip_grid_spec_t *spec, const GridLayout &L, GridTables *T);

// ---- in-stream timing ----
void timer_begin(int id, hipStream_t s);
void timer_end(int id, hipStream_t s);

// ---- kernel launchers (defined in the .hip files) ----
int launch_grid_build(const float *d_xy, const int32_t *d_offsets, int32_t n_scans, const int32_t *d_target_ids,
                      int32_t n_targets, const nhip_grid_spec_t *spec, const GridLayout &L,
                      uint8_t *d_grids, void *d_ws, int64_t ws_bytes, hipStream_t s, bool incremental = false);

int launch_csm_match(const float *d_xy, const int32_t *d_offsets, const IdBounds &ids, const uint8_t *d_grids,
                     const nhip_grid_spec_t *spec, const GridLayout &L, const int32_t *d_pair_src,
                     const int32_t *d_pair_slot, const double *d_rot0_cs, const double *d_delta_cs,
                     const int32_t *d_pair_origin, int32_t n_pairs, const nhip_search_t *search,
                     uint64_t *d_keys, nhip_match_t *d_out, int32_t *d_sums, hipStream_t s,
                     void *d_workspace = nullptr, int64_t workspace_bytes = 0, const int32_t *d_pair_kbase = nullptr);

// branch-and-bound matcher (nhip_bnb.hip); returns NHIP_ERR_STATE-free: `*handled` = 0 when the lattice does not fit it
int launch_csm_bnb(const float *d_xy, const int32_t *d_offsets, const IdBounds &ids, const uint8_t *d_grids,
                   const nhip_grid_spec_t *spec, const GridLayout &L, const int32_t *d_pair_src,
                   const int32_t *d_pair_slot, const double *d_rot0_cs, const double *d_delta_cs,
                   const int32_t *d_pair_origin, int32_t n_pairs, const nhip_search_t *search,
                   uint64_t *d_keys, nhip_match_t *d_out, int32_t *d_sums, hipStream_t s, int *handled,
                   void *d_workspace = nullptr, int64_t workspace_bytes = 0, const int32_t *d_pair_kbase = nullptr);
int64_t bnb_workspace_bytes(int32_t n_pairs);
int64_t bnb_workspace_bytes_lists(int32_t n_pairs);
void bnb_last_launch(int32_t out[8]);
bool bnb_fits(const GridLayout &L, const nhip_search_t *search);
int bnb_stats_read(unsigned long long out[16]);
int bnb_timeline_read(unsigned long long *out, int32_t n);
int bnb_timeline_cand_read(unsigned long long *out, int32_t n);
int bnb_stats_per_pair(unsigned long long *out, int32_t n);
void launch_csm_finalize(const uint64_t *d_keys, const int32_t *d_pair_src, const int32_t *d_offsets, int32_t n_scans, int32_t n_pairs,
                         int32_t nx, int32_t ny, const GridLayout &L, nhip_match_t *d_out, int32_t *d_sums, hipStream_t s);

int launch_csm_scores(const float *d_xy, const int32_t *d_offsets, const uint8_t *d_grids,
                      const nhip_grid_spec_t *spec, const GridLayout &L, int32_t src, int32_t slot,
                      const double *d_rot0_cs, const double *d_delta_cs, int32_t origin_x,
                      int32_t origin_y, const nhip_search_t *search, int32_t *d_sums,
                      hipStream_t s);

// the kernel that performs every add, 16-bit cells (nhip_csm16.hip); called by launch_csm_match / launch_csm_scores
int launch_csm16_match(const float *d_xy, const int32_t *d_offsets, const IdBounds &ids, const uint8_t *d_grids,
                       const nhip_grid_spec_t *spec, const GridLayout &L, const int32_t *d_pair_src,
                       const int32_t *d_pair_slot, const double *d_rot0_cs, const double *d_delta_cs,
                       const int32_t *d_pair_origin, int32_t n_pairs, const nhip_search_t *search,
                       uint64_t *d_keys, nhip_match_t *d_out, int32_t *d_sums, hipStream_t s);
int launch_csm16_scores(const float *d_xy, const int32_t *d_offsets, const uint8_t *d_grids,
                        const nhip_grid_spec_t *spec, const GridLayout &L, int32_t src, int32_t slot,
                        const double *d_rot0_cs, const double *d_delta_cs, int32_t origin_x, int32_t origin_y,
                        const nhip_search_t *search, int32_t *d_sums, hipStream_t s);
// every add for lattices of few translations (nx * ny <= 256), both cell widths (nhip_csm_small.hip)
// Internal bits of nhip_search_t::flags (never part of the ABI: the extern "C" entry points mask them off).  The chained
// GetTransformation sets them for the searches it sends through the small-plane kernel: the keys were zeroed by the
// kernel before (no memset node), and the kernel after decodes the keys itself (no finalize node) -- dropin_bridge_kernel for
// the coarse level, csm_exact_score_kernel for the fine one.
constexpr int32_t SEARCH_I_KEYS_ZERO = 1 << 28;
constexpr int32_t SEARCH_I_NO_FINALIZE = 1 << 29;
bool csm_small_plane_fits(const nhip_search_t *search);
bool csm_small_tiled_fits(const nhip_search_t *search, int32_t n_pairs, int32_t *tile_rows, int32_t *n_tiles);
int launch_csm_small_match(const float *d_xy, const int32_t *d_offsets, const IdBounds &ids, const uint8_t *d_grids,
                           const nhip_grid_spec_t *spec, const GridLayout &L, const int32_t *d_pair_src,
                           const int32_t *d_pair_slot, const double *d_rot0_cs, const double *d_delta_cs,
                           const int32_t *d_pair_origin, int32_t n_pairs, const nhip_search_t *search,
                           uint64_t *d_keys, nhip_match_t *d_out, int32_t *d_sums, hipStream_t s);
// skip maps of n finished 16-bit grids (the handle API's late build; occupancy unknown: every map tile is computed)
int launch_skipmap_build(uint8_t *d_grids, int32_t n_grids, const GridLayout &L, hipStream_t s);
// true when a search on these grids takes the kernel that performs every add
bool csm_takes_exhaustive(const GridLayout &L, const nhip_search_t *search);

int launch_resid_lidar(int kind, const float *d_corr, const int32_t *d_corr_block, int64_t n_corr,
                       const int32_t *d_block_src, const int32_t *d_block_tgt, int32_t n_blocks,
                       const double *d_poses, int32_t n_poses, double *d_block_consts,
                       double *d_res, double *d_jsrc, double *d_jtgt, hipStream_t s, double *d_jtgt_theta = nullptr,
                       int32_t block_base = 0, double *d_q = nullptr);

int launch_resid_normal_eq(int kind, const float *d_corr, const int32_t *d_block_offsets,
                           const int32_t *d_block_src, const int32_t *d_block_tgt, int32_t n_blocks,
                           const double *d_poses, int32_t n_poses, double *d_block_consts, double *d_out,
                           hipStream_t s);

int launch_corr_search(const float *d_xy, const float *d_normals, const int32_t *d_offsets, int32_t n_scans,
                       const int32_t *d_block_src, const int32_t *d_block_tgt, int32_t n_blocks,
                       const float *d_pose_aff, float thr, float min_cos, bool gate, const int64_t *d_cap_offsets,
                       float *d_corr_padded, int32_t *d_counts, hipStream_t s);

int launch_corr_compact(const float *d_corr_padded, const int64_t *d_cap_offsets,
                        const int32_t *d_counts, int32_t n_blocks, int32_t *d_block_offsets,
                        float *d_corr, int32_t *d_corr_block, hipStream_t s);

int launch_lc_scatter_scores(const float *d_xy, const int32_t *d_offsets, int32_t n_scans, double *d_scores, hipStream_t s);
int launch_lc_chi_square(const double *d_poses, int32_t n_poses, const int32_t *d_src, const int32_t *d_tgt, const float *d_cov,
                         int32_t n, double max_score, double *d_scores, uint8_t *d_flags, hipStream_t s);
int launch_lc_pair_gate(const double *d_poses, int32_t n_poses, const int32_t *d_cand, int32_t n, double max_range,
                        int32_t min_sep, uint8_t *d_flags, hipStream_t s);

int launch_resid_point_to_line(const float *d_segments, const float *d_points,
                               const int32_t *d_point_block, int64_t n_points,
                               const int32_t *d_block_pose, const int32_t *d_block_line,
                               int32_t n_blocks, const double *d_poses, int32_t n_poses, const double *d_line_poses,
                               int32_t n_line_poses, double *d_res, double *d_jpose, double *d_jline, hipStream_t s);

int launch_resid_odometry(const float *d_t_odom, const float *d_r_odom, const int32_t *d_pose_i,
                          const int32_t *d_pose_j, int32_t n_factors, double tw, double rw,
                          const double *d_poses, int32_t n_poses, double *d_res, double *d_ji, double *d_jj,
                          hipStream_t s);

}  // namespace nhip
