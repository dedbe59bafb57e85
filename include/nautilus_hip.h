/*
 * nautilus_hip.h -- C ABI of the MI355X (gfx950) implementation of nautilus's loop-closure
 * correlative scan matcher and batched Ceres-residual evaluation.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ / torch types.  Each entry
 * point cites the reference interface it replaces (paths relative to the nautilus tree):
 *
 *   CorrelativeScanMatcher(30, 2, 0.3, 0.01)              src/optimization/solver.cc:56,633
 *   CorrelativeScanMatcher::GetTransformation(...)        src/optimization/solver.cc:634-638
 *   lookup-table geometry (side, cell index, rasterise)   src/visualization/cimg_debug.h:20-64
 *   LIDARNormalResidual / LIDARPointResidual functors     src/optimization/slam_residuals.h:64-177
 *   PointToLineResidual functor                           src/optimization/slam_residuals.h:179-216
 *   OdometryResidual functor                              src/optimization/slam_residuals.h:17-61
 *   ceres::CostFunction::Evaluate(parameters, residuals, jacobians) contract (Ceres 1.14):
 *       row-major num_residuals x 3 Jacobian per parameter block, NULL = not requested
 *   double pose[3] = (x, y, theta) parameter block        src/util/slam_types.h:159-178
 *
 * Conventions: every function returns 0 on success and a negative NHIP_ERR_* code on
 * failure (never aborts, never throws across the ABI); nhip_last_error() returns a
 * thread-local message.  "_dev" entry points take DEVICE pointers owned by the caller and
 * only enqueue work on `stream` (a hipStream_t passed as void*; NULL = default stream):
 * they allocate nothing and do not synchronise, so they can be captured in a hipGraph -- on a device that was named to
 * nhip_init (the current one) or nhip_set_device beforehand: those calls allocate the device's 16 status bytes (below).
 * A "_dev" call on a device the library never heard of allocates them itself, once (one hipMalloc + one synchronising
 * memset): do not let that be the first thing inside a stream capture.
 * Ids in device memory: the scan ids, grid slots, block ids and pose indices that a "_dev" entry point reads from DEVICE
 * arrays cannot be validated by the host without a round trip, so the KERNELS check each of them against the count
 * passed beside its array (n_scans, n_grids, n_blocks, n_poses ...).  An id outside [0, count) is never dereferenced: the
 * entry is treated as empty (a target without points, a pair that scores nothing, a correspondence that is skipped) and
 * recorded in the device's status words; nhip_dev_status(stream) -- call it where the host synchronises anyway --
 * returns NHIP_ERR_ARG with the offending id in nhip_last_error().  (The reference aborts on such input with a glog
 * CHECK, src/optimization/slam_residuals.h:99-101,109; a stale id here costs an error code, not the process.)
 * Handle entry points take HOST pointers, own their device memory and synchronise.
 * There is no CPU fallback anywhere: without a gfx950 device compute calls fail with
 * NHIP_ERR_NODEV.
 */
#ifndef NAUTILUS_HIP_H_
#define NAUTILUS_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NHIP_OK 0
#define NHIP_ERR_ARG (-1)    /* bad argument / shape mismatch (glog CHECK_* in the reference) */
#define NHIP_ERR_NODEV (-2)  /* no HIP device visible */
#define NHIP_ERR_HIP (-3)    /* a HIP runtime call failed */
#define NHIP_ERR_ALLOC (-4)  /* device or host allocation failed */
#define NHIP_ERR_STATE (-5)  /* handle used in the wrong state */

/* ------------------------------------------------------------------ general */
int nhip_init(int *n_devices);
int nhip_set_device(int device);
const char *nhip_last_error(void);
const char *nhip_version(void);
/* Synchronises `stream` (NULL = the default stream) and reports whether a kernel since the last call met an id in device
 * memory that was out of range (see "Ids in device memory" above): NHIP_OK, or NHIP_ERR_ARG with the message in
 * nhip_last_error().  info (may be NULL): {OR of the kinds seen, kind, value, index of the first one reported}; kinds:
 * 1 target scan id of a grid build, 2 source scan id of a pair, 4 grid slot of a pair, 8 block id of a correspondence,
 * 16 pose index of a block, 32 scan id of a correspondence-search block.  Clears the record (in the order of `stream`).
 * ONE record per DEVICE, shared by every stream and host thread that uses the library on it: a host with several streams
 * on one device learns THAT an id was bad and which, not on which stream; a call that finds a record consumes it -- reports
 * of kernels still running on OTHER streams at that moment can be reported by this call or wiped by its clear.  Clients
 * that need their errors apart must check on their own device or serialise launch + nhip_dev_status.  The index reported is
 * the entry's position in the CALLER's array (also for lists the matcher works in rounds). */
int nhip_dev_status(void *stream, int32_t info[4]);

/* ------------------------------------------------------------------ likelihood grids
 * Replaces the lookup table CorrelativeScanMatcher builds from the target point cloud.
 * Geometry follows cimg_debug.h:20-37; the likelihood model (integer separable Gaussian
 * blur of the hit raster, floor, natural log, 8- or 16-bit quantisation) is the build-defined
 * spec of DESIGN.md section 3. */
typedef struct nhip_grid_spec {
  double range;       /* scanner range [m]: ctor arg 1 (solver.cc:633); side = floor(2*range/res) */
  double res;         /* cell size [m] */
  double sigma;       /* blur sigma in cells */
  double floor_p;     /* likelihood floor before the log (1e-10) */
  int32_t max_shift;  /* largest |cell shift| a search on these grids may use */
  int32_t cell_bits;  /* width of a quantised log-likelihood cell: 16 (or 0: the default) or 8.  16-bit cells
                         (65535 steps of 3.5e-4 nat) keep reported scores within 1e-5 relative of an unquantised
                         double table (cimg_debug.h:19 holds the reference's table as CImg<double>); 8-bit cells
                         (255 steps of 0.09 nat, scores within 1e-3) are the explicit opt-in for callers that only
                         gate on a threshold */
  int32_t flags;      /* 0, NHIP_GRID_SKIP_MAP or NHIP_GRID_NO_IMAGE */
  int32_t reserved;   /* 0 */
} nhip_grid_spec_t;
/* NHIP_GRID_SKIP_MAP: the slots of 16-bit grids carry a skip map too (8-bit grids always do).  Only the kernel that
 * performs every add reads it (NHIP_SEARCH_EXHAUSTIVE, lattices beyond the branch-and-bound matcher's envelope);
 * without it that kernel adds every window strip, zero or not -- same records, ~1.5x the time.  The flag describes
 * the buffer: pass the same spec to the build and to the match.  The handle API builds missing maps itself the first
 * time an exhaustive search needs them. */
#define NHIP_GRID_SKIP_MAP 1
/* NHIP_GRID_NO_IMAGE: the slots carry no row-major image (layout.grid_bytes = 0, layout.skip_bytes = 0): 8.4 MB per
 * 1200 x 1200 grid of 16-bit cells instead of 12.3, a third less to clear per rebuild, a quarter less for the build to
 * store.  The branch-and-bound matcher on scans of at most NHIP_SHORT_SCAN_POINTS points -- the product path: every
 * 1081-beam workload -- reads only the pooled tables, its tiled planes and (NHIP_SEARCH_EXACT_SCORE) the hit raster; the
 * pooled tables are then built from the tiled copy of the cells.  What needs the image is refused on such grids with
 * NHIP_ERR_ARG: the kernels that perform every add (NHIP_SEARCH_EXHAUSTIVE, lattices beyond the matcher's envelope, score
 * volumes), lists that hold a longer scan (the caller must set NHIP_SEARCH_SHORT_SCANS; the handle API sets it itself),
 * nhip_grids_download.  Not combinable with NHIP_GRID_SKIP_MAP.  The flag describes the buffer: pass the same spec to
 * the build and to the match. */
#define NHIP_GRID_NO_IMAGE 2

typedef struct nhip_grid_layout {
  int32_t side;        /* S: cells per side (cimg_debug.h:21-22) */
  int32_t pad;         /* zero border on every side: 2*max_shift + 16, multiple of 4 */
  int32_t pitch;       /* bytes per stored row = (S + 2*pad) * cell_bytes rounded up to a multiple of 16 */
  int32_t rows;        /* stored rows = S + 2*pad; cell (row, col) is at byte (row+pad)*pitch + (col+pad)*cell_bytes */
  int32_t blur_radius; /* R = ceil(3*sigma) */
  int32_t cell_bytes;  /* 1 or 2 */
  int64_t tap_sum;     /* K = sum of the integer blur taps */
  int64_t grid_bytes;  /* pitch*rows: bytes of one stored grid's row-major image (0 with NHIP_GRID_NO_IMAGE) */
  double score_floor;  /* Lf = ln(floor_p): value of cell 0 */
  double score_step;   /* log-likelihood per quantisation step = -Lf/255 (8-bit cells) or -Lf/65535 (16-bit) */
  int64_t skip_bytes;  /* bytes of the skip map stored right after each image: one bit per stored row
                          r and aligned dword column c (bit c&7 of byte c>>3, 8*ceil(pitch/256) bytes per
                          row) = "stored rows [r, r+21) x dwords [c, c+21*cell_bytes) hold a non-zero cell",
                          so the correlation kernel can leave out window strips that only add zeros (same
                          sums, bit for bit).  Built for 8-bit cells always, for 16-bit cells when the spec carries
                          NHIP_GRID_SKIP_MAP (the branch-and-bound matcher never reads it); zero otherwise */
  int64_t slot_bytes;  /* grid_bytes + skip_bytes + pool_bytes + pool4_bytes + hi_bytes + hits_bytes: grid t of a buffer
                          starts at byte t*slot_bytes */
  int64_t pool_bytes;  /* bytes of the max-pooled table stored after the skip map (branch-and-bound bounds):
                          pool_rows x pool_pitch bytes, entry (i, j) = max of stored cells [8i, 8i+15) x [8j, 8j+15)
                          (16-bit cells: ceil(max / 257)), so that the sum of pooled entries bounds every score of
                          an 8 x 8 block of translations from above */
  int32_t pool_pitch;
  int32_t pool_rows;
  int64_t pool4_bytes; /* the second-level table, stored after the first: pool4_rows x pool4_pitch bytes.  With
                          P4[i][j] = max of stored cells [4i, 4i+7) x [4j, 4j+7) (16-bit cells: ceil(max / 257)), the
                          bounds of the 4 x 4 sub-blocks of a block of translations, byte (i, 2j) holds P4[i][j] and
                          byte (i, 2j+1) holds P4[i+1][j]: one read returns both sub-block rows */
  int32_t pool4_pitch;
  int32_t pool4_rows;
  int64_t hi_bytes;    /* after the second table, the matcher's 8-BIT PLANE: the cells' HIGH BYTES (cell >> 8; 8-bit cells: the cells),
                          hi_bytes bytes: two copies tiled 8 rows x 16 bytes, the second shifted by 8 columns (nhip_common.h
                          hi_tiled(); nhip_grids_download_hi_plane returns it as rows x hi_pitch), then (16-bit cells) the image once
                          more, tiled 8 rows x 8 cells (nhip_grids_download_tiled16).  The branch-and-bound matcher
                          takes its exact block sums on this plane at the cost of 8-bit cells -- 256*sum(high bytes) +
                          255*points bounds a pose's 16-bit sum from above -- and reads 16-bit cells for the few poses
                          whose bound still reaches the best sum: same records, bit for bit */
  int32_t hi_pitch;
  int32_t hits_pitch;  /* bytes per bit row of the hit raster (below) */
  int64_t hits_bytes;  /* after the matcher's planes, the HIT RASTER the table was blurred from: one bit per cell, rows
                          side + 64 (a zero border of 32 cells on every side), hits_pitch bytes each; bit (row, col) is bit
                          (col + 32) & 31 of dword (col + 32) >> 5 of bit row row + 32.  NHIP_SEARCH_EXACT_SCORE reads it */
} nhip_grid_layout_t;

/* Pure host helpers (work without a GPU). */
int nhip_grid_layout(const nhip_grid_spec_t *spec, nhip_grid_layout_t *out);
/* bytes the caller must allocate for n grids (n*slot_bytes + 256 B read slack) */
int64_t nhip_grids_bytes(const nhip_grid_spec_t *spec, int64_t n_grids);
/* workspace for nhip_grid_build_dev processing `chunk` targets at a time: per target one occupancy byte, one list slot and
 * 20 bytes of line masks per 64 x 64 tile (what nhip_grid_rebuild_dev clears is read from there), + the 16-bit quantiser's table */
int64_t nhip_grid_workspace_bytes(const nhip_grid_spec_t *spec, int32_t chunk);
/* integer blur taps (2R+1 values) and the quantiser threshold table: 256 entries for 8-bit cells,
 * 65536 for 16-bit cells (thresholds[k] = smallest integer blur sum whose quantised value is >= k) */
int nhip_grid_tables(const nhip_grid_spec_t *spec, int32_t *taps, uint32_t *thresholds);

/* ------------------------------------------------------------------ search lattice
 * BASELINE config #2: n_theta = 61, theta_step = 1 deg, nx = ny = 81 (one cell = 5 cm).
 * Translations are integer cell shifts ix-(nx-1)/2, iy-(ny-1)/2; rotation k is
 * theta0 + (k-(n_theta-1)/2)*theta_step with theta0 = AngleMod(rot_a - rot_b)
 * (math_util.h:81-89), i.e. the odometry estimate of "A to B in B's frame"
 * (solver.cc:631-638). */
typedef struct nhip_search {
  int32_t n_theta;
  int32_t nx;
  int32_t ny;
  int32_t flags;     /* 0, or an OR of NHIP_SEARCH_* */
  double theta_step; /* radians */
} nhip_search_t;
/* The matcher's result is that of the exhaustive (theta, x, y) search, always.  By default it gets there by
 * branch and bound: upper bounds of every 8 x 8 block of translations from a max-pooled copy of the table, bounds
 * of the 4 x 4 sub-blocks of the blocks that reach the best sum found from a second one, then exact sums only for
 * the sub-blocks whose bound still reaches it (indices, sums and scores are identical to the exhaustive kernel's,
 * bit for bit: tests compare the two).  NHIP_SEARCH_EXHAUSTIVE (or, in a process started with NHIP_TUNABLES=1, the environment variable NHIP_CSM_EXHAUSTIVE=1)
 * forces the kernel that performs every add (csm_correlate_kernel for 8-bit, csm_correlate16_kernel for 16-bit cells;
 * also taken for lattices of more than 88 x 88 translations, or more rotations than fit the LDS beside the bounds:
 * ~230 -- e.g. GetTransformation with a rotation restriction of pi).
 * Scan length: sums are reported as int32, so with 16-bit cells a scan may hold at most 32,768 points (8-bit:
 * 8,421,504); the handle API checks it, callers of the _dev entry points must. */
#define NHIP_SEARCH_EXHAUSTIVE 1
/* NHIP_SEARCH_DENSE (with the kernel that performs every add): add every window strip, the all-zero ones the skip map
 * would leave out too -- literally every add of the exhaustive definition; same records, ~1.5x the time. */
#define NHIP_SEARCH_DENSE 2
/* NHIP_SEARCH_SHORT_SCANS: the caller vouches that every source scan of the list has at most NHIP_SHORT_SCAN_POINTS
 * points (a 1081-beam scan does).  The branch-and-bound matcher serves longer scans with a second instantiation of its
 * kernel, launched beside the first over all pairs (its workgroups return at once when their pair is not theirs); with
 * this flag it is not launched.  A PROMISE: pairs whose scan is longer are then not matched at all (their records are
 * whatever d_keys held).  The handle API (nhip_csm_match) knows the lengths and sets the flag itself. */
#define NHIP_SEARCH_SHORT_SCANS 4
#define NHIP_SHORT_SCAN_POINTS 1088
/* NHIP_SEARCH_EXACT_SCORE: the record's `score` is the winning pose's mean log-likelihood on the UNQUANTISED table -- what a
 * table of doubles (the reference's CImg<double>, cimg_debug.h:19) gives at that pose -- instead of Lf + step * sum / N on the
 * quantised cells.  Indices and integer sums are unchanged (the search itself runs on the quantised table); one more kernel
 * recomputes, for the one pose that won, the exact integer blur sum of each cell a point reads from the slot's hit raster
 * and its logarithm in double (~0.05 ms per 10,000 pairs).  Without the flag a reported score is within half a
 * quantisation step of that value per cell -- measured up to 2.3e-5 relative on 1,300 pairs with 16-bit cells, 8.5e-4 with
 * 8-bit cells; with it, within double rounding (the record's float: 6e-8). */
#define NHIP_SEARCH_EXACT_SCORE 8
/* NHIP_SEARCH_LATENCY (with NHIP_SEARCH_EXHAUSTIVE): the list is a FEW pairs and what counts is how soon the call returns, not
 * lookups per second.  Every add is then performed by the kernel whose lanes are poses (csm_small_plane_kernel), a plane of
 * more than 256 translations in tiles of whole rows, one workgroup per (pair, rotation, tile) -- as long as that is at most
 * 2,048 workgroups (otherwise the flag is ignored and the strip kernels take the list).  GetTransformation's fine level
 * (21 x 61 x 61 on a 6000 x 6000 table) takes ~45 us this way whatever the clouds.  Same records as every other form. */
#define NHIP_SEARCH_LATENCY 16

/* One result per candidate pair: 16 bytes, the record that is all-gathered across GPUs. */
typedef struct nhip_match {
  int32_t itheta;
  int32_t ix;
  int32_t iy;
  float score; /* mean log-likelihood (<= 0), cf. csm_score_threshold = -5 (default_config.lua:85) */
} nhip_match_t;

/* Device memory of the handle API.  Handles own their device buffers; a buffer a handle lets go of (nhip_*_free, the
 * workspace and scratch of nhip_csm_match, ...) is KEPT for the next allocation of the same device that it fits (at most
 * twice the size asked for) instead of going back to the driver: hipFree is a device-wide synchronisation whose cost is not
 * the library's to bound (measured: 0.2 - 8 ms per call on a quiet device, 0.33 s per call for seconds after another client
 * of the process released 130 GB).  A buffer is filed under the device it was ALLOCATED on, whatever
 * device is current when its handle is freed.  At most max_bytes are kept PER DEVICE (default 4 GB: the workspace of a
 * 10,000-pair match, the tables of a few hundred targets; least recently released out first; 0 = nothing is kept, every
 * release is a hipFree); a host that cycles larger tables through build / free raises it (bench.py's host-buffer leg: 32
 * GB).  What the pool holds is invisible to other allocators in the process (torch's caching allocator): call
 * nhip_device_pool_release() -- everything back to the driver now -- before handing the device's memory to them; _stats
 * reports what is held (all devices).  A failed allocation releases the device's pool and tries again.  On an error path a
 * handle call waits for the device before its buffers return to the pool.
 * The "_dev" entry points never allocate and are not concerned. */
int nhip_device_pool_configure(int64_t max_bytes);
int nhip_device_pool_release(void);
int nhip_device_pool_stats(int64_t *entries, int64_t *bytes);

/* Wall-clock seconds of the calling thread's last handle-API call (nhip_scans_upload, nhip_grids_build, nhip_csm_match and
 * the _free calls), by what the host waited for: out[0] hipMalloc, [1] zero-fill + host-to-device copies, [2] from the first
 * kernel launch to the end of the call (nhip_grids_build: the launches alone), [3] waiting for the kernels, [4] device-to-host
 * copies, [5] hipFree, [6] the whole call (nhip_csm_match), [7] unused.  Diagnostic: bench.py prints it per run of its
 * host-buffer leg, so that a slow call says where it was slow. */
int nhip_host_phases(double out[8]);

/* Host helpers: (cos, sin) of theta0 per pair and of the lattice offsets, in double. */
int nhip_csm_rot0(const double *rot_a, const double *rot_b, int32_t n, double *cs_out /* 2n */);
int nhip_csm_delta_table(const nhip_search_t *search, double *cs_out /* 2*n_theta */);
/* Host helper: (tx, ty, theta) of a match, as consumed at solver.cc:640-644. */
int nhip_match_to_transform(const nhip_match_t *m, const nhip_grid_spec_t *spec,
                            const nhip_search_t *search, double theta0, int32_t origin_x,
                            int32_t origin_y, float *tx, float *ty, float *theta);
double nhip_score_from_sum(const nhip_grid_spec_t *spec, int64_t sum, int32_t n_points);

/* ------------------------------------------------------------------ device-pointer API */
/* K1: build likelihood grids for n_targets scans (scan ids in d_target_ids) into
 * d_grids (nhip_grids_bytes(spec, n_targets) bytes): slot t = position in d_target_ids, at byte
 * t * slot_bytes, holds the stored image (grid_bytes) followed by its skip map (skip_bytes) and the two
 * max-pooled tables (pool_bytes, pool4_bytes).
 * d_xy: float2 points of all scans, d_offsets: n_scans+1 prefix offsets (in points).
 * d_target_ids live in device memory: the kernels check every id against n_scans; one outside [0, n_scans) gives an
 * all-floor grid (a target without points) and an error from nhip_dev_status().  The offsets themselves are the
 * caller's: they must be non-decreasing and end inside d_xy. */
int nhip_grid_build_dev(const float *d_xy, const int32_t *d_offsets, int32_t n_scans, const int32_t *d_target_ids,
                        int32_t n_targets, const nhip_grid_spec_t *spec, uint8_t *d_grids,
                        void *d_workspace, int64_t workspace_bytes, void *stream);

/* The same build INTO A BUFFER THE PREVIOUS BUILD FILLED: instead of zero-filling n_targets * slot_bytes (gigabytes at
 * 1000 targets) it clears the ~20 % of 64 x 64 tiles the previous build wrote -- their list is still in the workspace
 * -- and the small derived tables.  Contract: same d_workspace as the build that last wrote d_grids, and d_grids
 * untouched since.  The workspace header carries a tag of (d_grids, n_targets, geometry) that the clearing kernel checks
 * on the device: a header that does not vouch for this buffer (fresh or recycled workspace memory, another buffer,
 * another spec) makes the call clear everything, i.e. behave as nhip_grid_build_dev.  Same results, bit for bit. */
int nhip_grid_rebuild_dev(const float *d_xy, const int32_t *d_offsets, int32_t n_scans, const int32_t *d_target_ids,
                          int32_t n_targets, const nhip_grid_spec_t *spec, uint8_t *d_grids,
                          void *d_workspace, int64_t workspace_bytes, void *stream);

/* K2+K3: exhaustive (theta, x, y) correlation + argmax for n_pairs candidate pairs.
 * Pair i matches scan d_pair_src[i] (of the n_scans behind d_offsets) against grid slot d_pair_slot[i] (of the n_grids in
 * d_grids).  Both arrays live in device memory: a pair whose scan or slot is out of range scores nothing (record: pose 0,
 * sum 0, the floor score) and makes nhip_dev_status() return an error.
 * d_rot0_cs: 2 doubles (cos, sin theta0) per pair; d_delta_cs: 2 doubles per lattice rotation.
 * d_pair_origin: NULL, or 2 int32 per pair = (x, y) cell offset of the search centre (used by
 * the fine level of a coarse-to-fine search); |origin| + half-width must be <= max_shift.
 * d_keys: n_pairs uint64 scratch; d_out: n_pairs records; d_sums: n_pairs int32 or NULL.
 * d_workspace: NULL, or nhip_csm_workspace_bytes(n_pairs) bytes of scratch for the branch-and-bound matcher.  Lists of
 * fewer than 192 pairs use it for hand-over lists: a pair whose landscape is flat (hundreds of candidate blocks after
 * bounds and seeds; in lists of <= 64 pairs, e.g. a single GetTransformation call, every pair) hands all but its first
 * 8 rotations to a second kernel that works them with every wave of the chip instead of keeping its one workgroup busy
 * for milliseconds.  Lists of 192 pairs and more run as two kernels -- bounds + seeds, whose workgroups all take the
 * same time, then the candidates of all pairs, the heaviest pairs shared by several workgroups -- with each pair's rows
 * of bounds parked in the workspace in between (nhip_csm_workspace_bytes asks for 32 KB per pair there, for at most
 * 262,144 pairs: 328 MB at 10,000 pairs; 10,000 pairs: 8.2 -> 6.3 ms, 3,000 pairs: 4.1 -> 2.6 ms).  Lists of more than
 * 131,072 pairs go through in rounds of that many, the candidates of a round on an internal stream of the current
 * device (ordered before and after `stream` by events) beside the next round's bounds.  A workspace that is NULL or too
 * small takes the one-kernel form.  Same records in every form. */
int nhip_csm_match_dev(const float *d_xy, const int32_t *d_offsets, int32_t n_scans, const uint8_t *d_grids,
                       int32_t n_grids, const nhip_grid_spec_t *spec, const int32_t *d_pair_src,
                       const int32_t *d_pair_slot, const double *d_rot0_cs,
                       const double *d_delta_cs, const int32_t *d_pair_origin, int32_t n_pairs,
                       const nhip_search_t *search, uint64_t *d_keys, nhip_match_t *d_out,
                       int32_t *d_sums, void *d_workspace, int64_t workspace_bytes, void *stream);
int64_t nhip_csm_workspace_bytes(int32_t n_pairs);
/* The form the calling thread's last branch-and-bound match took (diagnostic; every form returns the same records):
 * out[0] = 0 one kernel per pair from start to end (+ the hand-over kernel when out[5]), 1 split form in one round,
 * 2 split form in several rounds on the caller's stream, 3 split form in rounds with the candidates on the library's
 * helper stream of the current device; out[1] pairs per round; out[2] rounds' state the workspace holds; out[3] rounds;
 * out[4] 1 when NHIP_SEARCH_SHORT_SCANS was honoured; out[5] hand-over kernel launched; out[6] instrumented build;
 * out[7] n_pairs. */
int nhip_csm_last_launch(int32_t out[8]);

/* With NHIP_BNB_STATS=1 in the environment the branch-and-bound matcher counts its work: blocks of 8 x 8
 * translations whose sums it evaluated exactly (four 4 x 4 sub-blocks count as one block), and blocks in all, since
 * the last call (synchronises; resets). */
int nhip_bnb_stats(uint64_t *evaluated, uint64_t *total);
/* ... by level: out[0..3] = {blocks evaluated whole, blocks in all, candidate blocks refined through their four
 * sub-block bounds, 4 x 4 sub-blocks evaluated exactly}; out[4..10] = shader-clock sums of the matcher's kernel:
 * wave time in the candidate phase, of which window origins / sub-block bounds / exact sums, the slowest wave of
 * each pair, seed phase and bound phase (per workgroup); out[11..13] further clocks, out[14] = poses of 16-bit grids whose exact
 * sums were read from the 16-bit image (the rest was settled on the plane of high bytes); (synchronises; resets) */
int nhip_bnb_stats_levels(uint64_t out[16]);
/* ... and per pair of the last launch (4 x 4 sub-blocks evaluated exactly, a whole block counting four), before
 * nhip_bnb_stats resets the totals */
int nhip_bnb_stats_per_pair(uint64_t *evaluated, int32_t n_pairs);
/* With NHIP_BNB_TIMELINE=1: per pair of the last launch four 100 MHz timestamps of its workgroup -- start, bounds
 * done, seeds done, end (the top 16 bits of the last one hold the hardware id of the CU it ran on); after the
 * n_pairs records two more values: first start and last end of the second kernel.  ticks: 4*n_pairs + 2 values */
int nhip_bnb_timeline(uint64_t *ticks, int32_t n_pairs);
/* ... and of the candidates' launch of the split form: ticks[i] = first start, ticks[n_pairs + i] = last end over the
 * workgroups that worked pair i of the last round (~0 / 0: none did).  ticks: 2*n_pairs values */
int nhip_bnb_timeline_candidates(uint64_t *ticks, int32_t n_pairs);

/* Full score volume of ONE pair (tests / debugging): sums[(k*nx + ix)*ny + iy]. */
int nhip_csm_scores_dev(const float *d_xy, const int32_t *d_offsets, int32_t n_scans, const uint8_t *d_grids,
                        int32_t n_grids, const nhip_grid_spec_t *spec, int32_t src, int32_t slot,
                        const double *d_rot0_cs, const double *d_delta_cs, int32_t origin_x,
                        int32_t origin_y, const nhip_search_t *search, int32_t *d_sums,
                        void *stream);

/* K4: batched LIDAR residual + Jacobian evaluation.
 * kind: NHIP_LIDAR_NORMAL (slam_residuals.h:65-89) or NHIP_LIDAR_POINT (:124-145).
 * d_corr: 8 floats per correspondence (source point, target point, source normal, target
 * normal -- the four vectors of PointCorrespondences, data_structures.h:62-66).
 * d_corr_block: block id of every correspondence; block b reads poses d_block_src[b]
 * ("source_pose", parameter block 0) and d_block_tgt[b] ("target_pose", block 1) from
 * d_poses[n_poses][3].  d_block_consts: n_blocks*8 doubles scratch.
 * Outputs: d_residuals 2 doubles / correspondence; d_jac_src, d_jac_tgt 6 doubles /
 * correspondence (rows 2i, 2i+1 of the row-major (2N x 3) Jacobian); either may be NULL. */
#define NHIP_LIDAR_NORMAL 0
#define NHIP_LIDAR_POINT 1
int nhip_resid_lidar_dev(int kind, const float *d_corr, const int32_t *d_corr_block,
                         int64_t n_corr, const int32_t *d_block_src, const int32_t *d_block_tgt,
                         int32_t n_blocks, const double *d_poses, int32_t n_poses,
                         double *d_block_consts, double *d_residuals, double *d_jac_src,
                         double *d_jac_tgt, void *stream);

/* On-device normal equations of the LIDAR blocks: instead of
 * shipping 96 B of Jacobian per correspondence to the host, each block is reduced to what a
 * Gauss-Newton / Levenberg-Marquardt step needs.  With J = [J_src | J_tgt] (2N x 6) and r (2N):
 *   d_out[28*b +  0..20] = upper triangle of J^T J, row-major (i <= j)
 *   d_out[28*b + 21..26] = J^T r
 *   d_out[28*b + 27]     = r^T r
 * Same functors, same closed-form Jacobians as nhip_resid_lidar_dev; d_block_offsets has
 * n_blocks+1 entries (rows of d_corr per block). */
int nhip_resid_lidar_normal_eq_dev(int kind, const float *d_corr, const int32_t *d_block_offsets,
                                   const int32_t *d_block_src, const int32_t *d_block_tgt,
                                   int32_t n_blocks, const double *d_poses, int32_t n_poses,
                                   double *d_block_consts, double *d_out, void *stream);

/* PointToLineResidual (slam_residuals.h:180-200): block b has line segment d_segments[4b..]
 * (x0 y0 x1 y1, LineSegment<float>), points d_points[2i..] with block id d_point_block[i],
 * parameter blocks pose = d_poses[d_block_pose[b]] and line_pose = d_line_poses[d_block_line[b]].
 * Outputs: 1 residual / point, 3 doubles / point per Jacobian (may be NULL). */
int nhip_resid_point_to_line_dev(const float *d_segments, const float *d_points,
                                 const int32_t *d_point_block, int64_t n_points,
                                 const int32_t *d_block_pose, const int32_t *d_block_line,
                                 int32_t n_blocks, const double *d_poses, int32_t n_poses,
                                 const double *d_line_poses, int32_t n_line_poses, double *d_residuals,
                                 double *d_jac_pose, double *d_jac_line, void *stream);

/* OdometryResidual (slam_residuals.h:18-40): factor f has T_odom d_t_odom[2f..] (Vector2f),
 * R_odom d_r_odom[f] (float), poses d_pose_i[f], d_pose_j[f].  3 residuals, 3x3 Jacobians. */
int nhip_resid_odometry_dev(const float *d_t_odom, const float *d_r_odom, const int32_t *d_pose_i,
                            const int32_t *d_pose_j, int32_t n_factors, double translation_weight,
                            double rotation_weight, const double *d_poses, int32_t n_poses, double *d_residuals,
                            double *d_jac_i, double *d_jac_j, void *stream);

/* K5: correspondence search, the step that feeds K4 (Solver::GetPointToPointMatching,
 * src/optimization/solver.cc:132-172; KDTree::FindNearestPoint, src/util/kdtree.cc:253-305).
 * Block b pairs source scan d_block_src[b] with target scan d_block_tgt[b]: every source point is
 * moved into the target frame by inverse(T_target) * T_source (float affines of the poses, see
 * nhip_pose_affines), matched with its nearest target point and kept if the distance is below
 * outlier_threshold (default_config.lua:66).  d_normals: one float2 per point of d_xy (the
 * reference reads normals from its trees, solver.cc:67-78).  Rows (8 floats: source point, target
 * point, source normal, target normal) are written in source order at
 * d_corr_padded + 8*d_cap_offsets[b] (capacity = source points of block b); d_counts[b] rows are
 * valid.  nhip_corr_compact_dev packs them into the contiguous layout nhip_resid_lidar_dev takes
 * (d_block_offsets: n_blocks+1, d_corr: 8 floats/row, d_corr_block: block id per row). */
int nhip_pose_affines(const double *poses, int32_t n, float *out /* 4n: cos sin x y */);
int nhip_corr_search_dev(const float *d_xy, const float *d_normals, const int32_t *d_offsets, int32_t n_scans,
                         const int32_t *d_block_src, const int32_t *d_block_tgt, int32_t n_blocks,
                         const float *d_pose_aff, float outlier_threshold,
                         const int64_t *d_cap_offsets, float *d_corr_padded, int32_t *d_counts,
                         void *stream);
/* The same search with the normal gate of Solver::GetPointToNormalMatching /
 * FindClosestPointWithSimilarNormal (solver.cc:177-260; defined but not called at this commit): among
 * the target points within outlier_threshold, the nearest whose normal satisfies
 * |n_target . n_source| > min_abs_cosine (NormalsSimilar, math_util.h:46-49; the reference passes
 * cos(20 deg)); n_source is the source point's normal in the SOURCE frame, as in the reference.
 * Exactly equal distances go to the lowest target index (the reference's std::sort leaves that open). */
int nhip_corr_search_normals_dev(const float *d_xy, const float *d_normals, const int32_t *d_offsets, int32_t n_scans,
                                 const int32_t *d_block_src, const int32_t *d_block_tgt, int32_t n_blocks,
                                 const float *d_pose_aff, float outlier_threshold, float min_abs_cosine,
                                 const int64_t *d_cap_offsets, float *d_corr_padded, int32_t *d_counts,
                                 void *stream);
int nhip_corr_compact_dev(const float *d_corr_padded, const int64_t *d_cap_offsets,
                          const int32_t *d_counts, int32_t n_blocks, int32_t *d_block_offsets,
                          float *d_corr, int32_t *d_corr_block, void *stream);

/* Loop-closure candidate gating, the step before the matcher.
 * nhip_lc_scatter_scores*: LCCandidateFilter's ComputeScatterMatrixScore (lc_candidate_filter.cc:22-51) for every
 * scan: float mean and scatter matrix summed in point order (the reference's float sums, bit for bit), min / max
 * eigenvalue (closed form in double); one double per scan, NaN for an empty scan.
 * nhip_lc_pair_gate*: for n candidate nodes (indices into poses[n_poses][3]), flags[i*n + j] = 1 iff candidates i and
 * j are different nodes more than min_separation apart in index whose translations (as Vector2f) are closer than
 * max_range -- a geometric stand-in for LCMatcher's per-pair ceres::Covariance test (lc_matcher.cc:28-74).
 * nhip_lc_chi_square_gate*: LCMatcher's own test given the covariance blocks -- for pair i, d = Vector2f(pose of
 * pair_tgt[i]) - Vector2f(pose of pair_src[i]), scores[i] = d^T cov_i^-1 d evaluated in float as ChiSquareScore does
 * (lc_matcher.cc:50-57; cov[i] = the Matrix2f of GetCovarianceMatrix, :28-46, row major m00 m01 m10 m11), widened to
 * double; flags[i] = 1 iff pair_src[i] != pair_tgt[i] and scores[i] < max_score (GetPossibleMatches, :59-74, which
 * passes 5000.0).  A singular block gives inf / NaN scores as the reference's inverse() does, and flag 0 for NaN.
 * d_cov must be 16-byte aligned.  The covariance solve itself (ceres::Covariance) stays with the host's solver. */
int nhip_lc_scatter_scores_dev(const float *d_xy, const int32_t *d_offsets, int32_t n_scans, double *d_scores,
                               void *stream);
int nhip_lc_pair_gate_dev(const double *d_poses, int32_t n_poses, const int32_t *d_candidates, int32_t n_candidates,
                          double max_range, int32_t min_separation, uint8_t *d_flags, void *stream);
int nhip_lc_chi_square_gate_dev(const double *d_poses, int32_t n_poses, const int32_t *d_pair_src, const int32_t *d_pair_tgt,
                                const float *d_cov, int32_t n_pairs, double max_score, double *d_scores,
                                uint8_t *d_flags, void *stream);

/* ------------------------------------------------------------------ handle API (host pointers) */
typedef struct nhip_scans nhip_scans_t;
typedef struct nhip_grids nhip_grids_t;
typedef struct nhip_resid_batch nhip_resid_batch_t;

/* std::vector<Eigen::Vector2f> point clouds of n_scans scans, concatenated. */
int nhip_scans_upload(const float *xy, const int32_t *offsets, int32_t n_scans, nhip_scans_t **out);
int nhip_scans_free(nhip_scans_t *scans);

int nhip_grids_build(const nhip_scans_t *scans, const int32_t *target_ids, int32_t n_targets,
                     const nhip_grid_spec_t *spec, nhip_grids_t **out);
int nhip_grids_free(nhip_grids_t *grids);
/* 1 when this handle's tables were built by an incremental REBUILD: nhip_grids_free hands the table buffer and its build
 * workspace back to the device buffer pool together, contents known; while nobody else has taken either, the next
 * nhip_grids_build of the same spec and target count takes the pair and clears what the previous build wrote (as
 * nhip_grid_rebuild_dev does, the workspace's tag checked on the device) instead of zero-filling every slot.  Same
 * tables, bit for bit.  0: a fresh allocation or a buffer of unknown contents, zero-filled (diagnostic). */
int nhip_grids_was_rebuilt(const nhip_grids_t *grids);
/* copy stored (padded) grid `slot` to host: layout.grid_bytes bytes (uint8 or uint16 cells) */
int nhip_grids_download(const nhip_grids_t *grids, int32_t slot, uint8_t *out);
/* copy the plane of high bytes of grid `slot` (16-bit cells) to host in plain row-major form: rows x hi_pitch bytes.
 * On the device the plane is stored as two copies tiled 8 rows x 16 bytes (layout.hi_bytes bytes in all; the second
 * copy's tiles are shifted by 8 columns); `_copy` selects the copy that is read back (0 / 1: both hold the same bytes). */
int nhip_grids_download_hi_plane(const nhip_grids_t *grids, int32_t slot, uint8_t *out);
int nhip_grids_download_hi_plane_copy(const nhip_grids_t *grids, int32_t slot, int32_t copy, uint8_t *out);
/* 16-bit cells: the matcher's copy of the 16-bit image (tiled 8 rows x 8 cells on the device; it follows the two copies
 * of the plane of high bytes inside layout.hi_bytes) in the plain form of nhip_grids_download: layout.grid_bytes bytes */
int nhip_grids_download_tiled16(const nhip_grids_t *grids, int32_t slot, uint8_t *out);
/* copy the skip map of grid `slot` to host: layout.skip_bytes bytes (rows x 8*ceil(pitch/256) bytes, then padding) */
int nhip_grids_download_skip_map(const nhip_grids_t *grids, int32_t slot, uint8_t *out);
/* copy the max-pooled table of grid `slot` to host: layout.pool_bytes bytes (pool_rows x pool_pitch) */
int nhip_grids_download_pool(const nhip_grids_t *grids, int32_t slot, uint8_t *out);
/* the same for the second-level table: layout.pool4_bytes bytes (pool4_rows x pool4_pitch) */
int nhip_grids_download_pool4(const nhip_grids_t *grids, int32_t slot, uint8_t *out);
/* the hit raster of grid `slot`: layout.hits_bytes bytes (bit rows of hits_pitch bytes; see nhip_grid_layout_t.hits_bytes) */
int nhip_grids_download_hits(const nhip_grids_t *grids, int32_t slot, uint8_t *out);

/* Batched GetTransformation: theta0[i] = AngleMod(rot_a - rot_b) of pair i;
 * pair_origin: NULL or 2 int32 per pair (search centre in cells). */
int nhip_csm_match(const nhip_scans_t *scans, const nhip_grids_t *grids, const int32_t *pair_src,
                   const int32_t *pair_slot, const double *theta0, const int32_t *pair_origin,
                   int32_t n_pairs, const nhip_search_t *search, nhip_match_t *out,
                   int32_t *out_sums);
int nhip_csm_scores(const nhip_scans_t *scans, const nhip_grids_t *grids, int32_t src, int32_t slot,
                    double theta0, int32_t origin_x, int32_t origin_y, const nhip_search_t *search,
                    int32_t *out_sums);

/* Host-pointer forms of the candidate gating. */
int nhip_lc_scatter_scores(const nhip_scans_t *scans, double *scores /* n_scans */);
int nhip_lc_pair_gate(const double *poses, int32_t n_poses, const int32_t *candidates, int32_t n_candidates,
                      double max_range, int32_t min_separation, uint8_t *flags /* n_candidates^2 */);
int nhip_lc_chi_square_gate(const double *poses, int32_t n_poses, const int32_t *pair_src, const int32_t *pair_tgt,
                            const float *cov /* n_pairs x 4 */, int32_t n_pairs, double max_score,
                            double *scores /* n_pairs */, uint8_t *flags /* n_pairs */);

/* The reference-shaped single-pair call: CorrelativeScanMatcher(scanner_range, trans_range, low_res, high_res)
 * .GetTransformation(pc_a, pc_b, rot_a, rot_b, rot_restriction) -> (score, ((tx, ty), theta))
 * (solver.cc:56, 633-644; the class lives in the absent third_party/csm).  Build-defined search:
 * exhaustive on the low_res grid over +-trans_range and +-rot_restriction in 1 degree steps, then exhaustive on the
 * high_res grid over +-low_res around the coarse optimum in 0.1 degree steps; the returned score is the fine optimum's
 * mean log-likelihood on the unquantised table (NHIP_SEARCH_EXACT_SCORE).  This is the ONE implementation of that
 * search: the C++ drop-in (adapters/CorrelativeScanMatcher.h) and the Python mirror (nautilus_amd/csm.py) both call it;
 * oracle/csm_oracle.c restates it independently for the tests.  Host pointers.  pc_a / pc_b: n x 2 floats
 * (std::vector<Eigen::Vector2f>).  The two tables built from pc_b stay in a cache of the last targets (keyed by the cloud's
 * bytes and the parameters, per device; least recently used out first under a byte cap, 3 GB by default -- one target of
 * the (30, 2, 0.3, 0.01) matcher is ~0.3 GB): SolveAutoLC -> GetRelativeTransform (solver.cc:630-649, 676-700) matches many
 * sources against one target in a row, and only the first of those calls pays the build.  Thread-safe. */
typedef struct nhip_csm_params {
  double scanner_range; /* ctor arg 1 (30) */
  double trans_range;   /* ctor arg 2 (2) */
  double low_res;       /* ctor arg 3 (0.3) */
  double high_res;      /* ctor arg 4 (0.01) */
  double sigma;         /* blur sigma in cells (2.0) */
  double floor_p;       /* likelihood floor (1e-10) */
  int32_t cell_bits;    /* 16 or 8 (0 = 16, as in nhip_grid_spec_t: scores within 1e-5 of an unquantised table) */
  int32_t reserved;
} nhip_csm_params_t;
/* The cache of nhip_csm_get_transformation: its byte cap (0 = keep nothing; entries beyond the new cap are freed), all
 * entries freed, and counters {entries, bytes held, hits, misses since the process started} (any pointer may be NULL). */
int nhip_csm_cache_configure(int64_t max_bytes);
int nhip_csm_cache_clear(void);
int nhip_csm_cache_stats(int64_t *entries, int64_t *bytes, int64_t *hits, int64_t *misses);
int nhip_csm_get_transformation(const nhip_csm_params_t *params, const float *pc_a, int32_t n_a, const float *pc_b,
                                int32_t n_b, double rot_a, double rot_b, double rot_restriction, double *score,
                                float *tx, float *ty, float *theta);
/* What the calling thread's last cached-target nhip_csm_get_transformation did: {the coarse optimum's score, the fine level's
 * form (0: the branch-and-bound matcher, 1: every add by the strip kernels, 2: every add by the kernel whose lanes are poses --
 * the default), 1 if the two levels were chained on the device, the coarse optimum's rotation index}.  (Measurement / tests.) */
int nhip_csm_get_transformation_info(double out[4]);

/* Residual batch: all LIDAR residual blocks of one ceres::Problem build (immutable after
 * creation, like the functors' copied vectors, slam_residuals.h:117-120). */
int nhip_resid_batch_create(int kind, const float *corr, const int32_t *block_offsets,
                            const int32_t *block_src, const int32_t *block_tgt, int32_t n_blocks,
                            int32_t n_poses, nhip_resid_batch_t **out);
/* poses: n_poses*3 doubles.  residuals: 2*n_corr; jac_src / jac_tgt: 6*n_corr or NULL. */
int nhip_resid_batch_eval(nhip_resid_batch_t *batch, const double *poses, double *residuals,
                          double *jac_src, double *jac_tgt);
/* The same evaluation with the target Jacobian in compact form: jac_tgt_theta holds 2 doubles per correspondence,
 * the theta column of its two rows.  The x and y columns of J_tgt are exactly the negated x and y columns of J_src
 * (dq/dt_t = -dq/dt_s), so a consumer rebuilds row i as (-J_src[i][0], -J_src[i][1], jac_tgt_theta[i]) while it
 * copies its slice: 80 instead of 112 bytes per correspondence cross PCIe. */
int nhip_resid_batch_eval_compact(nhip_resid_batch_t *batch, const double *poses, double *residuals,
                                  double *jac_src, double *jac_tgt_theta);
/* The same evaluation in its SMALLEST form over PCIe: per correspondence the residual pair and q = S2T p_s (the source point
 * in the target's frame, slam_residuals.h:78), per block the 8 constants {S2T's linear part l00 l01 l10 l11, its
 * translation tx ty, and i00 i01 of inverse(A(target_pose))'s linear part (i10 = -i01, i11 = i00)}.  Every Jacobian entry
 * ceres::AutoDiffCostFunction derives for these functors (slam_residuals.h:104-115, 160-171) is a closed form of q, those
 * constants and the correspondence's own points / normals, which the host holds: a consumer rebuilds the
 * rows of its block while it copies its slice -- nhip_resid_jacobians_from_q does exactly that, on the host, for `n` rows
 * of ONE block (corr: its n x 8 floats; q: its n x 2 doubles; block_consts: its 8 doubles; jac_src / jac_tgt: 6 n doubles
 * or NULL).  32 instead of 80 (compact) / 112 (full) bytes per correspondence cross PCIe; the Jacobians agree with
 * nhip_resid_batch_eval's to 1e-14 relative (u is recovered as q - t). */
int nhip_resid_batch_eval_q(nhip_resid_batch_t *batch, const double *poses, double *residuals, double *q, double *block_consts);
int nhip_resid_jacobians_from_q(int kind, const float *corr, const double *q, const double *block_consts, int64_t n,
                                double *jac_src, double *jac_tgt);
/* ONE block of the batch at explicitly given parameter blocks (source_pose[3], target_pose[3]): what a
 * ceres::CostFunction::Evaluate() called outside the batched evaluation point needs (Problem::Evaluate,
 * Covariance::Compute, a rejected trial step).  residuals: 2 n_b doubles; jac_src / jac_tgt: n_b x 2 rows of 3
 * doubles, either may be NULL. */
int nhip_resid_batch_eval_block(nhip_resid_batch_t *batch, int32_t block, const double *source_pose,
                                const double *target_pose, double *residuals, double *jac_src, double *jac_tgt);
int nhip_resid_batch_free(nhip_resid_batch_t *batch);

/* Page-locked host memory for buffers that cross PCIe every evaluation (device-to-host copies into pinned memory
 * run at the link rate; into pageable memory at about half of it). */
int nhip_host_alloc(size_t bytes, void **out);
int nhip_host_free(void *p);

/* Host-pointer forms of the two small functor families (OdometryResidual, slam_residuals.h:18-40:
 * one block per consecutive pose pair, solver.cc:370-387; PointToLineResidual, :180-200: HITL blocks,
 * solver.cc:515-532).  Same argument meaning as the *_dev entry points; poses / line_poses are
 * n x 3 doubles; outputs may be NULL where the *_dev form allows it.  They allocate, copy, launch
 * and synchronise per call (the inputs are a few KB). */
int nhip_resid_odometry(const float *t_odom, const float *r_odom, const int32_t *pose_i,
                        const int32_t *pose_j, int32_t n_factors, double translation_weight,
                        double rotation_weight, const double *poses, int32_t n_poses,
                        double *residuals, double *jac_i, double *jac_j);
int nhip_resid_point_to_line(const float *segments, const float *points, const int32_t *point_block,
                             int64_t n_points, const int32_t *block_pose, const int32_t *block_line,
                             int32_t n_blocks, const double *poses, int32_t n_poses,
                             const double *line_poses, int32_t n_line_poses, double *residuals,
                             double *jac_pose, double *jac_line);

/* ------------------------------------------------------------------ multi-GPU
 * The one collective of the path: every GPU matched its own shard of the candidate pairs
 * (pairs are independent; shard by target so a grid is built on exactly one GPU) and the
 * 16-byte records are all-gathered.  `comm` is the RCCL communicator (ncclComm_t, one rank
 * per GPU, created by the host with ncclCommInitRank) passed as void* so this header needs
 * no RCCL include; every rank contributes n_local records (pad the shards to equal length)
 * and receives world_size * n_local records ordered by rank.  librccl is bound on first use
 * (dlopen: the copy already loaded in the process if there is one), so single-GPU hosts
 * never need it.  A Python host does the same with torch.distributed (nautilus_amd/sharding.py). */
int nhip_allgather_matches(void *comm, const nhip_match_t *d_local, int32_t n_local,
                           nhip_match_t *d_all, void *stream);

/* ------------------------------------------------------------------ in-stream kernel timing
 * When enabled, the dominant kernels are bracketed by hipEvents on their own stream.
 * ids: 0 = csm match (bounds, candidates, exact sums), 1 = grid build (everything nhip_grid_build_dev enqueues),
 *      2 = resid_lidar, 3 = corr_search, 4 = resid_normal_eq, 5 = the clearing part of a grid build (inside 1). */
#define NHIP_TIMER_CSM 0
#define NHIP_TIMER_GRID 1
#define NHIP_TIMER_RESID 2
#define NHIP_TIMER_CORR 3
#define NHIP_TIMER_NORMEQ 4
#define NHIP_TIMER_GRID_CLEAR 5
#define NHIP_TIMER_CSM_BOUNDS 6 /* split form of the matcher: bounds + seeds (csm_bnb_kernel<., ., true, true>) ... */
#define NHIP_TIMER_CSM_CAND 7   /* ... and the candidates (csm_bnb_cand_kernel); both inside NHIP_TIMER_CSM */
#define NHIP_TIMER_EXACT_SCORE 8 /* the pass of NHIP_SEARCH_EXACT_SCORE (after, not inside, NHIP_TIMER_CSM) */
#define NHIP_TIMER_COUNT 9
int nhip_timing_enable(int on);
int nhip_timing_reset(void);
/* synchronises the recorded events; total_ms / launches since the last reset */
int nhip_timing_get(int id, double *total_ms, int32_t *launches);

#ifdef __cplusplus
}
#endif
#endif /* NAUTILUS_HIP_H_ */
