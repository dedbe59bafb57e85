// nhip_csm_small.hip -- K2 + K3, every add, for lattices of FEW TRANSLATIONS and many rotations.
//
// The coarse level of CorrelativeScanMatcher::GetTransformation (call site src/optimization/solver.cc:633-638; DESIGN.md
// section 3, item 7) is 181 rotations of 13 x 13 translations on a 200 x 200 table.  The two kernels that perform every
// add (nhip_csm.hip, nhip_csm16.hip) are built for planes of 81 x 81: a workgroup owns a 21-row strip of ONE rotation's
// plane and walks all of the scan's points through LDS tiles -- on a 13 x 13 plane most of its lanes idle and its
// lifetime (0.2 ms) is the whole call's.  The branch-and-bound matcher computes the bounds of all 181 rotations in the
// pair's one workgroup (23 rounds of its eight waves).  Here:
//   one 512-thread workgroup per (pair, rotation); LANES ARE POSES (up to 256 of them: four per lane, taken in the table's
//   ROW order, slot = iy * nx + ix: see the kernel), the eight waves split the scan's points; a point's window origin is computed by the lane that owns it
//   (window_cell: the spec's arithmetic, shared with the other kernels) and handed to the wave as a scalar, so every
//   lookup is one buffer load whose address is scalar base + the lane's constant pose offset -- no address arithmetic
//   in the vector unit at all -- from a table that is L2-resident (a 200 x 200 table with its border is 131 KB);
//   the waves' partial sums meet in LDS (integer adds: any order), the plane's best key goes to keys[pair] by atomicMax.
// Same sums, same argmax, same tie-break ((k * nx + ix) * ny + iy, first maximum wins) as every other kernel: tests
// compare them on every lattice this one takes.
//
// Round 6: planes of MORE than 256 translations in TILES of whole rows -- a workgroup takes `tile_rows` rows of one rotation's
// plane (at most 256 poses: four per lane), a rotation has n_tiles workgroups -- for searches of a FEW pairs, where the life
// of a call is what counts, not the lookups per second: the fine level of GetTransformation (21 rotations of 61 x 61
// translations on a 6000 x 6000 table: 336 workgroups of 4 rows) takes 36 us here whatever the clouds, against 110 us by
// branch and bound where they match and 0.2-10 ms where they do not, or 0.16-0.48 ms by the strip kernels.  Lists that
// would need more than SMALL_TILED_MAX_BLOCKS workgroups stay with the strip kernels, whose LDS tiles make 17x the lookups
// per second.
#include "nhip_csm_shared.h"

namespace nhip {

namespace {

using namespace csm;

// (round 6: 16 waves per workgroup measured no faster -- 78.7 us either way: what bounds a rotation's workgroup is its CU's
//  address unit, not a wave's chain of round trips; see the lanes' order below)
constexpr int SMALL_WAVES = 8;
constexpr int SMALL_THREADS = 64 * SMALL_WAVES;
constexpr int SMALL_PASSES = 4;  // poses per lane: lattices of up to 256 translations

__device__ __forceinline__ __amdgpu_buffer_rsrc_t small_rsrc(const void *base, int64_t bytes) {
  const uint64_t a = reinterpret_cast<uint64_t>(base);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
  const int n = __builtin_amdgcn_readfirstlane((int)(bytes < 0x7fffffffll ? bytes : 0x7fffffffll));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((uint64_t)hi << 32) | lo), 0, n, 0x00020000);
}

// PASSES: poses per lane = ceil(nx * ny / 64), a compile-time constant so that the loads of a batch of points are issued
// back to back (with the count read at run time every load sat in a basic block of its own, behind a branch: 3x slower
// than the strip kernel it was to replace)
template <int CB, int PASSES>
__global__ __launch_bounds__(SMALL_THREADS) void csm_small_plane_kernel(CsmParams P) {
  __shared__ uint32_t s_sum[64 * SMALL_PASSES];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // block -> (pair, rotation): consecutive rotations of a pair on consecutive blocks, i.e. dealt over the eight XCDs.
  // (The strip kernels keep a pair on ONE XCD for its L2; for the single pair of a GetTransformation call that put all
  //  181 workgroups on 32 of the 256 CUs: 440 us instead of 60.  A table this kernel serves is small enough for eight L2s.)
  // (... and a rotation's tiles on consecutive blocks: their windows overlap row by row)
  const uint32_t per_pair = (uint32_t)P.n_theta * (uint32_t)P.n_tiles;
  const int32_t pair = (int32_t)(blockIdx.x / per_pair);
  const int32_t k = (int32_t)((blockIdx.x % per_pair) / (uint32_t)P.n_tiles);
  const int32_t row0 = (int32_t)(blockIdx.x % (uint32_t)P.n_tiles) * P.tile_rows;  // first row (iy) of this workgroup's tile
  if (pair >= P.n_pairs) return;
  int32_t src = P.pair_src[pair], slot = P.pair_slot[pair];
  // (ids from device memory: a pair whose scan or slot lies outside the caller's counts scores nothing and is reported)
  const bool ids_ok = pair_ids_ok(P.ids, src, slot, pair, threadIdx.x == 0 && k == 0);
  if (!ids_ok) src = slot = 0;
  const int32_t beg = ids_ok ? P.offsets[src] : 0, n_pts = ids_ok ? P.offsets[src + 1] - beg : 0;
  const float2 *pts = P.xy + beg;
  const uint8_t *grid = P.grids + (size_t)slot * P.slot_bytes;
  const int32_t cx = P.pair_origin ? P.pair_origin[2 * pair] : 0;
  const int32_t cy = P.pair_origin ? P.pair_origin[2 * pair + 1] : 0;
  // (a centre the stored border cannot cover scores nothing: every sum stays 0 and pose 0 wins, as in the other kernels)
  const bool centre_ok = (abs(cx) + P.hx <= P.max_shift) && (abs(cy) + P.hy <= P.max_shift);

  // rotation k: R(theta0) * R(delta_k), composed in double with individually rounded ops
  const double c0 = P.rot0_cs[2 * pair], s0 = P.rot0_cs[2 * pair + 1];
  const double cd = P.delta_cs[2 * k], sd = P.delta_cs[2 * k + 1];
  const float cf = __double2float_rn(__dsub_rn(__dmul_rn(c0, cd), __dmul_rn(s0, sd)));
  const float sf = __double2float_rn(__dadd_rn(__dmul_rn(s0, cd), __dmul_rn(c0, sd)));

  // this lane's poses: slot = pass * 64 + lane = iy * nx + ix -- ROW-major in the table, so that the 64 lookups of one load
  // instruction fall into ~5 rows of 13 consecutive cells (5 cache lines) and a point's three loads touch each of its 13
  // rows once; with slot = ix * ny + iy (the order of the lattice's linear index, round 4) every load touched all 13
  // rows -- 39 line visits per point, and the kernel's life is the address unit's: one workgroup, i.e. one CU, per
  // rotation.  The key below still carries the lattice's own index (k * nx + ix) * ny + iy.
  const int32_t n_poses = P.nx * P.ny;
  uint32_t off[PASSES];
  int32_t lin_pose[PASSES];
  bool valid[PASSES];
#pragma unroll
  for (int p = 0; p < PASSES; p++) {
    const int32_t q = p * 64 + lane;
    valid[p] = q < P.tile_rows * P.nx && row0 + q / P.nx < P.ny;
    const int32_t iy = valid[p] ? row0 + q / P.nx : 0, ix = valid[p] ? q % P.nx : 0;
    off[p] = (uint32_t)(iy * P.pitch + ix * CB);
    lin_pose[p] = ix * P.ny + iy;
  }
  for (int i = threadIdx.x; i < 64 * SMALL_PASSES; i += SMALL_THREADS) s_sum[i] = 0u;
  __syncthreads();

  const __amdgpu_buffer_rsrc_t rs = small_rsrc(grid, P.grid_bytes);
  uint32_t acc[PASSES];
#pragma unroll
  for (int p = 0; p < PASSES; p++) acc[p] = 0u;
  // wave w takes the 64-point chunks w, w + 8, ...
  for (int32_t c = 64 * wave; c < n_pts && centre_ok; c += 64 * SMALL_WAVES) {
    const int32_t idx = c + lane;
    uint32_t base = 0u;  // (lanes past the scan's end are never read below)
    if (idx < n_pts) {
      const uint32_t cell = window_cell(pts[idx], cf, sf, P, 0, 0, cx, cy);
      base = (cell >> 16) * (uint32_t)P.pitch + (cell & 0xffffu) * (uint32_t)CB;
    }
    const int32_t n_here = min(64, n_pts - c);
    // the chunk's points one after the other, each as a scalar base: U points' lookups in flight together
    constexpr int U = 8;
    for (int32_t j0 = 0; j0 < n_here; j0 += U) {
      uint32_t v[U][PASSES];
#pragma unroll
      for (int u = 0; u < U; u++) {
        // (points past the chunk's end re-read its last point; their values are dropped below.  Lanes without a pose
        //  read the point's own cell: in range, dropped too)
        const int32_t jj = min(j0 + u, n_here - 1);
        const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)base, jj);
#pragma unroll
        for (int p = 0; p < PASSES; p++)
          v[u][p] = CB == 1 ? (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(rs, (int)off[p], (int)b, 0)
                            : (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(rs, (int)off[p], (int)b, 0);
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const uint32_t m = (j0 + u < n_here) ? (CB == 1 ? 0xffu : 0xffffu) : 0u;  // (wave-uniform)
#pragma unroll
        for (int p = 0; p < PASSES; p++) acc[p] += v[u][p] & m;
      }
    }
  }
#pragma unroll
  for (int p = 0; p < PASSES; p++)
    if (valid[p]) atomicAdd(&s_sum[p * 64 + lane], acc[p]);
  __syncthreads();
  // the plane's best key: sum << 32 | ~linear index
  if (wave == 0) {
    unsigned long long best = 0ull;
#pragma unroll
    for (int p = 0; p < PASSES; p++) {
      if (!valid[p]) continue;
      const uint32_t lin = (uint32_t)(k * n_poses + lin_pose[p]);  // (k * nx + ix) * ny + iy
      const unsigned long long key = ((unsigned long long)s_sum[p * 64 + lane] << 32) | (0xffffffffu - lin);
      best = key > best ? key : best;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      const unsigned long long o = shfl_xor_u64(best, m);
      best = o > best ? o : best;
    }
    if (lane == 0) atomicMax(&P.keys[pair], best);
  }
}

}  // namespace

bool csm_small_plane_fits(const nhip_search_t *search) { return (int64_t)search->nx * search->ny <= 64 * SMALL_PASSES; }

// Larger planes in tiles of whole rows, for lists of few pairs (see the top of the file): rows per workgroup and workgroups
// per rotation; false if a row alone exceeds a workgroup's 256 poses or the list would need too many workgroups.
constexpr int64_t SMALL_TILED_MAX_BLOCKS = 2048;
bool csm_small_tiled_fits(const nhip_search_t *search, int32_t n_pairs, int32_t *tile_rows, int32_t *n_tiles) {
  if (search->nx < 1 || search->ny < 1 || search->nx > 64 * SMALL_PASSES) return false;
  const int32_t rows = std::min<int32_t>(search->ny, (64 * SMALL_PASSES) / search->nx);
  const int32_t tiles = (search->ny + rows - 1) / rows;
  if (tile_rows) *tile_rows = rows;
  if (n_tiles) *n_tiles = tiles;
  return (int64_t)n_pairs * search->n_theta * tiles <= SMALL_TILED_MAX_BLOCKS;
}

int launch_csm_small_match(const float *d_xy, const int32_t *d_offsets, const IdBounds &ids, const uint8_t *d_grids,
                           const nhip_grid_spec_t *spec, const GridLayout &L, const int32_t *d_pair_src,
                           const int32_t *d_pair_slot, const double *d_rot0_cs, const double *d_delta_cs,
                           const int32_t *d_pair_origin, int32_t n_pairs, const nhip_search_t *search,
                           uint64_t *d_keys, nhip_match_t *d_out, int32_t *d_sums, hipStream_t s) {
  CsmParams P;
  memset(&P, 0, sizeof(P));
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
  P.n_theta = search->n_theta;
  P.nx = search->nx;
  P.ny = search->ny;
  P.hx = (search->nx - 1) / 2;
  P.hy = (search->ny - 1) / 2;
  P.S = L.S;
  P.pad = L.pad;
  P.pitch = L.pitch;
  P.rows = L.S + 2 * L.pad;
  P.max_shift = spec->max_shift;
  P.grid_bytes = L.grid_bytes;
  P.slot_bytes = L.slot_bytes;
  P.res = spec->res;
  P.inv_res = 1.0 / spec->res;
  if (csm_small_plane_fits(search)) {
    P.tile_rows = P.ny;
    P.n_tiles = 1;
  } else {
    NHIP_REQUIRE(csm_small_tiled_fits(search, n_pairs, &P.tile_rows, &P.n_tiles), "csm_match: lattice %d x %d x %d of %d pairs does not "
                 "fit the small-plane kernel", search->n_theta, search->nx, search->ny, n_pairs);
  }
  const int64_t blocks = (int64_t)n_pairs * (int64_t)P.n_theta * (int64_t)P.n_tiles;
  NHIP_REQUIRE(blocks < 0x7fffffffll, "csm_match: %lld workgroups exceed one launch; split the batch", (long long)blocks);
  if (!(search->flags & SEARCH_I_KEYS_ZERO)) NHIP_TRY_HIP(hipMemsetAsync(d_keys, 0, sizeof(uint64_t) * (size_t)n_pairs, s));
  timer_begin(NHIP_TIMER_CSM, s);
  const int passes = (P.tile_rows * P.nx + 63) / 64;
#define NHIP_SMALL_LAUNCH(CB_, PS_) hipLaunchKernelGGL((csm_small_plane_kernel<CB_, PS_>), dim3((uint32_t)blocks), dim3(SMALL_THREADS), 0, s, P)
  if (L.cb == 1) {
    if (passes == 1) NHIP_SMALL_LAUNCH(1, 1); else if (passes == 2) NHIP_SMALL_LAUNCH(1, 2);
    else if (passes == 3) NHIP_SMALL_LAUNCH(1, 3); else NHIP_SMALL_LAUNCH(1, 4);
  } else {
    if (passes == 1) NHIP_SMALL_LAUNCH(2, 1); else if (passes == 2) NHIP_SMALL_LAUNCH(2, 2);
    else if (passes == 3) NHIP_SMALL_LAUNCH(2, 3); else NHIP_SMALL_LAUNCH(2, 4);
  }
#undef NHIP_SMALL_LAUNCH
  timer_end(NHIP_TIMER_CSM, s);
  if (!(search->flags & SEARCH_I_NO_FINALIZE))
    launch_csm_finalize(d_keys, d_pair_src, d_offsets, ids.n_scans, n_pairs, P.nx, P.ny, L, d_out, d_sums, s);
  NHIP_TRY_HIP(hipGetLastError());
  return NHIP_OK;
}

}  // namespace nhip
