"""Host-side mirror of the loop-closure scan matcher interface, over the C ABI.

Mirrors (same names / argument meaning):
  CorrelativeScanMatcher(scanner_range, trans_range, low_res, high_res)
      .GetTransformation(pc_a, pc_b, rot_a, rot_b, rot_restriction)
          -> (score, ((tx, ty), theta))        /root/reference/src/optimization/solver.cc:633-644
plus the batched form the hot path is built for (ScanTable / LikelihoodGrids / match_pairs).
All compute happens in libnautilus_hip.so; nothing here falls back to the CPU.
"""
import ctypes as C
import math

import numpy as np

from . import _lib
from ._lib import GridSpec, GridLayout, Search, Match, check, ptr

MATCH_DTYPE = np.dtype([("itheta", "<i4"), ("ix", "<i4"), ("iy", "<i4"), ("score", "<f4")])
assert MATCH_DTYPE.itemsize == C.sizeof(Match) == 16


def grid_spec(range_m=30.0, res=0.05, sigma=2.0, floor_p=1e-10, max_shift=40, cell_bits=16, skip_map=False, no_image=False):
    """cell_bits: 16 (the default everywhere: 65535 quantisation steps over [ln floor_p, 0]) or 8 (255 steps: the explicit
    opt-in for callers that only gate on a threshold).  skip_map: 16-bit grids carry a skip map too (only the kernel that
    performs every add reads it; LikelihoodGrids builds a missing one the first time such a search needs it).  no_image:
    NHIP_GRID_NO_IMAGE -- slots without the row-major image (a third smaller; the branch-and-bound matcher on scans of at
    most 1088 points needs none)."""
    return GridSpec(float(range_m), float(res), float(sigma), float(floor_p), int(max_shift), int(cell_bits),
                    (_lib.NHIP_GRID_SKIP_MAP if skip_map else 0) | (_lib.NHIP_GRID_NO_IMAGE if no_image else 0), 0)


def search_spec(n_theta=61, nx=81, ny=81, theta_step=math.radians(1.0), exhaustive=False, dense=False, short_scans=False,
                exact_score=False, latency=False):
    """exhaustive=True forces the kernel that performs every add (dense=True: the all-zero strips too); the default
    (branch and bound) returns the same records bit for bit.  short_scans=True is the caller's promise that no source
    scan of the list has more than 1088 points (nautilus_hip.h, NHIP_SEARCH_SHORT_SCANS).  exact_score=True: the records'
    scores are the winning poses' scores on the unquantised table (NHIP_SEARCH_EXACT_SCORE).  latency=True (with exhaustive): a
    list of few pairs through the kernel whose lanes are poses, larger planes in tiles of rows (NHIP_SEARCH_LATENCY)."""
    flags = (_lib.NHIP_SEARCH_EXHAUSTIVE if exhaustive else 0) | (_lib.NHIP_SEARCH_DENSE if dense else 0) | \
            (_lib.NHIP_SEARCH_SHORT_SCANS if short_scans else 0) | (_lib.NHIP_SEARCH_EXACT_SCORE if exact_score else 0) | \
            (_lib.NHIP_SEARCH_LATENCY if latency else 0)
    return Search(int(n_theta), int(nx), int(ny), flags, float(theta_step))


def grid_layout(spec):
    out = GridLayout()
    check(_lib.load().nhip_grid_layout(C.byref(spec), C.byref(out)))
    return out


def angle_mod(a):
    """math_util.h:81-84"""
    return a - 2.0 * math.pi * np.rint(a / (2.0 * math.pi))


def rot0_table(rot_a, rot_b=None):
    rot_a = np.ascontiguousarray(rot_a, dtype=np.float64)
    rb = None if rot_b is None else np.ascontiguousarray(rot_b, dtype=np.float64)
    out = np.empty((rot_a.size, 2), dtype=np.float64)
    check(_lib.load().nhip_csm_rot0(ptr(rot_a), ptr(rb), rot_a.size, ptr(out)))
    return out


def delta_table(search):
    out = np.empty((search.n_theta, 2), dtype=np.float64)
    check(_lib.load().nhip_csm_delta_table(C.byref(search), ptr(out)))
    return out


def pack_scans(scans):
    """list of (N_i, 2) float32 clouds -> (xy (sum N, 2) float32, offsets (n+1) int32)."""
    offsets = np.zeros(len(scans) + 1, dtype=np.int32)
    for i, s in enumerate(scans):
        offsets[i + 1] = offsets[i] + len(s)
    xy = np.zeros((int(offsets[-1]), 2), dtype=np.float32)
    for i, s in enumerate(scans):
        if len(s):
            xy[offsets[i]:offsets[i + 1]] = np.asarray(s, dtype=np.float32).reshape(-1, 2)
    return xy, offsets


class ScanTable:
    """Device copy of the point clouds (std::vector<Vector2f> per node, slam_types.h:41-76)."""

    def __init__(self, xy, offsets):
        self.xy = np.ascontiguousarray(xy, dtype=np.float32).reshape(-1, 2)
        self.offsets = np.ascontiguousarray(offsets, dtype=np.int32)
        self.n_scans = self.offsets.size - 1
        self._h = C.c_void_p()
        check(_lib.load().nhip_scans_upload(ptr(self.xy), ptr(self.offsets), self.n_scans, C.byref(self._h)))

    @classmethod
    def from_list(cls, scans):
        return cls(*pack_scans(scans))

    def close(self):
        if self._h:
            _lib.load().nhip_scans_free(self._h)
            self._h = C.c_void_p()

    __del__ = close


class LikelihoodGrids:
    """Log-likelihood lookup tables of the target scans, resident in HBM."""

    def __init__(self, scans, target_ids, spec):
        self.spec = spec
        self.layout = grid_layout(spec)
        self.target_ids = np.ascontiguousarray(target_ids, dtype=np.int32)
        self._h = C.c_void_p()
        check(_lib.load().nhip_grids_build(scans._h, ptr(self.target_ids), self.target_ids.size,
                                           C.byref(spec), C.byref(self._h)))

    def was_rebuilt(self):
        """True when the tables were rebuilt into the buffers a released handle of the same shape left in the library's
        device pool (nhip_grids_was_rebuilt): only what that build wrote was cleared, nothing was zero-filled."""
        return bool(_lib.load().nhip_grids_was_rebuilt(self._h))

    def download(self, slot):
        """Stored (padded) grid as (rows, pitch / cell_bytes) uint8 or uint16 cells."""
        L = self.layout
        out = np.empty((L.rows, L.pitch), dtype=np.uint8)
        check(_lib.load().nhip_grids_download(self._h, int(slot), ptr(out)))
        return out.view(np.uint16) if L.cell_bytes == 2 else out

    def hi_plane(self, slot, copy=0):
        """16-bit grids: the plane of the cells' high bytes in plain form, (rows, hi_pitch) uint8 (on the device: two
        tiled copies; `copy` selects which one is read back)."""
        L = self.layout
        out = np.empty((L.rows, L.hi_pitch), dtype=np.uint8)
        check(_lib.load().nhip_grids_download_hi_plane_copy(self._h, int(slot), int(copy), ptr(out)))
        return out

    def tiled16(self, slot):
        """16-bit grids: the matcher's tiled copy of the image, read back in the plain form of download()."""
        L = self.layout
        out = np.empty((L.rows, L.pitch), dtype=np.uint8)
        check(_lib.load().nhip_grids_download_tiled16(self._h, int(slot), ptr(out)))
        return out.view(np.uint16)

    def skip_map(self, slot):
        """The slot's skip map as (rows, bytes per map row) uint8: bit c & 7 of byte c >> 3 of row r = "stored rows
        [r, r + 21) x aligned dwords [c, c + 21 * cell_bytes) hold a non-zero cell"."""
        L = self.layout
        out = np.empty(L.skip_bytes, dtype=np.uint8)
        check(_lib.load().nhip_grids_download_skip_map(self._h, int(slot), ptr(out)))
        mp = 8 * ((L.pitch // 4 + 63) // 64)
        return out[:L.rows * mp].reshape(L.rows, mp)

    def pooled(self, slot, level=1):
        """Max-pooled table uint8, the branch-and-bound matcher's bounds: level 1 (pool_rows, pool_pitch), 15 x 15
        cells at stride 8; level 2 (pool4_rows, pool4_pitch), 7 x 7 cells at stride 4 as byte pairs
        {P4[i][j], P4[i + 1][j]}."""
        L = self.layout
        if level == 1:
            out = np.empty((L.pool_rows, L.pool_pitch), dtype=np.uint8)
            check(_lib.load().nhip_grids_download_pool(self._h, int(slot), ptr(out)))
        else:
            out = np.empty((L.pool4_rows, L.pool4_pitch), dtype=np.uint8)
            check(_lib.load().nhip_grids_download_pool4(self._h, int(slot), ptr(out)))
        return out

    def hits(self, slot):
        """The hit raster the table was blurred from, as (side, side) uint8 of 0 / 1 (on the device: one bit per cell with a
        zero border of 32 cells; NHIP_SEARCH_EXACT_SCORE reads it)."""
        L = self.layout
        raw = np.empty(L.hits_bytes, dtype=np.uint8)
        check(_lib.load().nhip_grids_download_hits(self._h, int(slot), ptr(raw)))
        rows = raw[:L.hits_pitch * (L.side + 64)].reshape(L.side + 64, L.hits_pitch)
        bits = np.unpackbits(rows, axis=1, bitorder="little")
        return bits[32:32 + L.side, 32:32 + L.side]

    def interior(self, slot):
        L = self.layout
        return self.download(slot)[L.pad:L.pad + L.side, L.pad:L.pad + L.side]

    def close(self):
        if self._h:
            _lib.load().nhip_grids_free(self._h)
            self._h = C.c_void_p()

    __del__ = close


def match_pairs(scans, grids, pair_src, pair_slot, theta0, search, pair_origin=None):
    """Batched GetTransformation.  Returns (matches MATCH_DTYPE[n], sums int32[n])."""
    pair_src = np.ascontiguousarray(pair_src, dtype=np.int32)
    pair_slot = np.ascontiguousarray(pair_slot, dtype=np.int32)
    theta0 = np.ascontiguousarray(theta0, dtype=np.float64)
    n = pair_src.size
    if pair_slot.size != n or theta0.size != n:
        raise ValueError("pair arrays differ in length")
    org = None
    if pair_origin is not None:
        org = np.ascontiguousarray(pair_origin, dtype=np.int32).reshape(n, 2)
    out = np.zeros(n, dtype=MATCH_DTYPE)
    sums = np.zeros(n, dtype=np.int32)
    check(_lib.load().nhip_csm_match(scans._h, grids._h, ptr(pair_src), ptr(pair_slot), ptr(theta0),
                                     ptr(org), n, C.byref(search), ptr(out), ptr(sums)))
    return out, sums


def bnb_stats():
    """(blocks evaluated exactly, blocks in all) since the last call; needs NHIP_BNB_STATS=1 in the environment."""
    a, b = C.c_uint64(0), C.c_uint64(0)
    check(_lib.load().nhip_bnb_stats(C.byref(a), C.byref(b)))
    return a.value, b.value


def bnb_stats_levels():
    """{whole blocks evaluated, blocks in all, candidates refined, 4 x 4 sub-blocks evaluated} since the last call."""
    v = (C.c_uint64 * 16)()
    check(_lib.load().nhip_bnb_stats_levels(v))
    return {"blocks_whole": v[0], "blocks_total": v[1], "candidates_refined": v[2], "sub_blocks": v[3],
            "clk_wave_phase3": v[4], "clk_origins": v[5], "clk_sub_bounds": v[6], "clk_exact": v[7],
            "clk_slowest_wave": v[8], "clk_seeds": v[9], "clk_bounds": v[10], "clk_wave_phase3_100MHz": v[11],
            "pairs_handed_over": v[12], "clk_wave_second_kernel": v[13], "pose_evals16": v[14]}


def score_volume(scans, grids, src, slot, theta0, search, origin=(0, 0)):
    out = np.zeros((search.n_theta, search.nx, search.ny), dtype=np.int32)
    check(_lib.load().nhip_csm_scores(scans._h, grids._h, int(src), int(slot), float(theta0),
                                      int(origin[0]), int(origin[1]), C.byref(search), ptr(out)))
    return out


def match_to_transform(m, spec, search, theta0, origin=(0, 0)):
    """(tx, ty, theta) as consumed at solver.cc:640-644: T_AB = Translation(tx, ty) * Rotation(theta)."""
    tx = (origin[0] + int(m["ix"]) - (search.nx - 1) // 2) * spec.res
    ty = (origin[1] + int(m["iy"]) - (search.ny - 1) // 2) * spec.res
    th = theta0 + (int(m["itheta"]) - (search.n_theta - 1) // 2) * search.theta_step
    return np.float32(tx), np.float32(ty), np.float32(th)


class CorrelativeScanMatcher:
    """Drop-in shape of third_party/csm's class as nautilus uses it (solver.h:18,126; solver.cc:56,633).

    The two-level search itself (coarse on the low_res grid, refinement on the high_res grid around the coarse
    optimum: build-defined) lives behind the C ABI, nhip_csm_get_transformation -- the same
    entry point the C++ drop-in header calls; nothing is re-derived here."""

    def __init__(self, scanner_range, trans_range, low_res, high_res, sigma=2.0, cell_bits=16):
        self.params = _lib.CsmParams(float(scanner_range), float(trans_range), float(low_res), float(high_res),
                                     float(sigma), 1e-10, int(cell_bits), 0)

    def GetTransformation(self, pointcloud_a, pointcloud_b, rotation_a, rotation_b, rotation_restriction):
        a = np.ascontiguousarray(pointcloud_a, dtype=np.float32).reshape(-1, 2)
        b = np.ascontiguousarray(pointcloud_b, dtype=np.float32).reshape(-1, 2)
        score, tx, ty, th = C.c_double(0), C.c_float(0), C.c_float(0), C.c_float(0)
        check(_lib.load().nhip_csm_get_transformation(C.byref(self.params), ptr(a), len(a), ptr(b), len(b),
                                                      float(rotation_a), float(rotation_b), float(rotation_restriction),
                                                      C.byref(score), C.byref(tx), C.byref(ty), C.byref(th)))
        return score.value, ((np.float32(tx.value), np.float32(ty.value)), np.float32(th.value))


def last_launch():
    """What the calling thread's last branch-and-bound match did (nhip_csm_last_launch): a dict."""
    out = (C.c_int32 * 8)()
    check(_lib.load().nhip_csm_last_launch(out))
    form = {0: "fused", 1: "split, one round", 2: "split, rounds on one stream", 3: "split, rounds overlapped on the helper stream"}
    return {"form": form.get(out[0], "?"), "form_id": out[0], "pairs_per_round": out[1], "rounds_of_state": out[2],
            "rounds": out[3], "short_scans": bool(out[4]), "hand_over_kernel": bool(out[5]), "instrumented": bool(out[6]),
            "n_pairs": out[7]}


def drop_in_cache_stats():
    """{entries, bytes, hits, misses} of nhip_csm_get_transformation's cache of target tables."""
    v = [C.c_int64(0) for _ in range(4)]
    check(_lib.load().nhip_csm_cache_stats(*[C.byref(x) for x in v]))
    return dict(zip(("entries", "bytes", "hits", "misses"), (x.value for x in v)))


def drop_in_cache_clear():
    check(_lib.load().nhip_csm_cache_clear())


def drop_in_cache_configure(max_bytes):
    check(_lib.load().nhip_csm_cache_configure(int(max_bytes)))
