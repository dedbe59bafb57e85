"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol the header
declares, its pure-host helpers agree with the oracle's formulas, and compute entry points
fail loudly (NHIP_ERR_NODEV) instead of falling back when no GPU is present."""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest

from nautilus_amd import _lib, csm
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "nautilus_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(nhip_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    syms = header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), "symbol %s declared in include/nautilus_hip.h is not exported" % s
    assert sorted(_lib.PROTOTYPES) == syms, "python prototypes and header disagree"


def test_struct_sizes_match_header():
    assert C.sizeof(_lib.Match) == 16
    assert C.sizeof(_lib.GridSpec) == 48  # + flags, reserved
    assert C.sizeof(_lib.Search) == 24
    assert C.sizeof(_lib.GridLayout) == 128  # + pool_*, pool4_* (branch-and-bound tables), hi_* (high bytes of 16-bit cells), hits_*


def test_grid_layout_follows_cimg_debug():
    spec = csm.grid_spec(30.0, 0.05, 2.0, 1e-10, 40, cell_bits=8)
    L = csm.grid_layout(spec)
    assert L.side == 1200 == O.grid_side(O.grid_spec(30.0, 0.05))  # floor(2*30/0.05), cimg_debug.h:21
    assert L.pad == 96 and L.pitch == 1392 and L.rows == 1392
    assert L.blur_radius == 6
    assert L.grid_bytes == 1392 * 1392
    assert L.skip_bytes == 48 * 1392  # 1 bit per row and dword column
    # pooled table: one byte per 8 x 8 cells + zero rows / columns for the reach of an 11 x 11-block lattice
    assert L.pool_rows == 174 + 12 and L.pool_pitch == 192 and L.pool_bytes == 186 * 192
    # second level: a byte pair per 4 x 4 cells + the reach of 22 sub-blocks + a 16-byte read from an aligned offset
    assert L.pool4_rows == 348 + 24 and L.pool4_pitch == 768 and L.pool4_bytes == 372 * 768
    # the matcher's tiled 8-bit plane (8-bit grids: the cells themselves): two copies of 174 x 88 tiles of 8 rows x 16 bytes
    assert L.hi_pitch == 1392 and L.hi_bytes == 2 * 174 * 88 * 128
    # the hit raster: one bit per cell + a border of 32 cells: 1264 bit rows of 40 dwords
    # (+ padding to the next 128-byte boundary of the slot: the raster is the slot's last part)
    assert L.hits_pitch == 160 and (160 * 1264 + 8) <= L.hits_bytes < (160 * 1264 + 8) + 16 + 128
    assert L.slot_bytes == L.grid_bytes + L.skip_bytes + L.pool_bytes + L.pool4_bytes + L.hi_bytes + L.hits_bytes
    # the matcher's tiled planes hold one 128-byte cache line per tile: they start on a line boundary in EVERY slot
    for lay in (L, csm.grid_layout(csm.grid_spec(30.0, 0.05, 2.0, 1e-10, 40)), csm.grid_layout(csm.grid_spec(30.0, 0.05, 2.0, 1e-10, 40, no_image=True)),
                csm.grid_layout(csm.grid_spec(10.0, 0.03, 1.0, 1e-10, 7)), csm.grid_layout(csm.grid_spec(30.0, 0.3, 2.0, 1e-10, 6)),
                csm.grid_layout(csm.grid_spec(17.0, 0.07, 2.0, 1e-10, 11, cell_bits=8, no_image=True))):
        assert lay.slot_bytes % 128 == 0 and (lay.grid_bytes + lay.skip_bytes + lay.pool_bytes + lay.pool4_bytes) % 128 == 0
        assert lay.pool4_bytes >= lay.pool4_rows * lay.pool4_pitch
    assert abs(L.score_floor - math.log(1e-10)) < 1e-15
    L16 = csm.grid_layout(csm.grid_spec(30.0, 0.05, 2.0, 1e-10, 40))  # the default width: 16-bit cells (0 means 16 too)
    zero = csm.grid_spec(30.0, 0.05, 2.0, 1e-10, 40, cell_bits=0)
    assert csm.grid_layout(zero).cell_bytes == 2
    assert L16.cell_bytes == 2 and L16.pitch == 2 * 1392 and L16.rows == 1392 and L16.grid_bytes == 2 * 1392 * 1392
    # the plane of high bytes: plain form at the 8-bit pitch; stored as two copies of 174 x 88 tiles of 8 rows x 16 bytes
    # ... and the 16-bit image once more in 174 x 174 tiles of 8 rows x 8 cells
    assert L16.hi_pitch == 1392 and L16.hi_bytes == 2 * 174 * 88 * 128 + 174 * 174 * 128
    assert L16.slot_bytes == L16.grid_bytes + L16.skip_bytes + L16.pool_bytes + L16.pool4_bytes + L16.hi_bytes + L16.hits_bytes
    for (r, res, side) in [(30, 0.3, 200), (30, 0.01, 6000), (10, 0.03, 666)]:
        assert csm.grid_layout(csm.grid_spec(r, res, 2.0, 1e-10, 4)).side == side


def test_bad_specs_are_rejected_with_message():
    lib = _lib.load()
    out = _lib.GridLayout()
    for bad in [csm.grid_spec(-1, 0.05), csm.grid_spec(30, 0), csm.grid_spec(30, 0.05, sigma=9.0),
                csm.grid_spec(30, 0.05, floor_p=2.0)]:
        assert lib.nhip_grid_layout(C.byref(bad), C.byref(out)) == _lib.NHIP_ERR_ARG
        assert len(lib.nhip_last_error()) > 0


def test_threshold_table_reproduces_direct_quantiser():
    """thr[k] <= V  <=>  q(V) >= k, against the oracle's direct libm quantiser on a 1-point grid."""
    spec = csm.grid_spec(1.0, 0.05, 2.0, 1e-10, 2, cell_bits=8)
    L = csm.grid_layout(spec)
    taps = np.zeros(2 * L.blur_radius + 1, dtype=np.int32)
    thr = np.zeros(256, dtype=np.uint32)
    _lib.check(_lib.load().nhip_grid_tables(C.byref(spec), _lib.ptr(taps), _lib.ptr(thr)))
    assert taps.sum() == L.tap_sum and np.all(taps == taps[::-1]) and taps.argmax() == L.blur_radius
    assert np.all(np.diff(thr.astype(np.int64)) >= 0)
    # one hit in the middle of a 40x40 grid: V[r][c] = taps[i]*taps[j] exactly
    g = O.grid_build(np.array([[0.01, 0.01]], dtype=np.float32), O.grid_spec(1.0, 0.05, 2.0, 1e-10, 8))
    R = L.blur_radius
    for i in range(-R, R + 1):
        for j in range(-R, R + 1):
            V = int(taps[i + R]) * int(taps[j + R])
            q = int(np.searchsorted(thr[1:], V, side="right"))
            assert q == g[20 + i, 20 + j], (i, j, V, q, g[20 + i, 20 + j])


def test_threshold_table_16bit_reproduces_direct_quantiser():
    """The 65536-entry table of the 16-bit cells: thr[k] <= V  <=>  q16(V) >= k, checked against the oracle's direct
    quantiser on every blur sum a single hit produces and on random sums around the table's entries."""
    spec = csm.grid_spec(1.0, 0.05, 2.0, 1e-10, 2, cell_bits=16)
    L = csm.grid_layout(spec)
    assert L.cell_bytes == 2 and abs(L.score_step + L.score_floor / 65535.0) < 1e-18
    taps = np.zeros(2 * L.blur_radius + 1, dtype=np.int32)
    thr = np.zeros(65536, dtype=np.uint32)
    _lib.check(_lib.load().nhip_grid_tables(C.byref(spec), _lib.ptr(taps), _lib.ptr(thr)))
    reach = thr[thr != 0xffffffff].astype(np.int64)
    assert np.all(np.diff(reach) >= 0) and len(reach) > 60000
    g = O.grid_build(np.array([[0.01, 0.01]], dtype=np.float32), O.grid_spec(1.0, 0.05, 2.0, 1e-10, 16))
    assert g.dtype == np.uint16
    R = L.blur_radius
    for i in range(-R, R + 1):
        for j in range(-R, R + 1):
            V = int(taps[i + R]) * int(taps[j + R])
            q = int(np.searchsorted(thr[1:], V, side="right"))
            assert q == g[20 + i, 20 + j], (i, j, V, q, g[20 + i, 20 + j])
    # the entries themselves: q(thr[k]) >= k and q(thr[k] - 1) < k (numpy restatement of the quantiser)
    K2 = float(L.tap_sum) ** 2
    Lf = math.log(1e-10)

    def q16(V):
        v = np.maximum(V.astype(np.float64) / K2, 1e-10)
        return np.clip(np.floor((np.log(v) - Lf) / (-Lf / 65535.0) + 0.5), 0, 65535)
    ks = np.nonzero((thr != 0xffffffff) & (thr > 0))[0][::37]
    assert np.all(q16(thr[ks].astype(np.int64)) >= ks) and np.all(q16(thr[ks].astype(np.int64) - 1) < ks)


def test_rot0_and_delta_tables():
    a = np.array([0.3, 3.0, -3.0, 7.0])
    b = np.array([0.1, -3.0, 3.0, 0.0])
    cs = csm.rot0_table(a, b)
    d = a - b
    d = d - 2 * math.pi * np.rint(d / (2 * math.pi))  # math_util.h:81-89
    assert np.array_equal(cs[:, 0], np.cos(d)) and np.array_equal(cs[:, 1], np.sin(d))
    s = csm.search_spec(61, 81, 81, math.radians(1))
    t = csm.delta_table(s)
    assert t.shape == (61, 2) and t[30, 0] == 1.0 and t[30, 1] == 0.0
    assert t[0, 1] == math.sin(-30 * math.radians(1))


def test_score_from_sum_matches_oracle_formula():
    spec = csm.grid_spec(cell_bits=8)
    lib = _lib.load()
    Lf = math.log(1e-10)
    step = -Lf / 255.0
    assert lib.nhip_score_from_sum(C.byref(spec), 231113, 975) == Lf + (step * 231113) / 975
    assert lib.nhip_score_from_sum(C.byref(spec), 0, 0) == Lf


@pytest.mark.skipif(_lib.load() is not None and _lib.device_count() > 0, reason="GPU present")
def test_compute_fails_loudly_without_gpu():
    """No CPU fallback: with no device the product refuses to compute."""
    xy = np.zeros((4, 2), dtype=np.float32)
    off = np.array([0, 4], dtype=np.int32)
    h = C.c_void_p()
    rc = _lib.load().nhip_scans_upload(_lib.ptr(xy), _lib.ptr(off), 1, C.byref(h))
    assert rc == _lib.NHIP_ERR_NODEV
    assert b"no HIP device" in _lib.load().nhip_last_error()
    with pytest.raises(_lib.NhipError):
        csm.ScanTable(xy, off)
    spec = csm.grid_spec()
    rc = _lib.load().nhip_grid_build_dev(None, None, 0, None, 0, C.byref(spec), None, None, 0, None)
    assert rc == _lib.NHIP_ERR_NODEV


def test_product_never_imports_oracle():
    """The product package must not reference oracle/ (only tests, smoke and bench may)."""
    bad = re.compile(r"(import\s+oracle|from\s+oracle|from\s+\.+\s*oracle|liboracle|orc_[a-z_]+\s*\(|"
                     r"#include\s*[\"<][^\">]*oracle)")
    for sub in ("nautilus_amd", "include"):
        for dp, _, fs in os.walk(os.path.join(ROOT, sub)):
            for f in fs:
                if f.endswith((".py", ".h", ".hip", ".cc", ".cpp", "Makefile")):
                    txt = open(os.path.join(dp, f)).read()
                    m = bad.search(txt)
                    assert m is None, "%s references the oracle: %r" % (os.path.join(dp, f), m.group(0))


def test_matcher_workspace_follows_the_size_rule():
    """nhip_csm_workspace_bytes (no GPU needed): lists of fewer than 192 pairs only need the hand-over lists (64 B
    per pair); from 192 pairs the split form parks 61+ rows of 128 bounds per pair (32 KB) in the workspace, for at
    most 131,072 pairs in a round, and twice that for longer lists (two rounds in flight)."""
    from nautilus_amd import _lib
    lib = _lib.load()
    w = lib.nhip_csm_workspace_bytes
    assert w(0) > 0 and w(191) < 1 << 20 and w(-5) == w(0)
    per_pair = 16 + 6 + 64 * 512
    assert 192 * per_pair <= w(192) < 192 * per_pair + (1 << 20)
    assert 1000 * per_pair <= w(1000) < 1000 * per_pair + (1 << 20)
    assert 4096 * per_pair <= w(4096) < 4096 * per_pair + (1 << 20)
    assert 10000 * per_pair <= w(10000) < 10000 * per_pair + (1 << 20)
    assert w(131072) < w(131073) == w(1000000) and 2 * 131072 * per_pair <= w(1000000) < 2 * 131072 * per_pair + (8 << 20)

