#!/usr/bin/env python3
"""Round-5 study (CPU, numpy + the oracle's score volumes): what a cheaper FIRST pass over the rotations could prune.

The bounds phase computes, for every pair, the 121 block bounds of all 61 rotations (origins of 1081 points, run lists,
gather, reduction) before anything is pruned.  This script measures on pairs of the bench workload (configs[1]):
  * how many rotations still hold a block whose bound reaches the best sum -- the final one, and the one the seeds give
    (live rotations: only those need their rows of bounds);
  * how many rotations a LOOSE first pass would leave alive, for the loose bounds that are cheaper than the tight ones:
      clusters of consecutive beams (rho cells of path length; pool widened by rho cells per side),
      groups of g consecutive rotations bounded from the middle rotation's origins (pool dilated by the entries a point
      can move under +-(g-1)/2 degrees: by range),
      both together;
  * the cost of the two-pass scheme in units of today's bounds phase (origins + run lists 0.54, gather 0.30 per 150
    entries, reduction 0.15 per rotation; ISA census of DESIGN.md section 5).
Usage: r05_bounds_study.py [n_pairs] [per_target]"""
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import oracle as O  # noqa: E402
from numpy.lib.stride_tricks import sliding_window_view as swv  # noqa: E402

N_PAIRS = int(sys.argv[1]) if len(sys.argv) > 1 else 24
PER_TARGET = int(sys.argv[2]) if len(sys.argv) > 2 else 10
wl = bench.Workload("weak", 1, 1000, PER_TARGET)
gs, ss = O.grid_spec(cell_bits=16), O.search_spec(61, 81, 81, math.radians(1.0))
S, res, hx = O.grid_side(gs), 0.05, 40
PAD = 160
rng = np.random.default_rng(7)
sel = np.sort(rng.choice(wl.n_pairs, N_PAIRS, replace=False))
NB = 11


def pooled(Gp, rho):
    """P[R][C] = ceil(max(Gp[8R - rho : 8R + 15 + rho, 8C - rho : ...]) / 257)"""
    n = Gp.shape[0] // 8
    w = 15 + 2 * rho
    M = np.zeros((Gp.shape[0] + 128, Gp.shape[1] + 128), Gp.dtype)
    M[64:64 + Gp.shape[0], 64:64 + Gp.shape[1]] = Gp
    v = swv(M, (w, w))[64 - rho:64 - rho + 8 * n:8, 64 - rho:64 - rho + 8 * n:8].max(axis=(2, 3))
    return ((v.astype(np.int64) + 256) // 257).astype(np.int64)


def dilate(P, d):
    """max over the (2d+1)^2 neighbouring entries"""
    if d == 0:
        return P
    M = np.zeros((P.shape[0] + 2 * d, P.shape[1] + 2 * d), P.dtype)
    M[d:d + P.shape[0], d:d + P.shape[1]] = P
    return swv(M, (2 * d + 1, 2 * d + 1)).max(axis=(2, 3))


def clusters(pts, L, cap=32):
    d = np.hypot(np.diff(pts[:, 0]), np.diff(pts[:, 1]))
    first = [0]
    s = 0.0
    for i in range(1, len(pts)):
        s += d[i - 1]
        if not (s < L) or i - first[-1] >= cap:
            first.append(i)
            s = 0.0
    first = np.array(first)
    return first, np.diff(np.r_[first, len(pts)])


def origins(ref, th):
    cf, sf = np.float32(math.cos(th)), np.float32(math.sin(th))
    xr = cf * ref[:, 0] - sf * ref[:, 1]
    yr = sf * ref[:, 0] + cf * ref[:, 1]
    col = S // 2 + np.floor(xr.astype(np.float64) / res).astype(np.int64) - hx + PAD
    row = S // 2 + np.floor(yr.astype(np.float64) / res).astype(np.int64) - hx + PAD
    return np.clip(row, 0, S + 2 * PAD - 100) >> 3, np.clip(col, 0, S + 2 * PAD - 100) >> 3


def runs(A, B):
    return 1 + int(np.count_nonzero((np.diff(A) != 0) | (np.diff(B) != 0)))


def rows_of_bounds(Ws, which, ref, cnt, th):
    """U[Y][X] x 257 of one rotation; Ws[j] = windows of table j, which[i] = table of point i"""
    A, B = origins(ref, th)
    U = np.zeros((NB, NB), np.int64)
    for j, W in enumerate(Ws):
        m = which == j
        if m.any():
            U += (W[A[m], B[m]] * cnt[m, None, None]).sum(axis=0)
    return 257 * U, runs(A, B)


COST_ORG, COST_GATHER, COST_REDUCE = 0.54, 0.30, 0.15  # of one rotation of today's bounds phase (1081 points, 150 entries)
acc = {}
per_pair = []
for p in sel:
    s_, t_ = wl.src[p], wl.tgt[p]
    src = wl.xy[wl.off[s_]:wl.off[s_ + 1]]
    tg = wl.xy[wl.off[t_]:wl.off[t_ + 1]]
    G = O.grid_build(tg, gs)
    vol = O.csm_scores(src, G, gs, float(wl.th0[p]), ss).reshape(61, 81, 81)  # [k][ix][iy]
    best = int(vol.max())
    Gp = np.zeros((S + 2 * PAD, S + 2 * PAD), np.uint16)
    Gp[PAD:PAD + S, PAD:PAD + S] = G
    rng_m = np.hypot(src[:, 0].astype(np.float64), src[:, 1].astype(np.float64))
    th = [float(wl.th0[p]) + (k - 30) * math.radians(1.0) for k in range(61)]

    # ---- tight bounds (today's first level)
    P0 = pooled(Gp, 0)
    W0 = swv(P0, (NB, NB))
    ones = np.ones(len(src), np.int64)
    z = np.zeros(len(src), np.int64)
    U0 = np.zeros((61, NB, NB), np.int64)
    ent0 = 0
    for k in range(61):
        U0[k], e = rows_of_bounds([W0], z, src, ones, th[k])
        ent0 += e
    # valid blocks only (81 = 10 blocks + 1 cell)
    umax0 = U0.reshape(61, -1).max(axis=1)
    # the seeds: wave w owns rotations w, w + 8, ...; its highest-bound block, evaluated exactly
    seed_best = 0
    for w in range(8):
        ks = np.arange(w, 61, 8)
        flat = U0[ks].reshape(len(ks), -1)
        i = int(np.argmax(flat))
        k, b = int(ks[i // (NB * NB)]), i % (NB * NB)
        Y, X = b // NB, b % NB
        blk = vol[k, 8 * X:8 * X + 8, 8 * Y:8 * Y + 8]
        if blk.size:
            seed_best = max(seed_best, int(blk.max()))
    T = {"final": best, "seeds": seed_best}
    row = {"pair": int(p), "best": best, "seed_best": seed_best, "entries_tight": ent0 / 61.0}
    for name, thr in T.items():
        row["live_rot_tight_" + name] = int(np.count_nonzero(umax0 >= thr))
        row["cand_blocks_tight_" + name] = int(np.count_nonzero(U0 >= thr))

    # ---- loose first passes
    def scheme(tag, rho, g):
        """clusters of rho cells (0: points), groups of g rotations (1: every rotation)"""
        if rho == 0:
            first, cnt = np.arange(len(src)), ones
        else:
            first, cnt = clusters(src, (rho - 0.02) * res)
        ref = src[first]
        half = (g - 1) // 2
        # entries a point's pooled index can move under +- half degrees (+ 1 cell of rounding): by its range
        disp = rng_m[first] * math.sin(math.radians(half)) / res + (1.0 if half else 0.0)
        d = np.ceil(disp / 8.0).astype(np.int64) if half else np.zeros(len(first), np.int64)
        levels = sorted(set(d.tolist()))
        Pr = pooled(Gp, rho)
        Ws = [swv(dilate(Pr, lv), (NB, NB)) for lv in levels]
        which = np.searchsorted(np.array(levels), d)
        centres = list(range(half, 61, g))
        groups = [(c, [k for k in range(c - half, c + half + 1) if k < 61]) for c in centres]
        if groups[-1][1][-1] < 60:  # the tail
            groups.append((60 - half if 60 - half > groups[-1][1][-1] else 60, list(range(groups[-1][1][-1] + 1, 61))))
        ent = 0
        alive = {name: 0 for name in T}
        first_pass = 0.0
        for c, ks in groups:
            U, e = rows_of_bounds(Ws, which, ref, cnt, th[min(c, 60)])
            ent += e
            first_pass += COST_ORG * len(first) / 1081.0 + COST_GATHER * len(levels) * e / 150.0 + COST_REDUCE
            for name, thr in T.items():
                if U.max() >= thr:
                    alive[name] += len(ks)
        row["clusters_" + tag] = len(first)
        row["entries_" + tag] = ent / len(groups)
        row["first_pass_cost_" + tag] = first_pass / 61.0
        for name in T:
            row["live_rot_%s_%s" % (tag, name)] = alive[name]
            row["two_pass_cost_%s_%s" % (tag, name)] = first_pass / 61.0 + alive[name] / 61.0

    for tag, rho, g in (("c1", 1, 1), ("c2", 2, 1), ("c3", 3, 1), ("g3", 0, 3), ("g5", 0, 5), ("c1g3", 1, 3), ("c2g3", 2, 3)):
        scheme(tag, rho, g)
    per_pair.append(row)
    print(json.dumps(row), flush=True)
    for k_, v in row.items():
        if k_ != "pair":
            acc.setdefault(k_, []).append(v)

print("---- mean / median / p90 over %d pairs (per_target %d)" % (len(sel), PER_TARGET))
for k_, v in acc.items():
    a = np.asarray(v, dtype=np.float64)
    print("%-34s mean %10.2f  median %10.2f  p90 %10.2f" % (k_, a.mean(), np.median(a), np.percentile(a, 90)))
