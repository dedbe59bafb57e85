#!/usr/bin/env python3
"""Study (CPU, numpy): first-level bounds of the matcher from CLUSTERS of consecutive beams instead of single points.

A cluster = consecutive points whose path length from the cluster's first point stays below L = rho cells: under every
rotation its members' window origins lie within rho cells of the first point's, so the pooled entry of the first point's
block, pooled over rho more cells on every side, bounds every member.  Per rotation the origins and run lists are then
computed per cluster, not per point.  This script measures, on pairs of the bench workload: clusters per scan, list
entries per rotation, and how many (rotation, block) bounds still reach the optimum (the candidates' work) with the
wider pool."""
import math, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from oracle import oracle as O
from numpy.lib.stride_tricks import sliding_window_view as swv

N_PAIRS = int(sys.argv[1]) if len(sys.argv) > 1 else 16
wl = bench.Workload("weak", 1, 1000, 10)
gs, ss = O.grid_spec(cell_bits=16), O.search_spec(61, 81, 81, math.radians(1.0))
S, res, hx = O.grid_side(gs), 0.05, 40
PAD = 128
rng = np.random.default_rng(1)
sel = rng.choice(wl.n_pairs, N_PAIRS, replace=False)


def pooled(Gp, rho):
    # P[R][C] = ceil(max(Gp[8R - rho : 8R + 15 + rho, 8C - rho : ...]) / 257)
    n = Gp.shape[0] // 8
    w = 15 + 2 * rho
    M = np.zeros((Gp.shape[0] + 64, Gp.shape[1] + 64), Gp.dtype)
    M[32:32 + Gp.shape[0], 32:32 + Gp.shape[1]] = Gp
    v = swv(M, (w, w))[32 - rho:32 - rho + 8 * n:8, 32 - rho:32 - rho + 8 * n:8].max(axis=(2, 3))
    return ((v.astype(np.int64) + 256) // 257).astype(np.int64)


def clusters(pts, L, cap):
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


def runs(A, B):
    return 1 + int(np.count_nonzero((np.diff(A) != 0) | (np.diff(B) != 0)))


tot = {}
for p in sel:
    s_, t_ = wl.src[p], wl.tgt[p]
    src = wl.xy[wl.off[s_]:wl.off[s_ + 1]]
    tg = wl.xy[wl.off[t_]:wl.off[t_ + 1]]
    G = O.grid_build(tg, gs)
    m = O.csm_match(src, G, gs, float(wl.th0[p]), ss)
    best = m.sum
    Gp = np.zeros((S + 2 * PAD, S + 2 * PAD), np.uint16)
    Gp[PAD:PAD + S, PAD:PAD + S] = G
    rows = {}
    for rho, cap in ((0, 1), (1, 8), (1, 16), (2, 16), (2, 32), (3, 32)):
        P = pooled(Gp, rho)
        W = swv(P, (11, 11))
        if rho == 0:
            first, cnt = np.arange(len(src)), np.ones(len(src), np.int64)
        else:
            first, cnt = clusters(src, (rho - 0.02) * res, cap)
        ref = src[first]
        surv = ent = 0
        for k in range(61):
            th = float(wl.th0[p]) + (k - 30) * math.radians(1.0)
            cf, sf = np.float32(math.cos(th)), np.float32(math.sin(th))
            xr = cf * ref[:, 0] - sf * ref[:, 1]
            yr = sf * ref[:, 0] + cf * ref[:, 1]
            col = S // 2 + np.floor(xr.astype(np.float64) / res).astype(np.int64) - hx + PAD
            row = S // 2 + np.floor(yr.astype(np.float64) / res).astype(np.int64) - hx + PAD
            col = np.clip(col, 0, S + 2 * PAD - 96)
            row = np.clip(row, 0, S + 2 * PAD - 96)
            A, B = row >> 3, col >> 3
            U = (W[A, B] * cnt[:, None, None]).sum(axis=0)
            surv += int(np.count_nonzero(257 * U >= best))
            ent += runs(A, B)
        rows[(rho, cap)] = (len(first), ent / 61.0, surv)
    base = rows[(0, 1)]
    print("pair %5d best %8d | " % (p, best) + " | ".join("rho %d cap %2d: %4d clusters, %5.1f entries, %5d blocks (x%.2f)" % (r, c, v[0], v[1], v[2], v[2] / max(base[2], 1)) for (r, c), v in rows.items()), flush=True)
    for key, v in rows.items():
        t = tot.setdefault(key, [0, 0.0, 0])
        t[0] += v[0]; t[1] += v[1]; t[2] += v[2]
print("---- mean over %d pairs" % len(sel))
for key, t in tot.items():
    print("rho %d cap %2d: %6.1f clusters per scan, %6.1f list entries per rotation, %8.1f surviving blocks per pair (x%.3f)" % (key[0], key[1], t[0] / len(sel), t[1] / len(sel), t[2] / len(sel), t[2] / max(tot[(0, 1)][2], 1)))
