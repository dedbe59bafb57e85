#!/usr/bin/env python3
"""Round-5 study (CPU, numpy + the oracle's score volumes): what a THIRD bound level would prune in the candidates phase.

Today a candidate block (8 x 8 translations, first-level bound >= the pair's best) is refined through the bounds of its
four 4 x 4 sub-blocks (second-level table: 7 x 7 cells at stride 4) and every sub-block whose bound still reaches the best
gets its exact sums: four row loads per list entry (eight for a whole block).  The candidates kernel's time goes with its
load instructions (profiles/r05_cand_strip_loads.txt), 44 % of which are those sub-block sums.  A third level -- 2 x 2
translations, a table of 3 x 3-cell maxima at stride 2 (484 KB per target) -- would cost ONE more load per entry and live
sub-block and save the rows of the 2 x 2 quarters it kills.  This script counts, on pairs of the bench workload
(configs[1]) against the pair's final best sum (the most any scheme can know) and against the best after the seeds:
  cand      candidate blocks (first-level bound >= threshold)
  s4        their 4 x 4 sub-blocks with a second-level bound >= threshold (what is evaluated exactly today)
  s4_true   of those, the ones that really hold a sum >= threshold (what an oracle would evaluate)
  q2        2 x 2 quarters of the s4 sub-blocks with a third-level bound >= threshold
  s4_dead3  s4 sub-blocks all four of whose quarters the third level kills
  rows3     row pairs (2 rows x 8 bytes = one load per entry) the surviving quarters need
and prices both forms in load instructions per list entry: today 4 * s4; with the third level s4 + rows3.
Usage: r05_level3_study.py [n_pairs] [per_target]"""
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


def pooled(Gp, stride, win):
    """P[i][j] = ceil(max(Gp[stride i : stride i + win, stride j : ...]) / 257), zero beyond the raster"""
    M = np.zeros((Gp.shape[0] + win, Gp.shape[1] + win), Gp.dtype)
    M[:Gp.shape[0], :Gp.shape[1]] = Gp
    v = swv(M, (win, win))[:Gp.shape[0]:stride, :Gp.shape[1]:stride].max(axis=(2, 3))
    return ((v.astype(np.int64) + 256) // 257).astype(np.int64)


def origins(src, th):
    """window origins in padded cells (row, column): the spec's arithmetic"""
    cf, sf = np.float32(math.cos(th)), np.float32(math.sin(th))
    xr = cf * src[:, 0] - sf * src[:, 1]
    yr = sf * src[:, 0] + cf * src[:, 1]
    col = S // 2 + np.floor(xr.astype(np.float64) / res).astype(np.int64) - hx + PAD
    row = S // 2 + np.floor(yr.astype(np.float64) / res).astype(np.int64) - hx + PAD
    lim = S + 2 * PAD - 100
    return np.clip(row, 0, lim), np.clip(col, 0, lim)


acc = {}
for p in sel:
    s_, t_ = wl.src[p], wl.tgt[p]
    src = wl.xy[wl.off[s_]:wl.off[s_ + 1]]
    tg = wl.xy[wl.off[t_]:wl.off[t_ + 1]]
    G = O.grid_build(tg, gs)
    vol = O.csm_scores(src, G, gs, float(wl.th0[p]), ss).reshape(61, 81, 81)  # [k][ix][iy]
    best = int(vol.max())
    Gp = np.zeros((S + 2 * PAD, S + 2 * PAD), np.uint16)
    Gp[PAD:PAD + S, PAD:PAD + S] = G
    P8, P4, P2 = pooled(Gp, 8, 15), pooled(Gp, 4, 7), pooled(Gp, 2, 3)
    W8 = swv(P8, (NB, NB))
    th = [float(wl.th0[p]) + (k - 30) * math.radians(1.0) for k in range(61)]
    org = [origins(src, th[k]) for k in range(61)]
    U1 = np.zeros((61, NB, NB), np.int64)
    for k in range(61):
        r, c = org[k]
        U1[k] = 257 * W8[r >> 3, c >> 3].sum(axis=0)
    # the seeds' best (wave w owns rotations w, w + 8, ...; its highest-bound block, evaluated exactly)
    seed_best = 0
    for w in range(8):
        ks = np.arange(w, 61, 8)
        i = int(np.argmax(U1[ks].reshape(len(ks), -1)))
        k, b = int(ks[i // (NB * NB)]), i % (NB * NB)
        Y, X = b // NB, b % NB
        blk = vol[k, 8 * X:8 * X + 8, 8 * Y:8 * Y + 8]
        if blk.size:
            seed_best = max(seed_best, int(blk.max()))
    row = {"pair": int(p), "best": best, "seed_best": seed_best}
    for name, T in (("final", best), ("seeds", seed_best)):
        cand = s4 = s4_true = q2 = s4_dead3 = rows3 = 0
        for k, Y, X in zip(*np.nonzero(U1 >= T)):
            if 8 * X >= 81 or 8 * Y >= 81:
                continue  # (blocks past the 81 x 81 lattice)
            cand += 1
            r, c = org[k]
            for sy in range(2):
                for sx in range(2):
                    r4, c4 = r + 8 * Y + 4 * sy, c + 8 * X + 4 * sx
                    if 257 * int(P4[r4 >> 2, c4 >> 2].sum()) < T:
                        continue
                    s4 += 1
                    v = vol[k, 8 * X + 4 * sx:8 * X + 4 * sx + 4, 8 * Y + 4 * sy:8 * Y + 4 * sy + 4]
                    if v.size and int(v.max()) >= T:
                        s4_true += 1
                    live = np.zeros((2, 2), bool)
                    for a in range(2):
                        for b in range(2):
                            live[a, b] = 257 * int(P2[(r4 + 2 * a) >> 1, (c4 + 2 * b) >> 1].sum()) >= T
                    q2 += int(live.sum())
                    s4_dead3 += int(not live.any())
                    rows3 += int(live.any(axis=1).sum())
        row.update({"cand_" + name: cand, "s4_" + name: s4, "s4_true_" + name: s4_true, "q2_" + name: q2,
                    "s4_dead3_" + name: s4_dead3, "rows3_" + name: rows3,
                    "loads_today_" + name: 4 * s4, "loads_level3_" + name: s4 + rows3})
    print(json.dumps(row), flush=True)
    for k_, v in row.items():
        if k_ != "pair":
            acc.setdefault(k_, []).append(v)

print("---- mean / median / p90 / sum over %d pairs (per_target %d)" % (len(sel), PER_TARGET))
for k_, v in acc.items():
    a = np.asarray(v, dtype=np.float64)
    print("%-22s mean %12.1f  median %12.1f  p90 %12.1f  sum %14.0f" % (k_, a.mean(), np.median(a), np.percentile(a, 90), a.sum()))
for name in ("final", "seeds"):
    t, l3 = np.sum(acc["loads_today_" + name]), np.sum(acc["loads_level3_" + name])
    print("sub-block sums' loads with the third level / today (%s): %.3f" % (name, l3 / max(t, 1)))
