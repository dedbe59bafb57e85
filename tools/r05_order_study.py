#!/usr/bin/env python3
"""Round-5 study (CPU, numpy + the oracle's score volumes): how much of the candidates phase's exact-sum work is the ORDER
in which a pair's candidate blocks are evaluated.  At the pair's final best ONE sub-block per pair truly needs its sums
(tools/r05_level3_study.py: s4_true); the kernel evaluates 37 sub-blocks and 21 whole blocks per pair because its best
rises as it goes.  Simulated here, per pair, from the first-level bounds U1, the second-level bounds U2 and the true sums:
  A  today: rotations best-first by their highest bound, a rotation's candidate blocks in index order
  B  rotations best-first, a rotation's blocks by descending bound
  C  all candidate blocks of the pair by descending bound (what a global priority queue would do)
  Z  the final best known from the start (the floor)
Cost in row loads per list entry: 1 per refined block (strip bounds), 8 per whole block (three or four live sub-blocks),
4 per sub-block.  Usage: r05_order_study.py [n_pairs] [per_target]"""
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
    M = np.zeros((Gp.shape[0] + win, Gp.shape[1] + win), Gp.dtype)
    M[:Gp.shape[0], :Gp.shape[1]] = Gp
    v = swv(M, (win, win))[:Gp.shape[0]:stride, :Gp.shape[1]:stride].max(axis=(2, 3))
    return ((v.astype(np.int64) + 256) // 257).astype(np.int64)


def origins(src, th):
    cf, sf = np.float32(math.cos(th)), np.float32(math.sin(th))
    xr = cf * src[:, 0] - sf * src[:, 1]
    yr = sf * src[:, 0] + cf * src[:, 1]
    col = S // 2 + np.floor(xr.astype(np.float64) / res).astype(np.int64) - hx + PAD
    row = S // 2 + np.floor(yr.astype(np.float64) / res).astype(np.int64) - hx + PAD
    lim = S + 2 * PAD - 100
    return np.clip(row, 0, lim), np.clip(col, 0, lim)


def run(order, blocks, best0, U2, TS):
    """blocks: list of (u1, k, Y, X) in processing order; returns (cost, whole, subs, refined)"""
    best, cost, whole, subs, refined = best0, 0, 0, 0, 0
    for u1, k, Y, X in order:
        if u1 < best:
            continue
        refined += 1
        cost += 1
        u2 = U2[(k, Y, X)]
        live = [q for q in range(4) if u2[q] >= best and u2[q] > 0]
        if len(live) >= 3:
            whole += 1
            cost += 8
            best = max(best, int(TS[(k, Y, X)].max()))
        else:
            for q in live:
                if u2[q] < best:
                    continue
                subs += 1
                cost += 4
                best = max(best, int(TS[(k, Y, X)][q]))
    return cost, whole, subs, refined


acc = {}
for p in sel:
    s_, t_ = wl.src[p], wl.tgt[p]
    src = wl.xy[wl.off[s_]:wl.off[s_ + 1]]
    tg = wl.xy[wl.off[t_]:wl.off[t_ + 1]]
    G = O.grid_build(tg, gs)
    vol = O.csm_scores(src, G, gs, float(wl.th0[p]), ss).reshape(61, 81, 81)  # [k][ix][iy]
    final = int(vol.max())
    Gp = np.zeros((S + 2 * PAD, S + 2 * PAD), np.uint16)
    Gp[PAD:PAD + S, PAD:PAD + S] = G
    P8, P4 = pooled(Gp, 8, 15), pooled(Gp, 4, 7)
    W8 = swv(P8, (NB, NB))
    th = [float(wl.th0[p]) + (k - 30) * math.radians(1.0) for k in range(61)]
    org = [origins(src, th[k]) for k in range(61)]
    U1 = np.zeros((61, NB, NB), np.int64)
    for k in range(61):
        r, c = org[k]
        U1[k] = 257 * W8[r >> 3, c >> 3].sum(axis=0)
    seed_best, seeds = 0, set()
    for w in range(8):
        ks = np.arange(w, 61, 8)
        i = int(np.argmax(U1[ks].reshape(len(ks), -1)))
        k, b = int(ks[i // (NB * NB)]), i % (NB * NB)
        Y, X = b // NB, b % NB
        blk = vol[k, 8 * X:8 * X + 8, 8 * Y:8 * Y + 8]
        seeds.add((k, Y, X))
        if blk.size:
            seed_best = max(seed_best, int(blk.max()))
    cands = [(int(U1[k, Y, X]), int(k), int(Y), int(X)) for k, Y, X in zip(*np.nonzero(U1 >= seed_best))
             if 8 * X < 81 and 8 * Y < 81 and (int(k), int(Y), int(X)) not in seeds]
    U2, TS = {}, {}
    for u1, k, Y, X in cands:
        r, c = org[k]
        u2, ts = [], []
        for sy in range(2):
            for sx in range(2):
                r4, c4 = r + 8 * Y + 4 * sy, c + 8 * X + 4 * sx
                u2.append(257 * int(P4[r4 >> 2, c4 >> 2].sum()))
                v = vol[k, 8 * X + 4 * sx:8 * X + 4 * sx + 4, 8 * Y + 4 * sy:8 * Y + 4 * sy + 4]
                ts.append(int(v.max()) if v.size else 0)
        U2[(k, Y, X)], TS[(k, Y, X)] = u2, np.array(ts)
    rot_max = {}
    for u1, k, Y, X in cands:
        rot_max[k] = max(rot_max.get(k, 0), u1)
    rot_order = sorted(rot_max, key=lambda k: -rot_max[k])
    rank = {k: i for i, k in enumerate(rot_order)}
    A = sorted(cands, key=lambda t: (rank[t[1]], NB * t[2] + t[3]))
    B = sorted(cands, key=lambda t: (rank[t[1]], -t[0]))
    Cc = sorted(cands, key=lambda t: -t[0])
    row = {"pair": int(p), "final": final, "seed_best": seed_best, "cands": len(cands), "live_rotations": len(rot_order)}
    for name, order, b0 in (("A", A, seed_best), ("B", B, seed_best), ("C", Cc, seed_best), ("Z", A, final)):
        cost, whole, subs, refined = run(order, cands, b0, U2, TS)
        row.update({"cost_" + name: cost, "whole_" + name: whole, "subs_" + name: subs, "refined_" + name: refined})
    print(json.dumps(row), flush=True)
    for k_, v in row.items():
        if k_ != "pair":
            acc.setdefault(k_, []).append(v)

print("---- mean / median / p90 / sum over %d pairs (per_target %d)" % (len(sel), PER_TARGET))
for k_, v in acc.items():
    a = np.asarray(v, dtype=np.float64)
    print("%-18s mean %12.1f  median %12.1f  p90 %12.1f  sum %14.0f" % (k_, a.mean(), np.median(a), np.percentile(a, 90), a.sum()))
tA = np.sum(acc["cost_A"])
for name in "BCZ":
    print("cost %s / cost A: %.3f" % (name, np.sum(acc["cost_" + name]) / max(tA, 1)))
