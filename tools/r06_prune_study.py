#!/usr/bin/env python3
"""Round-6 study on the GPU (torch; every pair of the bench's list or a sample): what could prune the matcher's work
further.  Counts per pair, from the library's own tables and the 16-bit image:

  thresholds   final best / the seeds' best (each of 8 waves evaluates its highest-bound block of rotations w, w + 8, ...) /
               a PRIOR best (exact sums of the 3 x 3 blocks around the search centre at theta0 and its +-2 rotations: 45
               blocks, evaluated before any bound) / max(prior, seeds)
  at each      live rotations (a block's bound reaches the threshold), candidate blocks
  6(a)         per DEAD rotation (at the prior best): the fraction of the points, taken in beam order in 64-point chunks,
               after which  max_block(partial bound) + 255 * 257 * (points left)  falls below the threshold -- what an
               early-out of the bounds pass on its partial sums could skip
  groups       first-pass bounds of GROUPS of g consecutive rotations from the middle rotation's origins, with the pooled
               table dilated PER POINT by the cells its arc sweeps (window 15 + 2 * ceil(rho * sin(half group angle) /
               res) cells at stride 8, by range class) -- finer than r05's dilation by whole entries -- and the live
               rotations / two-pass cost they leave
Usage (GPU box):  python3 tools/r06_prune_study.py [n_pairs | all]      -> gpurun_out/r06_prune_study.json + summary"""
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

NB = 11
COST_ORG, COST_GATHER, COST_REDUCE = 0.54, 0.30, 0.15   # of one rotation of today's bounds phase (r05_bounds_study.txt)


def main():
    import torch
    import torch.nn.functional as F
    from nautilus_amd import csm, sharding
    dev = torch.device("cuda:0")
    wl = bench.Workload("weak", 1)
    plan = sharding.ShardPlan(wl.src, wl.tgt, wl.th0, 1, None)
    shard = plan.shard(0)
    m = bench.HipMatcher(wl, shard, dev, 16, exact_score=False, no_image=False)
    m.step()
    torch.cuda.synchronize()
    best_all = m.records()[1].to(torch.int64)
    n_pairs = m.n_pairs
    arg = sys.argv[1] if len(sys.argv) > 1 else "2000"
    sel = np.arange(n_pairs) if arg == "all" else np.sort(np.random.default_rng(7).choice(n_pairs, int(arg), replace=False))

    L = m.layout
    S, pad, h, res = L.side, L.pad, 40, 0.05
    half = S // 2
    rows = L.rows
    off1 = L.grid_bytes + L.skip_bytes
    G = m.d_grids
    d_delta = torch.from_numpy(csm.delta_table(m.search).reshape(-1, 2)).to(dev)
    xy, offs = m.d_xy, wl.off
    groups = {3: [], 5: []}
    acc = {k: [] for k in ("live_final", "cand_final", "live_seed", "cand_seed", "live_prior", "cand_prior", "live_ps", "cand_ps", "live_seedsub", "cand_seedsub", "seedsub_over_final",
                           "prior_over_final", "seed_over_final", "early_out_frac_dead", "dead_rot_prior", "far_frac_11m", "far_frac_23m")}
    gacc = {}
    t0 = time.time()
    cache = {}
    for n, p in enumerate(sel):
        s_, sl = int(m.src[p]), int(m.slot[p])
        if sl not in cache:
            cache.clear()
            base = sl * L.slot_bytes
            img = G[base: base + L.grid_bytes].view(torch.int16).view(rows, L.pitch // 2)[:, :rows].to(torch.int32) & 0xffff
            P1 = G[base + off1: base + off1 + L.pool_bytes].view(L.pool_rows, L.pool_pitch)
            imf = img.to(torch.float32)[None, None]
            tabs = {0: P1}
            for d in (1, 2, 3, 4, 6, 8, 12):   # window 15 + 2 d at stride 8, entry i covers stored cells [8 i - d, 8 i + 15 + d)
                t = F.max_pool2d(F.pad(imf, (d, 22 + d, d, 22 + d)), kernel_size=15 + 2 * d, stride=8)[0, 0]
                t = torch.div(t.to(torch.int64) + 256, 257, rounding_mode="floor").to(torch.uint8)
                tt = torch.zeros((L.pool_rows, L.pool_pitch), dtype=torch.uint8, device=dev)
                r_, c_ = min(t.shape[0], L.pool_rows), min(t.shape[1], L.pool_pitch)
                tt[:r_, :c_] = t[:r_, :c_]
                tabs[d] = tt
            off2 = off1 + L.pool_bytes
            P2 = G[base + off2: base + off2 + L.pool4_rows * L.pool4_pitch].view(L.pool4_rows, L.pool4_pitch)[:, 0::2]
            cache[sl] = (img, {d: t.unfold(0, NB, 1).unfold(1, NB, 1) for d, t in tabs.items()}, img.unfold(0, 8, 1).unfold(1, 8, 1), P2)
        img, W, W8, P2 = cache[sl]
        pts = xy[int(offs[s_]):int(offs[s_ + 1])]
        npts = pts.shape[0]
        c0, s0 = math.cos(float(m.h_th0[p])), math.sin(float(m.h_th0[p]))
        cd, sd = d_delta[:, 0], d_delta[:, 1]
        cf = (c0 * cd - s0 * sd).to(torch.float32)[:, None]
        sf = (s0 * cd + c0 * sd).to(torch.float32)[:, None]
        x, y = pts[None, :, 0], pts[None, :, 1]
        xr, yr = cf * x - sf * y, sf * x + cf * y
        ix = torch.floor(xr.to(torch.float64) / res).to(torch.int64).clamp(-h - 1 - half, S + h - half)
        iy = torch.floor(yr.to(torch.float64) / res).to(torch.int64).clamp(-h - 1 - half, S + h - half)
        pcol, prow = ix + (half - h + pad), iy + (half - h + pad)
        b_final = int(best_all[p])
        ent = W[0][prow >> 3, pcol >> 3]                                   # (61, n, 11, 11) uint8
        U1 = ent.sum(dim=1, dtype=torch.int32).to(torch.int64) * 257       # (61, 11, 11)

        def exact_block_max(k, Y, X):
            w = W8[(prow[k] + 8 * Y).clamp(max=rows - 8), (pcol[k] + 8 * X).clamp(max=rows - 8)]   # (n, 8, 8)
            sm = w.sum(dim=0, dtype=torch.int64)
            ny, nx = min(8, 81 - 8 * Y), min(8, 81 - 8 * X)
            return int(sm[:ny, :nx].max())

        def exact_sub_max(k, Y, X, qy, qx):
            w = W8[(prow[k] + 8 * Y).clamp(max=rows - 8), (pcol[k] + 8 * X).clamp(max=rows - 8)]   # (n, 8, 8)
            sm = w.sum(dim=0, dtype=torch.int64)[4 * qy:4 * qy + 4, 4 * qx:4 * qx + 4]
            ny, nx = max(0, min(4, 81 - 8 * Y - 4 * qy)), max(0, min(4, 81 - 8 * X - 4 * qx))
            return int(sm[:ny, :nx].max()) if ny and nx else 0

        b_seed, b_seed_sub = 0, 0
        for w_ in range(8):
            ks = torch.arange(w_, 61, 8, device=dev)
            flat = U1[ks].flatten(1)
            i = int(flat.argmax())
            k, bb = int(ks[i // (NB * NB)]), i % (NB * NB)
            b_seed = max(b_seed, exact_block_max(k, bb // NB, bb % NB))
            # the seed evaluated as ONE 4 x 4 sub-block: the one with the highest second-level bound of the wave's best block
            Yb, Xb = bb // NB, bb % NB
            r4, c4 = (prow[k] >> 2), (pcol[k] >> 2)
            bnds = [(int(P2[(r4 + 2 * Yb + qy).clamp(max=P2.shape[0] - 1), (c4 + 2 * Xb + qx).clamp(max=P2.shape[1] - 1)].sum(dtype=torch.int64)), qy, qx)
                    for qy in (0, 1) for qx in (0, 1)]
            _, qy, qx = max(bnds)
            b_seed_sub = max(b_seed_sub, exact_sub_max(k, Yb, Xb, qy, qx))
        b_prior = 0
        for k in range(28, 33):
            for Y in (4, 5, 6):
                for X in (4, 5, 6):
                    b_prior = max(b_prior, exact_block_max(k, Y, X))
        for name, thr in (("final", b_final), ("seed", b_seed), ("seedsub", b_seed_sub), ("prior", b_prior), ("ps", max(b_prior, b_seed))):
            c1 = U1 >= thr
            acc["live_" + name].append(int(c1.flatten(1).any(dim=1).sum()))
            acc["cand_" + name].append(int(c1.sum()))
        acc["prior_over_final"].append(b_prior / max(b_final, 1))
        acc["seed_over_final"].append(b_seed / max(b_final, 1))
        acc["seedsub_over_final"].append(b_seed_sub / max(b_final, 1))
        # 6(a): early-out on partial sums, threshold = max(prior, seeds) (the best anything could know before the bounds)
        thr = max(b_prior, b_seed)
        dead = torch.nonzero(U1.flatten(1).max(dim=1).values < thr).flatten()
        acc["dead_rot_prior"].append(int(len(dead)))
        if len(dead):
            nch = (npts + 63) // 64
            e = ent[dead].to(torch.int32)                                   # (d, n, 11, 11)
            padn = nch * 64 - npts
            if padn:
                e = torch.cat([e, torch.zeros((e.shape[0], padn, NB, NB), dtype=torch.int32, device=dev)], dim=1)
            part = e.view(e.shape[0], nch, 64, NB, NB).sum(dim=2).cumsum(dim=1)          # after chunk j
            left = torch.clamp(npts - 64 * torch.arange(1, nch + 1, device=dev), min=0)   # points after chunk j
            bound = (part.flatten(2).max(dim=2).values + 255 * left[None, :]) * 257       # (d, nch)
            ok = bound < thr
            first = torch.where(ok.any(dim=1), ok.to(torch.int32).argmax(dim=1) + 1, torch.full((e.shape[0],), nch, device=dev))
            acc["early_out_frac_dead"].append(float(first.to(torch.float64).mean()) / nch)
        rng = torch.hypot(pts[:, 0].to(torch.float64), pts[:, 1].to(torch.float64))
        acc["far_frac_11m"].append(float((rng > 11.4).to(torch.float64).mean()))
        acc["far_frac_23m"].append(float((rng > 22.9).to(torch.float64).mean()))
        # groups of g rotations from the middle rotation's origins, per-point dilation by arc
        # class sets: which dilations (cells each side; window 15 + 2 d at stride 8) a kernel would keep tables for; a point
        # takes the smallest class that covers its arc, a point beyond the largest scores the table maximum (255)
        CLASS_SETS = {"all": (1, 2, 3, 4, 6, 8, 12), "4_12": (4, 12), "4_8": (4, 8), "6": (6,), "8": (8,), "12": (12,)}
        for g in (2, 3, 4, 5):
            half_ang = math.radians(1.0) * (g - 1) / 2.0
            arc = torch.ceil(rng * math.sin(half_ang) / res + 1e-3).to(torch.int64)      # cells each side
            mids = []
            for lo_k in range(0, 61, g):
                hi_k = min(lo_k + g - 1, 60)
                ang = float(m.h_th0[p]) + ((lo_k + hi_k) / 2.0 - 30.0) * math.radians(1.0)   # the group's middle angle
                cm, sm = np.float32(math.cos(ang)), np.float32(math.sin(ang))
                gx = torch.floor((cm * pts[:, 0] - sm * pts[:, 1]).to(torch.float64) / res).to(torch.int64).clamp(-h - 1 - half, S + h - half)
                gy = torch.floor((sm * pts[:, 0] + cm * pts[:, 1]).to(torch.float64) / res).to(torch.int64).clamp(-h - 1 - half, S + h - half)
                mids.append((lo_k, hi_k, (gy + (half - h + pad)) >> 3, (gx + (half - h + pad)) >> 3))
            for cname, cset in CLASS_SETS.items():
                dcls = torch.full_like(arc, -1)
                for d in sorted(cset, reverse=True):
                    dcls = torch.where(arc <= d, torch.full_like(arc, d), dcls)
                over = dcls < 0
                n_over = int(over.sum())
                live_rot = 0
                for lo_k, hi_k, gr8, gc8 in mids:
                    Ug = torch.zeros((NB, NB), dtype=torch.int64, device=dev)
                    for d in cset:
                        msk = dcls == d
                        if bool(msk.any()):
                            Ug += W[d][gr8[msk], gc8[msk]].sum(dim=0, dtype=torch.int64)
                    Ug = (Ug + 255 * n_over) * 257
                    if bool((Ug >= thr).any()):
                        live_rot += hi_k - lo_k + 1
                key = "g%d_classes_%s" % (g, cname)
                gacc.setdefault(key, {"live": [], "first_pass_cost": [], "two_pass_cost": [], "points_beyond_largest_class": []})
                first_cost = len(mids) / 61.0
                gacc[key]["live"].append(live_rot)
                gacc[key]["first_pass_cost"].append(first_cost)
                gacc[key]["two_pass_cost"].append(first_cost + live_rot / 61.0)
                gacc[key]["points_beyond_largest_class"].append(n_over / float(npts))
        if n % 200 == 199:
            print("  %d / %d pairs, %.0f s" % (n + 1, len(sel), time.time() - t0), flush=True)

    q = lambda a: {"mean": float(np.mean(a)), "median": float(np.median(a)), "p90": float(np.percentile(a, 90)), "p99": float(np.percentile(a, 99))}
    out = {"source": "tools/r06_prune_study.py, %d pairs of the bench's list (configs[1], u16)" % len(sel),
           "thresholds": {k: q(v) for k, v in acc.items() if v},
           "groups_at_max_of_prior_and_seeds": {k: {kk: q(vv) for kk, vv in v.items()} for k, v in gacc.items()},
           "note": "group pass: g consecutive rotations bounded from the origins at the group's MIDDLE angle; "
                   "a point's table is the smallest class (cells of dilation each side; window 15 + 2 d at stride 8) of the class set that covers its arc, beyond the largest: 255; "
                   "two_pass_cost = groups / 61 + live rotations / 61 in units of today's bounds phase, before any overhead "
                   "(barriers, compaction, the extra tables' build and LDS)"}
    txt = json.dumps(out, indent=1)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", "r06_prune_study.json"), "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
