#!/usr/bin/env python3
"""What an IDEALLY-PRUNED matcher would cost on this chip for the bench's pair list (the `roofline.ideal_ms` of bench.py;
DESIGN.md section 6 derives it).  Runs on the GPU box:   python3 tools/ideal_matcher.py [n_pairs_sampled | all]

The branch-and-bound matcher must, whatever its order of work,
  (1) know a bound for every block that could hold the optimum: at the very least the rows of bounds of the rotations
      that still hold a block whose bound reaches the pair's FINAL best sum ("live rotations at the final best"), and
  (2) settle every block whose bound reaches that sum ("candidate blocks at the final best"): no bound-based matcher can
      skip those, because nothing but their exact sums tells them from the optimum.
Both counts are properties of the pair list and of the two bounding tables, not of the kernels: this
tool computes them for every pair from the tables the library built (torch on the GPU, no product code involved beyond the
table build and the records' best sums), and prices them with the chip's measured unit costs:
  bounds      live rotations x (vector instructions per rotation of the shipped bounds kernel: SQ_INSTS_VALU per launch /
              (pairs x 61), profiles/traffic.json) at the vector-instruction PEAK (1024 SIMDs x 2.4 GHz / 2 clk per wave64
              instruction, MI355X_MICROARCH.md)
  candidates  candidate blocks x (wave-level loads per refined block of the shipped candidates kernel: SQ_INSTS_VMEM_RD per
              launch / blocks it refined, counted by the instrumented build on the same list) x the cost of a wave-level
              gather of this shape from an L2-resident table, 27.6 clocks per wave-load and CU (13 lines per load:
              tools/ubench_gather.hip, profiles/r05_ubench_gather.txt, = max(17, 2.2 x lines)), over 256 CUs at 2.4 GHz
ideal_ms = both, per launch of the whole list.  It is a floor for THIS family of matchers (these tables, these unit costs),
not for the problem: a tighter bounding table would lower the counts themselves.

Writes profiles/ideal_matcher.json (gpurun_out/ideal_matcher.json on the GPU box) and prints the summary."""
import ctypes as C
import json
import math
import os
import sys
import time

os.environ.setdefault("NHIP_TUNABLES", "1")  # (the library reads its switches only then)
os.environ.setdefault("NHIP_BNB_STATS", "1")
os.environ.setdefault("NHIP_BNB_INSTRUMENT", "1")

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CLK_PER_WAVE_LOAD = 27.6     # profiles/r05_ubench_gather.txt, pattern 4 ("wall: 13 lines per load"), table L2-resident
CU, GHZ = 256, 2.4
NB, NB4 = 11, 21             # 8 x 8 blocks / 4 x 4 sub-blocks per axis of the 81 x 81 plane of translations


def main():
    import torch
    from nautilus_amd import _lib, csm
    lib = _lib.load()
    dev = torch.device("cuda:0")
    cell_bits = 16
    wl = bench.Workload("weak", 1)
    from nautilus_amd import sharding
    plan = sharding.ShardPlan(wl.src, wl.tgt, wl.th0, 1, None)
    shard = plan.shard(0)
    idx, src, tgt, th0, ids, slot = shard
    m = bench.HipMatcher(wl, shard, dev, cell_bits, exact_score=False)
    m.step()
    torch.cuda.synchronize()
    csm.bnb_stats_levels()            # reset: count ONE launch
    m.step()
    torch.cuda.synchronize()
    lv = csm.bnb_stats_levels()
    best = m.records()[1].to(torch.int64)          # the pairs' final best sums, shard order
    n_pairs = m.n_pairs
    arg = sys.argv[1] if len(sys.argv) > 1 else "all"
    if arg == "all":
        sel = np.arange(n_pairs)
    else:
        sel = np.sort(np.random.default_rng(7).choice(n_pairs, int(arg), replace=False))

    L = m.layout
    S, pad, h, res = L.side, L.pad, 40, 0.05
    half = S // 2
    slot_bytes = L.slot_bytes
    off1 = L.grid_bytes + L.skip_bytes
    off2 = off1 + L.pool_bytes
    G = m.d_grids
    delta = csm.delta_table(m.search).reshape(-1, 2)   # (61, 2) cos, sin of the rotation steps, double
    d_delta = torch.from_numpy(delta).to(dev)
    xy = m.d_xy
    offs = wl.off
    scale = 257 if cell_bits == 16 else 1

    live_rot = np.zeros(len(sel), np.int64)
    cand_blocks = np.zeros(len(sel), np.int64)
    live_sub = np.zeros(len(sel), np.int64)
    whole_blocks = np.zeros(len(sel), np.int64)   # candidate blocks with >= 3 live sub-blocks (the kernel evaluates those whole)
    t0 = time.time()
    cache = {}
    for n, p in enumerate(sel):
        s_, sl = int(m.src[p]), int(m.slot[p])
        if sl not in cache:
            cache.clear()
            base = sl * slot_bytes
            P1 = G[base + off1: base + off1 + L.pool_bytes].view(L.pool_rows, L.pool_pitch)
            P2 = G[base + off2: base + off2 + L.pool4_bytes].view(L.pool4_rows, L.pool4_pitch)[:, 0::2]
            cache[sl] = (P1.unfold(0, NB, 1).unfold(1, NB, 1), P2.unfold(0, NB4, 1).unfold(1, NB4, 1))
        W1, W2 = cache[sl]
        pts = xy[int(offs[s_]):int(offs[s_ + 1])]                     # (n, 2) float32
        c0, s0 = math.cos(float(m.h_th0[p])), math.sin(float(m.h_th0[p]))
        # R(theta_k) = R(theta0) R(delta_k) in double with individually rounded products, then float
        cd, sd = d_delta[:, 0], d_delta[:, 1]
        cf = (c0 * cd - s0 * sd).to(torch.float32)[:, None]
        sf = (s0 * cd + c0 * sd).to(torch.float32)[:, None]
        x, y = pts[None, :, 0], pts[None, :, 1]
        xr = cf * x - sf * y
        yr = sf * x + cf * y
        ix = torch.floor(xr.to(torch.float64) / res).to(torch.int64).clamp(-h - 1 - half, S + h - half)
        iy = torch.floor(yr.to(torch.float64) / res).to(torch.int64).clamp(-h - 1 - half, S + h - half)
        pcol = ix + (half - h + pad)
        prow = iy + (half - h + pad)
        b = int(best[p])
        U1 = W1[prow >> 3, pcol >> 3].sum(dim=1, dtype=torch.int32).to(torch.int64) * scale      # (61, 11, 11)
        c1 = U1 >= b
        live_rot[n] = int(c1.flatten(1).any(dim=1).sum())
        cand_blocks[n] = int(c1.sum())
        ks = torch.nonzero(c1.flatten(1).any(dim=1)).flatten()
        if len(ks):
            U2 = W2[prow[ks] >> 2, pcol[ks] >> 2].sum(dim=1, dtype=torch.int32).to(torch.int64) * scale   # (live, 21, 21)
            c2 = U2 >= b
            # the sub-blocks of candidate blocks only: sub-block (y4, x4) belongs to block (y4 >> 1, x4 >> 1)
            par = c1[ks].repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)[:, :NB4, :NB4]
            c2 = c2 & par
            live_sub[n] = int(c2.sum())
            pad4 = torch.zeros((len(ks), 2 * NB, 2 * NB), dtype=torch.bool, device=dev)
            pad4[:, :NB4, :NB4] = c2
            per_block = pad4.view(len(ks), NB, 2, NB, 2).sum(dim=(2, 4))
            whole_blocks[n] = int((per_block >= 3).sum())
        if n % 500 == 499:
            print("  %d / %d pairs, %.0f s" % (n + 1, len(sel), time.time() - t0), flush=True)

    d, st = bench._traffic_file()
    sq_b = d.get("csm_bnb_bounds_sq_per_launch_10000pairs_u16") or {}
    sq_c = d.get("csm_bnb_cand_sq_per_launch_10000pairs_u16") or {}
    k = n_pairs / float(len(sel))   # a sample is scaled to the list
    valu_per_rot = sq_b["SQ_INSTS_VALU"] / (10000.0 * 61)
    loads_per_block = sq_c["SQ_INSTS_VMEM_RD"] / float(lv["candidates_refined"])
    ideal_b = k * live_rot.sum() * valu_per_rot / bench.VALU_PEAK_WAVE_INSTR * 1e3
    ideal_c = k * cand_blocks.sum() * loads_per_block * CLK_PER_WAVE_LOAD / (CU * GHZ * 1e9) * 1e3
    q = lambda a: {"mean": float(a.mean()), "median": float(np.median(a)), "p90": float(np.percentile(a, 90)),
                   "p99": float(np.percentile(a, 99)), "max": int(a.max()), "sum": int(a.sum())}
    out = {
        "source": "tools/ideal_matcher.py on one MI355X: counts from the library's own bounding tables (torch), unit costs from "
                  "profiles/traffic.json (kernel sources %s%s) and profiles/r05_ubench_gather.txt"
                  % (d.get("kernel_source_hash"), ", STALE against this tree" if st["stale"] else ""),
        "workload": {"mode": "weak", "pairs": int(n_pairs), "scans": int(wl.n_scans), "per_target": int(wl.per_target),
                     "cell_bits": cell_bits, "pairs_counted": int(len(sel))},
        "at_the_final_best": {"live_rotations_per_pair": q(live_rot), "candidate_blocks_per_pair": q(cand_blocks),
                              "live_sub_blocks_per_pair": q(live_sub), "blocks_with_3_or_4_live_sub_blocks_per_pair": q(whole_blocks)},
        "shipped_kernels_on_the_same_list": {
            "candidate_blocks_refined_per_pair": lv["candidates_refined"] / float(n_pairs),
            "whole_blocks_evaluated_per_pair": lv["blocks_whole"] / float(n_pairs),
            "sub_blocks_evaluated_per_pair": lv["sub_blocks"] / float(n_pairs),
            "refined_over_ideal_candidate_blocks": lv["candidates_refined"] / max(k * float(cand_blocks.sum()), 1.0),
            "bounds_valu_instr_per_rotation": valu_per_rot, "candidates_wave_loads_per_refined_block": loads_per_block,
            "candidates_wave_loads_per_launch": sq_c["SQ_INSTS_VMEM_RD"]},
        "unit_costs": {"valu_peak_wave_instr_per_s": bench.VALU_PEAK_WAVE_INSTR, "clk_per_wave_load_and_cu": CLK_PER_WAVE_LOAD,
                       "cus": CU, "ghz": GHZ},
        "ideal_ms_bounds": ideal_b, "ideal_ms_candidates": ideal_c, "ideal_ms": ideal_b + ideal_c,
    }
    txt = json.dumps(out, indent=1)
    for dname in ("gpurun_out", "profiles"):
        pth = os.path.join(ROOT, dname)
        if os.path.isdir(pth):
            open(os.path.join(pth, "ideal_matcher.json"), "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
