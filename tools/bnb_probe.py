#!/usr/bin/env python3
"""Timing probe of the branch-and-bound matcher on the bench workload (GPU box): kernel ms with and without
its phases (NHIP_BNB_DEBUG), fraction of blocks evaluated exactly (NHIP_BNB_STATS)."""
import ctypes as C, json, math, os, sys, time
os.environ.setdefault("NHIP_TUNABLES", "1")  # (the library reads its switches only then)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from nautilus_amd import _lib, csm, sharding
os.environ["NHIP_BNB_STATS"] = os.environ["NHIP_BNB_INSTRUMENT"] = "1"  # the instrumented build of the kernels
lib = _lib.load()
wl = bench.Workload("weak", 1, int(sys.argv[1]) if len(sys.argv) > 1 else 1000, 10)
plan = sharding.ShardPlan(wl.src, wl.tgt, wl.th0, 1)
out = {}
MODES = os.environ.get("NHIP_PROBE_MODES", "0,L1,3,4,5,1,2").split(",")   # NHIP_BNB_DEBUG values; 26 / 27: bounds without reduction / gather
for bits in [int(b) for b in os.environ.get("NHIP_PROBE_BITS", "8,16").split(",")]:
    m = bench.HipMatcher(wl, plan.shard(0), torch.device("cuda", 0), bits)
    for dbg in MODES:
        os.environ["NHIP_BNB_LEVELS"] = "1" if dbg == "L1" else "2"   # L1: full run without the sub-block bounds
        os.environ["NHIP_BNB_DEBUG"] = "0" if dbg == "L1" else dbg
        m.step(); torch.cuda.synchronize(); csm.bnb_stats()
        lib.nhip_timing_reset(); lib.nhip_timing_enable(1)
        for _ in range(3):
            m.step()
        torch.cuda.synchronize(); lib.nhip_timing_enable(0)
        ms, n = bench._timer(lib, _lib, _lib.NHIP_TIMER_CSM)
        g_ms, g_n = bench._timer(lib, _lib, _lib.NHIP_TIMER_GRID)
        per = np.zeros(m.n_pairs, dtype=np.uint64)
        _lib.check(lib.nhip_bnb_stats_per_pair(_lib.ptr(per), m.n_pairs))
        lv = csm.bnb_stats_levels()
        ev, tot = lv["blocks_whole"] + lv["sub_blocks"] / 4, lv["blocks_total"]
        if dbg == "0":
            d = np.linalg.norm(wl.bag.truth[wl.src, :2] - wl.bag.truth[wl.tgt, :2], axis=1)
            q = np.percentile(per, [50, 90, 99, 99.9, 100])
            out["u%d_blocks_per_pair_percentiles_50_90_99_999_max" % bits] = [float(x) for x in q]
            out["u%d_blocks_by_distance" % bits] = {("%.0f-%.0fm" % (a, a + 1)): float(per[(d >= a) & (d < a + 1)].mean()) for a in range(4)}
            out["u%d_share_of_blocks_in_top_1pct_pairs" % bits] = float(np.sort(per)[-len(per) // 100:].sum() / max(per.sum(), 1))
        out["u%d_debug%s" % (bits, dbg)] = {"kernel_ms": ms / n, "grid_ms": g_ms / max(g_n, 1), "blocks_eval_per_pair": ev / 3 / m.n_pairs,
                                           "frac": ev / max(tot, 1), "whole_per_pair": lv["blocks_whole"] / 3 / m.n_pairs,
                                           "refined_per_pair": lv["candidates_refined"] / 3 / m.n_pairs,
                                           "sub_blocks_per_pair": lv["sub_blocks"] / 3 / m.n_pairs,
                                           "pose_evals16_per_pair": lv["pose_evals16"] / 3 / m.n_pairs,
                                           "clk_per_pair": {k_: v_ / 3 / m.n_pairs for k_, v_ in lv.items() if k_.startswith("clk_")},
                                           "pairs_handed_over": lv["pairs_handed_over"] / 3}
    m.free_grids()
os.environ.pop("NHIP_BNB_DEBUG")
os.environ.pop("NHIP_BNB_LEVELS")
print(json.dumps(out, indent=1))
