#!/usr/bin/env python3
"""How well does the cost model balance the ranks?  Builds the weak-scaling workload of `world` ranks, splits it by
predicted cost and by pair count, and times every shard's matcher on this one GPU, one after the other: max / mean of
the per-shard match times is what the slowest rank costs the job.   tools/shard_balance.py [world]"""
import os, sys
os.environ.setdefault("NHIP_TUNABLES", "1")  # (the library reads its switches only then)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from nautilus_amd import _lib, sharding
world = int(sys.argv[1]) if len(sys.argv) > 1 else 4
lib = _lib.load()
wl = bench.Workload("weak", world)
dev = torch.device("cuda", 0)
for name, w in (("by predicted cost", sharding.predicted_pair_cost(wl.bag.odom, wl.src, wl.tgt)), ("by pair count", None)):
    plan = sharding.ShardPlan(wl.src, wl.tgt, wl.th0, world, w)
    ms, pairs = [], []
    for r in range(world):
        m = bench.HipMatcher(wl, plan.shard(r), dev, 16)
        m.step(); torch.cuda.synchronize()
        lib.nhip_timing_reset(); lib.nhip_timing_enable(1)
        for _ in range(3):
            m.step()
        torch.cuda.synchronize(); lib.nhip_timing_enable(0)
        k, n = bench._timer(lib, _lib, _lib.NHIP_TIMER_CSM)
        g, gn = bench._timer(lib, _lib, _lib.NHIP_TIMER_GRID)
        ms.append(k / n + g / gn); pairs.append(m.n_pairs)
        m.free_grids()
    ms = np.array(ms)
    print("%s: pairs %s, match+tables ms %s, max/mean %.3f" % (name, pairs, np.round(ms, 2).tolist(), ms.max() / ms.mean()), flush=True)
