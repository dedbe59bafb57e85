#!/usr/bin/env python3
"""One rank's share of BASELINE configs[3] (1,000,000 pairs over WORLD ranks) on one MI355X, matched in rounds of the size
given by NHIP_BNB_SPLIT_BATCH (tools/share_rounds.sh loops over sizes): ms per step, bounds / candidates kernels, form.
  python tools/share_rounds.py WORLD [RANK]"""
import ctypes as C, json, os, sys, time
os.environ.setdefault("NHIP_TUNABLES", "1")  # (the library reads its switches only then)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from nautilus_amd import _lib, csm, sharding
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 0
lib = _lib.load()
wl = bench.Workload("config4", world)
w = sharding.predicted_pair_cost(wl.bag.odom, wl.src, wl.tgt)
plan = sharding.ShardPlan(wl.src, wl.tgt, wl.th0, world, w)
dev = torch.device("cuda", 0)
m = bench.HipMatcher(wl, plan.shard(rank), dev, 16)
for _ in range(2):
    m.step()
torch.cuda.synchronize()
ref = (m.records()[0].clone(), m.records()[1].clone())
for batch in [0] + [int(x) for x in os.environ.get("SHARE_BATCHES", "").split(",") if x]:
    if batch:
        os.environ["NHIP_BNB_SPLIT_BATCH"] = str(batch)
    else:
        os.environ.pop("NHIP_BNB_SPLIT_BATCH", None)
    # (the library read its environment once: a fresh process per setting is the clean way; the tunables are looked
    #  up per launch through getenv, so changing them here works as long as NHIP_TUNABLES was set at the first call)
    m.step(); torch.cuda.synchronize()
    lib.nhip_timing_reset(); lib.nhip_timing_enable(1)
    t0 = time.perf_counter()
    K = 5
    for _ in range(K):
        m.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    lib.nhip_timing_enable(0)
    ms, n = bench._timer(lib, _lib, _lib.NHIP_TIMER_CSM)
    mb, nb = bench._timer(lib, _lib, _lib.NHIP_TIMER_CSM_BOUNDS)
    mc, nc = bench._timer(lib, _lib, _lib.NHIP_TIMER_CSM_CAND)
    same = bool(torch.equal(m.records()[0], ref[0]) and torch.equal(m.records()[1], ref[1]))
    info = csm.last_launch()
    print(json.dumps({"world": world, "rank": rank, "pairs": m.n_pairs, "batch": batch or "default", "step_ms": round(1e3 * dt, 2),
                      "match_ms": round(ms / max(n, 1), 2), "bounds_ms_sum": round(mb / K, 2), "cand_ms_sum": round(mc / K, 2),
                      "Mpairs_per_s": round(m.n_pairs / dt / 1e6, 3), "form": info.get("form"), "rounds": info.get("rounds"),
                      "records_equal_default": same}), flush=True)
