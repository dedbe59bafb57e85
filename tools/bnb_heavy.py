#!/usr/bin/env python3
"""The pairs with the most candidates of the bench workload, each matched alone (GPU box): how long does one such
pair take, with and without handing rotations over?  usage: bnb_heavy.py [top_n]"""
import json, os, sys
os.environ.setdefault("NHIP_TUNABLES", "1")  # (the library reads its switches only then)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from nautilus_amd import _lib, csm, sharding
lib = _lib.load()
top_n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
wl = bench.Workload("weak", 1, 1000, 10)
plan = sharding.ShardPlan(wl.src, wl.tgt, wl.th0, 1)
shard = plan.shard(0)
dev = torch.device("cuda", 0)
os.environ["NHIP_BNB_STATS"] = os.environ["NHIP_BNB_INSTRUMENT"] = "1"
m = bench.HipMatcher(wl, shard, dev, 8)
m.step(); torch.cuda.synchronize(); csm.bnb_stats()
m.step(); torch.cuda.synchronize()
per = np.zeros(m.n_pairs, dtype=np.uint64)
_lib.check(lib.nhip_bnb_stats_per_pair(_lib.ptr(per), m.n_pairs))
csm.bnb_stats()
os.environ.pop("NHIP_BNB_STATS"); os.environ.pop("NHIP_BNB_INSTRUMENT")
m.free_grids(); del m
order = np.argsort(per)[::-1]
idx, src, tgt, th0, ids, slot = shard
out = {"sub_block_units_top": [int(per[i]) for i in order[:top_n]], "median": float(np.median(per))}


def time_subset(sel, env, bits=8, reps=5):
    for k_ in ("NHIP_BNB_KERNELS", "NHIP_BNB_HEAVY_MIN", "NHIP_BNB_KEEP_RANKS"):
        os.environ.pop(k_, None)
    os.environ.update(env)
    ids2 = np.unique(tgt[sel]).astype(np.int32)
    sh = (idx[sel], src[sel], tgt[sel], th0[sel], ids2, np.searchsorted(ids2, tgt[sel]).astype(np.int32))
    mm = bench.HipMatcher(wl, sh, dev, bits)
    mm.step(); torch.cuda.synchronize()
    lib.nhip_timing_reset(); lib.nhip_timing_enable(1)
    for _ in range(reps):
        mm.step()
    torch.cuda.synchronize(); lib.nhip_timing_enable(0)
    ms, n = bench._timer(lib, _lib, _lib.NHIP_TIMER_CSM)
    mm.free_grids()
    return ms / n


modes = {"no_handover": {"NHIP_BNB_KERNELS": "1"}, "default": {}, "second_kernel_always": {"NHIP_BNB_KERNELS": "2"}}
for r, i in enumerate(order[:top_n]):
    sel = np.array([i])
    out["pair_rank%d_alone_ms" % r] = {k: time_subset(sel, e) for k, e in modes.items()}
med = order[len(order) // 2]
out["median_pair_alone_ms"] = {k: time_subset(np.array([med]), e) for k, e in modes.items()}
out["top%d_plus_500_others_ms" % top_n] = {k: time_subset(np.concatenate([order[:top_n], order[2000:2500]]), e) for k, e in modes.items()}
print(json.dumps(out, indent=1))
