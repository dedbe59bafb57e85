#!/usr/bin/env python3
"""Timeline of the matcher's SPLIT form on a list of a few thousand pairs (GPU box): when each pair's bounds workgroup and
its candidates' workgroups started and ended, what the tail of the candidates' launch is made of.
   tools/bnb_timeline_split.py <scans> [cell bits]      (pairs = 10 x scans; environment: the form under test)"""
import json, os, sys
os.environ.setdefault("NHIP_TUNABLES", "1")  # (the library reads its switches only then)
os.environ["NHIP_BNB_TIMELINE"] = os.environ["NHIP_BNB_INSTRUMENT"] = os.environ["NHIP_BNB_STATS"] = "1"
os.environ.setdefault("NHIP_BNB_SPLIT", "1")
os.environ.setdefault("NHIP_BNB_KERNELS", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from nautilus_amd import _lib, csm, sharding
lib = _lib.load()
SCANS = int(sys.argv[1]) if len(sys.argv) > 1 else 300
BITS = int(sys.argv[2]) if len(sys.argv) > 2 else 16
wl = bench.Workload("weak", 1, SCANS, 10)
plan = sharding.ShardPlan(wl.src, wl.tgt, wl.th0, 1)
m = bench.HipMatcher(wl, plan.shard(0), torch.device("cuda", 0), BITS)
for _ in range(2):
    m.step()
torch.cuda.synchronize()
n = m.n_pairs
raw = np.zeros(4 * n + 2, dtype=np.uint64)
_lib.check(lib.nhip_bnb_timeline(_lib.ptr(raw), n))
cand = np.zeros(2 * n, dtype=np.uint64)
_lib.check(lib.nhip_bnb_timeline_candidates(_lib.ptr(cand), n))
work = np.zeros(n, dtype=np.uint64)
_lib.check(lib.nhip_bnb_stats_per_pair(_lib.ptr(work), n))
t = raw[:-2].reshape(n, 4).copy()
t[:, 3] &= np.uint64(0xfffffffffff)
t = t.astype(np.int64)
t0 = t[:, 0].min()
b_us = (t - t0) / 100.0
c0, c1 = cand[:n].astype(np.int64), cand[n:].astype(np.int64)
has = c1 > 0
cs, ce = (c0 - t0) / 100.0, (c1 - t0) / 100.0
dur = np.where(has, ce - cs, 0.0)
end = float(ce[has].max()) if has.any() else float(b_us[:, 3].max())
w = work.astype(np.int64)
out = {"pairs": int(n), "cell_bits": BITS, "form": csm.last_launch(), "env": {k: v for k, v in os.environ.items() if k.startswith("NHIP_BNB")},
       "bounds_kernel_us": [0.0, float(b_us[:, 3].max())],
       "candidates_kernel_us": [float(cs[has].min()), end] if has.any() else None,
       "matcher_us_first_start_to_last_end": end,
       "pairs_with_candidates": int(has.sum()),
       "cand_pair_span_us_p50_p90_p99_max": [float(x) for x in np.percentile(dur[has], [50, 90, 99, 100])] if has.any() else None,
       "work_units_per_pair_p50_p90_p99_max": [float(x) for x in np.percentile(w, [50, 90, 99, 100])],
       "sum_cand_span_us": float(dur.sum())}
if has.any():
    grid = np.linspace(cs[has].min(), end, 33)
    out["pairs_in_flight_over_the_candidates_launch"] = [int(((cs <= g) & (ce > g) & has).sum()) for g in grid]
    last = np.argsort(np.where(has, ce, -1))[-8:][::-1]
    out["last_8_pairs_to_finish"] = [{"pair": int(i), "cand_start_us": float(cs[i]), "cand_end_us": float(ce[i]), "span_us": float(dur[i]),
                                      "work_units": int(w[i])} for i in last]
    heavy = np.argsort(w)[-8:][::-1]
    out["heaviest_8_pairs"] = [{"pair": int(i), "cand_start_us": float(cs[i]), "cand_end_us": float(ce[i]), "span_us": float(dur[i]),
                                "work_units": int(w[i])} for i in heavy]
    # when would the launch end without its k heaviest pairs?
    o = np.argsort(np.where(has, ce, -1))
    out["end_us_without_the_last_k_pairs"] = {str(k): float(ce[o[-1 - k]]) for k in (1, 2, 4, 8, 16, 32) if k < has.sum()}
print(json.dumps(out))
