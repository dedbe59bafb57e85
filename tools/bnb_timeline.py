#!/usr/bin/env python3
"""Timeline of the matcher's workgroups on the bench workload (GPU box): when each pair's workgroup started and
ended, how many were resident over time, how long the tail is."""
import json, os, sys
os.environ["NHIP_BNB_TIMELINE"] = os.environ["NHIP_BNB_INSTRUMENT"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from nautilus_amd import _lib, sharding
lib = _lib.load()
wl = bench.Workload("weak", 1, 1000, 10)
plan = sharding.ShardPlan(wl.src, wl.tgt, wl.th0, 1)
BITS = int(sys.argv[1]) if len(sys.argv) > 1 else 16
m = bench.HipMatcher(wl, plan.shard(0), torch.device("cuda", 0), BITS)
for _ in range(2):
    m.step()
torch.cuda.synchronize()
raw = np.zeros(4 * m.n_pairs + 2, dtype=np.uint64)
_lib.check(lib.nhip_bnb_timeline(_lib.ptr(raw), m.n_pairs))
t = raw[:-2].reshape(m.n_pairs, 4).copy()
k2 = raw[-2:].astype(np.int64)
hw = (t[:, 3] >> np.uint64(48)).astype(np.int64)
t[:, 3] &= np.uint64(0xffffffffffff)
t = t.astype(np.int64)
t0 = t[:, 0].min()
us = (t - t0) / 100.0
dur = us[:, 3] - us[:, 0]
end = us[:, 3].max()
out = {"cell_bits": BITS, "lib": os.path.basename(_lib.LIB_PATH), "kernel_us": float(end), "wg_us_mean": float(dur.mean()), "wg_us_p50_p90_p99_max": [float(x) for x in np.percentile(dur, [50, 90, 99, 100])],
       "bounds_us_mean": float((us[:, 1] - us[:, 0]).mean()), "seeds_us_mean": float((us[:, 2] - us[:, 1]).mean()),
       "phase3_us_mean": float((us[:, 3] - us[:, 2]).mean()),
       "sum_wg_us_over_512_slots": float(dur.sum() / 512), "env": {k_: v_ for k_, v_ in os.environ.items() if k_.startswith("NHIP_BNB")}}
# residency over time
grid = np.linspace(0, end, 41)
out["resident_wgs_over_time"] = [int(((us[:, 0] <= g) & (us[:, 3] > g)).sum()) for g in grid]
# when did the last workgroup START, and how long were the ten last to finish
order = np.argsort(us[:, 3])[-3:]
out["last_3_to_finish"] = [{"pair": int(i), "start_us": float(us[i, 0]), "end_us": float(us[i, 3]), "phase3_us": float(us[i, 3] - us[i, 2])} for i in order]
out["last_start_us"] = float(us[:, 0].max())
out["second_kernel_us"] = [float((k2[0] - t0) / 100.0), float((k2[1] - t0) / 100.0)] if k2[1] else None
out["distinct_hw_ids"] = int(len(np.unique(hw)))
print(json.dumps(out, indent=1))
