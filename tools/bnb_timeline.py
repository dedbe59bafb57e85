#!/usr/bin/env python3
"""Timeline of the matcher's workgroups on the bench workload (GPU box): when each pair's workgroup started and
ended, how many were resident over time, how long the tail is."""
import json, os, sys
os.environ.setdefault("NHIP_TUNABLES", "1")  # (the library reads its switches only then)
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
hw = (t[:, 3] >> np.uint64(44)).astype(np.int64)   # cu | sh | se (8 bits), xcc (4 bits)
t[:, 3] &= np.uint64(0xfffffffffff)
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
out["distinct_cus"] = int(len(np.unique(hw)))
# per CU: workgroups resident over time (sampled), and the gap between one workgroup's end and the next one's start
# in the same "slot" (greedy assignment of a CU's workgroups to two slots in start order)
res_per_cu, gaps, busy = [], [], []
mid0, mid1 = 0.15 * end, 0.6 * end   # mid-launch: the chip is supposed to be full
for cu in np.unique(hw):
    sel = np.nonzero(hw == cu)[0]
    o = sel[np.argsort(us[sel, 0])]
    slots = []
    for i in o:
        placed = False
        for k in range(len(slots)):
            if slots[k] <= us[i, 0]:
                if mid0 < us[i, 0] < mid1:
                    gaps.append(us[i, 0] - slots[k])
                slots[k] = us[i, 3]
                placed = True
                break
        if not placed:
            slots.append(us[i, 3])
    res_per_cu.append(len(slots))
    g = np.linspace(mid0, mid1, 50)
    busy.append(np.mean([((us[sel, 0] <= x) & (us[sel, 3] > x)).sum() for x in g]))
out["max_concurrent_wgs_per_cu_hist"] = {int(k_): int(v_) for k_, v_ in zip(*np.unique(res_per_cu, return_counts=True))}
out["mean_resident_wgs_per_cu_mid_launch"] = float(np.mean(busy))
out["slot_gap_us_p10_p50_p90_mean"] = [float(x) for x in np.percentile(gaps, [10, 50, 90])] + [float(np.mean(gaps))] if gaps else None
print(json.dumps(out, indent=1))
