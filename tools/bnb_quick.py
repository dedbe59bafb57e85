#!/usr/bin/env python3
"""Kernel time of the branch-and-bound matcher on the bench workload under the environment given (GPU box)."""
import os, sys
os.environ.setdefault("NHIP_TUNABLES", "1")  # (the library reads its switches only then)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from nautilus_amd import _lib, sharding
if os.environ.get("NHIP_LIB_ALT"):  # (A/B experiments: a second build of the library)
    _lib.LIB_PATH = os.path.join(ROOT, "nautilus_amd", "lib", "alt_" + os.environ["NHIP_LIB_ALT"], "libnautilus_hip.so")
lib = _lib.load()
wl = bench.Workload("weak", 1, int(sys.argv[1]) if len(sys.argv) > 1 else 1000, int(sys.argv[2]) if len(sys.argv) > 2 else 10)
# NHIP_QUICK_ORDER=1: the pairs are launched heaviest first (sharding.predicted_pair_cost from the odometry poses)
w = sharding.predicted_pair_cost(wl.bag.odom, wl.src, wl.tgt) if os.environ.get("NHIP_QUICK_ORDER") == "1" else None
plan = sharding.ShardPlan(wl.src, wl.tgt, wl.th0, 1, w)
for bits in (8, 16):
    m = bench.HipMatcher(wl, plan.shard(0), torch.device("cuda", 0), bits, weights=plan.shard_weights(0))
    m.step(); torch.cuda.synchronize()
    lib.nhip_timing_reset(); lib.nhip_timing_enable(1)
    for _ in range(5):
        m.step()
    torch.cuda.synchronize(); lib.nhip_timing_enable(0)
    ms, n = bench._timer(lib, _lib, _lib.NHIP_TIMER_CSM)
    mb, nb = bench._timer(lib, _lib, _lib.NHIP_TIMER_CSM_BOUNDS)
    mc, nc = bench._timer(lib, _lib, _lib.NHIP_TIMER_CSM_CAND)
    import zlib
    crc = zlib.crc32(m.records()[0].cpu().numpy().tobytes()) ^ zlib.crc32(m.records()[1].cpu().numpy().tobytes())
    print("u%d kernel_ms %.3f bounds %.3f cand %.3f records_crc %08x" % (bits, ms / n, mb / max(nb, 1), mc / max(nc, 1), crc), flush=True)
    m.free_grids()
