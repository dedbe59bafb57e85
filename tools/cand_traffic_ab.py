#!/usr/bin/env python3
"""The candidates kernel's HBM traffic by slot layout: the bench's pair list matched four times on
NHIP_GRID_NO_IMAGE slots (8.3 MB per target), then four times on slots with the row-major image (12.3 MB), 16-bit cells.
Run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc TCC_HIT_sum TCC_MISS_sum` and read the dispatches in order
(tools/rocprof_summary.py <dir> --per-dispatch csm_bnb_cand): the first four are the image-less slots."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from nautilus_amd import _lib, sharding
lib = _lib.load()
wl = bench.Workload("weak", 1)
plan = sharding.ShardPlan(wl.src, wl.tgt, wl.th0, 1, None)
order = [True, False] if len(sys.argv) < 2 or sys.argv[1] != "image-first" else [False, True]
for no_image in order:
    m = bench.HipMatcher(wl, plan.shard(0), torch.device("cuda", 0), 16, no_image=no_image)
    lib.nhip_timing_reset(); lib.nhip_timing_enable(1)
    for _ in range(4):
        m.step()
    torch.cuda.synchronize(); lib.nhip_timing_enable(0)
    mb, nb = bench._timer(lib, _lib, _lib.NHIP_TIMER_CSM_BOUNDS)
    mc, nc = bench._timer(lib, _lib, _lib.NHIP_TIMER_CSM_CAND)
    print("no_image %d slot_bytes %d bounds_ms %.3f cand_ms %.3f (avg of %d incl. the first)" % (no_image, m.layout.slot_bytes, mb / max(nb, 1), mc / max(nc, 1), nc), flush=True)
    m.free_grids()
    del m
