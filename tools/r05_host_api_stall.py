#!/usr/bin/env python3
"""Round 5: where do the 4 seconds of one host-buffer call in five go?  (BENCH_r04.json secondary.host_buffer_api.runs_s =
[0.0123, 0.0117, 4.491]; profiles/r04_bench.json [3.823, 0.0118, 0.0118]; gpurun_out/r04_bench.json [0.0125, 4.172, 0.0121].)

The leg = nhip_scans_upload + nhip_grids_build (hipMalloc + zero-fill of 12 GB of tables) + nhip_csm_match (328 MB of
workspace) + the two frees, three times in a row, AFTER the bench had run configs[3] on one GPU through torch's caching
allocator (130 GB, released with empty_cache()).  This script runs the same calls with every call split by phase
(nhip_host_phases) in four situations and prints one JSON line per run:
  A  fresh process, nothing else on the device
  B  after 130 GB were allocated through torch, used and released (empty_cache) -- the bench's situation
  C  the same with the handle's tables kept between runs (only the match is repeated)
  D  after B, with a hipDeviceSynchronize + a 1 s sleep before every run (does the driver's reclaim need time?)
Usage: r05_host_api_stall.py [runs per situation] [GB for situation B]"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

RUNS = int(sys.argv[1]) if len(sys.argv) > 1 else 8
BIG_GB = float(sys.argv[2]) if len(sys.argv) > 2 else 130.0


def main():
    import torch
    from nautilus_amd import _lib, csm, sharding
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    wl = bench.Workload("weak", 1)
    plan = sharding.ShardPlan(wl.src, wl.tgt, wl.th0, 1, None)
    shard = plan.shard(0)
    spec = csm.grid_spec(30.0, 0.05, 2.0, 1e-10, 40, cell_bits=16)
    search = csm.search_spec(61, 81, 81, np.radians(1.0), short_scans=True)

    def situation(tag, before=None, n=RUNS):
        for i in range(n):
            if before:
                before()
            t_work, t_tot, calls, hm, hs = bench.host_api_run(wl, shard, spec, search)
            free_b, tot_b = torch.cuda.mem_get_info()
            print(json.dumps({"situation": tag, "run": i, "work_s": round(t_work, 5), "with_frees_s": round(t_tot, 5), "calls": calls,
                              "device_free_GB": round(free_b / 1e9, 2)}), flush=True)

    situation("A fresh")
    # B: what leg_config4_one_gpu leaves behind
    t0 = time.perf_counter()
    big = [torch.empty(int(1e9), dtype=torch.uint8, device=dev) for _ in range(int(BIG_GB))]
    for b in big[::8]:
        b.fill_(1)
    torch.cuda.synchronize()
    del big
    torch.cuda.empty_cache()
    print(json.dumps({"situation": "B setup", "torch_alloc_use_release_s": round(time.perf_counter() - t0, 3)}), flush=True)
    situation("B after %d GB through torch" % int(BIG_GB))
    # C: tables kept
    st = csm.ScanTable(wl.xy, wl.off)
    gr = csm.LikelihoodGrids(st, shard[4], spec)
    ph = (C.c_double * 8)()
    for i in range(RUNS):
        t0 = time.perf_counter()
        csm.match_pairs(st, gr, shard[1], shard[5], shard[3], search)
        dt = time.perf_counter() - t0
        lib.nhip_host_phases(ph)
        print(json.dumps({"situation": "C tables kept, match only", "run": i, "match_s": round(dt, 5), "phases": [round(x, 6) for x in ph]}), flush=True)
    gr.close()
    st.close()

    def settle():
        torch.cuda.synchronize()
        time.sleep(1.0)
    situation("D settle 1 s before every run", settle, n=max(RUNS // 2, 3))


if __name__ == "__main__":
    main()
