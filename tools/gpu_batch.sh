#!/bin/bash
# One gpurun call's worth of checks on the scan matcher (run from the repo root on the GPU box):
#   tools/gpu_batch.sh <tag>      -> gpurun_out/<tag>_{csm_tests.log,sweep.log,bnb_ab.log,bnb_probe.json}
T=${1:-x}
O=gpurun_out
timeout -k 10 400 python -m pytest tests/test_csm_gpu.py -x -q --timeout 120 > $O/${T}_csm_tests.log 2>&1; tail -3 $O/${T}_csm_tests.log
timeout -k 10 200 python tools/parity_sweep.py 600 > $O/${T}_sweep.log 2>&1; tail -1 $O/${T}_sweep.log
timeout -k 10 200 python tools/bnb_ab.py 3 > $O/${T}_bnb_ab.log 2>&1; tail -1 $O/${T}_bnb_ab.log
timeout -k 10 200 python tools/bnb_probe.py > $O/${T}_bnb_probe.json 2>/dev/null
python3 - <<PY
import json
d = json.load(open("$O/${T}_bnb_probe.json"))
for k, v in d.items():
    if isinstance(v, dict) and "kernel_ms" in v and k.endswith(("debug0", "debug2", "debug1")):
        c = v["clk_per_pair"]
        print(k, "ms %.2f" % v["kernel_ms"], "grid %.2f" % v["grid_ms"],
              "bounds %.0fk seeds %.0fk slowest %.0fk | p3 %.0fk: org %.0fk strip %.0fk exact %.0fk | sub %.1f whole %.1f"
              % (c["clk_bounds"] / 1e3, c["clk_seeds"] / 1e3, c["clk_slowest_wave"] / 1e3, c["clk_wave_phase3"] / 1e3,
                 c["clk_origins"] / 1e3, c["clk_sub_bounds"] / 1e3, c["clk_exact"] / 1e3, v["sub_blocks_per_pair"], v["whole_per_pair"]))
PY
