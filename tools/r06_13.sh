#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
timeout -k 10 600 python3 -m pytest tests/test_csm_gpu.py tests/test_adapters_gpu.py tests/test_golden.py tests/test_slam_loop_gpu.py -m gpu -x -q 2>&1 | tail -4
timeout -k 10 300 python3 tools/dropin_probe.py 2>&1 | grep -v amdgpu.ids | head -8
NHIP_TUNABLES=1 NHIP_DROPIN_UNFUSED=1 timeout -k 10 300 python3 tools/dropin_probe.py 2>&1 | grep -v amdgpu.ids | sed -n 2,3p
