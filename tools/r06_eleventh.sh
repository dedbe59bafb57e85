#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
timeout -k 10 600 python3 -m pytest tests/test_csm_gpu.py tests/test_adapters_gpu.py tests/test_golden.py tests/test_slam_loop_gpu.py -m gpu -x -q 2>&1 | tail -6
timeout -k 10 300 python3 tools/dropin_probe.py 2>&1 | grep -v amdgpu.ids > $O/r06_dropin_probe_tiled.txt; cat $O/r06_dropin_probe_tiled.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/dt -- python3 $R/tools/dropin_trace.py > $O/r06_dropin_trace.log 2>&1
python3 $R/tools/trace_gaps.py $O/dt 24 > $O/r06_dropin_timeline3.txt; rm -rf $O/dt; cat $O/r06_dropin_timeline3.txt
