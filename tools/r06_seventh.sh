#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
timeout -k 10 700 python3 -m pytest tests -m gpu -q > $O/r06_gputests_d.log 2>&1; tail -8 $O/r06_gputests_d.log
timeout -k 10 300 python3 tools/dropin_probe.py 2>&1 | grep -v amdgpu.ids > $O/r06_dropin_probe_hybrid.txt; cat $O/r06_dropin_probe_hybrid.txt
