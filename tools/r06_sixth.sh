#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
timeout -k 10 500 python3 tools/dropin_fine_probe.py > $O/r06_dropin_fine_probe.txt 2>&1 || { tail -20 $O/r06_dropin_fine_probe.txt; exit 1; }
tail -75 $O/r06_dropin_fine_probe.txt
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $O/r06_gputests_c.log 2>&1; tail -5 $O/r06_gputests_c.log
