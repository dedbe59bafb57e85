#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/dtn -- python3 $R/tools/dropin_trace_new.py > $O/r06_dropin_trace_new.log 2>&1
python3 $R/tools/trace_gaps.py $O/dtn 40 > $O/r06_dropin_timeline_new_target.txt; rm -rf $O/dtn; cat $O/r06_dropin_timeline_new_target.txt
