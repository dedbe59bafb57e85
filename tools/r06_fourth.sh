#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/dt -- python3 $R/tools/dropin_trace.py > $O/r06_dropin_trace.log 2>&1 || { tail -5 $O/r06_dropin_trace.log; exit 1; }
python3 $R/tools/trace_gaps.py $O/dt 36 > $O/r06_dropin_timeline.txt; rm -rf $O/dt
cat $O/r06_dropin_timeline.txt
cd $R
echo "# NHIP_BNB_FRONT_MIN sweep, tools/bnb_quick.py (u16 line), two runs each" > $O/r06_front_min2.txt
for f in 0 40 70 100 150 0 40 70 100 150; do
  echo "FRONT_MIN $f: $(NHIP_BNB_FRONT_MIN=$f timeout -k 10 200 python3 tools/bnb_quick.py 2>&1 | grep 'u16 kernel_ms')" >> $O/r06_front_min2.txt
done
cat $O/r06_front_min2.txt
