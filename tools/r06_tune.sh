#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
echo "# third sweep: a second, heavier front group (NHIP_BNB_FRONT_MIN2)" >> $O/r06_tune.txt
for v in 0 300 600 1200 2400 0 600 1200; do
  echo "FRONT_MIN2 $v: $(NHIP_BNB_FRONT_MIN2=$v timeout -k 10 200 python3 tools/bnb_quick.py 2>&1 | grep 'kernel_ms' | tr '\n' ' ')" >> $O/r06_tune.txt
done
tail -9 $O/r06_tune.txt
