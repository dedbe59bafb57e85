#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
echo "# second sweep" >> $O/r06_tune.txt
for v in "300 8 64" "300 4 64" "300 3 64" "300 5 64" "300 6 64" "400 4 64" "250 4 64" "300 4 64" "300 8 64" "300 5 64" "400 5 64"; do
  set -- $v
  echo "SPLIT_MIN $1 SPLIT_MAX $2 FRONT_MIN $3: $(NHIP_BNB_SPLIT_MIN=$1 NHIP_BNB_SPLIT_MAX=$2 NHIP_BNB_FRONT_MIN=$3 timeout -k 10 200 python3 tools/bnb_quick.py 2>&1 | grep 'kernel_ms' | tr '\n' ' ')" >> $O/r06_tune.txt
done
tail -12 $O/r06_tune.txt
