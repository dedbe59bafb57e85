#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for f in 0 64; do
  NHIP_BNB_FRONT_MIN=$f timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fm_$f -- python3 $R/tools/bnb_quick.py > $O/r06_fm_fetch_$f.log 2>&1 || exit 1
  echo "== NHIP_BNB_FRONT_MIN=$f" >> $O/r06_front_min_fetch.txt
  grep kernel_ms $O/r06_fm_fetch_$f.log >> $O/r06_front_min_fetch.txt
  python3 $R/tools/rocprof_summary.py $O/fm_$f --per-dispatch "csm_bnb_cand_kernel<2>" >> $O/r06_front_min_fetch.txt
  rm -rf $O/fm_$f
done
cat $O/r06_front_min_fetch.txt
