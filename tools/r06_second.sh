#!/bin/bash
# round 6, second GPU call: (1) which commit raised the candidates kernel's L2 fetch (FETCH_SIZE per dispatch on every
# tree of build/bisect/, each running ITS OWN tools/bnb_quick.py), (2) heavy pairs first in the candidates' list,
# (3) the GPU test suite on the library as it is now
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
echo "# FETCH_SIZE (KB) per dispatch of the matcher's kernels, tools/bnb_quick.py of each tree under rocprofv3 --pmc FETCH_SIZE" > $O/r06_bisect_fetch.txt
for c in bb3f128 d30c226 00b43ed 3ab88f4 f71c5fc 61056a2 HEAD; do
  T=$R/build/bisect/$c; [ $c = HEAD ] && T=$R
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/bis_$c -- python3 $T/tools/bnb_quick.py > $O/r06_bisect_$c.log 2>&1 || { tail -5 $O/r06_bisect_$c.log; exit 1; }
  echo "== $c  $(git -C $R log -1 --format=%s $c 2>/dev/null | cut -c1-100)" >> $O/r06_bisect_fetch.txt
  grep "kernel_ms" $O/r06_bisect_$c.log >> $O/r06_bisect_fetch.txt
  python3 $R/tools/rocprof_summary.py $O/bis_$c --per-dispatch csm_bnb >> $O/r06_bisect_fetch.txt
  rm -rf $O/bis_$c
done
tail -30 $O/r06_bisect_fetch.txt
cd $R
echo "# NHIP_BNB_FRONT_MIN (candidates from which a pair goes to the front of its XCD's list), tools/bnb_quick.py" > $O/r06_front_min.txt
for f in 0 100 200 400 800 0; do
  echo "== FRONT_MIN $f" >> $O/r06_front_min.txt
  NHIP_BNB_FRONT_MIN=$f timeout -k 10 200 python3 tools/bnb_quick.py 2>&1 | grep kernel_ms >> $O/r06_front_min.txt || exit 1
done
cat $O/r06_front_min.txt
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $O/r06_gputests_a.log 2>&1; tail -5 $O/r06_gputests_a.log
