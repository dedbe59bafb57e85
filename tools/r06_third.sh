#!/bin/bash
# round 6, third GPU call: the slot-alignment fix (FETCH_SIZE + kernel ms, a few slot paddings), the prune study, the
# chained GetTransformation, the GPU tests
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "# tools/bnb_quick.py: line-aligned slots (HEAD), extra slot padding in units of 128 B (NHIP_GRID_SLOT_PAD), heavy pairs first (NHIP_BNB_FRONT_MIN)" > $O/r06_aligned.txt
for v in "0 0" "1 0" "3 0" "17 0" "0 0" "0 100" "0 800" "1 100"; do
  set -- $v
  echo "== SLOT_PAD $1 FRONT_MIN $2" >> $O/r06_aligned.txt
  NHIP_GRID_SLOT_PAD=$1 NHIP_BNB_FRONT_MIN=$2 timeout -k 10 200 python3 tools/bnb_quick.py 2>&1 | grep kernel_ms >> $O/r06_aligned.txt || exit 1
done
cat $O/r06_aligned.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/al_f -- python3 $R/tools/bnb_quick.py > $O/r06_aligned_fetch.log 2>&1 || exit 1
python3 $R/tools/rocprof_summary.py $O/al_f --per-dispatch csm_bnb > $O/r06_aligned_fetch.txt; rm -rf $O/al_f
grep "<2" $O/r06_aligned_fetch.txt | tail -6
cd $R
timeout -k 10 400 python3 tools/r06_prune_study.py 1500 > $O/r06_prune_study.log 2>&1 || { tail -20 $O/r06_prune_study.log; exit 1; }
tail -3 $O/r06_prune_study.log
timeout -k 10 300 python3 tools/dropin_probe.py > $O/r06_dropin_probe.txt 2>&1; tail -12 $O/r06_dropin_probe.txt
NHIP_TUNABLES=1 NHIP_DROPIN_CHAIN=0 timeout -k 10 300 python3 tools/dropin_probe.py > $O/r06_dropin_probe_unchained.txt 2>&1; tail -12 $O/r06_dropin_probe_unchained.txt
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $O/r06_gputests_b.log 2>&1; tail -5 $O/r06_gputests_b.log
