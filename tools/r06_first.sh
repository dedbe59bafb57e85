#!/bin/bash
# round 6, first GPU call: the new bench line end to end, the ideal-matcher counts, the candidates kernel's traffic A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
true
true
true
timeout -k 10 300 python3 tools/ideal_matcher.py all > $O/r06_ideal.log 2>&1 || { tail -20 $O/r06_ideal.log; exit 1; }
tail -5 $O/r06_ideal.log
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  tag=$(echo $c | cut -d' ' -f1)
  timeout -k 10 240 rocprofv3 --pmc $c --output-format csv -d $O/ab_$tag -- python3 $R/tools/cand_traffic_ab.py > $O/r06_ab_$tag.log 2>&1 || { tail -5 $O/r06_ab_$tag.log; exit 1; }
  python3 $R/tools/rocprof_summary.py $O/ab_$tag --per-dispatch csm_bnb > $O/r06_ab_$tag.txt
  rm -rf $O/ab_$tag
done
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/ab_if -- python3 $R/tools/cand_traffic_ab.py image-first > $O/r06_ab_if.log 2>&1
python3 $R/tools/rocprof_summary.py $O/ab_if --per-dispatch csm_bnb > $O/r06_ab_image_first_FETCH_SIZE.txt
rm -rf $O/ab_if
cat $O/r06_ab_FETCH_SIZE.txt
