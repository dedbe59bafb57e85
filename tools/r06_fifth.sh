#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
for e in bnb every_add; do
  echo "== NHIP_DROPIN_FINE=$e" >> $O/r06_dropin_fine.txt
  NHIP_TUNABLES=1 NHIP_DROPIN_FINE=$e timeout -k 10 300 python3 tools/dropin_probe.py 2>&1 | grep -v amdgpu.ids >> $O/r06_dropin_fine.txt
done
cat $O/r06_dropin_fine.txt
NHIP_BENCH_REHEARSAL=1 timeout -k 10 500 python3 bench.py --gpus 4 --scans 2000 --per-target 20 --steps 3 --warmup 1 > $O/r06_rehearsal_4ranks.out 2> $O/r06_rehearsal_4ranks.err || { tail -20 $O/r06_rehearsal_4ranks.err; exit 1; }
tail -c 3000 $O/r06_rehearsal_4ranks.out; cp bench_details.json $O/r06_rehearsal_4ranks_details.json
