#!/bin/bash
# Matcher forms by batch size and pairs per target (bench.py legs off): match ms per step.
#   tools/bnb_size_ab.sh "<bench args>" "ENV=..." ...
cd "$(dirname "$0")/.."
args=$1; shift
for e in "$@"; do
  echo "[$args | $e]: $(env $e timeout -k 5 300 python3 bench.py $args --steps 2 --warmup 1 --cpu-seconds 0 --no-resid --no-drop-in 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms_per_step']; print('%.0f pairs, %.3f Mpairs/s, step %.2f ms, match %.2f (bounds %s, cand %s), grid %.2f' % (d['config']['pairs_total'], d['value']/1e6, d['ms_per_step'], k['csm_match'], k['of_which_bounds_and_seeds'], k['of_which_candidates'], k['grid_build']))")"
done
