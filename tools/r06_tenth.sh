#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
timeout -k 10 200 python3 tools/seeds_share.py 2>&1 | tail -2
timeout -k 10 600 python3 tools/r06_prune_study.py 500 > $O/r06_prune_study2.log 2>&1; python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r06_prune_study.json"))
for k,v in d["thresholds"].items():
    if "seed" in k or "final" in k: print(k, {a: round(b,3) for a,b in v.items()})
PY
for i in 1 2; do timeout -k 10 200 python3 tools/bnb_quick.py 2>&1 | grep kernel_ms | tr '\n' ' '; echo; done
