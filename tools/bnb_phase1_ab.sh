#!/bin/bash
export NHIP_TUNABLES=1  # (the library reads its environment switches only then)
# Bounds phase of the branch-and-bound matcher by part (instrumented build; NHIP_BNB_DEBUG timing experiments), for the
# product library and every build in build/variants (tools/bnb_variants.sh): kernel ms of the full run (0), bounds only
# (2), bounds without the reductions (26), without the gathers (27), without the run lists too (28), without the
# window origins too (29: the chunk loop's skeleton, the staging of the pooled table, the launch).
cd "$(dirname "$0")/.."
for lib in nautilus_amd/lib/libnautilus_hip.so build/variants/libbnb_*.so; do
  [ -f $lib ] || continue
  NHIP_LIB=$lib NHIP_PROBE_BITS=${BITS:-16} NHIP_PROBE_MODES=${MODES:-0,2,26,27,28,29} python3 tools/bnb_probe.py 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin)
print('$lib')
for k,v in d.items():
    if isinstance(v,dict) and 'kernel_ms' in v:
        c=v['clk_per_pair']
        print('  %-9s %.2f ms  bounds %.0fk clk (wave 0: staging %.0fk, rotations %.0fk, rest %.0fk)' % (k[3:], v['kernel_ms'], c['clk_bounds']/1e3, c['clk_origins']/1e3, c['clk_sub_bounds']/1e3, c['clk_exact']/1e3))
"
done
