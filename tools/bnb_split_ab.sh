#!/bin/bash
export NHIP_TUNABLES=1  # (the library reads its environment switches only then)
# The split form of the branch-and-bound matcher on the 10,000-pair bench workload: kernel ms (u8, u16) by how the
# heaviest pairs are shared out (NHIP_BNB_SPLIT_MIN candidates per additional workgroup, NHIP_BNB_SPLIT_MAX workgroups
# per pair), by batch size, with and without the candidates of one batch running beside the bounds of the next.
cd "$(dirname "$0")/.."
run() { echo "$1: $(env $2 timeout -k 5 100 python3 tools/bnb_quick.py 2>/dev/null | tr '\n' ' ')"; }
run "fused" "NHIP_BNB_SPLIT=0"
ONE="NHIP_BNB_SPLIT_BATCH=100000"
for mx in 1 2 4 8 16 32; do run "one batch split_max=$mx" "$ONE NHIP_BNB_SPLIT_MAX=$mx"; done
for mn in 50 100 200 800 1600; do run "one batch split_min=$mn max=16" "$ONE NHIP_BNB_SPLIT_MIN=$mn NHIP_BNB_SPLIT_MAX=16"; done
run "batch=4096 overlap" "NHIP_BNB_SPLIT_BATCH=4096"
