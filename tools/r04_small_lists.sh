#!/bin/bash
# Lists of a few thousand pairs (what configs[4] and a real bag produce): forms and sharing policies.
#   tools/r04_small_lists.sh <tag>   -> gpurun_out/<tag>_small_lists.txt, <tag>_timeline_*.json
export NHIP_TUNABLES=1  # (the library reads its environment switches only then)
cd "$(dirname "$0")/.."
T=${1:-x}
O=gpurun_out
{
echo "## 3,000 pairs (bench.py --scans 300): fused / split with round 3's per-XCD work lists / split with the spread lists"
tools/bnb_size_ab.sh "--scans 300" "NHIP_BNB_SPLIT=0" "NHIP_BNB_SPLIT=1 NHIP_BNB_SPREAD=0" "NHIP_BNB_SPLIT=1" \
  "NHIP_BNB_SPLIT=1 NHIP_BNB_SPLIT_MAX=16" "NHIP_BNB_SPLIT=1 NHIP_BNB_SPLIT_MIN=150 NHIP_BNB_SPLIT_MAX=16" "NHIP_BNB_SPLIT=1 NHIP_BNB_SPLIT_MIN=600"
echo "## 1,000 / 1,500 / 2,000 / 4,500 pairs"
tools/bnb_size_ab.sh "--scans 100" "NHIP_BNB_SPLIT=0" "NHIP_BNB_SPLIT=0 NHIP_BNB_KERNELS=2" "NHIP_BNB_SPLIT=1 NHIP_BNB_KERNELS=1"
tools/bnb_size_ab.sh "--scans 150" "NHIP_BNB_SPLIT=0" "NHIP_BNB_SPLIT=1 NHIP_BNB_KERNELS=1"
tools/bnb_size_ab.sh "--scans 200" "NHIP_BNB_SPLIT=0" "NHIP_BNB_SPLIT=1"
tools/bnb_size_ab.sh "--scans 450" "NHIP_BNB_SPLIT=0" "NHIP_BNB_SPLIT=1 NHIP_BNB_SPREAD=0" "NHIP_BNB_SPLIT=1"
echo "## 10,000 pairs"
tools/bnb_size_ab.sh "--scans 1000" "NHIP_BNB_SPREAD=0" "A=1" "NHIP_BNB_SPLIT_MAX=16" "NHIP_BNB_SPLIT_MIN=150 NHIP_BNB_SPLIT_MAX=16"
echo "## 500 pairs (hand-over lists)"
tools/bnb_size_ab.sh "--scans 50" "A=1" "NHIP_BNB_KERNELS=1"
} > $O/${T}_small_lists.txt 2>&1
timeout -k 5 120 python3 tools/bnb_timeline_split.py 300 > $O/${T}_timeline_3000_split.json 2>$O/${T}_timeline.err
cat $O/${T}_small_lists.txt
