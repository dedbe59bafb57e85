#!/bin/bash
export NHIP_TUNABLES=1  # (the library reads its environment switches only then)
# hand-over policy of the branch-and-bound matcher on the 10,000-pair bench workload: kernel ms (u8, u16) by
# NHIP_BNB_KERNELS / NHIP_BNB_HEAVY_MIN / NHIP_BNB_KEEP_RANKS
echo "default: $(timeout -k 5 100 python tools/bnb_quick.py 2>/dev/null | tr '\n' ' ')"
for hm in 150 384 800 1500; do for kr in 4 8 16; do
  echo "kernels=2 heavy_min=$hm keep=$kr: $(NHIP_BNB_KERNELS=2 NHIP_BNB_HEAVY_MIN=$hm NHIP_BNB_KEEP_RANKS=$kr timeout -k 5 100 python tools/bnb_quick.py 2>/dev/null | tr '\n' ' ')"
done; done
