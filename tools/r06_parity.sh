#!/bin/bash
# round 6: parity sweeps on the final sources + the GPU suite + smoke()
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
export NHIP_TUNABLES=1
S=$O/r06_parity_sweeps.txt
echo "Round 6 -- parity sweeps on the final sources (tools/parity_sweep.py), one MI355X" > $S
echo "  default: $(timeout -k 10 300 python3 tools/parity_sweep.py 1000 2>&1 | tail -1)" >> $S
echo "  split_rounds (NHIP_BNB_SPLIT=1 NHIP_BNB_SPLIT_BATCH=3 NHIP_BNB_SPLIT_MIN=20 NHIP_BNB_SPLIT_MAX=5): $(NHIP_BNB_SPLIT=1 NHIP_BNB_SPLIT_BATCH=3 NHIP_BNB_SPLIT_MIN=20 NHIP_BNB_SPLIT_MAX=5 timeout -k 10 300 python3 tools/parity_sweep.py 400 2>&1 | tail -1)" >> $S
echo "  split, heavy pairs first from 3 candidates (NHIP_BNB_SPLIT=1 NHIP_BNB_FRONT_MIN=3): $(NHIP_BNB_SPLIT=1 NHIP_BNB_FRONT_MIN=3 timeout -k 10 300 python3 tools/parity_sweep.py 300 2>&1 | tail -1)" >> $S
echo "  split, pair order throughout (NHIP_BNB_SPLIT=1 NHIP_BNB_FRONT_MIN=0): $(NHIP_BNB_SPLIT=1 NHIP_BNB_FRONT_MIN=0 timeout -k 10 300 python3 tools/parity_sweep.py 200 2>&1 | tail -1)" >> $S
echo "  queue (NHIP_BNB_QUEUE=1): $(NHIP_BNB_QUEUE=1 timeout -k 10 300 python3 tools/parity_sweep.py 200 2>&1 | tail -1)" >> $S
echo "  noimage (--no-image): $(timeout -k 10 300 python3 tools/parity_sweep.py 600 --no-image 2>&1 | tail -1)" >> $S
cat $S
unset NHIP_TUNABLES
timeout -k 10 700 python3 -m pytest tests -m gpu -q > $O/r06_gputests_e.log 2>&1; tail -4 $O/r06_gputests_e.log
echo "GPU suite: $(tail -1 $O/r06_gputests_e.log)" >> $S
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
