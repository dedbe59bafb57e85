#!/bin/bash
# round 6, last GPU call: GPU suite, smoke, the fine-level probe, the round's profiles, the ideal-matcher model, the bench
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
timeout -k 10 700 python3 -m pytest tests -m gpu -q > $O/r06_gputests_f.log 2>&1; tail -3 $O/r06_gputests_f.log
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 500 python3 tools/dropin_fine_probe.py > $O/r06_dropin_fine_probe2.txt 2>&1; tail -14 $O/r06_dropin_fine_probe2.txt
bash tools/profile_round.sh r06 > $O/r06_profile_round.log 2>&1 || { tail -20 $O/r06_profile_round.log; exit 1; }
cp $O/prof_r06/traffic.json profiles/traffic.json
timeout -k 10 300 python3 tools/ideal_matcher.py all > $O/r06_ideal.log 2>&1 || { tail -20 $O/r06_ideal.log; exit 1; }
tail -4 $O/r06_ideal.log
timeout -k 10 500 python3 bench.py --steps 20 --warmup 3 > $O/r06_bench.out 2> $O/r06_bench.err || { tail -20 $O/r06_bench.err; exit 1; }
cp bench_details.json $O/r06_bench_details.json
tail -c 3000 $O/r06_bench.out
