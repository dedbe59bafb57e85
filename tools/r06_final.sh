#!/bin/bash
# round 6: the round's profiles (rocprofv3 passes -> traffic.json), the ideal-matcher model on this build's counters, the bench
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
bash tools/profile_round.sh r06 > $O/r06_profile_round.log 2>&1 || { tail -20 $O/r06_profile_round.log; exit 1; }
cp $O/prof_r06/traffic.json profiles/traffic.json
timeout -k 10 300 python3 tools/ideal_matcher.py all > $O/r06_ideal.log 2>&1 || { tail -20 $O/r06_ideal.log; exit 1; }
tail -4 $O/r06_ideal.log
timeout -k 10 500 python3 bench.py --steps 20 --warmup 3 > $O/r06_bench.out 2> $O/r06_bench.err || { tail -20 $O/r06_bench.err; exit 1; }
cp bench_details.json $O/r06_bench_details.json
tail -c 3500 $O/r06_bench.out
