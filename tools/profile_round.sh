#!/bin/bash
# Collect the round's profiles on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <tag>        e.g. r02
# Writes gpurun_out/prof_<tag>/{kernel_stats.txt,pmc_traffic.txt,pmc_sq.txt,pmc_l1.txt,bench_under_rocprof.json,traffic.json}.
# Counters are collected in their own passes (never together with --kernel-trace / --stats).
set -e
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# default bench legs on the 10k-pair workload: headline (branch and bound, 8-bit cells), 16-bit cells, exhaustive kernel,
# residual kernels; without the small launches of the single-pair and 200-scan legs (they would blur the averages)
BENCH="python3 $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-drop-in"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $BENCH > $OUT/bench_under_rocprof.json 2> $OUT/kt.log
python3 $R/tools/rocprof_summary.py $OUT/kt > $OUT/kernel_stats.txt
echo "# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of: $BENCH" > $OUT/pmc_traffic.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -- $BENCH > /dev/null 2> $OUT/pmc_$c.log
  python3 $R/tools/rocprof_summary.py $OUT/pmc_$c --pmc >> $OUT/pmc_traffic.txt
done
echo "# rocprofv3 --pmc (two SQ passes, one TCP/TCC pass, one TA pass) of: $BENCH" > $OUT/pmc_sq.txt
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq1 -- $BENCH > /dev/null 2> $OUT/sq1.log
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/sq2 -- $BENCH > /dev/null 2> $OUT/sq2.log
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/l1 -- $BENCH > /dev/null 2> $OUT/l1.log
rocprofv3 --pmc TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum --output-format csv -d $OUT/ta -- $BENCH > /dev/null 2> $OUT/ta.log
for p in sq1 sq2 l1 ta; do python3 $R/tools/rocprof_summary.py $OUT/$p --pmc | grep -A9 "csm_\|resid_lidar_kernel<0, true>\|corr_search_kernel\|resid_normal_eq_kernel\|grid_blur_kernel\|grid_pool_kernel" >> $OUT/pmc_sq.txt || true; done
python3 $R/tools/make_traffic_json.py $OUT > $OUT/traffic.json
rm -rf $OUT/kt $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/sq1 $OUT/sq2 $OUT/l1 $OUT/ta
ls -la $OUT
