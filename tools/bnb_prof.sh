#!/bin/bash
# PMC passes over the branch-and-bound matcher (tools/bnb_probe.py workload, u8 + u16, all debug levels)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/bnbprof_${1:-x}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P="python3 $R/tools/bnb_probe.py"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/p1 -- $P > /dev/null 2> $OUT/p1.log
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/p2 -- $P > /dev/null 2> $OUT/p2.log
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/p3 -- $P > /dev/null 2> $OUT/p3.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/p4 -- $P > /dev/null 2> $OUT/p4.log
for p in p1 p2 p3 p4; do python3 $R/tools/rocprof_summary.py $OUT/$p --pmc --per-dispatch csm_bnb > $OUT/$p.txt 2>&1; rm -rf $OUT/$p; done
tail -2 $OUT/*.log | head -20
