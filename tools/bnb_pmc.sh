#!/bin/bash
# Counters of the scan matcher's kernels on the bench workload (tools/bnb_quick.py), one rocprofv3 --pmc pass per
# counter set (never together with a trace):   tools/bnb_pmc.sh <out-dir-under-gpurun_out>
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-bnb_pmc}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/tools/bnb_quick.py"
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
           "TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- $CMD > /dev/null 2> $OUT/p$i.log
  python3 $R/tools/rocprof_summary.py $OUT/p$i --pmc | grep -A9 "^csm_bnb" > $OUT/pmc_$i.txt || true
  rm -rf $OUT/p$i
done
cat $OUT/pmc_*.txt
