#!/bin/bash
# PMC passes for the memory side of the pipeline kernels (separate passes, --kernel-trace only).
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; tag=${1:-mem}
cd /tmp
i=0
for grp in "TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
           "TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_THRASHING_STALL_sum" \
           "TCC_WRITE_sum TCC_REQ_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_pmc_$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${tag}_$i.log 2>&1 || echo "pass $i failed"
done
ls $R/gpurun_out/ | grep ${tag}_pmc
