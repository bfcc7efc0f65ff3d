#!/bin/bash
# UTCL1 (per-CU TLB) counters of the pipeline kernels at two batch sizes (separate passes per size).
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp
for n in 1000000 4000000; do
  rocprofv3 --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum --kernel-trace --output-format csv -d $R/gpurun_out/tlb_$n -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --reads $n > $R/gpurun_out/tlb_$n.log 2>&1 || echo "pass $n failed"
done
