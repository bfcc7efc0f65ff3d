#!/bin/bash
# SQ / LDS counters of the headline pipeline's kernels (three separate --pmc passes, --kernel-trace only, as the guide prescribes):
# instructions issued by type, busy and wait cycles, LDS activity and bank conflicts.  usage (GPU box): bash scripts/prof_sq.sh <tag>
# -> gpurun_out/<tag>_sq_{a,b,c}/ ; scripts/summarize_sq.py <tag> prints per-kernel averages
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; tag=${1:-sq}
cd /tmp
B="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_sq_a -- $B > $R/gpurun_out/${tag}_sq_a.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_sq_b -- $B > $R/gpurun_out/${tag}_sq_b.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_sq_c -- $B > $R/gpurun_out/${tag}_sq_c.log 2>&1 || exit 1
echo done
