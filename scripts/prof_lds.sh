#!/bin/bash
# PMC passes for the LDS / issue side of the pipeline kernels (separate passes, --kernel-trace only).
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; tag=${1:-lds}
cd /tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_pmc_a -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${tag}_a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_pmc_b -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${tag}_b.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_pmc_c -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${tag}_c.log 2>&1
ls $R/gpurun_out/${tag}_pmc_a/*/ | head
