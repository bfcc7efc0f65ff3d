#!/bin/bash
# One GPU call: bench line + rocprofv3 kernel stats + PMC passes (separate passes, per the guide) for
# the default bench workload.  Outputs under gpurun_out/<tag>_*; copy what matters into profiles/.
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; tag=${1:-r01}; path=${2:-auto}
python3 $R/bench.py --steps 10 --warmup 2 --no-extras --insert-path $path > $R/gpurun_out/${tag}_bench.json 2> $R/gpurun_out/${tag}_bench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_stats -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --insert-path $path > $R/gpurun_out/${tag}_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --insert-path $path > $R/gpurun_out/${tag}_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --insert-path $path > $R/gpurun_out/${tag}_pmc_write.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_pmc_req -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --insert-path $path > $R/gpurun_out/${tag}_pmc_req.log 2>&1
cat $R/gpurun_out/${tag}_bench.json | cut -c1-300
