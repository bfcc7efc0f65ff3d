#!/bin/bash
# PMC passes (separate, per the guide) over the whole C3 flow: FETCH_SIZE, WRITE_SIZE, TCC_EA0 request counts per kernel.
# usage (GPU box): bash scripts/prof_c3_pmc.sh <tag>     -> gpurun_out/<tag>_c3pmc_{fetch,write,req}/ ; summarise with
# scripts/summarize_c3_pmc.py <tag> (needs the kernel stats of the same flow: gpurun_out/<tag>_c3prof/ or prof_c3.sh's)
set -o pipefail
tag=${1:-c3}
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
( while true; do sleep 45; date >> $R/gpurun_out/${tag}_c3pmc.alive; done ) & alive=$!
trap "kill $alive 2>/dev/null" EXIT
cd /tmp
args="50000000 4600000 0.005 0 0 0 31 auto"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_c3prof -- python3 $R/scripts/run_c3.py $args > $R/gpurun_out/${tag}_c3prof.json 2> $R/gpurun_out/${tag}_c3prof.err || exit 1
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_c3pmc_fetch -- python3 $R/scripts/run_c3.py $args > $R/gpurun_out/${tag}_c3pmc_fetch.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_c3pmc_write -- python3 $R/scripts/run_c3.py $args > $R/gpurun_out/${tag}_c3pmc_write.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_c3pmc_req -- python3 $R/scripts/run_c3.py $args > $R/gpurun_out/${tag}_c3pmc_req.log 2>&1 || exit 1
echo done
