set -o pipefail
export TMPDIR=/tmp; R=$PWD
( while true; do sleep 45; date >> $R/gpurun_out/r3c.alive; done ) & alive=$!
trap "kill $alive 2>/dev/null" EXIT
cd /tmp
timeout -k 10 400 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d $R/gpurun_out/v11_c3pmc_rdsz -- python3 $R/scripts/run_c3.py 50000000 4600000 0.005 0 0 0 31 auto > $R/gpurun_out/v11_c3pmc_rdsz.log 2>&1 || { tail -5 $R/gpurun_out/v11_c3pmc_rdsz.log; exit 1; }
timeout -k 10 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d $R/gpurun_out/v11_pmc_rdsz -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/v11_pmc_rdsz.log 2>&1 || { tail -5 $R/gpurun_out/v11_pmc_rdsz.log; exit 1; }
echo ok
