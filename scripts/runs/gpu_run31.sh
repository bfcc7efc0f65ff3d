set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp; R=$PWD
cd /tmp
for i in 1 2 3 4 5; do
timeout -k 10 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/r3a_host_$i -- python3 $R/scripts/trace_host.py > $R/gpurun_out/r3a_host_$i.log 2>&1 || { tail -5 $R/gpurun_out/r3a_host_$i.log; exit 1; }
grep "step 5" $R/gpurun_out/r3a_host_$i.log
done
