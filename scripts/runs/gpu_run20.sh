set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp; R=$PWD
python3 scripts/trace_host.py || exit 1
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/r2s_host -- python3 $R/scripts/trace_host.py > $R/gpurun_out/r2s_host.log 2>&1 || { tail -5 $R/gpurun_out/r2s_host.log; exit 1; }
cd $R; python3 scripts/timeline.py gpurun_out/r2s_host
