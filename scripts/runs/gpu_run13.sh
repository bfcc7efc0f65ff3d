set -o pipefail
mkdir -p gpurun_out
( while true; do sleep 45; date >> gpurun_out/r2l.alive; done ) & alive=$!
trap "kill $alive 2>/dev/null" EXIT
timeout -k 10 400 python -m pytest tests/test_table_gpu.py tests/test_configs_gpu.py tests/test_coverage_gpu.py -m gpu -x -q > gpurun_out/r2l_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r2l_tests.log
[ $rc -eq 0 ] || exit $rc
one() {  # label, args
  label=$1; shift
  timeout -k 10 100 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline "$@" > gpurun_out/r2l_b.json 2>> gpurun_out/r2l.err || return 1
  python - "$label" <<'PY'
import json,sys
d=json.load(open('gpurun_out/r2l_b.json')); r=d["roofline"]
print(sys.argv[1], "ms/step %.3f" % d["ms_per_step"], "kernel %.3f" % r["kernel_ms"], {k:round(v,3) for k,v in (r["phases_ms"] or {}).items()})
PY
}
one "U" --mode U || exit 1
one "G" --mode G || exit 1
one "U k55" --mode U --k 55 || exit 1
export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r2l_trace -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $R/gpurun_out/r2l_trace.log 2>&1 || exit 1
cd $R; python scripts/trace_step.py gpurun_out/r2l_trace
