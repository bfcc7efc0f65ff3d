set -o pipefail
mkdir -p gpurun_out
( while true; do sleep 45; date >> gpurun_out/r49.alive; done ) & alive=$!
trap "kill $alive 2>/dev/null" EXIT
run() {  # name, lib, options...
  name=$1; lib=$2; shift 2
  GK_LIB_PATH=$lib timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline "$@" > gpurun_out/r49_$name.json 2>> gpurun_out/r49.err || return 1
  python - $name gpurun_out/r49_$name.json <<'PY'
import json,sys
d=json.load(open(sys.argv[2])); r=d["roofline"]
print(sys.argv[1], "ms/step %.3f" % d["ms_per_step"], "kernel %.3f" % r["kernel_ms"], {k:round(v,3) for k,v in (r["phases_ms"] or {}).items()})
PY
}
V=genome_amd/variants/p5stride.so
run stride $V && run contig "" && run stride_G $V --mode G && run contig_G "" --mode G && run stride_k55 $V --k 55 && run contig_k55 "" --k 55 && run stride2 $V && run contig2 ""
timeout -k 10 300 python -m pytest tests/test_table_gpu.py tests/test_coverage_gpu.py -m gpu -x -q > gpurun_out/r49_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r49_tests.log
exit $rc
