set -o pipefail
mkdir -p gpurun_out
( while true; do sleep 45; date >> gpurun_out/r46.alive; done ) & alive=$!
trap "kill $alive 2>/dev/null" EXIT
timeout -k 10 600 python -m pytest tests/test_coverage_gpu.py tests/test_table_gpu.py tests/test_configs_gpu.py -m gpu -x -q > gpurun_out/r46_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r46_tests.log
[ $rc -eq 0 ] || exit $rc
run() {  # name, options...
  name=$1; shift
  timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/r46_$name.json 2>> gpurun_out/r46.err || return 1
  python - $name gpurun_out/r46_$name.json <<'PY'
import json,sys
d=json.load(open(sys.argv[2])); r=d["roofline"]
pc=d.get("pcie_inclusive") or {}
print(sys.argv[1], "ms/step %.3f" % d["ms_per_step"], "kernel %.3f" % r["kernel_ms"], {k:round(v,3) for k,v in (r["phases_ms"] or {}).items()}, "pcie", pc.get("ms_per_step"), "G", (d.get("mode_G") or {}).get("ms_per_step"))
PY
}
run p1 --no-c3 --opt p24_pieces=0 && run p2 --no-c3 --opt p24_pieces=2 && run p4 --no-c3 --opt p24_pieces=4 && run p8 --no-c3 --opt p24_pieces=8 && run p3 --no-c3 --opt p24_pieces=3
