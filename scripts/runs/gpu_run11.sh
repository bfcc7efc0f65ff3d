set -o pipefail
mkdir -p gpurun_out
( while true; do sleep 45; date >> gpurun_out/r2j.alive; done ) & alive=$!
trap "kill $alive 2>/dev/null" EXIT
timeout -k 10 300 python -m pytest tests/test_table_gpu.py tests/test_configs_gpu.py -m gpu -x -q > gpurun_out/r2j_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r2j_tests.log
[ $rc -eq 0 ] || exit $rc
one() {  # label, lib, args
  label=$1; lib=$2; shift 2
  GK_LIB_PATH=$lib timeout -k 10 100 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline "$@" > gpurun_out/r2j_b.json 2>> gpurun_out/r2j.err || return 1
  python - "$label" <<'PY'
import json,sys
d=json.load(open('gpurun_out/r2j_b.json')); r=d["roofline"]
print(sys.argv[1], "ms/step %.3f" % d["ms_per_step"], "kernel %.3f" % r["kernel_ms"], {k:round(v,3) for k,v in (r["phases_ms"] or {}).items()})
PY
}
one "U base" "" --mode U || exit 1
one "G base" "" --mode G || exit 1
for v in cap4224 cap4224b cap2816; do
  one "U $v" $PWD/genome_amd/variants/$v.so --mode U || exit 1
done
one "U k55 base" "" --mode U --k 55 || exit 1
GK_LIB_PATH=$PWD/genome_amd/variants/timers.so timeout -k 10 120 python scripts/run_timers.py > gpurun_out/r2j_timers.txt 2>&1 || exit 1
head -9 gpurun_out/r2j_timers.txt
