set -o pipefail
mkdir -p gpurun_out
( while true; do sleep 45; date >> gpurun_out/r2q.alive; done ) & alive=$!
trap "kill $alive 2>/dev/null" EXIT
GK_P45_STRIPES=4 timeout -k 10 400 python -m pytest tests/test_table_gpu.py tests/test_configs_gpu.py -m gpu -x -q > gpurun_out/r2q_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r2q_tests.log
[ $rc -eq 0 ] || exit $rc
one() {  # label, args
  label=$1; shift
  timeout -k 10 100 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline "$@" > gpurun_out/r2q_b.json 2>> gpurun_out/r2q.err || return 1
  python - "$label" <<'PY'
import json,sys
d=json.load(open('gpurun_out/r2q_b.json')); r=d["roofline"]
print(sys.argv[1], "ms/step %.3f" % d["ms_per_step"], "kernel %.3f" % r["kernel_ms"], "distinct", d["distinct_per_step"], {k:round(v,3) for k,v in (r["phases_ms"] or {}).items()})
PY
}
one "U base" --mode U || exit 1
for s in 2 4 8 16; do
  one "U stripes=$s" --mode U --opt p45_stripes=$s || exit 1
  one "U stripes=$s narrow P4" --mode U --opt p45_stripes=$s --opt p4_wide=0 || exit 1
done
one "G stripes=4" --mode G --opt p45_stripes=4 || exit 1
