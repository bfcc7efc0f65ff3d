set -o pipefail
mkdir -p gpurun_out
( while true; do sleep 45; date >> gpurun_out/r48.alive; done ) & alive=$!
trap "kill $alive 2>/dev/null" EXIT
timeout -k 10 300 python -m pytest tests/test_table_gpu.py -m gpu -x -q -k "stream_bench or abi or option" > gpurun_out/r48_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r48_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --no-cpu-baseline --no-c3 > gpurun_out/r48_bench.json 2> gpurun_out/r48_bench.err || { tail -5 gpurun_out/r48_bench.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r48_bench.json')); r=d['roofline']
print(d['ms_per_step'], r['frac'], r['measured_stream'], r.get('frac_of_measured_copy'), r.get('traffic_GBps'), r.get('traffic_over_measured_copy'))
PY
