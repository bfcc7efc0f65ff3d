set -o pipefail
mkdir -p gpurun_out
( while true; do sleep 45; date >> gpurun_out/r89.alive; done ) & alive=$!
trap "kill $alive 2>/dev/null" EXIT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r89_tests.log 2>&1; rc=$?
tail -2 gpurun_out/r89_tests.log
[ $rc -eq 0 ] || exit $rc
GK_MIN_LNB1=10 timeout -k 10 900 python -m pytest tests/test_table_gpu.py tests/test_coverage_gpu.py tests/test_configs_gpu.py -m gpu -x -q > gpurun_out/r89_tests_lnb10.log 2>&1; rc=$?
tail -2 gpurun_out/r89_tests_lnb10.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py > gpurun_out/r89_bench_full.json 2> gpurun_out/r89_bench_full.err || { tail -5 gpurun_out/r89_bench_full.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r89_bench_full.json')); r=d['roofline']
print("headline", round(d['ms_per_step'],3), "frac", round(r['frac'],3), r['phases_ms'], "G", round(d['mode_G']['ms_per_step'],3), "pcie", round(d['pcie_inclusive']['ms_per_step'],3), "c3 count", round(d['c3']['wall_ms']['count'],1))
PY
