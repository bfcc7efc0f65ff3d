set -o pipefail
mkdir -p gpurun_out
( while true; do sleep 45; date >> gpurun_out/r78.alive; done ) & alive=$!
trap "kill $alive 2>/dev/null" EXIT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r78_tests.log 2>&1; rc=$?
tail -2 gpurun_out/r78_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r78_smoke.log 2>&1 || { tail -5 gpurun_out/r78_smoke.log; exit 1; }
tail -1 gpurun_out/r78_smoke.log
timeout -k 10 400 python bench.py > gpurun_out/r78_bench_full.json 2> gpurun_out/r78_bench_full.err || { tail -5 gpurun_out/r78_bench_full.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r78_bench_full.json')); r=d['roofline']
print("headline", round(d['ms_per_step'],3), "frac", round(r['frac'],3), "traffic", r['traffic'], "G", round(d['mode_G']['ms_per_step'],3), "pcie", round(d['pcie_inclusive']['ms_per_step'],3), "c3 count", round(d['c3']['wall_ms']['count'],1))
PY
