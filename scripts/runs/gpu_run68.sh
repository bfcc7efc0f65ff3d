set -o pipefail
mkdir -p gpurun_out
( while true; do sleep 45; date >> gpurun_out/r68.alive; done ) & alive=$!
trap "kill $alive 2>/dev/null" EXIT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r68_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r68_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r68_smoke.log 2>&1 || { tail -5 gpurun_out/r68_smoke.log; exit 1; }
tail -1 gpurun_out/r68_smoke.log
timeout -k 10 400 python bench.py > gpurun_out/r68_bench_full.json 2> gpurun_out/r68_bench_full.err || { tail -5 gpurun_out/r68_bench_full.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r68_bench_full.json')); r=d['roofline']
print("headline", round(d['ms_per_step'],3), "frac", round(r['frac'],3), "G", round(d['mode_G']['ms_per_step'],3), "pcie", round(d['pcie_inclusive']['ms_per_step'],3), "cpu", d['cpu_baseline']['value'], r['measured_stream']['copy_GBps'])
print(d['c3']['wall_ms'])
PY
timeout -k 10 300 python bench.py --sharded --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r68_sharded.json 2> gpurun_out/r68_sharded.err || { tail -5 gpurun_out/r68_sharded.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r68_sharded.json')); print('sharded', round(d['ms_per_step'],3))"
timeout -k 10 600 bash scripts/profile_round.sh v13 > gpurun_out/r68_profile.log 2>&1 || { tail -5 gpurun_out/r68_profile.log; exit 1; }
echo profiled
