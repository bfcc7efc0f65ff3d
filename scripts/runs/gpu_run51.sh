set -o pipefail
mkdir -p gpurun_out
( while true; do sleep 45; date >> gpurun_out/r51.alive; done ) & alive=$!
trap "kill $alive 2>/dev/null" EXIT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r51_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r51_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py > gpurun_out/r51_bench_full.json 2> gpurun_out/r51_bench_full.err || { tail -5 gpurun_out/r51_bench_full.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r51_bench_full.json')); r=d['roofline']
print("headline", round(d['ms_per_step'],3), "frac", round(r['frac'],3), "G", d['mode_G']['ms_per_step'], "pcie", d['pcie_inclusive']['ms_per_step'], "cpu", d['cpu_baseline']['value'])
c=d['c3']; print({k:v for k,v in c.items() if k in ('count_s','filter_s','build_s','retain_s','times')})
PY
timeout -k 10 600 bash scripts/profile_round.sh v12 > gpurun_out/r51_profile.log 2>&1 || { tail -5 gpurun_out/r51_profile.log; exit 1; }
tail -2 gpurun_out/r51_profile.log
