set -o pipefail
mkdir -p gpurun_out
( while true; do sleep 45; date >> gpurun_out/r2y.alive; done ) & alive=$!
trap "kill $alive 2>/dev/null" EXIT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2y_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r2y_tests.log
[ $rc -eq 0 ] || exit $rc
echo "== host path after a resident map (closed)"; python3 scripts/trace_host.py after_resident close | tail -3 || exit 1
timeout -k 10 300 python3 scripts/c3_two_pass.py || exit 1
timeout -k 10 400 python bench.py > gpurun_out/r2y_bench_full.json 2> gpurun_out/r2y_bench_full.err || exit 1
python -c "
import json; d=json.load(open('gpurun_out/r2y_bench_full.json')); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic']); print({k:d['mode_G'][k] for k in ('ms_per_step',)}, {k:d['pcie_inclusive'][k] for k in ('ms_per_step',)}); print(d['c3']['wall_ms'], d['c3']['graph']['build_phase_ms'], d['c3']['count']['phases_ms'])"
