set -o pipefail
mkdir -p gpurun_out
( while true; do sleep 45; date >> gpurun_out/r3j.alive; done ) & alive=$!
trap "kill $alive 2>/dev/null" EXIT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3j_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r3j_tests.log
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 400 python bench.py > gpurun_out/r3j_bench_full.json 2> gpurun_out/r3j_bench_full.err || exit 1
python -c "
import json; d=json.load(open('gpurun_out/r3j_bench_full.json')); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic']); p=d['pcie_inclusive']; print('modeG', d['mode_G']['ms_per_step'], 'pcie', p['ms_per_step']); print('c3', d['c3']['wall_ms']); print(d['c3']['roofline'].get('moved_by_pmc', {}).get('k_classify<1>'))"
