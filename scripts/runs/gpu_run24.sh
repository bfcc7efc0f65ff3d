set -o pipefail
mkdir -p gpurun_out
( while true; do sleep 45; date >> gpurun_out/r2w.alive; done ) & alive=$!
trap "kill $alive 2>/dev/null" EXIT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2w_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r2w_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 500 bash scripts/profile_round.sh v10 auto 2>&1 | tail -3
python scripts/summarize_pmc.py v10 "round 2: P4 prefetch, pinned status copies" 2>&1 | tail -24
timeout -k 10 200 python bench.py --sharded --no-extras --no-cpu-baseline > gpurun_out/r2w_sharded.json 2> gpurun_out/r2w_sharded.err || exit 1
python -c "
import json; d=json.load(open('gpurun_out/r2w_sharded.json')); print('sharded', d['ms_per_step'], d['per_rank_step_ms'])"
timeout -k 10 400 python bench.py > gpurun_out/r2w_bench_full.json 2> gpurun_out/r2w_bench_full.err || exit 1
python -c "
import json; d=json.load(open('gpurun_out/r2w_bench_full.json')); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic']); print({k:d['mode_G'][k] for k in ('ms_per_step',)}, {k:d['pcie_inclusive'][k] for k in ('ms_per_step',)}); print(d['c3']['wall_ms'], d['c3']['graph']['build_phase_ms']); print(d['cpu_baseline'])"
