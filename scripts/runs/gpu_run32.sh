set -o pipefail
mkdir -p gpurun_out
( while true; do sleep 45; date >> gpurun_out/r3b.alive; done ) & alive=$!
trap "kill $alive 2>/dev/null" EXIT
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=5 > gpurun_out/r3b_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r3b_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 500 bash scripts/profile_round.sh v11 auto 2>&1 | tail -2
python scripts/summarize_pmc.py v11 "round 2 final: P4 prefetch, pinned status copies, block pool" 2>&1 | tail -3
timeout -k 10 200 python bench.py --sharded --no-extras --no-cpu-baseline > gpurun_out/r3b_sharded.json 2> gpurun_out/r3b_sharded.err || exit 1
timeout -k 10 400 python bench.py > gpurun_out/r3b_bench_full.json 2> gpurun_out/r3b_bench_full.err || exit 1
python -c "
import json; d=json.load(open('gpurun_out/r3b_bench_full.json')); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic']); p=d['pcie_inclusive']; print('modeG', d['mode_G']['ms_per_step'], 'pcie', p['ms_per_step'], p['phases_ms']['k_part_scatter1']); print('c3', d['c3']['wall_ms'])
s=json.load(open('gpurun_out/r3b_sharded.json')); print('sharded', s['ms_per_step'], s['per_rank_step_ms'])"
export TMPDIR=/tmp; R=$PWD; cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3b_c3prof -- python3 $R/scripts/run_c3.py 50000000 4600000 0.005 0 0 0 31 auto > $R/gpurun_out/r3b_c3.json 2> $R/gpurun_out/r3b_c3.err || { tail -3 $R/gpurun_out/r3b_c3.err; exit 1; }
cd $R; python -c "
import json; d=json.load(open('gpurun_out/r3b_c3.json')); print(d['times'], d['count_phases_ms'], {k: round(v,2) for k,v in d['build_stats']['phase_ms'].items()})"
