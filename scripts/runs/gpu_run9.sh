set -o pipefail
mkdir -p gpurun_out
( while true; do sleep 45; date >> gpurun_out/r2h.alive; done ) & alive=$!
trap "kill $alive 2>/dev/null" EXIT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2h_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r2h_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 120 python scripts/ubench_upload.py > gpurun_out/r2h_upload.txt 2>&1 || exit 1
cat gpurun_out/r2h_upload.txt
timeout -k 10 300 python scripts/run_c3.py 50000000 4600000 0.005 0 0 0 31 auto > gpurun_out/r2h_c3.json 2> gpurun_out/r2h_c3.err || exit 1
python -c "
import json; d=json.load(open('gpurun_out/r2h_c3.json')); print(d['times'], d['count_phases_ms'], d['table'], d['build_stats']['phase_ms'])"
timeout -k 10 200 python bench.py --sharded --no-extras --no-cpu-baseline > gpurun_out/r2h_sharded.json 2> gpurun_out/r2h_sharded.err || exit 1
python -c "
import json; d=json.load(open('gpurun_out/r2h_sharded.json')); print(d['ms_per_step'], d['per_rank_step_ms'])"
timeout -k 10 300 bash scripts/ab_headline2.sh r2h_ab || exit 1
GK_LIB_PATH=$PWD/genome_amd/variants/timers.so timeout -k 10 120 python scripts/run_timers.py > gpurun_out/r2h_timers_base.txt 2>&1 || exit 1
GK_LIB_PATH=$PWD/genome_amd/variants/timers.so timeout -k 10 120 python scripts/run_timers.py p2_sorted=1 > gpurun_out/r2h_timers_p2sorted.txt 2>&1 || exit 1
timeout -k 10 500 bash scripts/profile_round.sh v9 auto 2>&1 | tail -3
python scripts/summarize_pmc.py v9 "round 2 HEAD" 2>&1 | tail -30
