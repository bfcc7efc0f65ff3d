set -o pipefail
mkdir -p gpurun_out
python scripts/ubench_upload.py 2>&1 | tee gpurun_out/r2g_upload.txt
python scripts/run_c3.py 50000000 4600000 0.005 0 0 0 31 auto > gpurun_out/r2g_c3.json 2> gpurun_out/r2g_c3.err; python -c "
import json; d=json.load(open('gpurun_out/r2g_c3.json')); print(d['times'], d['count_phases_ms'], d['table'], d['build_stats']['phase_ms'])"
( while true; do sleep 45; date >> gpurun_out/r2g.alive; done ) & alive=$!
bash scripts/profile_round.sh v9 auto 2>&1 | tail -3
kill $alive
python scripts/summarize_pmc.py v9 "round 2 HEAD: P4 with 8192-key chunks" 2>&1 | tail -30
