set -o pipefail
mkdir -p gpurun_out
( while true; do sleep 45; date >> gpurun_out/r2u.alive; done ) & alive=$!
trap "kill $alive 2>/dev/null" EXIT
python3 scripts/trace_host.py | tail -2 || exit 1
for lp in 25 40 60 75; do
timeout -k 10 300 python scripts/run_c3.py 50000000 4600000 0.005 0 0 0 31 auto graph_load_pct=$lp > gpurun_out/r2u_c3_$lp.json 2> gpurun_out/r2u_c3.err || exit 1
python -c "
import json; d=json.load(open('gpurun_out/r2u_c3_$lp.json')); print($lp, d['times'], {k: round(v,2) for k,v in d['build_stats']['phase_ms'].items()})"
done
