set -o pipefail
mkdir -p gpurun_out
( while true; do sleep 45; date >> gpurun_out/r3e.alive; done ) & alive=$!
trap "kill $alive 2>/dev/null" EXIT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3e_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r3e_tests.log
[ $rc -eq 0 ] || exit $rc
for o in "graph_aligned=0" "graph_aligned=1" "graph_aligned=1,graph_load_pct=40" "graph_aligned=1,graph_load_pct=33"; do
timeout -k 10 300 python scripts/run_c3.py 50000000 4600000 0.005 0 0 0 31 auto $o > gpurun_out/r3e_c3.json 2> gpurun_out/r3e_c3.err || { tail -3 gpurun_out/r3e_c3.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r3e_c3.json')); print('$o', d['times'], {k: round(v,2) for k,v in d['build_stats']['phase_ms'].items()}, d['graph_built'], d['largest'])"
done
