set -o pipefail
mkdir -p gpurun_out
( while true; do sleep 45; date >> gpurun_out/r3d.alive; done ) & alive=$!
trap "kill $alive 2>/dev/null" EXIT
for o in "graph_mem=0" "graph_mem=1" "graph_mem=2"; do
timeout -k 10 300 python scripts/run_c3.py 50000000 4600000 0.005 0 0 0 31 auto $o > gpurun_out/r3d_c3.json 2> gpurun_out/r3d_c3.err || { tail -3 gpurun_out/r3d_c3.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r3d_c3.json')); print('$o', d['times'], {k: round(v,2) for k,v in d['build_stats']['phase_ms'].items()}, d['graph_built'])"
done
