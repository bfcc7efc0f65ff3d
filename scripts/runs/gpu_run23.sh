set -o pipefail
mkdir -p gpurun_out
( while true; do sleep 45; date >> gpurun_out/r2v.alive; done ) & alive=$!
trap "kill $alive 2>/dev/null" EXIT
timeout -k 10 600 python -m pytest tests/test_graph_gpu.py tests/test_vmap_gpu.py tests/test_host_cpp_gpu.py tests/test_c3_gpu.py tests/test_dist_gpu.py -m gpu -x -q > gpurun_out/r2v_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r2v_tests.log
[ $rc -eq 0 ] || exit $rc
for o in "graph_walk_queue=0" "graph_walk_queue=1"; do
timeout -k 10 300 python scripts/run_c3.py 50000000 4600000 0.005 0 0 0 31 auto $o > gpurun_out/r2v_c3.json 2> gpurun_out/r2v_c3.err || exit 1
python -c "
import json; d=json.load(open('gpurun_out/r2v_c3.json')); print('$o', d['times'], {k: round(v,2) for k,v in d['build_stats']['phase_ms'].items()}, d['graph_built'], d['largest'])"
done
