set -o pipefail
mkdir -p gpurun_out
( while true; do sleep 45; date >> gpurun_out/r3g.alive; done ) & alive=$!
trap "kill $alive 2>/dev/null" EXIT
timeout -k 10 600 python -m pytest tests/test_graph_gpu.py tests/test_configs_gpu.py tests/test_c3_gpu.py tests/test_host_cpp_gpu.py -m gpu -x -q > gpurun_out/r3g_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r3g_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python scripts/run_c3.py 50000000 4600000 0.005 0 0 0 31 auto > gpurun_out/r3g_c3.json 2> gpurun_out/r3g_c3.err || { tail -3 gpurun_out/r3g_c3.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r3g_c3.json')); print(d['times'], d['components'], d['largest'])"
