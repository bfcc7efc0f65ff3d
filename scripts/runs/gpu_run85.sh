set -o pipefail
mkdir -p gpurun_out
( while true; do sleep 45; date >> gpurun_out/r85.alive; done ) & alive=$!
trap "kill $alive 2>/dev/null" EXIT
for w in 1 2 1 2; do
timeout -k 10 300 python scripts/run_c3.py 50000000 4600000 0.005 0 0 0 31 auto p4_wide=$w > gpurun_out/r85_c3_w$w.json 2> gpurun_out/r85_c3.err || { tail -3 gpurun_out/r85_c3.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r85_c3_w$w.json')); print('p4_wide=$w', d['times']['count_s'], [round(x,1) for x in d['count_phases_ms']], d['good_kmers'], d['graph_built'])"
done
timeout -k 10 300 python -m pytest tests/test_c3_gpu.py tests/test_coverage_gpu.py -m gpu -x -q > gpurun_out/r85_tests.log 2>&1; rc=$?
tail -2 gpurun_out/r85_tests.log
exit $rc
