set -o pipefail
mkdir -p gpurun_out
( while true; do sleep 45; date >> gpurun_out/r58.alive; done ) & alive=$!
trap "kill $alive 2>/dev/null" EXIT
timeout -k 10 600 python -m pytest tests/test_configs_gpu.py -m gpu -q -x -k "beyond_34" > gpurun_out/r58_tests.log 2>&1; rc=$?
grep -E "^E  |^FAILED|passed|failed|Fatal" gpurun_out/r58_tests.log | head -20
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python scripts/time_big_table.py 16e6 31 > gpurun_out/r58_big31.json 2> gpurun_out/r58_big31.err || { tail -5 gpurun_out/r58_big31.err; exit 1; }
cat gpurun_out/r58_big31.json
