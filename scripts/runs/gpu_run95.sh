set -o pipefail
mkdir -p gpurun_out
( while true; do sleep 45; date >> gpurun_out/r95.alive; done ) & alive=$!
trap "kill $alive 2>/dev/null" EXIT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r95_tests.log 2>&1; rc=$?
tail -2 gpurun_out/r95_tests.log
[ $rc -eq 0 ] || exit $rc
GK_MIN_LNB1=10 timeout -k 10 900 python -m pytest tests/test_table_gpu.py tests/test_coverage_gpu.py tests/test_configs_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q > gpurun_out/r95_tests_lnb10.log 2>&1; rc=$?
tail -2 gpurun_out/r95_tests_lnb10.log
[ $rc -eq 0 ] || exit $rc
GK_FUZZ_EXAMPLES=1500 timeout -k 10 600 python -m pytest tests/test_fuzz_gpu.py -m gpu -x -q > gpurun_out/r95_fuzz.log 2>&1; rc=$?
tail -1 gpurun_out/r95_fuzz.log
exit $rc
