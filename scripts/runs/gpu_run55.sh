set -o pipefail
mkdir -p gpurun_out
( while true; do sleep 45; date >> gpurun_out/r55.alive; done ) & alive=$!
trap "kill $alive 2>/dev/null" EXIT
GK_MIN_LNB1=10 timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r55_tests_lnb10.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r55_tests_lnb10.log | tail -30
