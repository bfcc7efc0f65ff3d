set -o pipefail
mkdir -p gpurun_out
( while true; do sleep 45; date >> gpurun_out/r54.alive; done ) & alive=$!
trap "kill $alive 2>/dev/null" EXIT
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r54_tests_default.log 2>&1; rc=$?
tail -3 gpurun_out/r54_tests_default.log
[ $rc -eq 0 ] || exit $rc
GK_MIN_LNB1=10 timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r54_tests_lnb10.log 2>&1; rc=$?
tail -12 gpurun_out/r54_tests_lnb10.log
[ $rc -eq 0 ] || exit $rc
GK_MIN_LNB1=9 timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r54_tests_lnb9.log 2>&1; rc=$?
tail -12 gpurun_out/r54_tests_lnb9.log
exit $rc
