set -o pipefail
mkdir -p gpurun_out
( while true; do sleep 45; date >> gpurun_out/r84.alive; done ) & alive=$!
trap "kill $alive 2>/dev/null" EXIT
GK_FUZZ_EXAMPLES=3000 timeout -k 10 1000 python -m pytest tests/test_fuzz_gpu.py -m gpu -x -q > gpurun_out/r84_fuzz_default.log 2>&1; rc=$?
tail -3 gpurun_out/r84_fuzz_default.log
[ $rc -eq 0 ] || exit $rc
GK_MIN_LNB1=10 GK_FUZZ_EXAMPLES=2000 timeout -k 10 1000 python -m pytest tests/test_fuzz_gpu.py -m gpu -x -q > gpurun_out/r84_fuzz_lnb10.log 2>&1; rc=$?
tail -3 gpurun_out/r84_fuzz_lnb10.log
exit $rc
