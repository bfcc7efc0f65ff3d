set -o pipefail
mkdir -p gpurun_out
( while true; do sleep 45; date >> gpurun_out/r61.alive; done ) & alive=$!
trap "kill $alive 2>/dev/null" EXIT
GK_MIN_LNB1=10 timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r61_tests_lnb10.log 2>&1; rc=$?
echo "== lnb1 >= 10 rc=$rc"; grep -E "^E  |^FAILED|passed|failed|Fatal" gpurun_out/r61_tests_lnb10.log | head -20
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python scripts/time_big_table.py 16e6 31 > gpurun_out/r61_big31.json 2> gpurun_out/r61_big31.err || { tail -5 gpurun_out/r61_big31.err; exit 1; }
cat gpurun_out/r61_big31.json
