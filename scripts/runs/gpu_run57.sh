set -o pipefail
mkdir -p gpurun_out
( while true; do sleep 45; date >> gpurun_out/r57.alive; done ) & alive=$!
trap "kill $alive 2>/dev/null" EXIT
for l in 10 9; do
GK_MIN_LNB1=$l timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r57_tests_lnb$l.log 2>&1; rc=$?
echo "== lnb1 >= $l rc=$rc"; grep -E "^E  |^FAILED|passed|failed|Fatal" gpurun_out/r57_tests_lnb$l.log | head -20
[ $rc -eq 0 ] || exit $rc
done
