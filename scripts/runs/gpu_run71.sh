set -o pipefail
mkdir -p gpurun_out
( while true; do sleep 45; date >> gpurun_out/r71.alive; done ) & alive=$!
trap "kill $alive 2>/dev/null" EXIT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r71_tests.log 2>&1; rc=$?
tail -2 gpurun_out/r71_tests.log
[ $rc -eq 0 ] || exit $rc
for l in 10 9; do
GK_MIN_LNB1=$l timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r71_tests_lnb$l.log 2>&1; rc=$?
echo "== lnb1 >= $l rc=$rc"; grep -E "^E  |^FAILED|passed|failed|Fatal" gpurun_out/r71_tests_lnb$l.log | head -10
[ $rc -eq 0 ] || exit $rc
done
timeout -k 10 300 python bench.py --sharded --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r71_sharded.json 2> gpurun_out/r71_sharded.err || { tail -5 gpurun_out/r71_sharded.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r71_sharded.json')); print('sharded', round(d['ms_per_step'],3))"
