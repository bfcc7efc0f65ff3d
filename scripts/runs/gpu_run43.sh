set -o pipefail
mkdir -p gpurun_out
( while true; do sleep 45; date >> gpurun_out/r3l.alive; done ) & alive=$!
trap "kill $alive 2>/dev/null" EXIT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3l_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r3l_tests.log
[ $rc -eq 0 ] || exit $rc
for o in "filter_classic=1" "filter_classic=0"; do
timeout -k 10 300 python scripts/run_c3.py 50000000 4600000 0.005 0 0 0 31 auto $o > gpurun_out/r3l_c3.json 2> gpurun_out/r3l_c3.err || { tail -3 gpurun_out/r3l_c3.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r3l_c3.json')); print('$o', d['times'], d['good_kmers'], d['graph_built'], d['largest'], d['table']['slots'])"
done
