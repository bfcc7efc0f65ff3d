set -o pipefail
mkdir -p gpurun_out
( while true; do sleep 45; date >> gpurun_out/r62.alive; done ) & alive=$!
trap "kill $alive 2>/dev/null" EXIT
for l in 0 9 10; do
timeout -k 10 300 python scripts/run_c3.py 50000000 4600000 0.005 0 520000000 0 31 auto min_lnb1=$l > gpurun_out/r62_c3_l$l.json 2> gpurun_out/r62_c3.err || { tail -3 gpurun_out/r62_c3.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r62_c3_l$l.json')); print('min_lnb1=$l', d['times'], [round(x,1) for x in d['count_phases_ms']], d['table'])"
done
