set -o pipefail
mkdir -p gpurun_out
( while true; do sleep 45; date >> gpurun_out/r63.alive; done ) & alive=$!
trap "kill $alive 2>/dev/null" EXIT
for h in 2000000000 0; do
timeout -k 10 500 python scripts/run_c3.py 20000000 150000000 0.005 0 $h 0 55 auto > gpurun_out/r63_c4like_h$h.json 2> gpurun_out/r63_c4like.err || { tail -3 gpurun_out/r63_c4like.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r63_c4like_h$h.json')); print('hint=$h', d['times'], [round(x,1) for x in d['count_phases_ms']], d['table'], d.get('distinct_in_table'), d.get('good_kmers'))"
done
