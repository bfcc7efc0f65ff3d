set -o pipefail
mkdir -p gpurun_out
( while true; do sleep 45; date >> gpurun_out/r86.alive; done ) & alive=$!
trap "kill $alive 2>/dev/null" EXIT
for v in rc24 rc48 rc96 rc24 rc48 rc96; do
lib=""; [ $v != base ] && lib=genome_amd/variants/$v.so
GK_LIB_PATH=$lib timeout -k 10 300 python scripts/run_c3.py 50000000 4600000 0.005 0 0 0 31 auto > gpurun_out/r86_c3_$v.json 2> gpurun_out/r86_c3.err || { tail -3 gpurun_out/r86_c3.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r86_c3_$v.json')); print('$v', d['times']['count_s'], [round(x,1) for x in d['count_phases_ms']], d['good_kmers'])"
done
