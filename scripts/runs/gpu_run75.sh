set -o pipefail
mkdir -p gpurun_out
for g in 16 8 4 2; do
GK_EXP_SKM_GROUP=$g timeout -k 10 200 python bench.py --sharded --steps 20 --warmup 4 --no-cpu-baseline --opt dist_exchange_ahead=0 > gpurun_out/r75_sharded.json 2> gpurun_out/r75_sharded.err || { tail -5 gpurun_out/r75_sharded.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r75_sharded.json')); print('group $g sharded', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['roofline']['phases_ms'].items()})"
done
