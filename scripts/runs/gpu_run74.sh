set -o pipefail
mkdir -p gpurun_out
( while true; do sleep 45; date >> gpurun_out/r74.alive; done ) & alive=$!
trap "kill $alive 2>/dev/null" EXIT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r74_tests.log 2>&1; rc=$?
tail -2 gpurun_out/r74_tests.log
[ $rc -eq 0 ] || exit $rc
for o in "dist_exchange_ahead=0" "dist_exchange_ahead=1"; do
timeout -k 10 200 python bench.py --sharded --steps 20 --warmup 4 --no-cpu-baseline --opt $o > gpurun_out/r74_sharded.json 2> gpurun_out/r74_sharded.err || { tail -5 gpurun_out/r74_sharded.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r74_sharded.json')); print('$o sharded', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['roofline']['phases_ms'].items()})"
done
timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r74_plain.json 2>/dev/null && python -c "
import json; d=json.load(open('gpurun_out/r74_plain.json')); print('plain', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['roofline']['phases_ms'].items()})"
