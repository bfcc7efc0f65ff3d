set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_dist_gpu.py -m gpu -x -q > gpurun_out/r53_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r53_tests.log
[ $rc -eq 0 ] || exit $rc
for v in default ahead0; do
opt=""; [ $v = ahead0 ] && opt="--opt dist_exchange_ahead=0"
timeout -k 10 200 python bench.py --sharded --steps 20 --warmup 4 --no-cpu-baseline $opt > gpurun_out/r53_sharded_$v.json 2> gpurun_out/r53_sharded.err || { tail -5 gpurun_out/r53_sharded.err; exit 1; }
python - $v <<'PY'
import json,sys
d=json.load(open(f'gpurun_out/r53_sharded_{sys.argv[1]}.json'))
print(sys.argv[1], "sharded ms/step %.3f" % d["ms_per_step"], {k:(round(v,3) if isinstance(v,float) else '') for k,v in d["per_rank_step_ms"].items() if k!='what'}, {k:round(v,3) for k,v in d["roofline"]["phases_ms"].items()})
PY
done
