set -o pipefail
mkdir -p gpurun_out
( while true; do sleep 45; date >> gpurun_out/r50.alive; done ) & alive=$!
trap "kill $alive 2>/dev/null" EXIT
timeout -k 10 400 python -m pytest tests/test_dist_gpu.py tests/test_host_cpp_gpu.py -m gpu -x -q > gpurun_out/r50_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r50_tests.log
[ $rc -eq 0 ] || exit $rc
for i in 1 2; do
timeout -k 10 200 python bench.py --sharded --steps 20 --warmup 4 --no-cpu-baseline > gpurun_out/r50_sharded$i.json 2> gpurun_out/r50_sharded.err || { tail -5 gpurun_out/r50_sharded.err; exit 1; }
python - $i <<'PY'
import json,sys
d=json.load(open(f'gpurun_out/r50_sharded{sys.argv[1]}.json'))
print("sharded ms/step %.3f" % d["ms_per_step"], {k:(round(v,3) if isinstance(v,float) else '') for k,v in d["per_rank_step_ms"].items() if k!='what'}, {k:round(v,3) for k,v in d["roofline"]["phases_ms"].items()})
PY
done
