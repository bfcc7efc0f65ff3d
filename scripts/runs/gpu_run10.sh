set -o pipefail
mkdir -p gpurun_out
( while true; do sleep 45; date >> gpurun_out/r2i.alive; done ) & alive=$!
trap "kill $alive 2>/dev/null" EXIT
timeout -k 10 300 python -m pytest tests/test_table_gpu.py tests/test_configs_gpu.py -m gpu -x -q > gpurun_out/r2i_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r2i_tests.log
[ $rc -eq 0 ] || exit $rc
for mode in U G; do
for v in "" "--opt p4_wide=0" "--opt fine_exact=1" "--opt fine_exact=1 --opt p4_wide=0"; do
  timeout -k 10 100 python bench.py --steps 20 --warmup 3 --mode $mode --no-extras --no-cpu-baseline $v > gpurun_out/r2i_b.json 2>> gpurun_out/r2i.err || exit 1
  python - "$mode $v" <<'PY'
import json,sys
d=json.load(open('gpurun_out/r2i_b.json')); r=d["roofline"]
print(sys.argv[1], "ms/step %.3f" % d["ms_per_step"], "kernel %.3f" % r["kernel_ms"], {k:round(v,3) for k,v in (r["phases_ms"] or {}).items()})
PY
done; done
timeout -k 10 300 python scripts/run_c3.py 50000000 4600000 0.005 0 0 0 31 auto > gpurun_out/r2i_c3.json 2> gpurun_out/r2i_c3.err || exit 1
python -c "
import json; d=json.load(open('gpurun_out/r2i_c3.json')); print(d['times'], d['count_phases_ms'])"
GK_LIB_PATH=$PWD/genome_amd/variants/timers.so timeout -k 10 120 python scripts/run_timers.py > gpurun_out/r2i_timers.txt 2>&1 || exit 1
head -20 gpurun_out/r2i_timers.txt
