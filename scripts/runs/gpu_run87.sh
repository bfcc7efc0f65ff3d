set -o pipefail
mkdir -p gpurun_out
( while true; do sleep 45; date >> gpurun_out/r87.alive; done ) & alive=$!
trap "kill $alive 2>/dev/null" EXIT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r87_tests.log 2>&1; rc=$?
tail -2 gpurun_out/r87_tests.log
[ $rc -eq 0 ] || exit $rc
GK_MIN_LNB1=10 timeout -k 10 900 python -m pytest tests/test_table_gpu.py tests/test_coverage_gpu.py tests/test_c3_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q > gpurun_out/r87_tests_lnb10.log 2>&1; rc=$?
tail -2 gpurun_out/r87_tests_lnb10.log
[ $rc -eq 0 ] || exit $rc
for i in 1 2; do
timeout -k 10 300 python scripts/run_c3.py 50000000 4600000 0.005 0 0 0 31 auto > gpurun_out/r87_c3.json 2> gpurun_out/r87_c3.err || { tail -3 gpurun_out/r87_c3.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r87_c3.json')); print('c3', d['times'], [round(x,1) for x in d['count_phases_ms']], d['good_kmers'])"
done
timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline --opt fine_exact=1 > gpurun_out/r87_fe.json 2>/dev/null && python -c "
import json; d=json.load(open('gpurun_out/r87_fe.json')); print('C2 fine_exact=1', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['roofline']['phases_ms'].items()})"
