set -o pipefail
mkdir -p gpurun_out
( while true; do sleep 45; date >> gpurun_out/r80.alive; done ) & alive=$!
trap "kill $alive 2>/dev/null" EXIT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r80_tests.log 2>&1; rc=$?
tail -2 gpurun_out/r80_tests.log
[ $rc -eq 0 ] || exit $rc
V=genome_amd/variants/p2nofire.so
run() { name=$1; lib=$2; shift 2
  GK_LIB_PATH=$lib timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline "$@" > gpurun_out/r80_$name.json 2>> gpurun_out/r80.err || return 1
  python -c "
import json; d=json.load(open('gpurun_out/r80_$name.json')); print('$name', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['roofline']['phases_ms'].items()})"
}
run nofire_k55 $V --k 55 && run fire_k55 "" --k 55 && run nofire_k55b $V --k 55 && run fire_k55b "" --k 55 && run nofire_k63 $V --k 63 && run fire_k63 "" --k 63
GK_MIN_LNB1=10 timeout -k 10 900 python -m pytest tests/test_table_gpu.py tests/test_coverage_gpu.py -m gpu -x -q > gpurun_out/r80_tests_lnb10.log 2>&1; rc=$?
tail -2 gpurun_out/r80_tests_lnb10.log
exit $rc
