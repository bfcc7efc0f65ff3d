set -o pipefail
mkdir -p gpurun_out
( while true; do sleep 45; date >> gpurun_out/r79.alive; done ) & alive=$!
trap "kill $alive 2>/dev/null" EXIT
V=genome_amd/variants/p4fire.so
GK_LIB_PATH=$V timeout -k 10 600 python -m pytest tests/test_table_gpu.py tests/test_coverage_gpu.py tests/test_configs_gpu.py -m gpu -x -q > gpurun_out/r79_tests.log 2>&1; rc=$?
tail -2 gpurun_out/r79_tests.log
[ $rc -eq 0 ] || exit $rc
run() { name=$1; lib=$2; shift 2
  GK_LIB_PATH=$lib timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline "$@" > gpurun_out/r79_$name.json 2>> gpurun_out/r79.err || return 1
  python -c "
import json; d=json.load(open('gpurun_out/r79_$name.json')); print('$name', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['roofline']['phases_ms'].items()})"
}
run base "" && run fire $V && run base2 "" && run fire2 $V && run base_G "" --mode G && run fire_G $V --mode G
