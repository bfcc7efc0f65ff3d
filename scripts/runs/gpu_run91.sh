set -o pipefail
mkdir -p gpurun_out
run() { name=$1; lib=$2; shift 2
  GK_LIB_PATH=$lib timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline "$@" > gpurun_out/r91_$name.json 2>> gpurun_out/r91.err || return 1
  python -c "
import json; d=json.load(open('gpurun_out/r91_$name.json')); print('$name', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['roofline']['phases_ms'].items()})"
}
V=genome_amd/variants/p2big.so
run base "" && run big1024 $V --opt p2_wide=1 && run big512 $V && run base2 "" && run big1024b $V --opt p2_wide=1
