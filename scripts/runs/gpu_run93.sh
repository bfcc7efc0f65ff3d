set -o pipefail
mkdir -p gpurun_out
run() { name=$1; shift
  timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline "$@" > gpurun_out/r93_$name.json 2>> gpurun_out/r93.err || return 1
  python -c "
import json; d=json.load(open('gpurun_out/r93_$name.json')); print('$name', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['roofline']['phases_ms'].items()})"
}
for i in 1 2; do run g2 && run g1 --opt p4_grid=1 || exit 1; done
