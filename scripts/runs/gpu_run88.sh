set -o pipefail
mkdir -p gpurun_out
run() { name=$1; shift
  timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline "$@" > gpurun_out/r88_$name.json 2>> gpurun_out/r88.err || return 1
  python -c "
import json; d=json.load(open('gpurun_out/r88_$name.json')); print('$name', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['roofline']['phases_ms'].items()})"
}
run w1 && run w2 --opt p4_wide=2 && run w1b && run w2b --opt p4_wide=2 && run w1_G --mode G && run w2_G --mode G --opt p4_wide=2
