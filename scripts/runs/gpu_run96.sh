set -o pipefail
mkdir -p gpurun_out
run() { name=$1; shift
  timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline "$@" > gpurun_out/r96_$name.json 2>> gpurun_out/r96.err || { tail -3 gpurun_out/r96.err; return 1; }
  python -c "
import json; d=json.load(open('gpurun_out/r96_$name.json')); print('$name', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['roofline']['phases_ms'].items()})"
}
for i in 1 2; do run k55_fe_narrow --k 55 --opt fine_exact=1 --opt p4_wide=0 && run k55_fe_wide --k 55 --opt fine_exact=1 --opt p4_wide=1 || exit 1; done
