set -o pipefail
mkdir -p gpurun_out
run() { name=$1; lib=$2; shift 2
  GK_LIB_PATH=$lib timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline "$@" > gpurun_out/r92_$name.json 2>> gpurun_out/r92.err || return 1
  python -c "
import json; d=json.load(open('gpurun_out/r92_$name.json')); print('$name', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['roofline']['phases_ms'].items()})"
}
for i in 1 2; do run g24_G "" --mode G && run g48_G genome_amd/variants/p5g48.so --mode G && run g96_G genome_amd/variants/p5g96.so --mode G || exit 1; done
