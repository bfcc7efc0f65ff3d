set -o pipefail
mkdir -p gpurun_out
one() {  # label, lib, args
  label=$1; lib=$2; shift 2
  GK_LIB_PATH=$lib timeout -k 10 100 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline "$@" > gpurun_out/r2p_b.json 2>> gpurun_out/r2p.err || return 1
  python - "$label" <<'PY'
import json,sys
d=json.load(open('gpurun_out/r2p_b.json')); r=d["roofline"]
print(sys.argv[1], "ms/step %.3f" % d["ms_per_step"], "kernel %.3f" % r["kernel_ms"], "distinct", d["distinct_per_step"], {k:round(v,3) for k,v in (r["phases_ms"] or {}).items()})
PY
}
one "U base" "" --mode U || exit 1
for v in stagger1 stagger2 cap5760 cap4800; do
  one "U $v" $PWD/genome_amd/variants/$v.so --mode U || exit 1
done
