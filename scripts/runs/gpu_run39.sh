set -o pipefail
mkdir -p gpurun_out
one() {  # label, args
  label=$1; shift
  timeout -k 10 100 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline "$@" > gpurun_out/r3i_b.json 2>> gpurun_out/r3i.err || return 1
  python - "$label" <<'PY'
import json,sys
d=json.load(open('gpurun_out/r3i_b.json')); r=d["roofline"]
print(sys.argv[1], "ms/step %.3f" % d["ms_per_step"], "kernel %.3f" % r["kernel_ms"], "distinct", d["distinct_per_step"], {k:round(v,3) for k,v in (r["phases_ms"] or {}).items()})
PY
}
one "U base" --mode U || exit 1
one "U p2_sorted" --mode U --opt p2_sorted=1 || exit 1
one "G p2_sorted" --mode G --opt p2_sorted=1 || exit 1
one "U k55 base" --mode U --k 55 || exit 1
one "U k55 p2_sorted" --mode U --k 55 --opt p2_sorted=1 || exit 1
GK_LIB_PATH=$PWD/genome_amd/variants/timers.so timeout -k 10 120 python scripts/run_timers.py p2_sorted=1 2>&1 | head -9
