set -o pipefail
mkdir -p gpurun_out
one() {  # label, args
  label=$1; shift
  timeout -k 10 100 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline "$@" > gpurun_out/r2r_b.json 2>> gpurun_out/r2r.err || return 1
  python - "$label" <<'PY'
import json,sys
d=json.load(open('gpurun_out/r2r_b.json')); r=d["roofline"]
print(sys.argv[1], "ms/step %.3f" % d["ms_per_step"], "kernel %.3f" % r["kernel_ms"], {k:round(v,3) for k,v in (r["phases_ms"] or {}).items()})
PY
}
one "U base" --mode U || exit 1
one "U narrow grid1 (no stripes)" --mode U --opt p4_wide=0 --opt p4_grid=1 || exit 1
one "U wide grid1 (no stripes)" --mode U --opt p4_grid=1 || exit 1
for s in 2 4; do
  one "U stripes=$s narrow grid1" --mode U --opt p45_stripes=$s --opt p4_wide=0 --opt p4_grid=1 || exit 1
  one "U stripes=$s wide grid1" --mode U --opt p45_stripes=$s --opt p4_grid=1 || exit 1
  one "U stripes=$s narrow grid2" --mode U --opt p45_stripes=$s --opt p4_wide=0 --opt p4_grid=2 || exit 1
done
