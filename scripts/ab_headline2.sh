#!/bin/bash
# A/B of the headline pipeline at C2 mode U: context options and the table's size hint.  One line per variant.
tag=${1:-ab2}
run() {  # name, hint scale, options...
  name=$1; hs=$2; shift 2
  GK_HINT_SCALE=$hs python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline "$@" > gpurun_out/${tag}_$name.json 2>> gpurun_out/${tag}.err || return 1
  python - $name gpurun_out/${tag}_$name.json <<'PY'
import json,sys
d=json.load(open(sys.argv[2])); r=d["roofline"]
print(sys.argv[1], "ms/step %.3f" % d["ms_per_step"], "kernel %.3f" % r["kernel_ms"], "slots", d["config"]["table_slots_per_gpu"], {k:round(v,3) for k,v in (r["phases_ms"] or {}).items()})
PY
}
run base 1 && run p2sorted 1 --opt p2_sorted=1 && run p2sorted_wide 1 --opt p2_sorted=1 --opt p2_wide=1 \
 && run hint085 0.85 && run hint080 0.80 && run hint080_p2sorted 0.80 --opt p2_sorted=1
