#!/bin/bash
# A/B of the headline pipeline's variants at C2 (mode U and G): one JSON line per variant under gpurun_out/ab_<tag>_*.json
tag=${1:-ab}
for v in "base" "p2_wide=1" "p4_wide=1" "p2_wide=1,p4_wide=1" "fine_exact=1" "fine_exact=1,p4_wide=1"; do
  for mode in U G; do
    opts=""; if [ "$v" != "base" ]; then for kv in ${v//,/ }; do opts="$opts --opt $kv"; done; fi
    python bench.py --steps 20 --warmup 3 --mode $mode --no-extras --no-cpu-baseline $opts > gpurun_out/ab_${tag}_${v//[=,]/_}_$mode.json 2> gpurun_out/ab_${tag}.err
    python - "$v" "$mode" gpurun_out/ab_${tag}_${v//[=,]/_}_$mode.json <<'PY'
import json,sys
d=json.load(open(sys.argv[3])); r=d["roofline"]
print(sys.argv[1], sys.argv[2], "ms/step %.3f" % d["ms_per_step"], "kernel %.3f" % r["kernel_ms"], {k:round(v,3) for k,v in (r["phases_ms"] or {}).items()})
PY
  done
done
