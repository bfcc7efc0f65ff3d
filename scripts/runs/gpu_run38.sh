set -o pipefail
mkdir -p gpurun_out
for hs in 1 1.15 1.3 1.5; do
  GK_HINT_SCALE=$hs timeout -k 10 100 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r3h_b.json 2>> gpurun_out/r3h.err || exit 1
  python - "$hs" <<'PY'
import json,sys
d=json.load(open('gpurun_out/r3h_b.json')); r=d["roofline"]
print("hint", sys.argv[1], "ms/step %.3f" % d["ms_per_step"], "kernel %.3f" % r["kernel_ms"], "slots", d["config"]["table_slots_per_gpu"], {k:round(v,3) for k,v in (r["phases_ms"] or {}).items()})
PY
done
