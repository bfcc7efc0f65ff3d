set -o pipefail
mkdir -p gpurun_out
( while true; do sleep 45; date >> gpurun_out/r47.alive; done ) & alive=$!
trap "kill $alive 2>/dev/null" EXIT
run() {  # name, options...
  name=$1; shift
  timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-c3 "$@" > gpurun_out/r47_$name.json 2>> gpurun_out/r47.err || return 1
  python - $name gpurun_out/r47_$name.json <<'PY'
import json,sys
d=json.load(open(sys.argv[2])); r=d["roofline"]
pc=d.get("pcie_inclusive") or {}
print(sys.argv[1], "ms/step %.3f" % d["ms_per_step"], "pcie %.3f" % pc.get("ms_per_step"), {k:round(v,3) for k,v in (pc["phases_ms"] or {}).items()})
PY
}
run off --opt p24_pieces=0 && run low_default && GK_AUX_PRIO_NORMAL=1 run normal_default && run off2 --opt p24_pieces=0 && run low_default2
