set -o pipefail
mkdir -p gpurun_out
( while true; do sleep 45; date >> gpurun_out/r45.alive; done ) & alive=$!
trap "kill $alive 2>/dev/null" EXIT
run() {  # name, hint scale, lib
  name=$1; hs=$2; lib=$3; shift 3
  GK_LIB_PATH=$lib GK_HINT_SCALE=$hs timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline "$@" > gpurun_out/r45_$name.json 2>> gpurun_out/r45.err || return 1
  python - $name gpurun_out/r45_$name.json <<'PY'
import json,sys
d=json.load(open(sys.argv[2])); r=d["roofline"]
print(sys.argv[1], "ms/step %.3f" % d["ms_per_step"], "kernel %.3f" % r["kernel_ms"], "slots", d["config"]["table_slots_per_gpu"], {k:round(v,3) for k,v in (r["phases_ms"] or {}).items()})
PY
}
V=genome_amd/variants/p5rounds.so
run rounds_100 1 $V && run queue_100 1 "" && run rounds_085 0.85 $V && run queue_085 0.85 "" && run queue_080 0.80 "" && run queue_075 0.75 "" \
 && run rounds_100_G 1 $V --mode G && run queue_100_G 1 "" --mode G
