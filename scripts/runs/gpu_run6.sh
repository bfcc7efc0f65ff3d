set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=5 > gpurun_out/r2f_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -12 gpurun_out/r2f_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
bash scripts/ab_headline.sh r2f 2>&1 | tail -14
timeout -k 10 400 python bench.py --steps 10 --no-cpu-baseline > gpurun_out/r2f_bench.json 2> gpurun_out/r2f_bench.err; echo bench rc=$?
python - <<'PY'
import json
d=json.load(open("gpurun_out/r2f_bench.json"))
print({k:d[k] for k in ("value","ms_per_step")})
p=d["pcie_inclusive"]; print("pcie", p["ms_per_step"], p["kernel_ms"], p["phases_ms"])
c=d["c3"]; print("c3 wall", c["wall_ms"], "count", c["count"]["phases_ms"], "build", c["graph"]["build_phase_ms"])
PY
