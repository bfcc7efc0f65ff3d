set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r2e_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -18 gpurun_out/r2e_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
bash scripts/prof_c3.sh r2e 50000000 4600000 0.005 0 0 0 31 auto > gpurun_out/r2e_prof.log 2>&1; tail -42 gpurun_out/r2e_prof.log | cut -c1-260
timeout -k 10 400 python bench.py --steps 10 > gpurun_out/r2e_bench.json 2> gpurun_out/r2e_bench.err; echo bench rc=$?
python - <<'PY'
import json
try:
    d=json.load(open("gpurun_out/r2e_bench.json"))
    print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["phases_ms"])
    print(json.dumps(d.get("mode_G"))[:700]); print(json.dumps(d.get("pcie_inclusive"))[:800]); print(json.dumps(d.get("c3"))[:3500])
except Exception as e:
    print("bench json:", e)
PY
tail -5 gpurun_out/r2e_bench.err
timeout -k 10 200 python bench.py --steps 10 --sharded --no-extras --no-cpu-baseline > gpurun_out/r2e_bench_sharded.json 2> gpurun_out/r2e_bench_sharded.err; echo sharded rc=$?; cut -c1-1500 gpurun_out/r2e_bench_sharded.json; tail -3 gpurun_out/r2e_bench_sharded.err
