set -o pipefail
mkdir -p gpurun_out
export GK_MIN_LNB1=${1:-10}
for f in ALL; do
  timeout -k 10 400 python -m pytest tests/$f.py -m gpu -q -x --deselect "tests/test_coverage_gpu.py::test_high_coverage_batch_sizes_table_for_distinct_keys" > gpurun_out/r56_$f.log 2>&1
  echo "== $f rc=$?"; grep -E "^E  |^FAILED|passed|failed|Error" gpurun_out/r56_$f.log | head -20
done
