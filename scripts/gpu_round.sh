#!/bin/bash
# One gpurun call: GPU test suite, then the default bench line, then the C3 run.  A step that times out ends the call.
set -o pipefail
mkdir -p gpurun_out
tag=${1:-r2}
timeout -k 10 ${T_TESTS:-800} python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1
rc=$?
tail -5 gpurun_out/${tag}_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "tests timed out"; exit $rc; fi
timeout -k 10 300 python bench.py --steps 10 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
rb=$?
echo "bench rc=$rb"; head -c 1500 gpurun_out/${tag}_bench.json
if [ $rb -eq 124 ] || [ $rb -eq 137 ]; then exit $rb; fi
timeout -k 10 300 python scripts/run_c3.py 50000000 4600000 0.005 0 0 0 31 auto > gpurun_out/${tag}_c3.json 2> gpurun_out/${tag}_c3.err
echo "c3 rc=$?"; cat gpurun_out/${tag}_c3.json; tail -3 gpurun_out/${tag}_c3.err
exit $rc
