set -o pipefail
mkdir -p gpurun_out
python scripts/run_c3.py 50000000 4600000 0.005 0 0 0 31 auto > gpurun_out/r2c_c3_wide.json 2> gpurun_out/r2c_c3_wide.err; echo "wide rc=$?"; cat gpurun_out/r2c_c3_wide.json
python scripts/run_c3.py 50000000 4600000 0.005 0 0 0 31 auto p4_wide=0 > gpurun_out/r2c_c3_narrow.json 2> gpurun_out/r2c_c3_narrow.err; echo "narrow rc=$?"; cat gpurun_out/r2c_c3_narrow.json
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/r2c_tests.log 2>&1; echo "tests rc=$?"; tail -25 gpurun_out/r2c_tests.log
