#!/bin/bash
# round 3, GPU call 1: full GPU suite on 24-byte Slot<2> + rank-indexed pointer jumping + chunked gather, then the replica rehearsal, then bench
set -o pipefail
mkdir -p gpurun_out
tag=r3a
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1
rc=$?
tail -15 gpurun_out/${tag}_tests.log
if [ $rc -ne 0 ]; then echo "tests rc=$rc"; exit $rc; fi
timeout -k 10 400 python scripts/rehearse_replica.py --k 55 --genome 1500000000 --coverage 10 --out gpurun_out/${tag}_replica_c4.json > gpurun_out/${tag}_replica_c4.log 2>&1
rr=$?
echo "replica rc=$rr"; tail -5 gpurun_out/${tag}_replica_c4.log
if [ $rr -eq 124 ] || [ $rr -eq 137 ]; then exit $rr; fi
timeout -k 10 300 python bench.py --steps 10 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "bench rc=$?"; head -c 2500 gpurun_out/${tag}_bench.json
