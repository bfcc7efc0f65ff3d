#!/bin/bash
# round 3, GPU call 3: GPU suite with the device paired-end walks, walk-pairs throughput, replica rehearsal (C4 size)
set -o pipefail
mkdir -p gpurun_out
tag=t3c
md5sum genome_amd/libgenome_amd.so > gpurun_out/${tag}_so.md5
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/${tag}_tests.log 2>&1
rc=$?
tail -25 gpurun_out/${tag}_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "tests timed out"; exit $rc; fi
timeout -k 10 300 python scripts/time_walk_pairs.py 4000000 > gpurun_out/${tag}_walk_pairs.log 2>&1
echo "walk_pairs rc=$?"; tail -4 gpurun_out/${tag}_walk_pairs.log
timeout -k 10 500 python scripts/rehearse_replica.py --k 55 --genome 1500000000 --coverage 10 --out gpurun_out/${tag}_replica_c4.json > gpurun_out/${tag}_replica_c4.log 2>&1
rr=$?
echo "replica rc=$rr"; tail -12 gpurun_out/${tag}_replica_c4.log
exit $rc
