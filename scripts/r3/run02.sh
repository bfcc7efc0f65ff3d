#!/bin/bash
# round 3, GPU call 2: the whole GPU suite (no -x), the replica rehearsal at C4's size, the bench line
set -o pipefail
mkdir -p gpurun_out
tag=t3b
md5sum genome_amd/libgenome_amd.so > gpurun_out/${tag}_so.md5
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/${tag}_tests.log 2>&1
rc=$?
tail -25 gpurun_out/${tag}_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "tests timed out"; exit $rc; fi
timeout -k 10 500 python scripts/rehearse_replica.py --k 55 --genome 1500000000 --coverage 10 --out gpurun_out/${tag}_replica_c4.json > gpurun_out/${tag}_replica_c4.log 2>&1
rr=$?
echo "replica rc=$rr"; tail -5 gpurun_out/${tag}_replica_c4.log
if [ $rr -eq 124 ] || [ $rr -eq 137 ]; then exit $rr; fi
timeout -k 10 300 python bench.py --steps 10 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "bench rc=$?"; head -c 2500 gpurun_out/${tag}_bench.json
exit $rc
