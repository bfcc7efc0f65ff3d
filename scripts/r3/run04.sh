#!/bin/bash
# round 3, GPU call 4: suite with host-fed prefetch + component kernels; full bench line (pcie streaming leg, c3 host_fed leg)
set -o pipefail
mkdir -p gpurun_out
tag=t3d
md5sum genome_amd/libgenome_amd.so > gpurun_out/${tag}_so.md5
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/${tag}_tests.log 2>&1
rc=$?
tail -25 gpurun_out/${tag}_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "tests timed out"; exit $rc; fi
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "bench rc=$?"; tail -3 gpurun_out/${tag}_bench.err
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/t3d_bench.json'))
print('headline', d['ms_per_step'], d['roofline']['frac'], d['roofline']['phases_ms'])
print('pcie', {k:d['pcie_inclusive'].get(k) for k in ('ms_per_step','one_call_at_a_time_ms_per_step','phases_ms')})
c=d['c3']; print('c3 wall', c['wall_ms']); print('c3 host_fed', c.get('host_fed')); print(c['graph']['build_phase_ms'])
PY
exit $rc
