#!/bin/bash
# round 3, GPU call 5: suite; graph suites over the minimizer-bucketed table; C3 A/B of graph_mbt; bench (gated prefetch)
set -o pipefail
mkdir -p gpurun_out
tag=t3e
md5sum genome_amd/libgenome_amd.so > gpurun_out/${tag}_so.md5
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/${tag}_tests.log 2>&1
rc=$?
tail -8 gpurun_out/${tag}_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "tests timed out"; exit $rc; fi
GK_GRAPH_MBT=1 timeout -k 10 600 python -m pytest tests/test_graph_gpu.py tests/test_pairs_gpu.py tests/test_vmap_gpu.py tests/test_fuzz_gpu.py tests/test_dist_gpu.py -m gpu -q > gpurun_out/${tag}_tests_mbt.log 2>&1
rm=$?
echo "mbt tests rc=$rm"; tail -8 gpurun_out/${tag}_tests_mbt.log
if [ $rm -eq 124 ] || [ $rm -eq 137 ]; then exit $rm; fi
for o in "graph_mbt=0" "graph_mbt=1" "graph_mbt=1,graph_mbt_keys=64" "graph_mbt=1,graph_mbt_keys=1024"; do
  timeout -k 10 200 python scripts/run_c3.py 50000000 4600000 0.005 0 0 0 31 auto "$o" > gpurun_out/${tag}_c3_${o//[=,]/_}.json 2> gpurun_out/${tag}_c3.err
  r=$?; echo "c3 $o rc=$r"
  python3 -c "
import json,sys
d=json.load(open('gpurun_out/${tag}_c3_${o//[=,]/_}.json'))
print(d['times'], d['build_stats'], d['graph_built'], d['largest'])
" || tail -3 gpurun_out/${tag}_c3.err
  if [ $r -eq 124 ] || [ $r -eq 137 ]; then exit $r; fi
done
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "bench rc=$?"; tail -3 gpurun_out/${tag}_bench.err
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/t3e_bench.json'))
print('headline', d['ms_per_step'], d['roofline']['frac'], d['roofline']['phases_ms'])
print('pcie', {k:d['pcie_inclusive'].get(k) for k in ('ms_per_step','one_call_at_a_time_ms_per_step','phases_ms')})
c=d['c3']; print('c3 wall', c['wall_ms']); print('c3 host_fed', c.get('host_fed')); print(c['graph']['build_phase_ms'])
PY
exit $rc
