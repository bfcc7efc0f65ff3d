set -o pipefail
mkdir -p gpurun_out
( while true; do sleep 45; date >> gpurun_out/r2z.alive; done ) & alive=$!
trap "kill $alive 2>/dev/null" EXIT
for i in 1 2; do
timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/r2z_bench_$i.json 2> gpurun_out/r2z_bench_$i.err || exit 1
python -c "
import json; d=json.load(open('gpurun_out/r2z_bench_$i.json')); print(d['ms_per_step'], d['roofline']['frac']); p=d['pcie_inclusive']; print('pcie', p['ms_per_step'], p['phases_ms']['k_part_scatter1'], p['host_buffer_pages_per_numa_node'], p['gpu_numa_nodes_sysfs']); print('c3', d['c3']['wall_ms'])"
done
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2z_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r2z_tests.log
exit $rc
