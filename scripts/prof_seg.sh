#!/bin/bash
# usage: scripts/prof_seg.sh <tag>   (env passes through) -> per-kernel average times of the partitioned bench
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; tag=$1
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --insert-path ${GK_PATH:-partitioned} > $R/gpurun_out/prof_$tag.log 2>&1
python3 -c "
import csv,glob
for r in csv.DictReader(open(glob.glob('$R/gpurun_out/prof_$tag/*/*kernel_stats.csv')[0])):
    if 'k_' in r['Name']: print('$tag', r['Name'][:40].ljust(41), r['Calls'], round(float(r['AverageNs'])/1e3,1),'us')
" | head -6
