#!/bin/bash
# like sweep_variants.sh, but reads kernel durations from rocprofv3 (for variants whose results are
# wrong and so leave the pipeline through a fallback the bench's phase timers do not cover)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cp $R/genome_amd/libgenome_amd.so /tmp/orig.so
cd /tmp
for v in "$@"; do
  cp $R/genome_amd/variants/$v.so $R/genome_amd/libgenome_amd.so
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/vp_$v -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/vp_$v.log 2>&1 || exit 1
  echo "== $v"; head -8 $R/gpurun_out/vp_$v/*/*kernel_stats.csv | cut -c1-120
done
cp /tmp/orig.so $R/genome_amd/libgenome_amd.so
