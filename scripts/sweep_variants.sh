#!/bin/bash
# Run the default bench once per prebuilt tuning variant (genome_amd/variants/*.so, built with -D
# overrides of the gk_partition.hip tunables) — on the GPU box's copy of the tree only.
R=$GRAFT_REPO_ROOT; cp $R/genome_amd/libgenome_amd.so /tmp/orig.so
for v in "$@"; do
  cp $R/genome_amd/variants/$v.so $R/genome_amd/libgenome_amd.so
  echo "== $v" >> $R/gpurun_out/sweep.log
  python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 3 $BENCH_ARGS 2>/dev/null | grep -o '"ms_per_step": [0-9.]*\|"phases_ms": {[^}]*}' >> $R/gpurun_out/sweep.log || exit 1
done
cp /tmp/orig.so $R/genome_amd/libgenome_amd.so
cat $R/gpurun_out/sweep.log
