#!/bin/bash
# Run the default bench once per prebuilt tuning variant (genome_amd/variants/*.so, built with -D
# overrides of the gk_partition.hip tunables); the variant is loaded through GK_LIB_PATH, the product .so is untouched.
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  echo "== $v" >> $R/gpurun_out/sweep.log
  GK_LIB_PATH=$R/genome_amd/variants/$v.so python3 $R/bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 $BENCH_ARGS 2>/dev/null | grep -o '"ms_per_step": [0-9.]*\|"phases_ms": {[^}]*}' >> $R/gpurun_out/sweep.log || exit 1
done
cat $R/gpurun_out/sweep.log
