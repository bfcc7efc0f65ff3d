#!/bin/bash
# rocprofv3 kernel trace + stats of the whole C3 flow (count -> filter -> buildGraph -> bubbles -> simplify -> retain)
# usage (on the GPU box): bash scripts/prof_c3.sh <tag> [run_c3 args...]
set -o pipefail
tag=${1:-c3}; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 scripts/run_c3.py "$@" > gpurun_out/prof_${tag}.json 2> gpurun_out/prof_${tag}.err
echo "rc=$?"; cat gpurun_out/prof_${tag}.json
f=$(find $out -name '*kernel_stats.csv' | head -1); echo $f; head -40 $f
