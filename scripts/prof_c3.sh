#!/bin/bash
# rocprofv3 kernel trace + stats of the whole C3 flow (count -> filter -> buildGraph -> bubbles -> simplify -> retain)
# usage (on the GPU box): bash scripts/prof_c3.sh <tag> [run_c3 args...]
set -o pipefail
tag=${1:-c3}; shift
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
( while true; do sleep 45; date >> $R/gpurun_out/prof_${tag}.alive; done ) &      # the watchdog wants to see progress
alive=$!
cd /tmp
timeout -k 10 ${PROF_TIMEOUT:-300} rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $R/scripts/run_c3.py "$@" > $R/gpurun_out/prof_${tag}.json 2> $R/gpurun_out/prof_${tag}.err
echo "rc=$?"
kill $alive
cd $R
cat gpurun_out/prof_${tag}.json
f=$(find $out -name '*kernel_stats.csv' | head -1); echo $f; [ -n "$f" ] && head -40 $f
