#!/bin/bash
# usage: build genome_amd/variants/<name>.so with -DGK_TIMERS (see sweep_variants.sh), then: sweep_timers.sh <name>...
R=$GRAFT_REPO_ROOT; cp $R/genome_amd/libgenome_amd.so /tmp/orig.so
for v in "$@"; do
  cp $R/genome_amd/variants/$v.so $R/genome_amd/libgenome_amd.so
  echo "== $v"; python3 $R/scripts/run_timers.py || exit 1
done
cp /tmp/orig.so $R/genome_amd/libgenome_amd.so
