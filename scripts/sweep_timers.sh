#!/bin/bash
# usage: build genome_amd/variants/<name>.so with -DGK_TIMERS (scripts/build_variant.sh <name> -DGK_TIMERS), then: sweep_timers.sh <name>...
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  echo "== $v"; GK_LIB_PATH=$R/genome_amd/variants/$v.so python3 $R/scripts/run_timers.py || exit 1
done
