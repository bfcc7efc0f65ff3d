#!/bin/bash
# Build a variant of the library with extra compiler flags into genome_amd/variants/<name>.so (git-ignored; it travels to the
# GPU box with the tree).  Load it with GK_LIB_PATH (genome_amd/_lib.py); the product library is not touched.
# usage (here, no GPU needed): scripts/build_variant.sh timers -DGK_TIMERS
set -e
R=$(cd "$(dirname "$0")/.." && pwd); name=$1; shift
O=/tmp/gk_variant_$name; mkdir -p $O $R/genome_amd/variants
for f in $R/genome_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I$R/include ${SEGDEF:--DGK_SEG_BITS1=11} "$@" -c $f -o $O/$(basename $f .hip).o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/genome_amd/variants/$name.so $O/*.o -ldl
echo built $R/genome_amd/variants/$name.so
