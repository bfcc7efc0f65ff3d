set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for v in base p2noatomic; do
  lib=""; [ $v != base ] && lib=$R/genome_amd/variants/$v.so
  cd /tmp && GK_LIB_PATH=$lib timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r76_$v -- python3 $R/scripts/time_p2_variant.py > $R/gpurun_out/r76_$v.log 2>&1
  cd $R
  python3 - $v <<'PY'
import csv,glob,sys
f=glob.glob(f'gpurun_out/r76_{sys.argv[1]}/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'k_op_scatter1_reads' in r['Name']: print(sys.argv[1], r['Name'][:50], r['Calls'], round(float(r['AverageNs'])/1e3,1),'us avg')
PY
done
