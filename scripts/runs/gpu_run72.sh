set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r72_prof -- python3 $R/bench.py --sharded --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/r72_sharded.json 2> $R/gpurun_out/r72_sharded.err
cd $R
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r72_prof/*/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:10]: print(r['Name'][:80].ljust(82), r['Calls'], round(float(r['AverageNs'])/1e3,1),'us avg')
PY
