set -o pipefail
mkdir -p gpurun_out
( while true; do sleep 45; date >> gpurun_out/r65.alive; done ) & alive=$!
trap "kill $alive 2>/dev/null" EXIT
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r65_prof -- python3 $R/scripts/run_c3.py 20000000 150000000 0.005 0 0 0 55 auto > $R/gpurun_out/r65_c4like.json 2> $R/gpurun_out/r65_c4like.err
cd $R
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r65_prof/*/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:12]: print(r['Name'][:70].ljust(72), r['Calls'], round(float(r['TotalDurationNs'])/1e6,1),'ms')
PY
python3 -c "
import json; d=json.load(open('gpurun_out/r65_c4like.json')); print(d['times'], d['table'])"
