set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
( while true; do sleep 45; date >> $R/gpurun_out/r77.alive; done ) & alive=$!
trap "kill $alive 2>/dev/null" EXIT
cd /tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/r77_pmc$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/r77_pmc$i.log 2>&1 || { tail -3 $R/gpurun_out/r77_pmc$i.log; exit 1; }
done
cd $R
python3 - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/r77_pmc*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name'].replace('void ','').split('(')[0]
        agg[n][r['Counter_Name']].append(float(r['Counter_Value']))
for n in agg:
    if n.startswith(('k_op_scatter1_reads','k_part_scatter2','k_seg_insert')):
        print(n[:44], {c: round(sum(v)/len(v)) for c,v in sorted(agg[n].items())})
PY
