set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp; R=$PWD
( while true; do sleep 45; date >> $R/gpurun_out/r2o.alive; done ) & alive=$!
trap "kill $alive 2>/dev/null" EXIT
cd /tmp
for v in base p2exp1 p2exp2 p4exp1 p4exp2; do
  lib=$R/genome_amd/variants/$v.so; [ $v = base ] && lib=$R/genome_amd/libgenome_amd.so
  GK_LIB_PATH=$lib timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2o_$v -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $R/gpurun_out/r2o_$v.log 2>&1 || { echo "$v failed"; tail -3 $R/gpurun_out/r2o_$v.log; continue; }
  f=$(ls $R/gpurun_out/r2o_$v/*/*kernel_stats.csv | head -1)
  echo "== $v"; grep -E "k_op_scatter1|k_part_scatter2|k_seg_insert|k_count_reads" $f | cut -d, -f1-4 | cut -c1-150
done
