"""Print one bench step's dispatches (name, duration, gap to the previous one) from a rocprofv3 --kernel-trace CSV.
usage: python scripts/trace_step.py <dir with *_kernel_trace.csv> [which step, default 6]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_op_scatter1' in r['Kernel_Name'] or 'k_count_reads' in r['Kernel_Name']]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 6
i0, i1 = idx[which], idx[which + 1]
prev = None
for r in rows[i0 - 4:i1 - 3]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{r['Kernel_Name'][:64]:64s} {(e - s) / 1000:8.1f} us   gap {((s - prev) / 1000 if prev else 0):7.1f} us")
    prev = e
print(f"step span {(int(rows[i1]['Start_Timestamp']) - int(rows[i0]['Start_Timestamp'])) / 1000:.1f} us")
