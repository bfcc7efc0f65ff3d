"""Merge a rocprofv3 kernel trace and memory-copy trace into one timeline; print the last `count_reads` step.
usage: python scripts/timeline.py <output dir>"""
import csv, glob, sys
d = sys.argv[1]
ev = []
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K ' + r['Kernel_Name'][:56]))
for f in glob.glob(d + '/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'C %s %s B' % (r.get('Direction', '?'), r.get('Bytes', r.get('Size', '?')))))
ev.sort()
last = max(i for i, e in enumerate(ev) if 'k_seg_insert' in e[2])
first = max(i for i, e in enumerate(ev[:last]) if 'k_seg_insert' in e[2]) + 1
t0 = ev[first][0]
for s, e, n in ev[first:last + 4]:
    print(f"{(s - t0) / 1000:9.1f} .. {(e - t0) / 1000:9.1f} us  ({(e - s) / 1000:7.1f})  {n}")
