"""The host-fed count (gk_map_count_reads from a pinned `.bin` buffer) a few times: run under
rocprofv3 --kernel-trace --memory-copy-trace to see how the sub-chunk uploads and the L1 scatter interleave."""
import sys, time, ctypes as C
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
from genome_amd import synth
from genome_amd.dnamap import Context, HipDNAMap
ctx = Context(0)
n, L, k = 1_000_000, 150, 31
stride = synth.record_stride(L)
d = ctx.alloc(n * stride + 64)
ctx.synth_reads(d, n, L, "U", 2, 0, 0, 0.0)
import os
late = "late_alloc" in sys.argv


def numa_report(tag):
    try:
        node = [x for x in os.listdir("/sys/devices/system/node") if x.startswith("node")]
        cpu = os.sched_getcpu()
        mine = [nd for nd in node if os.path.exists(f"/sys/devices/system/node/{nd}/cpu{cpu}")]
        gpu = [open(f).read().strip() for f in __import__("glob").glob("/sys/class/drm/card*/device/numa_node")]
        print(f"[{tag}] cpu {cpu} on {mine}, {len(node)} NUMA nodes, affinity {len(os.sched_getaffinity(0))} cpus, GPU numa_node {gpu}")
    except Exception as e:
        print("numa_report:", e)


def make_hb():
    numa_report("host_alloc")
    hb_ = ctx.host_alloc(n * stride)
    hb_[:] = ctx.download(d, n * stride)
    return hb_


if not late:
    hb = make_hb()
if len(sys.argv) > 1 and sys.argv[1] == "after_resident":       # what bench.py does first: the device-resident headline on another map
    m0 = HipDNAMap(ctx, k, int(n * (L - k + 1) * 1.05))
    for it in range(12):
        m0.clear(); m0.count_reads_dev(d, n, L)
    ctx.sync()
    if len(sys.argv) > 2 and sys.argv[2] == "close":
        m0.close()
if late:
    hb = make_hb()
m = HipDNAMap(ctx, k, int(n * (L - k + 1) * 1.05))
for it in range(6):
    m.clear(); ctx.sync(); t0 = time.perf_counter(); m.count_reads(hb, n); ctx.sync()
    print(f"step {it}: {(time.perf_counter() - t0) * 1e3:.3f} ms, phases {[round(x, 3) for x in m.last_phase_ms()]}")
