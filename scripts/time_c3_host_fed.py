"""C3's count from pinned host memory under the streaming switches, beside the device-resident count (same map, same reads):
where the host-fed call's extra time goes.  usage: python scripts/time_c3_host_fed.py [reads=50000000]
Variants: default (704 MiB chunks, each uploaded beside the previous chunk's fine level),
host_prefetch=0 (every chunk piece-wise under its own L1 scatter), max_stage = 704 / 352 / 1408 MiB (an explicit max_stage
also switches the short first chunk off)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GK_LIB_PATH", os.path.join(sys.path[0], "genome_amd", "libgenome_amd_test.so"))
from genome_amd import synth
from genome_amd.dnamap import Context, HipDNAMap

N = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
L, k, G, err = 150, 31, 4_600_000, 0.005
ctx = Context(0)
stride = synth.record_stride(L)
nbytes = N * stride
d = ctx.alloc(nbytes + 64)
ctx.synth_reads(d, N, L, "G", 3, 0, G, err)
hbuf = ctx.host_alloc(nbytes)
step = 256 << 20
for o in range(0, nbytes, step):
    hbuf[o:o + step] = ctx.download(d + o, min(step, nbytes - o))
m = HipDNAMap(ctx, k, 0)
names = ["hist1", "P2", "P3", "P4", "P5"]
keys = ("partitioned_launches", "direct_launches", "grows", "slots")


def run(label, f, reps=2, filter_first=False):
    best = None
    for _ in range(reps):
        if filter_first:
            m.deleteAll_lt(3)          # what bench.py's legs do between counts: the table is rebuilt for the survivors, the next count re-creates it
        m.clear(); ctx.sync()
        t0 = time.perf_counter(); f(); ctx.sync(); ms = (time.perf_counter() - t0) * 1e3
        st = m.stats()
        row = {"what": label, "count_ms": round(ms, 2), "phases_ms": dict(zip(names, [round(x, 2) for x in m.last_phase_ms()])), "gap_ms": st.get("gap_ms"),
               **{k_: st[k_] for k_ in keys if k_ in st}}
        if best is None or ms < best["count_ms"]:
            best = row
    print(json.dumps(best), flush=True)
    return best


run("device-resident", lambda: m.count_reads_dev(d, N, L), reps=3)
variants = [("host default", {}), ("host_prefetch=0", {"host_prefetch": 0}), ("host default again", {}), ("max_stage=704MiB", {"test_max_stage": 704 << 20}),
            ("max_stage=352MiB", {"test_max_stage": 352 << 20}), ("max_stage=1408MiB", {"test_max_stage": 1408 << 20}), ("max_stage=2047MiB (one chunk)", {"test_max_stage": 2047 << 20}),
            ("max_stage=704MiB, host_prefetch=0", {"test_max_stage": 704 << 20, "host_prefetch": 0})]
for label, opts in variants:
    for name, v in opts.items():
        ctx.set_option(name, v)
    run(label, lambda: m.count_reads(hbuf, N))
    for name in opts:
        ctx.set_option(name, -1 if name == "host_prefetch" else 0)
run("device-resident after filter_lt", lambda: m.count_reads_dev(d, N, L), reps=3, filter_first=True)
run("host default after filter_lt", lambda: m.count_reads(hbuf, N), reps=3, filter_first=True)
print(json.dumps(ctx.mem_stats()))
