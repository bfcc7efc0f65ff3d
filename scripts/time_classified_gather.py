"""The classify by the keys' owners against the replicated classify, on the one GPU: `world` loopback ranks (threads, each with its
own context and partition; the test build's transport) count their share of C3-shaped reads, filter, and then build the graph
twice — from the plain gather (every rank classifies the WHOLE k-mer set) and from the classified gather (every rank classifies
its own partition, asks the owners about the neighbours it does not hold, and the masks travel with the keys).  All ranks share the
one GPU, so what shows is the total work: the replicated classify is done `world` times over, the owners' classify once.
usage: python scripts/time_classified_gather.py [world=2] [reads per rank=5000000] [k=31]"""
import json, os, random, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GK_LIB_PATH", os.path.join(sys.path[0], "genome_amd", "libgenome_amd_test.so"))
from genome_amd import synth
from genome_amd.dist import DistDNAMap, HipDist
from genome_amd.dnamap import Context
from genome_amd.graph import buildGraph

world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
k = int(sys.argv[3]) if len(sys.argv) > 3 else 31
L_, G, err = 150, 4_600_000, 0.005
id128 = bytes(random.Random(world).getrandbits(8) for _ in range(128))
out, errors = [None] * world, []


def run(rank):
    try:
        c = Context(0)
        hd = HipDist(c, rank, world, id128, loopback=True)
        pm = DistDNAMap(hd, k)
        stride = synth.record_stride(L_)
        d = c.alloc(n * stride + 64)
        c.synth_reads(d, n, L_, "G", 3, rank * n, G, err)
        pm.count_reads_dev(d, n, L_)
        pm.deleteAll_lt(3)
        c.free(d)
        res = {"rank": rank, "local_keys": pm.local.size(), "keys": pm.size()}
        for name, classified in (("plain", False), ("classified", True), ("plain_again", False), ("classified_again", True)):
            hd.barrier(); c.sync()
            t0 = time.perf_counter()
            full = pm.gathered(classified=classified)
            c.sync(); hd.barrier()
            t1 = time.perf_counter()
            g = buildGraph(k, full)
            c.sync(); hd.barrier()
            t2 = time.perf_counter()
            st = g.buildStats()
            res[name] = {"gather_ms": round((t1 - t0) * 1e3, 2), "build_ms": round((t2 - t1) * 1e3, 2), "classify_phase_ms": round(st["phase_ms"]["classify"], 2),
                         "by_owners": st["classified_by_owners"], "graph": [g.counts(), g.checksum()]}
            g.close(); full.close()
        res["queries_asked"] = pm.classify_queries()
        out[rank] = res
        hd.barrier()
        pm.close(); hd.close(); c.close()
    except BaseException as e:      # noqa: BLE001
        errors.append((rank, repr(e)))


threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
for t in threads:
    t.start()
for t in threads:
    t.join(timeout=500)
if errors or any(t.is_alive() for t in threads):
    print("FAILED", errors, file=sys.stderr)
    sys.exit(1)
assert all(o["plain"]["graph"] == o["classified"]["graph"] == out[0]["plain"]["graph"] for o in out)
for o in out:
    for name in ("plain", "classified", "plain_again", "classified_again"):
        o[name].pop("graph")
print(json.dumps({"world": world, "reads_per_rank": n, "k": k, "graph": out[0].get("graph"), "ranks": out}, indent=1))
