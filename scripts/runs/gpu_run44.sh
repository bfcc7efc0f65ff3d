set -o pipefail
mkdir -p gpurun_out
( while true; do sleep 45; date >> gpurun_out/r3m.alive; done ) & alive=$!
trap "kill $alive 2>/dev/null" EXIT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3m_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r3m_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python scripts/run_c3.py 50000000 4600000 0.005 0 0 0 31 auto > gpurun_out/r3m_c3.json 2> gpurun_out/r3m_c3.err || { tail -3 gpurun_out/r3m_c3.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r3m_c3.json')); print(d['times'], d['count_phases_ms'], d['good_kmers'], d['graph_built'], d['table'])"
# growth by rehash, timed: a table that must grow three times while 2e7 distinct keys go in through the direct path
python3 - <<'PY'
import sys, os, time
sys.path.insert(0, os.getcwd())
from genome_amd import synth
from genome_amd.dnamap import Context, HipDNAMap
for classic in (1, 0):
    ctx = Context(0); ctx.set_option("filter_classic", classic)
    n, L, k = 200000, 150, 31
    d = ctx.alloc(n * synth.record_stride(L) + 64)
    m = HipDNAMap(ctx, k, 1 << 20); m.set_insert_path("direct")
    t0 = time.perf_counter()
    for b in range(8):
        ctx.synth_reads(d, n, L, "U", 2, b * n, 0, 0.0); m.count_reads_dev(d, n, L)
    ctx.sync(); dt = time.perf_counter() - t0
    print("classic" if classic else "streaming", f"{dt * 1e3:.1f} ms, size {m.size()}, grows {m.stats()['grows']}, verify {m.verify()}")
    m.close(); ctx.free(d); ctx.close()
PY
