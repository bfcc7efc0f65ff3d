set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_table_gpu.py tests/test_coverage_gpu.py -m gpu -x -q -k "host or stream or uniform or ragged or bin" 2>&1 | tail -3
for o in 1 2; do
python3 - $o <<'PY' || exit 1
import sys, time, os
sys.path.insert(0, os.getcwd())
from genome_amd import synth
from genome_amd.dnamap import Context, HipDNAMap
ctx = Context(0)
ctx.set_option("upload_streams", int(sys.argv[1]))
n, L, k = 1_000_000, 150, 31
stride = synth.record_stride(L)
hb = ctx.host_alloc(n * stride)
d = ctx.alloc(n * stride + 64)
ctx.synth_reads(d, n, L, "U", 2, 0, 0, 0.0)
hb[:] = ctx.download(d, n * stride)
m = HipDNAMap(ctx, k, int(n * (L - k + 1) * 1.05))
best = 1e9
for it in range(8):
    m.clear(); ctx.sync(); t0 = time.perf_counter(); m.count_reads(hb, n); ctx.sync(); best = min(best, time.perf_counter() - t0)
print("upload_streams", sys.argv[1], f"best {best * 1e3:.3f} ms, phases", [round(x, 3) for x in m.last_phase_ms()], m.size())
PY
done
