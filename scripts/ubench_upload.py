"""H2D upload rate from pinned and pageable host memory through the library (gk_dev_upload = hipMemcpyAsync + sync)."""
import sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
from genome_amd.dnamap import Context
ctx = Context(0)
for mb in (1, 5, 10, 39, 156):
    n = mb << 20
    d = ctx.alloc(n)
    pinned = ctx.host_alloc(n); pinned[:] = 7
    pageable = np.full(n, 7, np.uint8)
    for name, buf in (("pinned", pinned), ("pageable", pageable)):
        ctx.upload(d, buf)
        t0 = time.perf_counter()
        for _ in range(10):
            ctx.upload(d, buf)
        dt = (time.perf_counter() - t0) / 10
        print(f"{mb:4d} MB {name:9s} {dt*1e3:7.3f} ms  {n/dt/1e9:6.1f} GB/s", flush=True)
    ctx.host_free(pinned); ctx.free(d)
