#!/usr/bin/env python3
"""bench.py — distinct k-mers/s inserted (k=31, 150 bp synthetic reads) on N MI355X.

One "step" = one pass of the hot path over one batch of synthetic reads that is already resident
in HBM: clear the table, then FreqFilter.add for every read (extract -> canonicalise -> insert +
count).  Workload = BASELINE.json configs[1] ("C2"): 1M x 150 bp reads per GPU, k=31.
  N = 1 : single-partition DNAMap kernel (gk_map_count_reads_dev).
  N > 1 : weak scaling, one rank per GPU: every rank owns 1M reads and one table partition; k-mers are
          routed by strand-symmetric minimizer owner as SUPER-K-MER records (runs of same-owner windows,
          2 bits/base in 16-B slots), exchanged with ONE all-to-all over RCCL/xGMI and counted by their
          owner with the same pipeline as reads — all of it inside the C-ABI (gk_dist_count_reads_dev,
          csrc/gk_dist.hip).  torch.distributed (gloo) only carries the 128-byte RCCL id from rank 0 to
          the others.  value = distinct k-mers over all partitions / max-over-ranks time.
          Launched by `torch.distributed.run` (RANK / LOCAL_RANK / WORLD_SIZE in the environment) or by itself: with
          --gpus N and no WORLD_SIZE it starts its N ranks as child processes before touching torch or HIP.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

if int(os.environ.get("WORLD_SIZE", "1")) > 1 and not os.environ.get("GK_BENCH_LAUNCH_ONLY"):
    # only to bootstrap the RCCL id across ranks — and FIRST: the HIP runtime (and RCCL) torch bundles must be the
    # process's only ones (the library dlopens whichever RCCL is already loaded)
    import torch  # noqa: E402
    import torch.distributed as dist  # noqa: E402
import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes_count_kernel(occ, distinct, L, k):
    """per occurrence 2L/(8 n_k) B of packed read + one key slot read (W B) + one
    int32 count read-modify-write (8 B); plus W B once per distinct key (first write)."""
    W = 8 if k <= 32 else 16
    nk = L - k + 1
    return occ * (2.0 * L / (8.0 * nk) + W + 8.0) + distinct * W


def algorithmic_bytes_insert_kernel(keys, distinct, k):
    """Owner-side count of routed super-k-mer records: ~1.6 B of record per window (16-B slot per
    ~10 windows) + slot read (W B) + count RMW (8 B); + W B per distinct key."""
    W = 8 if k <= 32 else 16
    return keys * (1.6 + W + 8.0) + distinct * W


def cpu_baseline(rec_host: np.ndarray, nreads_total: int, k: int, target_s: float = 10.0):
    """The oracle (C restatement of the reference's ArrayDNAMap / PartitionedDNAMap + FreqFilter.add,
    incl. `improve`, tombstones, 0.3/0.7 rescale) timed on this host on a bounded prefix of the same
    reads: first one thread / one partition (= one ArrayDNAMap), then P = usable cores threads and
    partitions (= PartitionedDNAMap without the network).  The multi-core figure is the reported value."""
    from oracle import oracle as O
    cores = max(1, min(len(os.sched_getaffinity(0)), 16))     # a 1-GPU box has a 16-core share
    probe = min(2000, nreads_total)
    pm = O.PMap(k, 1)
    t0 = time.perf_counter()
    pm.count_reads(rec_host[:probe].tobytes(), probe)
    dt = max(time.perf_counter() - t0, 1e-6)
    pm.close()
    sample1 = int(min(nreads_total, max(probe, probe * target_s / dt)))
    pm = O.PMap(k, 1)
    t0 = time.perf_counter()
    occ1 = pm.count_reads(rec_host[:sample1].tobytes(), sample1)
    dt1 = time.perf_counter() - t0
    distinct1 = pm.size()
    pm.close()
    sample = int(min(nreads_total, sample1 * (3 if cores > 2 else 1)))     # bounded: ~10-30 s of CPU work in all
    pm = O.PMap(k, cores)
    t0 = time.perf_counter()
    occ = pm.count_reads_mt(rec_host[:sample].tobytes(), sample, cores)
    dt = time.perf_counter() - t0
    distinct = pm.size()
    pm.close()
    return {"value": distinct / dt, "unit": "distinct k-mers/s", "cores": cores, "kind": "port",
            "occurrences_per_s": occ / dt,
            "sample": f"first {sample} of the reads ({occ} k-mer occurrences, {distinct} distinct) in {dt:.1f} s on {cores} threads / "
                      f"{cores} partitions: C restatement of PartitionedDNAMap (hashCode mod P, no network) + FreqFilter.add",
            "single_thread": {"value": distinct1 / dt1, "occurrences_per_s": occ1 / dt1, "cores": 1,
                              "sample": f"first {sample1} reads, {dt1:.1f} s, one ArrayDNAMap"}}


def c2_variant(ctx, rec_ptr, n, L, k, mode, steps, warmup, host_buf=None, prefetch=False):
    """C2 again, outside the headline: mode G (5 Mbp genome, 30x, 1 % error — repeats) device-resident, or the §8(d) reading
    of the metric: packed reads in PINNED HOST memory -> complete table in HBM (gk_map_count_reads: the upload runs in
    sub-chunks on a copy stream and overlaps the L1 scatter of the pipeline)."""
    from genome_amd.dnamap import HipDNAMap
    nk = L - k + 1
    m = HipDNAMap(ctx, k, int(n * nk * 1.05))
    kms, phases = [], []

    def step(more=False):
        m.clear()
        if host_buf is None:
            m.count_reads_dev(rec_ptr, n, L)
        else:
            if prefetch and more:                   # a streaming caller: the NEXT batch's upload is started before this batch is counted
                m.prefetch_reads(host_buf, n)
            m.count_reads(host_buf, n)
        kms.append(m.last_count_kernel()[0]); phases.append(m.last_phase_ms())

    for _ in range(warmup):
        step()
    kms.clear(); phases.clear()
    ctx.sync()
    if prefetch and host_buf is not None:
        m.prefetch_reads(host_buf, n)
    t0 = time.perf_counter()
    for i in range(steps):
        step(i + 1 < steps)
    ctx.sync()
    dt = (time.perf_counter() - t0) / steps
    distinct = m.size()
    st = m.stats()
    out = {"ms_per_step": dt * 1e3, "value": distinct / dt, "unit": "distinct k-mers/s", "occurrences_per_s": n * nk / dt,
           "distinct_per_step": distinct, "steps": steps, "kernel_ms": float(np.mean(kms)),
           "phases_ms": dict(zip(["k_part_hist1", "k_part_scatter1", "k_part_hist2", "k_part_scatter2", "k_seg_insert"],
                                 [float(x) for x in np.mean(np.array(phases), axis=0)])),
           "insert_path": "partitioned" if st["partitioned_launches"] else "direct", "table_slots": st["slots"]}
    abytes = algorithmic_bytes_count_kernel(n * nk, distinct, L, k)
    out["roofline_frac_of_step"] = abytes / dt / 1e9 / HBM_PEAK_GBS
    m.close()
    return out


def numa_of(addr):
    """NUMA node(s) holding the pages of the mapping that contains `addr` (/proc/self/numa_maps), e.g. {"N0": 9589}; {} if unknown."""
    try:
        best = None
        for line in open("/proc/self/numa_maps"):
            f = line.split()
            start = int(f[0], 16)
            if start <= addr and (best is None or start > best[0]):
                best = (start, {x.split("=")[0]: int(x.split("=")[1]) for x in f[1:] if x.startswith("N") and "=" in x})
        return best[1] if best else {}
    except Exception:
        return {}


def gpu_numa_node():
    try:
        import glob
        return sorted({open(f).read().strip() for f in glob.glob("/sys/class/drm/card*/device/numa_node")})
    except Exception:
        return []


def c3_object(ctx):
    """BASELINE.json configs[2] measured in THIS run: 50 M x 150 bp reads over a 4.6 Mbp genome (~1600x, 0.5 % error), k = 31, one
    GPU, all reads resident in HBM (1.95 GB): FreqFilter.extractFilteredKmers(.., 3) -> Graph.buildGraph -> removeBubbles ->
    simplifyGraph -> components + retain (GraphBuilder.scala:32-54).  One warm-up pass, one timed pass; wall ms per phase
    (every call returns synchronised), device ms where the library keeps them, and each phase's algorithmic bytes against HBM peak."""
    from genome_amd import synth
    from genome_amd.dnamap import HipDNAMap
    from genome_amd.graph import buildGraph
    N, L, k, G, err = 50_000_000, 150, 31, 4_600_000, 0.005
    nk = L - k + 1
    d = ctx.alloc(N * synth.record_stride(L) + 64)
    ctx.synth_reads(d, N, L, "G", 3, 0, G, err)
    ctx.sync()
    m = HipDNAMap(ctx, k, 0)
    res = None
    for timed in (False, True):
        t = {}
        m.clear()
        t0 = time.perf_counter(); occ = m.count_reads_dev(d, N, L); t["count"] = time.perf_counter() - t0
        count_kernel_ms = m.last_count_kernel()[0]; phases = m.last_phase_ms(); st = m.stats(); distinct = m.size()
        t0 = time.perf_counter(); m.deleteAll_lt(3); t["filter_lt"] = time.perf_counter() - t0
        good = m.size()
        t0 = time.perf_counter(); g = buildGraph(k, m); t["buildGraph"] = time.perf_counter() - t0
        built, bs = g.counts(), g.buildStats()
        t0 = time.perf_counter(); g.removeBubbles(); t["removeBubbles"] = time.perf_counter() - t0
        t0 = time.perf_counter(); g.simplifyGraph(); t["simplifyGraph"] = time.perf_counter() - t0
        t0 = time.perf_counter(); kept, comps = g.retainLargest(); t["retainLargest"] = time.perf_counter() - t0
        final = g.counts()
        g.close()
        if not timed:
            continue
        ph = bs["phase_ms"]
        walk_ms = ph["unitig_measure"] + ph["unitig_emit"]

        def roof(nbytes, ms):
            gbs = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            return {"algorithmic_bytes": nbytes, "ms": ms, "achieved_GB_s": gbs, "frac": gbs / HBM_PEAK_GBS}
        res = {"workload": f"C3: {N} x {L}bp reads (SplitMix64 mode G, {G} bp genome, e={err}), k={k}, 1 GPU, reads resident in HBM, no capacity hint",
               "occurrences": occ, "distinct": distinct, "good_kmers_ge3": good,
               "wall_ms": {k_: v * 1e3 for k_, v in t.items()}, "total_wall_ms": sum(t.values()) * 1e3,
               "count": {"kernel_ms": count_kernel_ms,
                         "phases_ms": dict(zip(["k_part_hist1", "k_part_scatter1", "k_part_hist2+scans", "k_part_scatter2", "k_seg_insert"], phases)),
                         "occurrences_per_s": occ / t["count"], "distinct_per_s": distinct / t["count"],
                         "partitioned_launches": st["partitioned_launches"], "direct_launches": st["direct_launches"],
                         "table_slots": st["slots"], "grows": st["grows"], "spilled_keys": st["spilled_keys"]},
               "graph": {"nodes_edges_bases_built": built, "build_phase_ms": ph, "pointer_jumping": bs["pointer_jumping"],
                         "components": comps, "largest_component_nodes": kept, "nodes_edges_bases_final": final},
               "roofline": {"count": roof(occ * (2.0 * L / (8.0 * nk) + 16.0) + distinct * 8.0, t["count"] * 1e3),
                            "filter_lt": roof((st["slot_bytes"] + 4.0) * st["slots"], t["filter_lt"] * 1e3),
                            "classify": roof(80.0 * good, ph["classify"]),
                            "unitig_walk": roof(9.25 * bs["walked_bases"], walk_ms),
                            "bytes_per_unit": "SURVEY.md §8d: 16.3 B/occurrence (+8 B/distinct key) count; 20 B/slot filter (12 B scan + 8 B tombstone; the wall "
                                              "time also holds the rebuild into a table sized for the survivors); 80 B/live key classify; 9.25 B/walked base"}}
    # what the memory system carried for the graph kernels, from the committed PMC passes over this same flow (every read request is
    # a 128-byte line, also for a 16-byte random probe: the algorithmic fractions above understate how busy HBM is)
    import glob as _glob
    cands = sorted(_glob.glob(os.path.join(ROOT, "profiles", "r03", "pmc_c3_*.json")))
    pmc = cands[-1] if cands else os.path.join(ROOT, "profiles", "r02", "pmc_c3_v11.json")
    if res is not None and os.path.exists(pmc):
        k_ = json.load(open(pmc))["kernels"]
        res["roofline"]["moved_by_pmc"] = {name: {"fetch_bytes": k_[name]["fetch_bytes"], "write_bytes": k_[name]["write_bytes"], "ms": k_[name]["total_ms"],
                                                   "frac_of_hbm_peak": k_[name]["moved_frac_of_hbm_peak"]}
                                            for name in ("k_classify<1>", "k_walk_q<1>", "k_compact_seg<1>", "k_filter_lt<1>", "k_rehash<1>", "k_cc_link") if name in k_}
        res["roofline"]["moved_by_pmc"]["source"] = os.path.relpath(pmc, ROOT)
    # the same count from PINNED HOST memory (1.95 GB of `.bin` -> filtered table): the stream is cut where the device-resident count cuts
    # its batches, the first chunk's upload overlaps its own L1 scatter piece by piece, every later chunk's upload runs beside the
    # previous chunk's fine level (two staging areas)
    if res is not None:
        try:
            nbytes = N * synth.record_stride(L)
            hbuf = ctx.host_alloc(nbytes)
            step_b = 256 << 20
            for o in range(0, nbytes, step_b):
                hbuf[o:o + step_b] = ctx.download(d + o, min(step_b, nbytes - o))
            for timed in (False, True):
                m.clear()
                t0 = time.perf_counter(); occ_h = m.count_reads(hbuf, N); t_count = time.perf_counter() - t0
                ph = m.last_phase_ms()
                t0 = time.perf_counter(); m.deleteAll_lt(3); t_filter = time.perf_counter() - t0
            res["host_fed"] = {"count_ms": t_count * 1e3, "filter_lt_ms": t_filter * 1e3, "occurrences": occ_h, "good_kmers_ge3": m.size(),
                               "phases_ms": dict(zip(["k_part_hist1", "k_part_scatter1", "k_part_hist2+scans", "k_part_scatter2", "k_seg_insert"], ph)),
                               "over_device_resident_count": t_count * 1e3 / res["wall_ms"]["count"], "host_bytes": nbytes,
                               "what": "gk_map_count_reads over the same reads in pinned host memory, then filter_lt(3)"}
            ctx.host_free(hbuf)
        except Exception as e:          # noqa: BLE001
            res["host_fed"] = {"error": repr(e)}
    m.close(); ctx.free(d)
    return res


def launch_ranks(n, argv, launch_only):
    """`bench.py --gpus N` without an outer launcher (WORLD_SIZE unset): start N fresh copies of this script, one rank per GPU,
    BEFORE this process has imported torch or touched HIP (the reference driver deploys its own partitions too,
    S/ds/PartitionedDNAMap.scala:20-28).  Rank 0's stdout — the one JSON line — is forwarded, everything else goes to stderr;
    the exit code is the worst child's.  launch_only (--launch-only): the children only report their environment and exit —
    the plumbing, testable without a GPU."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        if launch_only:
            env["GK_BENCH_LAUNCH_ONLY"] = "1"
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if (r == 0 or launch_only) else sys.stderr, stderr=sys.stderr))
    outs, worst = [], 0
    for r, p in enumerate(procs):
        o = p.communicate()[0] if (r == 0 or launch_only) else (p.wait(), None)[1]
        outs.append(o.decode() if o else "")
        if p.returncode != 0:
            print(f"bench.py: rank {r} exited with {p.returncode}", file=sys.stderr)
            worst = max(worst, abs(p.returncode) or 1)
    if launch_only:
        ranks = [json.loads(o.strip().splitlines()[-1]) for o in outs if o.strip()]
        print(json.dumps({"launch_only": True, "n_gpus": n, "rc": worst, "ranks": ranks}), flush=True)
    else:
        lines = [ln for ln in outs[0].splitlines() if ln.startswith("{")]
        if lines:
            print(lines[-1], flush=True)
        elif not worst:
            worst = 1
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=1_000_000, help="reads per GPU")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--mode", choices=["U", "G"], default="U", help="U: uniform reads (every k-mer distinct w.h.p.); "
                    "G: 5 Mbp genome, 30x, 1%% error")
    ap.add_argument("--insert-path", choices=["auto", "direct", "partitioned"], default="auto",
                    help="direct: one global CAS/add per k-mer; partitioned: radix-partition by table segment, build in LDS")
    ap.add_argument("--sharded", action="store_true", help="run the N>1 code path (owner bucketing + all-to-all + owner insert) "
                    "even with one rank: the only way to exercise it on a 1-GPU box")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--opt", action="append", default=[], help="gk_ctx_set_option name=value (A/B switches of the insert pipeline)")
    ap.add_argument("--no-extras", action="store_true", help="skip the objects measured next to the headline at N=1: mode_G, pcie_inclusive, c3")
    ap.add_argument("--no-c3", action="store_true", help="skip only the c3 object (A/B runs of the C2 pipeline)")
    ap.add_argument("--launch-only", action="store_true", help="with --gpus N and no WORLD_SIZE: start the N ranks, let each report its "
                    "environment and exit (checks the launcher without a GPU)")
    args = ap.parse_args()
    if os.environ.get("GK_BENCH_LAUNCH_ONLY"):      # a child of --launch-only: no torch, no HIP
        r = int(os.environ["RANK"])
        print(json.dumps({"rank": r, "local_rank": int(os.environ["LOCAL_RANK"]), "world": int(os.environ["WORLD_SIZE"]),
                          "master": os.environ["MASTER_ADDR"] + ":" + os.environ["MASTER_PORT"], "gpus_arg": args.gpus}), flush=True)
        sys.exit(int(os.environ.get("GK_BENCH_TEST_EXIT", "0")) if r == int(os.environ.get("GK_BENCH_TEST_EXIT_RANK", "-1")) else 0)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, [a for a in sys.argv[1:] if a != "--launch-only"], args.launch_only))
    # stdout carries exactly ONE JSON line: native libraries (RCCL prints a version banner) write to
    # fd 1 behind Python's back, so fd 1 is pointed at stderr until the result is ready.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: running with {world} ranks", file=sys.stderr)
    sharded = world > 1 or args.sharded
    if args.opt:        # A/B switches exist in the TEST build of the library only (the product library has no option switch)
        os.environ.setdefault("GK_LIB_PATH", os.path.join(ROOT, "genome_amd", "libgenome_amd_test.so"))

    from genome_amd import synth
    from genome_amd.dnamap import Context, HipDNAMap

    n, L, k = args.reads, args.read_len, args.k
    W = 1 if k <= 32 else 2
    nk = L - k + 1
    stride = synth.record_stride(L)
    ctx = Context(local_rank)
    for kv in args.opt:
        name, val = kv.split("=")
        ctx.set_option(name, int(val))
    # The pinned host buffer of the pcie_inclusive object is taken NOW, the way an application takes its I/O buffers at start-up:
    # pinned late, after the process has churned through host memory, the same buffer uploaded at 26 instead of 56 GB/s in three
    # runs out of four (pages on the right NUMA node both times: presumably small pages behind the IOMMU instead of large ones).
    want_extras = (world == 1 and not (world > 1 or args.sharded) and not args.no_extras
                   and args.mode == "U" and n == 1_000_000 and L == 150 and k == 31)
    hb = ctx.host_alloc(n * stride) if want_extras else None
    rec = ctx.alloc(n * stride + 64)
    G, err = 5_000_000 * world, 0.01
    ctx.synth_reads(rec, n, L, args.mode, 2, rank * n, G, err)     # config_id 2 = C2
    occ_rank = n * nk
    hint = int(occ_rank * 1.05 * float(os.environ.get('GK_HINT_SCALE', '1')))
    hd = None
    if sharded:
        from genome_amd.dist import DistDNAMap, HipDist, unique_id
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            from genome_amd.dist import share_id
            id128 = share_id(rank, world)               # gloo: only the 128-byte RCCL id travels over torch.distributed
        else:
            id128 = unique_id()
        hd = HipDist(ctx, rank, world, id128)
        pm = DistDNAMap(hd, k, hint)
        m = pm.local
    else:
        m = HipDNAMap(ctx, k, hint)
    m.set_insert_path(args.insert_path)

    kernel_ms, kernel_units, phase_ms, dist_ms = [], [], [], []
    begun = [0]

    def step(i=0, K=1):
        m.clear()
        if not sharded:
            m.count_reads_dev(rec, n, L)
            ms, kocc = m.last_count_kernel()
            kernel_ms.append(ms); kernel_units.append(kocc); phase_ms.append(m.last_phase_ms())
            return
        # a streaming loop, three batches deep: the routing of batch i+2 is launched on the second stream and the exchange of
        # batch i+1 posted on the communication stream before batch i is counted, so the route kernel and the wire time hide
        # behind the owner pipeline.  Every loop of K steps launches K routes and consumes K (its first two are not hidden).
        if i == 0:
            begun[0] = 0
        while begun[0] < min(i + 3, K):
            pm.route_begin(rec, n, L)
            begun[0] += 1
        _, owned = pm.count_routed()
        ms, _ = m.last_count_kernel()
        kernel_ms.append(ms); kernel_units.append(owned); phase_ms.append(m.last_phase_ms()); dist_ms.append(hd.last_ms())

    def fence():
        ctx.sync()
        if sharded:
            hd.barrier()
            ctx.sync()

    for i in range(args.warmup):
        step(i, args.warmup)
    kernel_ms.clear(); kernel_units.clear(); phase_ms.clear(); dist_ms.clear()
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, args.steps)
    fence()
    dt = time.perf_counter() - t0

    distinct_rank = m.size()
    if sharded:
        dt_max = float(hd.allreduce([dt], "max")[0])
        tsum = hd.allreduce([float(distinct_rank), float(occ_rank)], "sum")
        distinct_total, occ_total = float(tsum[0]), float(tsum[1])
    else:
        dt_max, distinct_total, occ_total = dt, float(distinct_rank), float(occ_rank)

    if rank == 0:
        ms_per_step = dt_max / args.steps * 1e3
        units = float(np.mean(kernel_units))
        phases = np.mean(np.array(phase_ms), axis=0)
        stats = m.stats()
        partitioned = stats["partitioned_launches"] > 0
        slot_b = stats["slot_bytes"]            # 12 (8-byte keys in a count table), 16 (the graph layout) or 24 (16-byte keys)
        # ALGORITHMIC bytes of one insert+count pass, whatever kernels carry it
        abytes = (algorithmic_bytes_count_kernel(units, distinct_rank, L, k) if not sharded
                  else algorithmic_bytes_insert_kernel(units, distinct_rank, k))
        if partitioned:
            names = ["k_part_hist1", "k_part_scatter1", "k_part_hist2", "k_part_scatter2", "k_seg_insert"]
            kernel_time_ms = float(phases.sum())
            kname = "partitioned insert pipeline: " + " + ".join(f"{n}<{W}>" for n in names)
            timing = "HIP events on the library stream around each phase (gk_map_last_phase_ms), summed"
            dom = int(np.argmax(phases))
            # what the dominant kernel itself must move: its keys in, its table segments out (+ in unless built from empty)
            src_b = 1.6 if sharded else 2.0 * L / (8 * nk)           # P1/P2 read super-k-mer records or packed reads
            dom_bytes = {4: units * 8 * W + m.slots() * slot_b, 3: units * 16 * W, 2: units * 8 * W,
                         1: units * (8 * W + src_b), 0: units * src_b}[dom]
            dominant = {"kernel": f"{names[dom]}<{W}>", "ms": float(phases[dom]), "own_streaming_bytes": dom_bytes,
                        "GB_per_s": dom_bytes / (float(phases[dom]) * 1e-3) / 1e9,
                        "frac_of_peak": dom_bytes / (float(phases[dom]) * 1e-3) / 1e9 / HBM_PEAK_GBS}
            phase_detail = {n: float(x) for n, x in zip(names, phases)}
        else:
            kernel_time_ms = float(np.mean(kernel_ms))
            kname = ("k_count_reads" if not sharded else "k_add_keys") + f"<{W}>"
            timing = "HIP events on the library stream (gk_map_last_count_kernel)"
            dominant, phase_detail = None, None
        avg_kernel_ms = kernel_time_ms
        achieved = abytes / (avg_kernel_ms * 1e-3) / 1e9
        traffic = None
        # HBM traffic of one step by the PMC counters: the newest committed pass over this very command (scripts/profile_round.sh)
        pmc_file = os.path.join(ROOT, "profiles", "r01", "pmc_count_reads_v2.json")
        if partitioned:
            import glob
            cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r03", "pmc_pipeline_*.json"))) or \
                sorted(glob.glob(os.path.join(ROOT, "profiles", "r02", "pmc_pipeline_v1[34].json")))
            pmc_file = cands[-1] if cands else pmc_file
        if not sharded and args.mode == "U" and n == 1_000_000 and L == 150 and k == 31 and os.path.exists(pmc_file):
            pj = json.load(open(pmc_file))
            traffic = pj.get("hbm_bytes_per_launch", pj.get("k_count_reads<1>", {}).get("hbm_bytes_per_launch"))
        out = {
            "metric": "distinct k-mers/s inserted (k=31, 150bp reads)",
            "value": distinct_total / (dt_max / args.steps),
            "unit": "distinct k-mers/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64" if W == 1 else "u128", "data": "synthetic",
            "config": {"workload": f"C2: {n} x {L}bp synthetic reads per GPU (SplitMix64 mode {args.mode}), k={k}, "
                                   + ("single-partition DNAMap kernel" if not sharded else
                                      f"minimizer-sharded PartitionedDNAMap, {world} partitions, RCCL all-to-all"),
                       "reads_per_gpu": n, "read_len": L, "k": k, "mode": args.mode,
                       "table_slots_per_gpu": m.slots(), "slot_bytes": stats["slot_bytes"],
                       "insert_path": "partitioned" if partitioned else "direct"},
            "occurrences_per_s": occ_total / (dt_max / args.steps),
            "distinct_per_step": distinct_total, "occurrences_per_step": occ_total,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel": kname,
                         "kernel_ms": avg_kernel_ms, "units_per_launch": units,
                         "algorithmic_bytes_per_launch": abytes, "timing": timing,
                         "phases_ms": phase_detail, "dominant": dominant,
                         "traffic_source": (os.path.relpath(pmc_file, ROOT) if traffic else None)},
        }
        # the same figures against what THIS card streams, measured in this run (after the timed region)
        # (the objects measured NEXT to the headline must never cost the line itself: one that fails is reported in place)
        def extra(name, fn):
            try:
                return fn()
            except Exception as e:          # noqa: BLE001
                print(f"bench.py: {name} failed: {e!r}", file=sys.stderr)
                return {"error": repr(e)}

        rl = out["roofline"]
        sb = extra("measured_stream", lambda: ctx.stream_bench(1 << 30, 10))
        if "error" in sb:
            rl["measured_stream"] = sb
        else:
            rl["measured_stream"] = dict(sb, what="gk_dev_stream_bench: 16-byte copy / fill / sum kernels over 1 GiB buffers, GB/s "
                                                 "(copy counts bytes read + written)")
            rl["frac_of_measured_copy"] = achieved / sb["copy_GBps"] if sb["copy_GBps"] else None
            if traffic:
                rl["traffic_GBps"] = traffic / (avg_kernel_ms * 1e-3) / 1e9
                rl["traffic_over_measured_copy"] = rl["traffic_GBps"] / sb["copy_GBps"] if sb["copy_GBps"] else None
        if sharded and dist_ms:
            out["per_rank_step_ms"] = {k_: float(np.mean([x[k_] for x in dist_ms])) for k_ in dist_ms[0]}
            out["per_rank_step_ms"]["what"] = ("rank 0, wall ms inside gk_dist_count_routed: route_wait = waiting for the routing kernel (reads -> super-k-mer records "
                                               "grouped by owner) that gk_dist_route_begin launched on the second stream two steps earlier; "
                                               "exchange = counts + records of the NEXT batch over RCCL, posted on the communication stream before this batch is counted (enqueue + the one host sync for the sizes); owner_count = "
                                               "the pipeline over what arrived (stream-ordered behind the receives)")
        default_workload = args.mode == "U" and n == 1_000_000 and L == 150 and k == 31
        if world == 1 and not sharded and not args.no_extras and default_workload:
            def mode_g():
                recg = ctx.alloc(n * stride + 64)
                ctx.synth_reads(recg, n, L, "G", 2, 0, 5_000_000, 0.01)
                o = c2_variant(ctx, recg, n, L, k, "G", 5, 2)
                o["workload"] = "C2 mode G: 5 Mbp genome, 30x, 1 % error (SURVEY.md §8d) — the same reads count, with repeats"
                ctx.free(recg)
                return o

            def pcie():
                hb[:] = ctx.download(rec, n * stride)
                single = c2_variant(ctx, None, n, L, k, "U", 10, 2, host_buf=hb)
                o = c2_variant(ctx, None, n, L, k, "U", 10, 2, host_buf=hb, prefetch=True)
                o["one_call_at_a_time_ms_per_step"] = single["ms_per_step"]
                o["one_call_at_a_time_phases_ms"] = single["phases_ms"]
                o["host_buffer_pages_per_numa_node"] = numa_of(hb.ctypes.data)
                o["gpu_numa_nodes_sysfs"] = gpu_numa_node()
                o["workload"] = ("SURVEY.md §8(d) reading of the metric: the headline's reads as a `.bin` stream in PINNED HOST memory -> complete "
                                 "table in HBM, 10 steps back to back the way a streaming caller runs them: gk_map_prefetch_reads starts batch i+1's "
                                 "upload before gk_map_count_reads counts batch i, so the copy runs beside batch i's fine level (one_call_at_a_time = no "
                                 "prefetch: the upload in sub-chunks overlapped with the L1 scatter only)")
                ctx.host_free(hb)
                return o

            out["mode_G"] = extra("mode_G", mode_g)
            out["pcie_inclusive"] = extra("pcie_inclusive", pcie)
            if not args.no_c3:
                out["c3"] = extra("c3", lambda: c3_object(ctx))
        if not args.no_cpu_baseline and world == 1 and not sharded:
            def cpu():
                sample_reads = min(n, 400_000)
                host = ctx.download(rec, sample_reads * stride).reshape(sample_reads, stride)
                return cpu_baseline(host, sample_reads, k)

            out["cpu_baseline"] = extra("cpu_baseline", cpu)
        elif not args.no_cpu_baseline:
            out["cpu_baseline"] = None
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    m.close()
    if sharded:
        hd.barrier()
        hd.close()
        if world > 1:
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
