#!/usr/bin/env python3
"""bench.py — distinct k-mers/s inserted (k=31, 150 bp synthetic reads) on N MI355X.

One "step" = one pass of the hot path over one batch of synthetic reads that is already resident
in HBM: clear the table, then FreqFilter.add for every read (extract -> canonicalise -> insert +
count).  Workload = BASELINE.json configs[1] ("C2"): 1M x 150 bp reads per GPU, k=31.
  N = 1 : single-partition DNAMap kernel (gk_map_count_reads_dev).
  N > 1 : weak scaling, one rank per GPU: every rank owns 1M reads and one table partition; k-mers are
          routed by strand-symmetric minimizer owner as SUPER-K-MER records (gk_shard_superkmers_dev:
          runs of same-owner windows, 2 bits/base in 16-B slots), exchanged with ONE RCCL all-to-all
          (torch.distributed, backend nccl) and counted by their owner with the same pipeline as
          reads (gk_map_count_superkmers_dev).  value = distinct k-mers over all partitions / max-over-ranks time.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402  (first: its bundled HIP runtime must be the process's only one)
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


def main():
    # stdout carries exactly ONE JSON line: native libraries (RCCL prints a version banner) write to
    # fd 1 behind Python's back, so fd 1 is pointed at stderr until the result is ready.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
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
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} needs WORLD_SIZE={args.gpus} (launch with torch.distributed.run)", file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    sharded = world > 1 or args.sharded
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    from genome_amd import synth
    from genome_amd.dnamap import Context, HipDNAMap
    from genome_amd.dnamap import skm_slot_bytes
    from genome_amd.partitioned import exchange_records

    n, L, k = args.reads, args.read_len, args.k
    W = 1 if k <= 32 else 2
    nk = L - k + 1
    stride = synth.record_stride(L)
    ctx = Context(local_rank)
    rec = torch.empty(n * stride + 64, dtype=torch.uint8, device=dev)
    G, err = 5_000_000 * world, 0.01
    ctx.synth_reads(rec.data_ptr(), n, L, args.mode, 2, rank * n, G, err)     # config_id 2 = C2
    occ_rank = n * nk
    m = HipDNAMap(ctx, k, int(occ_rank * 1.05 * float(os.environ.get('GK_HINT_SCALE', '1'))))
    m.set_insert_path(args.insert_path)
    if sharded:
        slot = skm_slot_bytes(k)
        send_cap = n * 24 // world * world                       # records, world regions of send_cap/world slots
        send = torch.empty(send_cap * slot, dtype=torch.uint8, device=dev)
        recv = torch.empty(send_cap * slot, dtype=torch.uint8, device=dev)

    kernel_ms, kernel_units, phase_ms = [], [], []

    def step():
        m.clear()
        if not sharded:
            m.count_reads_dev(rec.data_ptr(), n, L)
            ms, kocc = m.last_count_kernel()
            kernel_ms.append(ms); kernel_units.append(kocc); phase_ms.append(m.last_phase_ms())
            return
        nonlocal recv
        recs, kmers = ctx.shard_superkmers(k, rec.data_ptr(), n, L, world, send.data_ptr(), send_cap)
        recv, nrec, nrecv = exchange_records(dist, send, recs, kmers, slot, send_cap // world, recv)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m.count_superkmers_dev(recv.data_ptr(), nrec, nrecv)
        kernel_ms.append((time.perf_counter() - t0) * 1e3); kernel_units.append(nrecv); phase_ms.append(m.last_phase_ms())

    def fence():
        torch.cuda.synchronize()
        ctx_sync()
        if sharded:
            dist.barrier()
            torch.cuda.synchronize()

    def ctx_sync():
        from genome_amd import _lib
        _lib.check(_lib.lib().gk_ctx_sync(ctx.h), ctx.h)

    for _ in range(args.warmup):
        step()
    kernel_ms.clear(); kernel_units.clear(); phase_ms.clear()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0

    distinct_rank = m.size()
    tt = torch.tensor([dt, float(distinct_rank), float(occ_rank)], dtype=torch.float64, device=dev)
    if sharded:
        tmax = tt.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = tt.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt_max, distinct_total, occ_total = float(tmax[0]), float(tsum[1]), float(tsum[2])
    else:
        dt_max, distinct_total, occ_total = dt, float(distinct_rank), float(occ_rank)

    if rank == 0:
        ms_per_step = dt_max / args.steps * 1e3
        units = float(np.mean(kernel_units))
        phases = np.mean(np.array(phase_ms), axis=0)
        stats = m.stats()
        partitioned = stats["partitioned_launches"] > 0
        slot_b = 16 if W == 1 else 32
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
        pmc_file = os.path.join(ROOT, "profiles", "r01", "pmc_pipeline_v8.json" if partitioned else "pmc_count_reads_v2.json")
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
                       "table_slots_per_gpu": m.slots(), "slot_bytes": 16 if W == 1 else 32,
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
        if not args.no_cpu_baseline and world == 1 and not sharded:
            sample_reads = min(n, 400_000)
            host = ctx.download(rec.data_ptr(), sample_reads * stride).reshape(sample_reads, stride)
            out["cpu_baseline"] = cpu_baseline(host, sample_reads, k)
        elif not args.no_cpu_baseline:
            out["cpu_baseline"] = None
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    m.close()
    if sharded:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
