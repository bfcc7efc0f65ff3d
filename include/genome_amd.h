/*
 * genome_amd.h — C-ABI of the MI355X-native k-mer hashtable + de Bruijn graph-build core.
 *
 * This is the drop-in boundary for the data-parallel hot path of winger/genome
 * (S/ = /root/reference/src/main/scala/ru/ifmo/genome/).  Plain pointers and sizes only; no torch
 * types.  Each entry point names the reference interface it replaces.  The Scala-side binding a
 * maintainer would add (JNI stub + `HipDNAMap extends DNAMap[Int]`) is shown in INTEGRATION.md.
 *
 * Conventions
 *   - Every function returns a gk_status (0 = ok, <0 = error); the message of the last error of a
 *     context is available through gk_last_error().  Nothing aborts.  (Reference: `assert` =>
 *     AssertionError on wrong key length, S/ds/ArrayDNAMap.scala:182; failures surface via Future.)
 *   - Handles are opaque, created/destroyed by the caller and EXTERNALLY SYNCHRONISED: one call at
 *     a time per handle, from any thread (reference: an ArrayDNAMap is confined to its actor).
 *   - Calls are synchronous on return.  Host input buffers are borrowed for the duration of the
 *     call; export buffers are caller-allocated with a capacity and an out-count.
 *   - `_dev` variants take DEVICE pointers (hipMalloc'ed by the caller, e.g. a torch tensor's
 *     data_ptr()) valid on the context's device.
 *   - k-mers cross the ABI as little-endian uint64 lo[,hi] in the reference bit layout: base i at
 *     bits 2i of lo (i<32) / 2(i-32) of hi, codes A=0 G=1 C=2 T=3, unused high bits zero
 *     (S/dna/DNASeq.scala:74-215, S/dna/Base.scala:13-19).  `hi` arrays may be NULL when k<=32.
 *     Counts are int32 (the reference's DNAMap[Int]).
 *   - Supported k: 2..31 and 34..64, as in the reference: k=32/33 are broken in the reference itself
 *      and k>64 takes its un-specialised ArrayDNASeq path: GK_E_UNSUPPORTED_K.
 *   - There is no CPU fallback: without a gfx950 device every compute call fails with
 *     GK_E_NODEVICE / GK_E_HIP.
 *   - This is everything a host binds.  The library's test hooks (A/B switches of the kernels, failure injection, a loopback
 *     transport that runs several ranks on one GPU) are NOT in libgenome_amd.so: they live in the test build,
 *     libgenome_amd_test.so, behind include/genome_amd_test.h.
 */
#ifndef GENOME_AMD_H
#define GENOME_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    GK_OK = 0,
    GK_E_INVALID = -1,        /* bad argument (NULL handle, negative size, ...) */
    GK_E_KLEN = -2,           /* key length != the map's k  (ArrayDNAMap.scala:182,187,192,199,206) */
    GK_E_UNSUPPORTED_K = -3,  /* k outside 2..31, 34..64 */
    GK_E_CAPACITY = -4,       /* table could not grow / export buffer too small */
    GK_E_HIP = -5,            /* HIP runtime error (message has the hipError string) */
    GK_E_NODEVICE = -6,       /* no usable gfx950 device */
    GK_E_FORMAT = -7,         /* malformed `.bin` read stream */
    GK_E_STATE = -8,          /* operation not valid in the handle's current state */
    GK_E_COMM = -9            /* RCCL missing or a collective failed (message has the RCCL error string) */
} gk_status;

typedef struct gk_ctx gk_ctx;       /* one device + stream; replaces ActorsHome.system (S/scripts/ActorsHome.scala:20-30) */
typedef struct gk_map gk_map;       /* one ArrayDNAMap[Int] partition, resident in HBM */
typedef struct gk_graph gk_graph;   /* a MapGraph, resident in HBM */
typedef struct gk_prefilter gk_prefilter;   /* exact two-pass singleton pre-filter (2-bit counters) */
typedef struct gk_dist gk_dist;     /* one rank of a PartitionedDNAMap spread over the GPUs of a node (RCCL communicator) */

/* ---- context ---------------------------------------------------------------------------- */
int gk_device_count(void);                       /* number of HIP devices, 0 if none / no runtime */
int gk_ctx_create(int device, gk_ctx **out);
void gk_ctx_destroy(gk_ctx *ctx);
const char *gk_last_error(const gk_ctx *ctx);    /* ctx may be NULL: last error of the calling thread */
int gk_ctx_device(const gk_ctx *ctx);
int gk_ctx_sync(gk_ctx *ctx);                    /* hipStreamSynchronize on the context stream */
/* Device buffers of 1 MiB and more that the library frees (tables, key scratch, graph arrays, gk_dev_free) are parked in the
 * context — up to a third of the device's memory — and handed out again by size: hipMalloc / hipFree of multi-GB blocks cost
 * milliseconds to seconds, and freshly freed memory is scrubbed on the copy engines while the next upload wants them.
 * gk_ctx_trim gives everything parked back to the device (gk_map_trim does so too); gk_ctx_destroy always does. */
int gk_ctx_trim(gk_ctx *ctx);
/* Device memory this context holds in blocks of 1 MiB and more (tables, key scratch, graph arrays, gk_dev_alloc): *live = bytes
 * handed out now, *peak = their high-water mark since the last reset, *parked = bytes waiting in the block pool; any may be
 * NULL.  reset_peak != 0 restarts the high-water mark at the current value.  (What DESIGN.md's bytes-per-key tables are
 * checked against; the reference's counterpart is the JVM heap it logs, -Xmx in project/Build.scala:44.) */
int gk_ctx_mem_stats(gk_ctx *ctx, uint64_t *live, uint64_t *peak, uint64_t *parked, int reset_peak);
/* Plan as if the device had `bytes` of memory (0 = all of it, the default): the sizing decisions that look at free memory — the
 * load factor of the table the graph phase reads, the key scratch of an insert batch — then use min(free, bytes - live).  For a
 * GPU shared with other work, and for rehearsing an 8-GPU replica's budget at an eighth of its size. */
int gk_ctx_set_mem_budget(gk_ctx *ctx, uint64_t bytes);
/* Pinned host memory: gk_map_count_reads / gk_prefilter_add_reads read the caller's `.bin` buffer with asynchronous copies
 * that overlap the insert kernels only if the buffer is page-locked — allocate it here, or register an existing one (a JNI
 * direct ByteBuffer's address) for as long as it is passed in.  Pageable buffers work, at a staged copy's speed. */
int gk_host_alloc(gk_ctx *ctx, size_t nbytes, void **host_ptr);
int gk_host_free(gk_ctx *ctx, void *host_ptr);
int gk_host_register(gk_ctx *ctx, void *host_ptr, size_t nbytes);
int gk_host_unregister(gk_ctx *ctx, void *host_ptr);
/* raw device memory helpers for callers without their own allocator (tests, the C++ host side) */
int gk_dev_alloc(gk_ctx *ctx, size_t nbytes, void **dev_ptr);
int gk_dev_free(gk_ctx *ctx, void *dev_ptr);
int gk_dev_upload(gk_ctx *ctx, void *dev_dst, const void *host_src, size_t nbytes);
int gk_dev_download(gk_ctx *ctx, void *host_dst, const void *dev_src, size_t nbytes);
/* What this card streams, measured: gbps3 = {copy (bytes read + written), fill (written), sum (read)} in GB/s over two
 * scratch buffers of nbytes each, `reps` launches of each kernel.  The yardstick SURVEY.md §8(d) asks for beside the nominal
 * HBM peak (bench.py reports the pipeline's traffic rate against it); no reference counterpart. */
int gk_dev_stream_bench(gk_ctx *ctx, size_t nbytes, int reps, double *gbps3);

/* ---- DNAMap[Int]: trait at S/ds/ArrayDNAMap.scala:49-60 ----------------------------------- */
/* new ArrayDNAMap[Int](k)  (ArrayDNAMap.scala:62-72).  capacity_hint = expected number of distinct
 * keys (0 = default); the table is pre-sized from it and grows by rehashing when a batch could
 * push the load factor past 0.75 (the reference's 0.3/0.7 rescale, :217-230, is unobservable). */
int gk_map_create(gk_ctx *ctx, int k, uint64_t capacity_hint, gk_map **out);
/* The same for a map that is FILLED ONCE with `keys` keys (a merge of partitions, a load from a file) and then read by
 * gk_graph_build: its table is sized the way the graph phase wants it — as sparse as memory allows, load 0.25 down to 0.7
 *  — which is also how gk_dist_gather_map sizes the table it returns and gk_map_filter_lt the one it leaves. */
int gk_map_create_for_graph(gk_ctx *ctx, int k, uint64_t keys, gk_map **out);
void gk_map_destroy(gk_map *m);
int gk_map_k(const gk_map *m);
int gk_map_clear(gk_map *m);                               /* back to an empty table of the same capacity */
/* How batches are inserted.  0 = auto (cost model), 1 = direct: one global CAS/add per k-mer
 * (k_count_reads / k_add_keys), 2 = partitioned: radix-partition the batch by 64-KiB table segment
 * and build each segment in LDS (gk_partition.hip).  Results are identical; only speed differs. */
int gk_map_set_insert_path(gk_map *m, int path);
int gk_map_size(gk_map *m, uint64_t *n);                   /* DNAMap.size :50 (live keys) */
int gk_map_slots(gk_map *m, uint64_t *slots);              /* current table capacity in slots */
/* Invariants of the table, checked on the device (the reference prints `nodeMap.size` next to the expected total,
 * Graph.scala:117): *live = live slots (== gk_map_size), *bad_slots = keys that are stored twice or in a segment
 * their hash does not name (must be 0), *sum_counts = sum of all counts (== occurrences inserted), *checksum = an
 * order-independent 64-bit checksum of the (key, count) set: two maps hold the same table iff (live, sum, checksum)
 * agree — whatever their sizes, slot orders and insert histories (partitions of one table ADD UP: the checksum of a
 * PartitionedDNAMap is the sum of its partitions' checksums mod 2^64).  Any may be NULL. */
int gk_map_verify(gk_map *m, uint64_t *live, uint64_t *bad_slots, uint64_t *sum_counts, uint64_t *checksum);
/* Upper bound on the k-mer windows one partitioned insert batch holds when a call brings more windows than the
 * table has room for (the batch's key scratch is ~17 x W bytes per window); 0 = default (64 GiB of keys per buffer or as much scratch as the table itself holds, half of the free HBM at most). */
int gk_map_set_max_batch_keys(gk_map *m, uint64_t keys);
/* Release the scratch the handle keeps between calls (key buffers of the partitioned insert, staging of host streams,
 * point-query scratch).  The table is untouched; the next call allocates what it needs again. */
int gk_map_trim(gk_map *m);

/* FreqFilter.add over a stream of reads (S/data/FreqFilter.scala:28-36, 44-48):
 * for every read with len >= k, every window in order -> reverse complement -> orientation with
 * the smaller signed hashCode (tie: reverse complement) -> update(y, 1, _+1).
 * `bin` is the reference `.bin` record stream [len:u8][ceil(len/4) bytes] x nreads
 * (S/data/PairedEndData.scala:20-36); *occurrences (may be NULL) = number of windows counted.
 * The table grows as needed.  A call whose windows exceed the table's room is not assumed to bring that many NEW
 * keys (sequencing coverage: mostly repeats): the insert pipeline samples the distinct keys it sees and sizes the
 * table for those (gk_map_stats: "est_new_distinct_last_batch"). */
int gk_map_count_reads(gk_map *m, const uint8_t *bin_host, size_t nbytes, uint64_t nreads, uint64_t *occurrences);
/* Start uploading the head of a `.bin` stream NOW (asynchronously, on the copy stream, into the staging area the map is not
 * using) and return: the next gk_map_count_reads on THIS map whose stream starts at the same address then finds its first chunk
 * on the device and scatters it in one launch.  A streaming caller prefetches batch i+1 before it counts batch i: the upload
 * runs beside batch i's fine level.  (gk_map_count_reads does the same by itself between the chunks of ONE long stream.)  The
 * buffer must be page-locked (gk_host_alloc / gk_host_register) for the copy to be asynchronous, and must stay unchanged until
 * the count that consumes it has returned.  Reference: the reader thread of S/data/FreqFilter.scala:44-51 running ahead of
 * the inserts through the actors' mailboxes. */
int gk_map_prefetch_reads(gk_map *m, const uint8_t *bin_host, size_t nbytes, uint64_t nreads);
/* same, records already in HBM at a fixed stride 1+ceil(read_len/4); read_len = the LONGEST record (shorter ones
 * are fine).  Device records are untrusted input: a length byte above read_len is clamped, never followed, and
 * the call then fails with GK_E_FORMAT (the map's contents are unspecified: clear it). */
int gk_map_count_reads_dev(gk_map *m, const void *dev_records, uint64_t nreads, int read_len, uint64_t *occurrences);

/* DNAMap.update(key, 1, _+1) for a batch of keys taken verbatim (no canonicalisation):
 * PartitionedDNAMap's owner-side insert (Messages.update1, ArrayDNAMap.scala:39).  Keys are
 * W = 1 (k<=32) or 2 (k>32) uint64 each.  The map notices when a verbatim key is not the hash-rule orientation of
 * its k-mer (FreqFilter.scala:31-32): gk_graph_build then probes both strands for every `contains`, exactly as
 * Graph.scala:270 does, instead of the one probe a canonically filled table needs. */
int gk_map_update_inc(gk_map *m, const uint64_t *lo, const uint64_t *hi, uint64_t n);
/* device keys, interleaved W words per key ([lo] or [lo,hi]) as gk_shard_reads_dev emits them */
int gk_map_update_inc_dev(gk_map *m, const void *dev_keys, uint64_t n);
/* update(key, v0=c, f=_+c): adds counts of pre-aggregated keys (merging exported partitions) */
int gk_map_add_counts(gk_map *m, const uint64_t *lo, const uint64_t *hi, const int32_t *counts, uint64_t n);

/* update(key, c, _ + c) for every (key, c) of `src` (same context, same k), device to device in bounded chunks: merges the
 * partitions of a PartitionedDNAMap that share a device (the one-device counterpart of gk_dist_gather_map).  `src` is unchanged. */
int gk_map_add_map(gk_map *dst, gk_map *src);

/* DNAMap.deleteAll((k, v) => v < rounds)  (FreqFilter.scala:55; ArrayDNAMap.scala:164-173, 212-215) */
int gk_map_filter_lt(gk_map *m, int32_t rounds);

/* DNAMap.apply / contains for a batch (ArrayDNAMap.scala:90-101, 232): counts_out[i] = value or
 * -1 when absent; found_out[i] = 0/1.  Either output may be NULL.  One strand only, as the trait. */
int gk_map_get_batch(gk_map *m, const uint64_t *lo, const uint64_t *hi, uint64_t n, int32_t *counts_out, uint8_t *found_out);

/* Container.iterator / mapReduce(identity) (ArrayDNAMap.scala:175-178, 234-241): every live
 * (key, count) in slot order (unspecified; callers sort for comparison).  If cap < live count the
 * call fails with GK_E_CAPACITY and *n holds the required size. */
int gk_map_export(gk_map *m, uint64_t *lo, uint64_t *hi, int32_t *counts, uint64_t cap, uint64_t *n);

/* JSON counters: capacity, size, occurrences, grows, last kernel time ... (SURVEY.md §5 metrics) */
int gk_map_stats(gk_map *m, char *json, size_t cap);
/* duration (ms, HIP events on the context stream) and occurrence count of the most recent
 * insert+count kernel launched by gk_map_count_reads[_dev] — used by bench.py's roofline. */
int gk_map_last_count_kernel(gk_map *m, float *ms, uint64_t *occurrences);
/* per-phase device time (ms, HIP events) of the most recent insert: ms5 = {hist1, scatter1, hist2,
 * scatter2, seg_insert} for the partitioned path; {kernel, 0, 0, 0, 0} for the direct path. */
int gk_map_last_phase_ms(gk_map *m, float *ms5);

/* ---- PartitionedDNAMap: owner routing (S/ds/PartitionedDNAMap.scala:60-63) ----------------- */
/* Extract + canonicalise every k-mer of fixed-length device reads and bucket the canonical keys
 * by owner partition.  The owner is a strand-symmetric minimizer hash mod P (not `hashCode mod P`:
 * the partition function is unobservable in results), so both orientations of a
 * k-mer — and both candidates of the hash-rule tie — land on the same partition.
 * dev_keys_out receives the keys grouped by owner (W uint64 per key), capacity keys_cap keys;
 * counts_host[p] = number of keys for owner p (sum = occurrences).  Exchange the groups
 * (RCCL all-to-all) and feed what a rank receives to gk_map_update_inc_dev. */
int gk_shard_reads_dev(gk_ctx *ctx, int k, const void *dev_records, uint64_t nreads, int read_len, int P,
                       void *dev_keys_out, uint64_t keys_cap, uint64_t *counts_host);
/* The same routing, shipped as SUPER-K-MERS: every maximal run of consecutive same-owner windows of
 * a read becomes one record — the run's bases in the `.bin` framing [len:u8][2-bit bases] inside a
 * fixed slot of gk_skm_slot_bytes(k) bytes (16 for k<=31, 32 for k>=34; longer runs are split).
 * ~8x fewer bytes than 8/16-B keys cross xGMI; the owner re-extracts and canonicalises, exactly as
 * FreqFilter.add does on the original reads: feed what a rank receives to
 * gk_map_count_superkmers_dev.  dev_out is cut into P regions of out_cap_records / P slots; owner p's
 * records fill the start of region p; rec_counts_host[p] / kmer_counts_host[p] = records / windows
 * for owner p.  If a region is too small the call fails with GK_E_CAPACITY after filling the counts
 * (max of rec_counts_host = slots a region needs); retry with a larger buffer. */
int gk_skm_slot_bytes(int k);
int gk_shard_superkmers_dev(gk_ctx *ctx, int k, const void *dev_records, uint64_t nreads, int read_len, int P,
                            void *dev_out, uint64_t out_cap_records, uint64_t *rec_counts_host, uint64_t *kmer_counts_host);
int gk_map_count_superkmers_dev(gk_map *m, const void *dev_records, uint64_t nrecords, uint64_t kmers_total, uint64_t *occurrences);
/* owner of one key under the same function (host-side, for tests and for routing point queries) */
int gk_owner_of(int k, uint64_t lo, uint64_t hi, int P);
/* *foreign = live keys of partition p's table whose owner (of P) is not p — 0 for a correctly routed PartitionedDNAMap */
int gk_map_count_foreign(gk_map *m, int P, int p, uint64_t *foreign);

/* ---- DNAMap[T] with values and multimap inserts: the rest of the trait (S/ds/ArrayDNAMap.scala:49-60) -------------- */
/* A second kind of map, for T = a 64-bit value (GraphPosition, Long): putNew stores a key as often as it is put, getAll
 * returns every value stored under a key, update(key, v) inserts or overwrites, apply returns the first value in probe order.
 * Same HBM table layout as gk_map; nothing is deleted from these maps.  Used by gk_graph_position_map (Graph.getGraphMap). */
typedef struct gk_vmap gk_vmap;
int gk_vmap_create(gk_ctx *ctx, int k, uint64_t capacity_hint, gk_vmap **out);     /* new PartitionedDNAMap[GraphPosition](k), Graph.scala:92 */
void gk_vmap_destroy(gk_vmap *m);
int gk_vmap_k(const gk_vmap *m);
int gk_vmap_size(gk_vmap *m, uint64_t *n);                                         /* DNAMap.size :50 — entries, duplicates included */
/* DNAMap.putNew(key, v) (:55; Container.putNew ArrayDNAMap.scala:152-162), batched: no duplicate check — a multimap.
 * BOUND (the reference has none): all copies of a key live in the one 32 KiB / 24 KiB segment its hash names, so one key can be
 * stored at most 2048 times (k <= 31), 1024 times (34 <= k <= 63) or 256 times (k = 64), less what else shares the segment;
 * beyond that the call fails with GK_E_CAPACITY, the batch's other entries stay stored (a failed batch is NOT rolled back;
 * gk_vmap_size tells what is in).  Graph.getGraphMap stores a k-mer once per graph position: far from the bound. */
int gk_vmap_put_new_batch(gk_vmap *m, const uint64_t *lo, const uint64_t *hi, const uint64_t *values, uint64_t n);
/* DNAMap.update(key, v) (:53; Container.update :115-127), batched: insert, or overwrite the first entry of the key; for a key
 * that occurs several times in one batch the LAST occurrence wins, as in the reference's sequential loop */
int gk_vmap_update_batch(gk_vmap *m, const uint64_t *lo, const uint64_t *hi, const uint64_t *values, uint64_t n);
/* DNAMap.getAll(key) (:52; Container.getAll :103-113), batched, CSR output: the values of key i are
 * values_out[offsets_out[i] .. offsets_out[i+1]) (offsets_out has n + 1 entries); *total = offsets_out[n].  If values_cap <
 * *total the offsets are still filled and the call fails with GK_E_CAPACITY.  Order inside a key's run: most recently probed
 * first, as the reference's list — callers treat it as a set (GraphSimplifier.scala:192-217). */
int gk_vmap_get_all_batch(gk_vmap *m, const uint64_t *lo, const uint64_t *hi, uint64_t n, uint64_t *offsets_out, uint64_t *values_out,
                          uint64_t values_cap, uint64_t *total);
/* DNAMap.apply(key) (:51), batched: first value in probe order; found_out[i] = 0/1 (may be NULL) */
int gk_vmap_get_batch(gk_vmap *m, const uint64_t *lo, const uint64_t *hi, uint64_t n, uint64_t *values_out, uint8_t *found_out);
/* Container.iterator: every (key, value), unspecified order */
int gk_vmap_export(gk_vmap *m, uint64_t *lo, uint64_t *hi, uint64_t *values, uint64_t cap, uint64_t *n);

/* GraphPosition (S/data/graph/GraphPosition.scala) as a 64-bit value: NodeGraphPosition(id) = id;
 * EdgeGraphPosition(id, dist) = 1<<63 | id<<32 | dist.  Ids are this library's node / edge ids (gk_graph_*_by_id). */
#define GK_POS_IS_EDGE(v) (((v) >> 63) != 0)
#define GK_POS_ID(v) ((uint32_t)(GK_POS_IS_EDGE(v) ? (((v) >> 32) & 0x7fffffffu) : ((v) & 0xffffffffu)))
#define GK_POS_DIST(v) ((uint32_t)((v) & 0xffffffffu))

/* ---- PartitionedDNAMap over N GPUs: one rank (process) per GPU, RCCL over xGMI ---------------- */
/* Reference: `new PartitionedDNAMap[Int](k)` deploys one ArrayDNAMap actor per storage node and sends one message per k-mer
 * occurrence to its owner (S/ds/PartitionedDNAMap.scala:20-47).  Here every rank holds ONE partition (a gk_map it created on
 * its own context), routes the k-mers of ITS reads as super-k-mer records and exchanges them with one all-to-all per call.
 * Bootstrap: rank 0 calls gk_dist_unique_id and hands the 128 bytes to the other ranks by any means the host has (the JVM
 * driver's own channel, a file, MPI, torch.distributed in bench.py); every rank then calls gk_dist_create — collectively.
 * RCCL is loaded on first use (dlopen); without it these calls fail with GK_E_COMM and nothing else is affected.
 * All gk_dist_* calls that move data are COLLECTIVE: every rank of the communicator must make them, in the same order. */
int gk_dist_unique_id(void *id128);                                   /* out: 128 bytes */
int gk_dist_create(gk_ctx *ctx, int rank, int world, const void *id128, gk_dist **out);
void gk_dist_destroy(gk_dist *d);
int gk_dist_rank(const gk_dist *d);
int gk_dist_world(const gk_dist *d);
int gk_dist_barrier(gk_dist *d);
/* all-reduce of up to 32 doubles in place (op_max: 0 = sum, 1 = max) — for the host's own bookkeeping (timings, totals) */
int gk_dist_allreduce_f64(gk_dist *d, double *values, int n, int op_max);
/* FreqFilter.add over THIS rank's reads, every k-mer counted by its owner rank (PartitionedDNAMap.update, :41-43):
 * route (gk_shard_superkmers_dev, P = world) -> exchange counts -> exchange records -> gk_map_count_superkmers_dev on `local`.
 * *occurrences_sent = windows of this rank's reads, *occurrences_owned = windows this rank counted (sums over ranks agree). */
int gk_dist_count_reads_dev(gk_dist *d, gk_map *local, const void *dev_records, uint64_t nreads, int read_len,
                            uint64_t *occurrences_sent, uint64_t *occurrences_owned);
/* The same in two halves, for a streaming loop: gk_dist_route_begin launches the routing of a batch on the context's second
 * stream and returns at once; gk_dist_count_routed waits for it, exchanges and counts.  Calling route_begin for batch i+1
 * BEFORE count_routed for batch i overlaps the routing kernel with the owner pipeline; with route_begin for batch i+2 before
 * it as well, count_routed(i) first posts the exchange of batch i+1 on the handle's own communication stream, so that those
 * records travel over xGMI while batch i is counted (three send buffers, two receive buffers; the rule depends only on the
 * number of begun batches, so every rank issues the same sequence of RCCL operations — all ranks must run the same loop).
 * The records of a begun batch must stay valid until its count_routed returns.  At most three batches may be begun and not
 * yet counted (GK_E_STATE otherwise; count_routed with none begun is GK_E_STATE too); they are counted first in, first out:
 * begin(0), begin(1), [begin(i+2), count_routed(i)]*, count_routed(last-1), count_routed(last).  A batch whose exchange
 * fails is dropped and the error returned by the count_routed whose turn it is.  A malformed record in a begun batch
 * (GK_E_FORMAT) may be reported by whichever count call on the context checks the flags next. */
int gk_dist_route_begin(gk_dist *d, int k, const void *dev_records, uint64_t nreads, int read_len);
int gk_dist_count_routed(gk_dist *d, gk_map *local, uint64_t *occurrences_sent, uint64_t *occurrences_owned);
/* wall ms of the last gk_dist_count_routed on this rank: {waiting for the route, exchange, owner count, total} */
int gk_dist_last_ms(gk_dist *d, float *ms4);
int gk_dist_size(gk_dist *d, gk_map *local, uint64_t *total);         /* PartitionedDNAMap.size (:31): sum over the partitions */
/* deleteAll / filter_lt, stats, export are LOCAL: call gk_map_filter_lt etc. on `local` on every rank (:49-51 scatter, no data moves). */
/* The whole k-mer set on every rank, for Graph.buildGraph (the unitig walk crosses partitions arbitrarily):
 * all-gather of every partition's live (key, count), device to device and in bounded chunks (staging <= 0.7 GB to send, world x
 * that to receive), into a NEW map (*full, caller destroys it) sized the way the graph phase wants it.  A rank that fails
 * says so in the chunk's size word: every rank then returns an error for the same chunk and nobody waits in a receive. */
int gk_dist_gather_map(gk_dist *d, gk_map *local, gk_map **full);
/* The same gather with the classify of Graph.buildGraph done by the keys' OWNERS first (Graph.scala:320-329 shipped to every
 * partition, PartitionedDNAMap.scala:55-58; SURVEY.md section 8e "beyond counting"): every rank looks up the neighbours it owns
 * in its own partition and asks the other neighbours of their owners (per chunk: one all-to-all of canonical keys, one of answer
 * bytes), and each key's (incoming, outcoming) mask travels with it into *full.  gk_graph_build on *full then derives the
 * terminal k-mers from the masks in one streaming pass — the eight lookups per key are done once in the whole job instead of once
 * per rank.  The masks serve the FIRST gk_graph_build on *full; any change of its contents drops them (plain classify again).
 * Partitions that hold verbatim non-canonical keys on any rank: plain gather on every rank.  `local` is left in the graph
 * layout; its contents are unchanged.  Same failure behaviour as gk_dist_gather_map. */
int gk_dist_gather_classified_map(gk_dist *d, gk_map *local, gk_map **full);
/* neighbour lookups this rank has asked of other ranks in classified gathers since the handle was created */
int gk_dist_classify_queries(gk_dist *d, uint64_t *n);

/* ---- Graph: S/data/graph/Graph.scala ------------------------------------------------------- */
/* Graph.buildGraph(k, kmersFreq) (:269-382): degree classification of every live key through
 * `contains` on both strands, one node per terminal k-mer (both strands), one edge per
 * (node, outgoing base) walked to the next terminal k-mer.  The map must hold the whole k-mer set
 * (merge partitions with gk_map_export + gk_map_add_counts first). */
int gk_graph_build(gk_map *m, gk_graph **out);
void gk_graph_destroy(gk_graph *g);
int gk_graph_counts(gk_graph *g, uint64_t *nodes, uint64_t *edges, uint64_t *total_edge_len); /* live */
int gk_graph_simplify(gk_graph *g);          /* MapGraph.simplifyGraph :211-230 */
int gk_graph_remove_bubbles(gk_graph *g);    /* Graph.removeBubbles :125-149 */
/* MapGraph.removeEdge :191-195 for the edges leaving start[i] with first base base[i] */
int gk_graph_remove_edges(gk_graph *g, const uint64_t *start_lo, const uint64_t *start_hi, const uint8_t *base,
                          uint64_t n, uint64_t *removed);
/* Graph.components + retain(maxBy size) (:54-72, :161-165; GraphBuilder.scala:52-54); ties between
 * equal-size components go to the one holding the smallest k-mer. */
int gk_graph_retain_largest(gk_graph *g, uint64_t *kept_nodes, uint64_t *components);
/* GraphBuilder's two component histograms (GraphBuilder.scala:41-47) are group-bys of these two arrays: one entry per
 * connected component, nodes_per_component[i] = comp.size, edge_len_per_component[i] = sum of seq.size over the out-edges
 * of its nodes; order unspecified.  If cap < *n the call fails with GK_E_CAPACITY and *n holds the required size. */
int gk_graph_component_stats(gk_graph *g, uint32_t *nodes_per_component, uint64_t *edge_len_per_component, uint64_t cap, uint64_t *n);
/* Order-independent 64-bit checksums of the canonical serialisation: the node k-mer set, and the edge set
 * as (start k-mer, end k-mer, length, every base).  Two graphs with equal counts and checksums are the same graph. */
int gk_graph_checksum(gk_graph *g, uint64_t *nodes_checksum, uint64_t *edges_checksum);
/* How gk_graph_build spent its time: phase_ms6 = {degree classification (k_classify), terminals -> nodes + edge stubs,
 * unitig measurement (k_walk pass 0 or pointer jumping), pool reservation, unitig emission, node index + counts} (wall ms,
 * every phase ends in a stream sync); *walked_bases = bases emitted; *pointer_jumping = 1 if k_pj_* built the unitigs. */
int gk_graph_build_stats(gk_graph *g, float *phase_ms6, uint64_t *walked_bases, int *pointer_jumping);
/* When gk_graph_build ran on a minimizer-bucketed copy of the table (gk_ctx_set_option "graph_mbt" = 1): wall ms of building the
 * copy and its slots; 0 / 0 otherwise.  Diagnostics of an A/B switch. */
int gk_graph_bucketed_table_stats(gk_graph *g, float *build_ms, uint64_t *slots);
/* *flag = 1 when gk_graph_build took the degree masks that came with the table (gk_dist_gather_classified_map) instead of
 * running the neighbour lookups itself */
int gk_graph_classified_by_owners(gk_graph *g, int *flag);
/* Graph.getGraphMap (Graph.scala:90-119): putNew of every node's k-mer -> NodeGraphPosition(node id) and of the k-mers at
 * distance 1 .. len-1 along every edge -> EdgeGraphPosition(edge id, dist) into `vm` (same k, same context).  *entries =
 * number of entries added = sum of edge lengths + nodes - edges (the reference prints both side by side, :117; here the
 * equality is checked). */
int gk_graph_position_map(gk_graph *g, gk_vmap *vm, uint64_t *entries);
/* Ids.  Node and edge ids are array indices (arbitrary, like the reference's AtomicLong ids; stable for a graph's lifetime).
 * gk_graph_node_lookup: the live node with this k-mer (smallest id if a node split left several) and, for base in 0..3, the id
 * of its out-edge whose sequence starts with that base; 0xffffffff = none. */
int gk_graph_node_lookup(gk_graph *g, uint64_t lo, uint64_t hi, int base, uint32_t *node_id, uint32_t *edge_id);
int gk_graph_nodes_by_id(gk_graph *g, const uint32_t *ids, uint64_t n, uint64_t *lo, uint64_t *hi, uint8_t *alive, uint32_t *in_deg, uint32_t *out_deg);
int gk_graph_edges_by_id(gk_graph *g, const uint32_t *ids, uint64_t n, uint32_t *start_node, uint32_t *end_node, uint64_t *len,
                         uint8_t *first_base, uint8_t *alive);
/* MapGraph.addNode(seq) (:172-176): a new node without edges (a node split creates nodes that share a sequence) */
int gk_graph_add_node(gk_graph *g, uint64_t lo, uint64_t hi, uint32_t *node_id);
/* MapGraph.replaceStart / replaceEnd (:197-209): re-attach one end of an edge to another node */
int gk_graph_replace_start(gk_graph *g, uint32_t edge_id, uint32_t new_start_node);
int gk_graph_replace_end(gk_graph *g, uint32_t edge_id, uint32_t new_end_node);
/* ids run from 0 to these bounds (dead nodes / edges keep theirs) */
int gk_graph_id_bounds(gk_graph *g, uint64_t *node_ids, uint64_t *edge_ids);
/* MapGraph.removeEdge (:191-195) by edge id (each id once, as the reference's `toRemove` Set, GraphSimplifier.scala:270,316) */
int gk_graph_remove_edges_by_id(gk_graph *g, const uint32_t *edge_ids, uint64_t n, uint64_t *removed);

/* ---- paired-end walking: GraphSimplifier.startup (S/scripts/GraphSimplifier.scala:188-318) ------------------------------
 * gk_support = the reference's pathsMap (:209: (edge id, edge id) -> number of read pairs whose walk passes through the two
 * edges one after the other) and badPairs (:211).  It accumulates over calls of gk_graph_walk_pairs. */
typedef struct gk_support gk_support;
int gk_support_create(gk_ctx *ctx, gk_support **out);
void gk_support_destroy(gk_support *s);
int gk_support_size(const gk_support *s, uint64_t *pairs, uint64_t *bad_pairs, uint64_t *walked_orientations);
/* wall ms of the last gk_graph_walk_pairs into this support: {keys cut from the stream, getAll batch, graph snapshot + checks,
 * walks, merge of the per-thread counts} */
int gk_support_last_ms(const gk_support *s, float *ms5);
int gk_support_export(const gk_support *s, uint32_t *e1, uint32_t *e2, uint32_t *count, uint64_t cap, uint64_t *n);   /* unordered */
/* :213-247 for the first `npairs` pairs of a `.bin` stream (two records per pair; pairs with a mate shorter than k are
 * skipped, :213).  `positions` = gk_graph_position_map of THIS graph in its current state (GK_E_STATE otherwise).  For each
 * pair the four getAll (:214-217) run as one batch on the device; `annotate` (:192-206) drops an orientation whose mates lie
 * on one edge at a distance inside [range_lo, range_hi]; every (pos1, pos2) is walked as WalkingActor does (:78-125:
 * reachable set bounded by range_hi, paths whose length puts the mates range_lo..range_hi apart) on host threads, over a
 * snapshot of the graph's edge arrays; the edge pairs on successful paths are counted once per pair orientation.  The
 * reference's range is 180 to 250 (:146). */
int gk_graph_walk_pairs(gk_graph *g, gk_vmap *positions, gk_support *sup, const uint8_t *bin, size_t nbytes, uint64_t npairs, int range_lo,
                        int range_hi);
/* :272-316: for every node with in- and out-edges the matrix support[in][out]; its connected groups at `cutoff` (genome.cutoff);
 * a group without an out-edge has its in-edge removed, every other group moves to a copy of the node (addNode, replaceEnd,
 * replaceStart); out-edges that no group reached are removed.  Follow with gk_graph_simplify (:318).  Node ids: copies are
 * appended; nodes that existed when the call started are the ones visited (the reference iterates a live map: unspecified). */
int gk_graph_split_by_support(gk_graph *g, const gk_support *sup, int cutoff, uint64_t *removed_edges, uint64_t *new_nodes);
/* live nodes, unspecified order */
int gk_graph_export_nodes(gk_graph *g, uint64_t *lo, uint64_t *hi, uint64_t cap, uint64_t *n);
/* live edges, unspecified order: start/end k-mer, length in bases, and the edge sequence as 2-bit
 * codes packed LSB-first (4 per byte, each edge starting on a byte boundary at seq_off[i]). */
int gk_graph_export_edges(gk_graph *g, uint64_t *start_lo, uint64_t *start_hi, uint64_t *end_lo, uint64_t *end_hi,
                          int64_t *len, int64_t *seq_off, uint64_t cap, uint64_t *n,
                          uint8_t *seq2bit, uint64_t seq_cap, uint64_t *seq_bytes);
/* out-edge insertion order of one node (the reference's immutable Map1..Map4 order, used by
 * removeBubbles); bases4 gets up to 4 base codes, *count their number (-1 = no such node) */
int gk_graph_out_order(gk_graph *g, uint64_t lo, uint64_t hi, int *bases4, int *count);

/* ---- exact two-pass singleton pre-filter -------------------------- */
/* Analogue of the reference's unused Bloom filter (S/ds/BloomFilter.scala:17-70) in front of
 * FreqFilter.add (S/data/FreqFilter.scala:28-36): an array of 2-bit saturating counters, 4 per
 * expected distinct k-mer (1 byte per k-mer instead of a 16/32-byte table slot).
 *   pass 1: gk_prefilter_add_reads[_dev] over EVERY read that will be counted;
 *   pass 2: gk_map_count_reads_prefiltered[_dev] over the same reads: a window enters the table only
 *           if its counter says "seen at least twice".
 * Every k-mer with true count >= 2 then holds its exact count; a k-mer seen once is either absent or
 * present with count 1.  So for rounds >= 2, gk_map_filter_lt(rounds) leaves exactly the table that
 * plain counting + filter_lt(rounds) leaves (FreqFilter.scala:55) — whatever the filter's size. */
int gk_prefilter_create(gk_ctx *ctx, int k, uint64_t expected_distinct, gk_prefilter **out);
void gk_prefilter_destroy(gk_prefilter *pf);
int gk_prefilter_add_reads(gk_prefilter *pf, const uint8_t *bin_host, size_t nbytes, uint64_t nreads);
int gk_prefilter_add_reads_dev(gk_prefilter *pf, const void *dev_records, uint64_t nreads, int read_len);
/* *occurrences = windows looked at, *admitted = windows inserted (either may be NULL) */
int gk_map_count_reads_prefiltered(gk_map *m, gk_prefilter *pf, const uint8_t *bin_host, size_t nbytes, uint64_t nreads,
                                   uint64_t *occurrences, uint64_t *admitted);
int gk_map_count_reads_prefiltered_dev(gk_map *m, gk_prefilter *pf, const void *dev_records, uint64_t nreads, int read_len,
                                       uint64_t *occurrences, uint64_t *admitted);
/* counters in state "once" / "twice or more", buckets in total, windows fed to pass 1 (any may be NULL) */
int gk_prefilter_stats(gk_prefilter *pf, uint64_t *buckets, uint64_t *seen_once, uint64_t *seen_twice_or_more, uint64_t *windows_added);

/* ---- synthetic reads (bench / tests) ---------------------------------------- */
/* Fill dev_records with nreads fixed-length `.bin` records generated on device, bit-identical to
 * genome_amd/synth.py.  mode 0 = U (uniform bases); mode 1 = G (genome of G bases, error rate
 * err_thresh24 / 2^24).  first_read offsets the stream so chunks can be generated independently. */
int gk_synth_reads_dev(gk_ctx *ctx, void *dev_records, uint64_t nreads, int read_len, int mode, uint64_t config_id,
                       uint64_t first_read, uint64_t genome_len, uint32_t err_thresh24);

#ifdef __cplusplus
}
#endif
#endif
