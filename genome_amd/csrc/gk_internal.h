// gk_internal.h — host-side handle definitions shared by the C-ABI translation units.
#pragma once

#include <hip/hip_runtime.h>

#include <map>
#include <mutex>
#include <string>
#include <unordered_map>

#include "../../include/genome_amd.h"
#include "gk_device.h"

struct gk_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t pev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // phase boundaries of the partitioned path
    hipEvent_t gev2 = nullptr;                    // joins the second stream back into the first (striped P4/P5)
    hipEvent_t gev = nullptr;                     // "the host is back": the fine level's first launch (the GPU idles from pev[2] to here)
    bool copy_other_pending = false;              // the copy stream carries work that touches POOLED blocks (not just uploads into staging areas): pool_free waits for it
    hipStream_t copy_stream = nullptr;            // host -> device copies that overlap kernels on `stream` (host-fed inserts)
    hipEvent_t cev[16] = {};                      // "sub-chunk j has landed" (round robin; 8..15: "piece j is scattered")
    hipStream_t aux_stream = nullptr;             // P4 of the pieces of a pipelined batch, beside the next piece's scatter on `stream`
    int cu_count = 256;
    void *skm_counts = nullptr;      // device scratch of gk_shard_superkmers_dev (cursors, counts, overflow flag)
    uint32_t *d_flags = nullptr;     // [0] = a device record's length byte exceeded the declared read length (kernels without a map)
    // test / A-B hooks, read from the environment ONCE here (never on a hot path; a stray variable in the host JVM
    // cannot flip behaviour mid-run): GK_TEST_NO_RESERVE, GK_HOST_RAGGED, GK_PART_EXACT, GK_GRAPH_UNITIGS
    bool hook_no_reserve = false, hook_host_ragged = false, hook_part_exact = false;
    int hook_unitigs = 0;            // 0 auto, 1 walk, 2 pointer jumping
    int hook_p4_direct = -1;         // exact fine level: -1 auto (by nb2), 0 chunk sorted in LDS, 1 straight scatter with per-range cursors
    int hook_p2_wide = -1;           // over-provisioned L1 scatter: 1 = 1024 threads per tile (A/B)
    int hook_p2_sorted = -1;         // over-provisioned L1 scatter: 1 = bucket-ordered write-out (A/B)
    int hook_max_nb2 = 0;            // test hook: the partitioned path refuses tables of more fine buckets per L1 bucket than this (0: MAX_NB2), so that
                                     // a batch that outgrows the fan-out between its levels can be staged with a small table
    int hook_min_lnb1 = 0;           // test hook: tables of enough segments get at least 2^this L1 buckets (9, 10: the fan-out of tables beyond 34 GB)
    int hook_dist_small_send = 0;    // test hook: the next gk_dist_route_begin on this context gets a send buffer of so many records (forces the in-place re-route)
    int hook_target_load_pct = 0;    // A/B: load factor new tables are sized for, in percent (0: 65)
    int hook_cc_find = 3;            // A/B: the components' link pass (gk_graph.hip: cc_find / k_cc_link): 3 = path halving + look before the CAS (default);
                                     // 0 = halving only, 1 = no path writes, 2 = only the start node is re-pointed
    int hook_dist_fail_classify = 0; // test hook: this context's next classified gather cannot stage its queries (GK_E_CAPACITY after the sizes round)
    int hook_dist_fail = 0;          // test hook: this context's next exchange fails locally with this code before anything is posted (the peers must drop the batch too)
    int hook_dist_ahead = -1;        // gk_dist_count_routed with three batches begun: 0 = do not post the next batch's exchange ahead (every rank alike)
    int hook_p24_pieces = -1;        // pipelined batch (both levels over-provisioned): pieces whose P4 overlaps the next piece's scatter (-1: default, 0/1: off)
    int hook_p45_stripes = -1;       // over-provisioned fine level: stripes of L1 buckets whose P5 overlaps the next stripe's P4 (-1/1: none)
    int hook_graph_load_pct = -1;    // load factor (percent) of the table map_compact builds for the graph phase (-1: 40, 30 at k = 64)
    int hook_filter_classic = -1;    // deleteAll: 1 = tombstones + k_rehash (the older path), else the one-pass segment-wise filter + compaction
    int64_t hook_max_stage = 0;      // test hook: bytes of `.bin` records per staging area (0: 704 MiB), so that a small stream is cut into many chunks
    int hook_host_prefetch = -1;     // host-fed count: 0 = do not upload the next chunk beside the current chunk's P3 / P4 / P5 (A/B)
    int hook_pairs_host = -1;        // paired-end walks: 1 = all of them on host threads over a snapshot (the round-2 form), else one wave per pair orientation
    int hook_pairs_small_sets = -1;  // test hook: the device walks get tiny LDS sets, so that most orientations overflow to the host walker
    int hook_graph_mbt = -1;         // gk_graph_build: 1 = classify and walk on a minimizer-bucketed copy of the table (A/B)
    int hook_graph_mbt_keys = 256;   // ... keys per bucket on average
    int hook_walk_queue = -1;        // unitig walk: 0 = one edge per lane (k_walk pass 0), else lanes fed from a queue (k_walk_q)
    int hook_p4_grid = -1;           // over-provisioned fine level: P4 workgroups per CU (-1: 4, or 2 of the 1024-thread form)
    int hook_p4_wide = -1;           // exact fine level, 8-byte keys: -1 auto (by nb2), 0 sort 4096 keys at a time, 1 sort 8192 (1024 threads)
    int hook_fine_exact = -1;        // -1 auto, 0 never unless forced by the data path, 1 always (A/B of the two fine levels)
    // Block pool (gk::pool_malloc / pool_free): the big device buffers — tables, key scratch, graph arrays — are handed back
    // here instead of hipFree and reused by size.  hipMalloc / hipFree of multi-GB blocks cost milliseconds to seconds on this
    // platform and freed VRAM is scrubbed in the background on the copy engines: a second count over the same map paid 20 ms of
    // host time between the levels, and host-fed inserts right after a map was destroyed ran their uploads at half speed.
    std::multimap<size_t, void *> pool_free_blocks;          // by size
    std::unordered_map<void *, size_t> pool_sizes;           // every block the pool has handed out or holds
    size_t pool_held = 0, pool_limit = 0;                    // bytes parked; cap (a third of the device's memory)
    size_t pool_live = 0, pool_peak = 0;                     // bytes handed out (blocks of 1 MiB and more) and their high-water mark (gk_ctx_mem_stats)
    size_t mem_budget = 0;                                   // gk_ctx_set_mem_budget: plan as if the device had this much memory (0: all of it)
    uint64_t pool_hits = 0, pool_misses = 0;
    std::recursive_mutex pool_mu;                            // (gk_dist's helper thread grows its buffers beside the owner pipeline)
    // the path-choice model's coefficients for THIS device (gk_table.hip: path_cost): 0 not measured yet, 1 measured, 2 reference values
    int cost_state = 0;
    alignas(8) unsigned char cost_blob[96] = {};
    double measured_copy_tbps = 0, measured_cas_gps = 0;
    // what the same two kernels measure on the box of the round 1-2 profiles (64 MiB working sets: partly Infinity Cache)
    double ref_copy_tbps = 5.9, ref_cas_gps = 26.7;        // (measured by these kernels in run 6 of round 3: 5.91 TB/s, 26.66 G/s)
    std::string err;
};

namespace gk {
// A stream of `.bin` records resident in HBM (PairedEndData.scala:20-36): fixed stride (off == nullptr) or an offset
// table built by the host while it walked the framing.  max_len = the longest read the buffers downstream are sized
// for: kernels clamp a larger length byte to it and raise the format flag (gk_tile.h WindowLimits).
struct ReadSrc {
    const uint8_t *rec = nullptr;
    uint64_t nreads = 0;
    const uint32_t *off = nullptr;
    uint32_t stride = 0;
    int group = 64;          // lanes per read in the window loops (64 reads, 32/16 short records)
    int max_len = 255;
    // host-fed: the records are still in HOST memory at `host` (stride x nreads bytes) and `rec` is the device staging area they
    // go to.  Whoever consumes the source uploads them — the partitioned pipeline in sub-chunks on the copy stream, so that
    // the upload of sub-chunk j+1 overlaps the L1 scatter (P2) of sub-chunk j; everything else in one piece (stage_source).
    const uint8_t *host = nullptr;
    size_t host_bytes = 0;
    // the host did NOT walk this chunk's framing: it only saw that the byte count fits nreads records of the first record's
    // length.  The L1 scatter then checks every length byte for equality, and a mismatch gives the chunk back (PART_NOT_UNIFORM)
    // before anything but scratch has been touched.
    bool verify_uniform = false;
    // the records were PREFETCHED: their upload to `rec` is already queued on the copy stream and this event fires when they
    // have landed — the consumer waits for it instead of uploading (gk_map_count_reads looks one chunk ahead; gk_map_prefetch_reads)
    hipEvent_t ready = nullptr;
};
int stage_source(gk_ctx *ctx, const ReadSrc &src);      // upload a host-fed source in one piece, stream-ordered on ctx->stream
}

namespace gk {
struct PartScratch;
// pooled device memory of a context (blocks of 1 MiB and more; smaller requests go straight to hipMalloc).  pool_free waits
// for the context's two streams first, like the hipFree it replaces.  hipErrorOutOfMemory only after the pool was emptied.
hipError_t pool_malloc(gk_ctx *ctx, void **p, size_t bytes);
hipError_t pool_free(gk_ctx *ctx, void *p);
void pool_release(gk_ctx *ctx);                              // hipFree everything parked
size_t mem_available(gk_ctx *ctx);                           // free + parked, or what the caller's budget leaves
double graph_table_load(gk_ctx *ctx, int k, uint64_t keys);  // load factor of a table the graph phase will read
}

struct gk_map {
    gk_ctx *ctx = nullptr;
    int k = 0;
    int W = 1;                   // 64-bit words per key
    uint64_t capacity = 0;       // slots = nseg * 2^seg_bits
    uint32_t nb2 = 1, lnb1 = 0;  // segment geometry (gk::Table)
    void *slots = nullptr;       // CSlot / Slot<1> / Slot<2> [capacity], by W and `layout`
    int layout = 0;              // gk::LAYOUT_COUNT or gk::LAYOUT_GRAPH (always GRAPH-shaped Slot<2> for W = 2: the field then only says what the table is FOR)
    gk::Counters *d_ctr = nullptr;
    // pinned host landing area of the small device->host reads a batch ends with: [Counters][PartStatus] (a copy into
    // pageable memory is staged and synchronous: five of them cost 0.12 ms of a 2.1 ms step)
    unsigned char *h_status = nullptr;
    uint64_t occ_cached = 0;     // d_ctr->occurrences as of the last map_sync_counters (every launch path ends with one)
    uint64_t size = 0;           // host mirror of d_ctr->size (valid after every public call)
    uint64_t tombstones = 0;
    uint64_t total_occurrences = 0;
    uint64_t grows = 0;
    bool skewed = false;         // a batch overflowed the pipeline's L1 regions and spill list (pathological skew): auto mode stays on the direct path
    bool dirty = false;          // the table may hold keys that are not the hash-rule orientation of their k-mer (verbatim inserts):
                                 // Graph.buildGraph's `contains` then probes both strands, as the reference does (Graph.scala:270)
    bool masks_valid = false;    // every live slot's annotation word holds the (incoming, outcoming) mask its OWNER computed (gk_dist_gather_map,
                                 // classified form): gk_graph_build then skips the neighbour lookups.  Any change of the contents clears it
                                 // (map_sync_counters, gk_map_clear), and so does the build that consumes it.
    bool sample_dirty = false;   // the distinct-key sample holds keys: gk_map_clear must reset it
    void *d_scratch = nullptr;   // pooled device scratch of the point-query / scan entry points (get_batch, filter_lt, export)
    size_t scratch_bytes = 0;
    bool repeats = false;        // over-provisioned segment regions spilled: this data has many repeats, use the exact fine level
    // distinct-key sample (gk_partition.hip: Sampler): 1/1024 of the distinct keys offered since the last clear
    uint64_t *d_sample = nullptr;
    uint64_t sample_mask = 0;
    uint64_t sample_claims_seen = 0;     // sample claims already accounted for by earlier batches
    uint64_t est_distinct_last = 0;      // the last batch's estimate of NEW distinct keys (gk_map_stats)
    uint64_t max_batch_keys = 0;         // 0 = default: windows per partitioned batch when the call's reads exceed the table's room
    float last_count_ms = 0.f;
    uint64_t last_count_occ = 0;
    // two staging areas of host-fed count_reads: one being consumed, one being filled by the copy stream (gk_table.hip)
    struct StageSlot {
        void *d = nullptr;
        size_t cap = 0;
        const uint8_t *host = nullptr;   // what was PREFETCHED into this area and not consumed yet (valid): host address, bytes
        size_t bytes = 0;
        bool valid = false;
        bool armed = false;              // chosen and sized, the copy not issued yet (map_fire_prefetch)
        hipEvent_t ev = nullptr;         // fires when the prefetched bytes have landed
    } stage[2];
    int stage_cur = 0;                   // the area the current (or last) chunk lives in
    int stage_last_pf = 1;               // the area the last prefetch went to
    void *d_offsets = nullptr;
    size_t offsets_bytes = 0;
    // partitioned insert path (gk_partition.hip)
    gk::PartScratch *part = nullptr;
    int insert_path = 0;         // 0 auto, 1 direct (global atomics), 2 partitioned (LDS build)
    bool pending_clear = false;  // gk_map_clear deferred: slots are stale until materialised
    uint64_t part_launches = 0, direct_launches = 0;
    uint64_t spilled_keys = 0, failed_segments = 0, retries_direct = 0;   // partitioned-path skew counters
    float phase_ms[5] = {0, 0, 0, 0, 0};   // last insert: hist1, scatter1, hist2, scatter2, seg_insert (or [0] = direct kernel)
    float gap_ms = 0.f;                    // of phase_ms[2]: GPU idle while the host read the sample / sized the table / prepared the fine level
};

namespace gk {

void set_error(const gk_ctx *ctx, const std::string &msg);
int fail(const gk_ctx *ctx, int code, const std::string &msg);
int hip_fail(const gk_ctx *ctx, hipError_t e, const char *what);

#define GK_HIP(ctx, call)                                                   \
    do {                                                                    \
        hipError_t e__ = (call);                                            \
        if (e__ != hipSuccess) return gk::hip_fail((ctx), e__, #call);      \
    } while (0)

inline bool k_supported(int k) { return (k >= 2 && k <= 31) || (k >= 34 && k <= 64); }
inline int words_for_k(int k) { return k <= 32 ? 1 : 2; }
inline size_t slot_bytes(int W) { return W == 1 ? 16 : 24; }      // sizeof(gk::Slot<W>): value maps, and k-mer tables in the GRAPH layout
// A k-mer table with 8-byte keys comes in two layouts (gk_device.h): COUNT = 12-byte CSlot (key + count: what counting streams),
// GRAPH = 16-byte Slot<1> (+ the graph phase's annotation word: what deleteAll, the gather and gk_map_create_for_graph leave).
// 16-byte keys have one layout (24-byte Slot<2>).
enum { LAYOUT_COUNT = 0, LAYOUT_GRAPH = 1 };
inline size_t slot_bytes(int W, int layout) { return W == 2 ? 24 : (layout == LAYOUT_GRAPH ? 16 : 12); }
inline uint32_t seg_bits_for(int W) { return W == 1 ? gk::SegBits<1>::value : gk::SegBits<2>::value; }
// segment geometry for at least `want_slots` slots: nb1 = 2^lnb1 L1 buckets x nb2 fine buckets.  256 L1 buckets at most
// — unless that would take more fine buckets per L1 bucket than the partitioned insert pipeline handles (PLAN_MAX_NB2):
// then 512 or 1024 (tables beyond 34 GB).  min_lnb1: a test hook that forces a larger fan-out on small tables too;
// keep_lnb1 >= 0: the L1 bucket of a key must not change (growth between the two levels of a running batch).
static constexpr uint32_t PLAN_MAX_NB2 = 4096;
inline void plan_segments(int W, uint64_t want_slots, uint32_t *nb2, uint32_t *lnb1, uint64_t *capacity, uint32_t min_lnb1 = 0,
                          int keep_lnb1 = -1) {
    const uint64_t S = 1ull << seg_bits_for(W);
    uint64_t want_seg = (want_slots + S - 1) / S;
    if (want_seg < 1) want_seg = 1;
    uint32_t l = 0;
    while (l < 8 && (2ull << l) <= want_seg) l++;
    while (l < gk::MAX_LNB1 && (want_seg > ((uint64_t)PLAN_MAX_NB2 << l) || (l < min_lnb1 && (2ull << l) <= want_seg))) l++;
    if (keep_lnb1 >= 0) l = (uint32_t)keep_lnb1;
    const uint64_t nb1 = 1ull << l;
    *lnb1 = l;
    *nb2 = (uint32_t)((want_seg + nb1 - 1) / nb1);
    *capacity = ((uint64_t)*nb2 << l) * S;
}
inline uint64_t pow2ceil(uint64_t v) {
    uint64_t p = 1;
    while (p < v) p <<= 1;
    return p;
}

inline size_t map_slot_bytes(const gk_map *m) { return slot_bytes(m->W, m->layout); }
// run BODY with `W` (constexpr int) and `S` (the slot type) bound to what the map's table is made of
#define GK_BY_SLOT(m_, ...)                                                                              \
    do {                                                                                                 \
        if ((m_)->W == 2) { constexpr int W = 2; using S = gk::Slot<2>; (void)W; __VA_ARGS__; }          \
        else if ((m_)->layout == gk::LAYOUT_GRAPH) { constexpr int W = 1; using S = gk::Slot<1>; (void)W; __VA_ARGS__; } \
        else { constexpr int W = 1; using S = gk::CSlot; (void)W; __VA_ARGS__; }                         \
    } while (0)

// table ops used across translation units
int map_reserve(gk_map *m, uint64_t extra_keys);     // grow so that size+extra stays under the load limit
int map_sync_counters(gk_map *m);                    // refresh m->size, detect device-side error flag
// the classify over a PartitionedDNAMap (gk_graph.hip: k_dc_*; driven by gk_dist.hip)
int dclass_count(gk_map *m, int rank, int P, uint64_t s0, uint64_t s1, unsigned long long *d_cnt);
int dclass_fill(gk_map *m, int rank, int P, uint64_t s0, uint64_t s1, const unsigned long long *d_off, unsigned long long *d_cur, uint64_t *d_qkeys, uint64_t *d_qref);
int dclass_answer(gk_map *m, const uint64_t *d_keys, uint64_t n, uint8_t *d_ans);
int dclass_apply(gk_map *m, const uint64_t *d_qref, const uint8_t *d_ans, uint64_t n);
int map_materialize(gk_map *m);                      // run a deferred clear so that the slots are valid
int map_add_keys_direct(gk_map *m, const uint64_t *d_keys, uint64_t n);   // k_add_keys on device keys (W words each)
int map_insert_keys_dev(gk_map *m, const uint64_t *d_keys, uint64_t n, bool verbatim);
int map_add_unique_keys_dev(gk_map *m, const uint64_t *d_keys, const int32_t *d_counts, const uint8_t *d_masks, uint64_t n);   // absent, distinct keys into a graph-layout table
int map_add_counted_keys_dev(gk_map *m, const uint64_t *d_keys, const int32_t *d_counts, uint64_t n);   // update(key, c, _ + c), canonical device keys   // update(key, 1, _+1) for device keys, either path
int map_export_range_dev(gk_map *m, uint64_t s0, uint64_t s1, uint64_t *d_keys, int32_t *d_cnt, unsigned long long *d_cursor, uint64_t *n_out, uint8_t *d_masks = nullptr);
int map_to_graph_layout(gk_map *m);                  // 8-byte keys in 12-byte count slots -> 16-byte graph slots (streaming rebuild); no-op otherwise
int map_create_for_graph(gk_ctx *ctx, int k, uint64_t keys, gk_map **out);      // a new map sized the way the graph phase wants it
void *map_scratch(gk_map *m, size_t bytes);          // pooled scratch (grown, never shrunk); nullptr + error set on failure
// partitioned path (part_count may return PART_RETRY_DIRECT: take the direct path for this batch)
constexpr int PART_RETRY_DIRECT = 1;
constexpr int PART_OUTGREW = 3;          // the table grew between the levels past what its L1 fan-out lets the pipeline address: this batch goes the direct way
constexpr int PART_NOT_UNIFORM = 2;      // ReadSrc::verify_uniform failed: walk the framing on the host and come again
// estimate: keep the distinct-key sample and wait for it between the two partition levels — the table is then sized for the
// batch's NEW DISTINCT keys (and may be replaced, same lnb1) instead of having been sized for its windows up front;
// fine_exact: take the exact (range-matrix) fine level even where over-provisioned segment regions could be tried
// check_canon: the keys come verbatim from the caller: note in Counters::noncanon if one is not its k-mer's hash-rule orientation
// grow_ahead: when the table has to grow for this batch, the call still holds (grow_ahead - 1) x as many windows after it:
// leave room for part of what they will bring, so that the next batch does not rehash what this one just built
struct PartPlan { bool estimate = false; bool fine_exact = false; bool check_canon = false; double grow_ahead = 1.0; };
int part_count(gk_map *m, PartScratch **pps, const ReadSrc &src, const uint64_t *d_keys, uint64_t nkeys_in, uint64_t nkeys_bound,
               bool from_empty, const PartPlan &plan);
int map_fire_prefetch(gk_map *m, hipEvent_t after);   // issue the armed host -> staging uploads on the copy stream behind `after` (nullptr: at once)
int map_ensure_sample(gk_map *m);                    // allocate the distinct-key sample set on first use
// if the table cannot take `new_distinct` more keys, grow it (rehash, or plain re-allocation from empty) for `size_for` more
int map_make_room(gk_map *m, uint64_t new_distinct, uint64_t size_for, bool from_empty, bool keep_lnb1 = false);
uint64_t part_max_slots(int W);                      // largest table the partitioned path can address
int ctx_check_format(gk_ctx *ctx);                   // GK_E_FORMAT (and reset) if a map-less kernel raised ctx->d_flags[0]
// super-k-mer routing in two halves (gk_skm.hip): launch on any stream without waiting, finish after that stream was synchronised
constexpr int SKM_COUNT_WORDS = 2 * 64 + 1;
int skm_route_launch(gk_ctx *ctx, hipStream_t st, unsigned long long *d_counts, unsigned long long *h_counts, int k, const void *dev_records,
                     uint64_t nreads, int read_len, int P, void *dev_out, uint64_t out_cap_records);
int skm_route_finish(gk_ctx *ctx, const unsigned long long *h_counts, bool launched, int P, uint64_t out_cap_records, uint64_t *rec_counts_host,
                     uint64_t *kmer_counts_host);
// lanes per read in the window loops: 64 for reads, 32/16 for short records (super-k-mers)
inline int lanes_per_read(int max_windows) { return max_windows > 32 ? 64 : max_windows > 16 ? 32 : 16; }
bool part_supported(const gk_map *m);
void part_scratch_free(gk_ctx *ctx, PartScratch *ps);

}  // namespace gk

// Optional in-kernel phase timers (-DGK_TIMERS; scripts/run_timers.py): thread 0 of every workgroup adds
// the wall-clock ticks (s_memrealtime, 100 MHz) it spent in each phase.  A translation unit that uses
// them says GK_TIMERS_DEFINE(suffix) once at file scope: its counters and the C entry point
// gk_debug_timers_<suffix>(out[16], reset) that reads them back.
#ifdef GK_TIMERS
#define GK_TIMERS_DEFINE(suffix)                                                                                          \
    static __device__ unsigned long long g_timers[16];                                                                   \
    extern "C" int gk_debug_timers_##suffix(unsigned long long *out16, int reset) {                                      \
        if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_timers), sizeof(unsigned long long) * 16) != hipSuccess) return -1;  \
        unsigned long long z[16] = {0};                                                                                  \
        if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_timers), z, sizeof(z)) != hipSuccess) return -1;                     \
        return 0;                                                                                                        \
    }
#define GK_T0() unsigned long long t_prev = __builtin_amdgcn_s_memrealtime(), t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define GK_TARGS_DECL unsigned long long &t_prev, unsigned long long (&t_acc)[8]      /* pass the timers into a helper */
#define GK_TARGS t_prev, t_acc
#define GK_TICK(i) do { const unsigned long long t_now = __builtin_amdgcn_s_memrealtime(); t_acc[i] += t_now - t_prev; t_prev = t_now; } while (0)
#define GK_TFLUSH(base) do { if (threadIdx.x == 0) for (int q = 0; q < 8; q++) if (t_acc[q]) atomicAdd(&g_timers[(base) + q], t_acc[q]); } while (0)
#else
#define GK_TIMERS_DEFINE(suffix)
#define GK_T0() do {} while (0)
#define GK_TARGS_DECL int
#define GK_TARGS 0
#define GK_TICK(i) do {} while (0)
#define GK_TFLUSH(base) do {} while (0)
#endif

// Inside the library every device allocation goes through the context's block pool: the HIP names are redirected here, and a
// translation unit that really means the runtime's own writes (hipMalloc)(...) / (hipFree)(...).  Needs `ctx` in scope.
#ifndef GK_NO_POOL_REDIRECT
#define hipMalloc(p, n) gk::pool_malloc(ctx, (void **)(p), (size_t)(n))
#define hipFree(p) gk::pool_free(ctx, (void *)(p))
#endif

