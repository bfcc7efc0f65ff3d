// gk_table.hip — the DNAMap[Int] hot path as HIP kernels for gfx950, plus its C-ABI entry points.
//
// Reference path replaced here (S/ = /root/reference/src/main/scala/ru/ifmo/genome/):
//   FreqFilter.add            S/data/FreqFilter.scala:28-36      -> k_count_reads
//   PairedEndData.getPairs    S/data/PairedEndData.scala:20-36   -> record framing read in-kernel
//   Container.update(k,v0,f)  S/ds/ArrayDNAMap.scala:129-150     -> gk::table_add (gk_device.h)
//   Container.apply           S/ds/ArrayDNAMap.scala:90-101      -> k_get
//   Container.deleteAll       S/ds/ArrayDNAMap.scala:164-173     -> k_filter_lt
//   ArrayDNAMap.rescale       S/ds/ArrayDNAMap.scala:217-230     -> k_rehash (pre-sized / doubled)
//   Container.iterator        S/ds/ArrayDNAMap.scala:175-178     -> k_export
//
// Roofline: every kernel here is HBM-bound integer work (random 64-B sector touches for probes,
// streaming for scans); no MFMA.  Algorithmic bytes per unit are stated in DESIGN.md.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

#include "gk_internal.h"
#include "gk_tile.h"

using namespace gk;

// ============================================================================================
// error plumbing
// ============================================================================================
namespace gk {
static thread_local std::string tls_err;

void set_error(const gk_ctx *ctx, const std::string &msg) {
    tls_err = msg;
    if (ctx) const_cast<gk_ctx *>(ctx)->err = msg;
}
int fail(const gk_ctx *ctx, int code, const std::string &msg) {
    set_error(ctx, msg);
    return code;
}
int hip_fail(const gk_ctx *ctx, hipError_t e, const char *what) {
    (void)hipGetLastError();
    return fail(ctx, e == hipErrorNoDevice || e == hipErrorInvalidDevice ? GK_E_NODEVICE : GK_E_HIP,
                std::string(what) + ": " + hipGetErrorString(e));
}
}  // namespace gk

// ============================================================================================
// kernels
// ============================================================================================
template <int W> __global__ __launch_bounds__(BLOCK) void k_clear(Slot<W> *slots, u64 n) {
    u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x;
    u64 stride = (u64)gridDim.x * BLOCK;
    for (; i < n; i += stride) {
        if constexpr (W == 1) {
            slots[i] = Slot<1>{KEY_EMPTY, 0u, 0u};
        } else {
            slots[i] = Slot<2>{KEY_EMPTY, KEY_EMPTY, 0u, 0u, 0ull};
        }
    }
}

// FreqFilter.add (FreqFilter.scala:28-36) for a stream of `.bin` records: one workgroup stages a
// tile of 64 reads in LDS, each wave takes reads round-robin, each lane one window of the read:
// extract -> reverse complement -> hash rule -> insert-or-increment in the HBM table.
//   offsets == nullptr : fixed stride records (record r at r*stride)
//   offsets != nullptr : offsets[r] = byte offset of record r, offsets[nreads] = end
template <int W>
__global__ __launch_bounds__(BLOCK) void k_count_reads(const uint8_t *__restrict__ rec, u64 nreads,
                                                       const u32 *__restrict__ offsets, u32 stride, int k, int group,
                                                       Table<W> t, Counters *ctr) {
    __shared__ __attribute__((aligned(16))) u32 tile[TILE_WORDS];
    __shared__ u32 s_claimed, s_occ;
    if (threadIdx.x == 0) { s_claimed = 0; s_occ = 0; }
    u32 claimed = 0, occ = 0, err = 0;
    const u64 ntiles = (nreads + TILE_READS - 1) / TILE_READS;
    for (u64 tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const u64 r0 = tl * TILE_READS;
        const int nr = (int)min((u64)TILE_READS, nreads - r0);
        const u64 gb = offsets ? (u64)offsets[r0] : r0 * stride;
        const u64 ge = offsets ? (u64)offsets[r0 + nr] : (r0 + nr) * stride;
        __syncthreads();                       // previous tile fully consumed
        const u64 a0 = stage_tile(tile, rec, gb, ge);
        __syncthreads();
        for_each_window<W>(tile, a0, r0, nr, offsets, stride, k, group, [&](Kmer<W> x) {
            Kmer<W> y = canonical(x, k);                  // FreqFilter.scala:31-32
            claimed += table_add(t, y, 1u, &err);         // kmersFreq.update(y, 1, _ + 1)  :33
            occ++;
        });
    }
    atomicAdd(&s_claimed, claimed);
    atomicAdd(&s_occ, occ);
    if (err) ctr->error = 1;
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_claimed) atomicAdd(&ctr->size, (unsigned long long)s_claimed);
        if (s_occ) atomicAdd(&ctr->occurrences, (unsigned long long)s_occ);
    }
}

// DNAMap.update(key, c, _ + c) for keys already canonical and routed (owner-side insert of the
// PartitionedDNAMap exchange; also partition merge).  keys: W words per key, interleaved.
template <int W>
__global__ __launch_bounds__(BLOCK) void k_add_keys(const u64 *__restrict__ keys, const i32 *__restrict__ counts,
                                                    u64 n, Table<W> t, Counters *ctr) {
    __shared__ u32 s_claimed;
    if (threadIdx.x == 0) s_claimed = 0;
    __syncthreads();
    u32 claimed = 0, err = 0;
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) {
        Kmer<W> key;
        if constexpr (W == 1) key = Kmer<1>{keys[i]};
        else key = Kmer<2>{keys[2 * i], keys[2 * i + 1]};
        claimed += table_add(t, key, counts ? (u32)counts[i] : 1u, &err);
    }
    if (claimed) atomicAdd(&s_claimed, claimed);
    if (err) ctr->error = 1;
    __syncthreads();
    if (threadIdx.x == 0 && s_claimed) atomicAdd(&ctr->size, (unsigned long long)s_claimed);
}

// ArrayDNAMap.rescale (ArrayDNAMap.scala:217-230): move every live (key, count) into a new table.
template <int W>
__global__ __launch_bounds__(BLOCK) void k_rehash(const Slot<W> *__restrict__ old, u64 ncap, Table<W> t, Counters *ctr) {   // old and new share t.tagged
    u32 err = 0;
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < ncap; i += (u64)gridDim.x * BLOCK) {
        if (!slot_live(&old[i])) continue;
        Kmer<W> key = slot_key(old, i, t.tagged);
        // every key of the old table is unique: one CAS claims its new slot, the count goes in with a plain store
        const u64 h = slot_hash(key);
        Slot<W> *seg = t.slots + ((u64)seg_of(t, h) << SegBits<W>::value);
        const i64 at = seg_claim_unique(seg, seg_pos<W>(h), key, t.tagged);
        if (at < 0) { err = 1; continue; }
        seg[at].extra = old[i].extra;
    }
    if (err) ctr->error = 1;
}

// Container.deleteAll((k, v) => v < rounds) (ArrayDNAMap.scala:164-173): full scan, tombstone.
template <int W>
__global__ __launch_bounds__(BLOCK) void k_filter_lt(Slot<W> *slots, u64 ncap, i32 rounds, unsigned long long *removed) {
    __shared__ u32 s_rm;
    if (threadIdx.x == 0) s_rm = 0;
    __syncthreads();
    u32 rm = 0;
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < ncap; i += (u64)gridDim.x * BLOCK) {
        if (slot_live(&slots[i]) && (i32)slot_count(&slots[i]) < rounds) {
            slots[i].w0 = KEY_TOMB;
            rm++;
        }
    }
    if (rm) atomicAdd(&s_rm, rm);
    __syncthreads();
    if (threadIdx.x == 0 && s_rm) atomicAdd(removed, (unsigned long long)s_rm);
}

// Container.apply (ArrayDNAMap.scala:90-101) for a batch of keys.
template <int W>
__global__ __launch_bounds__(BLOCK) void k_get(const u64 *__restrict__ lo, const u64 *__restrict__ hi, u64 n,
                                               Table<W> t, i32 *counts, uint8_t *found) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) {
        Kmer<W> key;
        if constexpr (W == 1) key = Kmer<1>{lo[i]};
        else key = Kmer<2>{lo[i], hi[i]};
        i64 s = table_find(t, key);
        if (counts) counts[i] = s >= 0 ? (i32)slot_count(&t.slots[s]) : -1;
        if (found) found[i] = s >= 0;
    }
}

// block-wide exclusive scan of a 0/1 flag using ballots (wave64) + one LDS word per wave
__device__ __forceinline__ u32 block_scan_flag(bool flag, u32 *total, u32 *lds4) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long b = __ballot(flag);
    u32 wprefix = (u32)__popcll(b & ((1ull << lane) - 1ull));
    __syncthreads();
    if (lane == 0) lds4[wave] = (u32)__popcll(b);
    __syncthreads();
    u32 base = 0, tot = 0;
    for (int w = 0; w < BLOCK / 64; ++w) {
        u32 c = lds4[w];
        if (w < wave) base += c;
        tot += c;
    }
    *total = tot;
    return base + wprefix;
}

// Container.iterator (ArrayDNAMap.scala:175-178): compact every live (key, count) to dense arrays.
// One atomic per 256-slot group reserves the output range; order within a group is slot order.
template <int W>
__global__ __launch_bounds__(BLOCK) void k_export(const Slot<W> *__restrict__ slots, u64 ncap, u32 tagged, u64 *lo, u64 *hi, i32 *cnt,
                                                  unsigned long long *cursor) {
    __shared__ u32 lds4[BLOCK / 64];
    __shared__ unsigned long long s_base;
    const u64 ngroups = (ncap + BLOCK - 1) / BLOCK;
    for (u64 g = blockIdx.x; g < ngroups; g += gridDim.x) {
        u64 i = g * BLOCK + threadIdx.x;
        bool live = i < ncap && slot_live(&slots[i]);
        u32 tot;
        u32 pos = block_scan_flag(live, &tot, lds4);
        if (threadIdx.x == 0 && tot) s_base = atomicAdd(cursor, (unsigned long long)tot);
        __syncthreads();
        if (live) {
            Kmer<W> key = slot_key(slots, i, tagged);
            u64 o = s_base + pos;
            lo[o] = key.lo;
            if constexpr (W == 2) { if (hi) hi[o] = key.hi; }
            else { if (hi) hi[o] = 0; }
            cnt[o] = (i32)slot_count(&slots[i]);
        }
        __syncthreads();
    }
}

// ============================================================================================
// host side
// ============================================================================================
static inline int grid_for(const gk_ctx *ctx, u64 work_items, int per_block) {
    u64 blocks = (work_items + per_block - 1) / per_block;
    u64 cap = (u64)ctx->cu_count * 8;      // 8 resident 256-thread workgroups per CU
    if (blocks < 1) blocks = 1;
    return (int)std::min(blocks, cap);
}

template <int W> static Table<W> table_of(const gk_map *m) {
    return Table<W>{reinterpret_cast<Slot<W> *>(m->slots), m->nb2, m->lnb1, m->k == 64 ? 1u : 0u};
}

static int alloc_table(gk_ctx *ctx, int W, uint64_t cap, void **out) {
    GK_HIP(ctx, hipMalloc(out, cap * slot_bytes(W)));
    int grid = grid_for(ctx, cap, BLOCK * 4);
    if (W == 1) hipLaunchKernelGGL(k_clear<1>, dim3(grid), dim3(BLOCK), 0, ctx->stream, (Slot<1> *)*out, cap);
    else hipLaunchKernelGGL(k_clear<2>, dim3(grid), dim3(BLOCK), 0, ctx->stream, (Slot<2> *)*out, cap);
    GK_HIP(ctx, hipGetLastError());
    return GK_OK;
}

namespace gk {

int map_sync_counters(gk_map *m) {
    Counters c;
    GK_HIP(m->ctx, hipMemcpyAsync(&c, m->d_ctr, sizeof(c), hipMemcpyDeviceToHost, m->ctx->stream));
    GK_HIP(m->ctx, hipStreamSynchronize(m->ctx->stream));
    m->size = c.size;
    if (c.error) {
        GK_HIP(m->ctx, hipMemsetAsync(&m->d_ctr->error, 0, sizeof(u32), m->ctx->stream));
        return fail(m->ctx, GK_E_CAPACITY, "a table segment filled up (internal sizing error)");
    }
    return GK_OK;
}

// Load limits: grow before a batch could exceed max_load; size for target_load.  A tagged table
// (k = 64) is four interleaved sub-tables of a quarter segment each, so it runs emptier.
static inline double max_load(const gk_map *m) { return m->k == 64 ? 0.6 : 0.8; }
static inline double target_load(const gk_map *m) { return m->k == 64 ? 0.45 : 0.65; }

// ArrayDNAMap.rescale analogue: make room for `extra_keys` more distinct keys.
int map_reserve(gk_map *m, uint64_t extra_keys) {
    uint64_t need = m->size + m->tombstones + extra_keys;
    if ((double)need <= max_load(m) * (double)m->capacity) return GK_OK;
    uint32_t nnb2, nlnb1;
    uint64_t ncap;
    plan_segments(m->W, std::max<uint64_t>((uint64_t)((double)(m->size + extra_keys) / target_load(m)) + 1, m->capacity + m->capacity / 2), &nnb2, &nlnb1, &ncap);
    gk_ctx *ctx = m->ctx;
    void *nslots = nullptr;
    int rc = alloc_table(ctx, m->W, ncap, &nslots);
    if (rc) return fail(ctx, GK_E_CAPACITY, "cannot grow table to " + std::to_string(ncap) + " slots: " + ctx->err);
    int grid = grid_for(ctx, m->capacity, BLOCK);
    if (m->W == 1)
        hipLaunchKernelGGL(k_rehash<1>, dim3(grid), dim3(BLOCK), 0, ctx->stream, (const Slot<1> *)m->slots, m->capacity,
                           Table<1>{(Slot<1> *)nslots, nnb2, nlnb1, 0u}, m->d_ctr);
    else
        hipLaunchKernelGGL(k_rehash<2>, dim3(grid), dim3(BLOCK), 0, ctx->stream, (const Slot<2> *)m->slots, m->capacity,
                           Table<2>{(Slot<2> *)nslots, nnb2, nlnb1, m->k == 64 ? 1u : 0u}, m->d_ctr);
    GK_HIP(ctx, hipGetLastError());
    GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    GK_HIP(ctx, hipFree(m->slots));
    m->slots = nslots;
    m->capacity = ncap;
    m->nb2 = nnb2;
    m->lnb1 = nlnb1;
    m->tombstones = 0;
    m->grows++;
    return map_sync_counters(m);
}

int map_materialize(gk_map *m) {
    if (!m->pending_clear) return GK_OK;
    gk_ctx *ctx = m->ctx;
    int grid = grid_for(ctx, m->capacity, BLOCK * 4);
    if (m->W == 1) hipLaunchKernelGGL(k_clear<1>, dim3(grid), dim3(BLOCK), 0, ctx->stream, (Slot<1> *)m->slots, m->capacity);
    else hipLaunchKernelGGL(k_clear<2>, dim3(grid), dim3(BLOCK), 0, ctx->stream, (Slot<2> *)m->slots, m->capacity);
    GK_HIP(ctx, hipGetLastError());
    m->pending_clear = false;
    return GK_OK;
}

}  // namespace gk

static int check_map_lazy(const gk_map *m) {       // does not touch the slots
    if (!m || !m->ctx) return fail(nullptr, GK_E_INVALID, "null map handle");
    hipError_t e = hipSetDevice(m->ctx->device);
    if (e != hipSuccess) return hip_fail(m->ctx, e, "hipSetDevice");
    return GK_OK;
}
static int check_map(gk_map *m) {                  // slots must be valid afterwards
    if (int rc = check_map_lazy(m)) return rc;
    return map_materialize(m);
}

extern "C" {

int gk_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int gk_ctx_create(int device, gk_ctx **out) {
    if (!out) return fail(nullptr, GK_E_INVALID, "gk_ctx_create: out is NULL");
    *out = nullptr;
    int n = gk_device_count();
    if (n <= 0) return fail(nullptr, GK_E_NODEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= n) return fail(nullptr, GK_E_INVALID, "device index out of range");
    gk_ctx *ctx = new gk_ctx();
    ctx->device = device;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev0);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev1);
    for (int i = 0; i < 6 && e == hipSuccess; i++) e = hipEventCreate(&ctx->pev[i]);
    hipDeviceProp_t prop;
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) {
        int rc = hip_fail(nullptr, e, "gk_ctx_create");
        delete ctx;
        return rc;
    }
    ctx->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    *out = ctx;
    return GK_OK;
}

void gk_ctx_destroy(gk_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) { (void)hipStreamSynchronize(ctx->stream); (void)hipStreamDestroy(ctx->stream); }
    if (ctx->skm_counts) (void)hipFree(ctx->skm_counts);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    for (int i = 0; i < 6; i++) if (ctx->pev[i]) (void)hipEventDestroy(ctx->pev[i]);
    delete ctx;
}

const char *gk_last_error(const gk_ctx *ctx) { return ctx ? ctx->err.c_str() : tls_err.c_str(); }
int gk_ctx_device(const gk_ctx *ctx) { return ctx ? ctx->device : -1; }

int gk_ctx_sync(gk_ctx *ctx) {
    if (!ctx) return fail(nullptr, GK_E_INVALID, "null ctx");
    GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GK_OK;
}
int gk_dev_alloc(gk_ctx *ctx, size_t nbytes, void **dev_ptr) {
    if (!ctx || !dev_ptr) return fail(ctx, GK_E_INVALID, "gk_dev_alloc: null argument");
    GK_HIP(ctx, hipSetDevice(ctx->device));
    GK_HIP(ctx, hipMalloc(dev_ptr, nbytes ? nbytes : 1));
    return GK_OK;
}
int gk_dev_free(gk_ctx *ctx, void *dev_ptr) {
    if (!ctx) return fail(ctx, GK_E_INVALID, "null ctx");
    if (dev_ptr) GK_HIP(ctx, hipFree(dev_ptr));
    return GK_OK;
}
int gk_dev_upload(gk_ctx *ctx, void *dev_dst, const void *host_src, size_t nbytes) {
    if (!ctx) return fail(ctx, GK_E_INVALID, "null ctx");
    GK_HIP(ctx, hipMemcpyAsync(dev_dst, host_src, nbytes, hipMemcpyHostToDevice, ctx->stream));
    GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GK_OK;
}
int gk_dev_download(gk_ctx *ctx, void *host_dst, const void *dev_src, size_t nbytes) {
    if (!ctx) return fail(ctx, GK_E_INVALID, "null ctx");
    GK_HIP(ctx, hipMemcpyAsync(host_dst, dev_src, nbytes, hipMemcpyDeviceToHost, ctx->stream));
    GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GK_OK;
}

int gk_map_create(gk_ctx *ctx, int k, uint64_t capacity_hint, gk_map **out) {
    if (!ctx || !out) return fail(ctx, GK_E_INVALID, "gk_map_create: null argument");
    *out = nullptr;
    if (!k_supported(k))
        return fail(ctx, GK_E_UNSUPPORTED_K,
                    "k=" + std::to_string(k) + " unsupported (2..31 and 34..64; k=32,33 are broken in the reference)");
    GK_HIP(ctx, hipSetDevice(ctx->device));
    gk_map *m = new gk_map();
    m->ctx = ctx;
    m->k = k;
    m->W = words_for_k(k);
    uint64_t want = capacity_hint ? capacity_hint : 1024;
    plan_segments(m->W, (uint64_t)((double)want / target_load(m)) + 1, &m->nb2, &m->lnb1, &m->capacity);
    int rc = alloc_table(ctx, m->W, m->capacity, &m->slots);
    if (rc == GK_OK) {
        hipError_t e = hipMalloc((void **)&m->d_ctr, sizeof(Counters));
        if (e == hipSuccess) e = hipMemsetAsync(m->d_ctr, 0, sizeof(Counters), ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = hip_fail(ctx, e, "gk_map_create");
    }
    if (rc != GK_OK) {
        if (m->slots) (void)hipFree(m->slots);
        if (m->d_ctr) (void)hipFree(m->d_ctr);
        delete m;
        return rc == GK_E_HIP ? fail(ctx, GK_E_CAPACITY, "cannot allocate table: " + ctx->err) : rc;
    }
    *out = m;
    return GK_OK;
}

void gk_map_destroy(gk_map *m) {
    if (!m) return;
    (void)hipSetDevice(m->ctx->device);
    (void)hipStreamSynchronize(m->ctx->stream);
    if (m->slots) (void)hipFree(m->slots);
    if (m->d_ctr) (void)hipFree(m->d_ctr);
    if (m->d_stage) (void)hipFree(m->d_stage);
    if (m->d_offsets) (void)hipFree(m->d_offsets);
    part_scratch_free(m->part);
    delete m;
}

int gk_map_k(const gk_map *m) { return m ? m->k : 0; }

int gk_map_set_insert_path(gk_map *m, int path) {
    if (int rc = check_map_lazy(m)) return rc;
    if (path < 0 || path > 2) return fail(m->ctx, GK_E_INVALID, "insert path must be 0 (auto), 1 (direct) or 2 (partitioned)");
    if (path == 2 && !part_supported(m)) return fail(m->ctx, GK_E_INVALID, "table too large for the partitioned path");
    m->insert_path = path;
    return GK_OK;
}

int gk_map_clear(gk_map *m) {
    if (int rc = check_map_lazy(m)) return rc;
    gk_ctx *ctx = m->ctx;
    // Deferred: the next partitioned insert rebuilds every segment from EMPTY without reading it;
    // anything else materialises the clear first (map_materialize).
    GK_HIP(ctx, hipMemsetAsync(m->d_ctr, 0, sizeof(Counters), ctx->stream));     // stream-ordered: no host round trip needed
    m->pending_clear = true;
    m->size = 0;
    m->tombstones = 0;
    m->total_occurrences = 0;
    return GK_OK;
}

int gk_map_size(gk_map *m, uint64_t *n) {
    if (int rc = check_map_lazy(m)) return rc;
    if (!n) return fail(m->ctx, GK_E_INVALID, "gk_map_size: n is NULL");
    *n = m->size;
    return GK_OK;
}

int gk_map_slots(gk_map *m, uint64_t *slots) {
    if (int rc = check_map_lazy(m)) return rc;
    if (!slots) return fail(m->ctx, GK_E_INVALID, "gk_map_slots: slots is NULL");
    *slots = m->capacity;
    return GK_OK;
}

// launch k_count_reads over device-resident records; accumulates event time into last_count_ms
static int launch_count(gk_map *m, const uint8_t *d_rec, u64 nreads, const u32 *d_off, u32 stride, int group = 64) {
    gk_ctx *ctx = m->ctx;
    u64 ntiles = (nreads + TILE_READS - 1) / TILE_READS;
    int grid = (int)std::min<u64>(std::max<u64>(ntiles, 1), (u64)ctx->cu_count * 8);
    GK_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    if (m->W == 1)
        hipLaunchKernelGGL(k_count_reads<1>, dim3(grid), dim3(BLOCK), 0, ctx->stream, d_rec, nreads, d_off, stride, m->k, group,
                           table_of<1>(m), m->d_ctr);
    else
        hipLaunchKernelGGL(k_count_reads<2>, dim3(grid), dim3(BLOCK), 0, ctx->stream, d_rec, nreads, d_off, stride, m->k, group,
                           table_of<2>(m), m->d_ctr);
    GK_HIP(ctx, hipGetLastError());
    GK_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    if (int rc = map_sync_counters(m)) return rc;
    float ms = 0.f;
    GK_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    m->last_count_ms += ms;
    m->phase_ms[0] += ms;
    m->direct_launches++;
    return GK_OK;
}

static int reset_occ_counter(gk_map *m) {
    GK_HIP(m->ctx, hipMemsetAsync(&m->d_ctr->occurrences, 0, sizeof(unsigned long long), m->ctx->stream));
    return GK_OK;
}
static int read_occ_counter(gk_map *m, uint64_t *occ) {
    unsigned long long v = 0;
    GK_HIP(m->ctx, hipMemcpyAsync(&v, &m->d_ctr->occurrences, sizeof(v), hipMemcpyDeviceToHost, m->ctx->stream));
    GK_HIP(m->ctx, hipStreamSynchronize(m->ctx->stream));
    *occ = v;
    return GK_OK;
}

// Direct path: ~130 B of HBM traffic per occurrence (one 64-B sector read + one 64-B atomic,
// profiles/r01/pmc_count_reads_v2.json).  Partitioned path: ~40 B per occurrence per key word of
// streaming plus the table itself streamed out (and in, unless it is known to be empty).
// The pipeline is calibrated on batches of up to ~2.7e8 windows into a table of about their own size.  Much larger
// batches go with larger tables, where P4 has more fine buckets per chunk (1.2x the time per key at 4.8e8 windows, 1.8x
// at 9.6e8) and where the windows are usually repeats of far fewer k-mers: the over-provisioned regions are sized for
// near-distinct keys, repeats over-disperse the bucket sizes and spill (C3 fed in 1.9e9-window batches through the
// pipeline: 0.62 s against 0.24 s on the direct path, whose repeats hit cached slots).  The growth factor below is
// that experience, not a model of it; splitting a big batch instead would pay a pass over the table per piece
// (measured: 19 vs 11.5 ms for 4.8e8 windows into an empty table).
static bool use_partitioned(const gk_map *m, u64 occ) {
    if (m->insert_path == 1 || !part_supported(m)) return false;
    if (m->insert_path == 2) return true;
    if (m->skewed) return false;          // this map's data has already defeated the over-provisioned regions once
    const double calibrated = 2.7e8;
    const double growth = occ > calibrated ? 1.0 + 0.77 * ((double)occ / calibrated - 1.0) : 1.0;
    const double tb = (double)m->capacity * (double)slot_bytes(m->W);
    const double cost_direct = (double)occ * 130.0 + (m->pending_clear ? tb : 0.0);
    const double cost_part = (double)occ * 40.0 * m->W * growth + tb * (m->pending_clear ? 1.0 : 2.0) + 3e8;
    return cost_part < cost_direct;
}

// partitioned launch over device records or device keys, timed with the same events as launch_count
static int launch_partitioned(gk_map *m, const uint8_t *d_rec, u64 nreads, const u32 *d_off, u32 stride, const u64 *d_keys,
                              u64 nkeys_in, u64 bound, int group = 64) {
    gk_ctx *ctx = m->ctx;
    GK_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    const bool from_empty = m->pending_clear;
    m->pending_clear = false;
    const int prc = part_count(m, &m->part, d_rec, nreads, d_off, stride, group, d_keys, nkeys_in, bound, from_empty);
    if (prc < 0) return prc;
    if (prc == PART_RETRY_DIRECT) {        // extreme skew: nothing but scratch (or a table that was being rebuilt from empty) was touched
        m->pending_clear = from_empty;
        m->skewed = true;
        if (from_empty) m->size = 0;
        if (int rc = map_materialize(m)) return rc;
        if (int rc = map_reserve(m, bound)) return rc;
        if (d_rec) return launch_count(m, d_rec, nreads, d_off, stride, group);
        return map_add_keys_direct(m, d_keys, nkeys_in);
    }
    GK_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    GK_HIP(ctx, hipEventSynchronize(ctx->ev1));
    float ms = 0.f;
    GK_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    m->last_count_ms += ms;
    for (int i = 0; i < 5; i++) {
        float pm = 0.f;
        GK_HIP(ctx, hipEventElapsedTime(&pm, ctx->pev[i], ctx->pev[i + 1]));
        m->phase_ms[i] += pm;
    }
    m->part_launches++;
    return GK_OK;
}

// how many reads (of nk windows each) may go into one launch without risking the load limit
static u64 reads_per_launch(gk_map *m, u64 nk) {
    if (nk == 0) return ~0ull;
    double room = max_load(m) * (double)m->capacity - (double)(m->size + m->tombstones);
    u64 floor_occ = std::max<u64>(1ull << 24, m->capacity / 4);
    u64 occ = room > (double)floor_occ ? (u64)room : floor_occ;
    return std::max<u64>(1, occ / nk);
}

int gk_map_count_reads_dev(gk_map *m, const void *dev_records, uint64_t nreads, int read_len, uint64_t *occurrences) {
    if (int rc = check_map_lazy(m)) return rc;
    gk_ctx *ctx = m->ctx;
    if (occurrences) *occurrences = 0;
    if (!dev_records && nreads) return fail(ctx, GK_E_INVALID, "gk_map_count_reads_dev: null records");
    if (read_len < 0 || read_len > 255) return fail(ctx, GK_E_FORMAT, "read_len must be 0..255 (one length byte per record)");
    m->last_count_ms = 0.f;
    m->last_count_occ = 0;
    for (float &x : m->phase_ms) x = 0.f;
    if (nreads == 0) return GK_OK;
    const u32 stride = 1 + (read_len + 3) / 4;
    const u64 nk = read_len >= m->k ? (u64)(read_len - m->k + 1) : 0;
    if (int rc = reset_occ_counter(m)) return rc;
    const uint8_t *rec = (const uint8_t *)dev_records;
    u64 done = 0;
    while (done < nreads) {
        u64 chunk = std::min(nreads - done, reads_per_launch(m, nk));
        if (nk && use_partitioned(m, chunk * nk)) {
            // (GK_TEST_NO_RESERVE: test hook — leave the table too small on purpose so that segments fill up in P5
            //  and the failed-segment replay runs; hashed keys never get there on their own)
            const bool no_reserve = getenv("GK_TEST_NO_RESERVE") != nullptr;
            if (!m->pending_clear && !no_reserve) { if (int rc = map_reserve(m, chunk * nk)) return rc; }
            if (int rc = launch_partitioned(m, rec + done * stride, chunk, nullptr, stride, nullptr, 0, chunk * nk)) return rc;
        } else {
            if (int rc = map_materialize(m)) return rc;
            if (int rc = map_reserve(m, chunk * nk)) return rc;
            if (int rc = launch_count(m, rec + done * stride, chunk, nullptr, stride)) return rc;
        }
        done += chunk;
    }
    uint64_t occ = 0;
    if (int rc = read_occ_counter(m, &occ)) return rc;
    m->last_count_occ = occ;
    m->total_occurrences += occ;
    if (occurrences) *occurrences = occ;
    return GK_OK;
}

// Owner side of the super-k-mer exchange (gk_skm.hip): the records are short reads in fixed slots;
// kmers_total (exchanged next to the record counts) is the exact number of windows they hold.
int gk_map_count_superkmers_dev(gk_map *m, const void *dev_records, uint64_t nrecords, uint64_t kmers_total, uint64_t *occurrences) {
    if (int rc = check_map_lazy(m)) return rc;
    gk_ctx *ctx = m->ctx;
    if (occurrences) *occurrences = 0;
    if (!dev_records && nrecords) return fail(ctx, GK_E_INVALID, "gk_map_count_superkmers_dev: null records");
    m->last_count_ms = 0.f;
    m->last_count_occ = 0;
    for (float &x : m->phase_ms) x = 0.f;
    if (nrecords == 0) return GK_OK;
    const u32 stride = m->k <= 31 ? 16u : 32u;
    const u64 max_run = (u64)((stride - 1) * 4 - m->k + 1);
    if (kmers_total > nrecords * max_run) return fail(ctx, GK_E_INVALID, "kmers_total exceeds what the records can hold");
    if (int rc = reset_occ_counter(m)) return rc;
    const uint8_t *rec = (const uint8_t *)dev_records;
    // a record holds <= max_run windows but typically a minimizer's reach, ~(k-m+2)/2: pick the lane
    // group that keeps the wave full
    const int group = lanes_per_read(std::min<int>((int)max_run, std::max(8, (m->k - 9) * 3 / 4)));
    if (use_partitioned(m, kmers_total)) {
        if (!m->pending_clear) { if (int rc = map_reserve(m, kmers_total)) return rc; }
        if (int rc = launch_partitioned(m, rec, nrecords, nullptr, stride, nullptr, 0, kmers_total, group)) return rc;
    } else {
        if (int rc = map_materialize(m)) return rc;
        if (int rc = map_reserve(m, kmers_total)) return rc;
        if (int rc = launch_count(m, rec, nrecords, nullptr, stride, group)) return rc;
    }
    uint64_t occ = 0;
    if (int rc = read_occ_counter(m, &occ)) return rc;
    m->last_count_occ = occ;
    m->total_occurrences += occ;
    if (occurrences) *occurrences = occ;
    if (occ != kmers_total) return fail(ctx, GK_E_FORMAT, "records held " + std::to_string(occ) + " k-mers, caller announced " + std::to_string(kmers_total));
    return GK_OK;
}

int gk_map_count_reads(gk_map *m, const uint8_t *bin, size_t nbytes, uint64_t nreads, uint64_t *occurrences) {
    if (int rc = check_map_lazy(m)) return rc;
    gk_ctx *ctx = m->ctx;
    if (occurrences) *occurrences = 0;
    if (!bin && nreads) return fail(ctx, GK_E_INVALID, "gk_map_count_reads: null stream");
    m->last_count_ms = 0.f;
    m->last_count_occ = 0;
    for (float &x : m->phase_ms) x = 0.f;
    if (nreads == 0) return GK_OK;
    if (int rc = reset_occ_counter(m)) return rc;
    // Walk the record framing once on the host (one length byte per record, PairedEndData.scala:24-31),
    // cutting the stream into launches bounded in bytes (staging buffer) and in windows (load limit).
    const size_t MAX_STAGE = 256u << 20;
    size_t pos = 0;
    u64 r = 0;
    std::vector<u32> offs;
    while (r < nreads) {
        const size_t chunk_begin = pos;
        const u64 r_begin = r;
        u64 occ = 0;
        const u64 occ_limit = reads_per_launch(m, 1);
        offs.clear();
        int first_len = -1;
        bool uniform = true;            // every record of the chunk has the same length: fixed stride, no offset table
        // Fast prefix: a run of equal-length records (one sequencing run) is recognised by comparing one byte per
        // record, with no offset table built; it becomes a chunk of its own when it is long enough to be worth it.
        bool fast_prefix = false;
        if (pos < nbytes) {
            const int len0 = bin[pos];
            const size_t rb0 = 1 + (size_t)(len0 + 3) / 4;
            const u64 nk0 = len0 >= m->k ? (u64)(len0 - m->k + 1) : 0;
            u64 cap = std::min<u64>(nreads - r, (nbytes - pos) / rb0);
            cap = std::min<u64>(cap, std::max<u64>(1, MAX_STAGE / rb0));
            if (nk0) cap = std::min<u64>(cap, std::max<u64>(1, occ_limit / nk0));
            const uint8_t *p0 = bin + pos;
            u64 run = 0;
            while (run < cap && p0[run * rb0] == (uint8_t)len0) run++;
            if (run >= 4096 || (run == nreads - r && run > 0)) {
                fast_prefix = true;
                first_len = len0;
                pos += run * rb0;
                occ = run * nk0;
                r += run;
            }
        }
        while (!fast_prefix && r < nreads) {
            if (pos >= nbytes) return fail(ctx, GK_E_FORMAT, "truncated .bin stream: record " + std::to_string(r) + " starts past the end");
            int len = bin[pos];
            if (first_len < 0) first_len = len; else if (len != first_len) uniform = false;
            size_t rb = 1 + (size_t)(len + 3) / 4;
            if (pos + rb > nbytes) return fail(ctx, GK_E_FORMAT, "truncated .bin stream inside record " + std::to_string(r));
            u64 nk = len >= m->k ? (u64)(len - m->k + 1) : 0;
            if (r > r_begin && (pos + rb - chunk_begin > MAX_STAGE || occ + nk > occ_limit)) break;
            offs.push_back((u32)(pos - chunk_begin));
            pos += rb;
            occ += nk;
            r++;
        }
        if (!offs.empty()) offs.push_back((u32)(pos - chunk_begin));
        const size_t cbytes = pos - chunk_begin;
        const u64 creads = r - r_begin;
        const bool partitioned = occ && use_partitioned(m, occ);
        if (!partitioned) { if (int rc = map_materialize(m)) return rc; }
        if (!(partitioned && m->pending_clear)) { if (int rc = map_reserve(m, occ)) return rc; }
        if (m->stage_bytes < cbytes + 64) {
            if (m->d_stage) GK_HIP(ctx, hipFree(m->d_stage));
            m->d_stage = nullptr;
            m->stage_bytes = 0;
            GK_HIP(ctx, hipMalloc(&m->d_stage, cbytes + 64));
            m->stage_bytes = cbytes + 64;
        }
        if (m->offsets_bytes < offs.size() * sizeof(u32)) {
            if (m->d_offsets) GK_HIP(ctx, hipFree(m->d_offsets));
            m->d_offsets = nullptr;
            m->offsets_bytes = 0;
            GK_HIP(ctx, hipMalloc(&m->d_offsets, offs.size() * sizeof(u32)));
            m->offsets_bytes = offs.size() * sizeof(u32);
        }
        GK_HIP(ctx, hipMemcpyAsync(m->d_stage, bin + chunk_begin, cbytes, hipMemcpyHostToDevice, ctx->stream));
        // a chunk of equal-length records (the usual case: one sequencing run) is a fixed-stride array: no offset
        // table to upload, and the partitioned path can take its one-extraction form
        const u32 *d_off = nullptr;
        u32 stride = 0;
        if (uniform && first_len >= 0 && !getenv("GK_HOST_RAGGED")) stride = 1 + (u32)(first_len + 3) / 4;   // (env: A/B hook)
        else {
            GK_HIP(ctx, hipMemcpyAsync(m->d_offsets, offs.data(), offs.size() * sizeof(u32), hipMemcpyHostToDevice, ctx->stream));
            d_off = (const u32 *)m->d_offsets;
        }
        if (partitioned) {
            if (int rc = launch_partitioned(m, (const uint8_t *)m->d_stage, creads, d_off, stride, nullptr, 0, occ)) return rc;
        } else {
            if (int rc = launch_count(m, (const uint8_t *)m->d_stage, creads, d_off, stride)) return rc;
        }
    }
    uint64_t occ = 0;
    if (int rc = read_occ_counter(m, &occ)) return rc;
    m->last_count_occ = occ;
    m->total_occurrences += occ;
    if (occurrences) *occurrences = occ;
    return GK_OK;
}

static int add_keys_dev(gk_map *m, const u64 *d_keys, const i32 *d_counts, u64 n) {
    gk_ctx *ctx = m->ctx;
    u64 done = 0;
    while (done < n) {
        u64 chunk = std::min(n - done, reads_per_launch(m, 1));
        if (int rc = map_reserve(m, chunk)) return rc;
        int grid = grid_for(ctx, chunk, BLOCK);
        if (m->W == 1)
            hipLaunchKernelGGL(k_add_keys<1>, dim3(grid), dim3(BLOCK), 0, ctx->stream, d_keys + done,
                               d_counts ? d_counts + done : nullptr, chunk, table_of<1>(m), m->d_ctr);
        else
            hipLaunchKernelGGL(k_add_keys<2>, dim3(grid), dim3(BLOCK), 0, ctx->stream, d_keys + 2 * done,
                               d_counts ? d_counts + done : nullptr, chunk, table_of<2>(m), m->d_ctr);
        GK_HIP(ctx, hipGetLastError());
        if (int rc = map_sync_counters(m)) return rc;
        done += chunk;
    }
    return GK_OK;
}

}  // extern "C"
namespace gk {
int map_add_keys_direct(gk_map *m, const uint64_t *d_keys, uint64_t n) { return n ? add_keys_dev(m, d_keys, nullptr, n) : GK_OK; }
}
extern "C" {

int gk_map_update_inc_dev(gk_map *m, const void *dev_keys, uint64_t n) {
    if (int rc = check_map_lazy(m)) return rc;
    if (!dev_keys && n) return fail(m->ctx, GK_E_INVALID, "gk_map_update_inc_dev: null keys");
    if (n == 0) return GK_OK;
    m->last_count_ms = 0.f;
    m->last_count_occ = n;
    for (float &x : m->phase_ms) x = 0.f;
    const u64 *keys = (const u64 *)dev_keys;
    u64 done = 0;
    while (done < n) {
        const u64 chunk = std::min(n - done, reads_per_launch(m, 1));
        if (use_partitioned(m, chunk)) {
            if (!m->pending_clear) { if (int rc = map_reserve(m, chunk)) return rc; }
            if (int rc = launch_partitioned(m, nullptr, 0, nullptr, 0, keys + done * m->W, chunk, chunk)) return rc;
        } else {
            if (int rc = map_materialize(m)) return rc;
            gk_ctx *ctx = m->ctx;
            GK_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
            if (int rc = add_keys_dev(m, keys + done * m->W, nullptr, chunk)) return rc;
            GK_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
            GK_HIP(ctx, hipEventSynchronize(ctx->ev1));
            float ms = 0.f;
            GK_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
            m->last_count_ms += ms;
            m->direct_launches++;
        }
        done += chunk;
    }
    return GK_OK;
}

// validate that every key fits in 2k bits — the C-ABI form of `assert(key.length == k)`
// (ArrayDNAMap.scala:182,199): a key with bits above 2k cannot be a k-mer of this map.
static int check_key_bits(const gk_map *m, const uint64_t *lo, const uint64_t *hi, uint64_t n) {
    const int k = m->k;
    for (uint64_t i = 0; i < n; i++) {
        bool bad;
        if (m->W == 1) bad = (lo[i] >> (2 * k)) != 0 || (hi && hi[i] != 0);
        else bad = k < 64 && (hi[i] >> (2 * (k - 32))) != 0;
        if (bad) return fail(m->ctx, GK_E_KLEN, "key " + std::to_string(i) + " is not a " + std::to_string(k) + "-mer (bits set above 2k)");
    }
    return GK_OK;
}

static int upload_keys(gk_map *m, const uint64_t *lo, const uint64_t *hi, uint64_t n, u64 **d_keys) {
    gk_ctx *ctx = m->ctx;
    std::vector<u64> inter((size_t)n * m->W);
    for (uint64_t i = 0; i < n; i++) {
        if (m->W == 1) inter[i] = lo[i];
        else { inter[2 * i] = lo[i]; inter[2 * i + 1] = hi[i]; }
    }
    GK_HIP(ctx, hipMalloc((void **)d_keys, inter.size() * sizeof(u64) + 8));
    GK_HIP(ctx, hipMemcpyAsync(*d_keys, inter.data(), inter.size() * sizeof(u64), hipMemcpyHostToDevice, ctx->stream));
    GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GK_OK;
}

int gk_map_add_counts(gk_map *m, const uint64_t *lo, const uint64_t *hi, const int32_t *counts, uint64_t n) {
    if (int rc = check_map(m)) return rc;
    gk_ctx *ctx = m->ctx;
    if (n == 0) return GK_OK;
    if (!lo || (m->W == 2 && !hi)) return fail(ctx, GK_E_INVALID, "null key array");
    if (int rc = check_key_bits(m, lo, hi, n)) return rc;
    u64 *d_keys = nullptr;
    i32 *d_counts = nullptr;
    int rc = upload_keys(m, lo, hi, n, &d_keys);
    if (rc == GK_OK && counts) {
        hipError_t e = hipMalloc((void **)&d_counts, n * sizeof(i32));
        if (e == hipSuccess) e = hipMemcpyAsync(d_counts, counts, n * sizeof(i32), hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) rc = hip_fail(ctx, e, "gk_map_add_counts");
    }
    if (rc == GK_OK) rc = add_keys_dev(m, d_keys, d_counts, n);
    if (d_keys) (void)hipFree(d_keys);
    if (d_counts) (void)hipFree(d_counts);
    return rc;
}

int gk_map_update_inc(gk_map *m, const uint64_t *lo, const uint64_t *hi, uint64_t n) {
    return gk_map_add_counts(m, lo, hi, nullptr, n);
}

int gk_map_filter_lt(gk_map *m, int32_t rounds) {
    if (int rc = check_map(m)) return rc;
    gk_ctx *ctx = m->ctx;
    unsigned long long *d_removed = nullptr;
    GK_HIP(ctx, hipMalloc((void **)&d_removed, sizeof(unsigned long long)));
    GK_HIP(ctx, hipMemsetAsync(d_removed, 0, sizeof(unsigned long long), ctx->stream));
    int grid = grid_for(ctx, m->capacity, BLOCK * 4);
    if (m->W == 1) hipLaunchKernelGGL(k_filter_lt<1>, dim3(grid), dim3(BLOCK), 0, ctx->stream, (Slot<1> *)m->slots, m->capacity, rounds, d_removed);
    else hipLaunchKernelGGL(k_filter_lt<2>, dim3(grid), dim3(BLOCK), 0, ctx->stream, (Slot<2> *)m->slots, m->capacity, rounds, d_removed);
    hipError_t e = hipGetLastError();
    unsigned long long removed = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&removed, d_removed, sizeof(removed), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_removed);
    if (e != hipSuccess) return hip_fail(ctx, e, "gk_map_filter_lt");
    m->size -= removed;
    m->tombstones += removed;
    unsigned long long sz = m->size;
    GK_HIP(ctx, hipMemcpyAsync(&m->d_ctr->size, &sz, sizeof(sz), hipMemcpyHostToDevice, ctx->stream));
    GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    // The reference rescales after deleteAll (ArrayDNAMap.scala:214).  Here: rebuild into a table
    // sized for the survivors whenever tombstones exist, so that the read-only graph phase probes a
    // clean, cache-friendlier table.
    if (m->tombstones) {
        uint32_t nnb2, nlnb1;
        uint64_t ncap;
        plan_segments(m->W, (uint64_t)((double)m->size / target_load(m)) + 1, &nnb2, &nlnb1, &ncap);
        if (ncap > m->capacity) { nnb2 = m->nb2; nlnb1 = m->lnb1; ncap = m->capacity; }
        void *nslots = nullptr;
        if (alloc_table(ctx, m->W, ncap, &nslots) != GK_OK) return GK_OK;   // keep tombstones if memory is short
        int g2 = grid_for(ctx, m->capacity, BLOCK);
        if (m->W == 1)
            hipLaunchKernelGGL(k_rehash<1>, dim3(g2), dim3(BLOCK), 0, ctx->stream, (const Slot<1> *)m->slots, m->capacity,
                               Table<1>{(Slot<1> *)nslots, nnb2, nlnb1, 0u}, m->d_ctr);
        else
            hipLaunchKernelGGL(k_rehash<2>, dim3(g2), dim3(BLOCK), 0, ctx->stream, (const Slot<2> *)m->slots, m->capacity,
                               Table<2>{(Slot<2> *)nslots, nnb2, nlnb1, m->k == 64 ? 1u : 0u}, m->d_ctr);
        GK_HIP(ctx, hipGetLastError());
        GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        GK_HIP(ctx, hipFree(m->slots));
        m->slots = nslots;
        m->capacity = ncap;
        m->nb2 = nnb2;
        m->lnb1 = nlnb1;
        m->tombstones = 0;
        return map_sync_counters(m);
    }
    return GK_OK;
}

int gk_map_get_batch(gk_map *m, const uint64_t *lo, const uint64_t *hi, uint64_t n, int32_t *counts_out, uint8_t *found_out) {
    if (int rc = check_map(m)) return rc;
    gk_ctx *ctx = m->ctx;
    if (n == 0) return GK_OK;
    if (!lo || (m->W == 2 && !hi)) return fail(ctx, GK_E_INVALID, "null key array");
    if (int rc = check_key_bits(m, lo, hi, n)) return rc;
    u64 *d_lo = nullptr, *d_hi = nullptr;
    i32 *d_cnt = nullptr;
    uint8_t *d_found = nullptr;
    hipError_t e = hipMalloc((void **)&d_lo, n * 8);
    if (e == hipSuccess) e = hipMemcpyAsync(d_lo, lo, n * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && m->W == 2) {
        e = hipMalloc((void **)&d_hi, n * 8);
        if (e == hipSuccess) e = hipMemcpyAsync(d_hi, hi, n * 8, hipMemcpyHostToDevice, ctx->stream);
    }
    if (e == hipSuccess && counts_out) e = hipMalloc((void **)&d_cnt, n * 4);
    if (e == hipSuccess && found_out) e = hipMalloc((void **)&d_found, n);
    if (e == hipSuccess) {
        int grid = grid_for(ctx, n, BLOCK);
        if (m->W == 1) hipLaunchKernelGGL(k_get<1>, dim3(grid), dim3(BLOCK), 0, ctx->stream, d_lo, d_hi, n, table_of<1>(m), d_cnt, d_found);
        else hipLaunchKernelGGL(k_get<2>, dim3(grid), dim3(BLOCK), 0, ctx->stream, d_lo, d_hi, n, table_of<2>(m), d_cnt, d_found);
        e = hipGetLastError();
    }
    if (e == hipSuccess && counts_out) e = hipMemcpyAsync(counts_out, d_cnt, n * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && found_out) e = hipMemcpyAsync(found_out, d_found, n, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_lo); (void)hipFree(d_hi); (void)hipFree(d_cnt); (void)hipFree(d_found);
    if (e != hipSuccess) return hip_fail(ctx, e, "gk_map_get_batch");
    return GK_OK;
}

int gk_map_export(gk_map *m, uint64_t *lo, uint64_t *hi, int32_t *counts, uint64_t cap, uint64_t *n) {
    if (int rc = check_map(m)) return rc;
    gk_ctx *ctx = m->ctx;
    if (n) *n = m->size;
    if (m->size > cap) return fail(ctx, GK_E_CAPACITY, "export buffer too small: need " + std::to_string(m->size));
    if (m->size == 0) return GK_OK;
    if (!lo || !counts || (m->W == 2 && !hi)) return fail(ctx, GK_E_INVALID, "null export buffer");
    const u64 cnt = m->size;
    u64 *d_lo = nullptr, *d_hi = nullptr;
    i32 *d_cnt = nullptr;
    unsigned long long *d_cursor = nullptr;
    hipError_t e = hipMalloc((void **)&d_lo, cnt * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&d_hi, cnt * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&d_cnt, cnt * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&d_cursor, 8);
    if (e == hipSuccess) e = hipMemsetAsync(d_cursor, 0, 8, ctx->stream);
    if (e == hipSuccess) {
        int grid = grid_for(ctx, m->capacity, BLOCK);
        if (m->W == 1) hipLaunchKernelGGL(k_export<1>, dim3(grid), dim3(BLOCK), 0, ctx->stream, (const Slot<1> *)m->slots, m->capacity, 0u, d_lo, d_hi, d_cnt, d_cursor);
        else hipLaunchKernelGGL(k_export<2>, dim3(grid), dim3(BLOCK), 0, ctx->stream, (const Slot<2> *)m->slots, m->capacity, m->k == 64 ? 1u : 0u, d_lo, d_hi, d_cnt, d_cursor);
        e = hipGetLastError();
    }
    unsigned long long written = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&written, d_cursor, 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(lo, d_lo, cnt * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && hi) e = hipMemcpyAsync(hi, d_hi, cnt * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(counts, d_cnt, cnt * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_lo); (void)hipFree(d_hi); (void)hipFree(d_cnt); (void)hipFree(d_cursor);
    if (e != hipSuccess) return hip_fail(ctx, e, "gk_map_export");
    if (written != cnt) return fail(ctx, GK_E_STATE, "export wrote " + std::to_string(written) + " entries, size says " + std::to_string(cnt));
    return GK_OK;
}

int gk_map_stats(gk_map *m, char *json, size_t cap) {
    if (int rc = check_map_lazy(m)) return rc;
    if (!json || cap == 0) return fail(m->ctx, GK_E_INVALID, "null stats buffer");
    int w = snprintf(json, cap,
                     "{\"k\":%d,\"key_words\":%d,\"slot_bytes\":%zu,\"slots\":%llu,\"size\":%llu,\"tombstones\":%llu,"
                     "\"load\":%.6f,\"occurrences\":%llu,\"grows\":%llu,\"last_count_kernel_ms\":%.6f,"
                     "\"last_count_occurrences\":%llu,\"partitioned_launches\":%llu,\"direct_launches\":%llu,"
                     "\"spilled_keys\":%llu,\"failed_segments\":%llu,\"retries_direct\":%llu,\"device\":%d,\"cu_count\":%d}",
                     m->k, m->W, slot_bytes(m->W), (unsigned long long)m->capacity, (unsigned long long)m->size,
                     (unsigned long long)m->tombstones, m->capacity ? (double)m->size / (double)m->capacity : 0.0,
                     (unsigned long long)m->total_occurrences, (unsigned long long)m->grows, m->last_count_ms,
                     (unsigned long long)m->last_count_occ, (unsigned long long)m->part_launches,
                     (unsigned long long)m->direct_launches, (unsigned long long)m->spilled_keys,
                     (unsigned long long)m->failed_segments, (unsigned long long)m->retries_direct, m->ctx->device, m->ctx->cu_count);
    if (w < 0 || (size_t)w >= cap) return fail(m->ctx, GK_E_CAPACITY, "stats buffer too small");
    return GK_OK;
}

int gk_map_last_phase_ms(gk_map *m, float *ms5) {
    if (int rc = check_map_lazy(m)) return rc;
    if (!ms5) return fail(m->ctx, GK_E_INVALID, "null buffer");
    for (int i = 0; i < 5; i++) ms5[i] = m->phase_ms[i];
    return GK_OK;
}

int gk_map_last_count_kernel(gk_map *m, float *ms, uint64_t *occurrences) {
    if (int rc = check_map_lazy(m)) return rc;
    if (ms) *ms = m->last_count_ms;
    if (occurrences) *occurrences = m->last_count_occ;
    return GK_OK;
}

}  // extern "C"
