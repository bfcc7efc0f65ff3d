// gk_vmap.hip — DNAMap[T] with a 64-bit value per entry and MULTIMAP inserts: the rest of `trait DNAMap`
// (S/ds/ArrayDNAMap.scala:49-60) that the count table (DNAMap[Int], gk_table.hip) does not need, and the
// position map of the reference's simplifier built on it.
//
// Reference path replaced (S/ = /root/reference/src/main/scala/ru/ifmo/genome/):
//   Container.putNew(key, v)     S/ds/ArrayDNAMap.scala:152-162   k_vm_put_new   blind insert: the same key may be stored many times
//   Container.getAll(key)        :103-113                          k_vm_get_all   every value stored under the key (a probe run to the first free slot)
//   Container.update(key, v)     :115-127                          k_vm_find_or_claim + k_vm_store_last   insert or overwrite, last writer of a batch wins
//   Container.apply(key)         :90-101                           k_vm_get_all with a limit of one
//   Graph.getGraphMap            S/data/graph/Graph.scala:90-119   gk_graph_position_map (gk_graph.hip) fills one of these
//
// Layout: the SAME segmented open-addressed table as the count map (gk_device.h: 16/32-byte slots, linear probing that wraps
// inside a 32 KiB segment, key words claimed by 64-bit CAS); the slot's two 32-bit payload words (`extra`, `aux`) hold the
// 64-bit value.  Nothing is ever deleted from these maps in the reference's flows (no tombstones here).
// GraphPosition (S/data/graph/GraphPosition.scala) is encoded as  node: id  |  edge: 1<<63 | id << 32 | dist  (genome_amd.h).
// Bound: random 64-byte sector touches (one per insert, ~1.3 per lookup); HBM random-access rate, no MFMA.
#include <algorithm>
#include <string>
#include <vector>

#include "gk_internal.h"
#include "gk_tile.h"

using namespace gk;

struct gk_vmap {
    gk_ctx *ctx = nullptr;
    int k = 0, W = 1;
    uint64_t capacity = 0;
    uint32_t nb2 = 1, lnb1 = 0;
    void *slots = nullptr;
    unsigned long long *d_ctr = nullptr;    // [0] entries, [1] error (a probe wrapped its segment)
    uint64_t size = 0;
};

template <int W> static Table<W> vtable(const gk_vmap *m) { return Table<W>{reinterpret_cast<Slot<W> *>(m->slots), m->nb2, m->lnb1, m->k == 64 ? 1u : 0u}; }
template <int W> __device__ __forceinline__ void slot_set_value(Slot<W> *s, u64 v) { s->extra = (u32)v; s->aux = (u32)(v >> 32); }
template <int W> __device__ __forceinline__ u64 slot_value(const Slot<W> *s) { return (u64)s->extra | ((u64)s->aux << 32); }

template <int W> __global__ __launch_bounds__(BLOCK) void k_vm_clear(Slot<W> *slots, u64 n) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) {
        if constexpr (W == 1) slots[i] = Slot<1>{KEY_EMPTY, 0u, 0u};
        else slots[i] = Slot<2>{KEY_EMPTY, KEY_EMPTY, 0u, 0u};
    }
}

template <int W> __device__ __forceinline__ Kmer<W> key_at(const u64 *lo, const u64 *hi, u64 i) {
    if constexpr (W == 1) return Kmer<1>{lo[i]};
    else return Kmer<2>{lo[i], hi[i]};
}

// putNew (ArrayDNAMap.scala:152-162): the first free slot of the probe sequence takes (key, v), whatever keys sit before it
template <int W>
__global__ __launch_bounds__(BLOCK) void k_vm_put_new(Table<W> t, const u64 *__restrict__ lo, const u64 *__restrict__ hi,
                                                      const u64 *__restrict__ val, u64 n, unsigned long long *ctr) {
    u32 claimed = 0;
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) {
        const Kmer<W> key = key_at<W>(lo, hi, i);
        const u64 h = slot_hash(key);
        Slot<W> *seg = t.slots + ((u64)seg_of(t, h) << SegBits<W>::value);
        const i64 at = seg_claim_unique(seg, seg_pos<W>(h), key, t.tagged);
        if (at < 0) { ctr[1] = 1; continue; }
        slot_set_value(&seg[at], val[i]);
        claimed++;
    }
    for (int d = 32; d; d >>= 1) claimed += __shfl_down(claimed, d);
    if ((threadIdx.x & 63) == 0 && claimed) atomicAdd(&ctr[0], (unsigned long long)claimed);
}

// a 128-bit key is visible only once BOTH words are in place (seg_claim_unique writes w0 then w1): a reader that meets a slot
// whose w0 matches and whose w1 is still EMPTY is looking at an insert of ANOTHER launch phase, which these maps never overlap
// with lookups (handles are externally synchronised), so plain loads suffice here.
template <int W> __device__ __forceinline__ bool slot_has_key(const Slot<W> *s, Kmer<W> key, u32 tagged, u64 index) {
    if constexpr (W == 1) return s->w0 == key.lo;
    else {
        const Stored<2> k = to_stored(key);
        return s->w0 == k.w0 && s->w1 == k.w1 && (!tagged || (u32)(index & 3u) == key_tag(key));
    }
}

// getAll (ArrayDNAMap.scala:103-113): walk the probe sequence to the first free slot, every slot holding the key contributes.
// out == nullptr: count only.  The reference prepends as it goes (`ans ::= v`), so its list is in REVERSE probe order;
// values are written that way too (harmless: the simplifier treats them as sets).
template <int W>
__global__ __launch_bounds__(BLOCK) void k_vm_get_all(Table<W> t, const u64 *__restrict__ lo, const u64 *__restrict__ hi, u64 n, u32 limit,
                                                      const unsigned long long *__restrict__ offsets, u32 *__restrict__ counts, u64 *__restrict__ out) {
    constexpr u32 smask = (1u << SegBits<W>::value) - 1u;
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) {
        const Kmer<W> key = key_at<W>(lo, hi, i);
        const u64 h = slot_hash(key);
        const u64 base = (u64)seg_of(t, h) << SegBits<W>::value;
        const Slot<W> *seg = t.slots + base;
        const u32 step = t.tagged ? 4u : 1u;
        u32 p = t.tagged ? ((seg_pos<W>(h) & ~3u) | key_tag(key)) : seg_pos<W>(h);
        u32 found = 0;
        const u32 total = out ? counts[i] : 0u;
        for (u32 s = 0; s <= smask && found < limit; s += step) {
            if (seg[p].w0 == KEY_EMPTY) break;
            if (slot_has_key(&seg[p], key, t.tagged, base + p)) {
                if (out) out[offsets[i] + (total - 1 - found)] = slot_value(&seg[p]);
                found++;
            }
            p = (p + step) & smask;
        }
        if (!out) counts[i] = found;
    }
}

// update(key, v) (ArrayDNAMap.scala:115-127), two kernels so that the LAST entry of the batch wins for a repeated key, as the
// reference's sequential loop would have it: (1) find the key's slot or claim a new one, note the highest batch index per slot;
template <int W>
__global__ __launch_bounds__(BLOCK) void k_vm_find_or_claim(Table<W> t, const u64 *__restrict__ lo, const u64 *__restrict__ hi, u64 n,
                                                            u64 *__restrict__ slot_of, u32 *__restrict__ winner, unsigned long long *ctr) {
    u32 claimed = 0;
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) {
        const Kmer<W> key = key_at<W>(lo, hi, i);
        const u64 h = slot_hash(key);
        const u64 base = (u64)seg_of(t, h) << SegBits<W>::value;
        u32 err = 0;
        // Container.update stops at the first slot that holds the key or is free: exactly seg_add's walk (add = 0: no count)
        int r;
        i64 at = -1;
        {
            struct NoAdd { __device__ void operator()(u32 *, u32) const {} };
            r = seg_add(t.slots + base, seg_pos<W>(h), key, 0u, GlobalCas(), NoAdd(), t.tagged);
            if (r >= 0) at = seg_find(t.slots + base, seg_pos<W>(h), key, t.tagged);
        }
        if (r < 0 || at < 0) { ctr[1] = 1; slot_of[i] = ~0ull; err = 1; }
        if (err) continue;
        claimed += (u32)r;
        slot_of[i] = base + (u64)at;
        atomicMax(&winner[base + (u64)at], (u32)(i + 1));
    }
    for (int d = 32; d; d >>= 1) claimed += __shfl_down(claimed, d);
    if ((threadIdx.x & 63) == 0 && claimed) atomicAdd(&ctr[0], (unsigned long long)claimed);
}
// (2) the winners store their value and reset the note
template <int W>
__global__ __launch_bounds__(BLOCK) void k_vm_store_last(Table<W> t, const u64 *__restrict__ val, u64 n, const u64 *__restrict__ slot_of, u32 *__restrict__ winner) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) {
        const u64 s = slot_of[i];
        if (s == ~0ull || winner[s] != (u32)(i + 1)) continue;
        slot_set_value(&t.slots[s], val[i]);
    }
}
__global__ __launch_bounds__(BLOCK) void k_vm_reset_winner(const u64 *__restrict__ slot_of, u64 n, u32 *__restrict__ winner) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK)
        if (slot_of[i] != ~0ull) winner[slot_of[i]] = 0u;
}

template <int W>
__global__ __launch_bounds__(BLOCK) void k_vm_rehash(const Slot<W> *__restrict__ old, u64 ncap, Table<W> t, unsigned long long *ctr) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < ncap; i += (u64)gridDim.x * BLOCK) {
        if (old[i].w0 == KEY_EMPTY) continue;
        const Kmer<W> key = slot_key(old, i, t.tagged);
        const u64 h = slot_hash(key);
        Slot<W> *seg = t.slots + ((u64)seg_of(t, h) << SegBits<W>::value);
        const i64 at = seg_claim_unique(seg, seg_pos<W>(h), key, t.tagged);       // multimap: equal keys each take a slot of their own
        if (at < 0) { ctr[1] = 1; continue; }
        seg[at].extra = old[i].extra;
        seg[at].aux = old[i].aux;
    }
}

template <int W>
__global__ __launch_bounds__(BLOCK) void k_vm_export(const Slot<W> *__restrict__ slots, u64 ncap, u32 tagged, u64 *lo, u64 *hi, u64 *val,
                                                     unsigned long long *cursor) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < ncap; i += (u64)gridDim.x * BLOCK) {
        if (slots[i].w0 == KEY_EMPTY) continue;
        const Kmer<W> key = slot_key(slots, i, tagged);
        const u64 o = atomicAdd(cursor, 1ull);
        lo[o] = key.lo;
        if constexpr (W == 2) { if (hi) hi[o] = key.hi; }
        else { if (hi) hi[o] = 0; }
        val[o] = slot_value(&slots[i]);
    }
}

namespace {

int vgrid(const gk_ctx *ctx, u64 items) { return (int)std::min<u64>(std::max<u64>((items + BLOCK - 1) / BLOCK, 1), (u64)ctx->cu_count * 8); }

int vm_check(const gk_vmap *m) {
    if (!m || !m->ctx) return fail(nullptr, GK_E_INVALID, "null value-map handle");
    hipError_t e = hipSetDevice(m->ctx->device);
    if (e != hipSuccess) return hip_fail(m->ctx, e, "hipSetDevice");
    return GK_OK;
}

int vm_alloc_table(gk_ctx *ctx, int W, u64 cap, void **out) {
    hipError_t e = hipMalloc(out, cap * slot_bytes(W));
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(ctx, GK_E_CAPACITY, std::string("value map: cannot allocate table: ") + hipGetErrorString(e)); }
    if (W == 1) hipLaunchKernelGGL(k_vm_clear<1>, dim3(vgrid(ctx, cap / 4)), dim3(BLOCK), 0, ctx->stream, (Slot<1> *)*out, cap);
    else hipLaunchKernelGGL(k_vm_clear<2>, dim3(vgrid(ctx, cap / 4)), dim3(BLOCK), 0, ctx->stream, (Slot<2> *)*out, cap);
    GK_HIP(ctx, hipGetLastError());
    return GK_OK;
}

int vm_sync(gk_vmap *m) {
    unsigned long long h[2] = {0, 0};
    GK_HIP(m->ctx, hipMemcpyAsync(h, m->d_ctr, 16, hipMemcpyDeviceToHost, m->ctx->stream));
    GK_HIP(m->ctx, hipStreamSynchronize(m->ctx->stream));
    m->size = h[0];
    if (h[1]) {
        GK_HIP(m->ctx, hipMemsetAsync(m->d_ctr + 1, 0, 8, m->ctx->stream));
        // every copy of a key lives in the ONE segment its hash names (2048 entries of 8-byte keys, 1024 of 16-byte keys, a quarter
        // of that at k = 64): a key stored more often than that — or a heavy key plus its segment's other keys — cannot be helped by
        // growing the table (the reference's putNew probes the whole table and has no such bound: ArrayDNAMap.scala:152-162)
        return fail(m->ctx, GK_E_CAPACITY, "value map: a segment filled up — one key stored more often than a segment holds (" +
                                               std::to_string((1u << seg_bits_for(m->W)) / (m->k == 64 ? 4u : 1u)) +
                                               " entries)?  The batch's other entries ARE stored (gk_vmap_size counts them)");
    }
    return GK_OK;
}

// room for `extra` more entries (every putNew takes a slot; load kept under 0.7, sized for 0.5: multimap runs are longer than a set's)
int vm_reserve(gk_vmap *m, u64 extra) {
    const double max_load = m->k == 64 ? 0.5 : 0.7, target = m->k == 64 ? 0.35 : 0.5;
    if ((double)(m->size + extra) <= max_load * (double)m->capacity) return GK_OK;
    gk_ctx *ctx = m->ctx;
    uint32_t nnb2, nlnb1;
    uint64_t ncap;
    plan_segments(m->W, std::max<u64>((u64)((double)(m->size + extra) / target) + 1, m->capacity + m->capacity / 2), &nnb2, &nlnb1, &ncap);
    void *nslots = nullptr;
    if (int rc = vm_alloc_table(ctx, m->W, ncap, &nslots)) { if (nslots) (void)hipFree(nslots); return rc; }
    if (m->W == 1)
        hipLaunchKernelGGL(k_vm_rehash<1>, dim3(vgrid(ctx, m->capacity)), dim3(BLOCK), 0, ctx->stream, (const Slot<1> *)m->slots, m->capacity,
                           Table<1>{(Slot<1> *)nslots, nnb2, nlnb1, 0u}, m->d_ctr);
    else
        hipLaunchKernelGGL(k_vm_rehash<2>, dim3(vgrid(ctx, m->capacity)), dim3(BLOCK), 0, ctx->stream, (const Slot<2> *)m->slots, m->capacity,
                           Table<2>{(Slot<2> *)nslots, nnb2, nlnb1, m->k == 64 ? 1u : 0u}, m->d_ctr);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { (void)hipFree(nslots); return hip_fail(ctx, e, "value map rehash"); }
    if (int rc = vm_sync(m)) { (void)hipFree(nslots); return rc; }          // a failed rehash leaves the old table in place
    GK_HIP(ctx, hipFree(m->slots));
    m->slots = nslots; m->capacity = ncap; m->nb2 = nnb2; m->lnb1 = nlnb1;
    return GK_OK;
}

int vm_check_keys(const gk_vmap *m, const uint64_t *lo, const uint64_t *hi, uint64_t n) {
    if (!lo || (m->W == 2 && !hi)) return fail(m->ctx, GK_E_INVALID, "null key array");
    const int k = m->k;
    for (uint64_t i = 0; i < n; i++) {
        const bool bad = m->W == 1 ? ((lo[i] >> (2 * k)) != 0 || (hi && hi[i] != 0)) : (k < 64 && (hi[i] >> (2 * (k - 32))) != 0);
        if (bad) return fail(m->ctx, GK_E_KLEN, "key " + std::to_string(i) + " is not a " + std::to_string(k) + "-mer (bits set above 2k)");   // ArrayDNAMap.scala:182
    }
    return GK_OK;
}

struct DevBuf {      // a few device arrays for one call, freed together
    gk_ctx *ctx;
    explicit DevBuf(gk_ctx *c) : ctx(c) {}
    std::vector<void *> ptrs;
    ~DevBuf() { for (void *p : ptrs) if (p) (void)hipFree(p); }
    template <class T> hipError_t get(T **p, u64 n) {
        hipError_t e = hipMalloc((void **)p, std::max<u64>(n, 1) * sizeof(T));
        if (e == hipSuccess) ptrs.push_back(*p);
        return e;
    }
};

}  // namespace

namespace gk {
// device-side putNew for arrays already in HBM (gk_graph_position_map)
int vmap_put_new_dev(gk_vmap *m, const uint64_t *d_lo, const uint64_t *d_hi, const uint64_t *d_val, uint64_t n) {
    if (int rc = vm_reserve(m, n)) return rc;
    gk_ctx *ctx = m->ctx;
    if (n) {
        if (m->W == 1) hipLaunchKernelGGL(k_vm_put_new<1>, dim3(vgrid(ctx, n)), dim3(BLOCK), 0, ctx->stream, vtable<1>(m), d_lo, d_hi, d_val, n, m->d_ctr);
        else hipLaunchKernelGGL(k_vm_put_new<2>, dim3(vgrid(ctx, n)), dim3(BLOCK), 0, ctx->stream, vtable<2>(m), d_lo, d_hi, d_val, n, m->d_ctr);
        GK_HIP(ctx, hipGetLastError());
    }
    return vm_sync(m);
}
int vmap_k(const gk_vmap *m) { return m->k; }
gk_ctx *vmap_ctx(const gk_vmap *m) { return m->ctx; }
// getAll for keys already in HBM, results left in HBM (the paired-end stage, gk_pairs.hip): pass 0 counts into d_cnt[n]; the
// caller turns the counts into CSR offsets d_off[n + 1] and allocates d_out; pass 1 fills.  Stream-ordered, nothing is waited for.
int vmap_get_all_dev(gk_vmap *m, const uint64_t *d_lo, const uint64_t *d_hi, uint64_t n, const unsigned long long *d_off, uint32_t *d_cnt, uint64_t *d_out) {
    gk_ctx *ctx = m->ctx;
    if (n == 0) return GK_OK;
    const int grid = vgrid(ctx, n);
    if (m->W == 1) hipLaunchKernelGGL(k_vm_get_all<1>, dim3(grid), dim3(BLOCK), 0, ctx->stream, vtable<1>(m), d_lo, d_hi, n, 0xffffffffu, d_off, d_cnt, d_out);
    else hipLaunchKernelGGL(k_vm_get_all<2>, dim3(grid), dim3(BLOCK), 0, ctx->stream, vtable<2>(m), d_lo, d_hi, n, 0xffffffffu, d_off, d_cnt, d_out);
    GK_HIP(ctx, hipGetLastError());
    return GK_OK;
}
}  // namespace gk

extern "C" {

int gk_vmap_create(gk_ctx *ctx, int k, uint64_t capacity_hint, gk_vmap **out) {
    if (!ctx || !out) return fail(ctx, GK_E_INVALID, "gk_vmap_create: null argument");
    *out = nullptr;
    if (!k_supported(k)) return fail(ctx, GK_E_UNSUPPORTED_K, "k=" + std::to_string(k) + " unsupported (2..31 and 34..64)");
    GK_HIP(ctx, hipSetDevice(ctx->device));
    gk_vmap *m = new gk_vmap();
    m->ctx = ctx; m->k = k; m->W = words_for_k(k);
    plan_segments(m->W, (uint64_t)((double)std::max<uint64_t>(capacity_hint, 1024) / (k == 64 ? 0.35 : 0.5)) + 1, &m->nb2, &m->lnb1, &m->capacity);
    int rc = vm_alloc_table(ctx, m->W, m->capacity, &m->slots);
    if (rc == GK_OK) {
        hipError_t e = hipMalloc((void **)&m->d_ctr, 16);
        if (e == hipSuccess) e = hipMemsetAsync(m->d_ctr, 0, 16, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = hip_fail(ctx, e, "gk_vmap_create");
    }
    if (rc != GK_OK) {
        if (m->slots) (void)hipFree(m->slots);
        if (m->d_ctr) (void)hipFree(m->d_ctr);
        delete m;
        return rc;
    }
    *out = m;
    return GK_OK;
}

void gk_vmap_destroy(gk_vmap *m) {
    if (!m) return;
    gk_ctx *ctx = m->ctx;
    (void)hipSetDevice(m->ctx->device);
    (void)hipStreamSynchronize(m->ctx->stream);
    if (m->slots) (void)hipFree(m->slots);
    if (m->d_ctr) (void)hipFree(m->d_ctr);
    delete m;
}

int gk_vmap_k(const gk_vmap *m) { return m ? m->k : 0; }

int gk_vmap_size(gk_vmap *m, uint64_t *n) {
    if (int rc = vm_check(m)) return rc;
    if (!n) return fail(m->ctx, GK_E_INVALID, "gk_vmap_size: n is NULL");
    *n = m->size;
    return GK_OK;
}

int gk_vmap_put_new_batch(gk_vmap *m, const uint64_t *lo, const uint64_t *hi, const uint64_t *values, uint64_t n) {
    if (int rc = vm_check(m)) return rc;
    gk_ctx *ctx = m->ctx;
    if (n == 0) return GK_OK;
    if (!values) return fail(ctx, GK_E_INVALID, "null value array");
    if (int rc = vm_check_keys(m, lo, hi, n)) return rc;
    DevBuf b(ctx);
    u64 *d_lo = nullptr, *d_hi = nullptr, *d_val = nullptr;
    hipError_t e = b.get(&d_lo, n);
    if (e == hipSuccess && m->W == 2) e = b.get(&d_hi, n);
    if (e == hipSuccess) e = b.get(&d_val, n);
    if (e == hipSuccess) e = hipMemcpyAsync(d_lo, lo, n * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && d_hi) e = hipMemcpyAsync(d_hi, hi, n * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_val, values, n * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) return hip_fail(ctx, e, "gk_vmap_put_new_batch");
    return vmap_put_new_dev(m, d_lo, d_hi, d_val, n);
}

int gk_vmap_update_batch(gk_vmap *m, const uint64_t *lo, const uint64_t *hi, const uint64_t *values, uint64_t n) {
    if (int rc = vm_check(m)) return rc;
    gk_ctx *ctx = m->ctx;
    if (n == 0) return GK_OK;
    if (n >= 0xffffffffull) return fail(ctx, GK_E_INVALID, "gk_vmap_update_batch: at most 2^32 - 2 keys per call");
    if (!values) return fail(ctx, GK_E_INVALID, "null value array");
    if (int rc = vm_check_keys(m, lo, hi, n)) return rc;
    if (int rc = vm_reserve(m, n)) return rc;
    DevBuf b(ctx);
    u64 *d_lo = nullptr, *d_hi = nullptr, *d_val = nullptr, *d_slot = nullptr;
    u32 *d_win = nullptr;
    hipError_t e = b.get(&d_lo, n);
    if (e == hipSuccess && m->W == 2) e = b.get(&d_hi, n);
    if (e == hipSuccess) e = b.get(&d_val, n);
    if (e == hipSuccess) e = b.get(&d_slot, n);
    if (e == hipSuccess) e = b.get(&d_win, m->capacity);
    if (e == hipSuccess) e = hipMemsetAsync(d_win, 0, m->capacity * 4, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_lo, lo, n * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && d_hi) e = hipMemcpyAsync(d_hi, hi, n * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_val, values, n * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) return hip_fail(ctx, e, "gk_vmap_update_batch");
    const int grid = vgrid(ctx, n);
    if (m->W == 1) {
        hipLaunchKernelGGL(k_vm_find_or_claim<1>, dim3(grid), dim3(BLOCK), 0, ctx->stream, vtable<1>(m), d_lo, d_hi, n, d_slot, d_win, m->d_ctr);
        hipLaunchKernelGGL(k_vm_store_last<1>, dim3(grid), dim3(BLOCK), 0, ctx->stream, vtable<1>(m), d_val, n, d_slot, d_win);
    } else {
        hipLaunchKernelGGL(k_vm_find_or_claim<2>, dim3(grid), dim3(BLOCK), 0, ctx->stream, vtable<2>(m), d_lo, d_hi, n, d_slot, d_win, m->d_ctr);
        hipLaunchKernelGGL(k_vm_store_last<2>, dim3(grid), dim3(BLOCK), 0, ctx->stream, vtable<2>(m), d_val, n, d_slot, d_win);
    }
    GK_HIP(ctx, hipGetLastError());
    return vm_sync(m);
}

int gk_vmap_get_all_batch(gk_vmap *m, const uint64_t *lo, const uint64_t *hi, uint64_t n, uint64_t *offsets_out, uint64_t *values_out,
                          uint64_t values_cap, uint64_t *total) {
    if (int rc = vm_check(m)) return rc;
    gk_ctx *ctx = m->ctx;
    if (total) *total = 0;
    if (!offsets_out) return fail(ctx, GK_E_INVALID, "gk_vmap_get_all_batch: offsets_out is NULL");
    offsets_out[0] = 0;
    if (n == 0) return GK_OK;
    if (int rc = vm_check_keys(m, lo, hi, n)) return rc;
    DevBuf b(ctx);
    u64 *d_lo = nullptr, *d_hi = nullptr, *d_out = nullptr;
    u32 *d_cnt = nullptr;
    unsigned long long *d_off = nullptr;
    hipError_t e = b.get(&d_lo, n);
    if (e == hipSuccess && m->W == 2) e = b.get(&d_hi, n);
    if (e == hipSuccess) e = b.get(&d_cnt, n);
    if (e == hipSuccess) e = b.get(&d_off, n + 1);
    if (e == hipSuccess) e = hipMemcpyAsync(d_lo, lo, n * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && d_hi) e = hipMemcpyAsync(d_hi, hi, n * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) return hip_fail(ctx, e, "gk_vmap_get_all_batch");
    const int grid = vgrid(ctx, n);
    const u32 nolimit = 0xffffffffu;
    if (m->W == 1) hipLaunchKernelGGL(k_vm_get_all<1>, dim3(grid), dim3(BLOCK), 0, ctx->stream, vtable<1>(m), d_lo, d_hi, n, nolimit, nullptr, d_cnt, nullptr);
    else hipLaunchKernelGGL(k_vm_get_all<2>, dim3(grid), dim3(BLOCK), 0, ctx->stream, vtable<2>(m), d_lo, d_hi, n, nolimit, nullptr, d_cnt, nullptr);
    GK_HIP(ctx, hipGetLastError());
    std::vector<u32> cnt(n);
    GK_HIP(ctx, hipMemcpyAsync(cnt.data(), d_cnt, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (uint64_t i = 0; i < n; i++) offsets_out[i + 1] = offsets_out[i] + cnt[i];
    const uint64_t tot = offsets_out[n];
    if (total) *total = tot;
    if (tot > values_cap) return fail(ctx, GK_E_CAPACITY, "value buffer too small: need " + std::to_string(tot));
    if (tot == 0) return GK_OK;
    if (!values_out) return fail(ctx, GK_E_INVALID, "null value buffer");
    e = b.get(&d_out, tot);
    if (e == hipSuccess) e = hipMemcpyAsync(d_off, offsets_out, (n + 1) * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) return hip_fail(ctx, e, "gk_vmap_get_all_batch");
    if (m->W == 1) hipLaunchKernelGGL(k_vm_get_all<1>, dim3(grid), dim3(BLOCK), 0, ctx->stream, vtable<1>(m), d_lo, d_hi, n, nolimit, d_off, d_cnt, d_out);
    else hipLaunchKernelGGL(k_vm_get_all<2>, dim3(grid), dim3(BLOCK), 0, ctx->stream, vtable<2>(m), d_lo, d_hi, n, nolimit, d_off, d_cnt, d_out);
    GK_HIP(ctx, hipGetLastError());
    GK_HIP(ctx, hipMemcpyAsync(values_out, d_out, tot * 8, hipMemcpyDeviceToHost, ctx->stream));
    GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GK_OK;
}

int gk_vmap_get_batch(gk_vmap *m, const uint64_t *lo, const uint64_t *hi, uint64_t n, uint64_t *values_out, uint8_t *found_out) {
    if (int rc = vm_check(m)) return rc;
    gk_ctx *ctx = m->ctx;
    if (n == 0) return GK_OK;
    if (!values_out) return fail(ctx, GK_E_INVALID, "null value buffer");
    if (int rc = vm_check_keys(m, lo, hi, n)) return rc;
    DevBuf b(ctx);
    u64 *d_lo = nullptr, *d_hi = nullptr, *d_out = nullptr;
    u32 *d_cnt = nullptr;
    unsigned long long *d_off = nullptr;
    std::vector<unsigned long long> off(n + 1);
    for (uint64_t i = 0; i <= n; i++) off[i] = i;
    hipError_t e = b.get(&d_lo, n);
    if (e == hipSuccess && m->W == 2) e = b.get(&d_hi, n);
    if (e == hipSuccess) e = b.get(&d_cnt, n);
    if (e == hipSuccess) e = b.get(&d_off, n + 1);
    if (e == hipSuccess) e = b.get(&d_out, n);
    if (e == hipSuccess) e = hipMemcpyAsync(d_lo, lo, n * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && d_hi) e = hipMemcpyAsync(d_hi, hi, n * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_off, off.data(), (n + 1) * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_out, 0, n * 8, ctx->stream);
    if (e != hipSuccess) return hip_fail(ctx, e, "gk_vmap_get_batch");
    const int grid = vgrid(ctx, n);
    // apply (ArrayDNAMap.scala:90-101) = the first slot of the probe sequence that holds the key: getAll cut off at one
    for (int pass = 0; pass < 2; pass++) {
        if (m->W == 1) hipLaunchKernelGGL(k_vm_get_all<1>, dim3(grid), dim3(BLOCK), 0, ctx->stream, vtable<1>(m), d_lo, d_hi, n, 1u, d_off, d_cnt, pass ? d_out : nullptr);
        else hipLaunchKernelGGL(k_vm_get_all<2>, dim3(grid), dim3(BLOCK), 0, ctx->stream, vtable<2>(m), d_lo, d_hi, n, 1u, d_off, d_cnt, pass ? d_out : nullptr);
    }
    GK_HIP(ctx, hipGetLastError());
    std::vector<u32> cnt(n);
    GK_HIP(ctx, hipMemcpyAsync(cnt.data(), d_cnt, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    GK_HIP(ctx, hipMemcpyAsync(values_out, d_out, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (found_out) for (uint64_t i = 0; i < n; i++) found_out[i] = cnt[i] ? 1 : 0;
    return GK_OK;
}

int gk_vmap_export(gk_vmap *m, uint64_t *lo, uint64_t *hi, uint64_t *values, uint64_t cap, uint64_t *n) {
    if (int rc = vm_check(m)) return rc;
    gk_ctx *ctx = m->ctx;
    if (n) *n = m->size;
    if (m->size > cap) return fail(ctx, GK_E_CAPACITY, "export buffer too small: need " + std::to_string(m->size));
    if (m->size == 0) return GK_OK;
    if (!lo || !values || (m->W == 2 && !hi)) return fail(ctx, GK_E_INVALID, "null export buffer");
    DevBuf b(ctx);
    u64 *d_lo = nullptr, *d_hi = nullptr, *d_val = nullptr;
    unsigned long long *d_cur = nullptr;
    const u64 cnt = m->size;
    hipError_t e = b.get(&d_lo, cnt);
    if (e == hipSuccess) e = b.get(&d_hi, cnt);
    if (e == hipSuccess) e = b.get(&d_val, cnt);
    if (e == hipSuccess) e = b.get(&d_cur, 1);
    if (e == hipSuccess) e = hipMemsetAsync(d_cur, 0, 8, ctx->stream);
    if (e != hipSuccess) return hip_fail(ctx, e, "gk_vmap_export");
    if (m->W == 1) hipLaunchKernelGGL(k_vm_export<1>, dim3(vgrid(ctx, m->capacity)), dim3(BLOCK), 0, ctx->stream, (const Slot<1> *)m->slots, m->capacity, 0u, d_lo, d_hi, d_val, d_cur);
    else hipLaunchKernelGGL(k_vm_export<2>, dim3(vgrid(ctx, m->capacity)), dim3(BLOCK), 0, ctx->stream, (const Slot<2> *)m->slots, m->capacity, m->k == 64 ? 1u : 0u, d_lo, d_hi, d_val, d_cur);
    GK_HIP(ctx, hipGetLastError());
    GK_HIP(ctx, hipMemcpyAsync(lo, d_lo, cnt * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (hi) GK_HIP(ctx, hipMemcpyAsync(hi, d_hi, cnt * 8, hipMemcpyDeviceToHost, ctx->stream));
    GK_HIP(ctx, hipMemcpyAsync(values, d_val, cnt * 8, hipMemcpyDeviceToHost, ctx->stream));
    GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GK_OK;
}

}  // extern "C"
