// gk_dist.hip — PartitionedDNAMap across the GPUs of one node, behind the C-ABI: one rank per GPU, one table partition
// per rank, k-mers routed to their owner with ONE all-to-all over RCCL/xGMI per batch.
//
// Reference path replaced (S/ = /root/reference/src/main/scala/ru/ifmo/genome/):
//   PartitionedDNAMap.update / update1   S/ds/PartitionedDNAMap.scala:37-47   gk_dist_count_reads_dev (route -> exchange -> owner count)
//   PartitionedDNAMap.partition          :60-63                               gk_owner_of (strand-symmetric minimizer, gk_device.h)
//   PartitionedDNAMap.size               :31                                  gk_dist_size (all-reduce of one integer)
//   deleteAll / mapReduce scatter-gather :49-58                               local calls on every rank's gk_map (no data exchange)
//   "the whole k-mer set" for Graph.buildGraph (Graph.scala:269)              gk_dist_gather_map (all-gather of the survivors, device to device)
//
// The reference sends one Akka message per k-mer occurrence (driver -> owner actor).  Here a rank turns its reads into
// SUPER-K-MER records grouped by owner (gk_skm.hip: ~2 bits per base instead of 8/16 bytes per k-mer), exchanges the
// per-owner (records, k-mers) counts, then the records — sizes are exact, nothing is padded or packed: region p of the
// send buffer goes straight to rank p with ncclSend, what arrives lands back to back — and counts what it received with
// the same pipeline as reads (gk_map_count_superkmers_dev).  xGMI is point to point (7 links per GPU): an all-to-all puts
// one peer on each link, which is why this is a grouped send/recv and not a ring collective.
//
// RCCL is loaded at run time (dlopen), so the library has no link-time dependency on it: a single-GPU user never needs
// it, and a process that already holds an RCCL (PyTorch bundles one) shares that copy.
#include <dlfcn.h>
#include <time.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "gk_internal.h"
#include "gk_tile.h"

using namespace gk;

// ---- the handful of RCCL entry points used (rccl.h: NCCL-compatible ABI) ---------------------------------------------
namespace {
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { ncclSuccess = 0 };
enum { ncclInt8 = 0, ncclUint8 = 1, ncclUint64 = 5, ncclFloat64 = 8 };      // ncclDataType_t
enum { ncclSum = 0, ncclMax = 2 };                                           // ncclRedOp_t

struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(ncclUniqueId *) = nullptr;
    int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string why;
};

Rccl *rccl() {
    static Rccl r;
    static bool tried = false;
    if (tried) return &r;
    tried = true;
    // an RCCL this process already holds (PyTorch's) first, then the ROCm one
    for (const char *name : {"librccl.so", "librccl.so.1"}) {
        r.lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
        if (r.lib) break;
    }
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        if (r.lib) break;
        r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    }
    if (!r.lib) { r.why = std::string("cannot load librccl: ") + dlerror(); return &r; }
    auto sym = [&](const char *n) { void *p = dlsym(r.lib, n); if (!p && r.why.empty()) r.why = std::string("librccl lacks ") + n; return p; };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.Send = (decltype(r.Send))sym("ncclSend");
    r.Recv = (decltype(r.Recv))sym("ncclRecv");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
    r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    return &r;
}
}  // namespace

struct gk_dist {
    gk_ctx *ctx = nullptr;
    int rank = 0, world = 1;
    ncclComm_t comm = nullptr;
    // exchange scratch, kept between calls.  TWO send buffers: gk_dist_route_begin fills one on the second stream while the
    // records of the previous batch still leave the other one.
    uint8_t *d_sendbuf[2] = {nullptr, nullptr}, *d_recv = nullptr;
    u64 send_cap[2] = {0, 0}, recv_records = 0;      // capacities in record slots (send: world regions of send_cap / world)
    int cur = 0;                                     // the send buffer the next route goes to
    int slot = 0;                                    // record slot bytes the buffers were sized for
    // routes that were begun and not yet consumed: at most two (one per send buffer), consumed first in, first out
    struct Route { int k = 0, read_len = 0; const void *records = nullptr; u64 nreads = 0; };
    Route route[2];
    int npending = 0;                                // route[(cur - npending) & 1] is the oldest
    unsigned long long *d_route_cnt = nullptr;       // [2][SKM_COUNT_WORDS] counters of the routing kernels on the second stream
    unsigned long long *h_route_cnt = nullptr;       // pinned copy
    hipEvent_t route_done[2] = {nullptr, nullptr};   // recorded behind each route's counter copy
    unsigned long long *d_cnt = nullptr;             // [4 * world]: (records, k-mers) per peer to send, then as received
    unsigned long long *h_cnt = nullptr;             // pinned mirror
    float last_ms[4] = {0, 0, 0, 0};                 // route, exchange, owner count, total (wall)
};

#define GK_NCCL(ctx, call)                                                                                        \
    do {                                                                                                          \
        int r__ = (call);                                                                                         \
        if (r__ != ncclSuccess)                                                                                   \
            return gk::fail((ctx), GK_E_COMM, std::string(#call) + ": " + (rccl()->GetErrorString ? rccl()->GetErrorString(r__) : "RCCL error")); \
    } while (0)

// live (key, count) of a table packed for the wire: keys interleaved W words each (what k_add_keys takes), counts apart
template <int W>
__global__ __launch_bounds__(BLOCK) void k_export_packed(const Slot<W> *__restrict__ slots, u64 ncap, u32 tagged, u64 *keys, i32 *cnt,
                                                         unsigned long long *cursor) {
    __shared__ unsigned long long s_base;
    __shared__ u32 wsum[BLOCK / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u64 ngroups = (ncap + BLOCK - 1) / BLOCK;
    for (u64 g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const u64 i = g * BLOCK + threadIdx.x;
        const bool live = i < ncap && slot_live(&slots[i]);
        const unsigned long long b = __ballot(live);
        const u32 wprefix = (u32)__popcll(b & ((1ull << lane) - 1ull));
        __syncthreads();
        if (lane == 0) wsum[wave] = (u32)__popcll(b);
        __syncthreads();
        u32 base = 0, tot = 0;
        for (int w = 0; w < BLOCK / 64; ++w) { if (w < wave) base += wsum[w]; tot += wsum[w]; }
        if (threadIdx.x == 0 && tot) s_base = atomicAdd(cursor, (unsigned long long)tot);
        __syncthreads();
        if (live) {
            const Kmer<W> key = slot_key(slots, i, tagged);
            const u64 o = s_base + base + wprefix;
            if constexpr (W == 1) keys[o] = key.lo;
            else { keys[2 * o] = key.lo; keys[2 * o + 1] = key.hi; }
            cnt[o] = (i32)slot_count(&slots[i]);
        }
    }
}

static int dist_check(const gk_dist *d) {
    if (!d || !d->ctx || !d->comm) return fail(nullptr, GK_E_INVALID, "null or closed gk_dist handle");
    hipError_t e = hipSetDevice(d->ctx->device);
    if (e != hipSuccess) return hip_fail(d->ctx, e, "hipSetDevice");
    return GK_OK;
}

static double now_ms() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

extern "C" {

int gk_dist_unique_id(void *id128) {
    if (!id128) return fail(nullptr, GK_E_INVALID, "gk_dist_unique_id: null buffer");
    Rccl *r = rccl();
    if (!r->why.empty()) return fail(nullptr, GK_E_COMM, r->why);
    ncclUniqueId id;
    GK_NCCL(nullptr, r->GetUniqueId(&id));
    memcpy(id128, id.internal, 128);
    return GK_OK;
}

int gk_dist_create(gk_ctx *ctx, int rank, int world, const void *id128, gk_dist **out) {
    if (!ctx || !out || !id128) return fail(ctx, GK_E_INVALID, "gk_dist_create: null argument");
    *out = nullptr;
    if (world < 1 || world > 64 || rank < 0 || rank >= world) return fail(ctx, GK_E_INVALID, "gk_dist_create: need 0 <= rank < world <= 64");
    Rccl *r = rccl();
    if (!r->why.empty()) return fail(ctx, GK_E_COMM, r->why);
    GK_HIP(ctx, hipSetDevice(ctx->device));
    gk_dist *d = new gk_dist();
    d->ctx = ctx; d->rank = rank; d->world = world;
    ncclUniqueId id;
    memcpy(id.internal, id128, 128);
    int rc = r->CommInitRank(&d->comm, world, id, rank);
    if (rc != ncclSuccess) {
        delete d;
        return fail(ctx, GK_E_COMM, std::string("ncclCommInitRank: ") + (r->GetErrorString ? r->GetErrorString(rc) : "error"));
    }
    hipError_t e = hipMalloc((void **)&d->d_cnt, 4 * 64 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipHostMalloc((void **)&d->h_cnt, 4 * 64 * sizeof(unsigned long long), hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc((void **)&d->d_route_cnt, 2 * SKM_COUNT_WORDS * sizeof(unsigned long long));
    for (int i = 0; i < 2 && e == hipSuccess; i++) e = hipEventCreateWithFlags(&d->route_done[i], hipEventDisableTiming);
    if (e == hipSuccess) e = hipHostMalloc((void **)&d->h_route_cnt, 2 * SKM_COUNT_WORDS * sizeof(unsigned long long), hipHostMallocDefault);
    if (e != hipSuccess) { int c = hip_fail(ctx, e, "gk_dist_create"); r->CommDestroy(d->comm); delete d; return c; }
    *out = d;
    return GK_OK;
}

void gk_dist_destroy(gk_dist *d) {
    if (!d) return;
    gk_ctx *ctx = d->ctx;
    if (d->ctx) { (void)hipSetDevice(d->ctx->device); (void)hipStreamSynchronize(d->ctx->stream); }
    if (d->comm && rccl()->CommDestroy) (void)rccl()->CommDestroy(d->comm);
    if (d->ctx && d->ctx->copy_stream) (void)hipStreamSynchronize(d->ctx->copy_stream);
    for (int i = 0; i < 2; i++) if (d->d_sendbuf[i]) (void)hipFree(d->d_sendbuf[i]);
    if (d->d_recv) (void)hipFree(d->d_recv);
    for (int i = 0; i < 2; i++) if (d->route_done[i]) (void)hipEventDestroy(d->route_done[i]);
    if (d->d_route_cnt) (void)hipFree(d->d_route_cnt);
    if (d->h_route_cnt) (void)hipHostFree(d->h_route_cnt);
    if (d->d_cnt) (void)hipFree(d->d_cnt);
    if (d->h_cnt) (void)hipHostFree(d->h_cnt);
    delete d;
}

int gk_dist_rank(const gk_dist *d) { return d ? d->rank : -1; }
int gk_dist_world(const gk_dist *d) { return d ? d->world : 0; }

int gk_dist_barrier(gk_dist *d) {
    if (int rc = dist_check(d)) return rc;
    gk_ctx *ctx = d->ctx;
    GK_HIP(ctx, hipMemsetAsync(d->d_cnt, 0, 8, ctx->stream));
    GK_NCCL(ctx, rccl()->AllReduce(d->d_cnt, d->d_cnt, 1, ncclUint64, ncclSum, d->comm, ctx->stream));
    GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GK_OK;
}

int gk_dist_allreduce_f64(gk_dist *d, double *values, int n, int op_max) {
    if (int rc = dist_check(d)) return rc;
    gk_ctx *ctx = d->ctx;
    if (!values || n < 1 || n > 32) return fail(ctx, GK_E_INVALID, "gk_dist_allreduce_f64: 1..32 values");
    double *dv = reinterpret_cast<double *>(d->d_cnt);
    GK_HIP(ctx, hipMemcpyAsync(dv, values, n * 8, hipMemcpyHostToDevice, ctx->stream));
    GK_NCCL(ctx, rccl()->AllReduce(dv, dv, (size_t)n, ncclFloat64, op_max ? ncclMax : ncclSum, d->comm, ctx->stream));
    GK_HIP(ctx, hipMemcpyAsync(values, dv, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GK_OK;
}

int gk_dist_size(gk_dist *d, gk_map *local, uint64_t *total) {
    if (int rc = dist_check(d)) return rc;
    gk_ctx *ctx = d->ctx;
    if (!local || !total) return fail(ctx, GK_E_INVALID, "gk_dist_size: null argument");
    unsigned long long v = local->size;
    GK_HIP(ctx, hipMemcpyAsync(d->d_cnt, &v, 8, hipMemcpyHostToDevice, ctx->stream));
    GK_NCCL(ctx, rccl()->AllReduce(d->d_cnt, d->d_cnt, 1, ncclUint64, ncclSum, d->comm, ctx->stream));
    GK_HIP(ctx, hipMemcpyAsync(&v, d->d_cnt, 8, hipMemcpyDeviceToHost, ctx->stream));
    GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *total = v;
    return GK_OK;
}

static int dist_grow(gk_ctx *ctx, uint8_t **buf, u64 *have, u64 want_records, int slot) {
    if (*have >= want_records && *buf) return GK_OK;
    if (*buf) { GK_HIP(ctx, hipStreamSynchronize(ctx->stream)); GK_HIP(ctx, hipFree(*buf)); }
    *buf = nullptr; *have = 0;
    GK_HIP(ctx, hipMalloc((void **)buf, std::max<u64>(want_records, 1) * (u64)slot + 64));
    *have = want_records;
    return GK_OK;
}

static u64 route_want_records(int k, int P, u64 nreads, int read_len) {
    // a run of same-owner windows is about half a minimizer window long; a region that turns out too small is reported with
    // the size it needs and the route is taken again
    const u64 nk = read_len >= k ? (u64)(read_len - k + 1) : 0;
    const double per_read = std::max(2.0, (double)nk / std::max(1.0, (k - 9) / 2.0)) * 1.5 + 1.0;
    return std::max<u64>((u64)1024 * P, (u64)((double)nreads * per_read) / P * P);
}

// Route this rank's reads for the NEXT gk_dist_count_routed on the context's second stream and return at once: the routing
// kernel (0.6 ms per 10^6 reads, issue-bound) then overlaps whatever the main stream is doing — in a streaming loop, the
// owner pipeline of the previous batch.  The records buffer must stay valid until gk_dist_count_routed returns.
int gk_dist_route_begin(gk_dist *d, int k, const void *dev_records, uint64_t nreads, int read_len) {
    if (int rc = dist_check(d)) return rc;
    gk_ctx *ctx = d->ctx;
    if (d->npending >= 2) return fail(ctx, GK_E_STATE, "gk_dist_route_begin: two routes are already waiting (gk_dist_count_routed consumes one)");
    if (!dev_records && nreads) return fail(ctx, GK_E_INVALID, "gk_dist_route_begin: null records");
    if (read_len < 0 || read_len > 255) return fail(ctx, GK_E_FORMAT, "read_len must be 0..255 (one length byte per record)");
    if (!k_supported(k)) return fail(ctx, GK_E_UNSUPPORTED_K, "k=" + std::to_string(k) + " unsupported");
    const int P = d->world, slot = gk_skm_slot_bytes(k);
    if (d->slot != slot) {       // key width changed: the buffers were sized in other slots
        if (d->npending) return fail(ctx, GK_E_KLEN, "gk_dist_route_begin: a route for another key width is still waiting");
        GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        GK_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
        for (int i = 0; i < 2; i++) { if (d->d_sendbuf[i]) GK_HIP(ctx, hipFree(d->d_sendbuf[i])); d->d_sendbuf[i] = nullptr; d->send_cap[i] = 0; }
        if (d->d_recv) { GK_HIP(ctx, hipFree(d->d_recv)); d->d_recv = nullptr; }
        d->recv_records = 0;
        d->slot = slot;
    }
    const int b = d->cur;
    if (int rc = dist_grow(ctx, &d->d_sendbuf[b], &d->send_cap[b], std::max(route_want_records(k, P, nreads, read_len), d->send_cap[b]), slot)) return rc;
    if (int rc = skm_route_launch(ctx, ctx->copy_stream, d->d_route_cnt + b * SKM_COUNT_WORDS, d->h_route_cnt + b * SKM_COUNT_WORDS, k, dev_records, nreads,
                                  read_len, P, d->d_sendbuf[b], d->send_cap[b])) return rc;
    GK_HIP(ctx, hipEventRecord(d->route_done[b], ctx->copy_stream));
    d->route[b].k = k; d->route[b].read_len = read_len; d->route[b].records = dev_records; d->route[b].nreads = nreads;
    d->npending++;
    d->cur ^= 1;
    return GK_OK;
}

// The rest of the batch begun by gk_dist_route_begin: wait for its route, exchange counts and records, count what arrived.
int gk_dist_count_routed(gk_dist *d, gk_map *local, uint64_t *occurrences_sent, uint64_t *occurrences_owned) {
    if (int rc = dist_check(d)) return rc;
    gk_ctx *ctx = d->ctx;
    if (occurrences_sent) *occurrences_sent = 0;
    if (occurrences_owned) *occurrences_owned = 0;
    if (!d->npending) return fail(ctx, GK_E_STATE, "gk_dist_count_routed: no route was begun (gk_dist_route_begin)");
    const int b = (d->cur - d->npending) & 1;
    const gk_dist::Route rt = d->route[b];
    if (!local || local->ctx != ctx) return fail(ctx, GK_E_INVALID, "gk_dist_count_routed: the local map must live on the handle's context");
    if (local->k != rt.k) return fail(ctx, GK_E_KLEN, "gk_dist_count_routed: the route was begun for another k");
    d->npending--;
    const int k = rt.k, P = d->world, slot = d->slot;
    unsigned long long *d_rc = d->d_route_cnt + b * SKM_COUNT_WORDS, *h_rc = d->h_route_cnt + b * SKM_COUNT_WORDS;
    const double t0 = now_ms();
    // ---- 1. the route's counters (it ran on the second stream, possibly long ago)
    uint64_t recs[64], kmers[64];
    GK_HIP(ctx, hipEventSynchronize(d->route_done[b]));        // this route only: the next one may already be queued behind it
    int rrc = skm_route_finish(ctx, h_rc, rt.nreads && rt.read_len >= k, P, d->send_cap[b], recs, kmers);
    for (int attempt = 0; rrc == GK_E_CAPACITY && attempt < 4; attempt++) {       // a region was too small: route again, in place, bigger
        u64 worst = 0;
        for (int p = 0; p < P; p++) worst = std::max<u64>(worst, recs[p]);
        const u64 want = std::max<u64>(d->send_cap[b] * 2, (worst + worst / 8 + 1024) * P);
        if (int rc = dist_grow(ctx, &d->d_sendbuf[b], &d->send_cap[b], want, slot)) return rc;
        if (int rc = skm_route_launch(ctx, ctx->copy_stream, d_rc, h_rc, k, rt.records, rt.nreads, rt.read_len, P, d->d_sendbuf[b], d->send_cap[b])) return rc;
        GK_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
        rrc = skm_route_finish(ctx, h_rc, true, P, d->send_cap[b], recs, kmers);
    }
    if (rrc) return rrc;
    const double t1 = now_ms();
    // ---- 2. counts: (records, k-mers) for every peer, one tiny all-to-all; the sizes then reach the host
    const u64 region = d->send_cap[b] / (u64)P;
    u64 sent = 0;
    for (int p = 0; p < P; p++) { d->h_cnt[2 * p] = recs[p]; d->h_cnt[2 * p + 1] = kmers[p]; sent += kmers[p]; }
    unsigned long long *d_in = d->d_cnt, *d_out = d->d_cnt + 2 * 64;
    GK_HIP(ctx, hipMemcpyAsync(d_in, d->h_cnt, 2 * P * 8, hipMemcpyHostToDevice, ctx->stream));
    Rccl *r = rccl();
    GK_NCCL(ctx, r->GroupStart());
    for (int p = 0; p < P; p++) {
        GK_NCCL(ctx, r->Send(d_in + 2 * p, 2, ncclUint64, p, d->comm, ctx->stream));
        GK_NCCL(ctx, r->Recv(d_out + 2 * p, 2, ncclUint64, p, d->comm, ctx->stream));
    }
    GK_NCCL(ctx, r->GroupEnd());
    unsigned long long *h_out = d->h_cnt + 2 * 64;
    GK_HIP(ctx, hipMemcpyAsync(h_out, d_out, 2 * P * 8, hipMemcpyDeviceToHost, ctx->stream));
    GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    u64 nrec_in = 0, nkm_in = 0;
    for (int p = 0; p < P; p++) { nrec_in += h_out[2 * p]; nkm_in += h_out[2 * p + 1]; }
    // ---- 3. payload: region p -> rank p, straight out of the send buffer; arrivals land back to back
    if (int rc = dist_grow(ctx, &d->d_recv, &d->recv_records, std::max<u64>(nrec_in, d->recv_records), slot)) return rc;
    GK_NCCL(ctx, r->GroupStart());
    u64 roff = 0;
    for (int p = 0; p < P; p++) {
        if (recs[p]) GK_NCCL(ctx, r->Send(d->d_sendbuf[b] + (u64)p * region * slot, (size_t)recs[p] * slot, ncclUint8, p, d->comm, ctx->stream));
        if (h_out[2 * p]) GK_NCCL(ctx, r->Recv(d->d_recv + roff * slot, (size_t)h_out[2 * p] * slot, ncclUint8, p, d->comm, ctx->stream));
        roff += h_out[2 * p];
    }
    GK_NCCL(ctx, r->GroupEnd());
    const double t2 = now_ms();
    // ---- 4. the owner counts what it received: the records ARE short reads (stream-ordered behind the receives)
    uint64_t occ = 0;
    if (nrec_in) { if (int rc = gk_map_count_superkmers_dev(local, d->d_recv, nrec_in, nkm_in, &occ)) return rc; }
    else GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const double t3 = now_ms();
    d->last_ms[0] = (float)(t1 - t0); d->last_ms[1] = (float)(t2 - t1); d->last_ms[2] = (float)(t3 - t2); d->last_ms[3] = (float)(t3 - t0);
    if (occurrences_sent) *occurrences_sent = sent;
    if (occurrences_owned) *occurrences_owned = occ;
    return GK_OK;
}

int gk_dist_count_reads_dev(gk_dist *d, gk_map *local, const void *dev_records, uint64_t nreads, int read_len,
                            uint64_t *occurrences_sent, uint64_t *occurrences_owned) {
    if (int rc = dist_check(d)) return rc;
    if (occurrences_sent) *occurrences_sent = 0;
    if (occurrences_owned) *occurrences_owned = 0;
    if (!local || local->ctx != d->ctx) return fail(d->ctx, GK_E_INVALID, "gk_dist_count_reads_dev: the local map must live on the handle's context");
    if (int rc = gk_dist_route_begin(d, local->k, dev_records, nreads, read_len)) return rc;
    return gk_dist_count_routed(d, local, occurrences_sent, occurrences_owned);
}

int gk_dist_last_ms(gk_dist *d, float *ms4) {
    if (!d || !ms4) return fail(nullptr, GK_E_INVALID, "gk_dist_last_ms: null argument");
    for (int i = 0; i < 4; i++) ms4[i] = d->last_ms[i];
    return GK_OK;
}

int gk_dist_gather_map(gk_dist *d, gk_map *local, gk_map **full) {
    if (int rc = dist_check(d)) return rc;
    gk_ctx *ctx = d->ctx;
    if (!local || !full || local->ctx != ctx) return fail(ctx, GK_E_INVALID, "gk_dist_gather_map: bad argument");
    *full = nullptr;
    if (int rc = map_materialize(local)) return rc;
    const int P = d->world, W = local->W;
    Rccl *r = rccl();
    // sizes
    unsigned long long mine = local->size;
    GK_HIP(ctx, hipMemcpyAsync(d->d_cnt + 64, &mine, 8, hipMemcpyHostToDevice, ctx->stream));
    GK_NCCL(ctx, r->AllGather(d->d_cnt + 64, d->d_cnt, 1, ncclUint64, d->comm, ctx->stream));
    GK_HIP(ctx, hipMemcpyAsync(d->h_cnt, d->d_cnt, P * 8, hipMemcpyDeviceToHost, ctx->stream));
    GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    u64 total = 0, my_off = 0;
    for (int p = 0; p < P; p++) { if (p == d->rank) my_off = total; total += d->h_cnt[p]; }
    if (d->h_cnt[d->rank] != mine) return fail(ctx, GK_E_COMM, "gk_dist_gather_map: size exchange is inconsistent");
    // every rank's live (key, count) lands in one array, this rank's own part written in place by the export
    u64 *d_keys = nullptr;
    i32 *d_cnt = nullptr;
    unsigned long long *d_cursor = nullptr;
    auto done = [&](int code) {
        if (d_keys) (void)hipFree(d_keys);
        if (d_cnt) (void)hipFree(d_cnt);
        if (d_cursor) (void)hipFree(d_cursor);
        return code;
    };
    hipError_t e = hipMalloc((void **)&d_keys, std::max<u64>(total, 1) * 8 * W);
    if (e == hipSuccess) e = hipMalloc((void **)&d_cnt, std::max<u64>(total, 1) * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&d_cursor, 8);
    if (e == hipSuccess) e = hipMemsetAsync(d_cursor, 0, 8, ctx->stream);
    if (e != hipSuccess) return done(hip_fail(ctx, e, "gk_dist_gather_map: alloc"));
    if (mine) {
        const int grid = (int)std::min<u64>(std::max<u64>((local->capacity + BLOCK - 1) / BLOCK, 1), (u64)ctx->cu_count * 8);
        if (W == 1)
            hipLaunchKernelGGL(k_export_packed<1>, dim3(grid), dim3(BLOCK), 0, ctx->stream, (const Slot<1> *)local->slots, local->capacity, 0u,
                               d_keys + my_off, d_cnt + my_off, d_cursor);
        else
            hipLaunchKernelGGL(k_export_packed<2>, dim3(grid), dim3(BLOCK), 0, ctx->stream, (const Slot<2> *)local->slots, local->capacity,
                               local->k == 64 ? 1u : 0u, d_keys + 2 * my_off, d_cnt + my_off, d_cursor);
        e = hipGetLastError();
        if (e != hipSuccess) return done(hip_fail(ctx, e, "gk_dist_gather_map: export"));
    }
    // all-gather-v: my part to every peer, every peer's part to its offset (device to device over xGMI)
    int grc = r->GroupStart();
    u64 off = 0;
    for (int p = 0; p < P && grc == ncclSuccess; p++) {
        const u64 n = d->h_cnt[p];
        if (p != d->rank) {
            if (mine) grc = r->Send(d_keys + my_off * W, (size_t)mine * W, ncclUint64, p, d->comm, ctx->stream);
            if (mine && grc == ncclSuccess) grc = r->Send(d_cnt + my_off, (size_t)mine * 4, ncclUint8, p, d->comm, ctx->stream);
            if (n && grc == ncclSuccess) grc = r->Recv(d_keys + off * W, (size_t)n * W, ncclUint64, p, d->comm, ctx->stream);
            if (n && grc == ncclSuccess) grc = r->Recv(d_cnt + off, (size_t)n * 4, ncclUint8, p, d->comm, ctx->stream);
        }
        off += n;
    }
    if (grc == ncclSuccess) grc = r->GroupEnd();
    if (grc != ncclSuccess) return done(fail(ctx, GK_E_COMM, std::string("gk_dist_gather_map: ") + (r->GetErrorString ? r->GetErrorString(grc) : "RCCL error")));
    // one table holding every partition's keys (each key has exactly one owner: nothing merges)
    gk_map *m = nullptr;
    if (int rc = gk_map_create(ctx, local->k, total, &m)) return done(rc);
    int rc = total ? map_add_counted_keys_dev(m, d_keys, d_cnt, total) : GK_OK;
    if (rc == GK_OK) rc = map_sync_counters(m);
    if (rc != GK_OK) { gk_map_destroy(m); return done(rc); }
    m->dirty = local->dirty;
    *full = m;
    return done(GK_OK);
}

}  // extern "C"
