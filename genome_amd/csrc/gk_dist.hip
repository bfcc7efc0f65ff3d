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
#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "gk_dist.h"
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

static void helper_main(gk_dist *d) {
    (void)hipSetDevice(d->ctx->device);
    std::unique_lock<std::mutex> lk(d->wmu);
    for (;;) {
        d->wcv.wait(lk, [&]() { return d->job_pending || d->quit; });
        if (d->quit) return;
        std::function<void()> f = std::move(d->job);
        d->job_pending = false; d->job_running = true;
        lk.unlock();
        f();
        lk.lock();
        d->job_running = false;
        d->wcv.notify_all();
    }
}
static void helper_submit(gk_dist *d, std::function<void()> f) {
    std::unique_lock<std::mutex> lk(d->wmu);
    if (!d->worker.joinable()) d->worker = std::thread(helper_main, d);
    d->job = std::move(f);
    d->job_pending = true;
    d->wcv.notify_all();
}
static void helper_wait(gk_dist *d) {
    std::unique_lock<std::mutex> lk(d->wmu);
    d->wcv.wait(lk, [&]() { return !d->job_pending && !d->job_running; });
}
static void helper_stop(gk_dist *d) {
    {
        std::unique_lock<std::mutex> lk(d->wmu);
        d->wcv.wait(lk, [&]() { return !d->job_pending && !d->job_running; });
        d->quit = true;
        d->wcv.notify_all();
    }
    if (d->worker.joinable()) d->worker.join();
}
// buffers the helper thread replaced: their frees wait for the owner thread (pool_free synchronises the context's streams)
static void drain_garbage(gk_dist *d) {
    gk_ctx *ctx = d->ctx;
    std::vector<void *> g;
    { std::lock_guard<std::mutex> lk(d->wmu); g.swap(d->garbage); }
    for (void *p : g) (void)hipFree(p);
}

static thread_local const gk_dist *tl_dist = nullptr;       // whose transport explains an error code (set by dist_check)
static std::string comm_error_text(int code) {
    if (tl_dist && tl_dist->xport) return tl_dist->xport->error_text(code);
    Rccl *r = rccl();
    return r->GetErrorString ? r->GetErrorString(code) : "RCCL error";
}
#define GK_NCCL(ctx, call)                                                                                        \
    do {                                                                                                          \
        int r__ = (call);                                                                                         \
        if (r__ != ncclSuccess)                                                                                   \
            return gk::fail((ctx), GK_E_COMM, std::string(#call) + ": " + comm_error_text(r__)); \
    } while (0)

// ---- the RCCL transport -----------------------------------------------------------------------------------------------------
namespace {
int rccl_group_start(gk_dist *) { return rccl()->GroupStart(); }
int rccl_send(gk_dist *d, const void *p, size_t count, int dt, int peer, hipStream_t st) { return rccl()->Send(p, count, dt, peer, (ncclComm_t)d->comm, st); }
int rccl_recv(gk_dist *d, void *p, size_t count, int dt, int peer, hipStream_t st) { return rccl()->Recv(p, count, dt, peer, (ncclComm_t)d->comm, st); }
int rccl_group_end(gk_dist *) { return rccl()->GroupEnd(); }
int rccl_all_reduce(gk_dist *d, const void *in, void *out, size_t n, int dt, int op, hipStream_t st) { return rccl()->AllReduce(in, out, n, dt, op, (ncclComm_t)d->comm, st); }
int rccl_all_gather(gk_dist *d, const void *in, void *out, size_t n, int dt, hipStream_t st) { return rccl()->AllGather(in, out, n, dt, (ncclComm_t)d->comm, st); }
std::string rccl_error_text(int code) { Rccl *r = rccl(); return r->GetErrorString ? r->GetErrorString(code) : "RCCL error"; }
void rccl_close(gk_dist *d) { if (d->comm && rccl()->CommDestroy) (void)rccl()->CommDestroy((ncclComm_t)d->comm); d->comm = nullptr; }
const Transport RCCL_TRANSPORT = {rccl_group_start, rccl_send, rccl_recv, rccl_group_end, rccl_all_reduce, rccl_all_gather, rccl_error_text, rccl_close};
}  // namespace
static int xGroupStart(gk_dist *d) { return d->xport->group_start(d); }
static int xSend(gk_dist *d, const void *p, size_t count, int dt, int peer, hipStream_t st) { return d->xport->send(d, p, count, dt, peer, st); }
static int xRecv(gk_dist *d, void *p, size_t count, int dt, int peer, hipStream_t st) { return d->xport->recv(d, p, count, dt, peer, st); }
static int xGroupEnd(gk_dist *d) { return d->xport->group_end(d); }
static int xAllReduce(gk_dist *d, const void *in, void *out, size_t n, int dt, int op, hipStream_t st) { return d->xport->all_reduce(d, in, out, n, dt, op, st); }
static int xAllGather(gk_dist *d, const void *in, void *out, size_t n_per_rank, int dt, hipStream_t st) { return d->xport->all_gather(d, in, out, n_per_rank, dt, st); }

static int dist_check(const gk_dist *d) {
    if (!d || !d->ctx || !d->xport) return fail(nullptr, GK_E_INVALID, "null or closed gk_dist handle");
    tl_dist = d;
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

}  // extern "C"
namespace gk {
int dist_create_common(gk_ctx *ctx, int rank, int world, gk_dist **out) {
    *out = nullptr;
    gk_dist *d = new gk_dist();
    d->ctx = ctx; d->rank = rank; d->world = world;
    hipError_t e = hipMalloc((void **)&d->d_cnt, 8 * 64 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipHostMalloc((void **)&d->h_cnt, 8 * 64 * sizeof(unsigned long long), hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc((void **)&d->d_route_cnt, gk_dist::NROUTE * SKM_COUNT_WORDS * sizeof(unsigned long long));
    for (int i = 0; i < gk_dist::NROUTE && e == hipSuccess; i++) e = hipEventCreateWithFlags(&d->route_done[i], hipEventDisableTiming);
    for (int i = 0; i < gk_dist::NROUTE && e == hipSuccess; i++) e = hipEventCreateWithFlags(&d->exch_done[i], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&d->join, hipEventDisableTiming);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&d->comm_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipHostMalloc((void **)&d->h_route_cnt, gk_dist::NROUTE * SKM_COUNT_WORDS * sizeof(unsigned long long), hipHostMallocDefault);
    if (e != hipSuccess) { int c = hip_fail(ctx, e, "gk_dist_create"); gk_dist_destroy(d); return c; }
    *out = d;
    return GK_OK;
}
}  // namespace gk
extern "C" {

int gk_dist_create(gk_ctx *ctx, int rank, int world, const void *id128, gk_dist **out) {
    if (!ctx || !out || !id128) return fail(ctx, GK_E_INVALID, "gk_dist_create: null argument");
    *out = nullptr;
    if (world < 1 || world > 64 || rank < 0 || rank >= world) return fail(ctx, GK_E_INVALID, "gk_dist_create: need 0 <= rank < world <= 64");
    Rccl *r = rccl();
    if (!r->why.empty()) return fail(ctx, GK_E_COMM, r->why);
    GK_HIP(ctx, hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(id.internal, id128, 128);
    ncclComm_t comm = nullptr;
    const int rc = r->CommInitRank(&comm, world, id, rank);
    if (rc != ncclSuccess) return fail(ctx, GK_E_COMM, std::string("ncclCommInitRank: ") + (r->GetErrorString ? r->GetErrorString(rc) : "error"));
    gk_dist *d = nullptr;
    if (int c = dist_create_common(ctx, rank, world, &d)) { if (r->CommDestroy) (void)r->CommDestroy(comm); return c; }
    d->xport = &RCCL_TRANSPORT;
    d->comm = comm;
    *out = d;
    return GK_OK;
}

void gk_dist_destroy(gk_dist *d) {
    if (!d) return;
    gk_ctx *ctx = d->ctx;
    helper_stop(d);
    if (ctx) { (void)hipSetDevice(ctx->device); drain_garbage(d); }
    if (d->ctx) { (void)hipSetDevice(d->ctx->device); (void)hipStreamSynchronize(d->ctx->stream); }
    if (d->comm_stream) (void)hipStreamSynchronize(d->comm_stream);
    if (d->xport && d->xport->close) d->xport->close(d);
    if (d->ctx && d->ctx->copy_stream) (void)hipStreamSynchronize(d->ctx->copy_stream);
    for (int i = 0; i < gk_dist::NROUTE; i++) if (d->d_sendbuf[i]) (void)hipFree(d->d_sendbuf[i]);
    for (int i = 0; i < 2; i++) if (d->d_recv[i]) (void)hipFree(d->d_recv[i]);
    for (int i = 0; i < gk_dist::NROUTE; i++) if (d->route_done[i]) (void)hipEventDestroy(d->route_done[i]);
    for (int i = 0; i < gk_dist::NROUTE; i++) if (d->exch_done[i]) (void)hipEventDestroy(d->exch_done[i]);
    if (d->join) (void)hipEventDestroy(d->join);
    if (d->comm_stream) (void)hipStreamDestroy(d->comm_stream);
    if (d->d_route_cnt) (void)hipFree(d->d_route_cnt);
    if (d->h_route_cnt) (void)hipHostFree(d->h_route_cnt);
    if (d->d_cnt) (void)hipFree(d->d_cnt);
    if (d->h_cnt) (void)hipHostFree(d->h_cnt);
    delete d;
}

int gk_dist_rank(const gk_dist *d) { return d ? d->rank : -1; }
int gk_dist_world(const gk_dist *d) { return d ? d->world : 0; }

// The collectives below run on the context's stream; a batch whose exchange was posted ahead (gk_dist_count_routed) may still
// be on the wire on the communication stream, and two operations of one communicator must not run side by side.
static int dist_quiesce(gk_dist *d) {
    GK_HIP(d->ctx, hipStreamSynchronize(d->comm_stream));
    return GK_OK;
}

int gk_dist_barrier(gk_dist *d) {
    if (int rc = dist_check(d)) return rc;
    gk_ctx *ctx = d->ctx;
    if (int rc = dist_quiesce(d)) return rc;
    GK_HIP(ctx, hipMemsetAsync(d->d_cnt, 0, 8, ctx->stream));
    GK_NCCL(ctx, xAllReduce(d, d->d_cnt, d->d_cnt, 1, ncclUint64, ncclSum, ctx->stream));
    GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GK_OK;
}

int gk_dist_allreduce_f64(gk_dist *d, double *values, int n, int op_max) {
    if (int rc = dist_check(d)) return rc;
    gk_ctx *ctx = d->ctx;
    if (!values || n < 1 || n > 32) return fail(ctx, GK_E_INVALID, "gk_dist_allreduce_f64: 1..32 values");
    if (int rc = dist_quiesce(d)) return rc;
    double *dv = reinterpret_cast<double *>(d->d_cnt);
    GK_HIP(ctx, hipMemcpyAsync(dv, values, n * 8, hipMemcpyHostToDevice, ctx->stream));
    GK_NCCL(ctx, xAllReduce(d, dv, dv, (size_t)n, ncclFloat64, op_max ? ncclMax : ncclSum, ctx->stream));
    GK_HIP(ctx, hipMemcpyAsync(values, dv, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GK_OK;
}

int gk_dist_size(gk_dist *d, gk_map *local, uint64_t *total) {
    if (int rc = dist_check(d)) return rc;
    gk_ctx *ctx = d->ctx;
    if (!local || !total) return fail(ctx, GK_E_INVALID, "gk_dist_size: null argument");
    if (int rc = dist_quiesce(d)) return rc;
    unsigned long long v = local->size;
    GK_HIP(ctx, hipMemcpyAsync(d->d_cnt, &v, 8, hipMemcpyHostToDevice, ctx->stream));
    GK_NCCL(ctx, xAllReduce(d, d->d_cnt, d->d_cnt, 1, ncclUint64, ncclSum, ctx->stream));
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

// Route this rank's reads for a later gk_dist_count_routed on the context's second stream and return at once: the routing
// kernel (0.6 ms per 10^6 reads, issue-bound) then overlaps whatever the main stream is doing — in a streaming loop, the
// owner pipeline of an earlier batch.  The records buffer must stay valid until the batch's gk_dist_count_routed returns.
int gk_dist_route_begin(gk_dist *d, int k, const void *dev_records, uint64_t nreads, int read_len) {
    if (int rc = dist_check(d)) return rc;
    gk_ctx *ctx = d->ctx;
    constexpr int NR = gk_dist::NROUTE;
    drain_garbage(d);
    if (d->npending >= NR) return fail(ctx, GK_E_STATE, "gk_dist_route_begin: three routes are already waiting (gk_dist_count_routed consumes one)");
    if (!dev_records && nreads) return fail(ctx, GK_E_INVALID, "gk_dist_route_begin: null records");
    if (read_len < 0 || read_len > 255) return fail(ctx, GK_E_FORMAT, "read_len must be 0..255 (one length byte per record)");
    if (!k_supported(k)) return fail(ctx, GK_E_UNSUPPORTED_K, "k=" + std::to_string(k) + " unsupported");
    const int P = d->world, slot = gk_skm_slot_bytes(k);
    if (d->slot != slot) {       // key width changed: the buffers were sized in other slots
        if (d->npending) return fail(ctx, GK_E_KLEN, "gk_dist_route_begin: a route for another key width is still waiting");
        GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        GK_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
        GK_HIP(ctx, hipStreamSynchronize(d->comm_stream));
        for (int i = 0; i < NR; i++) { if (d->d_sendbuf[i]) GK_HIP(ctx, hipFree(d->d_sendbuf[i])); d->d_sendbuf[i] = nullptr; d->send_cap[i] = 0; }
        for (int i = 0; i < 2; i++) { if (d->d_recv[i]) GK_HIP(ctx, hipFree(d->d_recv[i])); d->d_recv[i] = nullptr; d->recv_records[i] = 0; }
        d->slot = slot;
    }
    const int b = (d->head + d->npending) % NR;
    // (the buffer's previous batch was counted before this slot could come round again: its records have left)
    u64 want = std::max(route_want_records(k, P, nreads, read_len), d->send_cap[b]);
    if (ctx->hook_dist_small_send > 0) {      // test hook: THIS route gets a send buffer of so many records — the in-place re-route must repair it
        GK_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
        if (d->d_sendbuf[b]) { GK_HIP(ctx, hipFree(d->d_sendbuf[b])); d->d_sendbuf[b] = nullptr; d->send_cap[b] = 0; }
        want = (u64)std::max(ctx->hook_dist_small_send, P);
        ctx->hook_dist_small_send = 0;
    }
    gk_dist::Route &rt = d->route[b];
    rt = gk_dist::Route();
    rt.k = k; rt.read_len = read_len; rt.records = dev_records; rt.nreads = nreads;
    // A failure from here on is THIS RANK'S ALONE, and its peers are going to exchange this batch: the batch is begun all the
    // same, carrying the failure, and gk_dist_count_routed tells everybody (status word of the counts exchange).
    int rc = dist_grow(ctx, &d->d_sendbuf[b], &d->send_cap[b], want, slot);
    ctx->copy_other_pending = true;
    if (!rc) rc = skm_route_launch(ctx, ctx->copy_stream, d->d_route_cnt + b * SKM_COUNT_WORDS, d->h_route_cnt + b * SKM_COUNT_WORDS, k, dev_records, nreads,
                                   read_len, P, d->d_sendbuf[b], d->send_cap[b]);
    if (rc) { rt.settled = true; rt.local_rc = rc; rt.local_err = ctx->err; (void)hipGetLastError(); }
    GK_HIP(ctx, hipEventRecord(d->route_done[b], ctx->copy_stream));
    d->npending++;
    return GK_OK;
}

// Owner thread: wait for route slot b's kernel, take the route again (bigger, in place) while a region was too small, and
// fix this rank's verdict on its half of the batch.  ms += the wait.
static void route_settle(gk_dist *d, int b, float *ms) {
    gk_ctx *ctx = d->ctx;
    gk_dist::Route &rt = d->route[b];
    if (rt.settled) return;
    const double t0 = now_ms();
    const int k = rt.k, P = d->world, slot = d->slot;
    unsigned long long *d_rc = d->d_route_cnt + b * SKM_COUNT_WORDS, *h_rc = d->h_route_cnt + b * SKM_COUNT_WORDS;
    int rrc = GK_OK;
    hipError_t e = hipEventSynchronize(d->route_done[b]);            // this route only: later ones may already be queued behind it
    if (e != hipSuccess) rrc = hip_fail(ctx, e, "gk_dist: waiting for the route");
    if (!rrc) rrc = skm_route_finish(ctx, h_rc, rt.nreads && rt.read_len >= k, P, d->send_cap[b], rt.recs, rt.kmers);
    for (int attempt = 0; rrc == GK_E_CAPACITY && attempt < 4; attempt++) {       // a region was too small: route again, in place, bigger
        u64 worst = 0;
        for (int p = 0; p < P; p++) worst = std::max<u64>(worst, rt.recs[p]);
        const u64 want = std::max<u64>(d->send_cap[b] * 2, (worst + worst / 8 + 1024) * P);
        if ((e = hipStreamSynchronize(ctx->copy_stream)) != hipSuccess) { rrc = hip_fail(ctx, e, "gk_dist: re-route"); break; }   // (later routes share the stream)
        if ((rrc = dist_grow(ctx, &d->d_sendbuf[b], &d->send_cap[b], want, slot)) != GK_OK) break;
        ctx->copy_other_pending = true;
        if ((rrc = skm_route_launch(ctx, ctx->copy_stream, d_rc, h_rc, k, rt.records, rt.nreads, rt.read_len, P, d->d_sendbuf[b], d->send_cap[b])) != GK_OK) break;
        if ((e = hipStreamSynchronize(ctx->copy_stream)) != hipSuccess) { rrc = hip_fail(ctx, e, "gk_dist: re-route"); break; }
        rrc = skm_route_finish(ctx, h_rc, true, P, d->send_cap[b], rt.recs, rt.kmers);
    }
    if (!rrc && ctx->hook_dist_fail) {        // test hook: this rank alone fails this batch, as an allocation or a hopeless route would
        rrc = fail(ctx, ctx->hook_dist_fail, "injected failure (test_dist_fail_exchange)");
        ctx->hook_dist_fail = 0;
    }
    rt.local_rc = rrc;
    if (rrc) rt.local_err = ctx->err;
    rt.settled = true;
    *ms += (float)(now_ms() - t0);
}

// Counts and records of route slot b (settled) over RCCL, on the communication stream — on the owner thread or on the
// helper.  Touches neither the context's error string nor its streams nor the pool's frees.  Three steps, and every rank
// takes the same ones whatever happens to it locally:
//   1. counts: (records, k-mers, STATUS) to every peer — a rank whose route failed says so here and sends nothing; if any
//      status is set, every rank drops the batch after this step;
//   2. room for what is coming, then one max-reduction of "I could not make room": any rank -> every rank drops the batch;
//   3. payload: region p -> rank p straight out of the send buffer, arrivals back to back in the receive buffer that is not
//      being counted.  exch_done[b] fires when they have arrived.
// Returns 0 with rt.exchanged set, or the code the batch's count_routed will report (rt.error / rt.error_text).
static int dist_exchange(gk_dist *d, int b, float *ms) {
    gk_ctx *ctx = d->ctx;
    gk_dist::Route &rt = d->route[b];
    const int P = d->world, slot = d->slot;
    const double t1 = now_ms();
    hipStream_t cs = d->comm_stream;
    auto give_up = [&](int code, const std::string &text) { rt.error = code; rt.error_text = text; *ms += (float)(now_ms() - t1); return code; };
    auto hip_text = [](hipError_t e, const char *what) { (void)hipGetLastError(); return std::string(what) + ": " + hipGetErrorString(e); };
    // ---- 1. counts + status
    const u64 region = d->send_cap[b] / (u64)P;
    u64 sent = 0;
    unsigned long long *h_in = d->h_cnt, *h_out = d->h_cnt + 3 * 64, *d_in = d->d_cnt, *d_out = d->d_cnt + 3 * 64;
    for (int p = 0; p < P; p++) {
        h_in[3 * p] = rt.local_rc ? 0 : rt.recs[p];
        h_in[3 * p + 1] = rt.local_rc ? 0 : rt.kmers[p];
        h_in[3 * p + 2] = rt.local_rc ? (unsigned long long)(-rt.local_rc) : 0ull;
        sent += h_in[3 * p + 1];
    }
    hipError_t e = hipMemcpyAsync(d_in, h_in, 3 * P * 8, hipMemcpyHostToDevice, cs);
    if (e != hipSuccess) return give_up(GK_E_HIP, hip_text(e, "gk_dist: counts upload") + " (before anything was posted: the peers will wait; destroy the communicator)");
    int grc = xGroupStart(d);
    for (int p = 0; p < P && grc == ncclSuccess; p++) {
        grc = xSend(d, d_in + 3 * p, 3, ncclUint64, p, cs);
        if (grc == ncclSuccess) grc = xRecv(d, d_out + 3 * p, 3, ncclUint64, p, cs);
    }
    { const int gend = xGroupEnd(d); if (grc == ncclSuccess) grc = gend; }          // (a group that was opened is always closed)
    if (grc != ncclSuccess) return give_up(GK_E_COMM, "counts exchange: " + comm_error_text(grc));
    e = hipMemcpyAsync(h_out, d_out, 3 * P * 8, hipMemcpyDeviceToHost, cs);
    if (e == hipSuccess) e = hipStreamSynchronize(cs);                              // (also: the previous batch's records have arrived)
    if (e != hipSuccess) return give_up(GK_E_HIP, hip_text(e, "gk_dist: counts download"));
    u64 nrec_in = 0, nkm_in = 0;
    int bad_rank = -1;
    for (int p = 0; p < P; p++) {
        nrec_in += h_out[3 * p]; nkm_in += h_out[3 * p + 1];
        if (h_out[3 * p + 2] && bad_rank < 0) bad_rank = p;
    }
    if (bad_rank >= 0) {
        std::string text = "rank " + std::to_string(bad_rank) + " could not route its share of this batch (status " + std::to_string(-(long long)h_out[3 * bad_rank + 2]) +
                           "): every rank dropped the batch";
        if (rt.local_rc) text += "; this rank: " + rt.local_err;
        return give_up(GK_E_COMM, text);
    }
    // ---- 2. room for the arrivals (the buffer that is NOT being counted: the batch counted two exchanges ago has returned)
    const int rb = (int)(d->nexchanged & 1);
    unsigned long long noroom = 0;
    std::string noroom_text;
    if (nrec_in > d->recv_records[rb] || !d->d_recv[rb]) {
        const u64 want = std::max<u64>(nrec_in + nrec_in / 8, 1024);
        uint8_t *nb = nullptr;
        e = hipMalloc((void **)&nb, want * (u64)slot + 64);                         // (the pool: a mutex, no stream is waited for)
        if (e == hipSuccess) {
            if (d->d_recv[rb]) { std::lock_guard<std::mutex> lk(d->wmu); d->garbage.push_back(d->d_recv[rb]); }
            d->d_recv[rb] = nb; d->recv_records[rb] = want;
        } else { noroom = 1; noroom_text = hip_text(e, "gk_dist: receive buffer"); }
    }
    unsigned long long *d_flag = d->d_cnt + 6 * 64, *h_flag = d->h_cnt + 6 * 64;
    h_flag[0] = noroom;
    e = hipMemcpyAsync(d_flag, h_flag, 8, hipMemcpyHostToDevice, cs);
    if (e != hipSuccess) return give_up(GK_E_HIP, hip_text(e, "gk_dist: ready flag"));
    grc = xAllReduce(d, d_flag, d_flag + 1, 1, ncclUint64, ncclMax, cs);
    if (grc != ncclSuccess) return give_up(GK_E_COMM, "ready reduction: " + comm_error_text(grc));
    e = hipMemcpyAsync(h_flag + 1, d_flag + 1, 8, hipMemcpyDeviceToHost, cs);
    if (e == hipSuccess) e = hipStreamSynchronize(cs);
    if (e != hipSuccess) return give_up(GK_E_HIP, hip_text(e, "gk_dist: ready flag"));
    if (h_flag[1]) return give_up(GK_E_COMM, "a rank had no room for the records addressed to it: every rank dropped the batch" + (noroom ? "; this rank: " + noroom_text : std::string()));
    // ---- 3. payload
    grc = xGroupStart(d);
    u64 roff = 0;
    for (int p = 0; p < P && grc == ncclSuccess; p++) {
        if (rt.recs[p]) grc = xSend(d, d->d_sendbuf[b] + (u64)p * region * slot, (size_t)rt.recs[p] * slot, ncclUint8, p, cs);
        if (h_out[3 * p] && grc == ncclSuccess) grc = xRecv(d, d->d_recv[rb] + roff * slot, (size_t)h_out[3 * p] * slot, ncclUint8, p, cs);
        roff += h_out[3 * p];
    }
    { const int gend = xGroupEnd(d); if (grc == ncclSuccess) grc = gend; }
    if (grc != ncclSuccess) return give_up(GK_E_COMM, "record exchange: " + comm_error_text(grc));
    e = hipEventRecord(d->exch_done[b], cs);
    if (e != hipSuccess) return give_up(GK_E_HIP, hip_text(e, "gk_dist: exchange event"));
    rt.exchanged = true; rt.rbuf = rb; rt.nrec_in = nrec_in; rt.nkm_in = nkm_in; rt.sent = sent;
    d->nexchanged++;
    *ms += (float)(now_ms() - t1);
    return GK_OK;
}

// The rest of the oldest batch begun by gk_dist_route_begin: exchange counts and records (unless that was done ahead), count
// what arrived.  With THREE batches begun, the exchange of the next one is posted first, so that its records travel while
// this batch is being counted (the rule depends on the number of begun batches only, never on timing: every rank issues
// the same sequence of RCCL operations).
int gk_dist_count_routed(gk_dist *d, gk_map *local, uint64_t *occurrences_sent, uint64_t *occurrences_owned) {
    if (int rc = dist_check(d)) return rc;
    gk_ctx *ctx = d->ctx;
    constexpr int NR = gk_dist::NROUTE;
    if (occurrences_sent) *occurrences_sent = 0;
    if (occurrences_owned) *occurrences_owned = 0;
    drain_garbage(d);
    if (!d->npending) return fail(ctx, GK_E_STATE, "gk_dist_count_routed: no route was begun (gk_dist_route_begin)");
    const int b = d->head;
    if (!local || local->ctx != ctx) return fail(ctx, GK_E_INVALID, "gk_dist_count_routed: the local map must live on the handle's context");
    if (local->k != d->route[b].k) return fail(ctx, GK_E_KLEN, "gk_dist_count_routed: the route was begun for another k");
    const double t0 = now_ms();
    float ms_route = 0, ms_exch = 0;
    auto drop_head = [&]() { d->head = (d->head + 1) % NR; d->npending--; };
    // (whatever this batch's own fate, the collective steps below are taken: the peers take them too)
    if (!d->route[b].exchanged && !d->route[b].error) {
        route_settle(d, b, &ms_route);
        (void)dist_exchange(d, b, &ms_exch);           // a failure is in route[b].error
    }
    // With three batches begun, the NEXT batch's exchange runs on the helper thread beside this batch's owner count: its host
    // side alone (RCCL group launches and two round trips for the sizes, 0.4-0.6 ms) would otherwise sit in front of the
    // owner pipeline's launches.  Its route is settled HERE first (a re-route allocates and waits for the context's second
    // stream: the owner thread's business).  The helper is the only thread that touches the communicator until it is waited for.
    bool helping = false;
    float hms = 0;
    if (d->npending >= NR && ctx->hook_dist_ahead != 0) {        // ("dist_exchange_ahead" = 0: the plain order, for A/B and as a fallback)
        const int b1 = (b + 1) % NR;
        if (!d->route[b1].exchanged && !d->route[b1].error) {
            route_settle(d, b1, &ms_route);
            helper_submit(d, [d, b1, &hms]() { (void)dist_exchange(d, b1, &hms); });
            helping = true;
        }
    }
    struct Join { gk_dist *d; bool on; ~Join() { if (on) helper_wait(d); } } join_helper{d, helping};
    const gk_dist::Route rt = d->route[b];
    drop_head();
    if (rt.error) return fail(ctx, rt.error, "gk_dist_count_routed: " + rt.error_text);     // (a batch that could not be exchanged is dropped — on every rank)
    const double t2 = now_ms();
    // ---- the owner counts what it received: the records ARE short reads (stream-ordered behind the receives)
    GK_HIP(ctx, hipStreamWaitEvent(ctx->stream, d->exch_done[b], 0));
    uint64_t occ = 0;
    if (rt.nrec_in) { if (int rc = gk_map_count_superkmers_dev(local, d->d_recv[rt.rbuf], rt.nrec_in, rt.nkm_in, &occ)) return rc; }
    else GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const double t3 = now_ms();
    if (helping) { helper_wait(d); join_helper.on = false; }
    const double t4 = now_ms();
    // {waiting for routes, exchanges done in front of the count, owner count, total}; [1] also carries what the helper's
    // exchange took beyond the owner count (0 when it was hidden completely)
    d->last_ms[0] = ms_route; d->last_ms[1] = ms_exch + (float)(t4 - t3); d->last_ms[2] = (float)(t3 - t2); d->last_ms[3] = (float)(t4 - t0);
    d->last_helper_ms = hms;
    if (occurrences_sent) *occurrences_sent = rt.sent;
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

// The whole k-mer set on every rank.  CHUNKED: every rank exports the live (key, count) of 2^25 of its slots at a time, the
// chunk's sizes go round (one word per rank — ~0 = "I failed": then EVERY rank gives up together, nobody is left waiting in a
// receive), the parts are exchanged and inserted, and the staging is reused: 0.67 GB to send and P x that to receive whatever
// the table's size (until round 3 the whole set was staged at 20 B per key — 62 GB at C5 — beside the table being built).
// The new table is sized the way the graph phase wants it (graph_table_load).
//
// CLASSIFIED form (gk_dist_gather_classified_map; SURVEY.md 8(e) "beyond counting", Graph.scala:320-329 through
// PartitionedDNAMap.mapReduce :55-58): before a chunk's keys travel, their owner classifies them — neighbours it owns itself
// are looked up in its own partition, the others are asked of THEIR owners (one all-to-all of canonical keys, one of answer
// bytes: k_dc_* in gk_graph.hip) — and the 8-bit (incoming, outcoming) mask travels with each key into the replica's annotation
// word.  gk_graph_build on the gathered table then derives terminal / secondary from the masks in one streaming pass instead of
// eight random lookups per key on EVERY rank: the classify's work is divided by the number of ranks.  A partition that holds
// verbatim non-canonical keys (dirty) on any rank turns the form off for everybody (plain gather).
static int dist_gather(gk_dist *d, gk_map *local, gk_map **full, bool classify) {
    if (int rc = dist_check(d)) return rc;
    gk_ctx *ctx = d->ctx;
    if (!local || !full || local->ctx != ctx) return fail(ctx, GK_E_INVALID, "gk_dist_gather_map: bad argument");
    *full = nullptr;
    if (int rc = dist_quiesce(d)) return rc;
    if (int rc = map_materialize(local)) return rc;
    const int P = d->world, W = local->W;
    // (the masks live in the annotation word: a partition that still counts in 12-byte slots is rebuilt first — a local step,
    //  whose failure is announced with the first chunk's word like any other)
    int pre_rc = GK_OK;
    std::string pre_err;
    if (classify) { pre_rc = map_to_graph_layout(local); if (pre_rc) pre_err = ctx->err; }
    constexpr u64 CHS = 1ull << 25;
    // live keys of every rank, and the number of chunks of the rank with the largest table
    // (one maximum over all ranks carries both: the chunk count of SOME rank in its low 40 bits and, above them, "some partition holds verbatim
    //  non-canonical keys" — the replica holds every partition's keys, so it is dirty if ANY of them is; the chunk count itself is a second maximum)
    unsigned long long mine[2] = {local->size, ((local->capacity + CHS - 1) / CHS) | (local->dirty ? 1ull << 40 : 0ull)};
    GK_HIP(ctx, hipMemcpyAsync(d->d_cnt + 2 * 64, mine, 16, hipMemcpyHostToDevice, ctx->stream));
    GK_NCCL(ctx, xAllGather(d, d->d_cnt + 2 * 64, d->d_cnt, 1, ncclUint64, ctx->stream));
    GK_NCCL(ctx, xAllReduce(d, d->d_cnt + 2 * 64 + 1, d->d_cnt + 64, 1, ncclUint64, ncclMax, ctx->stream));
    unsigned long long chunks_only = (local->capacity + CHS - 1) / CHS;
    GK_HIP(ctx, hipMemcpyAsync(d->d_cnt + 2 * 64 + 2, &chunks_only, 8, hipMemcpyHostToDevice, ctx->stream));
    GK_NCCL(ctx, xAllReduce(d, d->d_cnt + 2 * 64 + 2, d->d_cnt + 65, 1, ncclUint64, ncclMax, ctx->stream));
    GK_HIP(ctx, hipMemcpyAsync(d->h_cnt, d->d_cnt, (64 + 2) * 8, hipMemcpyDeviceToHost, ctx->stream));
    GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    u64 total = 0;
    const u64 nchunks = d->h_cnt[65];
    const bool any_dirty = (d->h_cnt[64] >> 40) != 0;
    classify = classify && !any_dirty;                 // (every rank takes the same decision: the word is a maximum over all of them)
    for (int p = 0; p < P; p++) total += d->h_cnt[p];
    if (d->h_cnt[d->rank] != mine[0]) return fail(ctx, GK_E_COMM, "gk_dist_gather_map: size exchange is inconsistent");
    gk_map *m = nullptr;
    u64 *d_send_k = nullptr, *d_recv_k = nullptr;
    i32 *d_send_c = nullptr, *d_recv_c = nullptr;
    unsigned long long *d_cur = nullptr;
    uint8_t *d_send_m = nullptr, *d_recv_m = nullptr;                  // classified form: the masks beside the keys
    unsigned long long *d_dc = nullptr;                                // [5][64]: counts to send | counts received | remote counts | region offsets | cursors
    u64 *d_qk = nullptr, *d_qref = nullptr, *d_rk = nullptr;           // queries out (keys, who asked), queries in
    uint8_t *d_ans_in = nullptr, *d_ans_out = nullptr;                 // answers to my queries, my answers to the others'
    u64 q_out_cap = 0, q_in_cap = 0;
    // every rank's chunk holds at most CHS keys: the receive staging is allocated ONCE, before anything is agreed — no allocation,
    // hence no local failure, between a chunk's size exchange and its sends and receives
    const u64 recv_cap = std::max<u64>(std::min<u64>(total, (u64)P * CHS), 1);
    auto done = [&](int code) {
        for (void *p : {(void *)d_send_k, (void *)d_send_c, (void *)d_recv_k, (void *)d_recv_c, (void *)d_cur, (void *)d_send_m, (void *)d_recv_m, (void *)d_dc,
                        (void *)d_qk, (void *)d_qref, (void *)d_rk, (void *)d_ans_in, (void *)d_ans_out}) if (p) (void)hipFree(p);
        if (code != GK_OK && m) { gk_map_destroy(m); m = nullptr; }
        return code;
    };
    // a local failure must not leave the peers in a receive: it is announced in the chunk's size word and everybody stops
    int my_rc = pre_rc;
    std::string my_err = pre_err;
    if (!my_rc) { my_rc = map_create_for_graph(ctx, local->k, total, &m); if (my_rc) my_err = ctx->err; }
    const u64 send_cap = std::min<u64>(CHS, local->capacity);
    if (!my_rc) {
        hipError_t e = hipMalloc((void **)&d_send_k, std::max<u64>(send_cap, 1) * 8 * W);
        if (e == hipSuccess) e = hipMalloc((void **)&d_send_c, std::max<u64>(send_cap, 1) * 4);
        if (e == hipSuccess) e = hipMalloc((void **)&d_cur, 8);
        if (e == hipSuccess) e = hipMalloc((void **)&d_recv_k, recv_cap * 8 * W);
        if (e == hipSuccess) e = hipMalloc((void **)&d_recv_c, recv_cap * 4);
        if (e == hipSuccess && classify) e = hipMalloc((void **)&d_send_m, std::max<u64>(send_cap, 1));
        if (e == hipSuccess && classify) e = hipMalloc((void **)&d_recv_m, recv_cap);
        if (e == hipSuccess && classify) e = hipMalloc((void **)&d_dc, 5 * 64 * 8);
        if (e != hipSuccess) { my_rc = hip_fail(ctx, e, "gk_dist_gather_map: staging"); my_err = ctx->err; }
    }
#define HIPD(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return done(hip_fail(ctx, e__, #call)); } while (0)
    for (u64 c = 0; c < nchunks; c++) {
        uint64_t n_mine = 0;
        if (classify && P > 1) {
            // ---- the owners' classify of this chunk's slots: count -> sizes round -> queries -> answers -> masks --------------
            unsigned long long h_send[64] = {0}, h_recv[64] = {0}, h_off[64] = {0};
            if (!my_rc) {
                my_rc = dclass_count(local, d->rank, P, c * CHS, (c + 1) * CHS, d_dc + 2 * 64);
                if (!my_rc) {
                    hipError_t e = hipMemcpyAsync(h_send, d_dc + 2 * 64, P * 8, hipMemcpyDeviceToHost, ctx->stream);
                    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
                    if (e != hipSuccess) my_rc = hip_fail(ctx, e, "gk_dist_gather_map: classify counts");
                }
                if (my_rc) my_err = ctx->err;
            }
            // what I will ask of every peer (or ~0: I failed, nobody goes on), against what every peer will ask of me
            unsigned long long words[64];
            for (int p = 0; p < P; p++) words[p] = my_rc ? ~0ull : h_send[p];
            unsigned long long *d_w_out = d->d_cnt + 4 * 64, *d_w_in = d->d_cnt + 5 * 64;      // (the handle's own words: they exist whatever failed above)
            HIPD(hipMemcpyAsync(d_w_out, words, P * 8, hipMemcpyHostToDevice, ctx->stream));
            {
                int grc = xGroupStart(d);
                for (int p = 0; p < P; p++) {
                    if (p == d->rank) continue;
                    if (grc == ncclSuccess) grc = xSend(d, d_w_out + p, 1, ncclUint64, p, ctx->stream);
                    if (grc == ncclSuccess) grc = xRecv(d, d_w_in + p, 1, ncclUint64, p, ctx->stream);
                }
                const int gend = xGroupEnd(d);
                if (grc == ncclSuccess) grc = gend;
                if (grc != ncclSuccess) return done(fail(ctx, GK_E_COMM, std::string("gk_dist_gather_map: ") + comm_error_text(grc)));
            }
            HIPD(hipMemcpyAsync(h_recv, d_w_in, P * 8, hipMemcpyDeviceToHost, ctx->stream));
            HIPD(hipStreamSynchronize(ctx->stream));
            h_recv[d->rank] = 0;
            bool peer_failed = false;
            for (int p = 0; p < P; p++) if (p != d->rank && h_recv[p] == ~0ull) peer_failed = true;
            // (a rank that failed told EVERY peer so: all of them are here, none has posted a payload receive)
            if (my_rc || peer_failed) return done(my_rc ? fail(ctx, my_rc, my_err) : fail(ctx, GK_E_COMM, "gk_dist_gather_map: another rank failed; the gather was abandoned on every rank"));
            u64 nq_out = 0, nq_in = 0;
            unsigned long long r_off[64] = {0};
            for (int p = 0; p < P; p++) { h_off[p] = nq_out; nq_out += h_send[p]; r_off[p] = nq_in; nq_in += h_recv[p]; }
            // room for both directions; a failure here is agreed on before anybody posts a receive
            if (ctx->hook_dist_fail_classify) {       // test hook: this rank alone cannot stage its queries
                ctx->hook_dist_fail_classify = 0;
                my_rc = fail(ctx, GK_E_CAPACITY, "injected failure (test_dist_fail_classify)");
                my_err = ctx->err;
            }
            if (!my_rc && nq_out > q_out_cap) {
                for (void *q : {(void *)d_qk, (void *)d_qref, (void *)d_ans_in}) if (q) (void)hipFree(q);
                d_qk = d_qref = nullptr; d_ans_in = nullptr; q_out_cap = 0;
                hipError_t e = hipMalloc((void **)&d_qk, nq_out * 8 * W);
                if (e == hipSuccess) e = hipMalloc((void **)&d_qref, nq_out * 8);
                if (e == hipSuccess) e = hipMalloc((void **)&d_ans_in, nq_out);
                if (e != hipSuccess) { my_rc = hip_fail(ctx, e, "gk_dist_gather_map: query staging"); my_err = ctx->err; } else q_out_cap = nq_out;
            }
            if (!my_rc && nq_in > q_in_cap) {
                for (void *q : {(void *)d_rk, (void *)d_ans_out}) if (q) (void)hipFree(q);
                d_rk = nullptr; d_ans_out = nullptr; q_in_cap = 0;
                hipError_t e = hipMalloc((void **)&d_rk, nq_in * 8 * W);
                if (e == hipSuccess) e = hipMalloc((void **)&d_ans_out, nq_in);
                if (e != hipSuccess) { my_rc = hip_fail(ctx, e, "gk_dist_gather_map: query staging"); my_err = ctx->err; } else q_in_cap = nq_in;
            }
            {
                unsigned long long word = my_rc ? 1ull : 0ull;
                HIPD(hipMemcpyAsync(d->d_cnt + 2 * 64, &word, 8, hipMemcpyHostToDevice, ctx->stream));
                const int g = xAllReduce(d, d->d_cnt + 2 * 64, d->d_cnt + 2 * 64, 1, ncclUint64, ncclMax, ctx->stream);
                if (g != ncclSuccess) return done(fail(ctx, GK_E_COMM, "gk_dist_gather_map: " + comm_error_text(g)));
                HIPD(hipMemcpyAsync(&word, d->d_cnt + 2 * 64, 8, hipMemcpyDeviceToHost, ctx->stream));
                HIPD(hipStreamSynchronize(ctx->stream));
                if (my_rc) return done(fail(ctx, my_rc, my_err));
                if (word) return done(fail(ctx, GK_E_COMM, "gk_dist_gather_map: another rank failed; the gather was abandoned on every rank"));
            }
            HIPD(hipMemcpyAsync(d_dc + 3 * 64, h_off, P * 8, hipMemcpyHostToDevice, ctx->stream));
            if (nq_out) { if (int rc = dclass_fill(local, d->rank, P, c * CHS, (c + 1) * CHS, d_dc + 3 * 64, d_dc + 4 * 64, d_qk, d_qref)) return done(rc); }
            {
                int grc = xGroupStart(d);
                for (int p = 0; p < P; p++) {
                    if (p == d->rank) continue;
                    if (h_send[p] && grc == ncclSuccess) grc = xSend(d, d_qk + h_off[p] * W, (size_t)h_send[p] * W, ncclUint64, p, ctx->stream);
                    if (h_recv[p] && grc == ncclSuccess) grc = xRecv(d, d_rk + r_off[p] * W, (size_t)h_recv[p] * W, ncclUint64, p, ctx->stream);
                }
                const int gend = xGroupEnd(d);
                if (grc == ncclSuccess) grc = gend;
                if (grc != ncclSuccess) return done(fail(ctx, GK_E_COMM, std::string("gk_dist_gather_map: ") + comm_error_text(grc)));
            }
            if (int rc = dclass_answer(local, d_rk, nq_in, d_ans_out)) return done(rc);
            {
                int grc = xGroupStart(d);
                for (int p = 0; p < P; p++) {
                    if (p == d->rank) continue;
                    if (h_recv[p] && grc == ncclSuccess) grc = xSend(d, d_ans_out + r_off[p], (size_t)h_recv[p], ncclUint8, p, ctx->stream);
                    if (h_send[p] && grc == ncclSuccess) grc = xRecv(d, d_ans_in + h_off[p], (size_t)h_send[p], ncclUint8, p, ctx->stream);
                }
                const int gend = xGroupEnd(d);
                if (grc == ncclSuccess) grc = gend;
                if (grc != ncclSuccess) return done(fail(ctx, GK_E_COMM, std::string("gk_dist_gather_map: ") + comm_error_text(grc)));
            }
            if (int rc = dclass_apply(local, d_qref, d_ans_in, nq_out)) return done(rc);
            d->classify_queries += nq_out;
        } else if (classify && !my_rc) {
            my_rc = dclass_count(local, d->rank, P, c * CHS, (c + 1) * CHS, d_dc + 2 * 64);       // one rank: every neighbour is local
            if (my_rc) my_err = ctx->err;
        }
        if (!my_rc && c * CHS < local->capacity) {
            my_rc = map_export_range_dev(local, c * CHS, (c + 1) * CHS, d_send_k, d_send_c, d_cur, &n_mine, classify ? d_send_m : nullptr);
            if (my_rc) my_err = ctx->err;
        }
        unsigned long long word = my_rc ? ~0ull : n_mine;
        HIPD(hipMemcpyAsync(d->d_cnt + 2 * 64, &word, 8, hipMemcpyHostToDevice, ctx->stream));
        { const int g = xAllGather(d, d->d_cnt + 2 * 64, d->d_cnt, 1, ncclUint64, ctx->stream); if (g != ncclSuccess) return done(fail(ctx, GK_E_COMM, "gk_dist_gather_map: " + comm_error_text(g))); }
        HIPD(hipMemcpyAsync(d->h_cnt, d->d_cnt, P * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPD(hipStreamSynchronize(ctx->stream));
        u64 tot_c = 0, my_off = 0;
        bool peer_failed = false;
        for (int p = 0; p < P; p++) {
            if (d->h_cnt[p] == ~0ull) { peer_failed = true; continue; }
            if (p == d->rank) my_off = tot_c;
            tot_c += d->h_cnt[p];
        }
        if (peer_failed) return done(my_rc ? fail(ctx, my_rc, my_err) : fail(ctx, GK_E_COMM, "gk_dist_gather_map: another rank failed; the gather was abandoned on every rank"));
        if (tot_c == 0) continue;
        if (tot_c > recv_cap) return done(fail(ctx, GK_E_STATE, "gk_dist_gather_map: a chunk holds more keys than its slots"));   // (cannot happen: <= CHS per rank)
        if (n_mine) {
            HIPD(hipMemcpyAsync(d_recv_k + my_off * W, d_send_k, n_mine * 8 * W, hipMemcpyDeviceToDevice, ctx->stream));
            HIPD(hipMemcpyAsync(d_recv_c + my_off, d_send_c, n_mine * 4, hipMemcpyDeviceToDevice, ctx->stream));
            if (classify) HIPD(hipMemcpyAsync(d_recv_m + my_off, d_send_m, n_mine, hipMemcpyDeviceToDevice, ctx->stream));
        }
        if (P > 1) {
            int grc = xGroupStart(d);
            u64 off = 0;
            for (int p = 0; p < P; p++) {
                const u64 n = d->h_cnt[p];
                if (p != d->rank) {
                    if (n_mine && grc == ncclSuccess) grc = xSend(d, d_send_k, (size_t)n_mine * W, ncclUint64, p, ctx->stream);
                    if (n_mine && grc == ncclSuccess) grc = xSend(d, d_send_c, (size_t)n_mine * 4, ncclUint8, p, ctx->stream);
                    if (classify && n_mine && grc == ncclSuccess) grc = xSend(d, d_send_m, (size_t)n_mine, ncclUint8, p, ctx->stream);
                    if (n && grc == ncclSuccess) grc = xRecv(d, d_recv_k + off * W, (size_t)n * W, ncclUint64, p, ctx->stream);
                    if (n && grc == ncclSuccess) grc = xRecv(d, d_recv_c + off, (size_t)n * 4, ncclUint8, p, ctx->stream);
                    if (classify && n && grc == ncclSuccess) grc = xRecv(d, d_recv_m + off, (size_t)n, ncclUint8, p, ctx->stream);
                }
                off += n;
            }
            const int gend = xGroupEnd(d);                 // (the group is always closed, whatever was posted)
            if (grc == ncclSuccess) grc = gend;
            if (grc != ncclSuccess) return done(fail(ctx, GK_E_COMM, std::string("gk_dist_gather_map: ") + comm_error_text(grc)));
        }
        // one table holding every partition's keys (each key has exactly one owner: nothing merges)
        // (each key has exactly one owner and the table was created for all of them: one CAS per key, count and mask beside it)
        my_rc = map_add_unique_keys_dev(m, d_recv_k, d_recv_c, classify ? d_recv_m : nullptr, tot_c);
        if (my_rc) my_err = ctx->err;
    }
    // the last chunk's insert may have failed after its word went out: agree on the outcome once more
    {
        unsigned long long word = my_rc ? ~0ull : 0ull;
        HIPD(hipMemcpyAsync(d->d_cnt + 2 * 64, &word, 8, hipMemcpyHostToDevice, ctx->stream));
        const int g = xAllReduce(d, d->d_cnt + 2 * 64, d->d_cnt + 2 * 64, 1, ncclUint64, ncclMax, ctx->stream);
        if (g != ncclSuccess) return done(fail(ctx, GK_E_COMM, "gk_dist_gather_map: " + comm_error_text(g)));
        HIPD(hipMemcpyAsync(&word, d->d_cnt + 2 * 64, 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPD(hipStreamSynchronize(ctx->stream));
        if (my_rc) return done(fail(ctx, my_rc, my_err));
        if (word) return done(fail(ctx, GK_E_COMM, "gk_dist_gather_map: another rank failed; the gathered table is void on every rank"));
    }
    if (int rc = map_sync_counters(m)) return done(rc);
    if (m->size != total) return done(fail(ctx, GK_E_STATE, "gk_dist_gather_map: gathered " + std::to_string(m->size) + " keys, the partitions hold " + std::to_string(total)));
    m->dirty = any_dirty;
    m->masks_valid = classify;           // (set after the last map_sync_counters of this table: any later change of its contents clears it)
    *full = m;
    return done(GK_OK);
}
#undef HIPD
int gk_dist_gather_map(gk_dist *d, gk_map *local, gk_map **full) { return dist_gather(d, local, full, false); }
int gk_dist_gather_classified_map(gk_dist *d, gk_map *local, gk_map **full) { return dist_gather(d, local, full, true); }
int gk_dist_classify_queries(gk_dist *d, uint64_t *n) {
    if (!d || !n) return fail(nullptr, GK_E_INVALID, "gk_dist_classify_queries: null argument");
    *n = d->classify_queries;
    return GK_OK;
}
#define HIPD(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return hip_fail(ctx, e__, #call); } while (0)
#undef HIPD

}  // extern "C"
