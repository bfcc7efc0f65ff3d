// gk_testhooks.hip — what ONLY the test build of the library has (libgenome_amd_test.so = the product's objects + this one):
//   * gk_ctx_set_option and the environment switches read at gk_ctx_create: A/B switches of the kernels, test hooks that stage
//     the large-table paths on small tables or inject failures;
//   * gk_dist_create_loopback: a transport whose ranks are threads of ONE process on ONE device, so that gk_dist_* runs with
//     world > 1 on a one-GPU box (RCCL refuses two ranks on one device).
// The product library (libgenome_amd.so) links none of this; its header is include/genome_amd.h, this file's is
// include/genome_amd_test.h.  The tests load the test build through GK_LIB_PATH (tests/conftest.py).
#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "gk_dist.h"
#include "../../include/genome_amd_test.h"

using namespace gk;

extern "C" {

void gk_testhooks_env(gk_ctx *ctx) {
    ctx->hook_no_reserve = getenv("GK_TEST_NO_RESERVE") != nullptr;
    ctx->hook_host_ragged = getenv("GK_HOST_RAGGED") != nullptr;
    ctx->hook_part_exact = getenv("GK_PART_EXACT") != nullptr;
    if (const char *u = getenv("GK_P45_STRIPES")) ctx->hook_p45_stripes = atoi(u);
    if (const char *u = getenv("GK_MIN_LNB1")) ctx->hook_min_lnb1 = std::max(0, std::min((int)gk::MAX_LNB1, atoi(u)));      // (tests: the whole suite over 512 / 1024 L1 buckets)
    if (const char *u = getenv("GK_GRAPH_UNITIGS")) ctx->hook_unitigs = !strcmp(u, "walk") ? 1 : !strcmp(u, "pj") ? 2 : 0;
    if (const char *u = getenv("GK_GRAPH_MBT")) ctx->hook_graph_mbt = atoi(u);            // (tests: the whole graph suite over the bucketed table)
}

int gk_ctx_set_option(gk_ctx *ctx, const char *name, int64_t value) {
    if (!ctx || !name) return fail(ctx, GK_E_INVALID, "gk_ctx_set_option: null argument");
    const std::string n(name);
    if (n == "test_no_reserve") ctx->hook_no_reserve = value != 0;
    else if (n == "host_ragged") ctx->hook_host_ragged = value != 0;
    else if (n == "part_exact") ctx->hook_part_exact = value != 0;
    else if (n == "graph_unitigs") {
        if (value < 0 || value > 2) return fail(ctx, GK_E_INVALID, "graph_unitigs: 0 auto, 1 walk, 2 pointer jumping");
        ctx->hook_unitigs = (int)value;
    } else if (n == "p4_direct") ctx->hook_p4_direct = value < 0 ? -1 : value != 0;
    else if (n == "p2_wide") ctx->hook_p2_wide = value < 0 ? -1 : value != 0;
    else if (n == "p2_sorted") ctx->hook_p2_sorted = value < 0 ? -1 : value != 0;
    else if (n == "p4_wide") ctx->hook_p4_wide = value < 0 ? -1 : (value >= 2 ? 2 : value != 0);
    else if (n == "p45_stripes") ctx->hook_p45_stripes = (int)value;
    else if (n == "p24_pieces") ctx->hook_p24_pieces = (int)value;
    else if (n == "dist_exchange_ahead") ctx->hook_dist_ahead = (int)value;
    else if (n == "test_dist_small_send") ctx->hook_dist_small_send = (int)std::max<int64_t>(0, value);
    else if (n == "target_load_pct") ctx->hook_target_load_pct = (int)value;
    else if (n == "cc_find") ctx->hook_cc_find = (int)value;
    else if (n == "test_dist_fail_classify") ctx->hook_dist_fail_classify = (int)value;
    else if (n == "test_dist_fail_exchange") ctx->hook_dist_fail = value > 0 ? -(int)value : (int)value;
    else if (n == "test_max_nb2") ctx->hook_max_nb2 = (int)std::max<int64_t>(0, value);
    else if (n == "min_lnb1") {
        if (value < 0 || value > (int64_t)gk::MAX_LNB1) return fail(ctx, GK_E_INVALID, "min_lnb1: 0..10");
        ctx->hook_min_lnb1 = (int)value;
    }
    else if (n == "p4_grid") ctx->hook_p4_grid = (int)value;
    else if (n == "graph_walk_queue") ctx->hook_walk_queue = (int)value;
    else if (n == "graph_mbt") ctx->hook_graph_mbt = (int)value;
    else if (n == "graph_mbt_keys") ctx->hook_graph_mbt_keys = (int)std::max<int64_t>(16, value);
    else if (n == "pairs_host") ctx->hook_pairs_host = (int)value;
    else if (n == "host_prefetch") ctx->hook_host_prefetch = (int)value;
    else if (n == "test_max_stage") ctx->hook_max_stage = (int64_t)std::max<int64_t>(0, value);
    else if (n == "test_pairs_small_sets") ctx->hook_pairs_small_sets = (int)value;
    else if (n == "filter_classic") ctx->hook_filter_classic = (int)value;
    else if (n == "graph_load_pct") ctx->hook_graph_load_pct = (int)value;
    else if (n == "fine_exact") ctx->hook_fine_exact = value < 0 ? -1 : value != 0;
    else return fail(ctx, GK_E_INVALID, "gk_ctx_set_option: unknown option '" + n + "'");
    return GK_OK;
}


}  // extern "C"

// ---- loopback transport (tests): the ranks of one "node" are threads of ONE process on ONE device --------------------------
// RCCL refuses two ranks on one GPU, and only one GPU is reachable from the build box: gk_dist_create_loopback gives every
// rank a handle whose sends, receives and reductions go through this hub instead — device-to-device copies between the
// ranks' buffers, matched pairwise in posting order like RCCL's — so that the exchange logic of gk_dist_* (sizes, regions,
// buffer rotation, the order of operations across ranks) runs with world > 1 before it ever meets a real communicator.  It is
// STRICTER than RCCL in two ways that make it a better test: a group's end blocks until every peer has posted the matching
// operation (an inconsistent order of operations across ranks deadlocks here at once), and a send whose size differs from
// the matching receive is an error.
namespace {
struct LoopHub {
    std::mutex mu;
    std::condition_variable cv;
    int world = 0, refs = 0;
    struct Op { void *ptr; size_t bytes; hipStream_t stream; hipEvent_t ready; bool *done; int *err; };
    std::deque<Op> sends[64][64], recvs[64][64];          // [src][dst], in posting order
    // reductions / gathers: one at a time, every rank contributes
    unsigned long long gen = 0;
    int arrived = 0;
    double contrib[64][32];
    double result[64 * 32];
    std::vector<hipEvent_t> events;                       // every event the transport made; destroyed with the hub
    ~LoopHub() { for (hipEvent_t e : events) if (e) (void)hipEventDestroy(e); }
};
std::mutex g_hubs_mu;
std::map<std::string, std::shared_ptr<LoopHub>> g_hubs;
}  // namespace


namespace {
struct LoopState {
    std::shared_ptr<LoopHub> hub;
    std::string key;
    struct Posted { bool send; void *ptr; size_t bytes; int peer; hipStream_t stream; };
    std::vector<Posted> group;                       // the operations of the group being built
    ~LoopState() {
        std::lock_guard<std::mutex> lk(g_hubs_mu);
        if (hub && --hub->refs == 0) g_hubs.erase(key);
    }
};
size_t dtype_bytes(int dt) { return dt == XP_UINT64 || dt == XP_FLOAT64 ? 8 : 1; }
int loop_group_start(gk_dist *d) { static_cast<LoopState *>(d->xstate.get())->group.clear(); return XP_SUCCESS; }
int loop_send(gk_dist *d, const void *p, size_t count, int dt, int peer, hipStream_t st) {
    static_cast<LoopState *>(d->xstate.get())->group.push_back({true, const_cast<void *>(p), count * dtype_bytes(dt), peer, st});
    return XP_SUCCESS;
}
int loop_recv(gk_dist *d, void *p, size_t count, int dt, int peer, hipStream_t st) {
    static_cast<LoopState *>(d->xstate.get())->group.push_back({false, p, count * dtype_bytes(dt), peer, st});
    return XP_SUCCESS;
}
int loop_group_end(gk_dist *d) {
    LoopState &ls = *static_cast<LoopState *>(d->xstate.get());
    LoopHub &h = *ls.hub;
    const size_t n = ls.group.size();
    std::unique_ptr<bool[]> done(new bool[n]());
    int err = 0;
    std::unique_lock<std::mutex> lk(h.mu);
    for (size_t i = 0; i < n; i++) {
        const LoopState::Posted &o = ls.group[i];
        hipEvent_t ev = nullptr;
        if (o.send) {                                 // the data is ready once the sender's stream reaches this point
            if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(ev, o.stream) != hipSuccess) err = 1;
            h.events.push_back(ev);
            h.sends[d->rank][o.peer].push_back({o.ptr, o.bytes, o.stream, ev, &done[i], &err});
        } else {
            h.recvs[o.peer][d->rank].push_back({o.ptr, o.bytes, o.stream, nullptr, &done[i], &err});
        }
    }
    // match whatever can be matched (any thread may complete any pair), then wait for the rest of this group
    auto match_all = [&]() {
        for (int s = 0; s < h.world; s++)
            for (int r = 0; r < h.world; r++)
                while (!h.sends[s][r].empty() && !h.recvs[s][r].empty()) {
                    LoopHub::Op a = h.sends[s][r].front(), b = h.recvs[s][r].front();
                    h.sends[s][r].pop_front(); h.recvs[s][r].pop_front();
                    int e = 0;
                    if (a.bytes != b.bytes) e = 2;                               // RCCL would hang or corrupt here
                    else if (a.bytes) {
                        hipEvent_t copied = nullptr;
                        if (hipStreamWaitEvent(b.stream, a.ready, 0) != hipSuccess) e = 1;
                        if (!e && hipMemcpyAsync(b.ptr, a.ptr, a.bytes, hipMemcpyDeviceToDevice, b.stream) != hipSuccess) e = 1;
                        // the sender may reuse its buffer only after the copy: its stream waits for it
                        if (!e && (hipEventCreateWithFlags(&copied, hipEventDisableTiming) != hipSuccess || hipEventRecord(copied, b.stream) != hipSuccess ||
                                   hipStreamWaitEvent(a.stream, copied, 0) != hipSuccess)) e = 1;
                        if (copied) h.events.push_back(copied);
                    }
                    if (e) { *a.err = e; *b.err = e; }
                    *a.done = true; *b.done = true;
                }
    };
    auto all_done = [&]() { for (size_t i = 0; i < n; i++) if (!done[i]) return false; return true; };
    match_all();
    h.cv.notify_all();
    while (!all_done()) {
        h.cv.wait(lk);
        match_all();
        h.cv.notify_all();
    }
    lk.unlock();
    ls.group.clear();
    return err ? 3 /* ncclInternalError */ : XP_SUCCESS;
}
static int loop_collective(gk_dist *d, const void *in, void *out, size_t n, int dt, int op, bool gather, hipStream_t st) {
    LoopHub &h = *static_cast<LoopState *>(d->xstate.get())->hub;
    if (n > 32 || (gather && n != 1)) return 3;
    double mine[32];
    if (hipStreamSynchronize(st) != hipSuccess || hipMemcpy(mine, in, n * 8, hipMemcpyDeviceToHost) != hipSuccess) return 3;
    double res[64 * 32];
    size_t nres = gather ? (size_t)h.world : n;
    {
        std::unique_lock<std::mutex> lk(h.mu);
        const unsigned long long gen = h.gen;
        memcpy(h.contrib[d->rank], mine, n * 8);
        if (++h.arrived == h.world) {
            for (size_t i = 0; i < nres; i++) {
                if (gather) { memcpy(&h.result[i], &h.contrib[i][0], 8); continue; }
                if (dt == XP_FLOAT64) {
                    double acc = h.contrib[0][i];
                    for (int r = 1; r < h.world; r++) acc = op == XP_MAX ? std::max(acc, h.contrib[r][i]) : acc + h.contrib[r][i];
                    h.result[i] = acc;
                } else {
                    unsigned long long acc = 0, v;
                    for (int r = 0; r < h.world; r++) { memcpy(&v, &h.contrib[r][i], 8); acc = op == XP_MAX ? std::max(acc, v) : acc + v; }
                    memcpy(&h.result[i], &acc, 8);
                }
            }
            h.arrived = 0;
            h.gen++;
            h.cv.notify_all();
        } else {
            h.cv.wait(lk, [&]() { return h.gen != gen; });
        }
        memcpy(res, h.result, nres * 8);
    }
    return hipMemcpyAsync(out, res, nres * 8, hipMemcpyHostToDevice, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess ? XP_SUCCESS : 3;
}
int loop_all_reduce(gk_dist *d, const void *in, void *out, size_t n, int dt, int op, hipStream_t st) { return loop_collective(d, in, out, n, dt, op, false, st); }
int loop_all_gather(gk_dist *d, const void *in, void *out, size_t n, int dt, hipStream_t st) { return loop_collective(d, in, out, n, dt, XP_SUM, true, st); }
std::string loop_error_text(int) { return "transport error (loopback: a send and its receive differ in size, or a HIP call failed)"; }
void loop_close(gk_dist *d) { d->xstate.reset(); }
const Transport LOOP_TRANSPORT = {loop_group_start, loop_send, loop_recv, loop_group_end, loop_all_reduce, loop_all_gather, loop_error_text, loop_close};
}  // namespace

extern "C" int gk_dist_create_loopback(gk_ctx *ctx, int rank, int world, const void *id128, gk_dist **out) {
    if (!ctx || !out || !id128) return fail(ctx, GK_E_INVALID, "gk_dist_create_loopback: null argument");
    *out = nullptr;
    if (world < 1 || world > 64 || rank < 0 || rank >= world) return fail(ctx, GK_E_INVALID, "gk_dist_create_loopback: need 0 <= rank < world <= 64");
    GK_HIP(ctx, hipSetDevice(ctx->device));
    // the ranks of one loopback "node" find each other by the id (any 128 bytes every rank was given alike)
    auto ls = std::make_shared<LoopState>();
    ls->key.assign((const char *)id128, 128);
    {
        std::lock_guard<std::mutex> lk(g_hubs_mu);
        std::shared_ptr<LoopHub> &h = g_hubs[ls->key];
        if (!h) { h = std::make_shared<LoopHub>(); h->world = world; }
        if (h->world != world) return fail(ctx, GK_E_INVALID, "gk_dist_create_loopback: the ranks of one id disagree about the world size");
        h->refs++;
        ls->hub = h;
    }
    gk_dist *d = nullptr;
    if (int rc = dist_create_common(ctx, rank, world, &d)) return rc;
    d->xport = &LOOP_TRANSPORT;
    d->xstate = ls;
    *out = d;
    return GK_OK;
}
