// gk_dist.h — one rank of a PartitionedDNAMap over the GPUs of a node (gk_dist.hip) and the TRANSPORT it talks through:
// RCCL in the product library; the test library (gk_testhooks.hip, libgenome_amd_test.so) adds a loopback hub whose ranks are
// threads of one process on one device.  Everything above the transport — routing, the exchange's three steps, the buffers'
// rotation, the helper thread — is the same code for both.
#pragma once

#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "gk_internal.h"

struct gk_dist;
namespace gk {
enum { XP_SUCCESS = 0 };
enum { XP_INT8 = 0, XP_UINT8 = 1, XP_UINT64 = 5, XP_FLOAT64 = 8 };      // ncclDataType_t values
enum { XP_SUM = 0, XP_MAX = 2 };                                         // ncclRedOp_t values
// NCCL-shaped operations on a handle's communicator; every function returns 0 or a transport error code (error_text explains it)
struct Transport {
    int (*group_start)(gk_dist *);
    int (*send)(gk_dist *, const void *, size_t count, int dtype, int peer, hipStream_t);
    int (*recv)(gk_dist *, void *, size_t count, int dtype, int peer, hipStream_t);
    int (*group_end)(gk_dist *);
    int (*all_reduce)(gk_dist *, const void *in, void *out, size_t n, int dtype, int op, hipStream_t);
    int (*all_gather)(gk_dist *, const void *in, void *out, size_t n_per_rank, int dtype, hipStream_t);
    std::string (*error_text)(int code);
    void (*close)(gk_dist *);                        // called by gk_dist_destroy before the handle's buffers go
};
// the transport-independent part of creating a handle (counters, events, the communication stream); *out is complete but for
// xport / comm / xstate, which the caller sets.  On failure nothing is left behind.
int dist_create_common(gk_ctx *ctx, int rank, int world, gk_dist **out);
}  // namespace gk

struct gk_dist {
    gk_ctx *ctx = nullptr;
    int rank = 0, world = 1;
    const gk::Transport *xport = nullptr;            // RCCL (gk_dist.hip), or the test library's loopback hub (gk_testhooks.hip)
    void *comm = nullptr;                            // RCCL: the ncclComm_t
    std::shared_ptr<void> xstate;                    // the transport's own per-handle state (loopback: its hub and the group being built)
    // Every RCCL call of this handle goes to ONE stream of its own (operations on a communicator must not run concurrently):
    // the exchange of batch i+1 can then be in flight while the owner pipeline of batch i runs on the context's stream.
    hipStream_t comm_stream = nullptr;
    // exchange scratch, kept between calls.  THREE send buffers (a route being written on the second stream, one whose
    // records are on the wire, one being counted) and TWO receive buffers (on the wire / being counted).
    static constexpr int NROUTE = 3;
    uint8_t *d_sendbuf[NROUTE] = {nullptr, nullptr, nullptr}, *d_recv[2] = {nullptr, nullptr};
    gk::u64 send_cap[NROUTE] = {0, 0, 0}, recv_records[2] = {0, 0};      // capacities in record slots (send: world regions of send_cap / world)
    int slot = 0;                                    // record slot bytes the buffers were sized for
    // routes that were begun and not yet counted: at most three, first in, first out
    struct Route {
        int k = 0, read_len = 0; const void *records = nullptr; gk::u64 nreads = 0;
        // settled (owner thread): the routing kernel has finished, a region that was too small has been routed again, and
        // this rank's verdict on its own half of the exchange is in local_rc — what it will tell its peers
        bool settled = false;
        int local_rc = 0;
        std::string local_err;
        uint64_t recs[64] = {}, kmers[64] = {};       // per owner
        bool exchanged = false;                      // counts known, records on the wire (or arrived) in d_recv[rbuf]
        int rbuf = 0;
        gk::u64 nrec_in = 0, nkm_in = 0, sent = 0;
        int error = 0;                               // the batch was dropped (by agreement of all ranks, or by the transport): reported when its turn comes
        std::string error_text;
    };
    Route route[NROUTE];
    int head = 0, npending = 0;                      // route[head] is the oldest; the next route goes to (head + npending) % NROUTE
    gk::u64 nexchanged = 0;                              // exchanges posted so far: the next one receives into d_recv[nexchanged & 1]
    unsigned long long *d_route_cnt = nullptr;       // [NROUTE][SKM_COUNT_WORDS] counters of the routing kernels on the second stream
    unsigned long long *h_route_cnt = nullptr;       // pinned copy
    hipEvent_t route_done[NROUTE] = {nullptr, nullptr, nullptr};   // recorded behind each route's counter copy
    hipEvent_t exch_done[NROUTE] = {nullptr, nullptr, nullptr};    // recorded behind each batch's receives
    hipEvent_t join = nullptr;                       // main stream -> communication stream
    unsigned long long *d_cnt = nullptr;             // [8 x 64]: (records, k-mers, status) per peer to send [0, 3 x 64), as received [3 x 64, 6 x 64), scalars behind
    unsigned long long *h_cnt = nullptr;             // pinned mirror
    float last_ms[4] = {0, 0, 0, 0};                 // route, exchange, owner count, total (wall)
    gk::u64 classify_queries = 0;                    // neighbour lookups this rank asked of other ranks (classified gathers, since creation)
    float last_helper_ms = 0;                        // host time of the exchange that ran beside the last owner count
    // ONE helper thread per handle, started on first use: it runs the exchange of the next batch beside the owner count
    // (a std::thread per step was 30-50 us of clone + join each).  It never touches the context's error string, its
    // streams or its block pool's frees: errors come back in the route, replaced buffers go to `garbage` for the owner thread.
    std::thread worker;
    std::mutex wmu;
    std::condition_variable wcv;
    std::function<void()> job;
    bool job_pending = false, job_running = false, quit = false;
    std::vector<void *> garbage;
};

