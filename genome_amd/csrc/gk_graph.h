// gk_graph.h — the device-side view of a MapGraph (S/data/graph/Graph.scala:153-209) and the host handle, shared by
// gk_graph.hip (build, structural edits) and gk_pairs.hip (the paired-end stage).
#pragma once

#include <memory>
#include <unordered_map>
#include <vector>

#include "gk_internal.h"

using namespace gk;

struct gk_vmap;
namespace gk {
int vmap_put_new_dev(gk_vmap *m, const uint64_t *d_lo, const uint64_t *d_hi, const uint64_t *d_val, uint64_t n);
int vmap_k(const gk_vmap *m);
gk_ctx *vmap_ctx(const gk_vmap *m);
}

static constexpr u32 NONE = 0xFFFFFFFFu;
static constexpr u32 AUX_TERMINAL = 1u << 8;
static constexpr u32 AUX_SECONDARY = 1u << 9;
// Once k_make_nodes has numbered the terminal k-mers, a terminal slot's annotation IS its node: AUX_NODE | j, where the stored
// orientation is node 2j and its reverse complement node 2j + 1.  (Until round 3 a separate u32 per table SLOT held that
// number: 4 bytes x capacity — 19 GB at C5 — and one more random read at every edge's end.)  Its degree masks are not
// needed any more at that point: a walk stops at a terminal k-mer, it never leaves one through the table.
static constexpr u32 AUX_NODE = 1u << 31;
__device__ __forceinline__ u32 aux_node(u32 aux, bool fwd) { return 2u * (aux & 0x7fffffffu) + (fwd ? 0u : 1u); }

// ---------------------------------------------------------------------------------------------
// device-side view of a graph
// ---------------------------------------------------------------------------------------------
struct GraphView {
    int k;
    u64 n_nodes, n_edges;
    u64 *node_lo, *node_hi;
    uint8_t *node_alive;
    u32 *out_edge;      // [n_nodes*4], by first base
    u32 *out_order;     // count in bits 0..2, i-th base in bits 4+2i..5+2i
    u32 *in_deg;
    u32 *e_start, *e_end;
    u64 *e_len, *e_off;
    uint8_t *e_alive, *e_first;
    uint8_t *pool;
    u32 *nidx;          // open-addressed k-mer -> node id index
    u64 nidx_mask;
};

struct gk_graph {
    gk_ctx *ctx = nullptr;
    int k = 0, W = 1;
    GraphView v{};
    void *node_blob = nullptr, *edge_blob = nullptr;      // the node / edge arrays of `v` are carved out of these two allocations
    u64 node_cap = 0, edge_cap = 0, pool_cap = 0, pool_used = 0;
    u64 live_nodes = 0, live_edges = 0, live_len = 0;
    // wall time of the phases of gk_graph_build (every phase ends in a stream sync): classify, terminals -> nodes + edge
    // stubs, unitig measure (k_walk pass 0 / pointer jumping), pool reservation, unitig emit, node index + counts
    float build_ms[6] = {0, 0, 0, 0, 0, 0};
    u64 walked_bases = 0;        // bases emitted by the unitig construction (= total edge length at build time)
    int used_pj = 0;
    bool index_ready = false;    // the k-mer -> node id index (nidx) exists: it is built on the first point query / by-k-mer edit, not by the build
    int used_masks = 0;          // the classify came with the table (masks computed by the keys' owners, gk_dist_gather_map): no k_classify ran
    float mbt_ms = 0;            // building the minimizer-bucketed copy of the table, when the build used one ("graph_mbt")
    u64 mbt_slots = 0;
    // host snapshot of the edge arrays for the paired-end walks, valid while `epoch` (bumped by every edit) has not moved:
    // a stream of gk_graph_walk_pairs batches downloads the graph once
    u64 epoch = 0, snap_epoch = ~0ull;
    std::shared_ptr<void> snap;
};

__device__ __forceinline__ int order_count(u32 o) { return (int)(o & 7u); }
__device__ __forceinline__ int order_base(u32 o, int i) { return (int)((o >> (4 + 2 * i)) & 3u); }
__device__ __forceinline__ u32 order_append(u32 o, int b) {
    int c = order_count(o);
    return ((o & ~7u) | (u32)(c + 1)) | ((u32)b << (4 + 2 * c));
}
__device__ __forceinline__ u32 order_remove(u32 o, int b) {
    u32 r = 0;
    for (int i = 0; i < order_count(o); i++) if (order_base(o, i) != b) r = order_append(r, order_base(o, i));
    return r;
}
__device__ __forceinline__ u32 rev4(u32 m) { return ((m & 1) << 3) | ((m & 2) << 1) | ((m & 4) >> 1) | ((m & 8) >> 3); }
__device__ __forceinline__ int pool_get(const uint8_t *pool, u64 off, u64 i) { return (pool[off + (i >> 2)] >> ((i & 3) * 2)) & 3; }


// host-side helpers of gk_graph.hip that the paired-end stage (gk_pairs.hip) uses
int check_graph(const gk_graph *g);
int ggrid(const gk_ctx *ctx, u64 items);                  // grid of BLOCK-thread workgroups for `items` work items
int graph_refresh_counts(gk_graph *g);                    // live nodes / edges / bases; bumps the graph's epoch
int graph_build_index(gk_graph *g);                       // k-mer -> node id index
inline int graph_ensure_index(gk_graph *g) { return g->index_ready ? 0 : graph_build_index(g); }   // before any kernel that calls node_find / reads nidx
int graph_grow_nodes(gk_graph *g, u64 new_cap);
