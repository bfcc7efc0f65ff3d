// gk_pairs.hip — the paired-end stage of the simplifier (S/scripts/GraphSimplifier.scala:33-127, 188-318) on the device.
//
//   :213-217  the mates' first k-mers and their four getAll          k_pair_keys (cut from the `.bin` pairs in HBM), k_vm_get_all
//                                                                    (count, scan, fill: positions stay in HBM as CSR)
//   :192-206  annotate                                               k_walk_pairs, per pair orientation
//   :43-72    WalkingActor.reachable (backward, bounded by range.last)   k_walk_pairs: label-correcting relaxation in LDS
//   :77-126   WalkingActor.dfs + memo                                k_walk_pairs: the states (previous edge, distance) as a set in
//                                                                    LDS, resolved to a fixpoint — same supported edge pairs as the
//                                                                    recursion (tests: the oracle's literal dfs)
//   :209-247  pathsMap / badPairs                                    a device hash table (edge, edge) -> count in gk_support
//   :272-316  support matrix, node split, edge removal               host over a snapshot + batch edit kernels (small graph)
//
// ONE WAVE PER PAIR ORIENTATION.  A walk is a few dozen states deep and branches on the graph: no lane-per-walk form keeps a
// wave busy, and the per-walk sets (reached nodes, states, the orientation's supported pairs) must be shared — they live in
// 8 KB of LDS per wave and the 64 lanes work them as a team: relax in-edges of the dirty reached nodes, expand the new
// states, resolve, emit.  A walk that outgrows its LDS sets (256 reached nodes, 384 states, 96 pairs) is not truncated: its
// orientation goes to an overflow list and the host walker (the round-2 form, kept below) does it — exact either way.
// Round 2 ran ALL walks on <= 16 host threads over a snapshot: 1.0-1.5e7 pairs/s, ten times everything before it at C3.
#include <algorithm>
#include <cstring>
#include <mutex>
#include <thread>

#include "gk_graph.h"
#include "gk_scan.h"
#include "gk_tile.h"

struct gk_vmap;
namespace gk {
int vmap_get_all_dev(gk_vmap *m, const uint64_t *d_lo, const uint64_t *d_hi, uint64_t n, const unsigned long long *d_off, uint32_t *d_cnt, uint64_t *d_out);
int vmap_k(const gk_vmap *m);
gk_ctx *vmap_ctx(const gk_vmap *m);
}

// ---------------------------------------------------------------------------------------------
// gk_support: pathsMap (:209) + badPairs (:211) on the device
// ---------------------------------------------------------------------------------------------
static constexpr u64 SUP_EMPTY = ~0ull;
struct SupView { u64 *keys; u32 *cnt; u64 mask; unsigned long long *ctr; };     // ctr: [0] distinct pairs [1] bad pairs [2] orientations walked [3] table full
struct gk_support {
    gk_ctx *ctx = nullptr;
    u64 *d_keys = nullptr;                     // (e1 << 32 | e2), open addressing, power-of-two capacity
    u32 *d_cnt = nullptr;
    u64 cap = 0;
    unsigned long long *d_ctr = nullptr;
    std::unordered_map<u64, u32> paths;        // host copy for the split (support_to_host), valid while host_valid
    bool host_valid = false;
    float last_ms[5] = {0, 0, 0, 0, 0};        // last gk_graph_walk_pairs: keys from the stream, getAll batch, in-edge lists + checks, walks, overflow walks on the host
    u64 last_overflow = 0;                     // orientations of the last call that went to the host walker
};

__device__ __forceinline__ void sup_add(const SupView &s, u64 key, u32 c) {
    u64 i = mix64(key) & s.mask;
    for (u64 n = 0; n <= s.mask; n++) {
        u64 cur = s.keys[i];
        if (cur == SUP_EMPTY) {
            cur = atomicCAS(reinterpret_cast<unsigned long long *>(&s.keys[i]), (unsigned long long)SUP_EMPTY, (unsigned long long)key);
            if (cur == SUP_EMPTY) { atomicAdd(&s.ctr[0], 1ull); cur = key; }
        }
        if (cur == key) { atomicAdd(&s.cnt[i], c); return; }
        i = (i + 1) & s.mask;
    }
    s.ctr[3] = 1;
}
__global__ __launch_bounds__(BLOCK) void k_sup_rehash(const u64 *okeys, const u32 *ocnt, u64 ocap, SupView s) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < ocap; i += (u64)gridDim.x * BLOCK)
        if (okeys[i] != SUP_EMPTY) sup_add(s, okeys[i], ocnt[i]);
}
__global__ __launch_bounds__(BLOCK) void k_sup_add_list(const u64 *keys, const u32 *cnt, u64 n, SupView s) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) sup_add(s, keys[i], cnt[i]);
}

// ---------------------------------------------------------------------------------------------
// the four keys of every pair (:213-217), cut on the device from equal-length `.bin` records:
// f1 = getAll(p1.take(k)), f2 = getAll(p2.take(k).revComplement), f3 = getAll(p2.take(k)), f4 = getAll(p1.take(k).revComplement)
// ---------------------------------------------------------------------------------------------
template <int W>
__global__ __launch_bounds__(BLOCK) void k_pair_keys(const uint8_t *__restrict__ bin, u64 npairs, u32 rb, int l0, int k, u64 *lo, u64 *hi, u32 *ragged) {
    for (u64 p = (u64)blockIdx.x * BLOCK + threadIdx.x; p < npairs; p += (u64)gridDim.x * BLOCK) {
        const uint8_t *r1 = bin + 2 * p * rb, *r2 = r1 + rb;
        if (r1[0] != (uint8_t)l0 || r2[0] != (uint8_t)l0) { *ragged = 1u; continue; }
        Kmer<W> m[2];
        for (int q = 0; q < 2; q++) {
            const uint8_t *pl = (q ? r2 : r1) + 1;
            u64 w0 = 0, w1 = 0;
            const int nbytes = (k + 3) / 4;
            for (int b = 0; b < nbytes && b < 8; b++) w0 |= (u64)pl[b] << (8 * b);
            for (int b = 8; b < nbytes; b++) w1 |= (u64)pl[b] << (8 * (b - 8));
            if constexpr (W == 1) m[q] = Kmer<1>{w0 & low_mask(2 * k)};
            else m[q] = Kmer<2>{w0, w1 & low_mask(2 * (k - 32))};
        }
        const Kmer<W> ra = revcomp(m[0], k), rbk = revcomp(m[1], k);
        const Kmer<W> four[4] = {m[0], rbk, m[1], ra};
        for (int q = 0; q < 4; q++) {
            lo[4 * p + q] = four[q].lo;
            if constexpr (W == 2) hi[4 * p + q] = four[q].hi;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// in-edge lists (Node.inEdgeIds) as CSR by end node, built on the device
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_in_count(GraphView g, u32 *cnt) {
    for (u64 e = (u64)blockIdx.x * BLOCK + threadIdx.x; e < g.n_edges; e += (u64)gridDim.x * BLOCK)
        if (g.e_alive[e]) atomicAdd(&cnt[g.e_end[e]], 1u);
}
__global__ __launch_bounds__(BLOCK) void k_in_fill(GraphView g, const unsigned long long *off, u32 *cursor, u32 *list) {
    for (u64 e = (u64)blockIdx.x * BLOCK + threadIdx.x; e < g.n_edges; e += (u64)gridDim.x * BLOCK)
        if (g.e_alive[e]) { const u32 v = g.e_end[e]; list[off[v] + atomicAdd(&cursor[v], 1u)] = (u32)e; }
}
// a position must name something of THIS graph (gk_graph_walk_pairs: GK_E_STATE otherwise)
__global__ __launch_bounds__(BLOCK) void k_check_positions(GraphView g, const u64 *vals, u64 n, u32 *bad) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) {
        const u64 v = vals[i];
        const u32 id = GK_POS_ID(v);
        const bool ok = GK_POS_IS_EDGE(v) ? (id < g.n_edges && g.e_alive[id] && (u64)GK_POS_DIST(v) < g.e_len[id]) : (id < g.n_nodes && g.node_alive[id]);
        if (!ok) *bad = 1u;
    }
}

// ---------------------------------------------------------------------------------------------
// the walks
// ---------------------------------------------------------------------------------------------
static constexpr u32 W_RCAP = 256, W_RMAX = 192;           // reached nodes: hash slots / entries before "overflow"
static constexpr u32 W_SCAP = 512, W_QCAP = 384;           // states: hash slots / entries
static constexpr u32 W_PCAP = 128, W_PMAX = 96;            // supported pairs of one orientation
static constexpr u32 W_ABSENT = 0xffffffffu;
static constexpr uint8_t WF_PRUNED = 1, WF_RES = 2;
enum { WM_QTAIL = 0, WM_RCOUNT = 1, WM_NPAIRS = 2, WM_OVF = 3 };
struct WaveLds {
    u64 st_key[W_SCAP];          // (previous edge << 16 | distance); ~0 = empty
    u64 pset[W_PCAP];            // (edge << 32 | edge) supported by this orientation
    u32 r_key[W_RCAP];           // node; NONE = empty
    u32 r_dist[W_RCAP];          // shortest distance back to node2 found so far
    u32 misc[8];
    uint16_t q[W_QCAP];          // state slots in order of discovery
    uint8_t r_dirty[W_RCAP];
    uint8_t st_flag[W_SCAP];
};
struct WalkArgs { int k, lo, hi; u32 rmax, qcap, pmax; };      // (the three set limits: W_RMAX / W_QCAP / W_PMAX, or a test's tiny ones)

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// reached[node] = min(reached[node], d); false if the set is full
__device__ __forceinline__ bool reach_relax(WaveLds &L, u32 node, u32 d, u32 rmax) {
    u32 h = hash32(node) & (W_RCAP - 1);
    for (u32 n = 0; n < W_RCAP; n++) {
        u32 cur = L.r_key[h];
        if (cur == NONE) {
            if (L.misc[WM_RCOUNT] >= rmax) return false;
            cur = atomicCAS(&L.r_key[h], NONE, node);
            if (cur == NONE) { atomicAdd(&L.misc[WM_RCOUNT], 1u); cur = node; }
        }
        if (cur == node) {
            if (atomicMin(&L.r_dist[h], d) > d) L.r_dirty[h] = 1;
            return true;
        }
        h = (h + 1) & (W_RCAP - 1);
    }
    return false;
}
__device__ __forceinline__ u32 reach_get(const WaveLds &L, u32 node) {
    u32 h = hash32(node) & (W_RCAP - 1);
    for (u32 n = 0; n < W_RCAP; n++) {
        const u32 cur = L.r_key[h];
        if (cur == node) return L.r_dist[h];
        if (cur == NONE) return W_ABSENT;
        h = (h + 1) & (W_RCAP - 1);
    }
    return W_ABSENT;
}
// the state (pe, d): its slot, new or old; -1 if the set is full
__device__ __forceinline__ int state_add(WaveLds &L, u32 pe, u32 d, u32 qcap) {
    const u64 key = ((u64)pe << 16) | d;
    u32 h = (u32)mix64(key) & (W_SCAP - 1);
    for (u32 n = 0; n < W_SCAP; n++) {
        u64 cur = L.st_key[h];
        if (cur == ~0ull) {
            cur = atomicCAS(reinterpret_cast<unsigned long long *>(&L.st_key[h]), ~0ull, (unsigned long long)key);
            if (cur == ~0ull) {
                const u32 qi = atomicAdd(&L.misc[WM_QTAIL], 1u);
                if (qi >= qcap) return -1;
                L.st_flag[h] = 0;
                L.q[qi] = (uint16_t)h;
                return (int)h;
            }
        }
        if (cur == key) return (int)h;
        h = (h + 1) & (W_SCAP - 1);
    }
    return -1;
}
__device__ __forceinline__ int state_find(const WaveLds &L, u32 pe, u32 d) {
    const u64 key = ((u64)pe << 16) | d;
    u32 h = (u32)mix64(key) & (W_SCAP - 1);
    for (u32 n = 0; n < W_SCAP; n++) {
        const u64 cur = L.st_key[h];
        if (cur == key) return (int)h;
        if (cur == ~0ull) return -1;
        h = (h + 1) & (W_SCAP - 1);
    }
    return -1;
}
__device__ __forceinline__ bool pair_add(WaveLds &L, u64 key, u32 pmax) {
    u32 h = (u32)mix64(key) & (W_PCAP - 1);
    for (u32 n = 0; n < W_PCAP; n++) {
        u64 cur = L.pset[h];
        if (cur == ~0ull) {
            if (L.misc[WM_NPAIRS] >= pmax) return false;
            cur = atomicCAS(reinterpret_cast<unsigned long long *>(&L.pset[h]), ~0ull, (unsigned long long)key);
            if (cur == ~0ull) { atomicAdd(&L.misc[WM_NPAIRS], 1u); return true; }
        }
        if (cur == key) return true;
        h = (h + 1) & (W_PCAP - 1);
    }
    return false;
}

// One (pos1, pos2) of WalkingActor.receive (:78-125), by the whole wave.  Supported (edge, edge) pairs go to L.pset; returns
// `good`; L.misc[WM_OVF] is raised if a set overflowed (the orientation is then redone on the host).
__device__ bool walk_one(WaveLds &L, const GraphView &g, const unsigned long long *__restrict__ in_off, const u32 *__restrict__ in_list, u64 v1, u64 v2,
                         const WalkArgs &A) {
    const int lane = threadIdx.x & 63;
    const bool e1 = GK_POS_IS_EDGE(v1), e2 = GK_POS_IS_EDGE(v2);
    const u32 id1 = GK_POS_ID(v1), id2 = GK_POS_ID(v2);
    const u32 node2 = e2 ? g.e_start[id2] : id2;
    const u32 dist2 = e2 ? GK_POS_DIST(v2) : 0u;
    const u32 end_edge = e2 ? id2 : NONE, start_edge = e1 ? id1 : NONE;
    const u32 node0 = e1 ? g.e_end[id1] : id1;
    const u64 dist0 = e1 ? g.e_len[id1] - GK_POS_DIST(v1) : 0ull;
    const u32 hi = (u32)A.hi, lo = (u32)A.lo;
    if (dist0 > (u64)hi) return false;                     // (the reference finds this out after `reachable`; nothing is recorded either way)
    // ---- clear the walk's sets
    for (u32 i = lane; i < W_RCAP; i += 64) { L.r_key[i] = NONE; L.r_dist[i] = 0xffffffffu; L.r_dirty[i] = 0; }
    for (u32 i = lane; i < W_SCAP; i += 64) L.st_key[i] = ~0ull;
    if (lane == 0) { L.misc[WM_QTAIL] = 0; L.misc[WM_RCOUNT] = 0; }
    wave_sync();
    bool ovf = false;
    // ---- reachable(node2) :43-72: shortest distance back along in-edges, <= hi.  Label correcting: a node whose distance
    //      dropped is dirty; every round relaxes the in-edges of the dirty nodes; done when a round changes nothing.
    if (lane == 0) ovf |= !reach_relax(L, node2, 0u, A.rmax);
    wave_sync();
    for (u32 round = 0; round < 65536u; round++) {
        bool did = false;
        for (u32 s = lane; s < W_RCAP; s += 64) {
            if (L.r_key[s] == NONE || !L.r_dirty[s]) continue;
            L.r_dirty[s] = 0;
            const u32 u = L.r_key[s], d = L.r_dist[s];
            for (u64 j = in_off[u]; j < in_off[u + 1]; j++) {
                const u32 e = in_list[j];
                const u64 d2 = (u64)d + g.e_len[e];
                if (d2 <= (u64)hi) ovf |= !reach_relax(L, g.e_start[e], (u32)d2, A.rmax);
            }
            did = true;
        }
        wave_sync();
        if (!__any(did) || __any(ovf)) break;
    }
    if (__any(ovf)) { if (lane == 0) L.misc[WM_OVF] = 1; return false; }
    // ---- forward: every state (previous edge, distance) reachable through states that are not pruned; a state is pruned when
    //      its node cannot reach node2 within what is left of the range
    int start_slot = -1;
    if (lane == 0) { start_slot = state_add(L, start_edge, (u32)dist0, A.qcap); ovf |= start_slot < 0; }
    start_slot = __shfl(start_slot, 0);
    wave_sync();
    u32 head = 0;
    for (;;) {
        const u32 tail = min(L.misc[WM_QTAIL], A.qcap);
        if (head >= tail || __any(ovf)) break;
        const u32 i = head + lane;
        if (i < tail) {
            const u32 s = L.q[i];
            const u64 key = L.st_key[s];
            const u32 pe = (u32)(key >> 16), d = (u32)(key & 0xffffu);
            const u32 n1 = pe == NONE ? node0 : g.e_end[pe];
            const u32 r = reach_get(L, n1);
            const bool pruned = r == W_ABSENT || (u64)d + dist2 + r > (u64)hi;
            L.st_flag[s] = pruned ? WF_PRUNED : 0;
            if (!pruned)
                for (int b = 0; b < 4; b++) {
                    const u32 e = g.out_edge[(u64)n1 * 4 + b];
                    if (e == NONE) continue;
                    const u64 d2 = (u64)d + g.e_len[e];
                    if (d2 <= (u64)hi) ovf |= state_add(L, e, (u32)d2, A.qcap) < 0;        // (beyond hi the child is pruned: res false, nothing to record)
                }
        }
        head = min(head + 64u, tail);
        wave_sync();
    }
    if (__any(ovf)) { if (lane == 0) L.misc[WM_OVF] = 1; return false; }
    // ---- backward: res(state) = arrival in range, or a child that succeeds (children have larger d): to a fixpoint
    const u32 ntot = min(L.misc[WM_QTAIL], A.qcap);
    for (u32 round = 0; round <= ntot; round++) {
        bool ch = false;
        for (u32 i = lane; i < ntot; i += 64) {
            const u32 s = L.q[i];
            const uint8_t f = L.st_flag[s];
            if (f & (WF_PRUNED | WF_RES)) continue;
            const u64 key = L.st_key[s];
            const u32 pe = (u32)(key >> 16), d = (u32)(key & 0xffffu);
            const u32 n1 = pe == NONE ? node0 : g.e_end[pe];
            bool cur = n1 == node2 && d + dist2 >= lo && d + dist2 <= hi;
            for (int b = 0; b < 4 && !cur; b++) {
                const u32 e = g.out_edge[(u64)n1 * 4 + b];
                if (e == NONE) continue;
                const u64 d2 = (u64)d + g.e_len[e];
                if (d2 > (u64)hi) continue;
                const int c = state_find(L, e, (u32)d2);
                if (c >= 0 && (L.st_flag[c] & WF_RES)) cur = true;
            }
            if (cur) { L.st_flag[s] = f | WF_RES; ch = true; }
        }
        wave_sync();
        if (!__any(ch)) break;
    }
    // ---- the supported pairs: (state's edge, end edge) on arrival, (state's edge, child's edge) for every child that succeeds
    for (u32 i = lane; i < ntot; i += 64) {
        const u32 s = L.q[i];
        if (L.st_flag[s] & WF_PRUNED) continue;
        const u64 key = L.st_key[s];
        const u32 pe = (u32)(key >> 16), d = (u32)(key & 0xffffu);
        if (pe == NONE) continue;
        const u32 n1 = g.e_end[pe];
        if (n1 == node2 && d + dist2 >= lo && d + dist2 <= hi && end_edge != NONE) ovf |= !pair_add(L, ((u64)pe << 32) | end_edge, A.pmax);
        for (int b = 0; b < 4; b++) {
            const u32 e = g.out_edge[(u64)n1 * 4 + b];
            if (e == NONE) continue;
            const u64 d2 = (u64)d + g.e_len[e];
            if (d2 > (u64)hi) continue;
            const int c = state_find(L, e, (u32)d2);
            if (c >= 0 && (L.st_flag[c] & WF_RES)) ovf |= !pair_add(L, ((u64)pe << 32) | e, A.pmax);
        }
    }
    wave_sync();
    if (__any(ovf)) { if (lane == 0) L.misc[WM_OVF] = 1; return false; }
    return start_slot >= 0 && (L.st_flag[start_slot] & WF_RES) != 0;
}

// :219-247 for every pair orientation o: positions p1 = vals[off[2o] .. off[2o+1]), p2 = vals[off[2o+1] .. off[2o+2])
__global__ __launch_bounds__(BLOCK) void k_walk_pairs(GraphView g, const unsigned long long *__restrict__ in_off, const u32 *__restrict__ in_list,
                                                      const unsigned long long *__restrict__ off, const u64 *__restrict__ vals, u64 norient, WalkArgs A, SupView sup,
                                                      u32 *ov_list, u32 ov_cap, u32 *ov_n) {
    __shared__ WaveLds lds[BLOCK / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    WaveLds &L = lds[wave];
    u32 n_walked = 0, n_bad = 0;
    for (u64 o = (u64)blockIdx.x * (BLOCK / 64) + wave; o < norient; o += (u64)gridDim.x * (BLOCK / 64)) {
        const u64 a0 = off[2 * o], a1 = off[2 * o + 1], a2 = off[2 * o + 2];
        const u64 n1 = a1 - a0, n2 = a2 - a1;
        if (n1 == 0 || n2 == 0) continue;                    // (`if !list.isEmpty` :231)
        // annotate :192-206: the mates on ONE edge at a distance inside the range
        bool same = false;
        for (u64 t = lane; t < n1 * n2; t += 64) {
            const u64 va = vals[a0 + t / n2], vb = vals[a1 + t % n2];
            if (!GK_POS_IS_EDGE(va) || !GK_POS_IS_EDGE(vb) || GK_POS_ID(va) != GK_POS_ID(vb)) continue;
            const long d = (long)GK_POS_DIST(vb) - (long)GK_POS_DIST(va) + A.k;
            if (d >= A.lo && d <= A.hi) same = true;
        }
        if (__any(same)) continue;
        for (u32 i = lane; i < W_PCAP; i += 64) L.pset[i] = ~0ull;
        if (lane == 0) { L.misc[WM_NPAIRS] = 0; L.misc[WM_OVF] = 0; }
        wave_sync();
        bool good = false;
        for (u64 i = 0; i < n1 && !L.misc[WM_OVF]; i++)
            for (u64 j = 0; j < n2 && !L.misc[WM_OVF]; j++) {
                good |= walk_one(L, g, in_off, in_list, vals[a0 + i], vals[a1 + j], A);
                wave_sync();
            }
        if (L.misc[WM_OVF]) {                                // a set overflowed: the host walker takes this orientation, nothing is counted here
            if (lane == 0) { const u32 at = atomicAdd(ov_n, 1u); if (at < ov_cap) ov_list[at] = (u32)o; }
            continue;
        }
        for (u32 i = lane; i < W_PCAP; i += 64) if (L.pset[i] != ~0ull) sup_add(sup, L.pset[i], 1u);       // counter.incrementAndGet() once per orientation  :235-241
        n_walked++;
        if (!good) n_bad++;
        wave_sync();
    }
    if (lane == 0) {
        if (n_bad) atomicAdd(&sup.ctr[1], (unsigned long long)n_bad);
        if (n_walked) atomicAdd(&sup.ctr[2], (unsigned long long)n_walked);
    }
}

// =============================================================================================
// host side
// =============================================================================================
namespace {

struct HostGraph {
    u64 n_nodes = 0, n_edges = 0;
    std::vector<u32> e_start, e_end, out_edge, in_off, in_list;
    std::vector<u64> e_len;
    std::vector<uint8_t> e_alive, node_alive;
};

int graph_snapshot(gk_graph *g, HostGraph &H) {
    gk_ctx *ctx = g->ctx;
    const GraphView &v = g->v;
    H.n_nodes = v.n_nodes; H.n_edges = v.n_edges;
    H.e_start.resize(H.n_edges); H.e_end.resize(H.n_edges); H.e_len.resize(H.n_edges); H.e_alive.resize(H.n_edges);
    H.out_edge.resize(H.n_nodes * 4); H.node_alive.resize(H.n_nodes);
    GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (H.n_edges) {
        GK_HIP(ctx, hipMemcpy(H.e_start.data(), v.e_start, H.n_edges * 4, hipMemcpyDeviceToHost));
        GK_HIP(ctx, hipMemcpy(H.e_end.data(), v.e_end, H.n_edges * 4, hipMemcpyDeviceToHost));
        GK_HIP(ctx, hipMemcpy(H.e_len.data(), v.e_len, H.n_edges * 8, hipMemcpyDeviceToHost));
        GK_HIP(ctx, hipMemcpy(H.e_alive.data(), v.e_alive, H.n_edges, hipMemcpyDeviceToHost));
    }
    if (H.n_nodes) {
        GK_HIP(ctx, hipMemcpy(H.out_edge.data(), v.out_edge, H.n_nodes * 16, hipMemcpyDeviceToHost));
        GK_HIP(ctx, hipMemcpy(H.node_alive.data(), v.node_alive, H.n_nodes, hipMemcpyDeviceToHost));
    }
    // in-edge lists (Node.inEdgeIds), CSR by end node
    H.in_off.assign(H.n_nodes + 1, 0);
    for (u64 e = 0; e < H.n_edges; e++) if (H.e_alive[e]) H.in_off[H.e_end[e] + 1]++;
    for (u64 n = 0; n < H.n_nodes; n++) H.in_off[n + 1] += H.in_off[n];
    H.in_list.resize(H.in_off[H.n_nodes]);
    std::vector<u32> cur(H.in_off.begin(), H.in_off.end() - 1);
    for (u64 e = 0; e < H.n_edges; e++) if (H.e_alive[e]) H.in_list[cur[H.e_end[e]]++] = (u32)e;
    return GK_OK;
}

int graph_snapshot_cached(gk_graph *g, const HostGraph **out) {
    if (!g->snap || g->snap_epoch != g->epoch) {
        auto h = std::make_shared<HostGraph>();
        if (int rc = graph_snapshot(g, *h)) return rc;
        g->snap = h;
        g->snap_epoch = g->epoch;
    }
    *out = static_cast<const HostGraph *>(g->snap.get());
    return GK_OK;
}

struct Pos { bool is_edge; u32 id; u32 dist; };
inline Pos decode_pos(u64 v) { return Pos{GK_POS_IS_EDGE(v), GK_POS_ID(v), GK_POS_IS_EDGE(v) ? GK_POS_DIST(v) : 0u}; }

// one (pos1, pos2) of WalkingActor.receive (:78-125): appends the supported (edge, edge) pairs, returns `good`
struct Walker {
    const HostGraph &G;
    const int lo, hi;
    // scratch reused between walks
    std::unordered_map<u32, int> reach;
    std::vector<std::vector<u32>> rq;                       // reach: nodes by distance
    struct State { u32 pe; u32 d; bool pruned, res; };
    std::vector<State> states;                              // in order of d
    std::unordered_map<u64, u32> index;                     // (pe, d) -> position in `states`
    std::vector<std::vector<u32>> sq;                       // states by distance
    std::vector<u32> rq_touched, sq_touched;                // the distances whose buckets the last walk filled
    Walker(const HostGraph &g, int lo_, int hi_) : G(g), lo(lo_), hi(hi_), rq(hi_ + 1), sq(hi_ + 1) {}

    bool walk(Pos p1, Pos p2, std::vector<u64> &pairs) {
        const u32 node2 = p2.is_edge ? G.e_start[p2.id] : p2.id;
        const u32 dist2 = p2.is_edge ? p2.dist : 0u;
        const u32 end_edge = p2.is_edge ? p2.id : NONE;
        const u32 start_edge = p1.is_edge ? p1.id : NONE;
        const u32 node0 = p1.is_edge ? G.e_end[p1.id] : p1.id;
        const u64 dist0 = p1.is_edge ? G.e_len[p1.id] - p1.dist : 0;
        if (dist0 > (u64)hi) return false;                  // (the reference finds this out after `reachable`; nothing is recorded either way)
        // ---- reachable(node2) :43-72: shortest distance back along in-edges, <= hi (edge lengths >= 1: buckets by distance)
        reach.clear();
        for (u32 d : rq_touched) rq[d].clear();
        rq_touched.clear();
        rq[0].push_back(node2);
        rq_touched.push_back(0);
        for (int d = 0; d <= hi; d++)
            for (size_t i = 0; i < rq[d].size(); i++) {
                const u32 u = rq[d][i];
                if (!reach.emplace(u, d).second) continue;
                for (u32 j = G.in_off[u]; j < G.in_off[u + 1]; j++) {
                    const u32 e = G.in_list[j];
                    const u64 d2 = (u64)d + G.e_len[e];
                    if (d2 <= (u64)hi) { if (rq[d2].empty()) rq_touched.push_back((u32)d2); rq[d2].push_back(G.e_start[e]); }
                }
            }
        // ---- forward: states (previous edge, distance) in order of distance
        states.clear(); index.clear();
        for (u32 d : sq_touched) sq[d].clear();
        sq_touched.clear();
        auto node_of = [&](u32 pe) { return pe == NONE ? node0 : G.e_end[pe]; };
        auto add_state = [&](u32 pe, u32 d) {
            const u64 key = ((u64)pe << 32) | d;
            if (index.count(key)) return;
            index.emplace(key, 0u);
            if (sq[d].empty()) sq_touched.push_back(d);
            sq[d].push_back(pe);
        };
        add_state(start_edge, (u32)dist0);
        for (int d = (int)dist0; d <= hi; d++)
            for (size_t i = 0; i < sq[d].size(); i++) {
                const u32 pe = sq[d][i], n1 = node_of(pe);
                auto r = reach.find(n1);
                const bool pruned = r == reach.end() || (u64)d + dist2 + (u64)r->second > (u64)hi;
                index[((u64)pe << 32) | (u32)d] = (u32)states.size();
                states.push_back(State{pe, (u32)d, pruned, false});
                if (pruned) continue;
                for (int b = 0; b < 4; b++) {
                    const u32 e = G.out_edge[(u64)n1 * 4 + b];
                    if (e == NONE) continue;
                    const u64 d2 = (u64)d + G.e_len[e];
                    if (d2 <= (u64)hi) add_state(e, (u32)d2);       // (beyond hi the child is pruned: res false, nothing to record)
                }
            }
        // ---- backward: res(state) = arrival in range, or a child that succeeds (children have larger d)
        for (size_t si = states.size(); si-- > 0;) {
            State &s = states[si];
            if (s.pruned) continue;
            const u32 n1 = node_of(s.pe);
            bool cur = false;
            if (n1 == node2 && (int)(s.d + dist2) >= lo && (int)(s.d + dist2) <= hi) {
                if (s.pe != NONE && end_edge != NONE) pairs.push_back(((u64)s.pe << 32) | end_edge);
                cur = true;
            }
            for (int b = 0; b < 4; b++) {
                const u32 e = G.out_edge[(u64)n1 * 4 + b];
                if (e == NONE) continue;
                const u64 d2 = (u64)s.d + G.e_len[e];
                if (d2 > (u64)hi) continue;
                const State &c = states[index[((u64)e << 32) | (u32)d2]];
                if (c.res) {
                    if (s.pe != NONE) pairs.push_back(((u64)s.pe << 32) | e);
                    cur = true;
                }
            }
            s.res = cur;
        }
        return states.empty() ? false : states[index[((u64)start_edge << 32) | (u32)dist0]].res;
    }
};

// the first k bases of a `.bin` record as (lo, hi) — the first 2k bits of its payload, LSB first (DNASeq.scala:285-303) —
// and its reverse complement.  `avail` = payload bytes that may be read (>= ceil(k / 4)).
inline void first_kmer(const uint8_t *payload, size_t avail, int k, u64 &lo, u64 &hi) {
    u64 w[2] = {0, 0};
    memcpy(w, payload, std::min<size_t>(16, avail));
    if (k <= 32) { lo = k == 32 ? w[0] : w[0] & ((1ull << (2 * k)) - 1ull); hi = 0; }
    else { lo = w[0]; hi = k == 64 ? w[1] : w[1] & ((1ull << (2 * (k - 32))) - 1ull); }
}
// the 32 two-bit groups of a word in reverse order
inline u64 reverse_groups(u64 x) {
    x = __builtin_bswap64(x);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
    return ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
}
// complement (A<->T, G<->C = b ^ 3, Base.scala:6-23), then reverse; the complemented padding ends up below bit 0 and is shifted out
inline void revcomp_host(u64 lo, u64 hi, int k, u64 &rlo, u64 &rhi) {
    if (k <= 32) { rlo = reverse_groups(~lo) >> (64 - 2 * k); rhi = 0; return; }
    const u64 nhi = reverse_groups(~lo), nlo = reverse_groups(~hi);       // the 128-bit value reversed: words swapped
    const int s = 128 - 2 * k;                                             // 0 <= s < 64
    rlo = s ? (nlo >> s) | (nhi << (64 - s)) : nlo;
    rhi = nhi >> s;
}

}  // namespace


namespace {
struct Tmp {      // device arrays of one call, freed together
    gk_ctx *ctx;
    std::vector<void *> ptrs;
    explicit Tmp(gk_ctx *c) : ctx(c) {}
    ~Tmp() { for (void *p : ptrs) if (p) (void)hipFree(p); }
    template <class T> hipError_t get(T **p, u64 n) {
        hipError_t e = hipMalloc((void **)p, std::max<u64>(n, 1) * sizeof(T));
        if (e == hipSuccess) ptrs.push_back(*p);
        return e;
    }
};

SupView sup_view(const gk_support *s) { return SupView{s->d_keys, s->d_cnt, s->cap - 1, s->d_ctr}; }

// room for `want` distinct pairs at load <= 0.5 (a power of two of slots); contents are kept
int support_reserve(gk_support *s, u64 want) {
    gk_ctx *ctx = s->ctx;
    const u64 need = pow2ceil(std::max<u64>(2 * want + 1024, 4096));
    if (s->cap >= need) return GK_OK;
    u64 *nk = nullptr;
    u32 *nc = nullptr;
    hipError_t e = hipMalloc((void **)&nk, need * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&nc, need * 4);
    if (e == hipSuccess) e = hipMemsetAsync(nk, 0xff, need * 8, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(nc, 0, need * 4, ctx->stream);
    if (e != hipSuccess) { if (nk) (void)hipFree(nk); if (nc) (void)hipFree(nc); return hip_fail(ctx, e, "gk_support: table"); }
    if (s->cap) {
        unsigned long long distinct = 0;
        GK_HIP(ctx, hipMemcpyAsync(&distinct, s->d_ctr, 8, hipMemcpyDeviceToHost, ctx->stream));
        GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        GK_HIP(ctx, hipMemsetAsync(s->d_ctr, 0, 8, ctx->stream));                   // (the rehash counts the distinct pairs again)
        hipLaunchKernelGGL(k_sup_rehash, dim3(ggrid(ctx, s->cap)), dim3(BLOCK), 0, ctx->stream, s->d_keys, s->d_cnt, s->cap, SupView{nk, nc, need - 1, s->d_ctr});
        GK_HIP(ctx, hipGetLastError());
        GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(s->d_keys); (void)hipFree(s->d_cnt);
        (void)distinct;
    }
    s->d_keys = nk; s->d_cnt = nc; s->cap = need;
    return GK_OK;
}

int support_counters(const gk_support *s, unsigned long long *h4) {
    gk_ctx *ctx = s->ctx;
    GK_HIP(ctx, hipMemcpyAsync(h4, s->d_ctr, 32, hipMemcpyDeviceToHost, ctx->stream));
    GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GK_OK;
}

// the counts as a host map (the split reads them by key)
int support_to_host(gk_support *s) {
    if (s->host_valid) return GK_OK;
    gk_ctx *ctx = s->ctx;
    s->paths.clear();
    if (s->cap) {
        std::vector<u64> k(s->cap);
        std::vector<u32> c(s->cap);
        GK_HIP(ctx, hipMemcpyAsync(k.data(), s->d_keys, s->cap * 8, hipMemcpyDeviceToHost, ctx->stream));
        GK_HIP(ctx, hipMemcpyAsync(c.data(), s->d_cnt, s->cap * 4, hipMemcpyDeviceToHost, ctx->stream));
        GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (u64 i = 0; i < s->cap; i++) if (k[i] != SUP_EMPTY) s->paths.emplace(k[i], c[i]);
    }
    s->host_valid = true;
    return GK_OK;
}
}  // namespace

extern "C" {

int gk_support_create(gk_ctx *ctx, gk_support **out) {
    if (!ctx || !out) return fail(ctx, GK_E_INVALID, "gk_support_create: null argument");
    *out = nullptr;
    GK_HIP(ctx, hipSetDevice(ctx->device));
    gk_support *s = new gk_support();
    s->ctx = ctx;
    hipError_t e = hipMalloc((void **)&s->d_ctr, 64);
    if (e == hipSuccess) e = hipMemsetAsync(s->d_ctr, 0, 64, ctx->stream);
    if (e != hipSuccess) { delete s; return hip_fail(ctx, e, "gk_support_create"); }
    *out = s;
    return GK_OK;
}
void gk_support_destroy(gk_support *s) {
    if (!s) return;
    gk_ctx *ctx = s->ctx;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (s->d_keys) (void)hipFree(s->d_keys);
    if (s->d_cnt) (void)hipFree(s->d_cnt);
    if (s->d_ctr) (void)hipFree(s->d_ctr);
    delete s;
}
int gk_support_size(const gk_support *s, uint64_t *pairs, uint64_t *bad_pairs, uint64_t *walked) {
    if (!s) return fail(nullptr, GK_E_INVALID, "null support handle");
    unsigned long long h[4] = {0, 0, 0, 0};
    if (int rc = support_counters(s, h)) return rc;
    if (pairs) *pairs = h[0];
    if (bad_pairs) *bad_pairs = h[1];
    if (walked) *walked = h[2];
    return GK_OK;
}
int gk_support_last_ms(const gk_support *s, float *ms5) {
    if (!s || !ms5) return fail(nullptr, GK_E_INVALID, "gk_support_last_ms: null argument");
    for (int i = 0; i < 5; i++) ms5[i] = s->last_ms[i];
    return GK_OK;
}
int gk_support_export(const gk_support *s, uint32_t *e1, uint32_t *e2, uint32_t *count, uint64_t cap, uint64_t *n) {
    if (!s) return fail(nullptr, GK_E_INVALID, "null support handle");
    if (int rc = support_to_host(const_cast<gk_support *>(s))) return rc;
    if (n) *n = s->paths.size();
    if (s->paths.size() > cap) return fail(s->ctx, GK_E_CAPACITY, "gk_support_export: need room for " + std::to_string(s->paths.size()) + " pairs");
    u64 i = 0;
    for (const auto &kv : s->paths) { e1[i] = (u32)(kv.first >> 32); e2[i] = (u32)kv.first; count[i] = kv.second; i++; }
    return GK_OK;
}

int gk_graph_id_bounds(gk_graph *g, uint64_t *node_ids, uint64_t *edge_ids) {
    if (int rc = check_graph(g)) return rc;
    if (node_ids) *node_ids = g->v.n_nodes;
    if (edge_ids) *edge_ids = g->v.n_edges;
    return GK_OK;
}

// GraphSimplifier.scala:188-247 over the first `npairs` pairs of a `.bin` stream (two records per pair)
int gk_graph_walk_pairs(gk_graph *g, gk_vmap *positions, gk_support *sup, const uint8_t *bin, size_t nbytes, uint64_t npairs, int range_lo,
                        int range_hi) {
    if (int rc = check_graph(g)) return rc;
    gk_ctx *ctx = g->ctx;
    if (!positions || !sup || (!bin && nbytes)) return fail(ctx, GK_E_INVALID, "gk_graph_walk_pairs: null argument");
    if (vmap_ctx(positions) != ctx || sup->ctx != ctx) return fail(ctx, GK_E_INVALID, "gk_graph_walk_pairs: the position map and the support must live on the graph's context");
    if (vmap_k(positions) != g->k) return fail(ctx, GK_E_KLEN, "gk_graph_walk_pairs: the position map has another k");
    if (range_lo < 0 || range_hi < range_lo || range_hi > 65535) return fail(ctx, GK_E_INVALID, "gk_graph_walk_pairs: range must satisfy 0 <= lo <= hi <= 65535");
    const int k = g->k, W = g->W;
    auto now = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
    const double t_begin = now();
    Tmp tmp(ctx);
    // ---- the pairs whose mates both hold k bases (:213), their four keys.  A stream of equal-length records (what a
    //      sequencer's run is) goes to the device as it is and is cut there; a ragged one is walked here.
    u64 *d_lo = nullptr, *d_hi = nullptr;
    u64 nq = 0;
    bool cut = false;
    if (npairs && nbytes && bin[0] >= k) {
        const int l0 = bin[0];
        const size_t rb = 1 + (size_t)(l0 + 3) / 4;
        if (nbytes >= 2 * npairs * rb) {
            uint8_t *d_bin = nullptr;
            u32 *d_rag = nullptr, h_rag = 0;
            hipError_t e = tmp.get(&d_bin, 2 * npairs * rb + 16);
            if (e == hipSuccess) e = tmp.get(&d_lo, 4 * npairs);
            if (e == hipSuccess && W == 2) e = tmp.get(&d_hi, 4 * npairs);
            if (e == hipSuccess) e = tmp.get(&d_rag, 1);
            if (e == hipSuccess) e = hipMemsetAsync(d_rag, 0, 4, ctx->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(d_bin, bin, 2 * npairs * rb, hipMemcpyHostToDevice, ctx->stream);
            if (e != hipSuccess) return hip_fail(ctx, e, "gk_graph_walk_pairs: upload");
            if (W == 1) hipLaunchKernelGGL(k_pair_keys<1>, dim3(ggrid(ctx, npairs)), dim3(BLOCK), 0, ctx->stream, d_bin, npairs, (u32)rb, l0, k, d_lo, d_hi, d_rag);
            else hipLaunchKernelGGL(k_pair_keys<2>, dim3(ggrid(ctx, npairs)), dim3(BLOCK), 0, ctx->stream, d_bin, npairs, (u32)rb, l0, k, d_lo, d_hi, d_rag);
            GK_HIP(ctx, hipGetLastError());
            GK_HIP(ctx, hipMemcpyAsync(&h_rag, d_rag, 4, hipMemcpyDeviceToHost, ctx->stream));
            GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (!h_rag) { cut = true; nq = 4 * npairs; }
        }
    }
    if (!cut) {
        std::vector<u64> klo, khi;
        klo.reserve((size_t)std::min<uint64_t>(npairs, nbytes / 2) * 4); khi.reserve(klo.capacity());
        size_t pos = 0;
        for (uint64_t p = 0; p < npairs && pos < nbytes; p++) {
            const uint8_t *r1 = bin + pos;
            const int l1 = r1[0];
            pos += 1 + (size_t)(l1 + 3) / 4;
            if (pos >= nbytes) return fail(ctx, GK_E_FORMAT, "gk_graph_walk_pairs: the stream ends inside a pair");
            const uint8_t *r2 = bin + pos;
            const int l2 = r2[0];
            pos += 1 + (size_t)(l2 + 3) / 4;
            if (pos > nbytes) return fail(ctx, GK_E_FORMAT, "gk_graph_walk_pairs: the stream ends inside a record");
            if (l1 < k || l2 < k) continue;
            u64 alo, ahi, blo, bhi, ralo, rahi, rblo, rbhi;
            first_kmer(r1 + 1, (size_t)(l1 + 3) / 4, k, alo, ahi); first_kmer(r2 + 1, (size_t)(l2 + 3) / 4, k, blo, bhi);
            revcomp_host(alo, ahi, k, ralo, rahi); revcomp_host(blo, bhi, k, rblo, rbhi);
            // f1 = getAll(p1.take(k)), f2 = getAll(p2.take(k).revComplement), f3 = getAll(p2.take(k)), f4 = getAll(p1.take(k).revComplement)
            const u64 lo4[4] = {alo, rblo, blo, ralo}, hi4[4] = {ahi, rbhi, bhi, rahi};
            klo.insert(klo.end(), lo4, lo4 + 4);
            khi.insert(khi.end(), hi4, hi4 + 4);
        }
        nq = klo.size();
        if (nq) {
            d_lo = d_hi = nullptr;
            hipError_t e = tmp.get(&d_lo, nq);
            if (e == hipSuccess && W == 2) e = tmp.get(&d_hi, nq);
            if (e == hipSuccess) e = hipMemcpyAsync(d_lo, klo.data(), nq * 8, hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess && d_hi) e = hipMemcpyAsync(d_hi, khi.data(), nq * 8, hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);          // (klo / khi die with this scope)
            if (e != hipSuccess) return hip_fail(ctx, e, "gk_graph_walk_pairs: keys");
        }
    }
    if (nq == 0) return GK_OK;
    const double t_keys = now();
    // ---- the four getAll of every pair as ONE batch, results left in HBM as CSR
    u32 *d_cnt = nullptr;
    unsigned long long *d_off = nullptr;
    u64 *d_sums = nullptr, *d_vals = nullptr;
    unsigned long long total = 0;
    {
        hipError_t e = tmp.get(&d_cnt, nq);
        if (e == hipSuccess) e = tmp.get(&d_off, nq + 1);
        if (e == hipSuccess) e = tmp.get(&d_sums, nq / SCAN_CHUNK + 2);
        if (e != hipSuccess) return hip_fail(ctx, e, "gk_graph_walk_pairs: lookup arrays");
        if (int rc = vmap_get_all_dev(positions, d_lo, d_hi, nq, nullptr, d_cnt, nullptr)) return rc;
        GK_HIP(ctx, scan_counts(ctx, d_cnt, nq, d_off, d_sums));
        GK_HIP(ctx, hipMemcpyAsync(&total, d_off + nq, 8, hipMemcpyDeviceToHost, ctx->stream));
        GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        GK_HIP(ctx, tmp.get(&d_vals, total));
        if (total) { if (int rc = vmap_get_all_dev(positions, d_lo, d_hi, nq, d_off, d_cnt, d_vals)) return rc; }
    }
    GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const double t_lookup = now();
    // ---- a position must name something of THIS graph; the in-edge lists of its current state
    const GraphView &v = g->v;
    u32 *d_flag = nullptr, h_flag[2] = {0, 0};                 // [0] bad position, [1] overflow count
    GK_HIP(ctx, tmp.get(&d_flag, 2));
    GK_HIP(ctx, hipMemsetAsync(d_flag, 0, 8, ctx->stream));
    if (total) {
        hipLaunchKernelGGL(k_check_positions, dim3(ggrid(ctx, total)), dim3(BLOCK), 0, ctx->stream, v, d_vals, (u64)total, d_flag);
        GK_HIP(ctx, hipGetLastError());
    }
    unsigned long long *d_in_off = nullptr;
    u32 *d_in_cnt = nullptr, *d_in_list = nullptr;
    u64 *d_sums2 = nullptr;
    {
        hipError_t e = tmp.get(&d_in_off, v.n_nodes + 1);
        if (e == hipSuccess) e = tmp.get(&d_in_cnt, v.n_nodes);
        if (e == hipSuccess) e = tmp.get(&d_in_list, v.n_edges);
        if (e == hipSuccess) e = tmp.get(&d_sums2, v.n_nodes / SCAN_CHUNK + 2);
        if (e == hipSuccess) e = hipMemsetAsync(d_in_cnt, 0, std::max<u64>(v.n_nodes, 1) * 4, ctx->stream);
        if (e != hipSuccess) return hip_fail(ctx, e, "gk_graph_walk_pairs: in-edge lists");
        if (v.n_edges) hipLaunchKernelGGL(k_in_count, dim3(ggrid(ctx, v.n_edges)), dim3(BLOCK), 0, ctx->stream, v, d_in_cnt);
        GK_HIP(ctx, scan_counts(ctx, d_in_cnt, v.n_nodes, d_in_off, d_sums2));
        GK_HIP(ctx, hipMemsetAsync(d_in_cnt, 0, std::max<u64>(v.n_nodes, 1) * 4, ctx->stream));
        if (v.n_edges) hipLaunchKernelGGL(k_in_fill, dim3(ggrid(ctx, v.n_edges)), dim3(BLOCK), 0, ctx->stream, v, d_in_off, d_in_cnt, d_in_list);
        GK_HIP(ctx, hipGetLastError());
    }
    GK_HIP(ctx, hipMemcpyAsync(h_flag, d_flag, 4, hipMemcpyDeviceToHost, ctx->stream));
    GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (h_flag[0]) return fail(ctx, GK_E_STATE, "gk_graph_walk_pairs: the position map does not belong to this graph (rebuild it after edits)");
    const double t_snap = now();
    // ---- the walks: one wave per pair orientation (two per pair: (f1, f2) and (f3, f4)  :219)
    const u64 norient = nq / 2;
    if (int rc = support_reserve(sup, 4 * v.n_edges)) return rc;       // supported pairs are (in-edge, out-edge) of a node: <= 4 per edge
    sup->host_valid = false;
    const bool host_only = ctx->hook_pairs_host > 0;                    // ("pairs_host" = 1: every walk on host threads, the round-2 form; A/B)
    u32 *d_ov = nullptr;
    const u32 ov_cap = (u32)std::min<u64>(norient, 1u << 22);
    GK_HIP(ctx, tmp.get(&d_ov, ov_cap));
    if (!host_only) {
        const int grid = (int)std::min<u64>((norient + BLOCK / 64 - 1) / (BLOCK / 64), (u64)ctx->cu_count * 16);
        hipLaunchKernelGGL(k_walk_pairs, dim3(std::max(grid, 1)), dim3(BLOCK), 0, ctx->stream, v, d_in_off, d_in_list, d_off, d_vals, norient,
                           ctx->hook_pairs_small_sets > 0 ? WalkArgs{k, range_lo, range_hi, 6u, 10u, 3u}       // (test: most walks outgrow their sets -> host walker)
                                                          : WalkArgs{k, range_lo, range_hi, W_RMAX, W_QCAP, W_PMAX},
                           sup_view(sup), d_ov, ov_cap, d_flag + 1);
        GK_HIP(ctx, hipGetLastError());
        GK_HIP(ctx, hipMemcpyAsync(h_flag + 1, d_flag + 1, 4, hipMemcpyDeviceToHost, ctx->stream));
        GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    const double t_walks = now();
    // ---- what the device could not hold (or everything, under the A/B switch): the host walker over a snapshot
    std::vector<u32> todo;
    if (host_only) { todo.resize(norient); for (u64 o = 0; o < norient; o++) todo[o] = (u32)o; }
    else if (h_flag[1]) {
        if (h_flag[1] > ov_cap) return fail(ctx, GK_E_CAPACITY, "gk_graph_walk_pairs: more than 2^22 walks outgrew the device's sets; pass fewer pairs per call");
        todo.resize(h_flag[1]);
        GK_HIP(ctx, hipMemcpy(todo.data(), d_ov, (size_t)h_flag[1] * 4, hipMemcpyDeviceToHost));
    }
    sup->last_overflow = host_only ? 0 : todo.size();
    if (!todo.empty()) {
        std::vector<unsigned long long> off(nq + 1);
        std::vector<u64> vals(std::max<u64>(total, 1));
        GK_HIP(ctx, hipMemcpy(off.data(), d_off, (nq + 1) * 8, hipMemcpyDeviceToHost));
        if (total) GK_HIP(ctx, hipMemcpy(vals.data(), d_vals, (size_t)total * 8, hipMemcpyDeviceToHost));
        const HostGraph *Hp = nullptr;
        if (int rc2 = graph_snapshot_cached(g, &Hp)) return rc2;
        const HostGraph &H = *Hp;
        const unsigned nthreads = (unsigned)std::max<u64>(1, std::min<u64>({(u64)std::thread::hardware_concurrency(), 16, todo.size() / 64 + 1}));
        std::vector<std::unordered_map<u64, u32>> local(nthreads);
        std::vector<u64> bad(nthreads, 0), walked(nthreads, 0);
        auto work = [&](unsigned t) {
            Walker w(H, range_lo, range_hi);
            std::vector<u64> pairs;
            for (u64 ti = t; ti < todo.size(); ti += nthreads) {
                const u64 o = todo[ti];
                const u64 *p1 = vals.data() + off[2 * o], *p2 = vals.data() + off[2 * o + 1];
                const u64 n1 = off[2 * o + 1] - off[2 * o], n2 = off[2 * o + 2] - off[2 * o + 1];
                bool same_edge = false;                          // annotate :192-206
                for (u64 i = 0; i < n1 && !same_edge; i++) {
                    const Pos a = decode_pos(p1[i]);
                    if (!a.is_edge) continue;
                    for (u64 j = 0; j < n2; j++) {
                        const Pos b = decode_pos(p2[j]);
                        const long d = (long)b.dist - (long)a.dist + k;
                        if (b.is_edge && a.id == b.id && d >= range_lo && d <= range_hi) { same_edge = true; break; }
                    }
                }
                if (same_edge || n1 == 0 || n2 == 0) continue;   // (`if !list.isEmpty` :231)
                pairs.clear();
                bool good = false;
                for (u64 i = 0; i < n1; i++)
                    for (u64 j = 0; j < n2; j++) good |= w.walk(decode_pos(p1[i]), decode_pos(p2[j]), pairs);
                std::sort(pairs.begin(), pairs.end());
                pairs.erase(std::unique(pairs.begin(), pairs.end()), pairs.end());
                for (u64 pr : pairs) local[t][pr]++;             // counter.incrementAndGet() once per pair orientation  :235-241
                if (!good) bad[t]++;
                walked[t]++;
            }
        };
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < nthreads; t++) pool.emplace_back(work, t);
        work(0);
        for (auto &th : pool) th.join();
        std::unordered_map<u64, u32> merged;
        u64 nbad = 0, nwalked = 0;
        for (unsigned t = 0; t < nthreads; t++) {
            for (const auto &kv : local[t]) merged[kv.first] += kv.second;
            nbad += bad[t]; nwalked += walked[t];
        }
        std::vector<u64> mk; std::vector<u32> mc;
        mk.reserve(merged.size()); mc.reserve(merged.size());
        for (const auto &kv : merged) { mk.push_back(kv.first); mc.push_back(kv.second); }
        if (!mk.empty()) {
            u64 *d_mk = nullptr; u32 *d_mc = nullptr;
            GK_HIP(ctx, tmp.get(&d_mk, mk.size()));
            GK_HIP(ctx, tmp.get(&d_mc, mc.size()));
            GK_HIP(ctx, hipMemcpy(d_mk, mk.data(), mk.size() * 8, hipMemcpyHostToDevice));
            GK_HIP(ctx, hipMemcpy(d_mc, mc.data(), mc.size() * 4, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_sup_add_list, dim3(ggrid(ctx, mk.size())), dim3(BLOCK), 0, ctx->stream, d_mk, d_mc, (u64)mk.size(), sup_view(sup));
            GK_HIP(ctx, hipGetLastError());
        }
        unsigned long long add[2] = {nbad, nwalked}, cur[4];
        if (int rc = support_counters(sup, cur)) return rc;
        cur[1] += add[0]; cur[2] += add[1];
        GK_HIP(ctx, hipMemcpy(sup->d_ctr + 1, cur + 1, 16, hipMemcpyHostToDevice));
    }
    unsigned long long ctr[4];
    if (int rc = support_counters(sup, ctr)) return rc;
    if (ctr[3]) return fail(ctx, GK_E_CAPACITY, "gk_graph_walk_pairs: the support table filled up (internal sizing error)");
    const double t_end = now();
    sup->last_ms[0] = (float)(t_keys - t_begin); sup->last_ms[1] = (float)(t_lookup - t_keys); sup->last_ms[2] = (float)(t_snap - t_lookup);
    sup->last_ms[3] = (float)(t_walks - t_snap); sup->last_ms[4] = (float)(t_end - t_walks);
    return GK_OK;
}

}  // extern "C"

// ---- batch edits of the split (device)
__global__ __launch_bounds__(BLOCK) void k_add_nodes(GraphView g, u32 first, const u32 *src, u64 n) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) {
        const u32 v = first + (u32)i, s = src[i];            // graph.addNode(node.seq)  :306
        g.node_lo[v] = g.node_lo[s]; g.node_hi[v] = g.node_hi[s];
        g.node_alive[v] = 1;
        g.out_order[v] = 0; g.in_deg[v] = 0;
        for (int b = 0; b < 4; b++) g.out_edge[(u64)v * 4 + b] = NONE;
    }
}
__global__ __launch_bounds__(BLOCK) void k_move_ends(GraphView g, const u32 *edge, const u32 *node, u64 n) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) {
        const u32 e = edge[i], nn = node[i];                 // graph.replaceEnd(e, newNode)  :307, Graph.scala:204-209
        atomicSub(&g.in_deg[g.e_end[e]], 1u);
        atomicAdd(&g.in_deg[nn], 1u);
        g.e_end[e] = nn;
    }
}
__global__ __launch_bounds__(BLOCK) void k_move_starts(GraphView g, const u32 *edge, const u32 *node, u64 n) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) {
        const u32 e = edge[i], nn = node[i];                 // graph.replaceStart(e, newNode)  :308, Graph.scala:197-202
        const u32 old = g.e_start[e];
        const int b = g.e_first[e];
        if (atomicCAS(&g.out_edge[(u64)old * 4 + b], e, NONE) == e) {
            u32 seen = g.out_order[old], prev;
            do { prev = seen; seen = atomicCAS(&g.out_order[old], prev, order_remove(prev, b)); } while (seen != prev);
        }
        g.out_edge[(u64)nn * 4 + b] = e;                     // (a fresh node: no two of its edges share a first base)
        u32 seen = g.out_order[nn], prev;
        do { prev = seen; seen = atomicCAS(&g.out_order[nn], prev, order_append(prev, b)); } while (seen != prev);
        g.e_start[e] = nn;
    }
}
__global__ __launch_bounds__(BLOCK) void k_remove_edges_by_id(GraphView g, const u32 *edge, u64 n, unsigned long long *removed) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) {
        const u32 e = edge[i];                               // MapGraph.removeEdge :191-195
        if (e >= g.n_edges || !g.e_alive[e]) continue;
        const u32 v = g.e_start[e];
        const int b = g.e_first[e];
        if (atomicCAS(&g.out_edge[(u64)v * 4 + b], e, NONE) == e) {
            u32 seen = g.out_order[v], prev;
            do { prev = seen; seen = atomicCAS(&g.out_order[v], prev, order_remove(prev, b)); } while (seen != prev);
        }
        g.e_alive[e] = 0;
        atomicSub(&g.in_deg[g.e_end[e]], 1u);
        atomicAdd(removed, 1ull);
    }
}

extern "C" {

int gk_graph_remove_edges_by_id(gk_graph *g, const uint32_t *edge_ids, uint64_t n, uint64_t *removed) {
    if (int rc = check_graph(g)) return rc;
    gk_ctx *ctx = g->ctx;
    if (removed) *removed = 0;
    if (n == 0) return GK_OK;
    if (!edge_ids) return fail(ctx, GK_E_INVALID, "gk_graph_remove_edges_by_id: null argument");
    std::vector<u32> ids(edge_ids, edge_ids + n);            // (a Set in the reference: each id once)
    std::sort(ids.begin(), ids.end());
    ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
    u32 *d_e = nullptr;
    unsigned long long *d_rm = nullptr, h_rm = 0;
    hipError_t e = hipMalloc((void **)&d_e, ids.size() * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&d_rm, 8);
    if (e == hipSuccess) e = hipMemcpyAsync(d_e, ids.data(), ids.size() * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_rm, 0, 8, ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_remove_edges_by_id, dim3(ggrid(ctx, ids.size())), dim3(BLOCK), 0, ctx->stream, g->v, d_e, (u64)ids.size(), d_rm);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&h_rm, d_rm, 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_e); (void)hipFree(d_rm);
    if (e != hipSuccess) return hip_fail(ctx, e, "gk_graph_remove_edges_by_id");
    if (removed) *removed = h_rm;
    return graph_refresh_counts(g);
}

// GraphSimplifier.scala:272-316: per node with in- and out-edges the support matrix, its connected groups at `cutoff`; a
// group without an out-edge loses its in-edge, every other group moves to a copy of the node; out-edges no group reached
// are removed.  (simplifyGraph :318 is the caller's next call.)  Nodes that exist when the call starts are visited.
int gk_graph_split_by_support(gk_graph *g, const gk_support *sup, int cutoff, uint64_t *removed_edges, uint64_t *new_nodes) {
    if (int rc = check_graph(g)) return rc;
    gk_ctx *ctx = g->ctx;
    if (removed_edges) *removed_edges = 0;
    if (new_nodes) *new_nodes = 0;
    if (!sup) return fail(ctx, GK_E_INVALID, "gk_graph_split_by_support: null support handle");
    const HostGraph *Hp = nullptr;
    if (int rc = graph_snapshot_cached(g, &Hp)) return rc;
    const HostGraph &H = *Hp;
    if (int rc = support_to_host(const_cast<gk_support *>(sup))) return rc;       // (the counts live on the device: one download)
    std::vector<u32> to_remove, new_src, end_edge, end_node, start_edge, start_node;
    const u32 first_new = (u32)H.n_nodes;
    for (u64 v = 0; v < H.n_nodes; v++) {
        if (!H.node_alive[v]) continue;
        const u32 *in = H.in_list.data() + H.in_off[v];
        const int nin = (int)(H.in_off[v + 1] - H.in_off[v]);
        u32 out[4];
        int nout = 0;
        for (int b = 0; b < 4; b++) if (H.out_edge[v * 4 + b] != NONE) out[nout++] = H.out_edge[v * 4 + b];
        if (nin == 0 || nout == 0) continue;                                         // :273
        auto support = [&](int i, int j) -> u32 {
            auto it = sup->paths.find(((u64)in[i] << 32) | out[j]);
            return it == sup->paths.end() ? 0u : it->second;
        };
        std::vector<char> col_l(nin, 0);
        bool col_r[4] = {false, false, false, false};
        for (int i0 = 0; i0 < nin; i0++) {
            if (col_l[i0]) continue;
            // the group of in-edge i0: alternate between the two sides until nothing is added (dfsLeft / dfsRight :280-301)
            std::vector<int> l{ i0 }, r, todo_l{ i0 }, todo_r;
            col_l[i0] = 1;
            while (!todo_l.empty() || !todo_r.empty()) {
                if (!todo_l.empty()) {
                    const int i = todo_l.back(); todo_l.pop_back();
                    for (int j = 0; j < nout; j++) if (!col_r[j] && (int)support(i, j) >= cutoff) { col_r[j] = true; r.push_back(j); todo_r.push_back(j); }
                } else {
                    const int j = todo_r.back(); todo_r.pop_back();
                    for (int i = 0; i < nin; i++) if (!col_l[i] && (int)support(i, j) >= cutoff) { col_l[i] = 1; l.push_back(i); todo_l.push_back(i); }
                }
            }
            if (r.empty()) to_remove.push_back(in[i0]);                             // :304
            else {
                const u32 nn = first_new + (u32)new_src.size();
                new_src.push_back((u32)v);
                for (int i : l) { end_edge.push_back(in[i]); end_node.push_back(nn); }          // :307
                for (int j : r) { start_edge.push_back(out[j]); start_node.push_back(nn); }     // :308
            }
        }
        for (int j = 0; j < nout; j++) if (!col_r[j]) to_remove.push_back(out[j]);  // :311
    }
    GraphView &v = g->v;
    const u64 nnew = new_src.size();
    if (v.n_nodes + nnew >= (u64)NONE) return fail(ctx, GK_E_CAPACITY, "more than 2^32 graph nodes");
    if (nnew) {
        if (v.n_nodes + nnew > g->node_cap) { if (int rc = graph_grow_nodes(g, std::max<u64>(g->node_cap * 2, v.n_nodes + nnew))) return rc; }
        u32 *d_src = nullptr, *d_a = nullptr, *d_b = nullptr;
        const u64 nmv = std::max<u64>(end_edge.size(), start_edge.size());
        hipError_t e = hipMalloc((void **)&d_src, nnew * 4);
        if (e == hipSuccess) e = hipMalloc((void **)&d_a, std::max<u64>(nmv, 1) * 4);
        if (e == hipSuccess) e = hipMalloc((void **)&d_b, std::max<u64>(nmv, 1) * 4);
        if (e == hipSuccess) e = hipMemcpyAsync(d_src, new_src.data(), nnew * 4, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_add_nodes, dim3(ggrid(ctx, nnew)), dim3(BLOCK), 0, ctx->stream, v, first_new, d_src, nnew);
            e = hipGetLastError();
        }
        v.n_nodes += nnew;                                    // (the kernels below index the new nodes)
        if (e == hipSuccess && !end_edge.empty()) {
            e = hipMemcpyAsync(d_a, end_edge.data(), end_edge.size() * 4, hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(d_b, end_node.data(), end_node.size() * 4, hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess) { hipLaunchKernelGGL(k_move_ends, dim3(ggrid(ctx, end_edge.size())), dim3(BLOCK), 0, ctx->stream, v, d_a, d_b, (u64)end_edge.size()); e = hipGetLastError(); }
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);      // d_a / d_b are reused below
        }
        if (e == hipSuccess && !start_edge.empty()) {
            e = hipMemcpyAsync(d_a, start_edge.data(), start_edge.size() * 4, hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(d_b, start_node.data(), start_node.size() * 4, hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess) { hipLaunchKernelGGL(k_move_starts, dim3(ggrid(ctx, start_edge.size())), dim3(BLOCK), 0, ctx->stream, v, d_a, d_b, (u64)start_edge.size()); e = hipGetLastError(); }
        }
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        (void)hipFree(d_src); (void)hipFree(d_a); (void)hipFree(d_b);
        if (e != hipSuccess) return hip_fail(ctx, e, "gk_graph_split_by_support");
        g->index_ready = false;                                // several nodes share a sequence now: the next point query rebuilds the index with all of them
    }
    uint64_t removed = 0;
    if (!to_remove.empty()) { if (int rc = gk_graph_remove_edges_by_id(g, to_remove.data(), to_remove.size(), &removed)) return rc; }   // :316
    if (removed_edges) *removed_edges = removed;
    if (new_nodes) *new_nodes = nnew;
    return graph_refresh_counts(g);
}

}  // extern "C"
