// genome.hpp — C++ host side above the C-ABI (include/genome_amd.h), mirroring the reference's
// Scala interface for the hot path name for name (the image has no JVM, so the host side is C++;
// the JNI stub + Scala adapter a maintainer would add are in INTEGRATION.md).
//
//   trait DNAMap[T]                       S/ds/ArrayDNAMap.scala:49-60      -> genome::DNAMap (T = Int)
//   FreqFilter.extractFilteredKmers       S/data/FreqFilter.scala:25-58     -> genome::FreqFilter::extractFilteredKmers
//   Graph.buildGraph / trait Graph        S/data/graph/Graph.scala:23-150,269-382 -> genome::Graph
//   PairedEndData                         S/data/PairedEndData.scala:11-36  -> genome::PairedEndData
//   PartitionedDNAMap                     S/ds/PartitionedDNAMap.scala:15-64 -> genome::PartitionedDNAMap (one rank per GPU, RCCL)
//
// Header-only, C++17, links against libgenome_amd.so.  Errors are exceptions (GkError) carrying the
// C-ABI status and message; a wrong key length throws KeyLengthError, the analogue of the
// reference's AssertionError (ArrayDNAMap.scala:182).  No CPU fallback: without a GPU the
// Context constructor throws.
#pragma once

#include <algorithm>
#include <cstdint>
#include <functional>
#include <map>
#include <optional>
#include <ostream>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "../../include/genome_amd.h"

namespace genome {

struct GkError : std::runtime_error {
    int code;
    GkError(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};
struct KeyLengthError : GkError {
    using GkError::GkError;
};

inline void check(int rc, const gk_ctx *ctx) {
    if (rc == GK_OK) return;
    std::string msg = gk_last_error(ctx);
    if (rc == GK_E_KLEN) throw KeyLengthError(rc, msg);
    throw GkError(rc, msg);
}

// S/dna/Base.scala:13-19 + S/dna/DNASeq.scala:74-215: a sequence of <= 64 bases, base i at bits 2i.
struct DNASeq {
    uint64_t lo = 0, hi = 0;
    int len = 0;
    static int code(char c) {
        switch (c) { case 'A': return 0; case 'G': return 1; case 'C': return 2; case 'T': return 3; }
        throw std::invalid_argument(std::string("not a base: ") + c);
    }
    static DNASeq fromString(const std::string &s) {
        if (s.size() > 64) throw std::invalid_argument("DNASeq longer than 64 bases");
        DNASeq r;
        r.len = (int)s.size();
        for (int i = 0; i < r.len; i++) (i < 32 ? r.lo : r.hi) |= (uint64_t)code(s[i]) << (2 * (i % 32));
        return r;
    }
    int apply(int i) const { return (int)(((i < 32 ? lo : hi) >> (2 * (i % 32))) & 3); }
    std::string toString() const {
        std::string s(len, 'A');
        for (int i = 0; i < len; i++) s[i] = "AGCT"[apply(i)];
        return s;
    }
    DNASeq revComplement() const {     // DNASeq.scala:27-28
        std::string s = toString(), r(s.rbegin(), s.rend());
        for (auto &c : r) c = c == 'A' ? 'T' : c == 'T' ? 'A' : c == 'G' ? 'C' : 'G';
        return fromString(r);
    }
    bool operator==(const DNASeq &o) const { return len == o.len && lo == o.lo && hi == o.hi; }
    bool operator<(const DNASeq &o) const { return std::tie(hi, lo) < std::tie(o.hi, o.lo); }
};

// replaces ActorsHome.system (S/scripts/ActorsHome.scala:20-30)
class Context {
  public:
    explicit Context(int device = 0) { check(gk_ctx_create(device, &h_), nullptr); }
    ~Context() { gk_ctx_destroy(h_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    gk_ctx *handle() const { return h_; }
    void trim() { check(gk_ctx_trim(h_), h_); }        // device buffers parked in the context's pool go back to the device

  private:
    gk_ctx *h_ = nullptr;
};

// S/data/PairedEndData.scala:11-36: `count` pairs in a `.bin` record stream, two records per pair.
struct PairedEndData {
    uint64_t count = 0;
    int insert = 0;
    std::vector<uint8_t> bin;
};

// the two closures the hot path passes to DNAMap
struct PlusOne {};                 // `_ + 1`            FreqFilter.scala:33
struct ValueLessThan { int rounds; };   // `(k, v) => v < rounds`   FreqFilter.scala:55

// trait DNAMap[Int] over one HBM-resident partition (ArrayDNAMap.scala:49-60, 62-243)
class DNAMap {
  public:
    DNAMap(Context &ctx, int k, uint64_t capacityHint = 0) : ctx_(ctx), k_(k) { check(gk_map_create(ctx.handle(), k, capacityHint, &h_), ctx.handle()); }
    ~DNAMap() { gk_map_destroy(h_); }
    DNAMap(const DNAMap &) = delete;
    DNAMap &operator=(const DNAMap &) = delete;
    DNAMap(DNAMap &&o) noexcept : ctx_(o.ctx_), k_(o.k_), h_(o.h_) { o.h_ = nullptr; }
    // adopt a handle the library created (gk_dist_gather_map)
    DNAMap(Context &ctx, int k, gk_map *adopted) : ctx_(ctx), k_(k), h_(adopted) {}

    int k() const { return k_; }
    gk_map *handle() const { return h_; }
    Context &context() const { return ctx_; }

    uint64_t size() const {                                        // :50
        uint64_t n = 0;
        check(gk_map_size(h_, &n), ctx_.handle());
        return n;
    }
    std::optional<int32_t> apply(const DNASeq &key) const {        // :51
        requireLen(key);
        int32_t v = -1;
        check(gk_map_get_batch(h_, &key.lo, &key.hi, 1, &v, nullptr), ctx_.handle());
        return v < 0 ? std::nullopt : std::optional<int32_t>(v);
    }
    bool contains(const DNASeq &key) const { return apply(key).has_value(); }   // :57
    void update(const DNASeq &key, int32_t v0, PlusOne) {          // :54 update(key, 1, _ + 1)
        requireLen(key);
        if (v0 != 1) throw GkError(GK_E_INVALID, "only update(key, 1, _ + 1) is a fused GPU form");
        check(gk_map_update_inc(h_, &key.lo, &key.hi, 1), ctx_.handle());
    }
    void updateAll(const std::vector<DNASeq> &keys, int32_t v0, PlusOne p) {   // batched form of the above
        (void)p;
        if (v0 != 1) throw GkError(GK_E_INVALID, "only update(key, 1, _ + 1) is a fused GPU form");
        std::vector<uint64_t> lo(keys.size()), hi(keys.size());
        for (size_t i = 0; i < keys.size(); i++) { requireLen(keys[i]); lo[i] = keys[i].lo; hi[i] = keys[i].hi; }
        check(gk_map_update_inc(h_, lo.data(), hi.data(), keys.size()), ctx_.handle());
    }
    void deleteAll(ValueLessThan p) { check(gk_map_filter_lt(h_, p.rounds), ctx_.handle()); }   // :56
    // :58-59 mapReduce / foreach: the table comes back to the host and `f` runs there
    void foreach(const std::function<void(const DNASeq &, int32_t)> &f) const {
        uint64_t n = size(), got = 0;
        std::vector<uint64_t> lo(n), hi(n);
        std::vector<int32_t> cnt(n);
        check(gk_map_export(h_, lo.data(), hi.data(), cnt.data(), n, &got), ctx_.handle());
        for (uint64_t i = 0; i < got; i++) f(DNASeq{lo[i], hi[i], k_}, cnt[i]);
    }
    // FreqFilter.add over a record stream (FreqFilter.scala:28-36, 44-48)
    uint64_t countReads(const uint8_t *bin, size_t nbytes, uint64_t nreads) {
        uint64_t occ = 0;
        check(gk_map_count_reads(h_, bin, nbytes, nreads, &occ), ctx_.handle());
        return occ;
    }
    // the table's invariants and an order-independent content checksum (gk_map_verify)
    struct Verify { uint64_t live, bad, sumCounts, checksum; };
    Verify verify() const {
        Verify v{};
        check(gk_map_verify(h_, &v.live, &v.bad, &v.sumCounts, &v.checksum), ctx_.handle());
        return v;
    }

  private:
    void requireLen(const DNASeq &key) const {                     // assert(key.length == k)  :182
        if (key.len != k_) throw KeyLengthError(GK_E_KLEN, "key length " + std::to_string(key.len) + " != k=" + std::to_string(k_));
    }
    Context &ctx_;
    int k_;
    gk_map *h_ = nullptr;
};

// One rank's view of a PartitionedDNAMap[Int] spread over the GPUs of a node (PartitionedDNAMap.scala:15-64): this rank's
// partition (an ordinary DNAMap on this rank's device) + the RCCL communicator.  One process per GPU; rank 0 obtains the id
// with PartitionedDNAMap::uniqueId() and hands it to the others through whatever channel the host program has.
// Every member that moves data is COLLECTIVE.  Owner of a k-mer = strand-symmetric minimizer hash mod world (gk_owner_of),
// not `hashCode mod P` (:60-63): the partition function is unobservable in results and keeps x and rc(x) together.
class PartitionedDNAMap {
  public:
    static std::vector<uint8_t> uniqueId() {
        std::vector<uint8_t> id(128);
        check(gk_dist_unique_id(id.data()), nullptr);
        return id;
    }
    PartitionedDNAMap(Context &ctx, int k, int rank, int world, const std::vector<uint8_t> &id128, uint64_t capacityHintPerRank = 0)
        : ctx_(ctx), local_(ctx, k, capacityHintPerRank) {
        if (id128.size() != 128) throw GkError(GK_E_INVALID, "the RCCL id is 128 bytes");
        check(gk_dist_create(ctx.handle(), rank, world, id128.data(), &d_), ctx.handle());
    }
    ~PartitionedDNAMap() { gk_dist_destroy(d_); }
    PartitionedDNAMap(const PartitionedDNAMap &) = delete;
    PartitionedDNAMap &operator=(const PartitionedDNAMap &) = delete;

    int rank() const { return gk_dist_rank(d_); }
    int world() const { return gk_dist_world(d_); }
    DNAMap &local() { return local_; }
    // FreqFilter.add over THIS rank's device-resident reads: update(key, 1, _ + 1) sent to each key's owner (:41-43)
    std::pair<uint64_t, uint64_t> countReadsDev(const void *devRecords, uint64_t nreads, int readLen) {
        uint64_t sent = 0, owned = 0;
        check(gk_dist_count_reads_dev(d_, local_.handle(), devRecords, nreads, readLen, &sent, &owned), ctx_.handle());
        return {sent, owned};
    }
    // the same in two halves for a streaming loop: routeBegin(batch i+1), and routeBegin(batch i+2) for the exchange to run
    // ahead as well, before countRouted() of batch i (at most three begun)
    void routeBegin(const void *devRecords, uint64_t nreads, int readLen) {
        check(gk_dist_route_begin(d_, local_.k(), devRecords, nreads, readLen), ctx_.handle());
    }
    std::pair<uint64_t, uint64_t> countRouted() {
        uint64_t sent = 0, owned = 0;
        check(gk_dist_count_routed(d_, local_.handle(), &sent, &owned), ctx_.handle());
        return {sent, owned};
    }
    uint64_t size() {                                              // :31
        uint64_t n = 0;
        check(gk_dist_size(d_, local_.handle(), &n), ctx_.handle());
        return n;
    }
    void deleteAll(ValueLessThan p) { local_.deleteAll(p); }      // :49-51 — every partition filters its own keys
    // every partition's keys in one table on this rank: what Graph.buildGraph needs.  classified: the keys' owners
    // classify them first (Graph.scala:320-329 on every partition, :55-58) and the degree masks travel with the keys
    DNAMap gathered(bool classified = false) {
        gk_map *full = nullptr;
        check(classified ? gk_dist_gather_classified_map(d_, local_.handle(), &full) : gk_dist_gather_map(d_, local_.handle(), &full), ctx_.handle());
        return DNAMap(ctx_, local_.k(), full);
    }
    void barrier() { check(gk_dist_barrier(d_), ctx_.handle()); }

  private:
    Context &ctx_;
    DNAMap local_;
    gk_dist *d_ = nullptr;
};

namespace FreqFilter {
// FreqFilter.extractFilteredKmers(data, k, rounds) (FreqFilter.scala:25-58); `takeFirst` is
// genome.takeFirst (:40, :44)
// prefilterDistinct > 0 (rounds >= 2 only): the exact two-pass singleton pre-filter sized for that many
// distinct k-mers runs first (include/genome_amd.h); same result, k-mers seen once take no table slot
inline DNAMap extractFilteredKmers(Context &ctx, const PairedEndData &data, int k, int rounds,
                                   uint64_t takeFirst = UINT64_MAX, uint64_t capacityHint = 0, uint64_t prefilterDistinct = 0) {
    DNAMap kmersFreq(ctx, k, capacityHint);
    const uint64_t pairs = std::min<uint64_t>(data.count, takeFirst);
    if (prefilterDistinct) {
        if (rounds < 2) throw GkError(GK_E_INVALID, "the singleton pre-filter needs rounds >= 2");
        gk_prefilter *pf = nullptr;
        check(gk_prefilter_create(ctx.handle(), k, prefilterDistinct, &pf), ctx.handle());
        int rc = gk_prefilter_add_reads(pf, data.bin.data(), data.bin.size(), 2 * pairs);
        if (rc == GK_OK) rc = gk_map_count_reads_prefiltered(kmersFreq.handle(), pf, data.bin.data(), data.bin.size(), 2 * pairs, nullptr, nullptr);
        gk_prefilter_destroy(pf);
        check(rc, ctx.handle());
    } else {
        kmersFreq.countReads(data.bin.data(), data.bin.size(), 2 * pairs);
    }
    kmersFreq.deleteAll(ValueLessThan{rounds});
    return kmersFreq;
}
}  // namespace FreqFilter

// DNAMap[GraphPosition] — the multimap Graph.getGraphMap fills (Graph.scala:90-119); values are GK_POS_* encoded
class PositionMap {
  public:
    PositionMap(Context &ctx, int k, uint64_t capacityHint = 0) : ctx_(ctx) { check(gk_vmap_create(ctx.handle(), k, capacityHint, &h_), ctx.handle()); }
    ~PositionMap() { gk_vmap_destroy(h_); }
    PositionMap(PositionMap &&o) noexcept : ctx_(o.ctx_), h_(o.h_) { o.h_ = nullptr; }
    PositionMap(const PositionMap &) = delete;
    gk_vmap *handle() const { return h_; }
    uint64_t size() const { uint64_t n = 0; check(gk_vmap_size(h_, &n), ctx_.handle()); return n; }

  private:
    Context &ctx_;
    gk_vmap *h_ = nullptr;
};

// pathsMap + badPairs of GraphSimplifier.scala:209-211
class Support {
  public:
    explicit Support(Context &ctx) : ctx_(ctx) { check(gk_support_create(ctx.handle(), &h_), ctx.handle()); }
    ~Support() { gk_support_destroy(h_); }
    Support(const Support &) = delete;
    gk_support *handle() const { return h_; }
    // -> (supported edge pairs, bad pairs, pair orientations walked)
    std::tuple<uint64_t, uint64_t, uint64_t> sizes() const {
        uint64_t a = 0, b = 0, c = 0;
        check(gk_support_size(h_, &a, &b, &c), ctx_.handle());
        return {a, b, c};
    }

  private:
    Context &ctx_;
    gk_support *h_ = nullptr;
};

struct Edge {                 // S/data/graph/Edge.scala:11 (ids are not part of the observable result)
    DNASeq start, end;
    std::string seq;
};

// trait Graph / MapGraph (Graph.scala:23-262), HBM resident
class Graph {
  public:
    // Graph.buildGraph(k, kmersFreq) :269-382
    static Graph buildGraph(int k, DNAMap &kmersFreq) {
        if (kmersFreq.k() != k) throw KeyLengthError(GK_E_KLEN, "map k != k");
        Graph g(kmersFreq.context(), k);
        check(gk_graph_build(kmersFreq.handle(), &g.h_), kmersFreq.context().handle());
        return g;
    }
    ~Graph() { gk_graph_destroy(h_); }
    Graph(Graph &&o) noexcept : ctx_(o.ctx_), k_(o.k_), h_(o.h_) { o.h_ = nullptr; }
    Graph(const Graph &) = delete;

    void simplifyGraph() { check(gk_graph_simplify(h_), ctx_.handle()); }           // :211-230
    void removeBubbles() { check(gk_graph_remove_bubbles(h_), ctx_.handle()); }     // :125-149
    bool removeEdge(const DNASeq &start, char firstBase) {                          // :191-195
        uint8_t b = (uint8_t)DNASeq::code(firstBase);
        uint64_t removed = 0;
        check(gk_graph_remove_edges(h_, &start.lo, &start.hi, &b, 1, &removed), ctx_.handle());
        return removed == 1;
    }
    // components.maxBy(_.size) + retain (:54-72, :161-165; GraphBuilder.scala:52-54)
    std::pair<uint64_t, uint64_t> retainLargestComponent() {
        uint64_t kept = 0, comps = 0;
        check(gk_graph_retain_largest(h_, &kept, &comps), ctx_.handle());
        return {kept, comps};
    }
    // GraphBuilder.scala:41-47: `hist` = components grouped by node count, `hist2` = grouped by the summed length of their
    // nodes' out-edges; each as value -> number of components, ascending
    std::pair<std::map<uint64_t, uint64_t>, std::map<uint64_t, uint64_t>> componentHistograms() const {
        uint64_t n = 0;
        int rc = gk_graph_component_stats(h_, nullptr, nullptr, 0, &n);
        if (rc != GK_OK && rc != GK_E_CAPACITY) check(rc, ctx_.handle());
        std::vector<uint32_t> nodes(n);
        std::vector<uint64_t> len(n);
        if (n) check(gk_graph_component_stats(h_, nodes.data(), len.data(), n, &n), ctx_.handle());
        std::map<uint64_t, uint64_t> h1, h2;
        for (uint64_t i = 0; i < n; i++) { h1[nodes[i]]++; h2[len[i]]++; }
        return {h1, h2};
    }
    // Graph.getGraphMap :90-119
    PositionMap getGraphMap() {
        auto [n, e, l] = counts();
        PositionMap pm(ctx_, k_, l + n);
        uint64_t entries = 0;
        check(gk_graph_position_map(h_, pm.handle(), &entries), ctx_.handle());
        return pm;
    }
    // GraphSimplifier.scala:213-247: the pairs' positions, annotate, the bounded walks -> support counts
    void walkPairs(PositionMap &positions, Support &support, const PairedEndData &data, uint64_t takeFirst, int rangeLo = 180, int rangeHi = 250) {
        check(gk_graph_walk_pairs(h_, positions.handle(), support.handle(), data.bin.data(), data.bin.size(), std::min<uint64_t>(data.count, takeFirst),
                                  rangeLo, rangeHi), ctx_.handle());
    }
    // :272-316 -> (edges removed, nodes added); simplifyGraph() is the next call (:318)
    std::pair<uint64_t, uint64_t> splitBySupport(const Support &support, int cutoff) {
        uint64_t rm = 0, nn = 0;
        check(gk_graph_split_by_support(h_, support.handle(), cutoff, &rm, &nn), ctx_.handle());
        return {rm, nn};
    }
    std::tuple<uint64_t, uint64_t, uint64_t> counts() const {
        uint64_t n = 0, e = 0, l = 0;
        check(gk_graph_counts(h_, &n, &e, &l), ctx_.handle());
        return {n, e, l};
    }
    std::vector<DNASeq> getNodes() const {                                          // sorted by k-mer
        auto [n, e, l] = counts();
        (void)e; (void)l;
        std::vector<uint64_t> lo(n), hi(n);
        uint64_t got = 0;
        check(gk_graph_export_nodes(h_, lo.data(), hi.data(), n, &got), ctx_.handle());
        std::vector<DNASeq> out(got);
        for (uint64_t i = 0; i < got; i++) out[i] = DNASeq{lo[i], hi[i], k_};
        std::sort(out.begin(), out.end());
        return out;
    }
    std::vector<Edge> getEdges() const {                                            // sorted by (start, first base)
        auto [n, ne, ln] = counts();
        (void)n;
        std::vector<uint64_t> slo(ne), shi(ne), elo(ne), ehi(ne);
        std::vector<int64_t> len(ne), off(ne);
        std::vector<uint8_t> seq((ln + 3 * ne) / 4 + 1);
        uint64_t got = 0, used = 0;
        check(gk_graph_export_edges(h_, slo.data(), shi.data(), elo.data(), ehi.data(), len.data(), off.data(), ne, &got,
                                    seq.data(), seq.size(), &used), ctx_.handle());
        std::vector<Edge> out(got);
        for (uint64_t i = 0; i < got; i++) {
            out[i].start = DNASeq{slo[i], shi[i], k_};
            out[i].end = DNASeq{elo[i], ehi[i], k_};
            out[i].seq.resize((size_t)len[i]);
            for (int64_t j = 0; j < len[i]; j++) out[i].seq[(size_t)j] = "AGCT"[(seq[(size_t)(off[i] + j / 4)] >> ((j % 4) * 2)) & 3];
        }
        std::sort(out.begin(), out.end(), [](const Edge &a, const Edge &b) {
            if (!(a.start == b.start)) return a.start < b.start;
            return DNASeq::code(a.seq[0]) < DNASeq::code(b.seq[0]);
        });
        return out;
    }

    // the `contigs` file of GraphSimplifier.scala:338-347: per edge its sequence, then ">abacaba<i>"
    // (edge order: canonical, since the reference's is ConcurrentHashMap order)
    void writeContigs(std::ostream &out) const {
        size_t i = 0;
        for (const auto &e : getEdges()) out << e.seq << "\n>abacaba" << i++ << "\n";
    }
    // Graph.writeDot (Graph.scala:74-88): `start -> end [label=seq|length]`, ids = rank of the node's k-mer
    void writeDot(std::ostream &out) const {
        const auto nodes = getNodes();
        auto id = [&](const DNASeq &s) { return (size_t)(std::lower_bound(nodes.begin(), nodes.end(), s) - nodes.begin()) + 1; };
        out << "digraph G {\n";
        for (const auto &e : getEdges())
            out << id(e.start) << " -> " << id(e.end) << " [label=" << (e.seq.size() <= 50 ? e.seq : std::to_string(e.seq.size())) << "]\n";
        out << "}\n";
    }

  private:
    Graph(Context &ctx, int k) : ctx_(ctx), k_(k) {}
    Context &ctx_;
    int k_;
    gk_graph *h_ = nullptr;
};

}  // namespace genome
