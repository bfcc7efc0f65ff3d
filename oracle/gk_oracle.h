/*
 * gk_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C, single-threaded, literal restatement of the reference's k-mer hashing and
 * de Bruijn graph-build path (winger/genome, Scala 2.9.1).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may link or call this; the product path (genome_amd/csrc) never
 * does.  Path shorthand: S/ = /root/reference/src/main/scala/ru/ifmo/genome/.
 *
 * PARITY STATUS: "parity unpinned".  The reference ships no tests, fixtures or golden vectors for
 * this path and cannot be compiled or run here (no JVM; dead SNAPSHOT
 * dependencies).  This restatement is pinned only by (i) an independent base-by-base Python
 * restatement (oracle/pyref.py), (ii) the known answers transliterated in SURVEY.md §8a-9/§8c,
 * (iii) the reference's in-source invariants and the one real data point at application.conf:73.
 * Third-party arithmetic restated here: scala-library 2.9.1 `Long.##` (ScalaRunTime.hash(Long)).
 */
#ifndef GK_ORACLE_H
#define GK_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* A k-mer (k<=64) in the reference bit layout: base i at bits 2i of lo (i<32) or 2(i-32) of hi;
 * unused high bits zero.  S/dna/DNASeq.scala:74-215 (Long1DNASeq / Long2DNASeq). */
typedef struct { uint64_t lo, hi; } gko_kmer;

/* 1 for the k the reference handles correctly: 2..31 and 34..64 (k=32 complement
 * bug, k=33 `:+` prepends). */
int gko_k_supported(int k);

/* S/dna/Base.scala:13-19 — A0 G1 C2 T3, complement A<->T G<->C. */
int gko_base_complement(int b);
int gko_base_from_char(char c);   /* -1 if not AGCT */
char gko_base_to_char(int b);

int gko_kmer_get(gko_kmer x, int i);
gko_kmer gko_kmer_from_bases(const uint8_t *bases, int len);          /* builder, DNASeq.scala:237-281 */
gko_kmer gko_kmer_from_packed(const uint8_t *packed, int pos, int k); /* window of an unpacked-on-the-fly read */
gko_kmer gko_revcomp(gko_kmer x, int k);   /* DNASeq.scala:28 complement.reverse (+ :155-168 for k<=32) */
int32_t gko_hash(gko_kmer x, int k);       /* DNASeq.scala:103 (k<=32) / BloomFilter.scala:12-15 + DNASeq.scala:204-208 */
gko_kmer gko_canon(gko_kmer x, int k);     /* FreqFilter.scala:31-32 */
int32_t gko_improve(int32_t h);            /* ArrayDNAMap.scala:267-272 */
int gko_partition(gko_kmer x, int k, int P); /* PartitionedDNAMap.scala:60-63 */
gko_kmer gko_prepend(int b, gko_kmer x, int k); /* b +: x.take(k-1)   Graph.scala:273 */
gko_kmer gko_append(gko_kmer x, int b, int k);  /* x.drop(1) :+ b     Graph.scala:279 */
int gko_kmer_cmp(gko_kmer a, gko_kmer b);       /* unsigned (hi,lo) order used by canonical serialisation */

/* ---- ArrayDNAMap[Int] (one partition): S/ds/ArrayDNAMap.scala:62-243 ---- */
typedef struct gko_map gko_map;
gko_map *gko_map_new(int k);
void gko_map_free(gko_map *m);
int gko_map_size(const gko_map *m);
int gko_map_bins(const gko_map *m);
int gko_map_rescales(const gko_map *m);
void gko_map_update_inc(gko_map *m, gko_kmer key);           /* update(key, 1, _+1)  :129-150,198-203 */
void gko_map_update_set(gko_map *m, gko_kmer key, int32_t v); /* update(key, v)       :115-127,191-196 */
void gko_map_put_new(gko_map *m, gko_kmer key, int32_t v);    /* putNew               :152-162,205-210 */
int gko_map_get(const gko_map *m, gko_kmer key, int32_t *v);  /* apply                :90-101 */
int gko_map_get_all(const gko_map *m, gko_kmer key, int32_t *out, int cap); /* getAll :103-113 (most recent first) */
void gko_map_delete_lt(gko_map *m, int32_t rounds);           /* deleteAll((k,v)=>v<rounds) :164-173,212-215 */
/* iterator in slot order (:175-178); returns number written (<= cap) */
size_t gko_map_export(const gko_map *m, uint64_t *lo, uint64_t *hi, int32_t *val, size_t cap);

/* ---- PartitionedDNAMap[Int]: P ArrayDNAMaps, owner = hashCode mod P ---- */
typedef struct gko_pmap gko_pmap;
gko_pmap *gko_pmap_new(int k, int P);
void gko_pmap_free(gko_pmap *pm);
int gko_pmap_k(const gko_pmap *pm);
int gko_pmap_parts(const gko_pmap *pm);
gko_map *gko_pmap_part(gko_pmap *pm, int p);
long gko_pmap_size(const gko_pmap *pm);
int gko_pmap_contains(const gko_pmap *pm, gko_kmer key);       /* one strand only, as DNAMap.contains */
int gko_pmap_get(const gko_pmap *pm, gko_kmer key, int32_t *v);
void gko_pmap_update_inc(gko_pmap *pm, gko_kmer key);
void gko_pmap_delete_lt(gko_pmap *pm, int32_t rounds);
/* all live (key,count) sorted by (hi,lo) unsigned: the canonical table serialisation */
size_t gko_pmap_export_sorted(const gko_pmap *pm, uint64_t *lo, uint64_t *hi, int32_t *val, size_t cap);

/* FreqFilter.extractFilteredKmers counting loop (FreqFilter.scala:28-51) over the reference `.bin`
 * record stream [len:u8][ceil(len/4) bytes] (PairedEndData.scala:20-36).  nreads records are
 * consumed (2 per pair).  Returns the number of k-mer occurrences processed, or -1 if the stream
 * is truncated. */
long gko_count_reads(gko_pmap *pm, const uint8_t *bin, size_t nbytes, uint64_t nreads);
/* the same on host threads: `nthreads` readers route k-mers to the P partitions of pm, then one thread
 * per partition inserts (PartitionedDNAMap without the network; the CPU baseline's multi-core form) */
long gko_count_reads_mt(gko_pmap *pm, const uint8_t *bin, size_t nbytes, uint64_t nreads, int nthreads);

/* ---- Graph (S/data/graph/Graph.scala) ---- */
typedef struct gko_graph gko_graph;
/* Graph.buildGraph (Graph.scala:269-382).  Node ids = 1.. in ascending k-mer order, edge ids =
 * 1.. in (node id, base A,G,C,T) order: a deterministic stand-in for the reference's
 * HashSet/.par-dependent ids (SURVEY §8c "caller-level nondeterminism"). */
gko_graph *gko_graph_build(const gko_pmap *pm);
void gko_graph_free(gko_graph *g);
int gko_graph_k(const gko_graph *g);
long gko_graph_num_nodes(const gko_graph *g);     /* live */
long gko_graph_num_edges(const gko_graph *g);     /* live */
long gko_graph_total_edge_len(const gko_graph *g);
void gko_graph_simplify(gko_graph *g);            /* MapGraph.simplifyGraph  :211-230, nodes in ascending id */
void gko_graph_remove_bubbles(gko_graph *g);      /* Graph.removeBubbles     :125-149, nodes in ascending id */
/* MapGraph.removeEdge (:191-195) for the edge leaving node `start` whose seq starts with `base`;
 * returns 0 if no such edge. */
int gko_graph_remove_edge(gko_graph *g, gko_kmer start, int base);
/* Graph.components + GraphBuilder retain(maxBy size) (:54-72, :161-165; GraphBuilder.scala:52-54);
 * tie between equal-size components -> the one holding the smallest k-mer.  Returns its size. */
long gko_graph_retain_largest(gko_graph *g);
long gko_graph_num_components(const gko_graph *g);

/* Canonical serialisation: nodes sorted by k-mer; edges sorted by (start k-mer, first base). */
size_t gko_graph_export_nodes(const gko_graph *g, uint64_t *lo, uint64_t *hi, size_t cap);
/* per edge: start/end k-mer, length, offset (in bases) into the base pool. `bases_out` gets one
 * base code per byte.  Returns edge count; *nbases_out = pool size needed/used. */
size_t gko_graph_export_edges(const gko_graph *g, uint64_t *slo, uint64_t *shi, uint64_t *elo,
                              uint64_t *ehi, int64_t *len, int64_t *off, size_t cap,
                              uint8_t *bases_out, size_t bases_cap, size_t *nbases_out);
/* out-edge insertion order of one node (Map1..Map4 order), as base codes; returns count or -1 */
int gko_graph_out_order(const gko_graph *g, gko_kmer node, int *bases4);
/* per-node degree for invariant checks; returns -1 if node unknown */
int gko_graph_degree(const gko_graph *g, gko_kmer node, int *in_deg, int *out_deg);

/* ---- ids, point edits, Graph.getGraphMap (what GraphSimplifier.scala:188-316 uses) ---- */
int64_t gko_graph_find_node(const gko_graph *g, gko_kmer x);            /* first live node with this sequence, 0 if none */
int64_t gko_graph_find_out_edge(const gko_graph *g, int64_t node, int base);
int64_t gko_graph_add_node(gko_graph *g, gko_kmer seq);                 /* MapGraph.addNode      Graph.scala:172-176 */
void gko_graph_replace_start(gko_graph *g, int64_t edge, int64_t new_start);  /* :197-202 */
void gko_graph_replace_end(gko_graph *g, int64_t edge, int64_t new_end);      /* :204-209 */
/* Graph.getGraphMap (:90-119) as the sequence of its putNew calls (key, NodeGraphPosition(id) | EdgeGraphPosition(id, dist)) */
size_t gko_graph_get_graph_map(const gko_graph *g, uint64_t *lo, uint64_t *hi, uint8_t *is_edge, int64_t *id, int32_t *dist, size_t cap);
gko_kmer gko_graph_node_seq(const gko_graph *g, int64_t id);
int gko_graph_edge_info(const gko_graph *g, int64_t id, int64_t *start, int64_t *end, int64_t *len, int *first);


/* ---- paired-end walking (S/scripts/GraphSimplifier.scala): literal restatement, test infrastructure */
typedef struct gko_support gko_support;             /* pathsMap :209 — (edge id, edge id) -> count — and badPairs :211 */
gko_support *gko_support_new(void);
void gko_support_free(gko_support *s);
long gko_support_bad_pairs(const gko_support *s);
size_t gko_support_export(const gko_support *s, int64_t *e1, int64_t *e2, int32_t *cnt, size_t cap);
/* :188-247 over the first npairs pairs of a `.bin` stream; range = range_lo to range_hi inclusive (:146); returns the
 * number of pair orientations walked */
long gko_graph_walk_pairs(const gko_graph *g, gko_support *sup, const uint8_t *bin, size_t nbytes, uint64_t npairs, int range_lo, int range_hi);
/* :272-318: support matrix per node, groups at `cutoff`, node split, removeEdge(toRemove), simplifyGraph */
void gko_graph_split_by_support(gko_graph *g, const gko_support *sup, int cutoff, long *removed_edges, long *new_nodes);

#ifdef __cplusplus
}
#endif
#endif
