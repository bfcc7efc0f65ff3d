/*
 * gk_oracle.c — CPU ORACLE (test infrastructure, NOT product code).  See gk_oracle.h.
 *
 * Literal single-threaded restatement of the reference algorithm; every function cites the
 * reference lines it follows (S/ = /root/reference/src/main/scala/ru/ifmo/genome/).
 * Java/Scala integer semantics are reproduced explicitly: 64-bit shifts use the count mod 64,
 * `>>` is arithmetic, `>>>` logical, Int arithmetic wraps at 32 bits.
 * PARITY STATUS: parity unpinned (no reference tests/fixtures exist; see header).
 */
#include "gk_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ---------- Java-semantics helpers ---------- */
static inline int64_t jshl(int64_t v, int n) { return (int64_t)((uint64_t)v << (n & 63)); }
static inline int64_t jshr(int64_t v, int n) { return v >> (n & 63); }               /* >>  */
static inline int64_t jushr(int64_t v, int n) { return (int64_t)((uint64_t)v >> (n & 63)); } /* >>> */

int gko_k_supported(int k) { return (k >= 2 && k <= 31) || (k >= 34 && k <= 64); }

/* S/dna/Base.scala:13-19 */
int gko_base_complement(int b) {
    static const int comp[4] = {3 /*A->T*/, 2 /*G->C*/, 1 /*C->G*/, 0 /*T->A*/};
    return comp[b & 3];
}
int gko_base_from_char(char c) {
    switch (c) { case 'A': return 0; case 'G': return 1; case 'C': return 2; case 'T': return 3; }
    return -1;
}
char gko_base_to_char(int b) { return "AGCT"[b & 3]; }

/* Long1DNASeq.apply :80-85 / Long2DNASeq.apply :178-184 */
int gko_kmer_get(gko_kmer x, int i) {
    uint64_t w = i < 32 ? x.lo : x.hi;
    return (int)((w >> ((i % 32) * 2)) & 3);
}

/* DNASeq.newBuilder :237-281 (count <= 64 branch) */
gko_kmer gko_kmer_from_bases(const uint8_t *bases, int len) {
    gko_kmer r = {0, 0};
    for (int c = 0; c < len; c++) {
        if (c < 32) r.lo |= (uint64_t)(bases[c] & 3) << (c * 2);
        else r.hi |= (uint64_t)(bases[c] & 3) << ((c % 32) * 2);
    }
    return r;
}

/* ArrayDNASeq.apply :46-51: base i = (data(i/4) >> (i%4*2)) & 3; then sliding(k) window via builder */
gko_kmer gko_kmer_from_packed(const uint8_t *packed, int pos, int k) {
    gko_kmer r = {0, 0};
    for (int c = 0; c < k; c++) {
        int i = pos + c;
        uint64_t b = (uint64_t)((packed[i / 4] >> ((i % 4) * 2)) & 3);
        if (c < 32) r.lo |= b << (c * 2);
        else r.hi |= b << ((c % 32) * 2);
    }
    return r;
}

/* Long1DNASeq.complement :165-168 (note `length == 64` test, kept literally) */
static int64_t long1_complement(int64_t v, int len) {
    int64_t mask = (len == 64) ? -1LL : (int64_t)((uint64_t)jshl(1, 2 * len) - 1ULL);
    return v ^ mask;
}
/* Long1DNASeq.reverse :155-163 */
static int64_t long1_reverse(int64_t v, int len) {
    int64_t i = v;
    i = jshl(i & 0x3333333333333333LL, 2) | (jushr(i, 2) & 0x3333333333333333LL);
    i = jshl(i & 0x0f0f0f0f0f0f0f0fLL, 4) | (jushr(i, 4) & 0x0f0f0f0f0f0f0f0fLL);
    i = jshl(i & 0x00ff00ff00ff00ffLL, 8) | (jushr(i, 8) & 0x00ff00ff00ff00ffLL);
    i = jshl(i, 48) | jshl(i & 0xffff0000LL, 16) | (jushr(i, 16) & 0xffff0000LL) | jushr(i, 48);
    return jushr(i, 2 * (32 - len));
}

/* DNASeq.revComplement :28 = complement.reverse.  k<=32: Long1 bit tricks; k>32: the generic
 * IndexedSeq map/reverse through the builder (Long2DNASeq has no specialisation, :172-215). */
gko_kmer gko_revcomp(gko_kmer x, int k) {
    gko_kmer r = {0, 0};
    if (k <= 32) {
        r.lo = (uint64_t)long1_reverse(long1_complement((int64_t)x.lo, k), k);
        return r;
    }
    uint8_t b[64];
    for (int i = 0; i < k; i++) b[i] = (uint8_t)gko_base_complement(gko_kmer_get(x, i));
    uint8_t rv[64];
    for (int i = 0; i < k; i++) rv[i] = b[k - 1 - i];
    return gko_kmer_from_bases(rv, k);
}

/* scala-library 2.9.1 `Long.##` = BoxesRunTime.hashFromLong: iv = n.intValue; if (iv == n) iv else
 * java.lang.Long.hashCode(n) = (int)(n ^ (n >>> 32)).  THIRD-PARTY, restated from the published
 * 2.9.x definition; unpinned by any reference test. */
static int32_t scala291_long_hash(int64_t lv) {
    int32_t iv = (int32_t)lv;
    if ((int64_t)iv == lv) return iv;
    return (int32_t)(lv ^ jushr(lv, 32));
}

/* Long2DNASeq.multiHashCode :204-208, seed sign-extended to Long */
static int32_t long2_multihash(int64_t l1, int64_t l2, int32_t seed) {
    int64_t s = (int64_t)seed;
    int64_t t = (int64_t)((uint64_t)(l1 ^ jshr(l1, 32)) * (uint64_t)s);
    int64_t t1 = (int64_t)((uint64_t)(l2 ^ jshr(l2, 32) ^ t) * (uint64_t)s);
    return (int32_t)(t1 ^ jshr(t1, 32));
}

/* hashCode: Long1DNASeq :103 (`long.##`); Long2DNASeq inherits MultiHash.hashCode =
 * multiHashCode(42) (S/ds/BloomFilter.scala:12-15). */
int32_t gko_hash(gko_kmer x, int k) {
    if (k <= 32) return scala291_long_hash((int64_t)x.lo);
    return long2_multihash((int64_t)x.lo, (int64_t)x.hi, 42);
}

/* FreqFilter.scala:31-32: y = if (x.hashCode < rcx.hashCode) x else rcx  (signed; tie -> rcx) */
gko_kmer gko_canon(gko_kmer x, int k) {
    gko_kmer rcx = gko_revcomp(x, k);
    return gko_hash(x, k) < gko_hash(rcx, k) ? x : rcx;
}

/* ArrayDNAMap.improve :267-272 (32-bit wrapping, >>> logical) */
int32_t gko_improve(int32_t hcode) {
    uint32_t hc = (uint32_t)hcode;
    uint32_t h = hc + ~(hc << 9);
    h = h ^ (h >> 14);
    h = h + (h << 4);
    return (int32_t)(h ^ (h >> 10));
}

/* PartitionedDNAMap.partition :60-63: fix(hashCode % P), Java remainder (sign of dividend) */
int gko_partition(gko_kmer x, int k, int P) {
    int32_t i = gko_hash(x, k) % P;
    return i < 0 ? i + P : i;
}

/* base +: x.take(k-1)  (Graph.scala:273; Long1 `+:` :135-143, generic otherwise) */
gko_kmer gko_prepend(int b, gko_kmer x, int k) {
    uint8_t s[64];
    s[0] = (uint8_t)b;
    for (int i = 0; i < k - 1; i++) s[i + 1] = (uint8_t)gko_kmer_get(x, i);
    return gko_kmer_from_bases(s, k);
}
/* x.drop(1) :+ base  (Graph.scala:279; Long1 `:+` :145-153 for k<=32 since len k-1 < 32) */
gko_kmer gko_append(gko_kmer x, int b, int k) {
    uint8_t s[64];
    for (int i = 1; i < k; i++) s[i - 1] = (uint8_t)gko_kmer_get(x, i);
    s[k - 1] = (uint8_t)b;
    return gko_kmer_from_bases(s, k);
}

int gko_kmer_cmp(gko_kmer a, gko_kmer b) {
    if (a.hi != b.hi) return a.hi < b.hi ? -1 : 1;
    if (a.lo != b.lo) return a.lo < b.lo ? -1 : 1;
    return 0;
}
static int kmer_eq(gko_kmer a, gko_kmer b) { return a.lo == b.lo && a.hi == b.hi; }

/* ================= ArrayDNAMap.Container (ArrayDNAMap.scala:74-179) ================= */
typedef struct {
    int bins, mask, size;
    gko_kmer *keys;
    int32_t *ar;
    uint8_t *set, *del; /* one byte per bin: the two BitSets */
} container;

struct gko_map {
    int k;
    int rescales;
    container *c;
};

static container *container_new(int bins) {
    container *c = (container *)calloc(1, sizeof(container));
    c->bins = bins;
    c->mask = bins - 1;
    c->keys = (gko_kmer *)calloc((size_t)bins, sizeof(gko_kmer));
    c->ar = (int32_t *)calloc((size_t)bins, sizeof(int32_t));
    c->set = (uint8_t *)calloc((size_t)bins, 1);
    c->del = (uint8_t *)calloc((size_t)bins, 1);
    return c;
}
static void container_free(container *c) {
    if (!c) return;
    free(c->keys); free(c->ar); free(c->set); free(c->del); free(c);
}

/* Container.putNew :152-162 */
static void container_put_new(container *c, int k, gko_kmer key, int32_t v) {
    int i = gko_improve(gko_hash(key, k)) & c->mask;
    while (!c->del[i] && c->set[i]) i = (i + 1) & c->mask;
    c->set[i] = 1;
    c->del[i] = 0;
    c->size += 1;
    c->keys[i] = key;
    c->ar[i] = v;
}

/* ArrayDNAMap.rescale :217-230; load factors :246-247; Int*Double compares */
static void map_rescale(gko_map *m) {
    container *c = m->c;
    double bins = (double)c->bins;
    if ((c->bins > 16 && (double)c->size < bins * 0.3) || bins * 0.7 < (double)c->size) {
        int newBins = 16;
        while ((double)newBins * 0.7 < (double)c->size) newBins *= 2;
        container *n = container_new(newBins);
        for (int i = 0; i < c->bins; i++)   /* container.iterator: slot order, live only */
            if (c->set[i] && !c->del[i]) container_put_new(n, m->k, c->keys[i], c->ar[i]);
        container_free(c);
        m->c = n;
        m->rescales++;
    }
}

gko_map *gko_map_new(int k) {
    gko_map *m = (gko_map *)calloc(1, sizeof(gko_map));
    m->k = k;
    m->c = container_new(16); /* :72 */
    return m;
}
void gko_map_free(gko_map *m) {
    if (!m) return;
    container_free(m->c);
    free(m);
}
int gko_map_size(const gko_map *m) { return m->c->size; }
int gko_map_bins(const gko_map *m) { return m->c->bins; }
int gko_map_rescales(const gko_map *m) { return m->rescales; }

/* Container.update(key, v0, f) :129-150 with v0 = 1, f = _+1 (FreqFilter.scala:33); then rescale :201 */
void gko_map_update_inc(gko_map *m, gko_kmer key) {
    container *c = m->c;
    int i = gko_improve(gko_hash(key, m->k)) & c->mask;
    int firstPos = -1;
    while (c->set[i] && (c->del[i] || !kmer_eq(c->keys[i], key))) {
        if (c->del[i]) firstPos = i;
        i = (i + 1) & c->mask;
    }
    if (!c->set[i]) {
        if (firstPos != -1) {
            i = firstPos;
            c->del[i] = 0;
        }
        c->set[i] = 1;
        c->keys[i] = key;
        c->size += 1;
        c->ar[i] = 1;
    } else {
        c->ar[i] = (int32_t)((uint32_t)c->ar[i] + 1u); /* Int wraps */
    }
    map_rescale(m);
}

/* Container.update(key, v) :115-127; then rescale :194 */
void gko_map_update_set(gko_map *m, gko_kmer key, int32_t v) {
    container *c = m->c;
    int i = gko_improve(gko_hash(key, m->k)) & c->mask;
    while (!c->del[i] && c->set[i] && !kmer_eq(c->keys[i], key)) i = (i + 1) & c->mask;
    if (c->del[i] || !c->set[i]) {
        c->set[i] = 1;
        c->del[i] = 0;
        c->keys[i] = key;
        c->size += 1;
    }
    c->ar[i] = v;
    map_rescale(m);
}

void gko_map_put_new(gko_map *m, gko_kmer key, int32_t v) {
    container_put_new(m->c, m->k, key, v);
    map_rescale(m);
}

/* Container.apply :90-101 */
int gko_map_get(const gko_map *m, gko_kmer key, int32_t *v) {
    const container *c = m->c;
    int i = gko_improve(gko_hash(key, m->k)) & c->mask;
    while (c->set[i]) {
        if (!c->del[i] && kmer_eq(c->keys[i], key)) {
            if (v) *v = c->ar[i];
            return 1;
        }
        i = (i + 1) & c->mask;
    }
    return 0;
}

/* Container.getAll :103-113 (`ans ::= v` prepends: last probed first) */
int gko_map_get_all(const gko_map *m, gko_kmer key, int32_t *out, int cap) {
    const container *c = m->c;
    int i = gko_improve(gko_hash(key, m->k)) & c->mask;
    int n = 0;
    while (c->set[i]) {
        if (!c->del[i] && kmer_eq(c->keys[i], key)) {
            if (n < cap) {
                memmove(out + 1, out, (size_t)n * sizeof(int32_t));
                out[0] = c->ar[i];
            }
            n++;
        }
        i = (i + 1) & c->mask;
    }
    return n;
}

/* Container.deleteAll :164-173 with p = (k,v) => v < rounds (FreqFilter.scala:55); rescale :214 */
void gko_map_delete_lt(gko_map *m, int32_t rounds) {
    container *c = m->c;
    for (int i = 0; i < c->bins; i++) {
        if (c->set[i] && !c->del[i] && c->ar[i] < rounds) {
            c->del[i] = 1;
            c->size -= 1;
        }
    }
    map_rescale(m);
}

/* Container.iterator :175-178 */
size_t gko_map_export(const gko_map *m, uint64_t *lo, uint64_t *hi, int32_t *val, size_t cap) {
    const container *c = m->c;
    size_t n = 0;
    for (int i = 0; i < c->bins; i++) {
        if (c->set[i] && !c->del[i]) {
            if (n < cap) {
                if (lo) lo[n] = c->keys[i].lo;
                if (hi) hi[n] = c->keys[i].hi;
                if (val) val[n] = c->ar[i];
            }
            n++;
        }
    }
    return n;
}

/* ================= PartitionedDNAMap (PartitionedDNAMap.scala:15-64) ================= */
struct gko_pmap {
    int k, P;
    gko_map **parts;
};

gko_pmap *gko_pmap_new(int k, int P) {
    gko_pmap *pm = (gko_pmap *)calloc(1, sizeof(gko_pmap));
    pm->k = k;
    pm->P = P;
    pm->parts = (gko_map **)calloc((size_t)P, sizeof(gko_map *));
    for (int p = 0; p < P; p++) pm->parts[p] = gko_map_new(k);
    return pm;
}
void gko_pmap_free(gko_pmap *pm) {
    if (!pm) return;
    for (int p = 0; p < pm->P; p++) gko_map_free(pm->parts[p]);
    free(pm->parts);
    free(pm);
}
int gko_pmap_k(const gko_pmap *pm) { return pm->k; }
int gko_pmap_parts(const gko_pmap *pm) { return pm->P; }
gko_map *gko_pmap_part(gko_pmap *pm, int p) { return pm->parts[p]; }
long gko_pmap_size(const gko_pmap *pm) { /* :31 */
    long s = 0;
    for (int p = 0; p < pm->P; p++) s += gko_map_size(pm->parts[p]);
    return s;
}
int gko_pmap_get(const gko_pmap *pm, gko_kmer key, int32_t *v) { /* :33 */
    return gko_map_get(pm->parts[gko_partition(key, pm->k, pm->P)], key, v);
}
int gko_pmap_contains(const gko_pmap *pm, gko_kmer key) { return gko_pmap_get(pm, key, NULL); } /* :53 */
void gko_pmap_update_inc(gko_pmap *pm, gko_kmer key) { /* :41-43 */
    gko_map_update_inc(pm->parts[gko_partition(key, pm->k, pm->P)], key);
}
void gko_pmap_delete_lt(gko_pmap *pm, int32_t rounds) { /* :49-51 */
    for (int p = 0; p < pm->P; p++) gko_map_delete_lt(pm->parts[p], rounds);
}

typedef struct { gko_kmer key; int32_t v; } kv;
static int kv_cmp(const void *a, const void *b) { return gko_kmer_cmp(((const kv *)a)->key, ((const kv *)b)->key); }

size_t gko_pmap_export_sorted(const gko_pmap *pm, uint64_t *lo, uint64_t *hi, int32_t *val, size_t cap) {
    size_t n = (size_t)gko_pmap_size(pm);
    if (n > cap) return n;
    kv *tmp = (kv *)malloc((n ? n : 1) * sizeof(kv));
    size_t w = 0;
    for (int p = 0; p < pm->P; p++) {
        const container *c = pm->parts[p]->c;
        for (int i = 0; i < c->bins; i++)
            if (c->set[i] && !c->del[i]) { tmp[w].key = c->keys[i]; tmp[w].v = c->ar[i]; w++; }
    }
    qsort(tmp, w, sizeof(kv), kv_cmp);
    for (size_t i = 0; i < w; i++) {
        if (lo) lo[i] = tmp[i].key.lo;
        if (hi) hi[i] = tmp[i].key.hi;
        if (val) val[i] = tmp[i].v;
    }
    free(tmp);
    return w;
}

/* FreqFilter.add :28-36 over PairedEndData.getPairs records :20-36.  Reads shorter than k are
 * skipped (:29); every window in order (`sliding(k)`, :30); canonical choice :31-32; update :33. */
long gko_count_reads(gko_pmap *pm, const uint8_t *bin, size_t nbytes, uint64_t nreads) {
    size_t pos = 0;
    long occ = 0;
    int k = pm->k;
    for (uint64_t r = 0; r < nreads; r++) {
        if (pos >= nbytes) return -1;
        int len = bin[pos++];
        int byteLen = (len + 3) / 4;
        if (pos + (size_t)byteLen > nbytes) return -1;
        const uint8_t *data = bin + pos;
        pos += (size_t)byteLen;
        if (len >= k) {
            for (int p = 0; p + k <= len; p++) {
                gko_kmer x = gko_kmer_from_packed(data, p, k);
                gko_pmap_update_inc(pm, gko_canon(x, k));
                occ++;
            }
        }
    }
    return occ;
}

/* PartitionedDNAMap on P host threads, without the network (BASELINE.md variant BP): phase 1, T
 * reader threads split the records and route every canonical k-mer to a per-(reader, owner) buffer
 * (owner = hashCode mod P, PartitionedDNAMap.scala:60-63 — the driver side of :41-43); phase 2, one
 * thread per partition applies its messages in reader order (the actor side, ArrayDNAMap.scala:39,
 * 198-203).  Per-partition content equals the sequential result (insertion order inside a partition
 * only changes slot order). */
typedef struct { gko_kmer *v; size_t n, cap; } keybuf;
typedef struct {
    gko_pmap *pm; const uint8_t *bin; const size_t *rec_off; uint64_t r0, r1; keybuf *out /* [P] */; long occ;
} mt_reader;
typedef struct { gko_pmap *pm; int p, T; keybuf *bufs /* [T][P] */; } mt_owner;

static void *mt_read_fn(void *arg) {
    mt_reader *a = (mt_reader *)arg;
    const int k = a->pm->k, P = a->pm->P;
    for (uint64_t r = a->r0; r < a->r1; r++) {
        const uint8_t *recp = a->bin + a->rec_off[r];
        const int len = recp[0];
        for (int p = 0; p + k <= len; p++) {
            gko_kmer y = gko_canon(gko_kmer_from_packed(recp + 1, p, k), k);
            keybuf *b = &a->out[gko_partition(y, k, P)];
            if (b->n == b->cap) { b->cap = b->cap ? b->cap * 2 : 4096; b->v = (gko_kmer *)realloc(b->v, b->cap * sizeof(gko_kmer)); }
            b->v[b->n++] = y;
            a->occ++;
        }
    }
    return NULL;
}
static void *mt_own_fn(void *arg) {
    mt_owner *a = (mt_owner *)arg;
    const int P = a->pm->P;
    for (int t = 0; t < a->T; t++) {
        keybuf *b = &a->bufs[(size_t)t * P + a->p];
        for (size_t i = 0; i < b->n; i++) gko_map_update_inc(a->pm->parts[a->p], b->v[i]);
    }
    return NULL;
}

long gko_count_reads_mt(gko_pmap *pm, const uint8_t *bin, size_t nbytes, uint64_t nreads, int nthreads) {
    const int P = pm->P, T = nthreads < 1 ? 1 : nthreads;
    size_t *rec_off = (size_t *)malloc((nreads + 1) * sizeof(size_t));
    size_t pos = 0;
    for (uint64_t r = 0; r < nreads; r++) {
        if (pos >= nbytes) { free(rec_off); return -1; }
        rec_off[r] = pos;
        pos += 1 + (size_t)(bin[pos] + 3) / 4;
        if (pos > nbytes) { free(rec_off); return -1; }
    }
    keybuf *bufs = (keybuf *)calloc((size_t)T * P, sizeof(keybuf));
    mt_reader *rd = (mt_reader *)calloc((size_t)T, sizeof(mt_reader));
    pthread_t *th = (pthread_t *)malloc((size_t)(T > P ? T : P) * sizeof(pthread_t));
    for (int t = 0; t < T; t++) {
        rd[t].pm = pm; rd[t].bin = bin; rd[t].rec_off = rec_off;
        rd[t].r0 = nreads * (uint64_t)t / T; rd[t].r1 = nreads * (uint64_t)(t + 1) / T;
        rd[t].out = bufs + (size_t)t * P;
        pthread_create(&th[t], NULL, mt_read_fn, &rd[t]);
    }
    long occ = 0;
    for (int t = 0; t < T; t++) { pthread_join(th[t], NULL); occ += rd[t].occ; }
    mt_owner *ow = (mt_owner *)calloc((size_t)P, sizeof(mt_owner));
    for (int p = 0; p < P; p++) {
        ow[p].pm = pm; ow[p].p = p; ow[p].T = T; ow[p].bufs = bufs;
        pthread_create(&th[p], NULL, mt_own_fn, &ow[p]);
    }
    for (int p = 0; p < P; p++) pthread_join(th[p], NULL);
    for (size_t i = 0; i < (size_t)T * P; i++) free(bufs[i].v);
    free(bufs); free(rd); free(ow); free(th); free(rec_off);
    return occ;
}

/* ================= Graph (S/data/graph/Graph.scala, Node.scala, Edge.scala) ================= */
typedef struct {
    gko_kmer seq;
    int alive;
    int nin, incap;
    int64_t *in;           /* inEdgeIds: Set[Long] */
    int nout;
    int out_base[4];       /* outEdgeIds: immutable Map1..Map4 keeps insertion order */
    int64_t out_edge[4];
} gnode;

typedef struct {
    int64_t start, end;    /* node ids (1-based) */
    uint8_t *seq;          /* one base code per byte */
    int64_t len;
    int alive;
} gedge;

struct gko_graph {
    int k;
    int64_t nsorted;       /* nodes[1..nsorted] are in ascending k-mer order (buildGraph's); addNode appends behind them */
    int64_t nnodes, ncap;  /* nodes[1..nnodes] */
    gnode *nodes;
    int64_t nedges, ecap;  /* edges[1..nedges] */
    gedge *edges;
    /* k-mer -> node id index: ids are assigned in ascending k-mer order, so binary search */
};

static int64_t graph_find_node(const gko_graph *g, gko_kmer x) {
    int64_t lo = 1, hi = g->nsorted ? g->nsorted : g->nnodes;
    const int64_t sorted = hi;
    while (lo <= hi) {
        int64_t mid = (lo + hi) / 2;
        int c = gko_kmer_cmp(g->nodes[mid].seq, x);
        if (c == 0) {
            if (g->nodes[mid].alive || sorted == g->nnodes) return mid;
            break;                                         /* dead, and nodes were appended: one of those may carry the sequence */
        }
        if (c < 0) lo = mid + 1; else hi = mid - 1;
    }
    for (int64_t i = sorted + 1; i <= g->nnodes; i++)      /* nodes added later (addNode): first live one with this sequence */
        if (g->nodes[i].alive && gko_kmer_cmp(g->nodes[i].seq, x) == 0) return i;
    return 0;
}

/* MapGraph.addEdge :178-184 */
static int64_t graph_add_edge(gko_graph *g, int64_t start, int64_t end, const uint8_t *seq, int64_t len) {
    if (g->nedges + 1 >= g->ecap) {
        g->ecap = g->ecap ? g->ecap * 2 : 64;
        g->edges = (gedge *)realloc(g->edges, (size_t)g->ecap * sizeof(gedge));
    }
    int64_t id = ++g->nedges;
    gedge *e = &g->edges[id];
    e->start = start; e->end = end; e->len = len; e->alive = 1;
    e->seq = (uint8_t *)malloc((size_t)(len ? len : 1));
    memcpy(e->seq, seq, (size_t)len);
    gnode *s = &g->nodes[start];
    int b = seq[0], found = 0;
    for (int i = 0; i < s->nout; i++)          /* Map `+ (k -> v)`: replace in place if key exists */
        if (s->out_base[i] == b) { s->out_edge[i] = id; found = 1; }
    if (!found) { s->out_base[s->nout] = b; s->out_edge[s->nout] = id; s->nout++; }
    gnode *t = &g->nodes[end];
    int have = 0;
    for (int i = 0; i < t->nin; i++) if (t->in[i] == id) have = 1;
    if (!have) {
        if (t->nin == t->incap) {
            t->incap = t->incap ? t->incap * 2 : 4;
            t->in = (int64_t *)realloc(t->in, (size_t)t->incap * sizeof(int64_t));
        }
        t->in[t->nin++] = id;
    }
    return id;
}

/* MapGraph.removeEdge :191-195: start drops its out entry BY BASE KEY, end drops the id */
static void graph_remove_edge(gko_graph *g, int64_t id) {
    gedge *e = &g->edges[id];
    if (g->nodes[e->start].alive) {
        gnode *s = &g->nodes[e->start];
        for (int i = 0; i < s->nout; i++)
            if (s->out_base[i] == e->seq[0]) {
                for (int j = i; j + 1 < s->nout; j++) { s->out_base[j] = s->out_base[j + 1]; s->out_edge[j] = s->out_edge[j + 1]; }
                s->nout--;
                break;
            }
    }
    if (g->nodes[e->end].alive) {
        gnode *t = &g->nodes[e->end];
        for (int i = 0; i < t->nin; i++)
            if (t->in[i] == id) { t->in[i] = t->in[t->nin - 1]; t->nin--; break; }
    }
    e->alive = 0;
}

/* Graph.buildGraph `contains` :270 — either strand, each strand asked of its own partition */
static int g_contains(const gko_pmap *pm, gko_kmer x) {
    return gko_pmap_contains(pm, x) || gko_pmap_contains(pm, gko_revcomp(x, pm->k));
}
/* incoming :272-276 / outcoming :278-282; bases tried in Base.fromInt order A,G,C,T */
static int g_incoming(const gko_pmap *pm, gko_kmer x, int *bases) {
    int n = 0;
    for (int b = 0; b < 4; b++) if (g_contains(pm, gko_prepend(b, x, pm->k))) bases[n++] = b;
    return n;
}
static int g_outcoming(const gko_pmap *pm, gko_kmer x, int *bases) {
    int n = 0;
    for (int b = 0; b < 4; b++) if (g_contains(pm, gko_append(x, b, pm->k))) bases[n++] = b;
    return n;
}

static int kmer_qcmp(const void *a, const void *b) { return gko_kmer_cmp(*(const gko_kmer *)a, *(const gko_kmer *)b); }

gko_graph *gko_graph_build(const gko_pmap *pm) {
    int k = pm->k;
    gko_graph *g = (gko_graph *)calloc(1, sizeof(gko_graph));
    g->k = k;
    /* op1 (:320-329): terminal <=> (in,out) not (1,1) and not (0,0), over every live stored key */
    size_t tcap = 1024, tn = 0;
    gko_kmer *term = (gko_kmer *)malloc(tcap * sizeof(gko_kmer));
    for (int p = 0; p < pm->P; p++) {
        const container *c = pm->parts[p]->c;
        for (int i = 0; i < c->bins; i++) {
            if (!(c->set[i] && !c->del[i])) continue;
            int tmp[4];
            int in = g_incoming(pm, c->keys[i], tmp);
            int out = g_outcoming(pm, c->keys[i], tmp);
            if ((in != 1 || out != 1) && (in != 0 || out != 0)) {
                if (tn + 2 > tcap) { tcap *= 2; term = (gko_kmer *)realloc(term, tcap * sizeof(gko_kmer)); }
                term[tn++] = c->keys[i];
                term[tn++] = gko_revcomp(c->keys[i], k);   /* termKmers = set ++ set.map(revComplement) :330-333 */
            }
        }
    }
    qsort(term, tn, sizeof(gko_kmer), kmer_qcmp);
    size_t un = 0;
    for (size_t i = 0; i < tn; i++) if (un == 0 || gko_kmer_cmp(term[un - 1], term[i]) != 0) term[un++] = term[i];
    /* nodeMap (:343-347): one node per terminal k-mer, ids ascending in k-mer order */
    g->nnodes = (int64_t)un;
    g->nodes = (gnode *)calloc(un + 2, sizeof(gnode));
    for (size_t i = 0; i < un; i++) { g->nodes[i + 1].seq = term[i]; g->nodes[i + 1].alive = 1; }
    free(term);
    /* buildEdges (:349-365) */
    size_t bcap = 256;
    uint8_t *builder = (uint8_t *)malloc(bcap);
    for (int64_t id = 1; id <= g->nnodes; id++) {
        gko_kmer read = g->nodes[id].seq;
        int outs[4];
        int no = g_outcoming(pm, read, outs);
        for (int oi = 0; oi < no; oi++) {
            size_t len = 0;
            builder[len++] = (uint8_t)outs[oi];
            gko_kmer seq = gko_append(read, outs[oi], k);
            int64_t endId;
            while ((endId = graph_find_node(g, seq)) == 0) {
                int o2[4];
                int n2 = g_outcoming(pm, seq, o2);
                if (n2 != 1) { /* reference: assert(out.size == 1) :357 */
                    abort();
                }
                if (len == bcap) { bcap *= 2; builder = (uint8_t *)realloc(builder, bcap); }
                builder[len++] = (uint8_t)o2[0];
                seq = gko_append(seq, o2[0], k);
            }
            graph_add_edge(g, id, endId, builder, (int64_t)len);
        }
    }
    free(builder);
    return g;
}

void gko_graph_free(gko_graph *g) {
    if (!g) return;
    for (int64_t i = 1; i <= g->nnodes; i++) free(g->nodes[i].in);
    for (int64_t i = 1; i <= g->nedges; i++) free(g->edges[i].seq);
    free(g->nodes); free(g->edges); free(g);
}
int gko_graph_k(const gko_graph *g) { return g->k; }
long gko_graph_num_nodes(const gko_graph *g) {
    long n = 0;
    for (int64_t i = 1; i <= g->nnodes; i++) n += g->nodes[i].alive;
    return n;
}
long gko_graph_num_edges(const gko_graph *g) {
    long n = 0;
    for (int64_t i = 1; i <= g->nedges; i++) n += g->edges[i].alive;
    return n;
}
long gko_graph_total_edge_len(const gko_graph *g) {
    long n = 0;
    for (int64_t i = 1; i <= g->nedges; i++) if (g->edges[i].alive) n += g->edges[i].len;
    return n;
}

/* MapGraph.simplifyGraph :211-230 */
void gko_graph_simplify(gko_graph *g) {
    for (int64_t id = 1; id <= g->nnodes; id++) {
        gnode *n = &g->nodes[id];
        if (!n->alive) continue;
        if (n->nin == 0 && n->nout == 0) {
            n->alive = 0;                                   /* removeNode */
        } else if (n->nin == 1 && n->nout == 1) {
            int64_t e1 = n->in[0], e2 = n->out_edge[0];
            if (e1 == e2) {
                graph_remove_edge(g, e1);
            } else {
                graph_remove_edge(g, e1);
                graph_remove_edge(g, e2);
                int64_t l1 = g->edges[e1].len, l2 = g->edges[e2].len;
                uint8_t *s = (uint8_t *)malloc((size_t)(l1 + l2));
                memcpy(s, g->edges[e1].seq, (size_t)l1);
                memcpy(s + l1, g->edges[e2].seq, (size_t)l2);
                int64_t st = g->edges[e1].start, en = g->edges[e2].end;
                graph_add_edge(g, st, en, s, l1 + l2);     /* may realloc g->edges */
                free(s);
                n = &g->nodes[id];
            }
            n->alive = 0;                                   /* removeNode */
        }
    }
}

/* Graph.similar :121-123 */
static int similar_len(int64_t a, int64_t b) {
    int64_t d = a > b ? a - b : b - a;
    int64_t mx = a > b ? a : b;
    return d * 5 < mx;
}

/* Graph.removeBubbles :125-149 */
void gko_graph_remove_bubbles(gko_graph *g) {
    for (int64_t id = 1; id <= g->nnodes; id++) {
        gnode *n = &g->nodes[id];
        if (!n->alive) continue;
        int cnt = n->nout;
        int64_t out[4];
        int removed[4] = {0, 0, 0, 0};
        for (int i = 0; i < cnt; i++) out[i] = n->out_edge[i];
        for (int i = 0; i < cnt; i++) {
            if (removed[i]) continue;
            for (int j = i + 1; j < cnt; j++)
                if (g->edges[out[i]].end == g->edges[out[j]].end &&
                    similar_len(g->edges[out[i]].len, g->edges[out[j]].len))
                    removed[j] = 1;
        }
        for (int j = 0; j < cnt; j++) if (removed[j]) graph_remove_edge(g, out[j]);
    }
}

int gko_graph_remove_edge(gko_graph *g, gko_kmer start, int base) {
    int64_t id = graph_find_node(g, start);
    if (!id || !g->nodes[id].alive) return 0;
    gnode *n = &g->nodes[id];
    for (int i = 0; i < n->nout; i++)
        if (n->out_base[i] == base) { graph_remove_edge(g, n->out_edge[i]); return 1; }
    return 0;
}

/* Graph.components :54-72 = connected components of the undirected node graph */
static int64_t *graph_component_labels(const gko_graph *g, int64_t *ncomp_out) {
    int64_t *label = (int64_t *)calloc((size_t)g->nnodes + 2, sizeof(int64_t));
    int64_t *stack = (int64_t *)malloc(((size_t)g->nnodes + 2) * sizeof(int64_t));
    /* undirected adjacency through live edges */
    int64_t ncomp = 0;
    /* build CSR of neighbours */
    int64_t *deg = (int64_t *)calloc((size_t)g->nnodes + 2, sizeof(int64_t));
    for (int64_t e = 1; e <= g->nedges; e++) if (g->edges[e].alive) { deg[g->edges[e].start]++; deg[g->edges[e].end]++; }
    int64_t *off = (int64_t *)calloc((size_t)g->nnodes + 3, sizeof(int64_t));
    for (int64_t i = 1; i <= g->nnodes; i++) off[i + 1] = off[i] + deg[i];
    int64_t *adj = (int64_t *)malloc(((size_t)off[g->nnodes + 1] + 1) * sizeof(int64_t));
    memset(deg, 0, ((size_t)g->nnodes + 2) * sizeof(int64_t));
    for (int64_t e = 1; e <= g->nedges; e++) if (g->edges[e].alive) {
        int64_t s = g->edges[e].start, t = g->edges[e].end;
        adj[off[s] + deg[s]++] = t;
        adj[off[t] + deg[t]++] = s;
    }
    for (int64_t i = 1; i <= g->nnodes; i++) {
        if (!g->nodes[i].alive || label[i]) continue;
        ncomp++;
        int64_t sp = 0;
        stack[sp++] = i;
        label[i] = ncomp;
        while (sp) {
            int64_t u = stack[--sp];
            for (int64_t a = off[u]; a < off[u + 1]; a++) {
                int64_t v = adj[a];
                if (g->nodes[v].alive && !label[v]) { label[v] = ncomp; stack[sp++] = v; }
            }
        }
    }
    free(stack); free(deg); free(off); free(adj);
    *ncomp_out = ncomp;
    return label;
}

long gko_graph_num_components(const gko_graph *g) {
    int64_t nc;
    int64_t *l = graph_component_labels(g, &nc);
    free(l);
    return (long)nc;
}

/* GraphBuilder.scala:52-54 maxBy(_.size) + MapGraph.retain :161-165 */
long gko_graph_retain_largest(gko_graph *g) {
    int64_t nc;
    int64_t *label = graph_component_labels(g, &nc);
    if (nc == 0) { free(label); return 0; }
    int64_t *sz = (int64_t *)calloc((size_t)nc + 1, sizeof(int64_t));
    for (int64_t i = 1; i <= g->nnodes; i++) if (g->nodes[i].alive) sz[label[i]]++;
    /* labels were handed out in ascending node id = ascending k-mer, so the first maximum is
     * the component holding the smallest k-mer among the largest ones */
    int64_t best = 1;
    for (int64_t c = 2; c <= nc; c++) if (sz[c] > sz[best]) best = c;
    for (int64_t e = 1; e <= g->nedges; e++)
        if (g->edges[e].alive && !(label[g->edges[e].start] == best && label[g->edges[e].end] == best))
            g->edges[e].alive = 0;
    for (int64_t i = 1; i <= g->nnodes; i++)
        if (g->nodes[i].alive && label[i] != best) g->nodes[i].alive = 0;
    long r = (long)sz[best];
    free(sz); free(label);
    return r;
}

size_t gko_graph_export_nodes(const gko_graph *g, uint64_t *lo, uint64_t *hi, size_t cap) {
    size_t n = 0;
    for (int64_t i = 1; i <= g->nnodes; i++) {
        if (!g->nodes[i].alive) continue;
        if (n < cap) { if (lo) lo[n] = g->nodes[i].seq.lo; if (hi) hi[n] = g->nodes[i].seq.hi; }
        n++;
    }
    return n;
}

typedef struct { gko_kmer s; int b; int64_t id; } esort;
static int esort_cmp(const void *a, const void *b) {
    const esort *x = (const esort *)a, *y = (const esort *)b;
    int c = gko_kmer_cmp(x->s, y->s);
    if (c) return c;
    if (x->b != y->b) return x->b < y->b ? -1 : 1;
    return x->id < y->id ? -1 : (x->id > y->id);
}

size_t gko_graph_export_edges(const gko_graph *g, uint64_t *slo, uint64_t *shi, uint64_t *elo,
                              uint64_t *ehi, int64_t *len, int64_t *off, size_t cap,
                              uint8_t *bases_out, size_t bases_cap, size_t *nbases_out) {
    size_t ne = (size_t)gko_graph_num_edges(g);
    esort *es = (esort *)malloc((ne ? ne : 1) * sizeof(esort));
    size_t w = 0;
    for (int64_t e = 1; e <= g->nedges; e++) if (g->edges[e].alive) {
        es[w].s = g->nodes[g->edges[e].start].seq;
        es[w].b = g->edges[e].seq[0];
        es[w].id = e;
        w++;
    }
    qsort(es, w, sizeof(esort), esort_cmp);
    size_t nb = 0;
    for (size_t i = 0; i < w; i++) {
        const gedge *e = &g->edges[es[i].id];
        if (i < cap) {
            if (slo) slo[i] = g->nodes[e->start].seq.lo;
            if (shi) shi[i] = g->nodes[e->start].seq.hi;
            if (elo) elo[i] = g->nodes[e->end].seq.lo;
            if (ehi) ehi[i] = g->nodes[e->end].seq.hi;
            if (len) len[i] = e->len;
            if (off) off[i] = (int64_t)nb;
            if (bases_out && nb + (size_t)e->len <= bases_cap) memcpy(bases_out + nb, e->seq, (size_t)e->len);
        }
        nb += (size_t)e->len;
    }
    free(es);
    if (nbases_out) *nbases_out = nb;
    return w;
}

int gko_graph_out_order(const gko_graph *g, gko_kmer node, int *bases4) {
    int64_t id = graph_find_node(g, node);
    if (!id || !g->nodes[id].alive) return -1;
    for (int i = 0; i < g->nodes[id].nout; i++) bases4[i] = g->nodes[id].out_base[i];
    return g->nodes[id].nout;
}

int gko_graph_degree(const gko_graph *g, gko_kmer node, int *in_deg, int *out_deg) {
    int64_t id = graph_find_node(g, node);
    if (!id || !g->nodes[id].alive) return -1;
    if (in_deg) *in_deg = g->nodes[id].nin;
    if (out_deg) *out_deg = g->nodes[id].nout;
    return 0;
}

/* ---- point edits by id and Graph.getGraphMap (GraphSimplifier's tools) ---- */
int64_t gko_graph_find_node(const gko_graph *g, gko_kmer x) {
    int64_t id = graph_find_node(g, x);
    return id && g->nodes[id].alive ? id : 0;
}
int64_t gko_graph_find_out_edge(const gko_graph *g, int64_t node, int base) {
    if (node < 1 || node > g->nnodes || !g->nodes[node].alive) return 0;
    const gnode *n = &g->nodes[node];
    for (int i = 0; i < n->nout; i++) if (n->out_base[i] == base) return n->out_edge[i];
    return 0;
}
/* MapGraph.addNode :172-176 */
int64_t gko_graph_add_node(gko_graph *g, gko_kmer seq) {
    if (!g->nsorted) g->nsorted = g->nnodes;
    if (g->nnodes + 2 >= g->ncap) {      /* (buildGraph sizes the array exactly and leaves ncap at what it allocated, possibly 0) */
        g->ncap = (g->ncap > g->nnodes ? g->ncap : g->nnodes) * 2 + 64;
        g->nodes = (gnode *)realloc(g->nodes, (size_t)g->ncap * sizeof(gnode));
    }
    gnode *n = &g->nodes[++g->nnodes];
    memset(n, 0, sizeof(*n));
    n->seq = seq;
    n->alive = 1;
    return g->nnodes;
}
/* MapGraph.replaceStart :197-202 */
void gko_graph_replace_start(gko_graph *g, int64_t edge, int64_t new_start) {
    gedge *e = &g->edges[edge];
    gnode *s = &g->nodes[e->start];
    const int b = e->seq[0];
    for (int i = 0; i < s->nout; i++)                      /* edge.start.outEdgeIds -= edge.seq(0) */
        if (s->out_base[i] == b) {
            for (int j = i; j + 1 < s->nout; j++) { s->out_base[j] = s->out_base[j + 1]; s->out_edge[j] = s->out_edge[j + 1]; }
            s->nout--;
            break;
        }
    gnode *t = &g->nodes[new_start];                       /* newStart.outEdgeIds += edge.seq(0) -> edge.id */
    int found = 0;
    for (int i = 0; i < t->nout; i++) if (t->out_base[i] == b) { t->out_edge[i] = edge; found = 1; }
    if (!found) { t->out_base[t->nout] = b; t->out_edge[t->nout] = edge; t->nout++; }
    e->start = new_start;                                  /* edges(edge.id) = new Edge(edge.id, newStart.id, ...) */
}
/* MapGraph.replaceEnd :204-209 */
void gko_graph_replace_end(gko_graph *g, int64_t edge, int64_t new_end) {
    gedge *e = &g->edges[edge];
    gnode *t = &g->nodes[e->end];
    for (int i = 0; i < t->nin; i++) if (t->in[i] == edge) { t->in[i] = t->in[t->nin - 1]; t->nin--; break; }
    gnode *n = &g->nodes[new_end];
    if (n->nin == n->incap) {
        n->incap = n->incap ? n->incap * 2 : 4;
        n->in = (int64_t *)realloc(n->in, (size_t)n->incap * sizeof(int64_t));
    }
    n->in[n->nin++] = edge;
    e->end = new_end;
}
/* Graph.getGraphMap :90-119 as the SEQUENCE of its putNew calls: for node in getNodes: (node.seq, NodeGraphPosition(id));
 * for edge in getEdges: seq = start.seq.drop(1) :+ edge.seq.head; dist = 1; for base in edge.seq.tail: (seq, EdgeGraphPosition(id, dist));
 * seq = seq.drop(1) :+ base; dist += 1.  Returns the number of calls; fills up to cap. */
size_t gko_graph_get_graph_map(const gko_graph *g, uint64_t *lo, uint64_t *hi, uint8_t *is_edge, int64_t *id, int32_t *dist, size_t cap) {
    size_t n = 0;
    for (int64_t i = 1; i <= g->nnodes; i++) {
        if (!g->nodes[i].alive) continue;
        if (n < cap) { lo[n] = g->nodes[i].seq.lo; hi[n] = g->nodes[i].seq.hi; is_edge[n] = 0; id[n] = i; dist[n] = 0; }
        n++;
    }
    for (int64_t e = 1; e <= g->nedges; e++) {
        const gedge *ed = &g->edges[e];
        if (!ed->alive) continue;
        gko_kmer seq = gko_append(g->nodes[ed->start].seq, ed->seq[0], g->k);
        int32_t d = 1;
        for (int64_t j = 1; j < ed->len; j++) {
            if (n < cap) { lo[n] = seq.lo; hi[n] = seq.hi; is_edge[n] = 1; id[n] = e; dist[n] = d; }
            n++;
            seq = gko_append(seq, ed->seq[j], g->k);
            d++;
        }
    }
    return n;
}
gko_kmer gko_graph_node_seq(const gko_graph *g, int64_t id) { return g->nodes[id].seq; }
int gko_graph_edge_info(const gko_graph *g, int64_t id, int64_t *start, int64_t *end, int64_t *len, int *first) {
    if (id < 1 || id > g->nedges) return 0;
    const gedge *e = &g->edges[id];
    if (start) *start = e->start;
    if (end) *end = e->end;
    if (len) *len = e->len;
    if (first) *first = e->seq[0];
    return e->alive;
}


/* ================= Paired-end walking (S/scripts/GraphSimplifier.scala) ======================================
 * A LITERAL restatement: reachable() is the priority-queue search of :43-72 (without its cache, which only saves time),
 * dfs() the memoised recursion of :90-112, annotate :192-206, the per-pair loop :213-247, the support matrix and the
 * node split :272-313, removeEdge + simplifyGraph :316-318.  Test infrastructure only. */
typedef struct { int is_edge; int64_t id; int32_t dist; } gpos;

/* getGraphMap as a multimap on the host: open addressing over the putNew calls */
typedef struct { gko_kmer key; gpos pos; int used; } gm_ent;
typedef struct { gm_ent *t; size_t cap; } gmap;
static uint64_t gm_hash(gko_kmer x) { uint64_t h = x.lo * 0x9E3779B97F4A7C15ull ^ (x.hi + 0x7F4A7C15ull) * 0xC2B2AE3D27D4EB4Full; return h ^ (h >> 29); }
static gmap gmap_build(const gko_graph *g) {
    size_t n = gko_graph_get_graph_map(g, NULL, NULL, NULL, NULL, NULL, 0);
    uint64_t *lo = (uint64_t *)malloc((n + 1) * 8), *hi = (uint64_t *)malloc((n + 1) * 8);
    uint8_t *ie = (uint8_t *)malloc(n + 1);
    int64_t *id = (int64_t *)malloc((n + 1) * 8);
    int32_t *ds = (int32_t *)malloc((n + 1) * 4);
    gko_graph_get_graph_map(g, lo, hi, ie, id, ds, n);
    gmap m;
    m.cap = 16;
    while (m.cap < 2 * n + 16) m.cap *= 2;
    m.t = (gm_ent *)calloc(m.cap, sizeof(gm_ent));
    for (size_t i = 0; i < n; i++) {
        gko_kmer key = {lo[i], hi[i]};
        size_t p = gm_hash(key) & (m.cap - 1);
        while (m.t[p].used) p = (p + 1) & (m.cap - 1);           /* putNew: no duplicate check */
        m.t[p].used = 1; m.t[p].key = key;
        m.t[p].pos.is_edge = ie[i]; m.t[p].pos.id = id[i]; m.t[p].pos.dist = ds[i];
    }
    free(lo); free(hi); free(ie); free(id); free(ds);
    return m;
}
static int gmap_get_all(const gmap *m, gko_kmer key, gpos *out, int cap) {
    int n = 0;
    for (size_t p = gm_hash(key) & (m->cap - 1); m->t[p].used; p = (p + 1) & (m->cap - 1))
        if (kmer_eq(m->t[p].key, key)) { if (n < cap) out[n] = m->t[p].pos; n++; }
    return n;
}

/* a small set / map of 128-bit keys (two int64) -> int32 */
typedef struct { int64_t a, b; int32_t v; int used; } pm_ent;
typedef struct { pm_ent *t; size_t cap, n; } pmap2;
static void pm2_init(pmap2 *m, size_t cap) { m->cap = 16; while (m->cap < cap) m->cap *= 2; m->t = (pm_ent *)calloc(m->cap, sizeof(pm_ent)); m->n = 0; }
static void pm2_free(pmap2 *m) { free(m->t); m->t = NULL; }
static pm_ent *pm2_slot(pmap2 *m, int64_t a, int64_t b, int create);
static void pm2_grow(pmap2 *m) {
    pmap2 n2; pm2_init(&n2, m->cap * 2);
    for (size_t i = 0; i < m->cap; i++) if (m->t[i].used) { pm_ent *e = pm2_slot(&n2, m->t[i].a, m->t[i].b, 1); e->v = m->t[i].v; }
    free(m->t); *m = n2;
}
static pm_ent *pm2_slot(pmap2 *m, int64_t a, int64_t b, int create) {
    if (create && (m->n + 1) * 2 > m->cap) pm2_grow(m);
    uint64_t h = (uint64_t)a * 0x9E3779B97F4A7C15ull ^ ((uint64_t)b + 0x632BE59BD9B4E019ull) * 0xC2B2AE3D27D4EB4Full;
    size_t p = (h ^ (h >> 31)) & (m->cap - 1);
    while (m->t[p].used) {
        if (m->t[p].a == a && m->t[p].b == b) return &m->t[p];
        p = (p + 1) & (m->cap - 1);
    }
    if (!create) return NULL;
    m->t[p].used = 1; m->t[p].a = a; m->t[p].b = b; m->t[p].v = 0; m->n++;
    return &m->t[p];
}

/* reachable(node) :43-72 — distances back along inEdges, bounded by range.last; a binary heap of (dist, node) */
typedef struct { int32_t d; int64_t u; } hq_ent;
static void reachable(const gko_graph *g, int64_t node, int range_hi, pmap2 *set) {
    size_t hn = 0, hcap = 64;
    hq_ent *hq = (hq_ent *)malloc(hcap * sizeof(hq_ent));
    hq[hn++] = (hq_ent){0, node};
    while (hn) {
        hq_ent top = hq[0];                                   /* dequeue the smallest dist (ordering y._1 - x._1 :49) */
        hq[0] = hq[--hn];
        for (size_t i = 0;;) {
            size_t l = 2 * i + 1, r = l + 1, s = i;
            if (l < hn && hq[l].d < hq[s].d) s = l;
            if (r < hn && hq[r].d < hq[s].d) s = r;
            if (s == i) break;
            hq_ent t = hq[i]; hq[i] = hq[s]; hq[s] = t; i = s;
        }
        if (pm2_slot(set, top.u, 0, 0)) continue;             /* if (!set.contains(u)) */
        pm2_slot(set, top.u, 0, 1)->v = top.d;                /* set += u -> dist */
        const gnode *u = &g->nodes[top.u];
        for (int i = 0; i < u->nin; i++) {
            const gedge *e = &g->edges[u->in[i]];
            const int64_t d2 = (int64_t)top.d + e->len;
            if (d2 <= range_hi) {
                if (hn == hcap) { hcap *= 2; hq = (hq_ent *)realloc(hq, hcap * sizeof(hq_ent)); }
                size_t c = hn++;
                hq[c] = (hq_ent){(int32_t)d2, e->start};
                while (c && hq[(c - 1) / 2].d > hq[c].d) { hq_ent t = hq[c]; hq[c] = hq[(c - 1) / 2]; hq[(c - 1) / 2] = t; c = (c - 1) / 2; }
            }
        }
    }
    free(hq);
}

typedef struct {
    const gko_graph *g;
    int range_lo, range_hi;
    int64_t node2; int32_t dist2; int64_t end_edge;      /* 0 = null */
    pmap2 reach, memo, *path_edges;
} walk_ctx;
/* dfs(node1, dist1, prevEdge) :90-112 */
static int walk_dfs(walk_ctx *w, int64_t node1, int32_t dist1, int64_t prev_edge) {
    pm_ent *m = pm2_slot(&w->memo, prev_edge, dist1, 0);
    if (m) return m->v;
    pm_ent *r = pm2_slot(&w->reach, node1, 0, 0);
    const int64_t rd = r ? r->v : (int64_t)w->range_hi + 1;
    if ((int64_t)dist1 + w->dist2 + rd > w->range_hi) return 0;
    int cur = 0;
    if (node1 == w->node2 && dist1 + w->dist2 >= w->range_lo && dist1 + w->dist2 <= w->range_hi) {
        if (prev_edge && w->end_edge) pm2_slot(w->path_edges, prev_edge, w->end_edge, 1)->v = 1;
        cur = 1;
    }
    const gnode *n = &w->g->nodes[node1];
    for (int i = 0; i < n->nout; i++) {
        const int64_t e = n->out_edge[i];
        const int res = walk_dfs(w, w->g->edges[e].end, (int32_t)(dist1 + w->g->edges[e].len), e);
        if (res && prev_edge) pm2_slot(w->path_edges, prev_edge, e, 1)->v = 1;
        cur |= res;
    }
    pm2_slot(&w->memo, prev_edge, dist1, 1)->v = cur;
    return cur;
}
/* WalkingActor.receive :78-125 for one (pos1, pos2): adds to path_edges, returns `good` */
static int walk_one(const gko_graph *g, gpos pos1, gpos pos2, int range_lo, int range_hi, pmap2 *path_edges) {
    walk_ctx w;
    w.g = g; w.range_lo = range_lo; w.range_hi = range_hi; w.path_edges = path_edges;
    w.node2 = pos2.is_edge ? g->edges[pos2.id].start : pos2.id;
    w.dist2 = pos2.is_edge ? pos2.dist : 0;
    w.end_edge = pos2.is_edge ? pos2.id : 0;
    const int64_t start_edge = pos1.is_edge ? pos1.id : 0;
    pm2_init(&w.reach, 64); pm2_init(&w.memo, 64);
    reachable(g, w.node2, range_hi, &w.reach);
    const int64_t node0 = pos1.is_edge ? g->edges[pos1.id].end : pos1.id;
    const int32_t dist0 = pos1.is_edge ? (int32_t)(g->edges[pos1.id].len - pos1.dist) : 0;
    const int good = walk_dfs(&w, node0, dist0, start_edge);
    pm2_free(&w.reach); pm2_free(&w.memo);
    return good;
}

struct gko_support { pmap2 paths; long bad_pairs; };
gko_support *gko_support_new(void) { gko_support *s = (gko_support *)calloc(1, sizeof(*s)); pm2_init(&s->paths, 1024); return s; }
void gko_support_free(gko_support *s) { if (s) { pm2_free(&s->paths); free(s); } }
long gko_support_bad_pairs(const gko_support *s) { return s->bad_pairs; }
size_t gko_support_export(const gko_support *s, int64_t *e1, int64_t *e2, int32_t *cnt, size_t cap) {
    size_t n = 0;
    for (size_t i = 0; i < s->paths.cap; i++) if (s->paths.t[i].used) {
        if (n < cap) { e1[n] = s->paths.t[i].a; e2[n] = s->paths.t[i].b; cnt[n] = s->paths.t[i].v; }
        n++;
    }
    return n;
}

/* the body of GraphSimplifier.startup :188-247 over the first `npairs` pairs of a `.bin` stream (two records per pair) */
long gko_graph_walk_pairs(const gko_graph *g, gko_support *sup, const uint8_t *bin, size_t nbytes, uint64_t npairs, int range_lo, int range_hi) {
    const int k = g->k;
    gmap gm = gmap_build(g);
    size_t pos = 0;
    long walked = 0;
    gpos *f[4];
    int fcap[4] = {64, 64, 64, 64};
    for (int i = 0; i < 4; i++) f[i] = (gpos *)malloc((size_t)fcap[i] * sizeof(gpos));
    for (uint64_t p = 0; p < npairs && pos < nbytes; p++) {
        const uint8_t *r1 = bin + pos; const int l1 = r1[0]; pos += 1 + (size_t)(l1 + 3) / 4;
        if (pos >= nbytes) break;
        const uint8_t *r2 = bin + pos; const int l2 = r2[0]; pos += 1 + (size_t)(l2 + 3) / 4;
        if (l1 < k || l2 < k) continue;                                           /* :213 */
        const gko_kmer a = gko_kmer_from_packed(r1 + 1, 0, k), b = gko_kmer_from_packed(r2 + 1, 0, k);
        int n[4];
        const gko_kmer keys[4] = {a,                      /* f1 = getAll(p1.take(k))               :214 */
                                  gko_revcomp(b, k),      /* f2 = getAll(p2.take(k).revComplement) :215 */
                                  b,                      /* f3 = getAll(p2.take(k))               :216 */
                                  gko_revcomp(a, k)};     /* f4 = getAll(p1.take(k).revComplement) :217 */
        for (int i = 0; i < 4; i++) {
            n[i] = gmap_get_all(&gm, keys[i], f[i], fcap[i]);
            if (n[i] > fcap[i]) {
                fcap[i] = n[i] * 2;
                f[i] = (gpos *)realloc(f[i], (size_t)fcap[i] * sizeof(gpos));
                n[i] = gmap_get_all(&gm, keys[i], f[i], fcap[i]);
            }
        }
        for (int o = 0; o < 2; o++) {                                               /* List((s1, s2), (s3, s4)) :219 */
            const gpos *p1 = f[2 * o], *p2 = f[2 * o + 1];
            const int n1 = n[2 * o], n2 = n[2 * o + 1];
            int same_edge = 0;                                                      /* annotate :192-206 */
            for (int i = 0; i < n1 && !same_edge; i++) if (p1[i].is_edge)
                for (int j = 0; j < n2; j++) if (p2[j].is_edge && p1[i].id == p2[j].id) {
                    const int d = (p2[j].dist - p1[i].dist) + k;
                    if (d >= range_lo && d <= range_hi) { same_edge = 1; break; }
                }
            if (same_edge) continue;
            if (n1 == 0 || n2 == 0) continue;                                       /* `if !list.isEmpty` :231 */
            pmap2 path_edges; pm2_init(&path_edges, 64);
            int good = 0;
            for (int i = 0; i < n1; i++) for (int j = 0; j < n2; j++) good |= walk_one(g, p1[i], p2[j], range_lo, range_hi, &path_edges);
            for (size_t i = 0; i < path_edges.cap; i++) if (path_edges.t[i].used)
                pm2_slot(&sup->paths, path_edges.t[i].a, path_edges.t[i].b, 1)->v++;   /* counter.incrementAndGet() :240 */
            if (!good) sup->bad_pairs++;
            pm2_free(&path_edges);
            walked++;
        }
    }
    for (int i = 0; i < 4; i++) free(f[i]);
    free(gm.t);
    return walked;
}

/* :272-318: support matrix per node, connected groups at `cutoff`, node split, removeEdge(toRemove), simplifyGraph.
 * Nodes in ascending id, only those that existed before the loop (the reference iterates a live map while it adds to it:
 * whether a copy is visited is unspecified, and a visited copy only produces another copy with the same edges). */
static void split_dfs_left(int i, int nin, int nout, const int32_t *mx, int cutoff, int *col_l, int *col_r, int *l, int *nl, int *r, int *nr);
static void split_dfs_right(int j, int nin, int nout, const int32_t *mx, int cutoff, int *col_l, int *col_r, int *l, int *nl, int *r, int *nr) {
    col_r[j] = 1; r[(*nr)++] = j;
    for (int i = 0; i < nin; i++) if (!col_l[i] && mx[i * nout + j] >= cutoff) split_dfs_left(i, nin, nout, mx, cutoff, col_l, col_r, l, nl, r, nr);
}
static void split_dfs_left(int i, int nin, int nout, const int32_t *mx, int cutoff, int *col_l, int *col_r, int *l, int *nl, int *r, int *nr) {
    col_l[i] = 1; l[(*nl)++] = i;
    for (int j = 0; j < nout; j++) if (!col_r[j] && mx[i * nout + j] >= cutoff) split_dfs_right(j, nin, nout, mx, cutoff, col_l, col_r, l, nl, r, nr);
}
void gko_graph_split_by_support(gko_graph *g, const gko_support *sup, int cutoff, long *removed_edges, long *new_nodes) {
    const int64_t n0 = g->nnodes;
    int64_t *to_remove = NULL; size_t nrm = 0, rmcap = 0;
    long added = 0;
#define PUSH_RM(id) do { if (nrm == rmcap) { rmcap = rmcap ? rmcap * 2 : 64; to_remove = (int64_t *)realloc(to_remove, rmcap * 8); } to_remove[nrm++] = (id); } while (0)
    for (int64_t v = 1; v <= n0; v++) {
        if (!g->nodes[v].alive) continue;
        const int nin = g->nodes[v].nin, nout = g->nodes[v].nout;
        if (nin == 0 || nout == 0) continue;                                         /* :273 */
        int64_t *in = (int64_t *)malloc((size_t)nin * 8), out[4];
        memcpy(in, g->nodes[v].in, (size_t)nin * 8);
        for (int j = 0; j < nout; j++) out[j] = g->nodes[v].out_edge[j];
        int32_t *mx = (int32_t *)calloc((size_t)nin * nout, 4);
        for (int i = 0; i < nin; i++) for (int j = 0; j < nout; j++) {
            pm_ent *e = pm2_slot((pmap2 *)&sup->paths, in[i], out[j], 0);
            mx[i * nout + j] = e ? e->v : 0;
        }
        int *col_l = (int *)calloc((size_t)nin, sizeof(int)), col_r[4] = {0, 0, 0, 0};
        int *l = (int *)malloc((size_t)nin * sizeof(int)), r[4];
        const gko_kmer seq = g->nodes[v].seq;
        for (int i = 0; i < nin; i++) {
            if (col_l[i]) continue;
            int nl = 0, nr = 0;
            split_dfs_left(i, nin, nout, mx, cutoff, col_l, col_r, l, &nl, r, &nr);
            if (nr == 0) PUSH_RM(in[i]);                                             /* toRemove += in(i) :304 */
            else {
                const int64_t nn = gko_graph_add_node(g, seq);                       /* :306 */
                added++;
                for (int q = 0; q < nl; q++) gko_graph_replace_end(g, in[l[q]], nn);    /* :307 */
                for (int q = 0; q < nr; q++) gko_graph_replace_start(g, out[r[q]], nn); /* :308 */
            }
        }
        for (int j = 0; j < nout; j++) if (!col_r[j]) PUSH_RM(out[j]);                /* :311 */
        free(in); free(mx); free(col_l); free(l);
    }
    long removed = 0;
    for (size_t i = 0; i < nrm; i++) if (g->edges[to_remove[i]].alive) { graph_remove_edge(g, to_remove[i]); removed++; }   /* :316 (a Set: once each) */
    free(to_remove);
    gko_graph_simplify(g);                                                            /* :318 */
    if (removed_edges) *removed_edges = removed;
    if (new_nodes) *new_nodes = added;
#undef PUSH_RM
}
