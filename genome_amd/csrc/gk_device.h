// gk_device.h — k-mer arithmetic and open-addressed table primitives shared by the HIP kernels
// (and, for the pure-arithmetic part, by the host side of the C-ABI).
//
// Everything here is integer/bit work laid out for gfx950: one lane = one k-mer, 64-bit VALU ops,
// no MFMA anywhere.  Reference semantics are cited per function
// (S/ = /root/reference/src/main/scala/ru/ifmo/genome/).
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GK_HD __host__ __device__ __forceinline__
#define GK_D __device__ __forceinline__
#else
#define GK_HD inline
#endif

namespace gk {

typedef uint32_t u32;
typedef uint64_t u64;
typedef int32_t i32;
typedef int64_t i64;

// ------------------------------------------------------------------------------------------
// K-mers.  W = number of 64-bit words: 1 for k<=31 (Long1DNASeq, DNASeq.scala:74-169),
// 2 for 34<=k<=63 (Long2DNASeq, :172-215).  Base i sits at bits 2i, first base at the LSB,
// codes A0 G1 C2 T3 (Base.scala:13-18); unused high bits are zero.
// ------------------------------------------------------------------------------------------
template <int W> struct Kmer;
template <> struct Kmer<1> { u64 lo; };
template <> struct Kmer<2> { u64 lo, hi; };

GK_HD u64 low_mask(int nbits) { return nbits >= 64 ? ~0ULL : ((1ULL << nbits) - 1ULL); }

GK_HD bool operator==(Kmer<1> a, Kmer<1> b) { return a.lo == b.lo; }
GK_HD bool operator==(Kmer<2> a, Kmer<2> b) { return a.lo == b.lo && a.hi == b.hi; }
GK_HD bool kmer_less(Kmer<1> a, Kmer<1> b) { return a.lo < b.lo; }
GK_HD bool kmer_less(Kmer<2> a, Kmer<2> b) { return a.hi != b.hi ? a.hi < b.hi : a.lo < b.lo; }

GK_HD u64 bitrev64(u64 v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __brevll(v);
#else
    v = ((v >> 1) & 0x5555555555555555ULL) | ((v & 0x5555555555555555ULL) << 1);
    v = ((v >> 2) & 0x3333333333333333ULL) | ((v & 0x3333333333333333ULL) << 2);
    v = ((v >> 4) & 0x0f0f0f0f0f0f0f0fULL) | ((v & 0x0f0f0f0f0f0f0f0fULL) << 4);
    return __builtin_bswap64(v);
#endif
}
// reverse the order of the 32 two-bit groups of a word
GK_HD u64 rev_groups(u64 v) {
    v = bitrev64(v);
    return ((v >> 1) & 0x5555555555555555ULL) | ((v & 0x5555555555555555ULL) << 1);
}

// revComplement = complement.reverse (DNASeq.scala:28; complement is b^3, Base.scala:19).
// ~x complements every base and turns the unused high groups into garbage that the final shift
// discards.
GK_HD Kmer<1> revcomp(Kmer<1> x, int k) {
    Kmer<1> r;
    r.lo = rev_groups(~x.lo) >> (64 - 2 * k);
    return r;
}
GK_HD Kmer<2> revcomp(Kmer<2> x, int k) {   // 34 <= k <= 64  =>  shift s in [0, 60]
    u64 nlo = rev_groups(~x.hi), nhi = rev_groups(~x.lo);
    int s = 128 - 2 * k;
    Kmer<2> r;
    if (s == 0) { r.lo = nlo; r.hi = nhi; return r; }
    r.lo = (nlo >> s) | (nhi << (64 - s));
    r.hi = nhi >> s;
    return r;
}

// hashCode.  k<=31: scala-library 2.9.1 `Long.##` (DNASeq.scala:103) — for the non-negative
// values a k<=31 k-mer takes this is (int)(v ^ (v >>> 32)) under either branch of
// BoxesRunTime.hashFromLong.  k>32: MultiHash.hashCode = multiHashCode(42)
// (BloomFilter.scala:12-15) with Long2DNASeq.multiHashCode (DNASeq.scala:204-208): arithmetic >>,
// wrapping 64-bit multiply.
GK_HD i32 ref_hash(Kmer<1> x) { return (i32)(u32)(x.lo ^ (x.lo >> 32)); }
GK_HD i32 ref_hash(Kmer<2> x) {
    i64 l1 = (i64)x.lo, l2 = (i64)x.hi;
    i64 t = (i64)((u64)(l1 ^ (l1 >> 32)) * 42ULL);
    i64 t1 = (i64)((u64)(l2 ^ (l2 >> 32) ^ t) * 42ULL);
    return (i32)(u32)(u64)(t1 ^ (t1 >> 32));
}

// FreqFilter.scala:31-32: y = if (x.hashCode < rcx.hashCode) x else rcx  (signed, tie -> rcx)
template <int W> GK_HD Kmer<W> canonical(Kmer<W> x, int k) {
    Kmer<W> rc = revcomp(x, k);
    return ref_hash(x) < ref_hash(rc) ? x : rc;
}

// x.drop(1) :+ b  (Graph.scala:279)
GK_HD Kmer<1> append_base(Kmer<1> x, int b, int k) {
    Kmer<1> r;
    r.lo = (x.lo >> 2) | ((u64)b << (2 * (k - 1)));
    return r;
}
GK_HD Kmer<2> append_base(Kmer<2> x, int b, int k) {
    Kmer<2> r;
    r.lo = (x.lo >> 2) | (x.hi << 62);
    r.hi = (x.hi >> 2) | ((u64)b << (2 * (k - 33)));
    return r;
}
// b +: x.take(k-1)  (Graph.scala:273)
GK_HD Kmer<1> prepend_base(int b, Kmer<1> x, int k) {
    Kmer<1> r;
    r.lo = ((x.lo << 2) | (u64)b) & ((1ULL << (2 * k)) - 1);
    return r;
}
GK_HD Kmer<2> prepend_base(int b, Kmer<2> x, int k) {
    Kmer<2> r;
    r.hi = ((x.hi << 2) | (x.lo >> 62)) & low_mask(2 * (k - 32));
    r.lo = (x.lo << 2) | (u64)b;
    return r;
}
GK_HD int first_base(Kmer<1> x) { return (int)(x.lo & 3); }
GK_HD int first_base(Kmer<2> x) { return (int)(x.lo & 3); }

// ------------------------------------------------------------------------------------------
// Slot placement hash (free choice: slot order is unobservable — the reference's
// improve(hashCode)&mask, ArrayDNAMap.scala:130,267-272, only fixes ITS slot order).
// ------------------------------------------------------------------------------------------
GK_HD u64 mix64(u64 x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}
GK_HD u64 slot_mix(u64 x) { return mix64(x); }        // (a one-multiply mixer here changes nothing measurable: profiles/r03/ab_slot_mixer_one_multiply.txt)
GK_HD u64 slot_hash(Kmer<1> x) { return slot_mix(x.lo); }
GK_HD u64 slot_hash(Kmer<2> x) { return slot_mix(x.lo ^ (slot_mix(x.hi) + 0x9e3779b97f4a7c15ULL)); }

// ------------------------------------------------------------------------------------------
// Owner partition = strand-symmetric minimizer: the minimum, over all m-mers of
// the k-mer, of a hash of the m-mer's own canonical form min(w, rc(w)).  The set of canonical
// m-mers of x and rc(x) is the same, so x, rc(x) — hence both hash-rule candidates — share an
// owner.  m = min(11, k).
// ------------------------------------------------------------------------------------------
GK_HD u32 hash32(u32 x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
GK_HD u64 window_bits(Kmer<1> x, int bit) { return x.lo >> bit; }
GK_HD u64 window_bits(Kmer<2> x, int bit) {
    if (bit >= 64) return x.hi >> (bit - 64);
    if (bit == 0) return x.lo;
    return (x.lo >> bit) | (x.hi << (64 - bit));
}
template <int W> GK_HD u32 minimizer_score(Kmer<W> x, int k) {
    const int m = k < 11 ? k : 11;
    const u32 mm = (u32)((1ULL << (2 * m)) - 1);
    Kmer<W> rc = revcomp(x, k);
    u32 best = 0xffffffffu;
    for (int i = 0; i + m <= k; i++) {
        u32 a = (u32)window_bits(x, 2 * i) & mm;
        u32 b = (u32)window_bits(rc, 2 * (k - m - i)) & mm;   // rc of window i of x
        u32 s = hash32(a < b ? a : b);
        best = s < best ? s : best;
    }
    return best;
}
template <int W> GK_HD int owner_of(Kmer<W> x, int k, int P) {
    // the minimum of many hashes is skewed towards 0: re-mix it (a bijection) before scaling to P
    return (int)(((u64)hash32(minimizer_score(x, k) ^ 0x5bd1e995u) * (u64)P) >> 32);
}

#if defined(__HIPCC__)
// ------------------------------------------------------------------------------------------
// HBM-resident open-addressed table, array-of-slots so that one probe touches one 64-B sector:
// key word(s), the int32 count and the graph-phase annotation word live side by side.
// ------------------------------------------------------------------------------------------
static constexpr u64 KEY_EMPTY = ~0ULL;       // never a valid key word (top bit clear in all valid words)
static constexpr u64 KEY_TOMB = ~0ULL - 1;    // deleteAll tombstone (ArrayDNAMap.scala:168): probes run through it

template <int W> struct Slot;
// `extra` = count - 1: the lane whose CAS claims a slot has thereby recorded the first occurrence
// (v0 = 1, ArrayDNAMap.scala:146), so a new key costs ONE atomic (the CAS) and only repeats pay an
// atomic add.  Readers add the 1 back (slot_count).
template <> struct __attribute__((aligned(16))) Slot<1> { u64 w0; u32 extra; u32 aux; };
// 24 bytes, nothing dead: the reference stores 16 bytes of key and a 4-byte value per slot (ArrayDNAMap.scala:74-89); `aux` is
// the graph phase's annotation.  (Until round 3 the slot carried 8 bytes of padding to make it 32: a quarter of every table
// pass and of C4/C5's HBM budget.)  A segment of 1024 slots is 24 KiB and still a whole number of 16-byte vectors.
template <> struct __attribute__((aligned(8))) Slot<2> { u64 w0; u64 w1; u32 extra; u32 aux; };
static_assert(sizeof(Slot<1>) == 16 && sizeof(Slot<2>) == 24, "slot layout");

// Stored form.  W=1: the key itself (k<=31 leaves the two top bits clear).  W=2: two 63-bit
// halves, bits 0..62 and 63..125 of the 128-bit k-mer, so each word has a spare top bit and each
// can be claimed by its own 64-bit CAS (there is no 128-bit CAS).  k=64 has two bits more than
// that (its last base, bits 126..127): the TAG.  A tagged table keeps them in the slot ADDRESS:
// a key may only sit in slots whose index is = tag (mod 4), probing steps by 4, so the stored 126
// bits plus the slot index give the key back.
template <int W> struct Stored;
template <> struct Stored<1> { u64 w0; };
template <> struct Stored<2> { u64 w0, w1; };
GK_HD Stored<1> to_stored(Kmer<1> x) { return Stored<1>{x.lo}; }
GK_HD Stored<2> to_stored(Kmer<2> x) { return Stored<2>{x.lo & 0x7fffffffffffffffULL, (x.lo >> 63) | ((x.hi << 1) & 0x7fffffffffffffffULL)}; }
GK_HD u32 key_tag(Kmer<1>) { return 0u; }
GK_HD u32 key_tag(Kmer<2> x) { return (u32)(x.hi >> 62); }
GK_HD Kmer<1> from_stored(Stored<1> s, u32 = 0u) { return Kmer<1>{s.w0}; }
GK_HD Kmer<2> from_stored(Stored<2> s, u32 tag = 0u) { return Kmer<2>{s.w0 | (s.w1 << 63), (s.w1 >> 1) | ((u64)tag << 62)}; }
GK_D Stored<1> load_stored(const Slot<1> *s) { return Stored<1>{s->w0}; }
GK_D Stored<2> load_stored(const Slot<2> *s) { return Stored<2>{s->w0, s->w1}; }

// An array of EMPTY slots seen as 16-byte vectors (how the segment kernels clear and copy segments): vector i.
// W = 1: one slot per vector {w0 = ~0, extra = 0, aux = 0}.  W = 2: three 64-bit words per slot {~0, ~0, 0}, two per vector.
template <int W> GK_D uint4 empty_vec(u32 i) {
    if constexpr (W == 1) { (void)i; return make_uint4(~0u, ~0u, 0u, 0u); }
    else {
        const u32 a = (2u * i) % 3u;                       // which word of its slot the vector's first 64-bit word is
        const u32 lo = a == 2u ? 0u : ~0u, hi = a == 1u ? 0u : ~0u;      // (second word: (a + 1) % 3 == 2  <=>  a == 1)
        return make_uint4(lo, lo, hi, hi);
    }
}
template <int W> GK_D uint4 empty_vec_of(const Slot<W> *, u32 i) { return empty_vec<W>(i); }
template <int W> GK_D void slot_tomb(Slot<W> *s) { s->w0 = ~0ULL - 1; }
// number of slots whose FIRST key word lies in vector i of a segment and is EMPTY (a slot is free iff its w0 is EMPTY)
template <int W> GK_D u32 empty_w0_in_vec(u32 i, uint4 v) {
    if constexpr (W == 1) { (void)i; return (v.x & v.y) == ~0u ? 1u : 0u; }
    else {
        const u32 a = (2u * i) % 3u;
        if (a == 0u) return (v.x & v.y) == ~0u ? 1u : 0u;
        if (a == 2u) return (v.z & v.w) == ~0u ? 1u : 0u;
        return 0u;
    }
}

// ------------------------------------------------------------------------------------------
// CSlot — the 12-byte slot of a COUNT table with 8-byte keys (k <= 31): what the reference stores per entry, an 8-byte key and a
// 4-byte value (ArrayDNAMap.scala:74-89), and nothing else.  The 62-bit key is kept as two 31-bit halves, each with a spare top
// bit and each claimed by its own 32-bit CAS — the two-word protocol of Slot<2> (claim w0, then w1; whoever sets w1 owns the
// slot) at half the width: a 12-byte slot is only 4-byte aligned, which rules out a 64-bit CAS on its key.  No `aux`: the graph
// phase's annotation exists only in the 16-byte Slot<1> tables that deleteAll / the gather / gk_map_create_for_graph build
// (gk_map::graph_layout).  A segment of 2048 slots is 24 KiB instead of 32: a quarter less to stream per table pass.
// ------------------------------------------------------------------------------------------
struct CSlot { u32 w0, w1, extra; };
static_assert(sizeof(CSlot) == 12, "count slot layout");
static constexpr u32 KEY_EMPTY32 = ~0u, KEY_TOMB32 = ~0u - 1u;
GK_HD u32 c_w0(Kmer<1> x) { return (u32)x.lo & 0x7fffffffu; }
GK_HD u32 c_w1(Kmer<1> x) { return (u32)(x.lo >> 31); }                   // (a k <= 31 k-mer is below 2^62: this is below 2^31)
GK_D bool slot_live(const CSlot *s) { return s->w0 != KEY_EMPTY32 && s->w0 != KEY_TOMB32; }
GK_D u32 slot_count(const CSlot *s) { return s->extra + 1u; }
GK_D Kmer<1> slot_key(const CSlot *slots, u64 i, u32 = 0u) { return Kmer<1>{(u64)slots[i].w0 | ((u64)slots[i].w1 << 31)}; }
GK_D void slot_tomb(CSlot *s) { s->w0 = KEY_TOMB32; }
// an array of EMPTY count slots as 16-byte vectors: 32-bit words {~0, ~0, 0} per slot, four per vector
GK_D uint4 empty_vec_of(const CSlot *, u32 i) {
    const u32 a = (4u * i) % 3u;                            // which word of its slot the vector's first 32-bit word is
    return make_uint4(a == 2u ? 0u : ~0u, a == 1u ? 0u : ~0u, a == 0u ? 0u : ~0u, a == 2u ? 0u : ~0u);
}
GK_D u32 empty_w0_in_vec_of(const CSlot *, u32 i, uint4 v) {
    const u32 a = (4u * i) % 3u;                            // word j of the vector is a w0 iff (a + j) % 3 == 0
    u32 n = 0;
    if (a == 0u) n = (v.x == ~0u) + (v.w == ~0u);
    else if (a == 1u) n = (v.z == ~0u);
    else n = (v.y == ~0u);
    return n;
}

template <int W> GK_D u32 empty_w0_in_vec_of(const Slot<W> *, u32 i, uint4 v) { return empty_w0_in_vec<W>(i, v); }

// The table is an array of SEGMENTS of 2^seg_bits slots (2048 16-B slots = 32 KiB, or 1024 24-B slots = 24 KiB), and linear probing wraps INSIDE a segment.  A segment is the unit one workgroup can hold
// in LDS, which is what lets a batch be radix-partitioned by segment and built there with LDS
// atomics and coalesced HBM traffic (gk_partition.hip) instead of one global atomic per key.
// Segment id = (L1 bucket, fine bucket): L1 = top lnb1 bits of the slot hash (<= 256 buckets; 512 or 1024 only
// for tables that would otherwise need more than MAX_NB2 fine buckets per L1 bucket, i.e. beyond 34 GB),
// fine = 32 middle bits scaled to nb2 — so the number of segments need not be a power of two and
// the table can be sized to the load factor wanted.  Start position = low seg_bits bits.
// The 32 fine bits are bits 24..55, directly below the L1 bits of a 256-bucket table; with 9 or 10 L1 bits they move down
// with them (bits 23..54, 22..53) — shared bits would pin the top of the fine value inside an L1 bucket and leave half or
// three quarters of its fine buckets empty.  They stay clear of the start position (bits 0..10) and of the sample bits (11..20).
template <int W> struct SegBits;
#ifndef GK_SEG_BITS1
#define GK_SEG_BITS1 11
#endif
template <> struct SegBits<1> { static constexpr u32 value = GK_SEG_BITS1; };
template <> struct SegBits<2> { static constexpr u32 value = GK_SEG_BITS1 - 1; };

static constexpr u32 MAX_LNB1 = 10;      // up to 1024 L1 buckets
template <int W, class S = Slot<W>> struct Table {
    S *slots;
    u32 nb2;           // fine buckets per L1 bucket
    u32 lnb1;          // log2(L1 buckets), 0..MAX_LNB1
    u32 tagged;        // 1 for k = 64: slot index mod 4 carries the key's last base
    u32 both;          // 1: the table may hold ANY orientation of a k-mer (keys inserted verbatim through the ABI), not only
                       // the hash-rule one: strand-agnostic lookups (table_find_either) must probe both, always
    GK_HD u64 nseg() const { return (u64)nb2 << lnb1; }
    GK_HD u64 capacity() const { return nseg() << SegBits<W>::value; }
};
template <int W, class S> GK_HD u32 seg_l1(const Table<W, S> &t, u64 h) { return t.lnb1 ? (u32)(h >> (64 - t.lnb1)) : 0u; }
template <int W, class S> GK_HD u32 seg_fine(const Table<W, S> &t, u64 h) {
    const u32 shift = t.lnb1 > 8 ? 32u - t.lnb1 : 24u;
    return (u32)((((h >> shift) & 0xffffffffULL) * (u64)t.nb2) >> 32);
}
template <int W, class S> GK_HD u32 seg_of(const Table<W, S> &t, u64 h) { return seg_l1(t, h) * t.nb2 + seg_fine(t, h); }
template <int W> GK_HD u32 seg_pos(u64 h) { return (u32)h & ((1u << SegBits<W>::value) - 1u); }
// where a key's probe starts in its segment
template <int W, class S> GK_HD u32 home_pos(const Table<W, S> &, u64 h) { return seg_pos<W>(h); }

struct Counters {      // device-resident, one per map
    unsigned long long size;        // live keys
    unsigned long long occurrences; // windows counted by the last count kernel
    u32 error;                      // 1 = a probe ran a whole segment (capacity exhausted)
    u32 format;                     // 1 = a device record's length byte exceeded the declared read length (clamped)
    unsigned long long sample_claims;   // keys the distinct-key sample has admitted since the last clear
    u32 noncanon;                   // 1 = a key inserted VERBATIM was not the hash-rule orientation of its k-mer (sticky until clear)
    u32 pad;
    unsigned long long rebuild_kept;    // k_compact_seg: keys moved into the new table
    unsigned long long rebuild_sample;  // k_count_ge_sample: survivors seen in the sampled segments
};

GK_D u64 cas64(u64 *p, u64 expect, u64 val) {
    __hip_atomic_compare_exchange_strong(p, &expect, val, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return expect;   // old value
}
GK_D void add32_noret(u32 *p, u32 v) { (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Container.update(key, v0, f) with v0 = add, f = _ + add (ArrayDNAMap.scala:129-150) on one
// segment (`seg` points at its first slot; in HBM here, in LDS in gk_partition.hip), lock-free:
// plain loads are only hints (a stale line can only read EMPTY, never a wrong key, because key
// words are write-once during an insert phase); the CAS decides.  Returns 1 if this call claimed
// a new slot, 0 if the key was there, -1 if the probe wrapped the whole segment (full).
template <class CAS, class ADD>
GK_D int seg_add(Slot<1> *seg, u32 pos, Kmer<1> key, u32 add, CAS cas, ADD addf, u32 = 0u) {
    constexpr u32 smask = (1u << SegBits<1>::value) - 1u;
    u32 i = pos;
    for (u32 n = 0; n <= smask; ++n) {
        Slot<1> *s = &seg[i];
        u64 cur = s->w0;
        int claimed = 0;
        if (cur == KEY_EMPTY) {
            cur = cas(&s->w0, KEY_EMPTY, key.lo);
            if (cur == KEY_EMPTY) { cur = key.lo; claimed = 1; }
        }
        if (cur == key.lo) {
            const u32 a = add - (u32)claimed;
            if (a) addf(&s->extra, a);
            return claimed;
        }
        i = (i + 1) & smask;
    }
    return -1;
}
// 128-bit keys: claim w0 then w1, each write-once.  Two keys sharing w0 may race for w1; the
// loser simply moves on to the next slot, and every later probe of that key makes the same
// decision from the (now immutable) slot contents, so a key never lands in two slots.
template <class CAS, class ADD>
GK_D int seg_add(Slot<2> *seg, u32 pos, Kmer<2> key, u32 add, CAS cas, ADD addf, u32 tagged = 0u) {
    constexpr u32 smask = (1u << SegBits<2>::value) - 1u;
    const Stored<2> k = to_stored(key);
    const u32 step = tagged ? 4u : 1u;
    u32 i = tagged ? ((pos & ~3u) | key_tag(key)) : pos;
    for (u32 n = 0; n <= smask; n += step) {
        Slot<2> *s = &seg[i];
        u64 c0 = s->w0;
        int claimed = 0;
        if (c0 == KEY_EMPTY) {
            c0 = cas(&s->w0, KEY_EMPTY, k.w0);
            if (c0 == KEY_EMPTY) c0 = k.w0;
        }
        if (c0 == k.w0) {
            u64 c1 = s->w1;
            if (c1 == KEY_EMPTY) {
                c1 = cas(&s->w1, KEY_EMPTY, k.w1);
                if (c1 == KEY_EMPTY) { c1 = k.w1; claimed = 1; }
            }
            if (c1 == k.w1) {
                const u32 a = add - (u32)claimed;
                if (a) addf(&s->extra, a);
                return claimed;
            }
        }
        i = (i + step) & smask;
    }
    return -1;
}
// Claim a slot for a key KNOWN to be absent and inserted by nobody else (rehash: the old table holds every key
// once): the index inside the segment, or -1 if the segment is full.  The caller then writes the value with a
// plain store — no second atomic.
GK_D i64 seg_claim_unique(Slot<1> *seg, u32 pos, Kmer<1> key, u32 = 0u) {
    constexpr u32 smask = (1u << SegBits<1>::value) - 1u;
    u32 i = pos;
    for (u32 n = 0; n <= smask; ++n) {
        if (seg[i].w0 == KEY_EMPTY && cas64(&seg[i].w0, KEY_EMPTY, key.lo) == KEY_EMPTY) return (i64)i;
        i = (i + 1) & smask;
    }
    return -1;
}
GK_D i64 seg_claim_unique(Slot<2> *seg, u32 pos, Kmer<2> key, u32 tagged = 0u) {
    constexpr u32 smask = (1u << SegBits<2>::value) - 1u;
    const Stored<2> k = to_stored(key);
    const u32 step = tagged ? 4u : 1u;
    u32 i = tagged ? ((pos & ~3u) | key_tag(key)) : pos;
    for (u32 n = 0; n <= smask; n += step) {
        // w0 decides the slot (the key is unique, so nobody races for w1 of a slot whose w0 is ours... except a key
        // sharing w0: it sees w0 == its own w0 and goes for w1 too — so w1 is claimed with a CAS as in seg_add)
        u64 c0 = seg[i].w0;
        if (c0 == KEY_EMPTY) { c0 = cas64(&seg[i].w0, KEY_EMPTY, k.w0); if (c0 == KEY_EMPTY) c0 = k.w0; }
        if (c0 == k.w0 && seg[i].w1 == KEY_EMPTY && cas64(&seg[i].w1, KEY_EMPTY, k.w1) == KEY_EMPTY) return (i64)i;
        i = (i + step) & smask;
    }
    return -1;
}
GK_D u32 cas32(u32 *p, u32 expect, u32 val) {
    __hip_atomic_compare_exchange_strong(p, &expect, val, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return expect;   // old value
}
struct GlobalCas {
    GK_D u64 operator()(u64 *p, u64 e, u64 v) const { return cas64(p, e, v); }
    GK_D u32 operator()(u32 *p, u32 e, u32 v) const { return cas32(p, e, v); }
};
// the two-word protocol of seg_add(Slot<2>) on 31-bit halves
template <class CAS, class ADD>
GK_D int seg_add(CSlot *seg, u32 pos, Kmer<1> key, u32 add, CAS cas, ADD addf, u32 = 0u) {
    constexpr u32 smask = (1u << SegBits<1>::value) - 1u;
    const u32 k0 = c_w0(key), k1 = c_w1(key);
    u32 i = pos;
    for (u32 n = 0; n <= smask; ++n) {
        CSlot *s = &seg[i];
        u32 c0 = s->w0;
        int claimed = 0;
        if (c0 == KEY_EMPTY32) {
            c0 = cas(&s->w0, KEY_EMPTY32, k0);
            if (c0 == KEY_EMPTY32) c0 = k0;
        }
        if (c0 == k0) {
            u32 c1 = s->w1;
            if (c1 == KEY_EMPTY32) {
                c1 = cas(&s->w1, KEY_EMPTY32, k1);
                if (c1 == KEY_EMPTY32) { c1 = k1; claimed = 1; }
            }
            if (c1 == k1) {
                const u32 a = add - (u32)claimed;
                if (a) addf(&s->extra, a);
                return claimed;
            }
        }
        i = (i + 1) & smask;
    }
    return -1;
}
GK_D i64 seg_claim_unique(CSlot *seg, u32 pos, Kmer<1> key, u32 = 0u) {
    constexpr u32 smask = (1u << SegBits<1>::value) - 1u;
    const u32 k0 = c_w0(key), k1 = c_w1(key);
    u32 i = pos;
    for (u32 n = 0; n <= smask; ++n) {
        u32 c0 = seg[i].w0;
        if (c0 == KEY_EMPTY32) { c0 = cas32(&seg[i].w0, KEY_EMPTY32, k0); if (c0 == KEY_EMPTY32) c0 = k0; }
        if (c0 == k0 && seg[i].w1 == KEY_EMPTY32 && cas32(&seg[i].w1, KEY_EMPTY32, k1) == KEY_EMPTY32) return (i64)i;
        i = (i + 1) & smask;
    }
    return -1;
}
struct GlobalAdd { GK_D void operator()(u32 *p, u32 v) const { add32_noret(p, v); } };

// the HBM form: one global CAS per new key, one no-return add per repeat
template <int W, class S> GK_D int table_add(const Table<W, S> &t, Kmer<W> key, u32 add, u32 *err) {
    const u64 h = slot_hash(key);
    S *seg = t.slots + ((u64)seg_of(t, h) << SegBits<W>::value);
    int r = seg_add(seg, home_pos(t, h), key, add, GlobalCas(), GlobalAdd(), t.tagged);
    if (r < 0) { *err = 1; return 0; }
    return r;
}

// Container.apply (ArrayDNAMap.scala:90-101) on a quiescent table: global slot index or -1.
GK_D i64 seg_find(const Slot<1> *seg, u32 pos, Kmer<1> key, u32 = 0u) {
    constexpr u32 smask = (1u << SegBits<1>::value) - 1u;
    u32 i = pos;
    for (u32 n = 0; n <= smask; ++n) {
        u64 cur = seg[i].w0;
        if (cur == key.lo) return (i64)i;
        if (cur == KEY_EMPTY) return -1;
        i = (i + 1) & smask;
    }
    return -1;
}
GK_D i64 seg_find(const Slot<2> *seg, u32 pos, Kmer<2> key, u32 tagged = 0u) {
    constexpr u32 smask = (1u << SegBits<2>::value) - 1u;
    const Stored<2> k = to_stored(key);
    const u32 step = tagged ? 4u : 1u;
    u32 i = tagged ? ((pos & ~3u) | key_tag(key)) : pos;
    for (u32 n = 0; n <= smask; n += step) {
        u64 c0 = seg[i].w0;
        if (c0 == k.w0 && seg[i].w1 == k.w1) return (i64)i;
        if (c0 == KEY_EMPTY) return -1;
        i = (i + step) & smask;
    }
    return -1;
}
GK_D i64 seg_find(const CSlot *seg, u32 pos, Kmer<1> key, u32 = 0u) {
    constexpr u32 smask = (1u << SegBits<1>::value) - 1u;
    const u32 k0 = c_w0(key), k1 = c_w1(key);
    u32 i = pos;
    for (u32 n = 0; n <= smask; ++n) {
        const u32 c0 = seg[i].w0;
        if (c0 == k0 && seg[i].w1 == k1) return (i64)i;
        if (c0 == KEY_EMPTY32) return -1;
        i = (i + 1) & smask;
    }
    return -1;
}
template <int W, class S> GK_D i64 table_find(const Table<W, S> &t, Kmer<W> key) {
    const u64 h = slot_hash(key);
    const u64 base = (u64)seg_of(t, h) << SegBits<W>::value;
    i64 r = seg_find(t.slots + base, home_pos(t, h), key, t.tagged);
    return r < 0 ? -1 : (i64)base + r;
}
template <int W> GK_D bool slot_live(const Slot<W> *s) { return s->w0 != KEY_EMPTY && s->w0 != KEY_TOMB; }
template <int W> GK_D u32 slot_count(const Slot<W> *s) { return s->extra + 1u; }
// the key held by slot i of a table (i = global slot index; segments are multiples of 4 slots)
// ... with the tag (a k = 64 key's last base) given explicitly: for a RANGE of slots whose first index the caller knows
template <int W> GK_D Kmer<W> slot_key_tag(const Slot<W> *slots, u64 i, u32 tag) { return from_stored(load_stored(&slots[i]), tag); }
GK_D Kmer<1> slot_key_tag(const CSlot *slots, u64 i, u32) { return slot_key(slots, i); }
template <int W> GK_D Kmer<W> slot_key(const Slot<W> *slots, u64 i, u32 tagged) {
    return from_stored(load_stored(&slots[i]), tagged ? (u32)(i & 3u) : 0u);
}

// ------------------------------------------------------------------------------------------
// MINIMIZER-BUCKETED table (the graph phase's own copy of the k-mer set, gk_graph.hip): bucket = owner_of(key, k, nb), i.e. a
// hash of the k-mer's strand-symmetric minimizer, so that consecutive k-mers of the genome — a k-mer and most of its 8
// de Bruijn neighbours, a unitig's next step — sit in ONE small region (a power of two of slots, load <= 0.5) instead of in
// eight random 128-byte lines of a hashed table: the classify pass and the unitig walks then find their probes in L1 / L2.
// Probing is linear inside the region from the key's slot hash.  Read-only after it is built; k = 64 keeps the hashed table.
// ------------------------------------------------------------------------------------------
template <int W> struct MbTable {
    Slot<W> *slots;          // the regions, back to back
    const u64 *off;          // [nb + 1]: first slot of every bucket's region
    u32 nb;
    u32 both;                // as Table::both
    u64 nslots;
    static constexpr u32 tagged = 0;
    GK_HD u64 capacity() const { return nslots; }
};
template <int W> GK_HD u32 mb_bucket(const MbTable<W> &t, Kmer<W> x, int k) { return (u32)owner_of(x, k, (int)t.nb); }
// where a key's probe starts, and the region it stays in: {first slot of the region, size - 1, start position}
template <int W> struct ProbeAt { const Slot<W> *reg; u64 base; u32 mask, pos, step; };
template <int W> GK_D ProbeAt<W> probe_at(const Table<W> &t, Kmer<W> q, int) {
    const u64 h = slot_hash(q);
    const u64 base = (u64)seg_of(t, h) << SegBits<W>::value;
    const u32 pos = t.tagged ? ((seg_pos<W>(h) & ~3u) | key_tag(q)) : home_pos(t, h);
    return ProbeAt<W>{t.slots + base, base, (1u << SegBits<W>::value) - 1u, pos, t.tagged ? 4u : 1u};
}
template <int W> GK_D ProbeAt<W> probe_at_bucket(const MbTable<W> &t, Kmer<W> q, u32 b) {
    const u64 base = t.off[b];
    const u32 mask = (u32)(t.off[b + 1] - base) - 1u;
    return ProbeAt<W>{t.slots + base, base, mask, (u32)slot_hash(q) & mask, 1u};
}
template <int W> GK_D ProbeAt<W> probe_at(const MbTable<W> &t, Kmer<W> q, int k) { return probe_at_bucket(t, q, mb_bucket(t, q, k)); }
// linear probe from p.pos (the caller may have looked at that slot already: `skip_first`); global slot index or -1
template <int W> GK_D i64 probe_find(const ProbeAt<W> &p, Kmer<W> key, bool skip_first = false) {
    const Stored<W> k = to_stored(key);
    u32 i = skip_first ? (p.pos + p.step) & p.mask : p.pos;
    for (u32 n = skip_first ? p.step : 0u; n <= p.mask; n += p.step) {
        const u64 c0 = p.reg[i].w0;
        if constexpr (W == 1) { if (c0 == k.w0) return (i64)(p.base + i); }
        else { if (c0 == k.w0 && p.reg[i].w1 == k.w1) return (i64)(p.base + i); }
        if (c0 == KEY_EMPTY) return -1;
        i = (i + p.step) & p.mask;
    }
    return -1;
}
template <int W> GK_D i64 table_find(const MbTable<W> &t, Kmer<W> key, int k) { return probe_find(probe_at(t, key, k), key); }
template <int W> GK_D i64 table_find(const Table<W> &t, Kmer<W> key, int) { return table_find(t, key); }
template <int W> GK_D Kmer<W> slot_key(const MbTable<W> &t, u64 i) { return from_stored(load_stored(&t.slots[i])); }
template <int W> GK_D Kmer<W> slot_key(const Table<W> &t, u64 i) { return slot_key(t.slots, i, t.tagged); }
// `contains` on either strand (see table_find_either below): x and rc(x) share their bucket, so ONE minimizer serves both probes
template <int W> GK_D i64 table_find_either(const MbTable<W> &t, Kmer<W> x, int k, bool *fwd) {
    Kmer<W> rc = revcomp(x, k);
    const i32 hx = ref_hash(x), hr = ref_hash(rc);
    const u32 b = mb_bucket(t, x, k);
    if (!t.both) {
        if (hx < hr) { *fwd = true; return probe_find(probe_at_bucket(t, x, b), x); }
        if (hx > hr) { *fwd = false; return probe_find(probe_at_bucket(t, rc, b), rc); }
    }
    const bool x_first = !kmer_less(rc, x);
    const i64 s = probe_find(probe_at_bucket(t, x_first ? x : rc, b), x_first ? x : rc);
    if (s >= 0) { *fwd = x_first; return s; }
    *fwd = !x_first;
    return probe_find(probe_at_bucket(t, x_first ? rc : x, b), x_first ? rc : x);
}

// Graph.buildGraph `contains` (Graph.scala:270): either strand.  Only the hash-rule canonical
// orientation can be a stored key, except in the tie h(x) == h(rc x) where occurrences seen as x
// are filed under rc x and vice versa (FreqFilter.scala:31-32) — then both are probed, the
// numerically smaller orientation first so that every caller resolves a tie pair to the same slot.
// Returns the slot (or -1) and whether the stored key is x itself (fwd) or its reverse complement.
// A table that took verbatim keys (Table::both) can hold either orientation of ANY k-mer — the reference's `contains`
// probes both unconditionally — so there the two-probe form is the rule, with the same order.
template <int W> GK_D i64 table_find_either(const Table<W> &t, Kmer<W> x, int k, bool *fwd) {
    Kmer<W> rc = revcomp(x, k);
    i32 hx = ref_hash(x), hr = ref_hash(rc);
    if (!t.both) {
        if (hx < hr) { *fwd = true; return table_find(t, x); }
        if (hx > hr) { *fwd = false; return table_find(t, rc); }
    }
    bool x_first = !kmer_less(rc, x);
    i64 s = table_find(t, x_first ? x : rc);
    if (s >= 0) { *fwd = x_first; return s; }
    *fwd = !x_first;
    return table_find(t, x_first ? rc : x);
}
#endif  // __HIPCC__

}  // namespace gk
