// gk_skm.hip — owner routing by SUPER-K-MERS for the PartitionedDNAMap path.
//
// Reference: PartitionedDNAMap.partition (S/ds/PartitionedDNAMap.scala:60-63) sends one Akka/Kryo
// message per k-mer to `hashCode mod P`.  Here the owner of a k-mer is a strand-symmetric minimizer
// hash (gk::owner_of), consecutive k-mers of a read almost always share their
// minimizer, and a maximal run of same-owner k-mers is shipped as ONE record: the run's bases, 2
// bits each, in the reference's own `.bin` framing [len:u8][bases] inside a fixed 16-B (k<=31) or
// 32-B (k>=34) slot.  A run of r k-mers costs (r+k-1)/4+1 bytes instead of 8r..16r; the receiver
// feeds the records to the ordinary read-counting pipeline (they ARE short reads: every k-mer of
// a record is one of the sender's windows, no window is sent twice, none is lost), so the
// all-to-all over xGMI moves ~8x fewer bytes and the receiver re-extracts and canonicalises.
//
// One kernel (k_skm_route) turns reads into records; owner p's records land in region p of the
// output buffer (out_cap_records / P slots each), counted per owner.
//
// The minimizer of a window is the minimum of the per-POSITION m-mer scores over the window: the
// scores are computed once per position (one lane each), parked in LDS, and each window lane
// takes the min of its k-m+1 neighbours from LDS — 1 hash per k-mer instead of k-m+1.
#include <algorithm>
#include <string>

#include "gk_internal.h"
#include "gk_tile.h"

using namespace gk;

GK_TIMERS_DEFINE(skm)

static constexpr int MAX_PARTS = 64;
static constexpr int MAX_POS = 256;       // positions per read (len <= 255)

__host__ __device__ static inline int skm_slot_bytes(int k) { return k <= 31 ? 16 : 32; }
__host__ __device__ static inline int skm_max_bases(int k) { return (skm_slot_bytes(k) - 1) * 4; }
__host__ __device__ static inline int skm_max_run(int k) { return skm_max_bases(k) - k + 1; }

// score of the m-mer at bit position `bit` of the LDS tile: hash of its own canonical form
__device__ __forceinline__ u32 mmer_score(const u32 *tile, u32 bit, int m) {
    const u32 wi = bit >> 5, sh = bit & 31;
    const u64 v = (u64)tile[wi] | ((u64)tile[wi + 1] << 32);
    const u32 mm = (u32)((1ull << (2 * m)) - 1);
    const u32 a = (u32)(v >> sh) & mm;
    // reverse complement of a 2m-bit value: complement, reverse 2-bit groups of the 32-bit word, shift down
    u32 r = __brev(~a);
    r = ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1);
    r >>= (32 - 2 * m);
    return hash32(a < r ? a : r);
}

// `nbases` bases starting at bit `bit` of the tile, as record bytes: byte 0 = nbases, then the
// bases 2 bits each LSB first (the `.bin` framing), zero padded to the slot.  16- or 32-byte slot.
template <int SLOT>
__device__ __forceinline__ void write_record(const u32 *tile, u32 bit, int nbases, uint8_t *dst) {
    constexpr int NW = SLOT / 4;                  // 32-bit words of the slot
    const u32 wi = bit >> 5, sh = bit & 31;
    u32 out[NW];
    // payload word j (bits 32j..32j+31 of the base stream)
    u32 prev = tile[wi];
    u32 pay[NW];
#pragma unroll
    for (int j = 0; j < NW; j++) {
        const u32 next = tile[wi + j + 1];
        pay[j] = sh ? (prev >> sh) | (next << (32 - sh)) : prev;
        prev = next;
    }
    // mask the payload to 2*nbases bits
    const int nbits = 2 * nbases;
#pragma unroll
    for (int j = 0; j < NW; j++) {
        const int lo = 32 * j;
        if (nbits <= lo) pay[j] = 0;
        else if (nbits < lo + 32) pay[j] &= (1u << (nbits - lo)) - 1u;
    }
    // shift the whole payload left by 8 bits and put the length byte in front
    out[0] = (pay[0] << 8) | (u32)nbases;
#pragma unroll
    for (int j = 1; j < NW; j++) out[j] = (pay[j] << 8) | (pay[j - 1] >> 24);
    uint4 *d4 = reinterpret_cast<uint4 *>(dst);
    d4[0] = make_uint4(out[0], out[1], out[2], out[3]);
    if constexpr (SLOT == 32) d4[1] = make_uint4(out[4], out[5], out[6], out[7]);
}

// One pass: owners -> runs -> records.  Per tile of 64 reads:
//   phase 1  each wave takes reads round-robin; per read the m-mer scores of its positions go to the
//            wave's LDS strip, each window lane takes the minimum over its k-m+1 positions -> owner;
//            run starts are found with a ballot (owner differs from the previous lane's; lane 0 of
//            every 64-window block starts a record too, so that everything stays wave-local), run
//            lengths with bit scans of the ballot — no loops.  Every run start appends ONE 32-bit
//            descriptor (read, first window, run length, owner) to an LDS list and counts its
//            records in an LDS histogram per owner.
//   reserve  one global atomic per (tile, owner) hands out a contiguous run of slots in the owner's
//            region; a region that would overflow sets the overflow flag (the caller retries larger)
//   phase 2  one LANE per descriptor (not one per window): it takes its slots from the owner's LDS
//            rank counter and writes its records: a 120-bit (or 248-bit) slice of the read, shifted
//            behind a length byte, as one or two 16-B stores.  (Re-deriving the runs from per-window
//            owner bytes with the same ballots left 1 lane in 10 busy during the record writes:
//            42 % of the kernel's time by the in-kernel phase timers; this form: see DESIGN.md.)
// The descriptor list holds DESC_CAP runs; a tile whose reads fragment into more than that (owners
// alternating from window to window — not something hashed minimizers do on real reads) is redone in
// groups of 16 reads, which always fit (16 x 254 windows).
static constexpr int DESC_CAP = 4096;
template <int SLOT>
__global__ __launch_bounds__(BLOCK) void k_skm_route(const uint8_t *__restrict__ rec, u64 nreads, u32 stride, int k, int P,
                                                     WindowLimits lim, u64 region_cap /* records per owner region */, unsigned long long *cursors,
                                                     unsigned long long *kmer_counts, u32 *overflow, uint8_t *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) u32 tile[TILE_WORDS];
    __shared__ u32 strips[(BLOCK / 64) * MAX_POS], strips2[(BLOCK / 64) * MAX_POS];
    __shared__ u32 desc[DESC_CAP];            // r (6 bits) | first window (8) << 6 | run length (8) << 14 | owner (6) << 22
    __shared__ u32 hist[MAX_PARTS], t_kmer[MAX_PARTS], h_kmer[MAX_PARTS];
    __shared__ unsigned long long base[MAX_PARTS];
    __shared__ u32 s_ndesc, s_over;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = k < 11 ? k : 11, w = k - m + 1, rmax = skm_max_run(k);
    if (threadIdx.x < MAX_PARTS) h_kmer[threadIdx.x] = 0;
    const u64 ntiles = (nreads + TILE_READS - 1) / TILE_READS;
    GK_T0();
    for (u64 tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const u64 r0 = tl * TILE_READS;
        const int nr = (int)min((u64)TILE_READS, nreads - r0);
        __syncthreads();
        GK_TICK(4);
        const u64 a0 = stage_tile(tile, rec, r0 * stride, (r0 + nr) * stride);
        const uint8_t *tb = reinterpret_cast<const uint8_t *>(tile);
        u32 *strip = strips + wave * MAX_POS, *strip2 = strips2 + wave * MAX_POS;
        int gsz = TILE_READS, g0 = 0;
        while (g0 < nr) {
            const int g1 = min(g0 + gsz, nr);
            __syncthreads();                  // tile staged / previous group's descriptors consumed
            if (threadIdx.x < MAX_PARTS) { hist[threadIdx.x] = 0; t_kmer[threadIdx.x] = 0; }
            if (threadIdx.x == 0) { s_ndesc = 0; s_over = 0; }
            __syncthreads();
            GK_TICK(0);
            // ---- phase 1
            for (int r = g0 + wave; r < g1; r += BLOCK / 64) {
                const u32 ro = (u32)((r0 + r) * stride - a0);
                const int len = record_len(tb, ro, lim), nk = len - k + 1, nm = len - m + 1;
                const u32 bit0 = (ro + 1) * 8;
                __builtin_amdgcn_wave_barrier();
                GK_TICK(1);
                for (int q = lane; q < nm; q += 64) strip[q] = mmer_score(tile, bit0 + 2 * q, m);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                GK_TICK(5);
                // window minimum over w = k-m+1 scores.  A plain loop is w dependent LDS round trips per
                // block (the reads are not batched: 90 % of the kernel by the phase timers); instead the
                // minimum of every 8 consecutive scores is parked once per position, and a window takes the
                // min of ceil(w / 8) of those (the last one clamped to the window's end), all reads issued together.
                const u32 *mins = strip;
                if (w >= 8) {
                    u32 *m8 = strip2;
                    for (int q = lane; q < nm - 7; q += 64) {
                        u32 v = strip[q];
#pragma unroll
                        for (int j = 1; j < 8; j++) v = min(v, strip[q + j]);
                        m8[q] = v;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    mins = m8;
                }
                GK_TICK(6);
                const int step = w >= 8 ? 8 : 1, last = w >= 8 ? w - 8 : w - 1;
                const int nterms = w >= 8 ? (w + 7) / 8 : w;         // 3 for k = 31 (w = 21): offsets 0, 8, 13; 7 at most (k = 63, or w = 7)
                for (int pb = 0; pb < nk; pb += 64) {
                    const int p = pb + lane;
                    const bool valid = p < nk;
                    u32 best = 0xffffffffu;
                    if (valid) {
#pragma unroll
                        for (int j = 0; j < 7; j++)
                            if (j < nterms) best = min(best, mins[p + min(j * step, last)]);
                    }
                    const int o = (int)(((u64)hash32(best ^ 0x5bd1e995u) * (u64)P) >> 32);      // == gk::owner_of
                    const int prev = __shfl_up(o, 1);
                    const bool start = valid && (lane == 0 || o != prev);
                    const unsigned long long S = __ballot(start), V = __ballot(valid);
                    u32 at = 0;
                    if (lane == 0 && S) at = atomicAdd(&s_ndesc, (u32)__popcll(S));
                    at = __shfl(at, 0);
                    if (start) {
                        const unsigned long long nxt = lane == 63 ? 0ull : (S >> (lane + 1));
                        const int rl = nxt ? __ffsll((long long)nxt) : (__popcll(V) - lane);
                        const u32 idx = at + (u32)__popcll(S & ((1ull << lane) - 1ull));
                        if (idx < (u32)DESC_CAP) desc[idx] = (u32)(r - g0) | ((u32)p << 6) | ((u32)rl << 14) | ((u32)o << 22);
                        else s_over = 1;
                        atomicAdd(&hist[o], (u32)((rl + rmax - 1) / rmax));
                        atomicAdd(&t_kmer[o], (u32)rl);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
            GK_TICK(1);
            __syncthreads();
            if (s_over) { gsz = 16; continue; }       // uniform: redo this group in pieces that always fit
            // ---- reserve
            if (threadIdx.x < (u32)P) {
                const u32 c = hist[threadIdx.x];
                unsigned long long b = ~0ull;
                if (c) {
                    const unsigned long long at = atomicAdd(&cursors[threadIdx.x], (unsigned long long)c);
                    if (at + c <= region_cap) b = (unsigned long long)threadIdx.x * region_cap + at;
                    else *overflow = 1;
                }
                base[threadIdx.x] = b;
                hist[threadIdx.x] = 0;      // becomes the rank counter
                h_kmer[threadIdx.x] += t_kmer[threadIdx.x];
            }
            __syncthreads();
            GK_TICK(2);
            // ---- phase 2
            const u32 ndesc = s_ndesc;
            for (u32 i = threadIdx.x; i < ndesc; i += BLOCK) {
                const u32 dsc = desc[i];
                const int r = g0 + (int)(dsc & 63u), p = (int)((dsc >> 6) & 255u), rl = (int)((dsc >> 14) & 255u), o = (int)(dsc >> 22);
                if (base[o] == ~0ull) continue;
                const u32 ro = (u32)((r0 + r) * stride - a0);
                const int nrec = (rl + rmax - 1) / rmax;
                const u64 slot0 = base[o] + atomicAdd(&hist[o], (u32)nrec);
                for (int c = 0; c < nrec; c++) {
                    const int ps = p + c * rmax;
                    const int wl = min(rmax, rl - c * rmax);
                    write_record<SLOT>(tile, (ro + 1) * 8 + 2 * ps, wl + k - 1, out + (slot0 + c) * SLOT);
                }
            }
            GK_TICK(3);
            g0 = g1;
        }
    }
    GK_TFLUSH(0);
    __syncthreads();
    if (threadIdx.x < (u32)P && h_kmer[threadIdx.x]) atomicAdd(&kmer_counts[threadIdx.x], (unsigned long long)h_kmer[threadIdx.x]);
}

namespace gk {
// The routing kernel, launched on `st` with its counters in d_counts (SKM_COUNT_WORDS words: cursors, k-mer counts, overflow)
// and their copy-back to h_counts queued behind it; nothing is waited for.  skm_route_finish reads h_counts once the stream
// has been synchronised.  (gk_dist routes batch i+1 on a second stream while the owner pipeline of batch i runs.)
int skm_route_launch(gk_ctx *ctx, hipStream_t st, unsigned long long *d_counts, unsigned long long *h_counts, int k, const void *dev_records,
                     uint64_t nreads, int read_len, int P, void *dev_out, uint64_t out_cap_records) {
    if (!k_supported(k)) return fail(ctx, GK_E_UNSUPPORTED_K, "k=" + std::to_string(k) + " unsupported");
    if (P < 1 || P > MAX_PARTS) return fail(ctx, GK_E_INVALID, "P must be 1.." + std::to_string(MAX_PARTS));
    if (!dev_records && nreads) return fail(ctx, GK_E_INVALID, "null argument");
    if (read_len < 0 || read_len > 255) return fail(ctx, GK_E_FORMAT, "read_len must be 0..255");
    GK_HIP(ctx, hipSetDevice(ctx->device));
    for (int i = 0; i < SKM_COUNT_WORDS; i++) h_counts[i] = 0;
    const u64 nk = read_len >= k ? (u64)(read_len - k + 1) : 0;
    if (nreads == 0 || nk == 0) return GK_OK;
    if (!dev_out) return fail(ctx, GK_E_INVALID, "null record buffer");
    const u64 region_cap = out_cap_records / (u64)P;
    if (region_cap == 0) return fail(ctx, GK_E_CAPACITY, "record buffer too small: out_cap_records must be at least P");
    GK_HIP(ctx, hipMemsetAsync(d_counts, 0, SKM_COUNT_WORDS * sizeof(unsigned long long), st));
    const u32 stride = 1 + (read_len + 3) / 4;
    const u64 ntiles = (nreads + TILE_READS - 1) / TILE_READS;
    const int grid = (int)std::min<u64>(ntiles, (u64)ctx->cu_count * 6);
    const uint8_t *rec = (const uint8_t *)dev_records;
    // the last counter word: [0] a region overflowed, [1] a record's length byte exceeded read_len (clamped).  Both travel back
    // with the counts, so whoever finishes the route — on any thread — needs no other stream and no shared flag word.
    u32 *d_overflow = reinterpret_cast<u32 *>(d_counts + 2 * MAX_PARTS);
    if (skm_slot_bytes(k) == 16)
        hipLaunchKernelGGL(k_skm_route<16>, dim3(grid), dim3(BLOCK), 0, st, rec, nreads, stride, k, P, WindowLimits{read_len, d_overflow + 1}, region_cap, d_counts,
                           d_counts + MAX_PARTS, d_overflow, (uint8_t *)dev_out);
    else
        hipLaunchKernelGGL(k_skm_route<32>, dim3(grid), dim3(BLOCK), 0, st, rec, nreads, stride, k, P, WindowLimits{read_len, d_overflow + 1}, region_cap, d_counts,
                           d_counts + MAX_PARTS, d_overflow, (uint8_t *)dev_out);
    GK_HIP(ctx, hipGetLastError());
    GK_HIP(ctx, hipMemcpyAsync(h_counts, d_counts, SKM_COUNT_WORDS * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    return GK_OK;
}
int skm_route_finish(gk_ctx *ctx, const unsigned long long *h, bool launched, int P, uint64_t out_cap_records, uint64_t *rec_counts_host,
                     uint64_t *kmer_counts_host) {
    const u64 region_cap = out_cap_records / (u64)std::max(P, 1);
    unsigned long long worst = 0;
    for (int p = 0; p < P; p++) {
        rec_counts_host[p] = h[p];
        kmer_counts_host[p] = h[MAX_PARTS + p];
        worst = std::max(worst, h[p]);
    }
    if (!launched) return GK_OK;
    if ((u32)(h[2 * MAX_PARTS] >> 32)) return fail(ctx, GK_E_FORMAT, "a device record's length byte exceeds the declared read length");
    if ((u32)h[2 * MAX_PARTS] || worst > region_cap)
        return fail(ctx, GK_E_CAPACITY, "record buffer too small: the fullest owner region needs " + std::to_string(worst) +
                                            " records, out_cap_records / P = " + std::to_string(region_cap));
    return GK_OK;
}
}  // namespace gk

extern "C" {

int gk_skm_slot_bytes(int k) { return k_supported(k) ? skm_slot_bytes(k) : 0; }

int gk_shard_superkmers_dev(gk_ctx *ctx, int k, const void *dev_records, uint64_t nreads, int read_len, int P,
                            void *dev_out, uint64_t out_cap_records, uint64_t *rec_counts_host, uint64_t *kmer_counts_host) {
    if (!ctx) return fail(nullptr, GK_E_INVALID, "null ctx");
    if (!rec_counts_host || !kmer_counts_host) return fail(ctx, GK_E_INVALID, "null argument");
    if (!ctx->skm_counts) GK_HIP(ctx, hipMalloc(&ctx->skm_counts, SKM_COUNT_WORDS * sizeof(unsigned long long)));
    unsigned long long h[SKM_COUNT_WORDS];
    if (int rc = skm_route_launch(ctx, ctx->stream, (unsigned long long *)ctx->skm_counts, h, k, dev_records, nreads, read_len, P, dev_out, out_cap_records)) return rc;
    GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return skm_route_finish(ctx, h, nreads && read_len >= k, P, out_cap_records, rec_counts_host, kmer_counts_host);
}

}  // extern "C"
