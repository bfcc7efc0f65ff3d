"""Deterministic synthetic read generator -> the reference `.bin` record stream.

Record framing is the reference's own: `[len:u8][ceil(len/4) bytes]`, base i of a read at bits
2(i%4) of byte i/4, codes A0 G1 C2 T3 (S/data/PairedEndData.scala:20-36, S/scripts/Convert2bin.scala:35-38).

PRNG: SplitMix64 in counter form (output i depends only on (seed, i)), so reads can be generated
in any chunking, on host (numpy, here) or on device (csrc/gk_synth.hip) with identical bytes.

  Mode U — every base independent uniform: base(r, j) = top 2 bits of out(seed_u, r*L + j).
  Mode G — genome base g = top 2 bits of out(seed_g, g); read r draws from stream seed_r at
           index r*(L+2): [0] start = out % (G-L+1); [1] strand = out >> 63 (1 = reverse
           complement); [2+j] error draw for emitted base j: error iff (out >> 40) < floor(e*2^24),
           substituted base = (b + 1 + (out & 0xffffffff) % 3) & 3.
"""
from __future__ import annotations

import numpy as np

GAMMA = np.uint64(0x9E3779B97F4A7C15)
SEED_U = 0xC0FFEE
SEED_G = 0xD1CE
SEED_R = 0xBEEF


def splitmix64_at(seed: int, idx: np.ndarray) -> np.ndarray:
    """SplitMix64 output number idx (0-based) of the stream started at `seed` (vectorised)."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed & (2**64 - 1)) + (idx.astype(np.uint64) + np.uint64(1)) * GAMMA
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def record_stride(L: int) -> int:
    return 1 + (L + 3) // 4


def pack_reads(bases: np.ndarray) -> np.ndarray:
    """(N, L) uint8 base codes -> (N, 1+ceil(L/4)) uint8 `.bin` records."""
    n, L = bases.shape
    nb = (L + 3) // 4
    pad = np.zeros((n, nb * 4), np.uint8)
    pad[:, :L] = bases
    q = pad.reshape(n, nb, 4)
    body = q[:, :, 0] | (q[:, :, 1] << 2) | (q[:, :, 2] << 4) | (q[:, :, 3] << 6)
    rec = np.empty((n, 1 + nb), np.uint8)
    rec[:, 0] = L
    rec[:, 1:] = body
    return rec


def reads_mode_u(n_reads: int, L: int, config_id: int = 0, first_read: int = 0) -> np.ndarray:
    """Mode U records for reads [first_read, first_read + n_reads)."""
    seed = SEED_U ^ config_id
    idx = (np.arange(n_reads, dtype=np.uint64)[:, None] + np.uint64(first_read)) * np.uint64(L) \
        + np.arange(L, dtype=np.uint64)[None, :]
    bases = (splitmix64_at(seed, idx) >> np.uint64(62)).astype(np.uint8)
    return pack_reads(bases)


def genome_bases(G: int, config_id: int = 0) -> np.ndarray:
    return (splitmix64_at(SEED_G ^ config_id, np.arange(G, dtype=np.uint64)) >> np.uint64(62)).astype(np.uint8)


def error_threshold(e: float) -> int:
    return int(e * (1 << 24))


def reads_mode_g(n_reads: int, L: int, G: int, e: float, config_id: int = 0, first_read: int = 0,
                 genome: np.ndarray | None = None) -> np.ndarray:
    """Mode G records for reads [first_read, first_read + n_reads)."""
    if genome is None:
        genome = genome_bases(G, config_id)
    seed = SEED_R ^ config_id
    r = np.arange(n_reads, dtype=np.uint64) + np.uint64(first_read)
    base_idx = r * np.uint64(L + 2)
    start = (splitmix64_at(seed, base_idx) % np.uint64(G - L + 1)).astype(np.int64)
    strand = (splitmix64_at(seed, base_idx + np.uint64(1)) >> np.uint64(63)).astype(bool)
    j = np.arange(L, dtype=np.int64)
    fwd = genome[start[:, None] + j[None, :]]
    rc = (3 - genome[start[:, None] + (L - 1 - j)[None, :]]).astype(np.uint8)
    bases = np.where(strand[:, None], rc, fwd).astype(np.uint8)
    draw = splitmix64_at(seed, base_idx[:, None] + np.uint64(2) + j.astype(np.uint64)[None, :])
    err = (draw >> np.uint64(40)) < np.uint64(error_threshold(e))
    sub = ((bases.astype(np.uint64) + np.uint64(1) + (draw & np.uint64(0xFFFFFFFF)) % np.uint64(3))
           & np.uint64(3)).astype(np.uint8)
    bases = np.where(err, sub, bases)
    return pack_reads(bases)


def bases_to_str(bases: np.ndarray) -> str:
    return "".join("AGCT"[int(b)] for b in bases)
