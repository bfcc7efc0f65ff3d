"""Host-side mirror of the reference's sequence primitives (S/dna/Base.scala, S/dna/DNASeq.scala):
just enough to move k-mers across the C-ABI — packing, unpacking and reverse complement of base
strings.  (Bit layout: base i at bits 2i, A0 G1 C2 T3, first base at the LSB; DNASeq.scala:84,256.)
"""
from __future__ import annotations

import numpy as np

BASES = "AGCT"                                   # Base.scala:13-18
_CODE = {c: i for i, c in enumerate(BASES)}
_COMP = {"A": "T", "T": "A", "G": "C", "C": "G"}  # Base.scala:19


def pack(seq: str) -> tuple[int, int]:
    """DNASeq.newBuilder (DNASeq.scala:237-281) -> (lo, hi)."""
    lo = hi = 0
    for i, c in enumerate(seq):
        if i < 32:
            lo |= _CODE[c] << (2 * i)
        else:
            hi |= _CODE[c] << (2 * (i - 32))
    return lo, hi


def unpack(lo: int, hi: int, k: int) -> str:
    return "".join(BASES[((lo if i < 32 else hi) >> (2 * (i % 32))) & 3] for i in range(k))


def rev_complement(seq: str) -> str:
    """DNASeq.revComplement (DNASeq.scala:27-28)."""
    return "".join(_COMP[c] for c in reversed(seq))


def pack_many(seqs) -> tuple[np.ndarray, np.ndarray]:
    lo = np.zeros(len(seqs), np.uint64)
    hi = np.zeros(len(seqs), np.uint64)
    for i, s in enumerate(seqs):
        a, b = pack(s)
        lo[i], hi[i] = a, b
    return lo, hi


def reads_to_bin(reads) -> bytes:
    """Convert2bin.write (S/scripts/Convert2bin.scala:35-38): [len:u8][toByteArray] per read."""
    out = bytearray()
    for r in reads:
        if len(r) > 255:
            raise ValueError("a .bin record holds at most 255 bases")
        out.append(len(r))
        data = bytearray((len(r) + 3) // 4)
        for i, c in enumerate(r):
            data[i // 4] |= _CODE[c] << (2 * (i % 4))
        out += data
    return bytes(out)


def unpack_2bit(buf: np.ndarray, nbases: int) -> str:
    """Edge sequences come back as 2-bit codes, 4 per byte, LSB first."""
    return "".join(BASES[(int(buf[i // 4]) >> (2 * (i % 4))) & 3] for i in range(nbases))
