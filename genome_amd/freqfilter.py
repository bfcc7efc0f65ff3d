"""FreqFilter.extractFilteredKmers (S/data/FreqFilter.scala:25-58) on the GPU."""
from __future__ import annotations

from .dnamap import Context, HipDNAMap
from .partitioned import PartitionedDNAMap


class PairedEndData:
    """S/data/PairedEndData.scala:11-36: `count` pairs in a `.bin` record stream (2 records/pair).
    The Java-serialised descriptor (:14-18, :39-42) is replaced by explicit fields."""

    def __init__(self, count: int, bin_bytes: bytes, insert: int = 0):
        self.count, self.bin, self.insert = count, bin_bytes, insert


def extractFilteredKmers(data: PairedEndData, k: int, rounds: int, ctx: Context | None = None,
                         take_first: int | None = None, partitions: int = 1, capacity_hint: int = 0,
                         prefilter_distinct: int = 0):
    """Count every canonical k-mer of the first `take_first` pairs (genome.takeFirst,
    FreqFilter.scala:40,44), then deleteAll(v < rounds) (:55).  Returns the DNAMap[Int].

    prefilter_distinct > 0 (single partition, rounds >= 2): run the exact two-pass singleton
    pre-filter sized for that many distinct k-mers first (genome_amd/prefilter.py) — same result,
    but k-mers seen once never take a table slot, so `capacity_hint` can be the number of k-mers
    seen at least twice."""
    ctx = ctx or Context(0)
    pairs = data.count if take_first is None else min(take_first, data.count)
    if prefilter_distinct:
        if partitions != 1 or rounds < 2:
            raise ValueError("the singleton pre-filter needs partitions == 1 and rounds >= 2 (it drops k-mers seen once)")
        from .prefilter import HipPrefilter
        kmers = HipDNAMap(ctx, k, capacity_hint)
        pf = HipPrefilter(ctx, k, prefilter_distinct)
        pf.add_reads(data.bin, 2 * pairs)
        pf.count_reads(kmers, data.bin, 2 * pairs)
        pf.close()
        kmers.deleteAll_lt(rounds)
        return kmers
    if partitions == 1:
        kmers = HipDNAMap(ctx, k, capacity_hint)
    else:
        kmers = PartitionedDNAMap(ctx, k, partitions, capacity_hint)
    kmers.count_reads(data.bin, 2 * pairs)
    kmers.deleteAll_lt(rounds)
    return kmers
