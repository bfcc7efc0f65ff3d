"""genome_amd — MI355X-native k-mer hashtable + de Bruijn graph-build core.

Host-side mirror of the reference's operator interface for the hot path
(`DNAMap[Int]`, `FreqFilter.extractFilteredKmers`, `Graph.buildGraph`) over a C-ABI HIP library
(include/genome_amd.h, genome_amd/csrc).  There is NO CPU fallback: every operation goes through
the HIP library and raises if it is missing or no gfx950 device is present.
"""
__version__ = "0.1.0"
