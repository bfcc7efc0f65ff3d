"""Emit the golden fixtures in this directory from the independent Python restatement
(oracle/pyref.py).  The reference itself cannot run here (no JVM), so these are
vectors of the build's own literal restatement — "parity unpinned" by the reference — and the
C oracle, and through it the HIP path, must reproduce them bit for bit.

Run:  python tests/golden/make_golden.py     (rewrites tests/golden/*.json; deterministic)
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from genome_amd import synth          # noqa: E402
from oracle import pyref as R         # noqa: E402

CASES = [  # name, k, P, rounds, n_reads, L, G, err, config_id
    ("g_k11_p1", 11, 1, 2, 90, 36, 240, 0.02, 101),
    ("g_k21_p3", 21, 3, 3, 160, 60, 300, 0.01, 102),
    ("g_k31_p1", 31, 1, 2, 120, 75, 320, 0.01, 103),
    ("g_k35_p2", 35, 2, 2, 120, 80, 300, 0.01, 104),
    ("g_k63_p1", 63, 1, 2, 100, 110, 340, 0.005, 105),
]


def snp_reads(G, L, cid, snps, copies=3, step=3):
    """Two haplotypes differing at `snps` positions, tiled by error-free reads from both strands:
    every SNP opens a bubble (two edges with the same start and end node, equal length)."""
    hap_a = synth.bases_to_str(synth.genome_bases(G, cid))
    hap_b = list(hap_a)
    for p in snps:
        hap_b[p] = "AGCT"[("AGCT".index(hap_b[p]) + 1) % 4]
    hap_b = "".join(hap_b)
    reads = []
    for hap in (hap_a, hap_b):
        for i in range(0, G - L + 1, step):
            r = hap[i:i + L]
            reads += [r, R.rev_comp(r)] * copies
        reads += [hap[G - L:]] * copies
    return reads


SNP_CASES = [  # name, k, P, rounds, L, G, config_id, snp positions
    ("snp_k11_p2", 11, 2, 3, 40, 260, 201, [60, 130, 200]),
    ("snp_k35_p1", 35, 1, 3, 90, 420, 202, [120, 300]),
]


def main():
    todo = []
    for name, k, P, rounds, n, L, G, err, cid in CASES:
        binb = synth.reads_mode_g(n, L, G, err, cid).tobytes()
        todo.append((name, k, P, rounds, n, L, G, err, cid, binb))
    for name, k, P, rounds, L, G, cid, snps in SNP_CASES:
        reads = snp_reads(G, L, cid, snps)
        todo.append((name, k, P, rounds, len(reads), L, G, 0.0, cid, R.reads_to_bin(reads)))
    for name, k, P, rounds, n, L, G, err, cid, binb in todo:
        reads = R.reads_from_bin(binb, n)
        m = R.extract_filtered_kmers(reads, k, rounds, P, do_filter=False)
        table = [[R.pack(s)[0], R.pack(s)[1], c] for s, c in m.sorted_items()]
        m.delete_lt(rounds)
        table_f = [[R.pack(s)[0], R.pack(s)[1], c] for s, c in m.sorted_items()]
        g = R.build_graph(k, m)
        nodes, edges = g.canonical()
        g.remove_bubbles()
        _, edges_b = g.canonical()
        g.simplify()
        nodes_s, edges_s = g.canonical()
        fx = dict(name=name, k=k, P=P, rounds=rounds, nreads=n, L=L, G=G, err=err, config_id=cid,
                  bin_hex=binb.hex(), occurrences=n * (L - k + 1), table=table, table_filtered=table_f,
                  nodes=nodes, edges=[list(e) for e in edges], edges_after_bubbles=[list(e) for e in edges_b],
                  nodes_after_simplify=nodes_s, edges_after_simplify=[list(e) for e in edges_s])
        with open(os.path.join(HERE, name + ".json"), "w") as f:
            json.dump(fx, f, separators=(",", ":"))
        print(name, "table", len(table), "filtered", len(table_f), "nodes", len(nodes), "edges", len(edges),
              "after bubbles", len(edges_b), "after simplify", len(nodes_s), len(edges_s))


if __name__ == "__main__":
    main()
