"""The C++ host side (genome_amd/host/genome.hpp + graph_builder.cpp, the twin of
GraphBuilder.startup, S/scripts/GraphBuilder.scala:18-59) run end to end on the GPU and compared
with the golden fixtures / the oracle."""
import json
import os
import subprocess

import pytest

from genome_amd import dna
from oracle import oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
EXE = os.path.join(ROOT, "genome_amd", "host", "graph_builder")


@pytest.fixture(scope="module")
def exe():
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "genome_amd", "csrc"), "host"])
    return EXE


@pytest.mark.parametrize("name", ["g_k11_p1", "g_k31_p1", "snp_k35_p1", "g_k63_p1", "snp_k11_p2"])
def test_graph_builder_cli_matches_golden(exe, tmp_path, name):
    fx = json.load(open(os.path.join(GOLDEN, name + ".json")))
    binf = tmp_path / "reads.bin"
    binf.write_bytes(bytes.fromhex(fx["bin_hex"]))
    out = tmp_path / "g"
    res = subprocess.run([exe, str(binf), str(fx["nreads"] // 2), str(fx["k"]), "--rounds", str(fx["rounds"]),
                          "--no-retain", "--out", str(out)], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    stats = json.loads(res.stdout)
    assert stats["good_kmers"] == len(fx["table_filtered"])
    assert stats["graph_nodes"] == len(fx["nodes"]) and stats["graph_edges"] == len(fx["edges"])
    assert stats["total_edges_length"] == sum(len(e[2]) for e in fx["edges"])
    assert open(str(out) + ".nodes.txt").read().split() == fx["nodes"]
    edges = [line.split() for line in open(str(out) + ".edges.txt").read().splitlines()]
    assert edges == fx["edges"]
    # contigs file (GraphSimplifier.scala:338-347): sequence line, then ">abacaba<i>"
    contigs = open(str(out) + ".contigs").read().splitlines()
    assert contigs[0::2] == [e[2] for e in fx["edges"]]
    assert contigs[1::2] == [f">abacaba{i}" for i in range(len(fx["edges"]))]
    dot = open(str(out) + ".dot").read().splitlines()                       # Graph.scala:74-88
    assert dot[0] == "digraph G {" and dot[-1] == "}" and len(dot) == len(fx["edges"]) + 2
    ids = {s: i + 1 for i, s in enumerate(fx["nodes"])}
    for line, (s, t, q) in zip(dot[1:-1], fx["edges"]):
        assert line == f"{ids[s]} -> {ids[t]} [label={q if len(q) <= 50 else len(q)}]"
    # --prefilter (exact two-pass singleton pre-filter): same outputs whenever rounds >= 2
    if fx["rounds"] >= 2:
        res = subprocess.run([exe, str(binf), str(fx["nreads"] // 2), str(fx["k"]), "--rounds", str(fx["rounds"]), "--prefilter", "1000",
                              "--no-retain", "--out", str(out) + "p"], capture_output=True, text=True)
        assert res.returncode == 0, res.stderr
        assert json.loads(res.stdout) == stats
        assert [line.split() for line in open(str(out) + "p.edges.txt").read().splitlines()] == fx["edges"]
    # --simplify = removeBubbles + simplifyGraph before writing
    res = subprocess.run([exe, str(binf), str(fx["nreads"] // 2), str(fx["k"]), "--rounds", str(fx["rounds"]),
                          "--no-retain", "--simplify", "--out", str(out) + "s"], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    assert [line.split() for line in open(str(out) + "s.edges.txt").read().splitlines()] == fx["edges_after_simplify"]


def test_graph_builder_cli_retain_and_errors(exe, tmp_path):
    fx = json.load(open(os.path.join(GOLDEN, "g_k11_p1.json")))
    binb = bytes.fromhex(fx["bin_hex"])
    binf = tmp_path / "reads.bin"
    binf.write_bytes(binb)
    res = subprocess.run([exe, str(binf), str(fx["nreads"] // 2), "11", "--rounds", "2", "--out", str(tmp_path / "r")],
                         capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    stats = json.loads(res.stdout)
    ref = O.PMap(11, 1)
    ref.count_reads(binb, fx["nreads"]); ref.delete_lt(2)
    og = O.Graph(ref)
    assert stats["components"] == og.num_components()
    # GraphBuilder.scala:41-47: components by node count / by summed out-edge length (computed before retain)
    h1, h2 = stats["components_histogram"], stats["components_histogram_2"]
    assert sum(c for _, c in h1) == sum(c for _, c in h2) == og.num_components()
    assert sum(v * c for v, c in h1) == og.num_nodes() and sum(v * c for v, c in h2) == og.total_edge_len()
    assert h1 == sorted(h1) and h2 == sorted(h2)
    assert stats["max_component_size"] == og.retain_largest() == stats["retained_nodes"]
    nlo, nhi = og.nodes()
    assert open(str(tmp_path / "r") + ".nodes.txt").read().split() == [dna.unpack(int(a), int(b), 11) for a, b in zip(nlo, nhi)]
    # takeFirst (genome.takeFirst, FreqFilter.scala:40,44): only the first pairs are counted
    res = subprocess.run([exe, str(binf), str(fx["nreads"] // 2), "11", "--rounds", "1", "--take-first", "10", "--no-retain"],
                         capture_output=True, text=True)
    ref2 = O.PMap(11, 1)
    ref2.count_reads(binb, 20)
    assert json.loads(res.stdout)["good_kmers"] == ref2.size()
    # error behaviour: unsupported k and a truncated stream are reported, not crashed on
    res = subprocess.run([exe, str(binf), "5", "32"], capture_output=True, text=True)
    assert res.returncode == 1 and "unsupported" in res.stderr
    res = subprocess.run([exe, str(binf), str(fx["nreads"]), "11"], capture_output=True, text=True)
    assert res.returncode == 1 and "truncated" in res.stderr
