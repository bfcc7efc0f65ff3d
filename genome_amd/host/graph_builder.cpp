// graph_builder.cpp — C++ twin of GraphBuilder.startup (S/scripts/GraphBuilder.scala:18-59) over
// genome.hpp: counts k-mers of a `.bin` read stream on the GPU, drops k-mers seen < rounds times,
// builds the de Bruijn graph, keeps the largest component, and writes the graph as text.
// The reference logs its counters through akka Logging (:34-53); here they are one JSON object.
//
//   graph_builder <reads.bin> <pairs> <k> [--rounds 3] [--take-first N] [--prefilter DISTINCT] [--no-retain] [--simplify]
//                 [--walk-pairs CUTOFF LO HI] [--out prefix]
//   --simplify runs removeBubbles + simplifyGraph (GraphSimplifier.scala:317-318) before writing;
//   --walk-pairs runs GraphSimplifier.startup's paired-end stage on the graph GraphBuilder hands over (:188-318): position
//   map, the pairs' walks with range LO to HI (the reference: 180 to 250, :146), node split at genome.cutoff = CUTOFF,
//   removeEdge, simplifyGraph; its counters join the JSON;
//   --out writes <prefix>.nodes.txt, .edges.txt, .contigs (GraphSimplifier.scala:338-347) and .dot (Graph.scala:74-88)
//
// Build: g++ -std=c++17 -O2 -I include genome_amd/host/graph_builder.cpp -L genome_amd -lgenome_amd
//        -Wl,-rpath,'$ORIGIN/..' -o genome_amd/host/graph_builder
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <iterator>

#include "genome.hpp"

int main(int argc, char **argv) {
    if (argc < 4) {
        std::fprintf(stderr, "usage: %s <reads.bin> <pairs> <k> [--rounds 3] [--take-first N] [--prefilter DISTINCT] [--no-retain] [--simplify] "
                             "[--walk-pairs CUTOFF LO HI] [--out prefix]\n", argv[0]);
        return 2;
    }
    const std::string infile = argv[1];
    genome::PairedEndData data;
    data.count = std::stoull(argv[2]);
    const int k = std::stoi(argv[3]);
    int rounds = 3;                               // GraphBuilder.scala:30
    uint64_t takeFirst = UINT64_MAX;              // genome.takeFirst
    uint64_t prefilter = 0;                       // expected distinct k-mers; 0 = no singleton pre-filter
    bool retain = true, simplify = false;
    int walkCutoff = -1, walkLo = 180, walkHi = 250;
    std::string out;
    for (int i = 4; i < argc; i++) {
        if (!std::strcmp(argv[i], "--rounds") && i + 1 < argc) rounds = std::stoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--take-first") && i + 1 < argc) takeFirst = std::stoull(argv[++i]);
        else if (!std::strcmp(argv[i], "--prefilter") && i + 1 < argc) prefilter = std::stoull(argv[++i]);
        else if (!std::strcmp(argv[i], "--no-retain")) retain = false;
        else if (!std::strcmp(argv[i], "--simplify")) simplify = true;
        else if (!std::strcmp(argv[i], "--walk-pairs") && i + 3 < argc) { walkCutoff = std::stoi(argv[++i]); walkLo = std::stoi(argv[++i]); walkHi = std::stoi(argv[++i]); }
        else if (!std::strcmp(argv[i], "--out") && i + 1 < argc) out = argv[++i];
        else { std::fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
    }
    try {
        std::ifstream f(infile, std::ios::binary);
        if (!f) throw std::runtime_error("cannot open " + infile);
        data.bin.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
        genome::Context ctx(0);
        auto kmersFreq = genome::FreqFilter::extractFilteredKmers(ctx, data, k, rounds, takeFirst, 0, prefilter);      // :32
        const uint64_t good = kmersFreq.size();                                                          // :34
        auto graph = genome::Graph::buildGraph(k, kmersFreq);                                            // :36
        auto [nodes, edges, totalLen] = graph.counts();                                                  // :39
        // :41-47 the two component histograms, on the graph as built (the reference computes them before retain)
        auto [hist, hist2] = graph.componentHistograms();
        uint64_t kept = nodes, comps = 0;
        if (retain) std::tie(kept, comps) = graph.retainLargestComponent();                              // :52-54
        // --simplify = GraphSimplifier.scala:317-318, applied to the graph GraphBuilder hands over (i.e. after retain)
        if (simplify) { graph.removeBubbles(); graph.simplifyGraph(); }
        uint64_t supPairs = 0, badPairs = 0, walked = 0, removedEdges = 0, newNodes = 0;
        if (walkCutoff >= 0) {
            auto graphMap = graph.getGraphMap();                                                         // GraphSimplifier.scala:188
            genome::Support support(ctx);
            graph.walkPairs(graphMap, support, data, takeFirst, walkLo, walkHi);                         // :213-263
            std::tie(supPairs, badPairs, walked) = support.sizes();                                      // :266 "Bad pairs"
            std::tie(removedEdges, newNodes) = graph.splitBySupport(support, walkCutoff);                // :272-316
            graph.simplifyGraph();                                                                       // :318
        }
        auto [n2, e2, l2] = graph.counts();
        std::printf("{\"k\":%d,\"rounds\":%d,\"good_kmers\":%llu,\"graph_nodes\":%llu,\"graph_edges\":%llu,"
                    "\"total_edges_length\":%llu,\"components\":%llu,\"max_component_size\":%llu,"
                    "\"retained_nodes\":%llu,\"retained_edges\":%llu,\"retained_edges_length\":%llu,",
                    k, rounds, (unsigned long long)good, (unsigned long long)nodes, (unsigned long long)edges,
                    (unsigned long long)totalLen, (unsigned long long)comps, (unsigned long long)kept,
                    (unsigned long long)n2, (unsigned long long)e2, (unsigned long long)l2);
        if (walkCutoff >= 0)
            std::printf("\"walk_pairs\":{\"supported_edge_pairs\":%llu,\"bad_pairs\":%llu,\"orientations_walked\":%llu,\"removed_edges\":%llu,\"new_nodes\":%llu},",
                        (unsigned long long)supPairs, (unsigned long long)badPairs, (unsigned long long)walked, (unsigned long long)removedEdges,
                        (unsigned long long)newNodes);
        auto dump = [](const char *name, const std::map<uint64_t, uint64_t> &h, const char *tail) {
            std::printf("\"%s\":[", name);
            bool first = true;
            for (const auto &p : h) { std::printf("%s[%llu,%llu]", first ? "" : ",", (unsigned long long)p.first, (unsigned long long)p.second); first = false; }
            std::printf("]%s", tail);
        };
        dump("components_histogram", hist, ",");          // GraphBuilder.scala:42 "Components histogram"
        dump("components_histogram_2", hist2, "}\n");     // :47 "Components histogram 2"
        if (!out.empty()) {                                                                              // :56 (Kryo file there)
            std::ofstream nf(out + ".nodes.txt"), ef(out + ".edges.txt");
            for (const auto &n : graph.getNodes()) nf << n.toString() << "\n";
            for (const auto &e : graph.getEdges()) ef << e.start.toString() << " " << e.end.toString() << " " << e.seq << "\n";
            std::ofstream cf(out + ".contigs"), df(out + ".dot");
            graph.writeContigs(cf);
            graph.writeDot(df);
        }
    } catch (const std::exception &e) {
        std::fprintf(stderr, "graph_builder: %s\n", e.what());
        return 1;
    }
    return 0;
}
