// sort_edges_oracle.cpp — TEST INFRASTRUCTURE, not product code.
//
// Sequential CPU restatement of OverlapGraph::sortEdges (reference src/OverlapGraph.cpp:722-764), the call that
// follows construct_edges in every workflow (src/ViralQuasispecies.cpp:297,359,434): every out-list is sorted by the
// edge's non-overlap length (src/Edge.h:58-63: len(read1) + len(read2) - 2 * overlap_len, in unsigned arithmetic), ties
// by vertex2, with std::sort — whose order among fully tied edges (the two orientation classes of one vertex pair) is
// a property of the host's libstdc++ in the reference too, so std::sort is what this file uses; adj_in is rebuilt
// from the sorted out-lists, vertex by vertex.  Only tests/ may load this.  Flat records of include/hcedge_host.h.
//
// PINNING: checked against the reference's own sortEdges, compiled as part of the fragment probe
// oracle/_ref/libhcref_edgecalc.so (oracle/Makefile `ref`), on the committed vectors tests/golden/sort_edges.json
// (tests/golden/make_golden_sort_edges.py): ties, out-lists beyond std::sort's insertion-sort threshold, wrapping lengths.
#include <stdint.h>

#include <algorithm>
#include <utility>
#include <vector>

#include "../include/hcedge_host.h"

extern "C" {

// in: the edges of adj_out, vertex by vertex in list order (v1 ascending, as hc_ec_get_edges / hc_host_graph_get
// return them); len_by_read[r]: Read::get_len() of read r (both mates for a pair, src/Read.h:203-212).
// out: the same after sortEdges; in_off (n_vertices + 1) / in_nodes (n): adj_in.  Returns 0, or 1 on a malformed input.
int hco_sort_edges(const hc_edge_rec* in, uint64_t n, uint64_t n_vertices, const uint32_t* len_by_read, hc_edge_rec* out, uint64_t* in_off,
                   uint64_t* in_nodes) {
    std::vector<std::vector<uint64_t>> adj_in(n_vertices);
    uint64_t k = 0, w = 0;
    for (uint64_t v = 0; v < n_vertices; v++) {
        std::vector<std::pair<hc_edge_rec, unsigned int>> pairs;  // :728-732
        for (; k < n && in[k].v1 == v; k++) {
            const hc_edge_rec& e = in[k];
            if (e.v2 >= n_vertices) return 1;
            const unsigned int nonoverlap = (unsigned int)len_by_read[e.read1] + (unsigned int)len_by_read[e.read2] - 2u * (unsigned int)e.len0;  // Edge.h:58-63
            pairs.push_back(std::make_pair(e, nonoverlap));
        }
        std::sort(pairs.begin(), pairs.end(), [](const std::pair<hc_edge_rec, unsigned int>& a, const std::pair<hc_edge_rec, unsigned int>& b) {  // :733-742
            if (a.second == b.second) return a.first.v2 < b.first.v2;
            return a.second < b.second;
        });
        for (const auto& p : pairs) {  // :745-749, :755-761
            out[w++] = p.first;
            adj_in[p.first.v2].push_back(p.first.v1);
        }
    }
    if (k != n) return 1;  // not grouped by v1 ascending
    uint64_t m = 0;
    for (uint64_t v = 0; v < n_vertices; v++) {
        in_off[v] = m;
        for (uint64_t x : adj_in[v]) in_nodes[m++] = x;
    }
    in_off[n_vertices] = m;
    return 0;
}

}  // extern "C"
