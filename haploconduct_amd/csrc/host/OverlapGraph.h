// OverlapGraph.h — the structure the hot path emits into (reference src/OverlapGraph.h:33-131;
// only the methods on the edge-calculation path: src/OverlapGraph.cpp:88-101,150-229,285-311).
// Adjacency "lists" are order-preserving vectors: push_back on insert, erase keeps order, so
// iteration order equals the reference's std::list order.
#pragma once
#include <memory>
#include <vector>

#include "Edge.h"
#include "FastqStorage.h"
#include "Types.h"

namespace hc {

class OverlapGraph {
public:
    OverlapGraph(unsigned int V, std::shared_ptr<FastqStorage> fastq, const ProgramSettings& ps)
        : adj_out(V), adj_in(V), inclusions(V, 0), fastq_storage(std::move(fastq)), program_settings(ps) {}

    node_id_t addVertex(read_id_t read_ID) {             // src/OverlapGraph.cpp:88-92
        vertex_to_read.push_back(read_ID);
        vertex_count++;
        return vertex_to_read.size() - 1;
    }
    void addEdge(const Edge& edge);                       // :94-101
    Edge removeEdgeWithOri(node_id_t v, node_id_t w, bool opposite_orientations);             // :150-194
    double checkEdgeWithOri(node_id_t v, node_id_t w, bool opposite_orientations) const;      // :198-229
    Edge* getEdgeInfoWithOri(node_id_t v, node_id_t w, bool opposite_orientations, bool reverse_allowed = true);  // :285-306
    unsigned int getEdgeCount() const { return edge_count; }
    unsigned int getVertexCount() const { return vertex_count; }

    std::vector<read_id_t> vertex_to_read;
    std::vector<std::vector<Edge>> adj_out;
    std::vector<std::vector<node_id_t>> adj_in;
    std::vector<uint8_t> inclusions;                      // boost::dynamic_bitset in the reference

private:
    unsigned int vertex_count = 0;
    unsigned int edge_count = 0;
    std::shared_ptr<FastqStorage> fastq_storage;
    ProgramSettings program_settings;
};

}  // namespace hc
