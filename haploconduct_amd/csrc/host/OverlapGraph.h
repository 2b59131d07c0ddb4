// OverlapGraph.h — the structure the hot path emits into (reference src/OverlapGraph.h:33-131;
// only the methods on the edge-calculation path: src/OverlapGraph.cpp:88-101,150-229,285-311).
// Adjacency "lists" are order-preserving arrays (ArenaList.h): push_back on insert, erase keeps order, so
// iteration order equals the reference's std::list order.
#pragma once
#include <atomic>
#include <memory>
#include <vector>

#include "../../../include/hcedge.h"
#include "ArenaList.h"
#include "Edge.h"
#include "FastqStorage.h"
#include "Types.h"

namespace hc {

// Which (unordered vertex pair, orientation class) slots hold an edge.  The reference answers that by walking
// two adjacency lists per candidate edge (checkEdgeWithOri); almost every answer is "none", and the list of the
// in-vertex is a cold place in memory.  One probe of this table (whose address is known from the candidate
// record alone, so it can be prefetched) settles the common case; the lists are only walked on a hit.
class EdgeSlotIndex {
public:
    EdgeSlotIndex() { rebuild(1u << 12); }
    static bool representable(uint64_t v, uint64_t w) { return (v | w) < ((uint64_t)1 << 31); }
    static uint64_t key(uint64_t v, uint64_t w, bool opposite_orientations) {
        const uint64_t lo = v < w ? v : w, hi = v < w ? w : v;
        return (lo << 33) | (hi << 1) | (uint64_t)opposite_orientations;
    }
    bool contains(uint64_t k) const {
        for (size_t h = slot_of(k);; h = (h + 1) & mask_) {
            if (tab_[h].key == k) return tab_[h].count != 0;
            if (tab_[h].key == kEmpty) return false;
        }
    }
    void add(uint64_t k) {
        if ((filled_ + 1) * 10 > (mask_ + 1) * 6) rebuild((mask_ + 1) * (live_ * 4 > filled_ ? 2 : 1));
        for (size_t h = slot_of(k);; h = (h + 1) & mask_) {
            if (tab_[h].key == k) {
                if (tab_[h].count++ == 0) live_++;
                return;
            }
            if (tab_[h].key == kEmpty) {
                tab_[h].key = k;
                tab_[h].count = 1;
                filled_++;
                live_++;
                return;
            }
        }
    }
    void remove(uint64_t k) {  // the slot stays (count 0) and is dropped at the next rebuild
        for (size_t h = slot_of(k);; h = (h + 1) & mask_) {
            if (tab_[h].key == k) {
                if (tab_[h].count && --tab_[h].count == 0) live_--;
                return;
            }
            if (tab_[h].key == kEmpty) return;
        }
    }
    void prefetch(uint64_t k) const { __builtin_prefetch(&tab_[slot_of(k)]); }
    void bulk_add(const uint64_t* keys, size_t n, unsigned n_threads);  // add(keys[0..n)), on several threads

private:
    struct Entry {
        uint64_t key;
        uint32_t count;
    };
    static constexpr uint64_t kEmpty = ~(uint64_t)0;
    size_t slot_of(uint64_t k) const {
        uint64_t x = k * 0x9E3779B97F4A7C15ull;
        x ^= x >> 29;
        return (size_t)x & mask_;
    }
    void rebuild(size_t cap) {
        std::vector<Entry> old;
        old.swap(tab_);
        tab_.assign(cap, Entry{kEmpty, 0});
        mask_ = cap - 1;
        filled_ = live_ = 0;
        for (const Entry& e : old)
            if (e.key != kEmpty && e.count) {
                size_t h = slot_of(e.key);
                while (tab_[h].key != kEmpty) h = (h + 1) & mask_;
                tab_[h] = e;
                filled_++;
                live_++;
            }
    }
    std::vector<Entry> tab_;
    size_t mask_ = 0, filled_ = 0, live_ = 0;
};

class OverlapGraph {
public:
    OverlapGraph(unsigned int V, std::shared_ptr<FastqStorage> fastq, const ProgramSettings& ps)
        : adj_out(V), adj_in(V), inclusions(V, 0), fastq_storage(std::move(fastq)), program_settings(ps) {}

    node_id_t addVertex(read_id_t read_ID) {             // src/OverlapGraph.cpp:88-92
        vertex_to_read.push_back(read_ID);
        vertex_count++;
        return vertex_to_read.size() - 1;
    }
    ~OverlapGraph();
    OverlapGraph(const OverlapGraph&) = delete;
    OverlapGraph& operator=(const OverlapGraph&) = delete;
    void addEdge(const Edge& edge);                       // :94-101
    // The whole graph at once, as the device's duplicate resolution hands it over (hc_graph_fetch): adj_out lists back
    // to back in vertex order with their offsets, adj_in likewise.  The graph must be empty.  Lists point into two
    // arenas owned by the graph; the slot index is built on the first call that needs it.  reads[r] = m_read_vec[r].
    // edges_arrived (optional): the edge array is still being filled from its front — records [0, *edges_arrived) are there; a worker
    // waits for its vertices' records (the stage fetches the edges in pieces and adopts behind the copy).  *abandon set by the filler
    // (its copy failed): the workers stop waiting and the call throws.
    void adopt_csr(const hc_edge_rec* edges, const uint64_t* out_off, const uint32_t* in_nodes, const uint64_t* in_off,
                   const uint8_t* inclusion_bits, Read* const* reads, size_t n_reads, unsigned n_threads,
                   const std::atomic<size_t>* edges_arrived = nullptr, const std::atomic<bool>* abandon = nullptr);
    // the addEdge calls of pool[order[0]], pool[order[1]], ... as one parallel fill
    void bulk_add_edges(const Edge* pool, const std::vector<uint32_t>& order, unsigned n_threads);
    // src/OverlapGraph.cpp:722-764, the call that follows construct_edges in every workflow: out-lists sorted by
    // non-overlap length, then vertex2 (std::sort, as the reference: its order among fully tied edges is part of the
    // behaviour), adj_in rebuilt from the sorted out-lists.  len_by_read[r] = Read::get_len() of m_read_vec[r];
    // lists are independent, so vertex ranges are sorted on n_threads threads.
    void sortEdges(const uint32_t* len_by_read, unsigned n_threads = 1);
    void sort_out_list(node_id_t v, const uint32_t* len_by_read);  // sortEdges' treatment of one out-list (:724-749)
    void rebuild_in_lists(unsigned n_threads);                     // adj_in from the out-lists, :751-762
    Edge removeEdge(node_id_t v, node_id_t w);                                                // :102-146
    double checkEdge(node_id_t v, node_id_t w, bool reverse_allowed) const;                   // :233-259
    // :608-719 (--add_duplicates; called at the end of construct_edges, EdgeCalculator.cpp:650-652): every edge once more
    // between the vertices of the reverse-complemented reads.  The mirrored edges carry no reverse offsets and no
    // mismatch rate in the reference (uninitialised / -1); here: pos3 = pos4 = 0, mismatch rate -1.
    void addEquivalentEdges(unsigned int* n_built = nullptr, unsigned int* n_doubles = nullptr);
    Edge removeEdgeWithOri(node_id_t v, node_id_t w, bool opposite_orientations);             // :150-194
    double checkEdgeWithOri(node_id_t v, node_id_t w, bool opposite_orientations) const;      // :198-229
    Edge* getEdgeInfoWithOri(node_id_t v, node_id_t w, bool opposite_orientations, bool reverse_allowed = true);  // :285-306
    // Hints for the serial insert (an edge lands at random places in memory): the slot of the pair in the
    // index and the header of the in-vertex's in-list first; a few edges later, once the header is in cache,
    // the end of that list.
    void prefetch_slot(node_id_t v, node_id_t w, bool opposite_orientations) const {
        if (slots_valid && EdgeSlotIndex::representable(v, w)) slots.prefetch(EdgeSlotIndex::key(v, w, opposite_orientations));
        if (w < adj_in.size()) __builtin_prefetch(&adj_in[w]);
    }
    void prefetch_in_list(node_id_t w) const {
        if (w < adj_in.size() && !adj_in[w].empty()) __builtin_prefetch(adj_in[w].data() + adj_in[w].size() - 1, 1);
    }
    unsigned int getEdgeCount() const { return edge_count; }
    unsigned int getVertexCount() const { return vertex_count; }

    std::vector<read_id_t> vertex_to_read;
    std::vector<ArenaList<Edge>> adj_out;
    std::vector<ArenaList<node_id_t>> adj_in;
    std::vector<uint8_t> inclusions;                      // boost::dynamic_bitset in the reference

private:
    void ensure_slots() const;  // after adopt_csr the index is built lazily
    mutable EdgeSlotIndex slots;  // kept in step with adj_out by addEdge / removeEdgeWithOri
    mutable bool slots_valid = true;
    Edge* out_arena = nullptr;        // storage of the lists adopt_csr made (lists that have grown since own theirs)
    node_id_t* in_arena = nullptr;
    unsigned int vertex_count = 0;
    unsigned int edge_count = 0;
    std::shared_ptr<FastqStorage> fastq_storage;
    ProgramSettings program_settings;
};

}  // namespace hc
