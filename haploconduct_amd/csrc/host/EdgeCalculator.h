// EdgeCalculator.h — reads the overlaps file and builds the edges of the overlap graph
// (reference src/EdgeCalculator.h:25-64).  Same constructor, same public methods and counters;
// the OpenMP loop of process_overlaps is replaced by hc_score_batch on the MI355X.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "../../../include/hcedge.h"
#include "Edge.h"
#include "FastqStorage.h"
#include "Overlap.h"
#include "OverlapGraph.h"
#include "OverlapsParser.h"
#include "Types.h"

namespace hc {

hc_settings to_hc_settings(const ProgramSettings& ps);

// The serial insert of process_overlaps (src/EdgeCalculator.cpp:441-538) as a free function so
// that it can be exercised without a device context.
struct InsertCounters {
    unsigned int inclusion_count = 0, dup_count = 0;
    uint64_t edges_added = 0;
};
void insert_edge(OverlapGraph& g, const ProgramSettings& ps, Edge& e, InsertCounters& c);

class EdgeCalculator {
public:
    unsigned int self_overlap_count = 0;   // never incremented by the reference either (its counting code is commented out)
    unsigned int inclusion_count = 0;
    unsigned int dup_count = 0;

    EdgeCalculator(std::shared_ptr<FastqStorage> fastq, std::shared_ptr<OverlapGraph> graph, const ProgramSettings& ps);
    ~EdgeCalculator();
    EdgeCalculator(const EdgeCalculator&) = delete;
    EdgeCalculator& operator=(const EdgeCalculator&) = delete;

    void construct_edges();                                // src/EdgeCalculator.cpp:561-666
    // src/EdgeCalculator.cpp:67-139 on arbitrary strings (used by SRBuilder::merge_self_overlap in the
    // reference): scored on the device through a two-read scratch store, finalised with the host libm.
    double overlap_score(const std::string& seq1, const std::string& seq2, const std::string& score1,
                         const std::string& score2, unsigned int pos, double& mismatch_rate);
    double phred_to_prob(int phred) const;                 // src/EdgeCalculator.cpp:59-63

    // statistics of the last construct_edges() (build-owned)
    struct Stats {
        uint64_t lines_read = 0, malformed = 0, self_overlaps = 0, prefilter_rejected = 0, scored = 0, edges_added = 0,
                 nonedges_written = 0, ambiguous = 0, silently_dropped = 0;
        double t_parse = 0, t_score = 0, t_insert = 0, t_write = 0;
    } stats;

private:
    void process_overlaps(const std::vector<ParsedOverlap>& batch);   // src/EdgeCalculator.cpp:389-557
    ProgramSettings program_settings;
    std::shared_ptr<FastqStorage> fastq_storage;
    std::shared_ptr<OverlapGraph> overlap_graph;
    hc_settings m_cs;
    hc_ctx* m_ctx = nullptr;
    hc_overlap_rec* m_rec = nullptr;  // page-locked staging (hc_host_alloc), grow-only
    hc_result_rec* m_res = nullptr;
    size_t m_cap = 0;
    std::string m_nonedge_buf;
};

}  // namespace hc
