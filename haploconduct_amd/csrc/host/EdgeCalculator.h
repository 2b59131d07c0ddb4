// EdgeCalculator.h — reads the overlaps file and builds the edges of the overlap graph
// (reference src/EdgeCalculator.h:25-64).  Same constructor, same public methods and counters;
// the OpenMP loop of process_overlaps is replaced by hc_score_batch on the MI355X.
#pragma once
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "../../../include/hcedge.h"
#include "Edge.h"
#include "FastqStorage.h"
#include "Overlap.h"
#include "OverlapGraph.h"
#include "OverlapsParser.h"
#include "WorkerPool.h"
#include "Types.h"

namespace hc {

hc_settings to_hc_settings(const ProgramSettings& ps);
// The stage's device sequence once on two dummy reads (hc_ec_api.cpp): starts the HIP runtime and loads the kernels, best effort; meant for
// a thread beside the caller's FASTQ parsing on a process's first use of a device.
void warm_device_code(hc_settings cs) noexcept;
bool warm_up_pays(const ProgramSettings& ps);  // FASTQ files of 64 MiB and more (HC_WARM=0 / 1 overrides)

// The serial insert of process_overlaps (src/EdgeCalculator.cpp:441-538) as a free function so
// that it can be exercised without a device context.
struct InsertCounters {
    unsigned int inclusion_count = 0, dup_count = 0;
    uint64_t edges_added = 0;
};
void insert_edge(OverlapGraph& g, const ProgramSettings& ps, Edge& e, InsertCounters& c);

// The same result as calling insert_edge() on every element of `admitted` in order, computed by
// sorting instead of scanning adjacency lists:
//   * candidates for one graph slot — same unordered vertex pair and same (ori1 == ori2) class — are
//     independent of all others, so the admitted edges are grouped by that key (stable: sequence
//     order is kept inside a group);
//   * inside a group the reference's replace / keep decisions (score, then the tie-break chain of
//     src/EdgeCalculator.cpp:470-521) are replayed in sequence order: the survivor, the number of
//     duplicates and the record that was inserted FIRST (it alone decides OverlapGraph::inclusions,
//     :459-468) come out exactly as in the sequential loop;
//   * an adjacency list holds its surviving edges in the order of their own insertion (a replaced
//     edge is erased and the winner appended, :523-530), i.e. ordered by the survivors' sequence
//     numbers: survivors are appended to the graph in that order.
// The graph must not hold edges yet.  `admitted[0..n)` is consumed (edges are normalised in place).
void resolve_admitted_edges(OverlapGraph& g, const ProgramSettings& ps, Edge* admitted, size_t n, InsertCounters& c);

// The Edge compute_overlap builds for an admitted candidate (src/EdgeCalculator.cpp:219-232, :254-270, :292-308,
// :353-379).  What it needs to know about a read comes in one row per read.
struct ReadInfo {
    Read* read;
    node_id_t vertex;       // get_vertex_id(true)
    node_id_t vertex_rev;   // get_vertex_id(false): the reverse-complemented read's vertex (--add_duplicates only)
    uint32_t len_a, len_b;  // get_seq_len(0) of a single-end read; get_seq_len(1), get_seq_len(2) of a pair
    uint8_t paired, vertex_set, vertex_rev_set, pad;
};
// add_duplicates: vertices by orientation (src/EdgeCalculator.cpp:176-179) instead of the normal one (:181-182)
Edge edge_from_admit(const hc_admit_rec& a, const ReadInfo* read_info, bool add_duplicates = false);

// The resident process: a stage's devices (contexts, text blocks, small blocks) outlive it and serve the process's next stage
// (EdgeCalculator's constructor takes them over when the device list and block size fit).  off: frees what is parked.
void keep_devices_resident(bool on);

class EdgeCalculator {
public:
    unsigned int self_overlap_count = 0;   // never incremented by the reference either (its counting code is commented out)
    unsigned int inclusion_count = 0;
    unsigned int dup_count = 0;

    EdgeCalculator(std::shared_ptr<FastqStorage> fastq, std::shared_ptr<OverlapGraph> graph, const ProgramSettings& ps);
    ~EdgeCalculator();
    EdgeCalculator(const EdgeCalculator&) = delete;
    EdgeCalculator& operator=(const EdgeCalculator&) = delete;

    void construct_edges() { run_stage(false); }           // src/EdgeCalculator.cpp:561-666
    // construct_edges() followed by OverlapGraph::sortEdges() (src/ViralQuasispecies.cpp:281,297 — what every workflow
    // does) as one call: the adjacency lists come from the device already in sortEdges order.
    void construct_edges_sorted() { run_stage(true); }
    // Reads -> graph with no overlaps file in between (what savage.py:664,713 does with rust-overlaps, scripts/sfo2overlaps.py
    // and an overlaps.txt): the candidates are found on the device among the reads of this stage's own store
    // (hc_find_overlaps: err_rate, min_overlap, HC_FIND_* flags), the SFO ingest runs on the records where they are
    // (hc_found_to_overlaps' sort + matching), and the overlaps file's text goes from memory into the stage's text blocks.
    // Same graph as writing that file and calling construct_edges[_sorted] on it.  *n_found / *n_lines: SFO records, overlap lines.
    void construct_edges_from_reads(double err_rate, uint32_t min_overlap, uint32_t find_flags, bool then_sort, uint64_t* n_found,
                                    uint64_t* n_lines);
    // The same with NOTHING between the reads and the graph but device memory (round 6): the finder's records stay on the
    // device, the script's flip / sort / matching / uniq run there (hc_found_to_lines_device) and leave the overlaps file's lines as parsed
    // records, which the stage's text blocks take as they are (hc_textblock_submit_lines): no text is written, copied or parsed.  Where the
    // device cannot decide (an assert of the script, ids outside its sort keys) the call takes construct_edges_from_reads' route, which
    // raises what the script raises.  Graph, inclusions, counters and nonedge_overlaps.txt are that route's, byte for byte.
    // *device_route (may be null): whether the lines stayed on the device.
    void construct_edges_from_store(double err_rate, uint32_t min_overlap, uint32_t find_flags, bool then_sort, uint64_t* n_found,
                                    uint64_t* n_lines, int* device_route);
    // The pipelines' own input to stage a — the SFO file `rust-overlaps` wrote (savage.py:664, polyte.py:514) — straight to the graph: what
    // scripts/sfo2overlaps.py, original_overlaps.txt and the binary's text parser do in three steps.  A canonical file (single tabs, plain
    // decimal numbers: what the tool writes) is read on the device into records that take the finder's place (hc_set_found_from_sfo_text), and the
    // rest is construct_edges_from_store's; any other file, and any input the device does not decide, goes through the host's ingest
    // (hc_sfo2overlaps' code), its text in memory, and the text blocks — which raise what the script raises.  Pinned end to end: the ingest
    // by the script's own outputs (tests/golden/sfo), the stage by the reference's construct_edges + sortEdges.
    void construct_edges_from_sfo(const std::string& sfo_path, bool then_sort, uint64_t* n_records, uint64_t* n_lines, int* device_route);
    // src/EdgeCalculator.cpp:67-139 on arbitrary strings (used by SRBuilder::merge_self_overlap in the
    // reference): scored on the device through a two-read scratch store, finalised with the host libm.
    double overlap_score(const std::string& seq1, const std::string& seq2, const std::string& score1,
                         const std::string& score2, unsigned int pos, double& mismatch_rate);
    double phred_to_prob(int phred) const;                 // src/EdgeCalculator.cpp:59-63
    unsigned int device_count() const { return (unsigned int)m_dev.size(); }

    // statistics of the last construct_edges() (build-owned)
    struct Stats {
        uint64_t lines_read = 0, malformed = 0, self_overlaps = 0, prefilter_rejected = 0, scored = 0, edges_added = 0,
                 nonedges_written = 0, ambiguous = 0, silently_dropped = 0;
        double t_parse = 0, t_score = 0, t_insert = 0, t_write = 0;
        // blocks of the overlaps file's text: parsed on the device / taken over by the host's tokeniser (a line that is not
        // plain, an unknown id, more lines than room) / device blocks that ran twice because their row buffers had to grow
        uint64_t device_blocks = 0, host_blocks = 0, regrown_blocks = 0;
        uint64_t host_lines = 0;  // lines of device-parsed blocks the host's tokeniser read one by one (per-line fallback)
    } stats;

private:
    // One device of the stage: a context with its own copy of the read store and the blocks it has in flight.
    struct Device {
        hc_ctx* ctx = nullptr;
        int device_id = 0;
        hc_block* blk[2] = {nullptr, nullptr};       // blocks of host-parsed records
        std::vector<hc_textblock*> tblk;             // blocks of the file's text (the device parses): m_text_depth per device
    };
    // What the collector makes of one scored block, in sequence order.
    struct BlockOut {
        std::vector<hc_admit_rec> admitted;
        std::string nonedge_text;
        uint64_t nonedges = 0, ambiguous = 0;
    };
    // keep_devices_resident (the resident process): the devices of the last stage, parked for the next one
    struct Park {
        std::mutex mu;
        bool keep = false;
        std::vector<Device> devices;
        std::vector<int> device_ids;
        size_t text_block = 0, odd_blk_cap = 0;
        uint32_t odd_line_cap = 0;
        hc_block* odd_blk = nullptr;
    };
    static Park g_park;
    static void park_destroy_locked();
    friend void keep_devices_resident(bool on);
    void run_stage(bool then_sort);
    std::shared_ptr<const std::string> m_text_override;  // construct_edges_from_reads: the overlaps file's text, in memory
    const hc_line_rec* m_lines_override = nullptr;       // construct_edges_from_store: the overlaps file's lines, parsed, in device memory
    uint64_t m_lines_override_n = 0;
    void score_device_lines(OverlapsParser& parser, std::vector<Overlap>& rejected, ParseCounters& pc);  // ... sent through the text blocks
    bool run_stage_from_found(bool then_sort, uint64_t* n_lines);  // the found records of m_ctx -> lines on the device -> run_stage; false: not the device's
    void score_host_parsed(OverlapsParser& parser, std::vector<Overlap>& rejected, ParseCounters& pc);   // the file tokenised on host threads
    void score_device_parsed(OverlapsParser& parser, std::vector<Overlap>& rejected, ParseCounters& pc); // the file's text sent to the device
    void finalize_text_block(const IdIndex& ids, const hc_text_row* rows, uint64_t n_rows, BlockOut& out, unsigned threads = 0);
    // per-line fallback (score_device_parsed): what the host makes of the lines of a block the device's parser did not read
    struct OddLines {
        std::vector<hc_text_row> rows;  // the block's rows with the odd lines' rows spliced in (file order)
        std::vector<std::pair<uint32_t, Overlap>> rejected;  // (line number in the block, line) the prefilter rejected
        ParseCounters pc;
        uint64_t scored = 0;
    };
    void score_odd_lines(const OverlapsParser& parser, const char* block_text, const hc_text_result& tr, OddLines& odd);
    hc_block* m_odd_blk = nullptr;  // the passing odd lines of a block are scored as one small block
    size_t m_odd_blk_cap = 0;
    std::mutex m_odd_mu;
    uint32_t m_odd_line_cap = 4096;  // entries of a text block's list of such lines (HC_PARSE_FALLBACK=block: 0, the whole block goes to the host)
    void collect_read_info();
    void finalize_block(const ParsedBatch& batch, const hc_gather_row* rows, uint64_t n_rows, uint64_t base, BlockOut& out);
    void consume_block(BlockOut& out);  // serial half: insert (or collect) + nonedge_overlaps.txt, :431-555
    void resolve_on_device(bool sorted);
    void resolve_on_host();
    std::vector<ReadInfo> m_read_info;
    std::unique_ptr<WorkerPool> m_build_pool;  // the per-block finalisation runs on these
    ProgramSettings program_settings;
    std::shared_ptr<FastqStorage> fastq_storage;
    std::shared_ptr<OverlapGraph> overlap_graph;
    hc_settings m_cs;
    hc_ctx* m_ctx = nullptr;          // = m_dev[0].ctx, the primary device (duplicate resolution runs there)
    std::vector<Device> m_dev;
    size_t m_block_cap = 0;           // candidates the hc_block objects were created for
    bool m_serial_insert = false;     // HC_INSERT_MODE=serial: per-edge inserts even into an empty graph
    bool m_host_resolve = false;      // HC_RESOLVE=host: duplicate resolution on the host threads instead of the device
    bool m_host_parse = false;        // HC_PARSE=host: the overlaps file is tokenised on the host threads instead of the device
    // CPUs of the NUMA node the primary device hangs on (empty: one node, unknown, or HC_NUMA=0): the stage's threads copy
    // the file's text into page-locked buffers next to that device, so they run there
    std::vector<int> m_node_cpus;
    void bind_here() const;  // the calling thread -> m_node_cpus
    std::unique_ptr<WorkerPool> m_pool;  // the stage's worker threads (lent to the overlaps parser of every call: starting and
                                         // joining 31 threads per file cost 25 ms)
    std::thread m_cleanup;            // frees of large buffers, off the caller's clock (defer_cleanup)
    void defer_cleanup(std::function<void()> work);
    size_t m_text_block = 16u << 20;  // bytes of text per device-parsed block (HC_TEXT_BLOCK)
    size_t m_text_depth = 10;         // text blocks in flight per device at most (HC_TEXT_DEPTH; fewer for a short file): copies of later blocks beside the device's work on k, k+1
    bool m_collect = false;           // this call collects the admitted candidates and resolves them after the last block
    bool m_device_resolve = false;    // ... on the device: every block's admitted records are appended there as they come
    std::vector<std::vector<hc_admit_rec>> m_admitted;  // admitted candidates of the whole file, block by block, in sequence order
    // The device's copy of a block's admitted records leaves from a thread of its own: the in-order half of the collectors only
    // queues (pointer, count) — the staging copy and the three HIP calls of hc_graph_append took 0.25 ms of every block's turn.
    struct Appender;
    std::unique_ptr<Appender> m_appender;
    void start_appender();
    void finish_appender(bool rethrow);  // every queued append issued, the thread joined; its first error thrown if asked
};

}  // namespace hc
