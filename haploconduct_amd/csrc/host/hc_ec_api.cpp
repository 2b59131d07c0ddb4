// hc_ec_api.cpp — extern "C" face (include/hcedge_host.h) of the host-side stage.
#include <cstring>
#include <chrono>
#include <atomic>
#include <memory>
#include <vector>
#include <thread>

#include "../../../include/hcedge_host.h"
#include "EdgeCalculator.h"

#include <sys/stat.h>
#include "NumaBind.h"
#include "api_helpers.h"

namespace hc {
int set_last_error(int status, const std::string& what);  // hc_api.cpp
}

using namespace hc;

struct hc_ec {
    ProgramSettings ps;
    std::shared_ptr<FastqStorage> fastq;
    std::shared_ptr<OverlapGraph> graph;
    std::unique_ptr<EdgeCalculator> calc;
};

// The runtime loads a kernel's code at its first launch in a process: 12 ms in the text blocks' launch sequence, 11 ms in
// the graph kernels and their sorts — as long again as a whole C2-sized stage.  hc_ec_open runs the stage's device
// sequence once on two dummy reads, in a thread beside its FASTQ parsing (which is host work): the code is then in place
// when construct_edges wants it, and opening takes no longer.  Best effort: errors end the warm-up, not the open.
// Worth it when the FASTQ parsing beside it lasts longer than the extra work costs: on the SAVAGE example (2 ms of parsing) a process
// with the warm-up took 0.178 s, without 0.163; with 0.3 GB of FASTQ it saves 0.05 s.
bool hc::warm_up_pays(const ProgramSettings& ps) {
    if (const char* w = getenv("HC_WARM")) return atoi(w) != 0;
    uint64_t bytes = 0;
    for (const std::string* f : {&ps.singles_file, &ps.paired1_file, &ps.paired2_file}) {
        struct stat sb;
        if (!f->empty() && *f != "None" && stat(f->c_str(), &sb) == 0 && S_ISREG(sb.st_mode)) bytes += (uint64_t)sb.st_size;
    }
    return bytes >= ((uint64_t)64 << 20);
}

void hc::warm_device_code(hc_settings cs) noexcept try {
    hc_ctx* c = nullptr;
    if (hc_create(&c, &cs) != HC_OK) return;
    hc_textblock* tb = nullptr;
    hc_linechain* chain = nullptr;
    do {
        const uint32_t L = 64;
        std::vector<uint8_t> bases(2 * L), quals(2 * L, (uint8_t)'I');
        for (uint32_t i = 0; i < 2 * L; i++) bases[i] = (uint8_t)"ACGT"[(i * 7 + i / 5) & 3];
        const uint64_t seq_off[3] = {0, L, 2 * L};
        const uint32_t first[3] = {0, 1, 2};
        const uint64_t ids[2] = {0, 1};
        if (hc_set_reads(c, bases.data(), quals.data(), seq_off, first, 2) != HC_OK || hc_text_set_ids(c, ids, 2) != HC_OK) break;
        if (hc_textblock_create(c, 4096, &tb) != HC_OK || hc_linechain_create(c, 1, &chain) != HC_OK) break;
        char* buf = hc_textblock_buffer(tb);
        if (!buf) break;
        const char line[] = "0\t1\t8\t-\t-\t+\t+\t100\t-\t56\t-\ts\ts\n";
        memcpy(buf, line, sizeof line - 1);
        hc_text_result tr;
        if (hc_textblock_submit_from(tb, buf, sizeof line - 1, chain, 0, nullptr, 0) != HC_OK || hc_textblock_wait(tb, &tr) != HC_OK) break;
        hc_admit_rec a;
        memset(&a, 0, sizeof a);
        a.score = 0.99;
        a.read1 = 0;
        a.read2 = 1;
        a.pos1 = 8;
        a.n = 56;
        a.len1 = 56;
        a.perc = 90;
        a.ori1 = a.ori2 = 1;
        a.ord = (uint8_t)'-';
        hc_graph_counts gc;
        if (hc_graph_begin(c) != HC_OK || hc_graph_append(c, &a, 1) != HC_OK || hc_graph_resolve(c, nullptr, 1, 2, nullptr, HC_GRAPH_SORTED, &gc) != HC_OK) break;
        hc_edge_rec e[2];
        uint64_t oo[3], io[3];
        uint32_t in_nodes[2];
        uint8_t incl[2];
        (void)hc_graph_fetch(c, e, oo, in_nodes, io, nullptr, incl, nullptr);
    } while (false);
    hc_linechain_destroy(chain);
    hc_textblock_destroy(tb);
    hc_destroy(c);
} catch (...) {  // a thread body: nothing may leave it
}

extern "C" {

int hc_ec_open(hc_ec** out, const hc_settings* settings, const hc_ec_paths* paths) {
    if (!out || !settings || !paths) return set_last_error(HC_ERR_ARG, "hc_ec_open: null argument");
    *out = nullptr;
    std::unique_ptr<hc_ec> ec(new hc_ec());
    int rc = guarded("hc_ec_open", [&] {
        const bool timing = getenv("HC_STAGE_TIMING") != nullptr;
        auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double t0 = now();
        std::thread warm;  // once per process and device
        static std::atomic<uint64_t> warmed{0};
        const uint64_t dev_bit = 1ull << ((uint32_t)settings->device & 63u);
        const bool first_open = !(warmed.fetch_or(dev_bit) & dev_bit);
        ec->ps = make_ps(settings, paths);
        if (first_open && hc::warm_up_pays(ec->ps)) warm = std::thread(hc::warm_device_code, *settings);
        struct Join {
            std::thread& t;
            ~Join() {
                if (t.joinable()) t.join();
            }
        } join_warm{warm};
        // The FASTQ reader's threads (they inherit this thread's CPUs) fill the arrays the read store is uploaded from: next to
        // the device.  Not on a process's first open: asking where the device sits starts the HIP runtime, which that open
        // leaves to the warm-up thread beside the parsing (the stage's own threads find their place once it is up).
        hc::BoundForNow bound(first_open ? std::vector<int>() : hc::cpus_near_device(settings->device));
        ec->fastq = std::make_shared<FastqStorage>(ec->ps);                                   // ViralQuasispecies.cpp:233
        const double t1 = now();
        const unsigned int R = ec->fastq->get_readcount();
        ec->graph = std::make_shared<OverlapGraph>(ec->ps.add_duplicates ? 2 * R : R, ec->fastq, ec->ps);  // :246-261
        for (Read* r : ec->fastq->m_read_vec) r->set_vertex_id(true, ec->graph->addVertex(r->get_read_id()));  // :259-263
        if (ec->ps.add_duplicates)  // a vertex for every reverse-complemented read as well, :265-271
            for (Read* r : ec->fastq->m_read_vec) r->set_vertex_id(false, ec->graph->addVertex(r->get_read_id()));
        const double t2 = now();
        ec->calc.reset(new EdgeCalculator(ec->fastq, ec->graph, ec->ps));                     // :279
        if (timing) fprintf(stderr, "[hc stage] open: FASTQ -> FastqStorage %.3f s, graph vertices %.3f s, device contexts + store + text blocks %.3f s\n", t1 - t0, t2 - t1, now() - t2);
    });
    if (rc) return rc;
    *out = ec.release();
    return HC_OK;
}

int hc_ec_construct_edges(hc_ec* ec) {
    if (!ec) return set_last_error(HC_ERR_ARG, "hc_ec_construct_edges: null");
    return guarded("construct_edges", [&] { ec->calc->construct_edges(); });
}

int hc_ec_construct_edges_sorted(hc_ec* ec) {
    if (!ec) return set_last_error(HC_ERR_ARG, "hc_ec_construct_edges_sorted: null");
    const double t0 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    const int rc = guarded("construct_edges", [&] { ec->calc->construct_edges_sorted(); });
    if (getenv("HC_STAGE_TIMING"))
        fprintf(stderr, "[hc stage] hc_ec_construct_edges_sorted took %.3f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - t0);
    return rc;
}

int hc_ec_construct_edges_from_sfo(hc_ec* ec, const char* sfo_path, int sorted, uint64_t* n_records, uint64_t* n_lines, int* device_route) {
    if (!ec || !sfo_path) return set_last_error(HC_ERR_ARG, "hc_ec_construct_edges_from_sfo: null");
    return guarded("construct_edges", [&] { ec->calc->construct_edges_from_sfo(sfo_path, sorted != 0, n_records, n_lines, device_route); });
}

int hc_ec_construct_edges_from_store(hc_ec* ec, double err_rate, uint32_t min_overlap, uint32_t find_flags, int sorted, uint64_t* n_found,
                                     uint64_t* n_lines, int* device_route) {
    if (!ec) return set_last_error(HC_ERR_ARG, "hc_ec_construct_edges_from_store: null");
    return guarded("construct_edges", [&] { ec->calc->construct_edges_from_store(err_rate, min_overlap, find_flags, sorted != 0, n_found, n_lines, device_route); });
}

int hc_ec_construct_edges_from_reads(hc_ec* ec, double err_rate, uint32_t min_overlap, uint32_t find_flags, int sorted, uint64_t* n_found,
                                     uint64_t* n_lines) {
    if (!ec) return set_last_error(HC_ERR_ARG, "hc_ec_construct_edges_from_reads: null");
    return guarded("construct_edges", [&] { ec->calc->construct_edges_from_reads(err_rate, min_overlap, find_flags, sorted != 0, n_found, n_lines); });
}

uint32_t hc_ec_device_count(hc_ec* ec) { return ec ? ec->calc->device_count() : 0; }

int hc_ec_get_counters(hc_ec* ec, hc_ec_counters* c) {
    if (!ec || !c) return set_last_error(HC_ERR_ARG, "hc_ec_get_counters: null");
    memset(c, 0, sizeof *c);
    const auto& s = ec->calc->stats;
    c->self_overlap_count = ec->calc->self_overlap_count;
    c->inclusion_count = ec->calc->inclusion_count;
    c->dup_count = ec->calc->dup_count;
    c->edges_added = s.edges_added;
    c->nonedges_written = s.nonedges_written;
    c->prefilter_rejected = s.prefilter_rejected;
    c->malformed_lines = s.malformed;
    c->lines_read = s.lines_read;
    c->scored = s.scored;
    c->ambiguous = s.ambiguous;
    c->silently_dropped = s.silently_dropped;
    c->t_parse = s.t_parse;
    c->t_score = s.t_score;
    c->t_insert = s.t_insert;
    c->t_write = s.t_write;
    c->device_blocks = s.device_blocks;
    c->host_blocks = s.host_blocks;
    c->regrown_blocks = s.regrown_blocks;
    c->host_lines = s.host_lines;
    return HC_OK;
}

uint64_t hc_ec_read_count(hc_ec* ec) { return ec ? ec->fastq->get_readcount() : 0; }
uint64_t hc_ec_vertex_count(hc_ec* ec) { return ec ? ec->graph->adj_out.size() : 0; }
uint64_t hc_ec_edge_count(hc_ec* ec) { return ec ? ec->graph->getEdgeCount() : 0; }

int hc_ec_get_edges(hc_ec* ec, hc_edge_rec* out, uint64_t cap, uint64_t* n_out) {
    if (!ec || !n_out) return set_last_error(HC_ERR_ARG, "hc_ec_get_edges: null");
    *n_out = dump_edges(*ec->graph, out, cap);
    return HC_OK;
}

int hc_ec_get_inclusions(hc_ec* ec, uint8_t* out, uint64_t cap) {
    if (!ec || !out) return set_last_error(HC_ERR_ARG, "hc_ec_get_inclusions: null");
    const auto& inc = ec->graph->inclusions;
    for (uint64_t i = 0; i < cap && i < inc.size(); i++) out[i] = inc[i];
    return HC_OK;
}

int hc_ec_sort_edges(hc_ec* ec) {
    if (!ec) return set_last_error(HC_ERR_ARG, "hc_ec_sort_edges: null");
    return guarded("sortEdges", [&] {
        std::vector<uint32_t> len(ec->fastq->m_read_vec.size());
        for (size_t r = 0; r < len.size(); r++) len[r] = ec->fastq->m_read_vec[r]->get_len();
        ec->graph->sortEdges(len.data(), ec->ps.n_threads);
    });
}

int hc_ec_get_in_lists(hc_ec* ec, uint64_t* in_off, uint64_t* in_nodes, uint64_t cap) {
    if (!ec || !in_off) return set_last_error(HC_ERR_ARG, "hc_ec_get_in_lists: null");
    dump_in_lists(*ec->graph, in_off, in_nodes, cap);
    return HC_OK;
}

int hc_ec_overlap_score(hc_ec* ec, const char* seq1, const char* seq2, const char* phred1, const char* phred2,
                        uint32_t pos, double* score, double* mismatch_rate) {
    if (!ec || !seq1 || !seq2 || !phred1 || !phred2 || !score || !mismatch_rate)
        return set_last_error(HC_ERR_ARG, "hc_ec_overlap_score: null");
    return guarded("overlap_score", [&] { *score = ec->calc->overlap_score(seq1, seq2, phred1, phred2, pos, *mismatch_rate); });
}

int hc_ec_close(hc_ec* ec) {
    delete ec;
    return HC_OK;
}

int hc_ec_keep_devices(int on) {
    hc::keep_devices_resident(on != 0);
    return HC_OK;
}

}  // extern "C"
