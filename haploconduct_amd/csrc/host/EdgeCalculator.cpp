// EdgeCalculator.cpp — construct_edges / process_overlaps of the reference
// (src/EdgeCalculator.cpp:389-666) with the OpenMP scoring loop replaced by the HIP path
// behind include/hcedge.h.  The serial insert (duplicate resolution with the reference's
// tie-break chain) and the nonedge_overlaps.txt bookkeeping stay on the host.
#include "EdgeCalculator.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <sys/mman.h>
#include <cstring>
#include <type_traits>
#include <thread>

namespace hc {

static double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static void check(int status, const char* where) {
    if (status != HC_OK) throw FatalError{status, std::string(where) + ": " + hc_strerror(status) + " " + hc_last_error()};
}

EdgeCalculator::EdgeCalculator(std::shared_ptr<FastqStorage> fastq, std::shared_ptr<OverlapGraph> graph,
                               const ProgramSettings& ps)
    : program_settings(ps), fastq_storage(std::move(fastq)), overlap_graph(std::move(graph)) {
    if (ps.add_duplicates) throw FatalError{HC_ERR_ARG, "--add_duplicates is not supported (the pipelines never set it)"};
    if (const char* m = getenv("HC_INSERT_MODE")) m_serial_insert = std::string(m) == "serial";
    m_cs = to_hc_settings(ps);
    check(hc_create(&m_ctx, &m_cs), "hc_create");
    const FastqStorage& f = *fastq_storage;
    check(hc_set_reads(m_ctx, f.bases().data(), f.quals().data(), f.seq_off().data(), f.read_first_seq().data(),
                       f.get_readcount()),
          "hc_set_reads");
}

EdgeCalculator::~EdgeCalculator() {
    if (m_ctx) {
        hc_host_free(m_ctx, m_res);
        hc_host_free(m_ctx, m_idx);
        hc_destroy(m_ctx);
    }
}

double EdgeCalculator::phred_to_prob(int phred) const { return pow(10, -phred / 10.0); }  // :59-63

double EdgeCalculator::overlap_score(const std::string& seq1, const std::string& seq2, const std::string& score1,
                                     const std::string& score2, unsigned int pos, double& mismatch_rate) {
    // a scratch context holding the two strings as two single-end reads
    if (seq1.empty() || seq2.empty() || seq1.size() != score1.size() || seq2.size() != score2.size())
        throw FatalError{HC_ERR_ARG, "overlap_score: empty or mismatched strings"};
    hc_settings cs = m_cs;
    hc_ctx* ctx = nullptr;
    check(hc_create(&ctx, &cs), "hc_create");
    std::string bases = seq1 + seq2, quals = score1 + score2;
    const uint64_t off[3] = {0, seq1.size(), seq1.size() + seq2.size()};
    const uint32_t first[3] = {0, 1, 2};
    hc_overlap_rec rec;
    memset(&rec, 0, sizeof rec);
    rec.read1 = 0; rec.read2 = 1; rec.pos1 = pos; rec.ori1 = rec.ori2 = 1; rec.ord = '-';
    hc_result_rec res;
    int st = hc_set_reads(ctx, (const uint8_t*)bases.data(), (const uint8_t*)quals.data(), off, first, 2);
    if (st == HC_OK) st = hc_score_batch(ctx, &rec, 1, &res);
    hc_destroy(ctx);
    check(st, "overlap_score");
    double score;
    uint32_t cls;
    check(hc_finalize(&cs, &res, &score, &mismatch_rate, &cls), "hc_finalize");
    return score;
}

void EdgeCalculator::collect_read_info() {
    const FastqStorage& f = *fastq_storage;
    const std::vector<uint32_t>& first = f.read_first_seq();  // lengths straight from the flat layout (Read::get_seq_len goes there too)
    const std::vector<uint64_t>& off = f.seq_off();
    const size_t n = f.m_read_vec.size();
    m_read_info.resize(n);
    auto fill = [&](size_t b, size_t e) {
        for (size_t i = b; i < e; i++) {
            Read* r = f.m_read_vec[i];
            ReadInfo& x = m_read_info[i];
            const uint32_t q = first[i];
            x.read = r;
            x.paired = r->is_paired();
            x.len_a = (uint32_t)(off[q + 1] - off[q]);
            x.len_b = x.paired ? (uint32_t)(off[q + 2] - off[q + 1]) : 0;
            x.vertex_set = r->has_vertex_id(true);
            x.vertex = x.vertex_set ? r->get_vertex_id(true) : 0;
        }
    };
    const unsigned T = n < (1u << 16) ? 1u : std::max(1u, std::min<unsigned>(program_settings.n_threads, 8u));
    if (T == 1) {
        fill(0, n);
        return;
    }
    std::vector<std::thread> th;
    for (unsigned t = 1; t < T; t++) th.emplace_back(fill, n * t / T, n * (t + 1) / T);
    fill(0, n / T);
    for (auto& x : th) x.join();
}

// src/EdgeCalculator.cpp:395-414 (+ the Edge construction of compute_overlap): device scoring, then finalise and
// build on a few host threads; sequence order is kept by concatenating the threads' pieces in order.
void EdgeCalculator::score_and_build(const ParsedBatch& batch, BuiltBlock& out) {
    out.edges.clear();
    out.nonedge_text.clear();
    out.nonedges = 0;
    const size_t n = batch.size();
    if (n == 0) return;
    double t0 = now_s();
    if (n > m_cap) {
        hc_host_free(m_ctx, m_res);
        hc_host_free(m_ctx, m_idx);
        m_res = nullptr;
        m_idx = nullptr;
        m_cap = 0;
        const size_t cap = n + n / 8;
        check(hc_host_alloc(m_ctx, (void**)&m_res, cap * sizeof(hc_result_rec)), "hc_host_alloc");
        check(hc_host_alloc(m_ctx, (void**)&m_idx, cap * sizeof(uint32_t)), "hc_host_alloc");
        m_cap = cap;
    }
    // the omp-for of :395-414 on the device, straight from the (page-locked) records the parser wrote; only the
    // records that are not dropped come back
    const hc_overlap_rec* m_rec = batch.recs;
    uint64_t n_kept = 0;
    check(hc_score_batch_compact(m_ctx, m_rec, n, m_idx, m_res, m_cap, &n_kept), "hc_score_batch_compact");
    stats.scored += n;
    const double t_dev = now_s();
    if (program_settings.verbose) puts("build edges / write overlaps to file");

    if (m_read_info.size() != fastq_storage->m_read_vec.size()) collect_read_info();
    struct Piece {
        std::vector<Edge> edges;
        std::string nonedge_text;
        uint64_t nonedges = 0, ambiguous = 0;
        FatalError error{0, ""};
    };
    auto build = [&](uint64_t kb, uint64_t ke, Piece& pc) {
        char linebuf[192];
        pc.edges.reserve((size_t)(ke - kb));
        constexpr uint64_t kAhead = 12;  // the two rows of m_read_info are random places in a table of n_reads * 32 bytes
        for (uint64_t k = kb; k < ke; k++) {
            if (k + kAhead < ke) {
                const hc_overlap_rec& a = m_rec[m_idx[k + kAhead]];
                __builtin_prefetch(&m_read_info[a.read1]);
                __builtin_prefetch(&m_read_info[a.read2]);
            }
            const size_t i = m_idx[k];
            const hc_result_rec& r = m_res[k];
            uint32_t cls = HC_RES_CLS(r);
            if (cls == HC_CLS_DROP) continue;
            if (cls == HC_CLS_ERROR) {
                pc.error = FatalError{HC_ERR_DATA, "overlap " + batch.lines[i].get_overlap_line() + " touches an invalid base or quality byte"};
                return;
            }
            if (cls == HC_CLS_NONEDGE) {  // :410-413
                pc.nonedge_text.append(linebuf, batch.lines[i].write_line(linebuf));
                pc.nonedges++;
                continue;
            }
            double score, mismatch_rate;
            if (cls == HC_CLS_AMBIG) pc.ambiguous++;
            const int st = hc_finalize(&m_cs, &r, &score, &mismatch_rate, &cls);  // exp() with the host libm
            if (st != HC_OK) {
                pc.error = FatalError{st, "hc_finalize"};
                return;
            }
            if (cls == HC_CLS_DROP) continue;
            if (cls == HC_CLS_NONEDGE) {
                pc.nonedge_text.append(linebuf, batch.lines[i].write_line(linebuf));
                pc.nonedges++;
                continue;
            }
            // build the Edge as compute_overlap does, :219-232 / :254-270 / :292-308 / :353-379
            const hc_overlap_rec& o = m_rec[i];
            const ReadInfo& i1 = m_read_info[o.read1];
            const ReadInfo& i2 = m_read_info[o.read2];
            if (!i1.vertex_set || !i2.vertex_set) {
                pc.error = FatalError{HC_ERR_STATE, "Read::get_vertex_id: vertex id not set"};  // :180-183 asserts it
                return;
            }
            const bool p1 = i1.paired, p2 = i2.paired;
            const int pos1 = (int)o.pos1, pos2 = (int)o.pos2;
            int pos3, pos4 = 0;
            if (!p1 && !p2) {
                pos3 = (int)i1.len_a - pos1 - (int)i2.len_a;                 // :222
            } else if (!p1 && p2) {
                pos3 = (int)i1.len_a - pos2 - (int)i2.len_b;                 // :262
                pos4 = (int)i1.len_a - pos1 - (int)i2.len_a;                 // :263
            } else if (p1 && !p2) {
                pos3 = (int)i1.len_b + pos2 - (int)i2.len_a;                 // :300
                pos4 = (int)i2.len_a + pos1 - (int)i1.len_a;                 // :301
            } else {
                pos3 = o.ord == '1' ? (int)i1.len_b - pos2 - (int)i2.len_b   // :363
                                    : (int)i1.len_b + pos2 - (int)i2.len_b;  // :370
                pos4 = (int)i1.len_a - pos1 - (int)i2.len_a;                 // :372
            }
            Edge e(score, pos1, pos2, o.ori1 != 0, o.ori2 != 0, std::string(1, (char)o.ord), i1.read, i2.read);
            e.set_vertices(i1.vertex, i2.vertex);  // :180-183
            e.set_extra_pos(pos3, pos4);
            e.set_perc((int)o.perc);
            e.set_len((int)o.len1, (!p1 && !p2) ? 0 : (int)o.len2);  // :227 / :268
            e.set_mismatch(mismatch_rate);
            pc.edges.push_back(e);
        }
    };
    static const unsigned build_cap = getenv("HC_BUILD_THREADS") ? (unsigned)atoi(getenv("HC_BUILD_THREADS")) : 8u;  // experiment knob
    unsigned T = program_settings.n_threads > 1 ? std::min<unsigned>(program_settings.n_threads, std::max(1u, build_cap)) : 1;
    if (n_kept < 4096) T = 1;
    std::vector<Piece> pieces(T);
    if (T == 1) {
        build(0, n_kept, pieces[0]);
    } else {
        if (!m_build_pool || m_build_pool->workers() + 1 < T) m_build_pool.reset(new WorkerPool(T - 1));
        m_build_pool->run(T, [&](unsigned int t) {
            try {
                build(n_kept * t / T, n_kept * (t + 1) / T, pieces[t]);
            } catch (const FatalError& e) {  // Edge's own checks (src/Edge.h:43-57, :211-218); reported in sequence order below
                pieces[t].error = e;
            }
        });
    }
    const double t_built = now_s();
    size_t n_edges = 0;
    for (const Piece& pc : pieces) {
        if (pc.error.status) throw pc.error;  // the first one in sequence order
        n_edges += pc.edges.size();
    }
    out.edges.reserve(n_edges);
    for (Piece& pc : pieces) {
        out.edges.insert(out.edges.end(), pc.edges.begin(), pc.edges.end());
        out.nonedge_text += pc.nonedge_text;
        out.nonedges += pc.nonedges;
        stats.ambiguous += pc.ambiguous;
    }
    const double t1 = now_s();
    stats.t_score += t1 - t0;
    if (static const bool detail = getenv("HC_STAGE_TIMING") != nullptr; detail) {  // where the score stage spends its time
        static double dev = 0, build_s = 0, concat = 0;
        dev += t_dev - t0, build_s += t_built - t_dev, concat += t1 - t_built;
        fprintf(stderr, "[hc stage] score stage so far: device %.3f s, build %.3f s, concat %.3f s\n", dev, build_s, concat);
    }
}

// src/EdgeCalculator.cpp:431-555: the serial half
void EdgeCalculator::insert_block(BuiltBlock& blk) {
    const double t1 = now_s();
    const unsigned int dups_before = dup_count;
    const uint64_t added_before = stats.edges_added;
    stats.nonedges_written += blk.nonedges;
    if (m_sorted_insert) {
        m_admitted.emplace_back(std::move(blk.edges));  // resolved once, after the last block
        blk.edges = std::vector<Edge>();
    } else {
        // The second read of an edge is a random place in the graph's slot index and in the in-lists: ask for the
        // slot and the list header 2*kAhead edges early, and for the end of the list (its header is in cache by
        // then) kAhead edges early.
        constexpr size_t kAhead = 8;
        const size_t m = blk.edges.size();
        for (size_t k = 0; k < m; k++) {
            if (k + 2 * kAhead < m) {
                const Edge& a = blk.edges[k + 2 * kAhead];
                overlap_graph->prefetch_slot(a.get_vertex(1), a.get_vertex(2), a.get_ori(1) == a.get_ori(2));
                if (a.get_pos(1) == 0) overlap_graph->prefetch_slot(a.get_vertex(2), a.get_vertex(1), false);  // may be swapped, :443-448
            }
            if (k + kAhead < m) {
                const Edge& a = blk.edges[k + kAhead];
                overlap_graph->prefetch_in_list(a.get_vertex(2));
                if (a.get_pos(1) == 0) overlap_graph->prefetch_in_list(a.get_vertex(1));
            }
            InsertCounters ic;
            insert_edge(*overlap_graph, program_settings, blk.edges[k], ic);
            inclusion_count += ic.inclusion_count;
            dup_count += ic.dup_count;
            stats.edges_added += ic.edges_added;
        }
    }
    const double t2 = now_s();
    stats.t_insert += t2 - t1;
    if (program_settings.verbose && !m_sorted_insert) {
        printf("Number of edges found: %lu\n", (unsigned long)(stats.edges_added - added_before));
        printf("Number of duplicates: %u\n", dup_count - dups_before);
    }
    // :546-555 (the file is opened in append mode even when nothing is written)
    FILE* fo = fopen((program_settings.output_dir + "nonedge_overlaps.txt").c_str(), "a");
    if (fo) {
        fwrite(blk.nonedge_text.data(), 1, blk.nonedge_text.size(), fo);
        fclose(fo);
    }
    stats.t_write += now_s() - t2;
}

// src/EdgeCalculator.cpp:389-557
void EdgeCalculator::process_overlaps(const ParsedBatch& batch) {
    BuiltBlock blk;
    score_and_build(batch, blk);
    insert_block(blk);
}

// src/EdgeCalculator.cpp:561-666
void EdgeCalculator::construct_edges() {
    collect_read_info();  // vertex ids may have been assigned since the last call
    // An empty graph (every pipeline call) takes the bulk path: admitted edges are collected in sequence order and
    // resolved + filled in once after the last block (resolve_admitted_edges).  A graph that already holds edges,
    // or HC_INSERT_MODE=serial, takes the per-edge insert of the reference's serial half.
    m_sorted_insert = !m_serial_insert && overlap_graph->getEdgeCount() == 0 &&
                      EdgeSlotIndex::representable(overlap_graph->adj_out.size(), overlap_graph->adj_out.size());
    m_admitted.clear();
    std::remove("nonedge_overlaps.txt");  // :566 — in the cwd, whatever --output says (kept as is)
    std::vector<Overlap> rejected;
    OverlapsParser parser(program_settings.overlaps_file, program_settings, *fastq_storage);
    if (!parser.is_open()) throw FatalError{HC_ERR_IO, "Unable to open overlaps file"};  // :662-665
    if (program_settings.verbose) puts("reading overlaps file... ");
    size_t overlaps_per_vec = 250000;
    if (const char* e = getenv("HC_STAGE_BLOCK")) overlaps_per_vec = (size_t)strtoull(e, nullptr, 10);  // experiment knob  // the reference batches 1,000,000 (:571); batch boundaries do not influence the result
    // Three-stage pipeline: block k+1 is tokenised by the parser's worker threads while block k is scored on the
    // device and its edges are built, while the edges of block k-1 are inserted into the graph.  Every stage
    // consumes the blocks strictly in file order, so the graph, the counters and nonedge_overlaps.txt are those
    // of the sequential loop.
    ParsedBatch::RecStorage pinned;  // the parser writes its records where the device reads them
    pinned.ctx = m_ctx;
    pinned.alloc = [](void* ctx, size_t n) -> hc_overlap_rec* {
        void* p = nullptr;
        check(hc_host_alloc((hc_ctx*)ctx, &p, n * sizeof(hc_overlap_rec)), "hc_host_alloc");
        return (hc_overlap_rec*)p;
    };
    pinned.release = [](void* ctx, hc_overlap_rec* p) { hc_host_free((hc_ctx*)ctx, p); };
    ParsedBatch batch[2] = {ParsedBatch(pinned), ParsedBatch(pinned)};
    ParseCounters pc;
    bool more[2] = {false, false};
    FatalError parse_error{0, ""};
    bool parse_failed = false;
    auto parse_into = [&](int slot) {
        try {
            more[slot] = parser.next_batch(batch[slot], overlaps_per_vec, rejected, pc, /*print_malformed=*/true);
        } catch (const FatalError& e) {
            parse_failed = true;
            parse_error = e;
            more[slot] = false;
        }
    };
    BuiltBlock built[2];
    std::thread inserter;
    FatalError insert_error{0, ""};
    auto finish_insert = [&] {
        if (inserter.joinable()) inserter.join();
        if (insert_error.status) throw insert_error;
    };
    double t0 = now_s();
    parse_into(0);
    stats.t_parse += now_s() - t0;
    int cur = 0, slot = 0;
    while (more[cur]) {
        if (parse_failed) {
            if (inserter.joinable()) inserter.join();
            throw parse_error;
        }
        std::thread ahead(parse_into, cur ^ 1);
        try {
            if (!batch[cur].empty()) {  // :636-644
                score_and_build(batch[cur], built[slot]);
                finish_insert();  // block k-1 is in the graph
                BuiltBlock* blk = &built[slot];
                inserter = std::thread([this, blk, &insert_error] {
                    try {
                        insert_block(*blk);
                    } catch (const FatalError& e) {
                        insert_error = e;
                    } catch (const std::exception& e) {
                        insert_error = FatalError{HC_ERR_STATE, e.what()};
                    }
                });
                slot ^= 1;
            }
        } catch (...) {
            ahead.join();
            if (inserter.joinable()) inserter.join();
            throw;
        }
        const double t2 = now_s();
        ahead.join();
        stats.t_parse += now_s() - t2;  // only the part of the parse that was not hidden
        cur ^= 1;
    }
    finish_insert();
    if (parse_failed) throw parse_error;
    if (m_sorted_insert) {
        const double tr = now_s();
        InsertCounters ic;
        // one array in sequence order (the blocks are copied side by side by a few threads), resolved and filled in
        static_assert(std::is_trivially_copyable<Edge>::value, "Edge is copied with memcpy");
        std::vector<size_t> at(m_admitted.size() + 1, 0);
        for (size_t b = 0; b < m_admitted.size(); b++) at[b + 1] = at[b] + m_admitted[b].size();
        const size_t total = at.back();
        Edge* all = nullptr;
        if (total) {
            const size_t bytes = (total * sizeof(Edge) + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
            if (posix_memalign((void**)&all, (size_t)2 << 20, bytes) != 0) throw FatalError{HC_ERR_NOMEM, "construct_edges: out of memory"};
            madvise(all, bytes, MADV_HUGEPAGE);  // a hint: 2 MiB pages where the system grants them (fewer first-touch faults)
        }
        {
            const unsigned T = total < (1u << 16) ? 1u : std::max(1u, std::min<unsigned>(program_settings.n_threads, 16u));
            std::vector<std::thread> th;
            auto copy = [&](unsigned t) {
                for (size_t b = t; b < m_admitted.size(); b += T)
                    if (!m_admitted[b].empty()) memcpy((void*)(all + at[b]), (const void*)m_admitted[b].data(), m_admitted[b].size() * sizeof(Edge));
            };
            for (unsigned t = 1; t < T; t++) th.emplace_back(copy, t);
            copy(0);
            for (auto& x : th) x.join();
        }
        if (getenv("HC_STAGE_TIMING")) fprintf(stderr, "[hc stage] admitted edges side by side: %.3f s\n", now_s() - tr);
        try {
            resolve_admitted_edges(*overlap_graph, program_settings, all, total, ic);
        } catch (...) {
            free(all);
            throw;
        }
        free(all);
        inclusion_count += ic.inclusion_count;
        dup_count += ic.dup_count;
        stats.edges_added += ic.edges_added;
        if (program_settings.verbose) {  // totals instead of the reference's per-batch lines
            printf("Number of edges found: %lu\n", (unsigned long)ic.edges_added);
            printf("Number of duplicates: %u\n", ic.dup_count);
        }
        m_admitted.clear();
        m_admitted.shrink_to_fit();
        if (getenv("HC_STAGE_TIMING")) fprintf(stderr, "[hc stage] resolve total %.3f s\n", now_s() - tr);
        m_sorted_insert = false;
        stats.t_insert += now_s() - tr;
    }
    stats.lines_read = pc.lines_read;
    stats.malformed = pc.malformed;
    stats.self_overlaps = pc.self_overlaps;
    stats.prefilter_rejected = pc.prefilter_rejected;
    stats.silently_dropped = pc.silently_dropped;
    if (program_settings.verbose) {  // :646-649
        printf("Number of self-overlapping reads: %u\n", self_overlap_count);
        printf("Number of inclusion edges: %u\n", inclusion_count);
    }
    t0 = now_s();
    FILE* fo = fopen((program_settings.output_dir + "nonedge_overlaps.txt").c_str(), "a");  // :654-660
    if (fo) {
        char linebuf[192];
        std::string buf;
        for (const Overlap& o : rejected) buf.append(linebuf, o.write_line(linebuf));
        fwrite(buf.data(), 1, buf.size(), fo);
        fclose(fo);
    }
    stats.t_write += now_s() - t0;
}

}  // namespace hc
