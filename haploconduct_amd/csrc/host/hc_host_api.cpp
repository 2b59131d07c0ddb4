// hc_host_api.cpp — the host-only entry points of include/hcedge_host.h (no device needed): tokenizer,
// Overlap record parsing, FastqStorage, parser + prefilter, serial insert on a bare graph.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <thread>
#include <vector>

#include "api_helpers.h"

struct hc_fastq {
    ProgramSettings ps;
    std::shared_ptr<FastqStorage> fastq;
    std::vector<uint64_t> ids;
};

struct hc_host_graph {
    ProgramSettings ps;
    std::unique_ptr<OverlapGraph> graph;
    std::vector<Read> reads;
    InsertCounters counters;
};

namespace hc {
std::string sfo_to_overlaps(const char* sfo_text, size_t sfo_bytes, long ns, long np, uint64_t& n_lines);  // Sfo2Overlaps.cpp
std::string sfo_records_to_overlaps(const hc_sfo_rec* recs, uint64_t n, long ns, long np, uint64_t& n_lines);
}

extern "C" {

int hc_sfo2overlaps(const char* sfo_path, const char* out_path, uint64_t num_singles, uint64_t num_pairs, uint64_t* n_lines) {
    if (!sfo_path || !out_path) return set_last_error(HC_ERR_ARG, "hc_sfo2overlaps: null path");
    return guarded("sfo2overlaps", [&] {
        // the file mapped read-only (a pipe or another unmappable input is read whole)
        const int fd = open(sfo_path, O_RDONLY);
        if (fd < 0) throw FatalError{HC_ERR_IO, std::string("cannot open ") + sfo_path};
        struct stat stt;
        std::string owned;
        const char* text = nullptr;
        size_t bytes = 0;
        void* map = nullptr;
        if (fstat(fd, &stt) == 0 && S_ISREG(stt.st_mode) && stt.st_size > 0) {
            map = mmap(nullptr, (size_t)stt.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (map == MAP_FAILED) map = nullptr;
        }
        if (map) {
            text = (const char*)map;
            bytes = (size_t)stt.st_size;
        } else {
            char buf[1 << 16];
            ssize_t k;
            while ((k = read(fd, buf, sizeof buf)) > 0) owned.append(buf, (size_t)k);
            text = owned.data();
            bytes = owned.size();
        }
        struct Unmap {
            void* p; size_t n; int fd;
            ~Unmap() { if (p) munmap(p, n); close(fd); }
        } unmap{map, bytes, fd};
        uint64_t n = 0;
        const std::string out = hc::sfo_to_overlaps(text, bytes, (long)num_singles, (long)num_pairs, n);
        FILE* o = fopen(out_path, "wb");
        if (!o) throw FatalError{HC_ERR_IO, std::string("cannot write ") + out_path};
        fwrite(out.data(), 1, out.size(), o);
        fclose(o);
        if (n_lines) *n_lines = n;
    });
}

int hc_host_write_overlaps(const char* path, const hc_overlap_rec* recs, uint64_t n, const uint64_t* read_ids, const uint8_t* read_paired,
                           uint64_t n_reads, uint32_t n_threads) {
    if (!path || (n && (!recs || !read_ids || !read_paired))) return set_last_error(HC_ERR_ARG, "hc_host_write_overlaps: null");
    return guarded("write_overlaps", [&] {
        FILE* o = fopen(path, "wb");
        if (!o) throw FatalError{HC_ERR_IO, std::string("cannot write ") + path};
        unsigned T = n_threads ? n_threads : std::thread::hardware_concurrency();
        if (T == 0) T = 1;
        if (T > 64) T = 64;
        auto put = [](char* p, uint64_t v) {
            char tmp[24];
            int k = 0;
            do {
                tmp[k++] = (char)('0' + v % 10);
                v /= 10;
            } while (v);
            while (k) *p++ = tmp[--k];
            return p;
        };
        const uint64_t kBlock = 1u << 22;  // records per round: bounds the text held in memory
        std::vector<std::vector<char>> buf(T);
        bool bad = false;
        for (uint64_t base = 0; base < n && !bad; base += kBlock) {
            const uint64_t m = std::min<uint64_t>(kBlock, n - base);
            std::vector<std::thread> th;
            std::vector<int> err(T, 0);
            for (unsigned t = 0; t < T; t++)
                th.emplace_back([&, t] {
                    const uint64_t b = base + m * t / T, e = base + m * (t + 1) / T;
                    buf[t].resize((e - b) * 128);
                    char* p = buf[t].data();
                    for (uint64_t i = b; i < e; i++) {
                        const hc_overlap_rec& r = recs[i];
                        if (r.read1 >= n_reads || r.read2 >= n_reads) {
                            err[t] = 1;
                            break;
                        }
                        const bool p1 = read_paired[r.read1], p2 = read_paired[r.read2], ss = !p1 && !p2;
                        p = put(p, read_ids[r.read1]);
                        *p++ = '\t';
                        p = put(p, read_ids[r.read2]);
                        *p++ = '\t';
                        p = put(p, r.pos1);
                        *p++ = '\t';
                        if (ss) *p++ = '-';
                        else p = put(p, r.pos2);
                        *p++ = '\t';
                        *p++ = (char)r.ord;
                        *p++ = '\t';
                        *p++ = r.ori1 ? '+' : '-';
                        *p++ = '\t';
                        *p++ = r.ori2 ? '+' : '-';
                        *p++ = '\t';
                        p = put(p, r.perc);
                        *p++ = '\t';
                        if (ss) *p++ = '-';
                        else p = put(p, r.perc);
                        *p++ = '\t';
                        p = put(p, r.len1);
                        *p++ = '\t';
                        if (ss) *p++ = '-';
                        else p = put(p, r.len2);
                        *p++ = '\t';
                        *p++ = p1 ? 'p' : 's';
                        *p++ = '\t';
                        *p++ = p2 ? 'p' : 's';
                        *p++ = '\n';
                    }
                    buf[t].resize((size_t)(p - buf[t].data()));
                });
            for (auto& x : th) x.join();
            for (unsigned t = 0; t < T; t++) {
                if (err[t]) bad = true;
                if (!bad && !buf[t].empty() && fwrite(buf[t].data(), 1, buf[t].size(), o) != buf[t].size()) {
                    fclose(o);
                    throw FatalError{HC_ERR_IO, std::string("short write to ") + path};
                }
            }
        }
        if (fclose(o) != 0) throw FatalError{HC_ERR_IO, std::string("cannot close ") + path};
        if (bad) throw FatalError{HC_ERR_BAD_OVERLAP, "hc_host_write_overlaps: read index out of range"};
    });
}

int hc_sfo_records_to_overlaps(const hc_sfo_rec* recs, uint64_t n, const char* out_path, uint64_t num_singles, uint64_t num_pairs,
                               uint64_t* n_lines) {
    if (!out_path || (n && !recs)) return set_last_error(HC_ERR_ARG, "hc_sfo_records_to_overlaps: null");
    return guarded("sfo_records_to_overlaps", [&] {
        uint64_t k = 0;
        const std::string out = hc::sfo_records_to_overlaps(recs, n, (long)num_singles, (long)num_pairs, k);
        FILE* o = fopen(out_path, "wb");
        if (!o) throw FatalError{HC_ERR_IO, std::string("cannot write ") + out_path};
        const size_t w = out.empty() ? 0 : fwrite(out.data(), 1, out.size(), o);
        if (fclose(o) != 0 || w != out.size()) throw FatalError{HC_ERR_IO, std::string("short write to ") + out_path};
        if (n_lines) *n_lines = k;
    });
}

int hc_host_write_sfo(const char* path, const hc_sfo_rec* recs, uint64_t n) {
    if (!path || (n && !recs)) return set_last_error(HC_ERR_ARG, "hc_host_write_sfo: null");
    return guarded("write_sfo", [&] {
        FILE* o = fopen(path, "wb");
        if (!o) throw FatalError{HC_ERR_IO, std::string("cannot write ") + path};
        unsigned T = std::thread::hardware_concurrency();
        if (T == 0) T = 1;
        if (T > 32) T = 32;
        auto put = [](char* p, int64_t v) {  // "%d" / "%u"
            uint64_t u = v < 0 ? 0ull - (uint64_t)v : (uint64_t)v;
            if (v < 0) *p++ = '-';
            char tmp[24];
            int k = 0;
            do {
                tmp[k++] = (char)('0' + u % 10);
                u /= 10;
            } while (u);
            while (k) *p++ = tmp[--k];
            return p;
        };
        const uint64_t kBlock = 1u << 22;  // records per round: bounds the text held in memory
        std::vector<std::vector<char>> buf(T);
        for (uint64_t base = 0; base < n; base += kBlock) {
            const uint64_t m = std::min<uint64_t>(kBlock, n - base);
            const unsigned Tr = m < 65536 ? 1u : T;
            std::vector<std::thread> th;
            auto fill = [&](unsigned t) {
                const uint64_t b = base + m * t / Tr, e = base + m * (t + 1) / Tr;
                buf[t].resize((e - b) * 96);
                char* p = buf[t].data();
                for (uint64_t i = b; i < e; i++) {  // idA idB N|I OHA OHB OLA OLB K, tab-separated (scripts/sfo2overlaps.py:36)
                    const hc_sfo_rec& r = recs[i];
                    p = put(p, r.idA); *p++ = '\t';
                    p = put(p, r.idB); *p++ = '\t';
                    *p++ = r.inverted ? 'I' : 'N'; *p++ = '\t';
                    p = put(p, r.OHA); *p++ = '\t';
                    p = put(p, r.OHB); *p++ = '\t';
                    p = put(p, r.OLA); *p++ = '\t';
                    p = put(p, r.OLB); *p++ = '\t';
                    p = put(p, r.K); *p++ = '\n';
                }
                buf[t].resize((size_t)(p - buf[t].data()));
            };
            for (unsigned t = 1; t < Tr; t++) th.emplace_back(fill, t);
            fill(0);
            for (auto& x : th) x.join();
            for (unsigned t = 0; t < Tr; t++)
                if (!buf[t].empty() && fwrite(buf[t].data(), 1, buf[t].size(), o) != buf[t].size()) {
                    fclose(o);
                    throw FatalError{HC_ERR_IO, std::string("short write to ") + path};
                }
        }
        if (fclose(o) != 0) throw FatalError{HC_ERR_IO, std::string("cannot close ") + path};
    });
}

int hc_host_split_line(const char* line, uint64_t n, int allow_spaces, uint32_t* off, uint32_t* len, int max_fields) {
    if (!line || max_fields < 0 || max_fields > 64) return HC_ERR_ARG;
    const char* f[64];
    size_t l[64];
    const int nf = split_overlap_line(line, n, allow_spaces != 0, f, l, max_fields);
    for (int i = 0; i < nf && i < max_fields; i++) {
        if (off) off[i] = (uint32_t)(f[i] - line);
        if (len) len[i] = (uint32_t)l[i];
    }
    return nf;
}

int hc_host_parse_overlap(const char* line, uint64_t n, int allow_spaces, hc_overlap_fields* out, char* text) {
    if (!line || !out) return set_last_error(HC_ERR_ARG, "hc_host_parse_overlap: null");
    const char* f[14];
    size_t l[14];
    Overlap plain;
    // as the stage does: the one-pass reader for plain lines first, the reference's steps for every other line
    const bool is_plain = !(allow_spaces & 2) && Overlap::from_plain_line(line, n, plain);
    if (!is_plain && split_overlap_line(line, n, (allow_spaces & 1) != 0, f, l, 14) != 13) return HC_ERR_ARG;
    return guarded("Overlap", [&] {
        const Overlap o = is_plain ? plain : Overlap::from_fields(f, l);
        memset(out, 0, sizeof *out);
        out->id1 = o.m_id1; out->id2 = o.m_id2;
        out->pos1 = o.m_pos1; out->pos2 = o.m_pos2;
        out->perc1 = o.m_perc1; out->perc2 = o.m_perc2;
        out->len1 = o.m_len1; out->len2 = o.m_len2;
        out->perc = o.get_perc();
        out->ord = o.m_ord; out->ori1 = o.m_ori1; out->ori2 = o.m_ori2; out->type1 = o.m_type1; out->type2 = o.m_type2;
        if (text) {
            const size_t k = o.write_line(text);
            text[k] = 0;
        }
    });
}

int hc_host_fastq_load(hc_fastq** out, const hc_ec_paths* paths, hc_fastq_view* view) {
    if (!out || !paths || !view) return set_last_error(HC_ERR_ARG, "hc_host_fastq_load: null");
    *out = nullptr;
    std::unique_ptr<hc_fastq> f(new hc_fastq());
    hc_settings s;
    memset(&s, 0, sizeof s);
    s.max_overlaps = 100000000;
    int rc = guarded("FastqStorage", [&] {
        f->ps = make_ps(&s, paths);
        f->fastq = std::make_shared<FastqStorage>(f->ps);
        for (Read* r : f->fastq->m_read_vec) f->ids.push_back(r->get_read_id());
    });
    if (rc) return rc;
    const FastqStorage& q = *f->fastq;
    view->bases = q.bases().data();
    view->quals = q.quals().data();
    view->seq_off = q.seq_off().data();
    view->read_first_seq = q.read_first_seq().data();
    view->read_ids = f->ids.data();
    view->n_reads = q.get_readcount();
    view->n_seq = (uint32_t)(q.seq_off().size() - 1);
    view->n_single = q.m_readcount_single;
    view->n_paired = q.m_readcount_paired;
    *out = f.release();
    return HC_OK;
}

int hc_host_fastq_free(hc_fastq* f) {
    delete f;
    return HC_OK;
}

static int parse_with(const hc_settings* settings, hc_fastq* f, const char* overlaps_path, std::shared_ptr<const std::string> text, hc_overlap_rec* out,
                      uint64_t cap, uint64_t* n_out, hc_ec_counters* counters);

int hc_host_parse_file(const hc_settings* settings, hc_fastq* f, const char* overlaps_path, hc_overlap_rec* out,
                       uint64_t cap, uint64_t* n_out, hc_ec_counters* counters) {
    if (!settings || !f || !overlaps_path || !n_out) return set_last_error(HC_ERR_ARG, "hc_host_parse_file: null");
    return parse_with(settings, f, overlaps_path, nullptr, out, cap, n_out, counters);
}

int hc_host_parse_text(const hc_settings* settings, hc_fastq* f, const char* text, uint64_t n_bytes, hc_overlap_rec* out, uint64_t cap,
                       uint64_t* n_out, hc_ec_counters* counters) {
    if (!settings || !f || (n_bytes && !text) || !n_out) return set_last_error(HC_ERR_ARG, "hc_host_parse_text: null");
    return parse_with(settings, f, nullptr, std::make_shared<const std::string>(text ? text : "", (size_t)n_bytes), out, cap, n_out, counters);
}

static int parse_with(const hc_settings* settings, hc_fastq* f, const char* overlaps_path, std::shared_ptr<const std::string> text, hc_overlap_rec* out,
                      uint64_t cap, uint64_t* n_out, hc_ec_counters* counters) {
    *n_out = 0;
    return guarded("parse", [&] {
        ProgramSettings ps = make_ps(settings, nullptr);
        if (overlaps_path) ps.overlaps_file = overlaps_path;
        std::unique_ptr<OverlapsParser> owner(text ? new OverlapsParser(text, ps, *f->fastq) : new OverlapsParser(ps.overlaps_file, ps, *f->fastq));
        OverlapsParser& parser = *owner;
        if (!parser.is_open()) throw FatalError{HC_ERR_IO, "Unable to open overlaps file"};
        ParsedBatch batch;
        std::vector<Overlap> rejected;
        ParseCounters pc;
        uint64_t n = 0;
        for (;;) {
            const bool more = parser.next_batch(batch, 1000000, rejected, pc, false);
            if (!more) break;
            for (size_t i = 0; i < batch.size(); i++) {
                if (out && n < cap) out[n] = make_overlap_rec(batch.lines[i], batch.recs[i].read1, batch.recs[i].read2);
                n++;
            }
        }
        *n_out = n;
        if (counters) {
            memset(counters, 0, sizeof *counters);
            counters->lines_read = pc.lines_read;
            counters->malformed_lines = pc.malformed;
            counters->prefilter_rejected = pc.prefilter_rejected;
            counters->silently_dropped = pc.silently_dropped;
            counters->scored = n;
        }
    });
}

int hc_host_graph_new(hc_host_graph** out, uint64_t n_vertices, const hc_settings* settings) {
    if (!out || !settings) return set_last_error(HC_ERR_ARG, "hc_host_graph_new: null");
    std::unique_ptr<hc_host_graph> g(new hc_host_graph());
    g->ps = make_ps(settings, nullptr);
    g->graph.reset(new OverlapGraph((unsigned int)n_vertices, nullptr, g->ps));
    // --add_duplicates: the second half of the vertices are the reverse-complemented reads (ViralQuasispecies.cpp:246-271)
    const uint64_t n_reads = g->ps.add_duplicates ? n_vertices / 2 : n_vertices;
    g->reads.reserve(n_reads);
    for (uint64_t i = 0; i < n_reads; i++) {
        g->reads.emplace_back(nullptr, (unsigned int)i, false, (read_id_t)i);
        g->reads.back().set_vertex_id(true, g->graph->addVertex(i));
    }
    if (g->ps.add_duplicates)
        for (uint64_t i = 0; i < n_reads; i++) g->reads[i].set_vertex_id(false, g->graph->addVertex(i));
    *out = g.release();
    return HC_OK;
}

int hc_host_graph_adopt(hc_host_graph* g, const hc_edge_rec* edges, const uint64_t* out_off, const uint32_t* in_nodes, const uint64_t* in_off,
                        const uint8_t* inclusions) {
    if (!g || !out_off || !in_off) return set_last_error(HC_ERR_ARG, "hc_host_graph_adopt: null");
    return guarded("adopt_csr", [&] {
        std::vector<Read*> reads(g->reads.size());
        for (size_t i = 0; i < reads.size(); i++) reads[i] = &g->reads[i];
        g->graph->adopt_csr(edges, out_off, in_nodes, in_off, inclusions, reads.data(), reads.size(), g->ps.n_threads);
    });
}

int hc_host_graph_add_equivalent_edges(hc_host_graph* g) {
    if (!g) return set_last_error(HC_ERR_ARG, "hc_host_graph_add_equivalent_edges: null");
    return guarded("addEquivalentEdges", [&] { g->graph->addEquivalentEdges(); });
}

int hc_host_graph_insert(hc_host_graph* g, const hc_edge_rec* r) {
    if (!g || !r) return set_last_error(HC_ERR_ARG, "hc_host_graph_insert: null");
    return guarded("insert", [&] {
        if (r->read1 >= g->reads.size() || r->read2 >= g->reads.size()) throw FatalError{HC_ERR_ARG, "read index out of range"};
        Edge e(r->score, r->pos1, r->pos2, r->ori1 != 0, r->ori2 != 0, std::string(1, (char)r->ord), &g->reads[r->read1],
               &g->reads[r->read2]);
        e.set_vertices(r->v1, r->v2);
        e.set_extra_pos(r->pos3, r->pos4);
        e.set_perc(r->perc);
        e.set_len(r->len1, r->len2);
        e.set_mismatch(r->mismatch_rate);
        insert_edge(*g->graph, g->ps, e, g->counters);
    });
}

int hc_host_graph_resolve(hc_host_graph* g, const hc_edge_rec* edges, uint64_t n) {
    if (!g || (!edges && n)) return set_last_error(HC_ERR_ARG, "hc_host_graph_resolve: null");
    return guarded("resolve", [&] {
        if (g->graph->getEdgeCount() != 0) throw FatalError{HC_ERR_STATE, "hc_host_graph_resolve needs an empty graph"};
        std::vector<Edge> adm;
        adm.reserve(n);
        for (uint64_t i = 0; i < n; i++) {
            const hc_edge_rec* r = &edges[i];
            if (r->read1 >= g->reads.size() || r->read2 >= g->reads.size()) throw FatalError{HC_ERR_ARG, "read index out of range"};
            Edge e(r->score, r->pos1, r->pos2, r->ori1 != 0, r->ori2 != 0, std::string(1, (char)r->ord), &g->reads[r->read1],
                   &g->reads[r->read2]);
            e.set_vertices(r->v1, r->v2);
            e.set_extra_pos(r->pos3, r->pos4);
            e.set_perc(r->perc);
            e.set_len(r->len1, r->len2);
            e.set_mismatch(r->mismatch_rate);
            adm.push_back(e);
        }
        resolve_admitted_edges(*g->graph, g->ps, adm.data(), adm.size(), g->counters);
    });
}

int hc_host_graph_sort_edges(hc_host_graph* g, const uint32_t* len_by_read, uint64_t n_reads) {
    if (!g || !len_by_read) return set_last_error(HC_ERR_ARG, "hc_host_graph_sort_edges: null");
    if (n_reads != g->reads.size()) return set_last_error(HC_ERR_ARG, "hc_host_graph_sort_edges: one length per read");
    return guarded("sortEdges", [&] { g->graph->sortEdges(len_by_read, g->ps.n_threads); });
}

int hc_host_graph_get_in_lists(hc_host_graph* g, uint64_t* in_off, uint64_t* in_nodes, uint64_t cap) {
    if (!g || !in_off) return set_last_error(HC_ERR_ARG, "hc_host_graph_get_in_lists: null");
    dump_in_lists(*g->graph, in_off, in_nodes, cap);
    return HC_OK;
}

int hc_host_graph_get(hc_host_graph* g, hc_edge_rec* out, uint64_t cap, uint64_t* n_out, uint8_t* inclusions,
                      hc_ec_counters* counters) {
    if (!g || !n_out) return set_last_error(HC_ERR_ARG, "hc_host_graph_get: null");
    *n_out = dump_edges(*g->graph, out, cap);
    if (inclusions)
        for (size_t i = 0; i < g->graph->inclusions.size(); i++) inclusions[i] = g->graph->inclusions[i];
    if (counters) {
        memset(counters, 0, sizeof *counters);
        counters->inclusion_count = g->counters.inclusion_count;
        counters->dup_count = g->counters.dup_count;
        counters->edges_added = g->counters.edges_added;
    }
    return HC_OK;
}

int hc_host_graph_free(hc_host_graph* g) {
    delete g;
    return HC_OK;
}

}  // extern "C"
