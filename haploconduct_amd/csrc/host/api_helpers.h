// api_helpers.h — shared by hc_ec_api.cpp (device-backed stage) and hc_host_api.cpp (host-only pieces).
#pragma once
#include <cstring>
#include <memory>
#include <new>
#include <string>

#include "../../../include/hcedge_host.h"
#include "EdgeCalculator.h"

namespace hc {
int set_last_error(int status, const std::string& what);  // hc_api.cpp (or the sanitizer build's stub)
}

namespace {
using namespace hc;

inline ProgramSettings make_ps(const hc_settings* s, const hc_ec_paths* p) {
    ProgramSettings ps;
    ps.edge_threshold = s->edge_threshold;
    ps.ov_threshold = s->ov_threshold;
    ps.merge_contigs = s->merge_contigs;
    ps.mismatch = s->mismatch;
    ps.min_read_len = s->min_read_len;
    ps.min_overlap_len = s->min_overlap_len;
    ps.min_overlap_perc = s->min_overlap_perc;
    ps.add_duplicates = s->flags & HC_FLAG_ADD_DUPLICATES;
    ps.resolve_orientations = s->flags & HC_FLAG_RESOLVE_ORIENTATIONS;
    ps.ignore_inclusions = s->flags & HC_FLAG_IGNORE_INCLUSIONS;
    ps.relax_PE_edges = s->flags & HC_FLAG_RELAX_PE_EDGES;
    ps.allow_spaces = s->flags & HC_FLAG_ALLOW_SPACES;
    ps.verbose = s->flags & HC_FLAG_VERBOSE;
    ps.max_overlaps = s->max_overlaps;
    ps.n_threads = s->n_threads ? s->n_threads : 1;
    ps.device = s->device;
    ps.device_mask = s->device_mask;
    if (p) {
        auto str = [](const char* c) { return std::string(c ? c : ""); };
        ps.singles_file = str(p->singles_file);
        ps.paired1_file = str(p->paired1_file);
        ps.paired2_file = str(p->paired2_file);
        ps.id_correspondence = str(p->id_correspondence);
        ps.overlaps_file = str(p->overlaps_file);
        ps.output_dir = str(p->output_dir);
        if (p->max_reads) ps.max_reads = p->max_reads;
    }
    return ps;
}

template <typename F>
int guarded(const char* where, F&& f) {
    try {
        f();
        return HC_OK;
    } catch (const FatalError& e) {
        return set_last_error(e.status, std::string(where) + ": " + e.what);
    } catch (const std::bad_alloc&) {
        return set_last_error(HC_ERR_NOMEM, std::string(where) + ": out of memory");
    } catch (const std::exception& e) {
        return set_last_error(HC_ERR_FORMAT, std::string(where) + ": " + e.what());
    }
}

inline void fill_edge_rec(const Edge& e, hc_edge_rec& r) {
    memset(&r, 0, sizeof r);
    r.score = e.get_score();
    r.mismatch_rate = e.get_mismatch_rate();
    r.pos1 = e.get_pos(1);
    r.pos2 = e.get_pos(2);
    r.pos3 = e.get_extra_pos(1);
    r.pos4 = e.get_extra_pos(2);
    r.ori1 = e.get_ori(1);
    r.ori2 = e.get_ori(2);
    r.ord = (uint8_t)e.get_ord();
    r.read1 = e.get_read(1) ? e.get_read(1)->get_index() : 0;
    r.read2 = e.get_read(2) ? e.get_read(2)->get_index() : 0;
    r.v1 = e.get_vertex(1);
    r.v2 = e.get_vertex(2);
    r.perc = e.get_perc();
    r.len0 = e.get_len(0);
    r.len1 = e.get_len(1);
    r.len2 = e.get_len(2);
}

inline void dump_in_lists(const OverlapGraph& g, uint64_t* in_off, uint64_t* in_nodes, uint64_t cap) {
    uint64_t m = 0;
    for (size_t v = 0; v < g.adj_in.size(); v++) {
        in_off[v] = m;
        for (node_id_t w : g.adj_in[v]) {
            if (in_nodes && m < cap) in_nodes[m] = w;
            m++;
        }
    }
    in_off[g.adj_in.size()] = m;
}

inline uint64_t dump_edges(const OverlapGraph& g, hc_edge_rec* out, uint64_t cap) {
    uint64_t n = 0;
    for (const auto& L : g.adj_out)
        for (const Edge& e : L) {
            if (out && n < cap) fill_edge_rec(e, out[n]);
            n++;
        }
    return n;
}


}  // namespace
