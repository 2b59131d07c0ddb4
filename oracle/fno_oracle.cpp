// fno_oracle.cpp — TEST INFRASTRUCTURE, not product code.
//
// Sequential CPU restatement of the reference's "find next overlaps" step, written from a reading of
//   src/FindNextOverlaps.cpp   (FNO=1)  and  src/FindNextOverlaps3.cpp  (FNO=3)
// with the same containers the reference uses where their ordering is observable
// (std::set<std::string> for the output of FNO=1, std::unordered_map<unsigned long, unsigned long> for the
// walk order of FNO=3 — that order is a property of the host's libstdc++ in the reference too).
// Only tests/ may load this.  Inputs/outputs use the flat records of include/hcfno.h.
//
// PINNING: the reference's FNO translation units need Boost (absent here) and cannot be built.  Almost all of their
// text is Boost-free, though, and is compiled as a FRAGMENT PROBE (oracle/Makefile `ref`, _ref/libhcref_fno.so) behind
// build-owned class shells: FindNextOverlaps.cpp:25-631 and :816-958 (updateOverlap, findCliqueIndex, computeOverlapData,
// processOverlaps, reconsiderEdgeOverlaps, findInclusionOverlaps, findNextOverlaps), OverlapGraph.cpp:233-259 (checkEdge)
// and FindNextOverlaps3.cpp:20-406 (the whole of FNO=3).  This file is checked against it on the committed vectors
// tests/golden/fno/*.json: whole findNextOverlaps() and findNextOverlaps3() runs (the files they write), updateOverlap
// sequences, computeOverlapData and deduceOverlap calls.  reconsiderNonedgeOverlaps (FindNextOverlaps.cpp:635-813) is in the
// probe as well, around its one Boost call (boost::trim_if at :652, replaced by a build-owned statement): the stored non-edges,
// the checkEdge filter in front of them (:702) and the opposite overlaps of --add_duplicates (:699-793) are pinned by
// fno1_run_nonedges.json and fno1_run_add_duplicates.json.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <deque>
#include <list>
#include <map>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

#include "../include/hcfno.h"

namespace {

struct RefAbort {
    std::string why;
};
#define REF_ASSERT(c) \
    do {              \
        if (!(c)) throw RefAbort{"reference assert: " #c}; \
    } while (0)

struct Rd {  // the facts FNO reads of a Read
    unsigned long id;
    int len1, len2;
    bool paired;
    unsigned int get_len() const { return paired ? (unsigned)len1 + (unsigned)len2 : (unsigned)len1; }
};

Rd rd_of(const hc_fno_read& r) { return Rd{(unsigned long)r.id, (int)r.len1, (int)r.len2, r.paired != 0}; }

struct Sr {
    Rd rd;
    std::unordered_map<unsigned long, hc_fno_subread> sub;
};

struct Ed {
    unsigned long v1, v2;
    double score;
    int pos1, pos2, len1, len2, perc;
    char ord;
    bool ori1, ori2;
};

Ed ed_of(const hc_fno_edge& e) {
    return Ed{(unsigned long)e.v1, (unsigned long)e.v2, e.score, e.pos1, e.pos2, e.len1, e.len2, e.perc, (char)e.ord, e.ori1 != 0, e.ori2 != 0};
}

// FindNextOverlaps.cpp:331-347
int find_clique_index(unsigned long node, const Sr& sr, bool leftside, bool second_occ) {
    REF_ASSERT(!sr.sub.empty());
    auto it = sr.sub.find(node);
    if (it == sr.sub.end()) throw RefAbort{"subreadMap.at(node): node not in super-read"};
    const hc_fno_subread& s = it->second;
    if (leftside && !second_occ) {
        REF_ASSERT(s.index1 >= 0 && s.startpos1 >= 0);
        REF_ASSERT(!(s.index1 > 0 && s.startpos1 > 0));
        return s.index1 - s.startpos1;
    }
    REF_ASSERT(leftside || !second_occ);
    REF_ASSERT(s.index2 >= 0 && s.startpos2 >= 0);
    if (sr.rd.paired) REF_ASSERT(!(s.index2 > 0 && s.startpos2 > 0));
    return s.index2 - s.startpos2;
}

int perc_of(int ov, int la, int lb) {  // (int)floor(std::max(ov/float(la), ov/float(lb))*100)
    const float a = ov / float(la), b = ov / float(lb);
    const float m = std::max(a, b) * 100;
    return (int)floorf(m);
}

// FindNextOverlaps.cpp:351-565
bool compute_overlap_data(const Rd& s1, const Rd& s2, int idx1l, int idx1r, int idx2l, int idx2r, const Ed& edge, int& new_pos1,
                          int& new_pos2, char& ord1, char& ord2, char& type1, char& type2, int& overlap_perc, int& overlap_len1,
                          int& overlap_len2) {
    const int pos1 = edge.pos1, pos2 = edge.pos2;
    if (!s1.paired && !s2.paired) {
        type1 = 's';
        type2 = 's';
        new_pos1 = (pos1 + idx1l) - idx2l;
        const int len1 = s1.len1, len2 = s2.len1;
        int len;
        REF_ASSERT(len1 > 0 && len2 > 0);
        if (new_pos1 < 0) {
            ord1 = '2';
            new_pos1 = -new_pos1;
            len = len2;
        } else {
            ord1 = '1';
            len = len1;
        }
        overlap_len1 = std::min(std::min(len - new_pos1, len1), len2);
        overlap_len2 = 0;
        overlap_perc = perc_of(overlap_len1, len1, len2);
        ord2 = '-';
        new_pos2 = 0;
        if (new_pos1 >= len) return false;
    } else if (s1.paired && !s2.paired) {
        type1 = 'p';
        type2 = 's';
        const int len1 = s1.len1 + s1.len2, len2 = s2.len1;
        REF_ASSERT(len1 > 0 && len2 > 0);
        new_pos1 = (pos1 + idx1l) - idx2l;
        if (new_pos1 < 0) {
            ord1 = '2';
            new_pos1 = -new_pos1;
            if (new_pos1 >= s2.len1) return false;
            overlap_len1 = s1.len1;
        } else {
            ord1 = '1';
            if (new_pos1 >= s1.len1) return false;
            overlap_len1 = s1.len1 - new_pos1;
        }
        if (edge.ord == '1') new_pos2 = idx2r - (idx1r + pos2);
        else new_pos2 = (pos2 + idx2r) - idx1r;
        if (new_pos2 >= s2.len1) return false;
        if (new_pos2 < 0) return false;
        ord2 = '-';
        overlap_len2 = (int)std::min((size_t)s2.len1 - (size_t)new_pos2, (size_t)s1.len2);
        const int total = overlap_len1 + overlap_len2;
        overlap_perc = std::min(perc_of(total, len1, len2), 100);
    } else if (!s1.paired && s2.paired) {
        type1 = 's';
        type2 = 'p';
        const int len1 = s1.len1, len2 = s2.len1 + s2.len2;
        REF_ASSERT(len1 > 0 && len2 > 0);
        new_pos1 = pos1 + idx1l - idx2l;
        if (new_pos1 < 0) {
            ord1 = '2';
            new_pos1 = -new_pos1;
            if (new_pos1 >= s2.len1) return false;
            overlap_len1 = s2.len1 - new_pos1;
        } else {
            ord1 = '1';
            if (new_pos1 >= s1.len1) return false;
            overlap_len1 = s2.len1;
        }
        if (edge.ord == '2') new_pos2 = idx1r - (pos2 + idx2r);
        else new_pos2 = idx1r + pos2 - idx2r;
        if (new_pos2 >= s1.len1) return false;
        if (new_pos2 < 0) return false;
        ord2 = '-';
        overlap_len2 = (int)std::min((size_t)s1.len1 - (size_t)new_pos2, (size_t)s2.len2);
        const int total = overlap_len1 + overlap_len2;
        overlap_perc = std::min(perc_of(total, len1, len2), 100);
    } else {
        type1 = 'p';
        type2 = 'p';
        new_pos1 = (pos1 + idx1l) - idx2l;
        if (new_pos1 < 0) {
            ord1 = '2';
            new_pos1 = -new_pos1;
            if (new_pos1 >= s2.len1) return false;
            overlap_len1 = (int)std::min((size_t)s1.len1, (size_t)s2.len1 - (size_t)new_pos1);
        } else {
            ord1 = '1';
            if (new_pos1 >= s1.len1) return false;
            overlap_len1 = (int)std::min((size_t)s1.len1 - (size_t)new_pos1, (size_t)s2.len1);
        }
        if (edge.ord == '1') new_pos2 = (pos2 + idx1r) - idx2r;
        else new_pos2 = idx1r - (pos2 + idx2r);
        if (new_pos2 < 0) {
            ord2 = ord1 == '1' ? '2' : '1';
            new_pos2 = -new_pos2;
            if (new_pos2 >= s2.len2) return false;
            overlap_len2 = (int)std::min((size_t)s1.len2, (size_t)s2.len2 - (size_t)new_pos2);
        } else {
            ord2 = ord1 == '1' ? '1' : '2';
            if (new_pos2 >= s1.len2) return false;
            overlap_len2 = (int)std::min((size_t)s1.len2 - (size_t)new_pos2, (size_t)s2.len2);
        }
        const int total = overlap_len1 + overlap_len2;
        overlap_perc = std::min(perc_of(total, s1.len1 + s1.len2, s2.len1 + s2.len2), 100);
    }
    REF_ASSERT(new_pos1 >= 0);
    REF_ASSERT(new_pos2 >= 0);
    REF_ASSERT(overlap_perc >= 0 && overlap_perc <= 100);
    return true;
}

struct Fno1 {
    const hc_fno1_input* in;
    std::vector<Rd> nodes;
    std::deque<Sr> srs;
    std::vector<std::vector<const Sr*>> nodes_to_SR;
    std::vector<std::set<unsigned long>> overlaps_found;
    std::vector<std::list<Ed>> adj_out;
    std::set<std::string> lines;
    unsigned long copied = 0, u2sr = 0, v2sr = 0, sr2sr = 0;

    bool visited(unsigned long u) const {
        if (u >= in->n_nodes) throw RefAbort{"vertex out of range"};
        return in->nodes[u].visited != 0;
    }
    unsigned long new_id(unsigned long u) const { return (unsigned long)in->nodes[u].id; }

    void indices(unsigned long node, const Sr& sr, bool read_paired, int& l, int& r) {
        if (sr.rd.paired) {
            l = find_clique_index(node, sr, true, false);
            r = find_clique_index(node, sr, false, false);
        } else if (read_paired) {
            l = find_clique_index(node, sr, true, false);
            r = find_clique_index(node, sr, true, true);
        } else {
            l = find_clique_index(node, sr, true, false);
            r = l;
        }
    }

    bool already_found(unsigned long id1, unsigned long id2) {
        const unsigned long smallest = std::min(id1, id2), largest = std::max(id1, id2);
        if (smallest >= overlaps_found.size()) throw RefAbort{"overlaps_found.at(): id >= new_read_count"};
        if (overlaps_found[smallest].count(largest)) return true;
        overlaps_found[smallest].insert(largest);
        return false;
    }

    std::string tail(int pos1, int pos2, char ord, const std::string& ori1, const std::string& ori2, int perc, int len1, int len2, char t1,
                     char t2) {
        std::string s = std::to_string(pos1) + "\t" + std::to_string(pos2) + "\t";
        REF_ASSERT(ord == '-' || ord == '1' || ord == '2');
        s += ord;
        s += "\t" + ori1 + "\t" + ori2 + "\t" + std::to_string(perc) + "\t0\t" + std::to_string(len1) + "\t" + std::to_string(len2) + "\t";
        s += t1;
        s += "\t";
        s += t2;
        return s;
    }

    // FindNextOverlaps.cpp:25-327
    void update_overlap(const Ed& e) {
        const unsigned long u = e.v1, v = e.v2;
        const bool vu = visited(u), vv = visited(v);
        const Rd &read1 = nodes[u], &read2 = nodes[v];
        std::string ori1 = "+", ori2 = "+";
        if ((in->flags & HC_FNO_RESOLVE_ORIENTATIONS) && e.score == 0) {
            ori1 = (e.ori1 == (in->nodes[u].orientation != 0)) ? "+" : "-";
            ori2 = (e.ori2 == (in->nodes[v].orientation != 0)) ? "+" : "-";
        }
        const bool no_incl = in->flags & HC_FNO_NO_INCLUSIONS;
        int pos1, pos2, perc, l1, l2;
        char ord1, ord2, type1, type2;
        if (!vu && !vv) {
            REF_ASSERT(e.perc >= 0);
            std::string line = std::to_string(new_id(u)) + "\t" + std::to_string(new_id(v)) + "\t";
            line += tail(e.pos1, e.pos2, e.ord, ori1, ori2, e.perc, e.len1, e.len2, read1.paired ? 'p' : 's', read2.paired ? 'p' : 's');
            if (!(no_incl && e.perc == 100)) {
                lines.insert(line);
                copied++;
            }
        } else if (!vu) {
            const unsigned long id1 = new_id(u);
            for (const Sr* sr : nodes_to_SR.at(v)) {
                const unsigned long id2 = sr->rd.id;
                REF_ASSERT(id1 != id2);
                if (already_found(id1, id2)) continue;
                int idx2l, idx2r;
                indices(v, *sr, read2.paired, idx2l, idx2r);
                if (!compute_overlap_data(read1, sr->rd, 0, 0, idx2l, idx2r, e, pos1, pos2, ord1, ord2, type1, type2, perc, l1, l2)) continue;
                std::string line;
                char t1, t2;
                if (ord1 == '1') {
                    line = std::to_string(id1) + "\t" + std::to_string(id2) + "\t";
                    t1 = type1;
                    t2 = type2;
                } else {
                    line = std::to_string(id2) + "\t" + std::to_string(id1) + "\t";
                    t1 = type2;
                    t2 = type1;
                }
                line += tail(pos1, pos2, ord2, ori1, ori2, perc, l1, l2, t1, t2);
                if (!(no_incl && perc == 100)) {
                    lines.insert(line);
                    u2sr++;
                }
            }
        } else if (!vv) {
            const unsigned long id1 = new_id(v);
            for (const Sr* sr : nodes_to_SR.at(u)) {
                const unsigned long id2 = sr->rd.id;
                REF_ASSERT(id1 != id2);
                if (already_found(id1, id2)) continue;
                int idx1l, idx1r;
                indices(u, *sr, read1.paired, idx1l, idx1r);
                if (!compute_overlap_data(sr->rd, read2, idx1l, idx1r, 0, 0, e, pos1, pos2, ord1, ord2, type1, type2, perc, l1, l2)) continue;
                std::string line;
                char t1, t2;
                if (ord1 == '1') {
                    line = std::to_string(id2) + "\t" + std::to_string(id1) + "\t";
                    t1 = type1;
                    t2 = type2;
                } else {
                    line = std::to_string(id1) + "\t" + std::to_string(id2) + "\t";
                    t1 = type2;
                    t2 = type1;
                }
                line += tail(pos1, pos2, ord2, ori1, ori2, perc, l1, l2, t1, t2);
                if (!(no_incl && perc == 100)) {
                    lines.insert(line);
                    v2sr++;
                }
            }
        } else {
            for (const Sr* sr1 : nodes_to_SR.at(u)) {
                const unsigned long id1 = sr1->rd.id;
                int idx1l, idx1r;
                indices(u, *sr1, read1.paired, idx1l, idx1r);
                for (const Sr* sr2 : nodes_to_SR.at(v)) {
                    const unsigned long id2 = sr2->rd.id;
                    if (id1 == id2) continue;
                    if (already_found(id1, id2)) continue;
                    int idx2l, idx2r;
                    indices(v, *sr2, read2.paired, idx2l, idx2r);
                    if (!compute_overlap_data(sr1->rd, sr2->rd, idx1l, idx1r, idx2l, idx2r, e, pos1, pos2, ord1, ord2, type1, type2, perc, l1, l2))
                        continue;
                    std::string line;
                    char t1, t2;
                    if (ord1 == '1') {
                        line = std::to_string(id1) + "\t" + std::to_string(id2) + "\t";
                        t1 = type1;
                        t2 = type2;
                    } else {
                        line = std::to_string(id2) + "\t" + std::to_string(id1) + "\t";
                        t1 = type2;
                        t2 = type1;
                    }
                    line += tail(pos1, pos2, ord2, ori1, ori2, perc, l1, l2, t1, t2);
                    if (!(no_incl && perc == 100)) {
                        lines.insert(line);
                        sr2sr++;
                    }
                }
            }
        }
    }

    // OverlapGraph::checkEdge (src/OverlapGraph.cpp:233-259)
    double check_edge(unsigned long v, unsigned long w, bool reverse_allowed) {
        for (const Ed& e : adj_out.at(v))
            if (e.v2 == w) return e.score;
        if (reverse_allowed)
            for (const Ed& e : adj_out.at(w))
                if (e.v2 == v) return e.score;
        return -1;
    }

    void run() {
        for (uint64_t i = 1; i < in->n_srs; ++i)  // single_SR_vec, then paired_SR_vec (:893-906)
            if (in->srs[i - 1].paired && !in->srs[i].paired) throw RefAbort{"input contract: super-reads single-end first, then paired"};
        for (uint64_t i = 0; i < in->n_nodes; ++i) nodes.push_back(rd_of(in->nodes[i]));
        for (uint64_t i = 0; i < in->n_srs; ++i) {
            Sr s;
            s.rd = rd_of(in->srs[i]);
            for (uint64_t k = in->subread_off[i]; k < in->subread_off[i + 1]; ++k) s.sub[(unsigned long)in->subreads[k].node] = in->subreads[k];
            srs.push_back(std::move(s));
        }
        // :891-906
        overlaps_found.assign(in->new_read_count, std::set<unsigned long>());
        nodes_to_SR.assign(in->n_nodes, std::vector<const Sr*>());
        for (uint64_t i = 0; i < in->n_srs; ++i) {
            REF_ASSERT(in->clique_off[i + 1] > in->clique_off[i]);
            for (uint64_t k = in->clique_off[i]; k < in->clique_off[i + 1]; ++k) nodes_to_SR.at(in->clique_nodes[k]).push_back(&srs[i]);
        }
        adj_out.assign(in->n_nodes, std::list<Ed>());
        for (uint64_t i = 0; i < in->n_graph_edges; ++i) adj_out.at(in->graph_edges[i].v1).push_back(ed_of(in->graph_edges[i]));
        // reconsiderEdgeOverlaps :605-631
        for (uint64_t i = 0; i < in->n_graph_edges; ++i) update_overlap(ed_of(in->graph_edges[i]));
        for (uint64_t i = 0; i < in->n_branching_edges; ++i) update_overlap(ed_of(in->branching_edges[i]));
        // reconsiderNonedgeOverlaps :635-813.  With --add_duplicates the graph has a vertex per read and strand (vertex r: read r,
        // vertex r + n_nodes / 2: its reverse complement, src/ViralQuasispecies.cpp:246-270), the line's vertices are taken by its
        // orientations (:672-675) and every line that passes :702 is followed by the same overlap seen from the other strand (:699-793).
        if (!(in->flags & HC_FNO_OPTIMIZE)) {
            const bool dup = (in->flags & HC_FNO_ADD_DUPLICATES) != 0;
            const unsigned long half = in->n_nodes / 2;
            if (dup) {
                if (in->n_nodes % 2) throw RefAbort{"input contract: --add_duplicates needs every read on both strands"};
                for (unsigned long r = 0; r < half; ++r)
                    if (nodes[r].len1 != nodes[r + half].len1 || nodes[r].len2 != nodes[r + half].len2 || nodes[r].paired != nodes[r + half].paired)
                        throw RefAbort{"input contract: vertices r and r + n_nodes / 2 are one read"};
            }
            auto other_strand = [&](unsigned long v) { return v < half ? v + half : v - half; };  // Read::get_vertex_id(!ori)
            for (uint64_t i = 0; i < in->n_nonedges; ++i) {
                Ed e = ed_of(in->nonedges[i]);
                e.score = 0;
                REF_ASSERT(e.len1 > 0);   // Edge::set_len
                REF_ASSERT(e.len2 >= 0);
                if (dup && ((e.v1 < half) != e.ori1 || (e.v2 < half) != e.ori2))
                    throw RefAbort{"input contract: under --add_duplicates a line's vertex is its read's vertex on the strand the orientation names (:672-675)"};
                if (check_edge(e.v1, e.v2, true) > 0) continue;
                update_overlap(e);
                if (!dup) continue;
                const Rd &read1 = nodes.at(e.v1), &read2 = nodes.at(e.v2);
                const unsigned long v1 = other_strand(e.v1), v2 = other_strand(e.v2);  // :700-701
                // the reference's `int pos = size() - get_pos() - size()`: size_t arithmetic, then the conversion to int
                const size_t a1 = (size_t)read1.len1, a2 = (size_t)read1.len2, b1 = (size_t)read2.len1, b2 = (size_t)read2.len2;
                const size_t p1 = (size_t)(long)e.pos1, p2 = (size_t)(long)e.pos2;
                int pos1, pos2 = 0;
                char ord = e.ord;
                if (!read1.paired && !read2.paired) {  // S-S :702-717
                    pos1 = (int)(a1 - p1 - b1);
                } else if (read1.paired && !read2.paired) {  // P-S :718-735
                    pos1 = (int)(a2 + p2 - b1);
                    pos2 = (int)(b1 + p1 - a1);
                } else if (!read1.paired && read2.paired) {  // S-P :736-753
                    pos1 = (int)(a1 - p2 - b2);
                    pos2 = (int)(a1 - p1 - b1);
                } else {  // P-P :754-792
                    if (e.ord == '1') {
                        pos1 = (int)(a2 - p2 - b2);
                    } else {
                        REF_ASSERT(e.ord == '2');
                        pos1 = (int)(a2 + p2 - b2);
                    }
                    pos2 = (int)(a1 - p1 - b1);
                    if (pos1 < 0) {
                        if (pos2 < 0) {
                            pos2 = -pos2;
                            ord = '1';
                        } else {
                            ord = '2';
                        }
                    } else {
                        if (pos2 < 0) {
                            pos2 = -pos2;
                            ord = '2';
                        } else {
                            ord = '1';
                        }
                    }
                }
                Ed opp;
                if (pos1 < 0) opp = Ed{v2, v1, 0.0, -pos1, pos2, e.len1, e.len2, e.perc, ord, !e.ori2, !e.ori1};
                else opp = Ed{v1, v2, 0.0, pos1, pos2, e.len1, e.len2, e.perc, ord, !e.ori1, !e.ori2};
                update_overlap(opp);
            }
        }
        // findInclusionOverlaps :816-883
        std::vector<Ed> edge_vec;
        for (uint64_t g = 0; g < in->n_inclusion_groups; ++g) {
            const uint64_t b = in->inclusion_off[g];
            const unsigned l = (unsigned)(in->inclusion_off[g + 1] - b);
            for (unsigned i = 0; i < l; ++i) {
                for (unsigned j = i + 1; j < l; ++j) {
                    const Ed e1 = ed_of(in->inclusion_edges[b + i]), e2 = ed_of(in->inclusion_edges[b + j]);
                    unsigned long node1, node2;
                    int pos1;
                    bool ori1, ori2;
                    if (e1.v1 == e2.v1) continue;
                    else if (e1.v1 == e2.v2) {
                        node1 = e2.v1;
                        node2 = e1.v2;
                        pos1 = e2.pos1;
                        ori1 = e2.ori1;
                        ori2 = e1.ori2;
                    } else if (e1.v2 == e2.v1) {
                        node1 = e1.v1;
                        node2 = e2.v2;
                        pos1 = e1.pos1;
                        ori1 = e1.ori1;
                        ori2 = e2.ori2;
                    } else {
                        REF_ASSERT(e1.v2 == e2.v2);
                        continue;
                    }
                    const Rd &r1 = nodes.at(node1), &r2 = nodes.at(node2);
                    if (r1.paired || r2.paired) continue;
                    const int len = (int)std::min(r1.get_len() - (unsigned)pos1, r2.get_len());
                    REF_ASSERT(std::min(r1.get_len(), r2.get_len()) != 0);  // SIGFPE otherwise
                    const int perc = (int)floor((double)((unsigned)(100 * len) / std::min(r1.get_len(), r2.get_len())));
                    const double score = in->edge_threshold;
                    REF_ASSERT(score == 0 || score == -1 || score > 0);  // Edge ctor
                    REF_ASSERT(len > 0);                                // Edge::set_len
                    Ed ne{node1, node2, score, pos1, 0, len, 0, perc, '-', ori1, ori2};
                    if (check_edge(node1, node2, true) == -1) edge_vec.push_back(ne);
                }
            }
        }
        for (const Ed& e : edge_vec) update_overlap(e);
    }
};

// ---- FNO=3 ----------------------------------------------------------------
struct Sr3 {
    Rd rd;
    std::vector<hc_fno_original> list;                       // iteration order of get_original_reads()
    std::unordered_map<unsigned long, hc_fno_original> map;  // .at(original_id)
};

struct Ov3 {  // the Overlap object deduceOverlap returns
    unsigned long id1, id2;
    unsigned pos1, pos2;
    char ord, ori1, ori2;
    unsigned perc1, perc2, len1, len2;
    char type1, type2;
};

Ov3 make_overlap(unsigned long id1, unsigned long id2, unsigned pos1, unsigned pos2, char ord, char ori1, char ori2, unsigned perc1,
                 unsigned perc2, unsigned len1, unsigned len2, char t1, char t2) {
    // Overlap value constructor, src/Overlap.h:74-103: the check_* members take int
    if ((int)pos1 < 0 || (int)pos2 < 0) throw RefAbort{"overlap.m_pos < 0"};
    if ((int)perc1 < 0 || (int)perc1 > 100 || (int)perc2 < 0 || (int)perc2 > 100) throw RefAbort{"overlap.m_perc not in 0..100"};
    if ((int)len1 < 0 || (int)len2 < 0) throw RefAbort{"overlap.m_len < 0"};
    if (t1 == 's' || t2 == 's') REF_ASSERT(ord == '-');
    else REF_ASSERT(ord == '1' || ord == '2');
    return Ov3{id1, id2, pos1, pos2, ord, ori1, ori2, perc1, perc2, len1, len2, t1, t2};
}

int floor_ratio100(int a, int b) { return (int)floorf(a / float(b) * 100); }

// FindNextOverlaps3.cpp:176-406
Ov3 deduce_overlap(const Sr3& SR1, const Sr3& SR2, unsigned long original_id) {
    const hc_fno_original& o1 = SR1.map.at(original_id);
    const hc_fno_original& o2 = SR2.map.at(original_id);
    unsigned long id1, id2;
    int pos1, pos2, len1, len2;
    unsigned perc1, perc2;
    char ord, type1, type2;
    if (!SR1.rd.paired && !SR2.rd.paired) {
        const int idx1 = (int)o1.index1, idx2 = (int)o2.index1;
        const int lenA = SR1.rd.len1, lenB = SR2.rd.len1;
        if (idx1 - idx2 >= 0) {
            id1 = SR1.rd.id;
            id2 = SR2.rd.id;
            pos1 = idx1 - idx2;
            if (pos1 > lenA) return make_overlap(0, 0, 0, 0, '-', '-', '-', 0, 0, 0, 0, 's', 's');
            len1 = std::min(lenA - pos1, lenB);
        } else {
            id1 = SR2.rd.id;
            id2 = SR1.rd.id;
            pos1 = idx2 - idx1;
            if (pos1 > lenB) return make_overlap(0, 0, 0, 0, '-', '-', '-', 0, 0, 0, 0, 's', 's');
            len1 = std::min(lenA, lenB - pos1);
        }
        perc1 = (unsigned)perc_of(len1, lenA, lenB);
        pos2 = 0;
        ord = '-';
        perc2 = 0;
        len2 = 0;
        type1 = 's';
        type2 = 's';
    } else if (SR1.rd.paired && !SR2.rd.paired) {
        const int idx1l = (int)o1.index1, idx1r = (int)o1.index2, idx2l = (int)o2.index1, idx2r = (int)o2.index2;
        const int lenA1 = SR1.rd.len1, lenA2 = SR1.rd.len2, lenB = SR2.rd.len1;
        if (idx1l - idx2l >= 0) {
            id1 = SR1.rd.id;
            id2 = SR2.rd.id;
            pos1 = idx1l - idx2l;
            len1 = lenA1 - pos1;
            if (len1 <= 0) return make_overlap(0, 0, 0, 0, '-', '-', '-', 0, 0, 0, 0, 'p', 's');
            type1 = 'p';
            type2 = 's';
        } else {
            id1 = SR2.rd.id;
            id2 = SR1.rd.id;
            pos1 = idx2l - idx1l;
            len1 = std::min(lenA1, lenB - pos1);
            if (len1 <= 0) return make_overlap(0, 0, 0, 0, '-', '-', '-', 0, 0, 0, 0, 'p', 's');
            type1 = 's';
            type2 = 'p';
        }
        perc1 = (unsigned)floor_ratio100(len1, lenA1);
        pos2 = idx2r - idx1r;
        len2 = std::min(lenA2, lenB - pos2);
        if (len2 <= 0 || pos2 < 0) return make_overlap(0, 0, 0, 0, '-', '-', '-', 0, 0, 0, 0, 'p', 's');
        perc2 = (unsigned)floor_ratio100(len2, lenA2);
        ord = '-';
    } else if (!SR1.rd.paired && SR2.rd.paired) {
        const int idx1l = (int)o1.index1, idx1r = (int)o1.index2, idx2l = (int)o2.index1, idx2r = (int)o2.index2;
        const int lenA = SR1.rd.len1, lenB1 = SR2.rd.len1, lenB2 = SR2.rd.len2;
        if (idx1l - idx2l >= 0) {
            id1 = SR1.rd.id;
            id2 = SR2.rd.id;
            pos1 = idx1l - idx2l;
            len1 = std::min(lenB1, lenA - pos1);
            if (len1 <= 0) return make_overlap(0, 0, 0, 0, '-', '-', '-', 0, 0, 0, 0, 's', 'p');
            type1 = 's';
            type2 = 'p';
        } else {
            id1 = SR2.rd.id;
            id2 = SR1.rd.id;
            pos1 = idx2l - idx1l;
            len1 = lenB1 - pos1;
            if (len1 <= 0) return make_overlap(0, 0, 0, 0, '-', '-', '-', 0, 0, 0, 0, 's', 'p');
            type1 = 'p';
            type2 = 's';
        }
        perc1 = (unsigned)floor_ratio100(len1, lenB1);
        pos2 = idx1r - idx2r;
        len2 = std::min(lenB2, lenA - pos2);
        if (len2 <= 0 || pos2 < 0) return make_overlap(0, 0, 0, 0, '-', '-', '-', 0, 0, 0, 0, 's', 'p');
        perc2 = (unsigned)floor_ratio100(len2, lenB2);
        ord = '-';
    } else {
        const int idx1l = (int)o1.index1, idx1r = (int)o1.index2, idx2l = (int)o2.index1, idx2r = (int)o2.index2;
        const int lenA = SR1.rd.len1, lenB = SR2.rd.len1, lenC = SR1.rd.len2, lenD = SR2.rd.len2;
        bool front_ord, back_ord;
        if (idx1l - idx2l >= 0) {
            id1 = SR1.rd.id;
            id2 = SR2.rd.id;
            pos1 = idx1l - idx2l;
            len1 = std::min(lenA - pos1, lenB);
            front_ord = true;
        } else {
            id1 = SR2.rd.id;
            id2 = SR1.rd.id;
            pos1 = idx2l - idx1l;
            len1 = std::min(lenA, lenB - pos1);
            front_ord = false;
        }
        if (idx1r - idx2r >= 0) {
            pos2 = idx1r - idx2r;
            len2 = std::min(lenC - pos2, lenD);
            back_ord = true;
        } else {
            pos2 = idx2r - idx1r;
            len2 = std::min(lenC, lenD - pos2);
            back_ord = false;
        }
        if (len1 <= 0 || len2 <= 0) return make_overlap(0, 0, 0, 0, '1', '-', '-', 0, 0, 0, 0, 'p', 'p');
        perc1 = (unsigned)perc_of(len1, lenA, lenB);
        perc2 = (unsigned)perc_of(len2, lenC, lenD);
        REF_ASSERT(perc1 <= 100 && perc2 <= 100);
        ord = (front_ord == back_ord) ? '1' : '2';
        type1 = 'p';
        type2 = 'p';
    }
    return make_overlap(id1, id2, (unsigned)pos1, (unsigned)pos2, ord, '+', '+', perc1, perc2, (unsigned)len1, (unsigned)len2, type1, type2);
}

std::string overlap_line(const Ov3& o) {  // Overlap::get_overlap_line, src/Overlap.h:234-237
    std::string s = std::to_string(o.id1) + "\t" + std::to_string(o.id2) + "\t" + std::to_string(o.pos1) + "\t" + std::to_string(o.pos2) + "\t";
    s += o.ord;
    s += "\t";
    s += o.ori1;
    s += "\t";
    s += o.ori2;
    s += "\t" + std::to_string(o.perc1) + "\t" + std::to_string(o.perc2) + "\t" + std::to_string(o.len1) + "\t" + std::to_string(o.len2) + "\t";
    s += o.type1;
    s += "\t";
    s += o.type2;
    s += "\n";
    return s;
}

char* dup_text(const std::string& s, uint64_t* n) {
    char* p = (char*)malloc(s.size() + 1);
    memcpy(p, s.data(), s.size());
    p[s.size()] = 0;
    *n = s.size();
    return p;
}

}  // namespace

extern "C" {

// 0 = ok; 1 = the reference would have aborted (why[] has the reason)
int oracle_fno1(const hc_fno1_input* in, char** text, uint64_t* n_bytes, hc_fno_counters* c, char* why, uint64_t why_cap) {
    try {
        Fno1 f;
        f.in = in;
        f.run();
        std::string all;
        for (const std::string& l : f.lines) {
            all += l;
            all += "\n";
        }
        *text = dup_text(all, n_bytes);
        memset(c, 0, sizeof *c);
        c->n_lines = f.lines.size();
        c->copied = f.copied;
        c->u2sr = f.u2sr;
        c->v2sr = f.v2sr;
        c->sr2sr = f.sr2sr;
        return 0;
    } catch (const RefAbort& a) {
        if (why && why_cap) snprintf(why, why_cap, "%s", a.why.c_str());
        return 1;
    } catch (const std::exception& e) {
        if (why && why_cap) snprintf(why, why_cap, "exception: %s", e.what());
        return 1;
    }
}

int oracle_fno3(const hc_fno3_input* in, char** text, uint64_t* n_bytes, hc_fno_counters* c, char* why, uint64_t why_cap) {
    try {
        const uint64_t n = in->n_single + in->n_paired + in->n_trivial;
        std::deque<Sr3> srs;
        for (uint64_t i = 0; i < n; ++i) {
            Sr3 s;
            s.rd = rd_of(in->srs[i]);
            for (uint64_t k = in->orig_off[i]; k < in->orig_off[i + 1]; ++k) {
                s.list.push_back(in->originals[k]);
                s.map[(unsigned long)in->originals[k].original_id] = in->originals[k];
            }
            srs.push_back(std::move(s));
        }
        // findNextOverlaps3 :20-87
        std::unordered_map<unsigned long, unsigned long> original_to_index;
        std::deque<std::vector<const Sr3*>> nodes_to_SR(in->original_readcount);
        unsigned long index = 0;
        for (uint64_t i = 0; i < n; ++i) {
            for (const hc_fno_original& o : srs[i].list) {
                auto it = original_to_index.find((unsigned long)o.original_id);
                if (it == original_to_index.end()) {
                    original_to_index.insert(std::make_pair((unsigned long)o.original_id, index));
                    nodes_to_SR.at(index).push_back(&srs[i]);
                    index++;
                } else {
                    nodes_to_SR.at(it->second).push_back(&srs[i]);
                }
            }
        }
        // nodeDictApproach :89-177
        std::vector<std::set<unsigned long>> overlaps_found(in->new_read_count);
        struct Cand {
            const Sr3 *a, *b;
            unsigned long original_id;
        };
        std::list<Cand> overlaps_list;
        const std::unordered_map<unsigned long, unsigned long> by_value(original_to_index);  // nodeDictApproach takes it by value (:89)
        for (const auto& kv : by_value) {
            const std::vector<const Sr3*>& SR_list = nodes_to_SR.at(kv.second);
            for (size_t i = 0; i < SR_list.size(); ++i) {
                for (size_t j = i + 1; j < SR_list.size(); ++j) {
                    const unsigned long id1 = SR_list[i]->rd.id, id2 = SR_list[j]->rd.id;
                    const unsigned long smallest = std::min(id1, id2), largest = std::max(id1, id2);
                    if (overlaps_found.at(smallest).count(largest)) continue;
                    overlaps_found.at(smallest).insert(largest);
                    overlaps_list.push_back(Cand{SR_list[i], SR_list[j], kv.first});
                }
            }
        }
        std::string all;
        uint64_t count = 0;
        for (const Cand& cd : overlaps_list) {
            const Ov3 o = deduce_overlap(*cd.a, *cd.b, cd.original_id);
            const unsigned perc = o.perc2 > 0 ? (unsigned)(0.5 * (o.perc1 + o.perc2)) : o.perc1;  // Overlap::get_perc :203-210
            if ((in->flags & HC_FNO_NO_INCLUSIONS) && perc == 100) continue;
            if ((int)o.len1 > 0) {
                all += overlap_line(o);
                count++;
            }
        }
        *text = dup_text(all, n_bytes);
        memset(c, 0, sizeof *c);
        c->n_lines = count;
        c->candidates = overlaps_list.size();
        return 0;
    } catch (const RefAbort& a) {
        if (why && why_cap) snprintf(why, why_cap, "%s", a.why.c_str());
        return 1;
    } catch (const std::exception& e) {
        if (why && why_cap) snprintf(why, why_cap, "exception: %s", e.what());
        return 1;
    }
}

// computeOverlapData alone. Returns 0 ok / 1 reference abort; *ok = the reference's bool.
int oracle_fno_compute_overlap_data(const hc_fno_read* s1, const hc_fno_read* s2, const int32_t idx[4], const hc_fno_edge* e, int32_t* ok,
                                    int32_t out9[9]) {
    try {
        int np1 = 0, np2 = 0, perc = 0, l1 = 0, l2 = 0;
        char o1 = 0, o2 = 0, t1 = 0, t2 = 0;
        *ok = compute_overlap_data(rd_of(*s1), rd_of(*s2), idx[0], idx[1], idx[2], idx[3], ed_of(*e), np1, np2, o1, o2, t1, t2, perc, l1, l2);
        const int32_t v[9] = {np1, np2, o1, o2, t1, t2, perc, l1, l2};
        memcpy(out9, v, sizeof v);
        return 0;
    } catch (const RefAbort&) {
        return 1;
    }
}

void oracle_fno_free(char* p) { free(p); }

}  // extern "C"
