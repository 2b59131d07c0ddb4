// ref_headers_driver.cpp — build-owned driver (NOT reference code) that exposes the
// Boost-free, header-only parts of the reference through a C ABI so that the oracle
// restatement (hc_oracle.c) can be checked against genuine reference code.
//
// Compiled by oracle/Makefile as
//     g++ -std=c++11 -O2 -I/root/reference/src oracle/ref_headers_driver.cpp -shared -fPIC
// i.e. the reference headers Types.h, Overlap.h, Read.h and Edge.h are included from
// where they lie; nothing of them is copied into this repository.  The standard
// headers below are included first because the reference headers rely on their
// includer (normally Boost) to have pulled them in (std::vector in Overlap.h:39,
// std::cerr in Overlap.h:104, std::count in Read.h:219).
//
// The reference's scoring TU (EdgeCalculator.cpp) is NOT built here: it needs
// Boost, which this image lacks.
#include <algorithm>
#include <cstring>
#include <iostream>
#include <list>
#include <map>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

#include "Types.h"
#include "Overlap.h"
#include "Read.h"
#include "Edge.h"

extern "C" {

struct ref_overlap_out {
    unsigned long id1, id2;
    int pos1, pos2;
    char ord, ori1, ori2, type1, type2;
    unsigned int perc, len1, len2;
    char line[256];
};

// Overlap(std::vector<std::string>) + getters (Overlap.h:39-237).  The reference exits /
// asserts on malformed fields, so callers only pass well-formed ones.
int ref_overlap_parse(const char* const fields[13], ref_overlap_out* out) {
    std::vector<std::string> v;
    for (int i = 0; i < 13; i++) v.push_back(fields[i]);
    Overlap o(v);
    out->id1 = o.get_id(1);
    out->id2 = o.get_id(2);
    out->pos1 = o.get_pos(1);
    out->pos2 = o.get_pos(2);
    out->ord = o.get_ord()[0];
    out->ori1 = o.get_ori(1)[0];
    out->ori2 = o.get_ori(2)[0];
    out->type1 = o.get_type(1)[0];
    out->type2 = o.get_type(2)[0];
    out->perc = o.get_perc();
    out->len1 = o.get_len(1);
    out->len2 = o.get_len(2);
    std::string l = o.get_overlap_line();
    std::strncpy(out->line, l.c_str(), sizeof(out->line) - 1);
    out->line[sizeof(out->line) - 1] = 0;
    return 0;
}

// build_rev_comp (Types.h:109-129); out must hold len+1 bytes.
int ref_build_rev_comp(const char* seq, char* out) {
    std::string r = build_rev_comp(std::string(seq));
    std::memcpy(out, r.c_str(), r.size() + 1);
    return (int)r.size();
}

// str_to_read_id (Types.h:99-102)
unsigned long ref_str_to_read_id(const char* s) { return str_to_read_id(std::string(s)); }

// Read getters (Read.h:144-201): which = 0 seq, 1 phred, 2 rev_comp, 3 rev_phred; i = 0/1/2.
int ref_read_get(int is_paired, const char* seq1, const char* seq2, const char* ph1, const char* ph2, int which,
                 int i, char* out) {
    Read r(is_paired != 0, false, 7, seq1, seq2, ph1, ph2);
    std::string s;
    if (which == 0) s = r.get_seq(i);
    else if (which == 1) s = r.get_phred(i);
    else if (which == 2) s = r.get_rev_comp(i);
    else s = r.get_rev_phred(i);
    std::memcpy(out, s.c_str(), s.size() + 1);
    return (int)s.size();
}

struct ref_edge_io {
    double score, mismatch;
    int pos1, pos2, pos3, pos4;
    int ori1, ori2;
    char ord;
    unsigned long v1, v2;
    int perc, len0, len1, len2;
    int read1_is_a; // after the operation: does read(1) point at the read that was passed first?
};

// Edge construction as compute_overlap does it (EdgeCalculator.cpp:223-231) followed,
// if do_swap, by Edge::swap_reads (Edge.h:74-88).
int ref_edge_build(ref_edge_io* io, int do_swap) {
    Read a(false, false, 1, "A", "", "I", "");
    Read b(false, false, 2, "A", "", "I", "");
    std::string ord(1, io->ord);
    Edge e(io->score, io->pos1, io->pos2, io->ori1 != 0, io->ori2 != 0, ord, &a, &b);
    e.set_vertices(io->v1, io->v2);
    e.set_extra_pos(io->pos3, io->pos4);
    e.set_perc(io->perc);
    e.set_len(io->len1, io->len2);
    e.set_mismatch(io->mismatch);
    if (do_swap) e.swap_reads();
    io->score = e.get_score();
    io->mismatch = e.get_mismatch_rate();
    io->pos1 = e.get_pos(1);
    io->pos2 = e.get_pos(2);
    io->pos3 = e.get_extra_pos(1);
    io->pos4 = e.get_extra_pos(2);
    io->ori1 = e.get_ori(1);
    io->ori2 = e.get_ori(2);
    io->ord = e.get_ord();
    io->v1 = e.get_vertex(1);
    io->v2 = e.get_vertex(2);
    io->perc = e.get_perc();
    io->len0 = e.get_len(0);
    io->len1 = e.get_len(1);
    io->len2 = e.get_len(2);
    io->read1_is_a = (e.get_read(1) == &a);
    return 0;
}

// Default-constructed-by-the-8-arg-ctor field values (Edge.h:43-57).
int ref_edge_defaults(double* mismatch) {
    Read a(false, false, 1, "A", "", "I", "");
    Read b(false, false, 2, "A", "", "I", "");
    Edge e(0, 0, 0, true, true, "-", &a, &b);
    *mismatch = e.get_mismatch_rate();
    return 0;
}

} // extern "C"
