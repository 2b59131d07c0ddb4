// Edge.h — an admitted (or candidate) edge of the overlap graph (reference src/Edge.h:18-276).
#pragma once
#include <string>
#include <utility>

#include "Read.h"
#include "Types.h"

namespace hc {

class Edge {
public:
    Edge() = default;
    // src/Edge.h:43-57: perc / len / mismatch start at -1
    Edge(double s, int p1, int p2, bool orientation1, bool orientation2, const std::string& o, Read* r1, Read* r2)
        : score(s), pos1(p1), pos2(p2), ori1(orientation1), ori2(orientation2), read1(r1), read2(r2),
          ord(o.empty() ? '\0' : o[0]) {
        if (!(score == 0 || score == -1 || score > 0)) throw FatalError{-9, "Edge: score must be 0, -1 or positive"};
    }

    Edge(double s, int p1, int p2, bool orientation1, bool orientation2, char o, Read* r1, Read* r2)
        : score(s), pos1(p1), pos2(p2), ori1(orientation1), ori2(orientation2), read1(r1), read2(r2), ord(o) {
        if (!(score == 0 || score == -1 || score > 0)) throw FatalError{-9, "Edge: score must be 0, -1 or positive"};
    }

    double get_score() const { return score; }
    void set_mismatch(double m) { mismatch_rate = m; }
    double get_mismatch_rate() const { return mismatch_rate; }
    void set_vertices(node_id_t v1, node_id_t v2) { vertex1 = v1; vertex2 = v2; }
    node_id_t get_vertex(int i) const { return i == 1 ? vertex1 : vertex2; }
    int get_pos(int i) const { return i == 1 ? pos1 : pos2; }
    bool get_ori(int i) const { return i == 1 ? ori1 : ori2; }
    char get_ord() const { return ord; }
    Read* get_read(int i) const { return i == 1 ? read1 : read2; }
    void set_extra_pos(int p3, int p4 = 0) { pos3 = p3; pos4 = p4; }
    int get_extra_pos(int i) const { return i == 1 ? pos3 : pos4; }
    int get_perc() const { return overlap_perc; }
    void set_perc(int p) { overlap_perc = p; }
    int get_len(int i) const { return i == 0 ? overlap_len : (i == 1 ? overlap_len1 : overlap_len2); }
    void set_len(int len1, int len2) {                    // src/Edge.h:211-218
        if (!(len1 > 0) || !(len2 >= 0)) throw FatalError{-9, "Edge::set_len: len1 must be > 0 and len2 >= 0"};
        overlap_len = len1 + len2;
        overlap_len1 = len1;
        overlap_len2 = len2;
    }
    // src/Edge.h:74-88: only legal when pos1 == 0 and vertex1 > vertex2
    void swap_reads() {
        std::swap(read1, read2);
        std::swap(vertex1, vertex2);
        std::swap(ori1, ori2);
        if (ord == '1') ord = '2';
        else if (ord == '2') ord = '1';
        pos3 = -pos3;
        pos4 = -pos4;
    }

private:
    double score = 0;
    int pos1 = 0, pos2 = 0, pos3 = 0, pos4 = 0;
    bool ori1 = true, ori2 = true;
    Read* read1 = nullptr;
    Read* read2 = nullptr;
    char ord = '-';
    node_id_t vertex1 = 0, vertex2 = 0;
    int overlap_perc = -1, overlap_len = -1, overlap_len1 = -1, overlap_len2 = -1;
    double mismatch_rate = -1;
};

}  // namespace hc
