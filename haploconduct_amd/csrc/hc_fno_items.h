// hc_fno_items.h — the records find-next-overlaps hands to its device form (hc_fno_device.h) and the host-side entry
// points of that form (hc_api_fno.cpp).  Plain C++: the host code that fills the records builds without HIP.
#ifndef HC_FNO_ITEMS_H_
#define HC_FNO_ITEMS_H_
#include <cstdint>
#include <functional>

#include "../../include/hcfno.h"

namespace hc {

// One combination, everything the arithmetic needs already looked up (host: ids, lengths, findCliqueIndex offsets).
//   kind 0 (copied, :44-68):  v = { pos1, pos2, perc, len1, len2 } of the edge
//   kind 1..3:                v = { e.pos1, e.pos2, i1l, i1r, i2l, i2r, a.len1, a.len2, b.len1, b.len2 }
struct FnoItem {
    uint64_t ida, idb;
    int32_t v[10];
    uint8_t kind, a_paired, b_paired, e_ord, ori1, ori2, pad[2];
};
static_assert(sizeof(FnoItem) == 64, "FnoItem is 64 bytes");

struct FnoRec {  // the 13 columns of one line (perc2 is always "0")
    uint64_t id1, id2;
    int32_t pos1, pos2, perc, len1, len2;
    uint8_t ord2, ori1, ori2, type1, type2, kind, valid, pad;
};
static_assert(sizeof(FnoRec) == 48, "FnoRec is 48 bytes");

// HC_FNO=host | device; by default the device takes batches of 200 000 combinations and more when there is one.
// Throws FatalError{HC_ERR_NO_DEVICE} for HC_FNO=device without a device.
bool fno_device_wanted(uint64_t n_items);
// computeOverlapData per item, the std::set<std::string> order and unique, the text: `text_of(bytes)` returns where the
// text goes; counters[0..3] lines per kind before the unique, counters[4] lines of the text.  false: the device met
// something the host path has to report (a stop of the reference) or to handle (numbers beyond the keys' range);
// text_of was not called.  seconds[0..1] (may be null): copy + deduce + sorts + unique + scan, text.
bool fno_lines_on_device(const FnoItem* items, uint64_t n, bool no_inclusions, const std::function<char*(uint64_t)>& text_of,
                         uint64_t counters[5], double* seconds);

// FNO=1 whole on the device.  The edges updateOverlap is called on, in the reference's order: adj_out (graph_edges, vertex by
// vertex — anything else makes the call return false), branching_edges, the stored non-edges (ALL of them when use_nonedges: which
// pass :702 is decided on the device against adj_out), the inclusion-induced edges (the host's: few); the cliques nodes_to_SR is
// made of (:893-906); every super-read's subreads sorted by node.
#if defined(__HIPCC__)
#define HC_FNO_HD __host__ __device__
#else
#define HC_FNO_HD
#endif
// --add_duplicates (program_settings.add_duplicates; HC_FNO_ADD_DUPLICATES): the graph has a vertex per read AND strand — vertex r for
// read r as it is, r + half for its reverse complement (src/ViralQuasispecies.cpp:246-270) — and reconsiderNonedgeOverlaps adds every
// stored non-edge that passes :702 a second time, seen from the other strand (src/FindNextOverlaps.cpp:699-793).  `e` is the line's
// own edge (vertices by orientation, :672-675), r1 / r2 its two reads; *o receives the opposite edge.  The reference computes the
// positions as `size_t - int - size_t` assigned to an int: exact arithmetic modulo 2^32.
// Returns 0; 1 where the reference's assert at :755 fires (two paired reads, ord neither "1" nor "2"); 2 when a vertex does not lie
// on the strand the line's orientation names (not a record reconsiderNonedgeOverlaps could have built).
HC_FNO_HD inline int fno_mirror_nonedge(const hc_fno_edge& e, const hc_fno_read& r1, const hc_fno_read& r2, uint64_t half, hc_fno_edge* o) {
    if ((e.v1 < half) != (e.ori1 != 0) || (e.v2 < half) != (e.ori2 != 0)) return 2;
    const uint64_t w1 = e.v1 < half ? e.v1 + half : e.v1 - half;  // get_vertex_id(!ori1), :700-701
    const uint64_t w2 = e.v2 < half ? e.v2 + half : e.v2 - half;
    const uint32_t p1 = (uint32_t)e.pos1, p2 = (uint32_t)e.pos2;
    uint32_t q1, q2 = 0;
    bool both_paired = false;
    if (!r1.paired && !r2.paired) {  // S-S, :702-717
        q1 = r1.len1 - p1 - r2.len1;
    } else if (r1.paired && !r2.paired) {  // P-S, :718-735
        q1 = r1.len2 + p2 - r2.len1;
        q2 = r2.len1 + p1 - r1.len1;
    } else if (!r1.paired && r2.paired) {  // S-P, :736-753
        q1 = r1.len1 - p2 - r2.len2;
        q2 = r1.len1 - p1 - r2.len1;
    } else {  // P-P, :754-792
        if (e.ord == '1') q1 = r1.len2 - p2 - r2.len2;
        else if (e.ord == '2') q1 = r1.len2 + p2 - r2.len2;
        else return 1;
        q2 = r1.len1 - p1 - r2.len1;
        both_paired = true;
    }
    int32_t pos1 = (int32_t)q1, pos2 = (int32_t)q2;
    const bool turned = pos1 < 0;  // the opposite overlap starts in read2: it leaves from read2's vertex
    uint8_t ord = e.ord;
    if (both_paired) {
        if (pos2 < 0) {
            pos2 = (int32_t)(0u - (uint32_t)pos2);
            ord = turned ? '1' : '2';
        } else {
            ord = turned ? '2' : '1';
        }
    }
    *o = e;
    o->score = 0;
    o->pos1 = turned ? (int32_t)(0u - (uint32_t)pos1) : pos1;
    o->pos2 = pos2;
    o->ord = ord;
    o->v1 = turned ? w2 : w1;
    o->v2 = turned ? w1 : w2;
    o->ori1 = turned ? !e.ori2 : !e.ori1;
    o->ori2 = turned ? !e.ori1 : !e.ori2;
    return 0;
}

struct FnoEdgeSpan {
    const hc_fno_edge* p;
    uint64_t n;
};
struct FnoWalkHost {
    FnoEdgeSpan graph, branching, nonedges, induced;
    const hc_fno_read* nodes;
    uint64_t n_nodes;
    const hc_fno_read* srs;
    uint64_t n_srs;
    const uint64_t* clique_off;
    const uint64_t* clique_nodes;
    const uint64_t* subread_off;
    const hc_fno_subread* subreads;
    uint64_t new_read_count;
    bool resolve_orientations, no_inclusions;
    uint64_t dup_half;  // --add_duplicates: n_nodes / 2 (every kept stored non-edge is followed by its opposite, fno_mirror_nonedge); else 0
};
// counters as fno_lines_on_device; *n_items: combinations kept (copied edges + first combination per pair); seconds[0..2] (may be
// null): copies + walk + look-ups, deduce + sorts + unique + scan, text.  false: something the host form has to report or handle.
bool fno1_walk_on_device(const FnoWalkHost& h, const std::function<char*(uint64_t)>& text_of, uint64_t counters[5], uint64_t* n_items,
                         double* seconds);

// FNO=3 (src/FindNextOverlaps3.cpp:176-406, deduceOverlap): one candidate pair of super-reads that share an original read,
// everything looked up on the host —
//   kind 4: ida / idb = the super-reads' ids, a_paired / b_paired, v = { index1 of the original in A, index2 in A, index1 in B,
//           index2 in B, A.len1, A.len2, B.len1, B.len2 }
// — deduced on the device in the order given (the walk's), the lines' text written in that order.  false: the device met
// something the reference would stop at; the host form then runs and reports.  *n_lines: lines written.
bool fno3_lines_on_device(const FnoItem* items, uint64_t n, bool no_inclusions, const std::function<char*(uint64_t)>& text_of, uint64_t* n_lines,
                          double* seconds);

}  // namespace hc
#endif
