// hc_fno_device.h — the device form of find-next-overlaps' second half: computeOverlapData
// (src/FindNextOverlaps.cpp:351-565) for a batch of (edge, super-read, super-read) combinations, then a radix sort +
// unique in place of the reference's std::set<std::string> (:937-948), then the text of the surviving lines.
// The order of a std::set<std::string> of 13-column lines is the order of the column tuples with every number compared
// as its DECIMAL TEXT (a tab ends the shorter text and sorts below '-' and every digit), so each column maps to an
// order-preserving integer and the line to one 256-bit key; equal keys <=> equal lines.
#ifndef HC_FNO_DEVICE_H_
#define HC_FNO_DEVICE_H_
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/hcfno.h"
#include "hc_fno_items.h"

namespace hc {

enum : unsigned long long {
    kFnoStatusRequire = 1,  // an assert / .at() of the reference would fire: the host path reports which
    kFnoStatusRange = 2,    // a number outside what the keys hold (id >= 10^10, perc > 999): the host path takes over
    kFnoStatusUnsorted = 4  // graph_edges do not come vertex by vertex: the host form builds adj_out by counting
};
// counters: [0..3] lines per kind before the unique (hc_fno_counters), [4] status bits, [5] lines after the unique
constexpr int kFnoCounters = 6;

hipError_t fno_deduce(const FnoItem* items, uint64_t n, uint32_t no_inclusions, FnoRec* rec, uint64_t* k0, uint64_t* k1, uint64_t* k2,
                      uint64_t* k3, uint32_t* iota, unsigned long long* counters, hipStream_t s);
hipError_t fno_gather_keys(const uint64_t* key, const uint32_t* perm, uint64_t n, uint64_t* out, hipStream_t s);
// len[i] = bytes (with the newline) of the line at sorted place i if it is the first of its run of equal lines, else 0; len[n] = 0
hipError_t fno_mark_lines(const FnoRec* rec, const uint32_t* perm, uint64_t n, uint64_t* len, unsigned long long* counters, hipStream_t s);
hipError_t fno_format(const FnoRec* rec, const uint32_t* perm, const uint64_t* len, const uint64_t* off, uint64_t n, char* text, hipStream_t s);
// ---- the walk of FNO=1 on the device (hc_fno_kernels.hip) ---------------------------------------------------------------------
struct FnoWalkInput {  // device pointers
    const hc_fno_edge* edges;  // every edge updateOverlap is called on, in the reference's order
    uint64_t n_edges;
    const hc_fno_read* nodes;
    uint64_t n_nodes;
    const hc_fno_read* srs;
    uint64_t n_srs;
    const uint64_t* n2s_off;  // nodes_to_SR as CSR (:893-906)
    const uint32_t* n2s;
    const uint64_t* subread_off;
    const hc_fno_subread* subreads;  // of every super-read, sorted by node
    uint64_t new_read_count;
    uint32_t id_bits;  // ids < new_read_count < 2^id_bits; a pair's key is lo << id_bits | hi
    uint32_t resolve_orientations;
};
// what the walk needs first: adj_out's offsets from the sorted graph_edges, the stored non-edges that pass :702, nodes_to_SR from the cliques
hipError_t fno_adj_offsets(const hc_fno_edge* ge, uint64_t G, uint64_t n_nodes, uint64_t* off, unsigned long long* counters, hipStream_t s);
hipError_t fno_nonedge_filter(const hc_fno_edge* nonedges, uint64_t n, const hc_fno_edge* ge, const uint64_t* off, uint64_t n_nodes, uint8_t* keep,
                              unsigned long long* counters, hipStream_t s);
hipError_t fno_gather_edges(const hc_fno_edge* in, const uint32_t* idx, uint64_t n, hc_fno_edge* out, hipStream_t s);
// --add_duplicates: out[2 i] = in[idx[i]], out[2 i + 1] = its opposite (fno_mirror_nonedge); a record the host form has to report sets the status
hipError_t fno_gather_mirrored(const hc_fno_edge* in, const uint32_t* idx, uint64_t n, const hc_fno_read* nodes, uint64_t half, hc_fno_edge* out,
                               unsigned long long* counters, hipStream_t s);
hipError_t fno_clique_pairs(const uint64_t* clique_nodes, const uint64_t* clique_off, uint64_t n_srs, uint64_t total, uint64_t n_nodes, uint64_t* key,
                            uint32_t* sr, unsigned long long* counters, hipStream_t s);
hipError_t fno_offsets(const uint64_t* sorted, uint64_t n, uint64_t n_nodes, uint64_t* off, hipStream_t s);
// cnt_*[i], i <= n_edges ([n_edges] = 0): combinations / copied items of edge i
hipError_t fno_walk_count(const FnoWalkInput& w, uint64_t* cnt_comb, uint64_t* cnt_direct, unsigned long long* counters, hipStream_t s);
// key[c] = the unordered pair of new ids of combination c (walk order), ~0 for a skipped one; iota[c] = c
hipError_t fno_walk_expand(const FnoWalkInput& w, const uint64_t* off_comb, uint64_t n_comb, uint64_t* key, uint32_t* iota, unsigned long long* counters,
                           hipStream_t s);
hipError_t fno_walk_heads(const uint64_t* key_sorted, uint64_t n_comb, uint8_t* flag, hipStream_t s);  // first of every run of equal pairs
hipError_t fno_walk_direct(const uint64_t* off_direct, uint64_t n_edges, uint32_t* direct_edge, hipStream_t s);
hipError_t fno_walk_items(const FnoWalkInput& w, const uint64_t* off_comb, const uint32_t* direct_edge, uint64_t n_direct, const uint32_t* val_sorted,
                          const uint32_t* head_pos, uint64_t n_heads, FnoItem* items, unsigned long long* counters, hipStream_t s);

// FNO=3: deduceOverlap per candidate pair (in walk order); len[i] = bytes of its line (0: no line), len[n] = 0; counters[4] status
// bits, counters[5] lines.  Then the text at off[i] (the exclusive scan of len).
struct Fno3Rec {
    uint64_t id1, id2;
    int32_t pos1, pos2, perc1, perc2, len1, len2;
    uint8_t ord, type1, type2, pad[5];
};
static_assert(sizeof(Fno3Rec) == 48, "Fno3Rec is 48 bytes");
hipError_t fno3_deduce(const FnoItem* items, uint64_t n, uint32_t no_inclusions, Fno3Rec* rec, uint64_t* len, unsigned long long* counters, hipStream_t s);
hipError_t fno3_format(const Fno3Rec* rec, const uint64_t* len, const uint64_t* off, uint64_t n, char* text, hipStream_t s);
// stable radix sort of (64-bit key, 32-bit value) pairs (hc_graph_kernels.hip owns the instantiation)
hipError_t sort_pairs_u64_u32(void* temp, size_t& temp_bytes, const uint64_t* k_in, uint64_t* k_out, const uint32_t* v_in, uint32_t* v_out,
                              uint32_t n, int end_bit, hipStream_t s);

}  // namespace hc
#endif
