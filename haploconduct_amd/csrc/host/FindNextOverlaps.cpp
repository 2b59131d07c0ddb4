// FindNextOverlaps.cpp — the overlaps file of the next iteration, induced from this iteration's edges
// (include/hcfno.h).  Mirrors SRBuilder::findNextOverlaps (src/FindNextOverlaps.cpp) and
// SRBuilder::findNextOverlaps3 (src/FindNextOverlaps3.cpp) on flat arrays.
//
// Shape of the work:
//   walk     enumerate (edge, super-read, super-read) combinations and keep the FIRST one, in the reference's
//            walk order, per unordered super-read pair (the reference's overlaps_found sets: which combination
//            wins is observable).  FNO=1: every combination carries its walk sequence number, combinations are
//            partitioned by a hash of the pair and each partition keeps the minimum per pair — a parallel
//            group-by, no shared set.  FNO=3: the walk order is a std::unordered_map's, kept sequential.
//   deduce   parallel over the kept combinations: positions, lengths, percentage, the text line;
//   emit     FNO=1: parallel sample sort + unique of the lines in std::string order; FNO=3: walk order.
// The reference does the three interleaved on one thread with a std::set<std::string> per batch.
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <stdlib.h>
#include <sys/mman.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <memory>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../../include/hcfno.h"
#include "../hc_fno_items.h"
#include "DefaultInit.h"
#include "Types.h"

namespace hc {
int set_last_error(int status, const std::string& what);
}

struct hc_fno_output {
    std::vector<char, hc::DefaultInitAllocator<char>> text;  // sized, then written in full: not zero-filled first
    hc_fno_counters counters;
    bool on_device = false;       // the arithmetic, the ordering and the text were the device's
    bool walk_on_device = false;  // ... and the walk and the look-ups too (FNO=1)
};

namespace {
using hc::FatalError;

[[noreturn]] void ref_abort(const char* what) { throw FatalError{HC_ERR_FORMAT, std::string("the reference stops here: ") + what}; }
#define FNO_REQUIRE(c) \
    do {               \
        if (!(c)) ref_abort(#c); \
    } while (0)

// the output's text sized for the device's copy: fresh pages, 2 MiB ones where the system grants them (the copy of 0.4 GB into
// untouched 4 KiB pages runs at 12 GB/s, a tenth of the pages' faults is most of that)
char* sized_text(hc_fno_output& out, uint64_t bytes) {
    out.text.resize(bytes);
    const uintptr_t huge = (uintptr_t)2 << 20, b = ((uintptr_t)out.text.data() + huge - 1) & ~(huge - 1), e = ((uintptr_t)out.text.data() + bytes) & ~(huge - 1);
    if (e > b) (void)madvise((void*)b, e - b, MADV_HUGEPAGE);
    return out.text.data();
}

unsigned thread_count(uint32_t asked) {
    unsigned n = asked ? asked : std::thread::hardware_concurrency();
    if (!asked && n > 64) n = 64;  // past 64 threads the merge/scatter steps only add overhead (tools/fno_bench.py)
    return n ? n : 1;
}

template <typename F>
void parallel_chunks(uint64_t n, unsigned threads, F&& f) {  // f(begin, end, thread_index)
    if (n == 0) return;
    if (threads > n) threads = (unsigned)n;
    if (n >= 65536 && threads > n / 8192 + 1) threads = (unsigned)(n / 8192 + 1);  // item loops (task lists are short): starting a thread costs as much as a few thousand items
    if (threads <= 1) {
        f((uint64_t)0, n, 0u);
        return;
    }
    std::vector<std::thread> th;
    std::vector<FatalError> errs(threads, FatalError{0, ""});
    for (unsigned t = 0; t < threads; ++t)
        th.emplace_back([&, t] {
            try {
                f(n * t / threads, n * (t + 1) / threads, t);
            } catch (const FatalError& e) {
                errs[t] = e;
            } catch (const std::exception& e) {
                errs[t] = FatalError{HC_ERR_FORMAT, e.what()};
            }
        });
    for (auto& x : th) x.join();
    for (auto& e : errs)
        if (e.status) throw e;
}

// ---- set of unordered id pairs (the reference's vector<set<read_id_t>> overlaps_found) --------------------
class PairSet {
public:
    explicit PairSet(uint64_t bound) : bound_(bound) { rehash(1u << 12); }
    // true if the pair was already present; inserts it otherwise
    bool test_and_set(uint64_t a, uint64_t b) {
        uint64_t lo = a < b ? a : b, hi = a < b ? b : a;
        if (lo >= bound_) ref_abort("overlaps_found.at(id): id >= new_read_count");
        if ((used_ + 1) * 10 > cap_ * 7) rehash(cap_ * 2);
        uint64_t h = mix(lo, hi) & (cap_ - 1);
        for (;;) {
            Slot& s = slots_[h];
            if (s.hi == kEmpty) {
                s.lo = lo;
                s.hi = hi;
                ++used_;
                return false;
            }
            if (s.lo == lo && s.hi == hi) return true;
            h = (h + 1) & (cap_ - 1);
        }
    }

private:
    struct Slot {
        uint64_t lo, hi;
    };
    static constexpr uint64_t kEmpty = ~(uint64_t)0;
    static uint64_t mix(uint64_t a, uint64_t b) {
        uint64_t x = a * 0x9E3779B97F4A7C15ull ^ (b + 0x7F4A7C15ull + (a << 6) + (a >> 2));
        x ^= x >> 32;
        x *= 0xD6E8FEB86659FD93ull;
        x ^= x >> 32;
        return x;
    }
    void rehash(uint64_t cap) {
        std::vector<Slot> old;
        old.swap(slots_);
        slots_.assign(cap, Slot{0, kEmpty});
        cap_ = cap;
        used_ = 0;
        for (const Slot& s : old)
            if (s.hi != kEmpty) {
                uint64_t h = mix(s.lo, s.hi) & (cap_ - 1);
                while (slots_[h].hi != kEmpty) h = (h + 1) & (cap_ - 1);
                slots_[h] = s;
                ++used_;
            }
    }
    std::vector<Slot> slots_;
    uint64_t cap_ = 0, used_ = 0, bound_;
};

// ---- small text helpers -----------------------------------------------------------------------------------
inline char* put_u64(char* p, uint64_t v) {
    char tmp[24];
    int n = 0;
    do {
        tmp[n++] = (char)('0' + v % 10);
        v /= 10;
    } while (v);
    while (n) *p++ = tmp[--n];
    return p;
}
inline char* put_i32(char* p, int32_t v) {  // std::to_string(int)
    if (v < 0) {
        *p++ = '-';
        return put_u64(p, (uint64_t)(-(int64_t)v));
    }
    return put_u64(p, (uint64_t)v);
}

struct ReadFacts {
    int len1, len2;
    bool paired;
};
inline ReadFacts facts(const hc_fno_read& r) { return ReadFacts{(int)r.len1, (int)r.len2, r.paired != 0}; }

inline int perc_max(int ov, int la, int lb) {  // (int)floor(std::max(ov/float(la), ov/float(lb))*100), :375
    const float a = (float)ov / (float)la, b = (float)ov / (float)lb;
    const float m = (a < b ? b : a) * 100.0f;
    return (int)floorf(m);
}

struct Induced {
    int pos1, pos2, perc, len1, len2;
    char ord1, ord2, type1, type2;
};

// computeOverlapData, src/FindNextOverlaps.cpp:351-565.  false = "failure" (too much of the read was trimmed
// away from the super-read, or a paired read that should have been merged).
bool induced_overlap(const ReadFacts& a, const ReadFacts& b, int i1l, int i1r, int i2l, int i2r, const hc_fno_edge& e, Induced& o) {
    const int shift1 = (e.pos1 + i1l) - i2l;
    o.ord1 = shift1 < 0 ? '2' : '1';
    o.pos1 = shift1 < 0 ? -shift1 : shift1;
    o.type1 = a.paired ? 'p' : 's';
    o.type2 = b.paired ? 'p' : 's';
    if (!a.paired && !b.paired) {  // :358-385
        FNO_REQUIRE(a.len1 > 0 && b.len1 > 0);
        const int len = shift1 < 0 ? b.len1 : a.len1;
        o.len1 = std::min(std::min(len - o.pos1, a.len1), b.len1);
        o.len2 = 0;
        o.perc = perc_max(o.len1, a.len1, b.len1);
        o.ord2 = '-';
        o.pos2 = 0;
        if (o.pos1 >= len) return false;
    } else if (a.paired != b.paired) {  // P-S :387-440, S-P :442-486
        const ReadFacts& P = a.paired ? a : b;  // the paired one
        const ReadFacts& S = a.paired ? b : a;
        FNO_REQUIRE(P.len1 + P.len2 > 0 && S.len1 > 0);
        // which sequence the first overlap starts in: the one that is shifted right
        const bool starts_in_pair = a.paired ? (shift1 >= 0) : (shift1 < 0);
        if (o.pos1 >= (starts_in_pair ? P.len1 : S.len1)) return false;
        o.len1 = starts_in_pair ? P.len1 - o.pos1 : P.len1;
        if (a.paired) o.pos2 = e.ord == '1' ? i2r - (i1r + e.pos2) : (e.pos2 + i2r) - i1r;
        else o.pos2 = e.ord == '2' ? i1r - (e.pos2 + i2r) : i1r + e.pos2 - i2r;
        if (o.pos2 >= S.len1 || o.pos2 < 0) return false;
        o.ord2 = '-';
        o.len2 = std::min(S.len1 - o.pos2, P.len2);
        const int total = o.len1 + o.len2;
        o.perc = std::min(a.paired ? perc_max(total, P.len1 + P.len2, S.len1) : perc_max(total, S.len1, P.len1 + P.len2), 100);
    } else {  // P-P :488-547
        if (o.pos1 >= (shift1 < 0 ? b.len1 : a.len1)) return false;
        o.len1 = shift1 < 0 ? std::min(a.len1, b.len1 - o.pos1) : std::min(a.len1 - o.pos1, b.len1);
        const int shift2 = e.ord == '1' ? (e.pos2 + i1r) - i2r : i1r - (e.pos2 + i2r);
        if (shift2 < 0) {
            o.ord2 = o.ord1 == '1' ? '2' : '1';
            o.pos2 = -shift2;
            if (o.pos2 >= b.len2) return false;
            o.len2 = std::min(a.len2, b.len2 - o.pos2);
        } else {
            o.ord2 = o.ord1 == '1' ? '1' : '2';
            o.pos2 = shift2;
            if (o.pos2 >= a.len2) return false;
            o.len2 = std::min(a.len2 - o.pos2, b.len2);
        }
        o.perc = std::min(perc_max(o.len1 + o.len2, a.len1 + a.len2, b.len1 + b.len2), 100);
    }
    FNO_REQUIRE(o.perc >= 0 && o.perc <= 100);  // :562
    return true;
}

// ---- FNO=1 ------------------------------------------------------------------------------------------------
enum Kind : uint8_t { kCopied = 0, kU2SR = 1, kV2SR = 2, kSR2SR = 3 };

struct Item {  // one line to deduce: an edge and the (super-)reads standing in for its two ends
    const hc_fno_edge* edge;
    uint32_t sr1, sr2;  // super-read indices (kU2SR uses sr2 only, kV2SR sr1 only)
    Kind kind;
    uint8_t nonedge;  // edge.score == 0
};

class Fno1 {
public:
    explicit Fno1(const hc_fno1_input& in) : in_(in) {}

    void run(hc_fno_output& out) {
        threads_ = thread_count(in_.n_threads);
        const bool timing = getenv("HC_FNO_TIMING") != nullptr;
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto t0 = now();
        check_input();
        index_subreads();
        auto t0b = now();
        if (walk_on_device(out)) {  // takes the caller's arrays as they are: nodes_to_SR, adj_out and the non-edge test happen there
            if (timing)
                fprintf(stderr, "hc_fno1_run: subread maps sorted %.3f s, the whole rest on the device %.3f s\n", std::chrono::duration<double>(t0b - t0).count(),
                        std::chrono::duration<double>(now() - t0b).count());
            return;
        }
        build_nodes_to_sr();
        auto t1 = now();
        if (timing)
            fprintf(stderr, "hc_fno1_run: index: subread maps sorted %.3f s, nodes_to_SR %.3f s\n", std::chrono::duration<double>(t0b - t0).count(),
                    std::chrono::duration<double>(t1 - t0b).count());
        collect_work();
        walk();
        auto t2 = now();
        deduce_and_emit(out);
        auto t3 = now();
        if (timing)
            fprintf(stderr, "hc_fno1_run: index %.3f s, walk %.3f s (%zu items), deduce+sort+emit %.3f s\n",
                    std::chrono::duration<double>(t1 - t0).count(), std::chrono::duration<double>(t2 - t1).count(), items_.size(),
                    std::chrono::duration<double>(t3 - t2).count());
    }

private:
    struct Combo {  // one (edge, super-read, super-read) combination that competes for its pair of ids
        uint64_t lo, hi;  // the unordered pair of new ids
        uint32_t edge;    // walk position of the edge
        uint32_t nth;     // position of the combination inside the edge's loops
        uint32_t sr1, sr2;
        Kind kind;
        bool before(const Combo& o) const { return edge != o.edge ? edge < o.edge : nth < o.nth; }
    };
    const hc_fno1_input& in_;
    uint64_t dup_half_ = 0;  // --add_duplicates: n_nodes / 2, else 0
    unsigned threads_ = 1;
    std::vector<hc_fno_subread> sub_sorted_;  // subreads of every super-read, sorted by node within the super-read
    std::vector<uint64_t> n2s_off_;           // nodes_to_SR as CSR
    std::vector<uint32_t> n2s_;
    // every edge updateOverlap is called on, in the reference's order: four runs of edges — adj_out and branching_edges where
    // the caller keeps them, the stored non-edges that pass :702 (copied together), the inclusion-induced edges
    hc::FnoEdgeSpan work_[4] = {{nullptr, 0}, {nullptr, 0}, {nullptr, 0}, {nullptr, 0}};
    uint64_t n_work_ = 0;
    std::vector<hc_fno_edge, hc::DefaultInitAllocator<hc_fno_edge>> kept_nonedges_;
    const hc_fno_edge* work_at(uint64_t i) const {
        for (int k = 0; k < 3; ++k) {
            if (i < work_[k].n) return work_[k].p + i;
            i -= work_[k].n;
        }
        return work_[3].p + i;
    }
    std::vector<Item, hc::DefaultInitAllocator<Item>> items_;  // sized once, filled by the threads: not zero-filled first
    std::vector<hc_fno_edge> induced_;  // inclusion-induced edges (owned)
    bool adj_identity_ = false;         // graph_edges lie vertex by vertex already (the documented order): adj_ is not built
    std::vector<uint64_t> adj_off_;     // OverlapGraph::adj_out as CSR over graph_edges (stable by v1)
    std::vector<uint32_t> adj_;

    void check_input() {
        if (in_.n_srs >= 0xFFFFFFFFull) throw FatalError{HC_ERR_ARG, "too many super-reads"};
        // single_SR_vec comes before paired_SR_vec (:893-906): the order of nodes_to_SR decides which combination of a
        // pair of super-reads is met first, i.e. it is observable
        for (uint64_t i = 1; i < in_.n_srs; ++i)
            if (in_.srs && in_.srs[i - 1].paired && !in_.srs[i].paired)
                throw FatalError{HC_ERR_ARG, "super-reads must be listed single-end first, then paired (single_SR_vec, paired_SR_vec)"};
        if (in_.n_nodes >= ((uint64_t)1 << 32)) throw FatalError{HC_ERR_ARG, "too many vertices"};
        if ((in_.n_nodes && !in_.nodes) || (in_.n_srs && (!in_.srs || !in_.clique_off || !in_.subread_off)))
            throw FatalError{HC_ERR_ARG, "null array"};
        if ((in_.n_graph_edges && !in_.graph_edges) || (in_.n_branching_edges && !in_.branching_edges) ||
            (in_.n_inclusion_groups && (!in_.inclusion_off || !in_.inclusion_edges)))
            throw FatalError{HC_ERR_ARG, "null array"};
        if (in_.flags & HC_FNO_ADD_DUPLICATES) {  // a vertex per read and strand: r and r + n_nodes / 2 are one Read (src/ViralQuasispecies.cpp:246-270)
            if (in_.n_nodes & 1) throw FatalError{HC_ERR_ARG, "HC_FNO_ADD_DUPLICATES: an even number of vertices (every read on both strands) is expected"};
            dup_half_ = in_.n_nodes / 2;
            for (uint64_t r = 0; r < dup_half_; ++r) {
                const hc_fno_read &a = in_.nodes[r], &b = in_.nodes[r + dup_half_];
                if (a.len1 != b.len1 || a.len2 != b.len2 || (a.paired != 0) != (b.paired != 0))
                    throw FatalError{HC_ERR_ARG, "HC_FNO_ADD_DUPLICATES: vertices r and r + n_nodes / 2 must describe the same read (lengths, paired)"};
            }
            if (!(in_.flags & HC_FNO_OPTIMIZE) && in_.n_nonedges && in_.nonedges)  // :672-675: a line's vertices are taken by its orientations
                parallel_chunks(in_.n_nonedges, threads_, [&](uint64_t b, uint64_t e, unsigned) {
                    for (uint64_t i = b; i < e; ++i) {
                        const hc_fno_edge& x = in_.nonedges[i];
                        if ((x.v1 < dup_half_) != (x.ori1 != 0) || (x.v2 < dup_half_) != (x.ori2 != 0))
                            throw FatalError{HC_ERR_ARG, "HC_FNO_ADD_DUPLICATES: a stored non-edge's vertex is not on the strand its orientation names"};
                    }
                });
        }
    }

    void index_subreads() {
        const uint64_t total = in_.n_srs ? in_.subread_off[in_.n_srs] : 0;
        sub_sorted_.assign(in_.subreads, in_.subreads + total);
        parallel_chunks(in_.n_srs, threads_, [&](uint64_t b, uint64_t e, unsigned) {
            for (uint64_t i = b; i < e; ++i)
                std::sort(sub_sorted_.begin() + in_.subread_off[i], sub_sorted_.begin() + in_.subread_off[i + 1],
                          [](const hc_fno_subread& x, const hc_fno_subread& y) { return x.node < y.node; });
        });
    }

    const hc_fno_subread& subread(uint32_t sr, uint64_t node) const {  // Read::get_subread_info, src/Read.h:294-301
        const hc_fno_subread* b = sub_sorted_.data() + in_.subread_off[sr];
        const hc_fno_subread* e = sub_sorted_.data() + in_.subread_off[sr + 1];
        FNO_REQUIRE(b != e);  // assert(!subreadMap.empty())
        const hc_fno_subread* it = std::lower_bound(b, e, node, [](const hc_fno_subread& x, uint64_t n) { return x.node < n; });
        if (it == e || it->node != node) ref_abort("subreadMap.at(node): vertex is not part of the super-read");
        return *it;
    }

    // the pair of findCliqueIndex calls of :95-109 etc.: start offsets of `node` inside super-read `sr`
    void clique_indices(uint64_t node, uint32_t sr, bool read_paired, int& left, int& right) const {
        const hc_fno_subread& s = subread(sr, node);
        FNO_REQUIRE(s.index1 >= 0 && s.startpos1 >= 0);
        FNO_REQUIRE(!(s.index1 > 0 && s.startpos1 > 0));
        left = s.index1 - s.startpos1;
        if (!in_.srs[sr].paired && !read_paired) {
            right = left;
            return;
        }
        FNO_REQUIRE(s.index2 >= 0 && s.startpos2 >= 0);
        if (in_.srs[sr].paired) FNO_REQUIRE(!(s.index2 > 0 && s.startpos2 > 0));
        right = s.index2 - s.startpos2;
    }

    void build_nodes_to_sr() {  // :893-906
        n2s_off_.assign(in_.n_nodes + 1, 0);
        for (uint64_t i = 0; i < in_.n_srs; ++i) {
            FNO_REQUIRE(in_.clique_off[i + 1] > in_.clique_off[i]);  // get_sorted_clique asserts size() > 0
            for (uint64_t k = in_.clique_off[i]; k < in_.clique_off[i + 1]; ++k) {
                if (in_.clique_nodes[k] >= in_.n_nodes) ref_abort("nodes_to_SR.at(node)");
                ++n2s_off_[in_.clique_nodes[k] + 1];
            }
        }
        for (uint64_t v = 0; v < in_.n_nodes; ++v) n2s_off_[v + 1] += n2s_off_[v];
        n2s_.resize(n2s_off_[in_.n_nodes]);
        std::vector<uint64_t> cur(n2s_off_.begin(), n2s_off_.end() - 1);
        for (uint64_t i = 0; i < in_.n_srs; ++i)
            for (uint64_t k = in_.clique_off[i]; k < in_.clique_off[i + 1]; ++k) n2s_[cur[in_.clique_nodes[k]]++] = (uint32_t)i;
    }

    void build_adjacency() {  // only what checkEdge needs
        const uint64_t G = in_.n_graph_edges;
        const hc_fno_edge* ge = in_.graph_edges;
        adj_off_.assign(in_.n_nodes + 1, 0);
        // adj_out "vertex by vertex" (hcfno.h) is sorted by v1: every thread checks its stretch and, where v1 steps up, writes the
        // offsets of the vertices in between; anything else takes the counting sort
        std::atomic<bool> sorted{true};
        parallel_chunks(G, threads_, [&](uint64_t b, uint64_t e, unsigned) {
            for (uint64_t i = b; i < e; ++i) {
                if (ge[i].v1 >= in_.n_nodes || ge[i].v2 >= in_.n_nodes) ref_abort("edge vertex out of range");
                if (i && ge[i - 1].v1 > ge[i].v1) sorted.store(false, std::memory_order_relaxed);
            }
        });
        adj_identity_ = sorted.load();
        if (adj_identity_) {
            parallel_chunks(G + 1, threads_, [&](uint64_t b, uint64_t e, unsigned) {
                for (uint64_t i = b; i < e; ++i) {  // offsets of the vertices in (v1[i - 1], v1[i]], i = G: up to n_nodes
                    const uint64_t from = i ? ge[i - 1].v1 + 1 : 0, to = i < G ? ge[i].v1 : in_.n_nodes;
                    for (uint64_t v = from; v <= to; ++v) adj_off_[v] = i;
                }
            });
            return;
        }
        for (uint64_t i = 0; i < G; ++i) ++adj_off_[ge[i].v1 + 1];
        for (uint64_t v = 0; v < in_.n_nodes; ++v) adj_off_[v + 1] += adj_off_[v];
        adj_.resize(G);
        std::vector<uint64_t> cur(adj_off_.begin(), adj_off_.end() - 1);
        for (uint64_t i = 0; i < G; ++i) adj_[cur[ge[i].v1]++] = (uint32_t)i;
    }

    const hc_fno_edge& adj_edge(uint64_t k) const { return in_.graph_edges[adj_identity_ ? k : adj_[k]]; }
    double check_edge(uint64_t v, uint64_t w) const {  // OverlapGraph::checkEdge(v, w, true), src/OverlapGraph.cpp:233-259
        for (uint64_t k = adj_off_[v]; k < adj_off_[v + 1]; ++k)
            if (adj_edge(k).v2 == w) return adj_edge(k).score;
        for (uint64_t k = adj_off_[w]; k < adj_off_[w + 1]; ++k)
            if (adj_edge(k).v2 == v) return adj_edge(k).score;
        return -1.0;
    }

    // findInclusionOverlaps (:816-887): the inclusion-induced edges, checked against adj_out (build_adjacency must have run if there are groups)
    void collect_induced() {
        induced_.clear();
        // u->w and w->v both inclusions  =>  try u->v
        uint64_t n_induced = 0;
        for (uint64_t g = 0; g < in_.n_inclusion_groups; ++g) {
            const uint64_t l = in_.inclusion_off[g + 1] - in_.inclusion_off[g];
            n_induced += l * (l - (l ? 1 : 0)) / 2;
        }
        induced_.reserve(n_induced);  // pointers into induced_ must stay valid
        for (uint64_t g = 0; g < in_.n_inclusion_groups; ++g) {
            const hc_fno_edge* grp = in_.inclusion_edges + in_.inclusion_off[g];
            const unsigned l = (unsigned)(in_.inclusion_off[g + 1] - in_.inclusion_off[g]);
            for (unsigned i = 0; i < l; ++i)
                for (unsigned j = i + 1; j < l; ++j) {
                    const hc_fno_edge &e1 = grp[i], &e2 = grp[j];
                    const hc_fno_edge *head, *tail;  // head->v1 = u, tail->v2 = v
                    if (e1.v1 == e2.v1) continue;
                    if (e1.v1 == e2.v2) {
                        head = &e2;
                        tail = &e1;
                    } else if (e1.v2 == e2.v1) {
                        head = &e1;
                        tail = &e2;
                    } else {
                        FNO_REQUIRE(e1.v2 == e2.v2);
                        continue;
                    }
                    if (head->v1 >= in_.n_nodes || tail->v2 >= in_.n_nodes) ref_abort("edge vertex out of range");
                    const hc_fno_read &r1 = in_.nodes[head->v1], &r2 = in_.nodes[tail->v2];
                    if (r1.paired || r2.paired) continue;
                    const unsigned L1 = r1.len1, L2 = r2.len1;
                    const int len = (int)std::min(L1 - (unsigned)head->pos1, L2);
                    FNO_REQUIRE(std::min(L1, L2) != 0);  // the reference divides by it
                    const int perc = (int)((unsigned)(100 * len) / std::min(L1, L2));
                    FNO_REQUIRE(in_.edge_threshold == 0 || in_.edge_threshold == -1 || in_.edge_threshold > 0);  // Edge ctor, src/Edge.h:46
                    FNO_REQUIRE(len > 0);                                                                        // Edge::set_len
                    hc_fno_edge ne;
                    memset(&ne, 0, sizeof ne);
                    ne.v1 = head->v1;
                    ne.v2 = tail->v2;
                    ne.score = in_.edge_threshold;
                    ne.pos1 = head->pos1;
                    ne.len1 = len;
                    ne.perc = perc;
                    ne.ord = '-';
                    ne.ori1 = head->ori1;
                    ne.ori2 = tail->ori2;
                    if (check_edge(ne.v1, ne.v2) == -1) induced_.push_back(ne);
                }
        }
    }

    // the edges updateOverlap sees, in order: adj_out, branching_edges, the stored non-edges that pass :702, the
    // inclusion-induced edges (:612-630, :635-813, :816-887)
    void collect_work() {
        if (in_.n_graph_edges + in_.n_branching_edges + in_.n_nonedges >= 0xFFFFFFF0ull) throw FatalError{HC_ERR_ARG, "too many edges"};
        const bool timing = getenv("HC_FNO_TIMING") != nullptr;
        auto tl = std::chrono::steady_clock::now();
        auto lap = [&](const char* what) {
            if (!timing) return;
            const auto t = std::chrono::steady_clock::now();
            fprintf(stderr, "hc_fno1_run: edges in walk order: %s %.3f s\n", what, std::chrono::duration<double>(t - tl).count());
            tl = t;
        };
        work_[0] = {in_.graph_edges, in_.n_graph_edges};
        work_[1] = {in_.branching_edges, in_.n_branching_edges};
        const bool use_nonedges = !(in_.flags & HC_FNO_OPTIMIZE) && in_.n_nonedges;  // :914
        if (use_nonedges || in_.n_inclusion_groups) build_adjacency();
        lap("adjacency");
        if (use_nonedges) {
            if (!in_.nonedges) throw FatalError{HC_ERR_ARG, "null array"};
            const unsigned T = (unsigned)std::min<uint64_t>(threads_, in_.n_nonedges);
            std::vector<std::vector<uint32_t>> kept(T);  // indices into nonedges
            parallel_chunks(in_.n_nonedges, T, [&](uint64_t b, uint64_t e, unsigned t) {
                for (uint64_t i = b; i < e; ++i) {
                    const hc_fno_edge* ed = in_.nonedges + i;
                    if (ed->score != 0) throw FatalError{HC_ERR_ARG, "a stored non-edge must carry score 0"};
                    FNO_REQUIRE(ed->len1 > 0 && ed->len2 >= 0);  // Edge::set_len, src/Edge.h:212-214
                    if (ed->v1 >= in_.n_nodes || ed->v2 >= in_.n_nodes) ref_abort("edge vertex out of range");
                    if (check_edge(ed->v1, ed->v2) > 0) continue;  // :702
                    kept[t].push_back((uint32_t)i);
                }
            });
            std::vector<uint64_t> at(T + 1, 0);
            for (unsigned t = 0; t < T; ++t) at[t + 1] = at[t] + kept[t].size();
            if (dup_half_) {  // every kept line twice: itself, then the same overlap seen from the other strand (:699-793)
                if (2 * at[T] + in_.n_graph_edges + in_.n_branching_edges >= 0xFFFFFFF0ull) throw FatalError{HC_ERR_ARG, "too many edges"};
                kept_nonedges_.resize(2 * at[T]);
                parallel_chunks(T, T, [&](uint64_t tb, uint64_t te, unsigned) {
                    for (uint64_t t = tb; t < te; ++t)
                        for (size_t k = 0; k < kept[t].size(); ++k) {
                            const hc_fno_edge& e = in_.nonedges[kept[t][k]];
                            hc_fno_edge& o = kept_nonedges_[2 * (at[t] + k) + 1];
                            kept_nonedges_[2 * (at[t] + k)] = e;
                            const int rc = hc::fno_mirror_nonedge(e, in_.nodes[e.v1], in_.nodes[e.v2], dup_half_, &o);
                            if (rc == 2) throw FatalError{HC_ERR_ARG, "HC_FNO_ADD_DUPLICATES: a stored non-edge's vertex is not on the strand its orientation names"};
                            if (rc) ref_abort("overlap.get_ord() == \"2\" (two paired reads, --add_duplicates)");
                        }
                });
            } else {
                kept_nonedges_.resize(at[T]);
                parallel_chunks(T, T, [&](uint64_t tb, uint64_t te, unsigned) {
                    for (uint64_t t = tb; t < te; ++t)
                        for (size_t k = 0; k < kept[t].size(); ++k) kept_nonedges_[at[t] + k] = in_.nonedges[kept[t][k]];
                });
            }
            work_[2] = {kept_nonedges_.data(), kept_nonedges_.size()};
        }
        lap("stored non-edges");
        collect_induced();
        work_[3] = {induced_.data(), induced_.size()};
        lap("inclusion-induced edges");
        n_work_ = work_[0].n + work_[1].n + work_[2].n + work_[3].n;
        if (n_work_ >= 0xFFFFFFFFull) throw FatalError{HC_ERR_ARG, "too many edges"};
    }

    static uint64_t pair_hash(uint64_t a, uint64_t b) {
        uint64_t x = a * 0x9E3779B97F4A7C15ull ^ (b + 0x7F4A7C15ull + (a << 6) + (a >> 2));
        x ^= x >> 32;
        x *= 0xD6E8FEB86659FD93ull;
        x ^= x >> 29;
        return x;
    }

    // updateOverlap's case analysis (:44,69,151,231) without the arithmetic.  The reference keeps, per unordered
    // pair of new ids, the first combination it meets (overlaps_found); here every combination is stamped with its
    // place in that order and each hash partition keeps the earliest one per pair.
    void walk() {
        const bool timing = getenv("HC_FNO_TIMING") != nullptr;
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto lap = [&](const char* what, std::chrono::steady_clock::time_point& since) {
            if (!timing) return;
            const auto t = now();
            fprintf(stderr, "hc_fno1_run: walk: %s %.3f s\n", what, std::chrono::duration<double>(t - since).count());
            since = t;
        };
        auto tw = now();
        const uint64_t E = n_work_;
        const unsigned T = (unsigned)std::min<uint64_t>(threads_, E ? E : 1);
        const unsigned P = T;  // partitions
        std::vector<std::vector<std::vector<Combo>>> scattered(T, std::vector<std::vector<Combo>>(P));
        std::vector<std::vector<Item>> direct(T);
        const uint64_t bound = in_.new_read_count;
        parallel_chunks(E, T, [&](uint64_t b, uint64_t e_end, unsigned t) {
            auto& parts = scattered[t];
            auto push = [&](uint64_t id1, uint64_t id2, uint32_t edge, uint32_t nth, uint32_t sr1, uint32_t sr2, Kind kind) {
                Combo c{id1 < id2 ? id1 : id2, id1 < id2 ? id2 : id1, edge, nth, sr1, sr2, kind};
                if (c.lo >= bound) ref_abort("overlaps_found.at(id): id >= new_read_count");
                parts[pair_hash(c.lo, c.hi) % P].push_back(c);
            };
            for (uint64_t i = b; i < e_end; ++i) {
                const hc_fno_edge* e = work_at(i);
                if (e->v1 >= in_.n_nodes || e->v2 >= in_.n_nodes) ref_abort("edge vertex out of range");
                const uint64_t u = e->v1, v = e->v2;
                const bool vu = in_.nodes[u].visited, vv = in_.nodes[v].visited;
                if (!vu && !vv) {
                    direct[t].push_back(Item{e, 0, 0, kCopied, (uint8_t)(e->score == 0)});
                } else if (!vu) {
                    const uint64_t id1 = in_.nodes[u].id;
                    for (uint64_t k = n2s_off_[v]; k < n2s_off_[v + 1]; ++k) {
                        const uint64_t id2 = in_.srs[n2s_[k]].id;
                        FNO_REQUIRE(id1 != id2);
                        push(id1, id2, (uint32_t)i, (uint32_t)(k - n2s_off_[v]), 0, n2s_[k], kU2SR);
                    }
                } else if (!vv) {
                    const uint64_t id1 = in_.nodes[v].id;
                    for (uint64_t k = n2s_off_[u]; k < n2s_off_[u + 1]; ++k) {
                        const uint64_t id2 = in_.srs[n2s_[k]].id;
                        FNO_REQUIRE(id1 != id2);
                        push(id1, id2, (uint32_t)i, (uint32_t)(k - n2s_off_[u]), n2s_[k], 0, kV2SR);
                    }
                } else {
                    const uint64_t n2 = n2s_off_[v + 1] - n2s_off_[v];
                    if ((n2s_off_[u + 1] - n2s_off_[u]) * n2 >= 0xFFFFFFFFull) throw FatalError{HC_ERR_ARG, "too many super-reads share one vertex"};
                    for (uint64_t k1 = n2s_off_[u]; k1 < n2s_off_[u + 1]; ++k1) {
                        const uint64_t id1 = in_.srs[n2s_[k1]].id;
                        for (uint64_t k2 = n2s_off_[v]; k2 < n2s_off_[v + 1]; ++k2) {
                            const uint64_t id2 = in_.srs[n2s_[k2]].id;
                            if (id1 == id2) continue;
                            push(id1, id2, (uint32_t)i, (uint32_t)((k1 - n2s_off_[u]) * n2 + (k2 - n2s_off_[v])), n2s_[k1], n2s_[k2], kSR2SR);
                        }
                    }
                }
            }
        });
        lap("combinations dealt by pair", tw);
        // per partition: the earliest combination of every pair
        std::vector<std::vector<Item>> winners(P);
        parallel_chunks(P, P, [&](uint64_t pb, uint64_t pe, unsigned) {
            for (uint64_t p = pb; p < pe; ++p) {
                size_t n = 0;
                for (unsigned t = 0; t < T; ++t) n += scattered[t][p].size();
                if (!n) continue;
                size_t cap = 16;
                while (cap < 2 * n) cap <<= 1;
                std::vector<const Combo*> table(cap, nullptr);
                size_t distinct = 0;
                for (unsigned t = 0; t < T; ++t)
                    for (const Combo& c : scattered[t][p]) {
                        size_t h = (size_t)(pair_hash(c.lo, c.hi) >> 20) & (cap - 1);
                        for (;;) {
                            const Combo*& slot = table[h];
                            if (!slot) {
                                slot = &c;
                                ++distinct;
                                break;
                            }
                            if (slot->lo == c.lo && slot->hi == c.hi) {
                                if (c.before(*slot)) slot = &c;
                                break;
                            }
                            h = (h + 1) & (cap - 1);
                        }
                    }
                winners[p].reserve(distinct);
                for (const Combo* c : table)
                    if (c) winners[p].push_back(Item{work_at(c->edge), c->sr1, c->sr2, c->kind, (uint8_t)(work_at(c->edge)->score == 0)});
            }
        });
        lap("earliest combination per pair", tw);
        size_t total = 0;
        for (auto& d : direct) total += d.size();
        for (auto& w : winners) total += w.size();
        items_.resize(total);  // leaves the items uninitialised; every source list is copied to its place by some thread
        std::vector<const std::vector<Item>*> lists;
        for (auto& d : direct) lists.push_back(&d);
        for (auto& w : winners) lists.push_back(&w);
        std::vector<size_t> at(lists.size() + 1, 0);
        for (size_t k = 0; k < lists.size(); ++k) at[k + 1] = at[k] + lists[k]->size();
        parallel_chunks(lists.size(), threads_, [&](uint64_t b, uint64_t e, unsigned) {
            for (uint64_t k = b; k < e; ++k)
                if (!lists[k]->empty()) memcpy(items_.data() + at[k], lists[k]->data(), lists[k]->size() * sizeof(Item));
        });
        lap("one list", tw);
    }

    // one line of overlaps.txt (without the newline) into p; returns the end, or nullptr if no line results
    char* deduce(const Item& it, char* p, uint64_t counts[4]) const {
        const hc_fno_edge& e = *it.edge;
        const hc_fno_read &n1 = in_.nodes[e.v1], &n2 = in_.nodes[e.v2];
        char ori1 = '+', ori2 = '+';
        if ((in_.flags & HC_FNO_RESOLVE_ORIENTATIONS) && it.nonedge) {  // :35-38
            ori1 = ((e.ori1 != 0) == (n1.orientation != 0)) ? '+' : '-';
            ori2 = ((e.ori2 != 0) == (n2.orientation != 0)) ? '+' : '-';
        }
        uint64_t first_id, second_id;
        Induced o;
        if (it.kind == kCopied) {  // :44-68
            FNO_REQUIRE(e.perc >= 0);
            first_id = n1.id;
            second_id = n2.id;
            o = Induced{e.pos1, e.pos2, e.perc, e.len1, e.len2, '1', (char)e.ord, n1.paired ? 'p' : 's', n2.paired ? 'p' : 's'};
        } else {
            int i1l = 0, i1r = 0, i2l = 0, i2r = 0;
            ReadFacts a, b;
            uint64_t ida, idb;
            if (it.kind == kU2SR) {
                a = facts(n1);
                ida = n1.id;
            } else {
                a = facts(in_.srs[it.sr1]);
                ida = in_.srs[it.sr1].id;
                clique_indices(e.v1, it.sr1, n1.paired, i1l, i1r);
            }
            if (it.kind == kV2SR) {
                b = facts(n2);
                idb = n2.id;
            } else {
                b = facts(in_.srs[it.sr2]);
                idb = in_.srs[it.sr2].id;
                clique_indices(e.v2, it.sr2, n2.paired, i2l, i2r);
            }
            if (!induced_overlap(a, b, i1l, i1r, i2l, i2r, e, o)) return nullptr;
            if (o.ord1 == '1') {
                first_id = ida;
                second_id = idb;
            } else {
                first_id = idb;
                second_id = ida;
                std::swap(o.type1, o.type2);
            }
        }
        FNO_REQUIRE(o.ord2 == '-' || o.ord2 == '1' || o.ord2 == '2');
        if ((in_.flags & HC_FNO_NO_INCLUSIONS) && o.perc == 100) return nullptr;
        ++counts[it.kind];
        p = put_u64(p, first_id);
        *p++ = '\t';
        p = put_u64(p, second_id);
        *p++ = '\t';
        p = put_i32(p, o.pos1);
        *p++ = '\t';
        p = put_i32(p, o.pos2);
        *p++ = '\t';
        *p++ = o.ord2;
        *p++ = '\t';
        *p++ = ori1;
        *p++ = '\t';
        *p++ = ori2;
        *p++ = '\t';
        p = put_i32(p, o.perc);
        *p++ = '\t';
        *p++ = '0';
        *p++ = '\t';
        p = put_i32(p, o.len1);
        *p++ = '\t';
        p = put_i32(p, o.len2);
        *p++ = '\t';
        *p++ = o.type1;
        *p++ = '\t';
        *p++ = o.type2;
        return p;
    }

    // ---- the device form --------------------------------------------------------------------
    // The look-ups of deduce() — ids, lengths, findCliqueIndex offsets, orientation signs — stay with the host threads
    // (they chase the caller's arrays); the arithmetic, the ordering, the unique and the text are the device's.
    void build_item(const Item& it, hc::FnoItem& o) const {
        const hc_fno_edge& e = *it.edge;
        const hc_fno_read &n1 = in_.nodes[e.v1], &n2 = in_.nodes[e.v2];
        memset(&o, 0, sizeof o);
        o.ori1 = o.ori2 = '+';
        if ((in_.flags & HC_FNO_RESOLVE_ORIENTATIONS) && it.nonedge) {  // :35-38
            o.ori1 = ((e.ori1 != 0) == (n1.orientation != 0)) ? '+' : '-';
            o.ori2 = ((e.ori2 != 0) == (n2.orientation != 0)) ? '+' : '-';
        }
        o.kind = (uint8_t)it.kind;
        o.e_ord = (uint8_t)e.ord;
        if (it.kind == kCopied) {  // :44-68
            FNO_REQUIRE(e.perc >= 0);
            o.ida = n1.id;
            o.idb = n2.id;
            o.v[0] = e.pos1; o.v[1] = e.pos2; o.v[2] = e.perc; o.v[3] = e.len1; o.v[4] = e.len2;
            o.a_paired = n1.paired != 0;
            o.b_paired = n2.paired != 0;
            return;
        }
        int i1l = 0, i1r = 0, i2l = 0, i2r = 0;
        ReadFacts a, b;
        if (it.kind == kU2SR) {
            a = facts(n1);
            o.ida = n1.id;
        } else {
            a = facts(in_.srs[it.sr1]);
            o.ida = in_.srs[it.sr1].id;
            clique_indices(e.v1, it.sr1, n1.paired, i1l, i1r);
        }
        if (it.kind == kV2SR) {
            b = facts(n2);
            o.idb = n2.id;
        } else {
            b = facts(in_.srs[it.sr2]);
            o.idb = in_.srs[it.sr2].id;
            clique_indices(e.v2, it.sr2, n2.paired, i2l, i2r);
        }
        o.v[0] = e.pos1; o.v[1] = e.pos2;
        o.v[2] = i1l; o.v[3] = i1r; o.v[4] = i2l; o.v[5] = i2r;
        o.v[6] = a.len1; o.v[7] = a.len2; o.v[8] = b.len1; o.v[9] = b.len2;
        o.a_paired = a.paired;
        o.b_paired = b.paired;
    }

    // false: the device met something the host path has to report (a stop of the reference) or to handle (numbers
    // beyond the keys' range); nothing was written
    bool deduce_on_device(hc_fno_output& out) {
        const uint64_t n = items_.size();
        if (n == 0 || n >= 0x7FFFFFF0ull) return false;
        const bool timing = getenv("HC_FNO_TIMING") != nullptr;
        auto now = [] { return std::chrono::steady_clock::now(); };
        const auto t0 = now();
        // 64 bytes per combination, written once by the threads below: 2 MiB pages where the system grants them (the
        // first touch of 0.7 GB in 4 KiB pages costs more than the look-ups)
        struct Freed {
            void operator()(hc::FnoItem* p) const { free(p); }
        };
        const size_t huge = (size_t)2 << 20, bytes = (n * sizeof(hc::FnoItem) + huge - 1) & ~(huge - 1);
        void* mem = nullptr;
        if (posix_memalign(&mem, huge, bytes) != 0) throw FatalError{HC_ERR_NOMEM, "find-next-overlaps: out of memory"};
        madvise(mem, bytes, MADV_HUGEPAGE);
        std::unique_ptr<hc::FnoItem[], Freed> h_items((hc::FnoItem*)mem);
        parallel_chunks(n, threads_, [&](uint64_t b, uint64_t e, unsigned) {
            for (uint64_t i = b; i < e; ++i) build_item(items_[i], h_items[i]);
        });
        const auto t1 = now();
        uint64_t counters[5] = {0, 0, 0, 0, 0};
        double seconds[2] = {0, 0};
        auto text_of = [&](uint64_t bytes) { return sized_text(out, bytes); };
        try {
            if (!hc::fno_lines_on_device(h_items.get(), n, (in_.flags & HC_FNO_NO_INCLUSIONS) != 0, text_of, counters, seconds)) return false;
        } catch (const FatalError& e) {
            if (e.status != HC_ERR_NOMEM) throw;
            return false;  // a batch beyond the device's memory: the host threads take it
        }
        memset(&out.counters, 0, sizeof out.counters);
        out.counters.copied = counters[kCopied];
        out.counters.u2sr = counters[kU2SR];
        out.counters.v2sr = counters[kV2SR];
        out.counters.sr2sr = counters[kSR2SR];
        out.counters.n_lines = counters[4];
        out.on_device = true;
        if (timing)
            fprintf(stderr, "hc_fno1_run (device): look-ups on the host %.3f s, copy + deduce + 4 sorts + unique + scan %.3f s, text %.3f s, device blocks released %.3f s\n",
                    std::chrono::duration<double>(t1 - t0).count(), seconds[0], seconds[1],
                    std::chrono::duration<double>(now() - t1).count() - seconds[0] - seconds[1]);
        return true;
    }

    // FNO=1 whole on the device (hc_fno_items.h: fno1_walk_on_device): the edges in walk order go over as they lie in the caller's
    // arrays (the kept non-edges gathered first), with nodes_to_SR and the sorted subread maps; false = the host form runs.
    bool walk_on_device(hc_fno_output& out) {
        if (const char* e = getenv("HC_FNO_WALK"))
            if (strcmp(e, "host") == 0) return false;
        const bool use_nonedges = !(in_.flags & HC_FNO_OPTIMIZE) && in_.n_nonedges;  // :914
        const uint64_t E = in_.n_graph_edges + in_.n_branching_edges + (use_nonedges ? in_.n_nonedges : 0);
        if (!hc::fno_device_wanted(E)) return false;
        if (use_nonedges && !in_.nonedges) throw FatalError{HC_ERR_ARG, "null array"};
        const bool timing = getenv("HC_FNO_TIMING") != nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        for (uint64_t i = 0; i < in_.n_srs; ++i) FNO_REQUIRE(in_.clique_off[i + 1] > in_.clique_off[i]);  // get_sorted_clique asserts size() > 0
        if (in_.n_inclusion_groups) {  // few edges, on the host: they need checkEdge, i.e. adj_out here as well
            build_adjacency();
            collect_induced();
        }
        hc::FnoWalkHost h{};
        h.graph = {in_.graph_edges, in_.n_graph_edges};
        h.branching = {in_.branching_edges, in_.n_branching_edges};
        h.nonedges = {in_.nonedges, use_nonedges ? in_.n_nonedges : 0};
        h.induced = {induced_.data(), induced_.size()};
        h.nodes = in_.nodes;
        h.n_nodes = in_.n_nodes;
        h.srs = in_.srs;
        h.n_srs = in_.n_srs;
        h.clique_off = in_.clique_off;
        h.clique_nodes = in_.clique_nodes;
        h.subread_off = in_.subread_off;
        h.subreads = sub_sorted_.data();
        h.new_read_count = in_.new_read_count;
        h.resolve_orientations = (in_.flags & HC_FNO_RESOLVE_ORIENTATIONS) != 0;
        h.no_inclusions = (in_.flags & HC_FNO_NO_INCLUSIONS) != 0;
        h.dup_half = use_nonedges ? dup_half_ : 0;
        uint64_t counters[5] = {0, 0, 0, 0, 0}, n_items = 0;
        double seconds[3] = {0, 0, 0};
        auto text_of = [&](uint64_t bytes) { return sized_text(out, bytes); };
        try {
            if (!hc::fno1_walk_on_device(h, text_of, counters, &n_items, seconds)) return false;
        } catch (const FatalError& e) {
            if (e.status != HC_ERR_NOMEM) throw;
            return false;  // beyond the device's memory: the host threads take it
        }
        memset(&out.counters, 0, sizeof out.counters);
        out.counters.copied = counters[kCopied];
        out.counters.u2sr = counters[kU2SR];
        out.counters.v2sr = counters[kV2SR];
        out.counters.sr2sr = counters[kSR2SR];
        out.counters.n_lines = counters[4];
        out.on_device = true;
        out.walk_on_device = true;
        if (timing)
            fprintf(stderr, "hc_fno1_run (device walk): copies + walk + look-ups %.3f s (%llu items), deduce + 4 sorts + unique + scan %.3f s, text %.3f s, buffers released %.3f s\n",
                    seconds[0], (unsigned long long)n_items, seconds[1], seconds[2],
                    std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() - seconds[0] - seconds[1] - seconds[2]);
        return true;
    }

    struct Line {
        const char* p;
        uint32_t n;
    };
    static bool line_less(const Line& a, const Line& b) {  // std::string::compare
        const int c = memcmp(a.p, b.p, a.n < b.n ? a.n : b.n);
        return c ? c < 0 : a.n < b.n;
    }

    void deduce_and_emit(hc_fno_output& out) {
        constexpr size_t kMaxLine = 160;  // 2 x 20 digits + 6 x 11 + separators
        const uint64_t n = items_.size();
        if (hc::fno_device_wanted(n) && deduce_on_device(out)) return;
        const unsigned T = (unsigned)std::min<uint64_t>(threads_, n ? n : 1);
        std::vector<std::unique_ptr<char[]>> arena(T);  // new char[]: not zero-filled, only the bytes written get touched
        std::vector<std::vector<Line>> lines(T);
        std::vector<uint64_t> counts(4 * (size_t)T, 0);
        parallel_chunks(n, T, [&](uint64_t b, uint64_t e, unsigned t) {
            arena[t].reset(new char[(e - b) * kMaxLine]);
            lines[t].reserve(e - b);
            char* p = arena[t].get();
            uint64_t local[4] = {0, 0, 0, 0};  // not counts[] directly: neighbouring threads would share its cache lines
            std::vector<Line>& mine = lines[t];
            for (uint64_t i = b; i < e; ++i) {
                char* q = deduce(items_[i], p, local);
                if (!q) continue;
                mine.push_back(Line{p, (uint32_t)(q - p)});
                p = q;
            }
            for (int k = 0; k < 4; ++k) counts[4 * (size_t)t + k] = local[k];
        });
        memset(&out.counters, 0, sizeof out.counters);
        for (unsigned t = 0; t < T; ++t) {
            out.counters.copied += counts[4 * (size_t)t + kCopied];
            out.counters.u2sr += counts[4 * (size_t)t + kU2SR];
            out.counters.v2sr += counts[4 * (size_t)t + kV2SR];
            out.counters.sr2sr += counts[4 * (size_t)t + kSR2SR];
        }
        // sample sort in std::string order: splitters from a regular sample, every thread deals its lines into
        // the buckets, every bucket is sorted and made unique on its own (equal lines always share a bucket)
        size_t total = 0;
        for (auto& l : lines) total += l.size();
        const unsigned B = total < 4096 ? 1u : T * 4u;
        std::vector<Line> splitters;
        if (B > 1) {
            std::vector<Line> sample;
            const size_t want = (size_t)B * 32, step = std::max<size_t>(1, total / want);
            size_t seen = 0;
            for (auto& l : lines)
                for (size_t i = 0; i < l.size(); ++i, ++seen)
                    if (seen % step == 0) sample.push_back(l[i]);
            std::sort(sample.begin(), sample.end(), line_less);
            for (unsigned k = 1; k < B; ++k) splitters.push_back(sample[k * sample.size() / B]);
        }
        std::vector<std::vector<std::vector<Line>>> dealt(T, std::vector<std::vector<Line>>(B));
        parallel_chunks(T, T, [&](uint64_t tb, uint64_t te, unsigned) {
            for (uint64_t t = tb; t < te; ++t)
                for (const Line& l : lines[t]) {
                    const size_t k = (size_t)(std::upper_bound(splitters.begin(), splitters.end(), l, line_less) - splitters.begin());
                    dealt[t][k].push_back(l);
                }
        });
        std::vector<std::vector<Line>> bucket(B);
        std::vector<size_t> bucket_bytes(B, 0);
        parallel_chunks(B, threads_, [&](uint64_t bb, uint64_t be, unsigned) {
            for (uint64_t k = bb; k < be; ++k) {
                std::vector<Line>& v = bucket[k];
                size_t m = 0;
                for (unsigned t = 0; t < T; ++t) m += dealt[t][k].size();
                v.reserve(m);
                for (unsigned t = 0; t < T; ++t) v.insert(v.end(), dealt[t][k].begin(), dealt[t][k].end());
                std::sort(v.begin(), v.end(), line_less);
                size_t kept = 0, bytes = 0;
                for (size_t i = 0; i < v.size(); ++i) {
                    if (kept && v[i].n == v[kept - 1].n && memcmp(v[i].p, v[kept - 1].p, v[i].n) == 0) continue;
                    v[kept++] = v[i];
                    bytes += v[i].n + 1;
                }
                v.resize(kept);
                bucket_bytes[k] = bytes;
            }
        });
        std::vector<size_t> at(B + 1, 0);
        size_t n_lines = 0;
        for (unsigned k = 0; k < B; ++k) {
            at[k + 1] = at[k] + bucket_bytes[k];
            n_lines += bucket[k].size();
        }
        out.text.resize(at[B]);
        char* dst = out.text.empty() ? nullptr : &out.text[0];
        parallel_chunks(B, threads_, [&](uint64_t bb, uint64_t be, unsigned) {
            for (uint64_t k = bb; k < be; ++k) {
                char* q = dst + at[k];
                for (const Line& l : bucket[k]) {
                    memcpy(q, l.p, l.n);
                    q[l.n] = '\n';
                    q += l.n + 1;
                }
            }
        });
        out.counters.n_lines = n_lines;
    }
};

// ---- FNO=3 ------------------------------------------------------------------------------------------------
inline int ratio100(int a, int b) { return (int)floorf((float)a / (float)b * 100.0f); }  // FindNextOverlaps3.cpp:259

class Fno3 {
public:
    explicit Fno3(const hc_fno3_input& in) : in_(in), n_(in.n_single + in.n_paired + in.n_trivial), found_(in.new_read_count) {}

    void run(hc_fno_output& out) {
        if (n_ >= 0xFFFFFFFFull) throw FatalError{HC_ERR_ARG, "too many super-reads"};
        if (n_ && (!in_.srs || !in_.orig_off || !in_.originals)) throw FatalError{HC_ERR_ARG, "null array"};
        threads_ = thread_count(in_.n_threads);
        index_originals();
        walk();
        emit(out);
    }

private:
    struct Cand {
        uint32_t a, b;
        uint64_t original_id;
    };
    const hc_fno3_input& in_;
    const uint64_t n_;
    PairSet found_;
    unsigned threads_ = 1;
    std::vector<hc_fno_original> by_id_;  // originals of every super-read, sorted by original_id within the super-read
    std::vector<Cand> cands_;

    void index_originals() {
        by_id_.assign(in_.originals, in_.originals + (n_ ? in_.orig_off[n_] : 0));
        parallel_chunks(n_, threads_, [&](uint64_t b, uint64_t e, unsigned) {
            for (uint64_t i = b; i < e; ++i)
                std::stable_sort(by_id_.begin() + in_.orig_off[i], by_id_.begin() + in_.orig_off[i + 1],
                                 [](const hc_fno_original& x, const hc_fno_original& y) { return x.original_id < y.original_id; });
        });
    }

    const hc_fno_original& original(uint32_t sr, uint64_t id) const {  // originals.at(original_id), :202-203
        const hc_fno_original* b = by_id_.data() + in_.orig_off[sr];
        const hc_fno_original* e = by_id_.data() + in_.orig_off[sr + 1];
        const hc_fno_original* it = std::lower_bound(b, e, id, [](const hc_fno_original& x, uint64_t v) { return x.original_id < v; });
        if (it == e || it->original_id != id) ref_abort("originals.at(original_id)");
        return *it;
    }

    void walk() {
        // :26-76.  The walk order of :100 is the iteration order of this very container type in the reference;
        // it is kept as is so that the order (a property of the host's libstdc++) is the reference's.
        const bool timing = getenv("HC_FNO_TIMING") != nullptr;
        auto tl = std::chrono::steady_clock::now();
        auto lap = [&](const char* what) {
            if (!timing) return;
            const auto t = std::chrono::steady_clock::now();
            fprintf(stderr, "hc_fno3_run: walk: %s %.3f s\n", what, std::chrono::duration<double>(t - tl).count());
            tl = t;
        };
        std::unordered_map<unsigned long, unsigned long> original_to_index;
        std::vector<uint64_t> count;  // super-reads per original, by index
        std::vector<uint64_t> slot_of(n_ ? in_.orig_off[n_] : 0);
        for (uint64_t i = 0; i < n_; ++i)
            for (uint64_t k = in_.orig_off[i]; k < in_.orig_off[i + 1]; ++k) {
                auto ins = original_to_index.emplace((unsigned long)in_.originals[k].original_id, (unsigned long)count.size());
                if (ins.second) {
                    if (count.size() >= in_.original_readcount) ref_abort("nodes_to_SR.at(index): more originals than original_readcount");
                    count.push_back(0);
                }
                slot_of[k] = ins.first->second;
                ++count[ins.first->second];
            }
        lap("originals numbered in order of appearance (the reference's unordered_map)");
        std::vector<uint64_t> off(count.size() + 1, 0);
        for (size_t i = 0; i < count.size(); ++i) off[i + 1] = off[i] + count[i];
        std::vector<uint32_t> members(off.back());
        {
            std::vector<uint64_t> cur(off.begin(), off.end() - 1);
            for (uint64_t i = 0; i < n_; ++i)
                for (uint64_t k = in_.orig_off[i]; k < in_.orig_off[i + 1]; ++k) members[cur[slot_of[k]]++] = (uint32_t)i;
        }
        lap("super-reads per original");
        // nodeDictApproach :100-132
        for (const auto& kv : original_to_index) {
            const uint32_t* m = members.data() + off[kv.second];
            const uint64_t cnt = count[kv.second];
            for (uint64_t i = 0; i < cnt; ++i)
                for (uint64_t j = i + 1; j < cnt; ++j)
                    if (!found_.test_and_set(in_.srs[m[i]].id, in_.srs[m[j]].id)) cands_.push_back(Cand{m[i], m[j], kv.first});
        }
        lap("pairs in the map's order, first met kept");
    }

    // deduceOverlap :180-406 and the two tests of :149-157; returns the end of the line or nullptr
    char* deduce(const Cand& c, char* p) const {
        const hc_fno_read &A = in_.srs[c.a], &B = in_.srs[c.b];
        const hc_fno_original &oa = original(c.a, c.original_id), &ob = original(c.b, c.original_id);
        const int a_l = (int)oa.index1, a_r = (int)oa.index2, b_l = (int)ob.index1, b_r = (int)ob.index2;
        const bool a_first = a_l - b_l >= 0;  // the super-read whose copy of the original starts further right comes first
        uint64_t id1 = a_first ? A.id : B.id, id2 = a_first ? B.id : A.id;
        int pos1 = a_first ? a_l - b_l : b_l - a_l, pos2 = 0, len1, len2 = 0;
        unsigned perc1, perc2 = 0;
        char ord = '-', t1, t2;
        const int A1 = (int)A.len1, A2 = (int)A.len2, B1 = (int)B.len1, B2 = (int)B.len2;
        if (!A.paired && !B.paired) {  // :204-237
            if (pos1 > (a_first ? A1 : B1)) return nullptr;
            len1 = a_first ? std::min(A1 - pos1, B1) : std::min(A1, B1 - pos1);
            perc1 = (unsigned)perc_max(len1, A1, B1);
            t1 = t2 = 's';
        } else if (A.paired != B.paired) {  // :238-281 (P-S), :282-324 (S-P)
            const bool a_pair = A.paired != 0;
            const int P1 = a_pair ? A1 : B1, P2 = a_pair ? A2 : B2, S = a_pair ? B1 : A1;
            const bool pair_first = a_pair == a_first;
            len1 = pair_first ? P1 - pos1 : std::min(P1, S - pos1);
            if (len1 <= 0) return nullptr;
            t1 = pair_first ? 'p' : 's';
            t2 = pair_first ? 's' : 'p';
            perc1 = (unsigned)ratio100(len1, P1);
            pos2 = a_pair ? b_r - a_r : a_r - b_r;  // offset of the pair's /2 inside the single
            len2 = std::min(P2, S - pos2);
            if (len2 <= 0 || pos2 < 0) return nullptr;
            perc2 = (unsigned)ratio100(len2, P2);
        } else {  // :325-399
            len1 = a_first ? std::min(A1 - pos1, B1) : std::min(A1, B1 - pos1);
            const bool back = a_r - b_r >= 0;
            pos2 = back ? a_r - b_r : b_r - a_r;
            len2 = back ? std::min(A2 - pos2, B2) : std::min(A2, B2 - pos2);
            if (len1 <= 0 || len2 <= 0) return nullptr;
            perc1 = (unsigned)perc_max(len1, A1, B1);
            perc2 = (unsigned)perc_max(len2, A2, B2);
            FNO_REQUIRE(perc1 <= 100 && perc2 <= 100);
            ord = a_first == back ? '1' : '2';
            t1 = t2 = 'p';
        }
        // the Overlap constructor's checks, src/Overlap.h:88-102
        if ((int)perc1 < 0 || (int)perc1 > 100 || (int)perc2 < 0 || (int)perc2 > 100) ref_abort("overlap.m_perc not in 0..100");
        if (len1 < 0 || len2 < 0) ref_abort("overlap.m_len < 0");
        const unsigned perc = perc2 > 0 ? (unsigned)(0.5 * (perc1 + perc2)) : perc1;  // Overlap::get_perc
        if ((in_.flags & HC_FNO_NO_INCLUSIONS) && perc == 100) return nullptr;
        if (len1 <= 0) return nullptr;
        p = put_u64(p, id1);
        *p++ = '\t';
        p = put_u64(p, id2);
        *p++ = '\t';
        p = put_u64(p, (unsigned)pos1);
        *p++ = '\t';
        p = put_u64(p, (unsigned)pos2);
        *p++ = '\t';
        *p++ = ord;
        *p++ = '\t';
        *p++ = '+';
        *p++ = '\t';
        *p++ = '+';
        *p++ = '\t';
        p = put_u64(p, perc1);
        *p++ = '\t';
        p = put_u64(p, perc2);
        *p++ = '\t';
        p = put_u64(p, (unsigned)len1);
        *p++ = '\t';
        p = put_u64(p, (unsigned)len2);
        *p++ = '\t';
        *p++ = t1;
        *p++ = '\t';
        *p++ = t2;
        *p++ = '\n';
        return p;
    }

    // The device form: the walk above stays on the host — its order is the iteration order of a
    // std::unordered_map, a property of the reference's libstdc++ — and hands over one 64-byte record per candidate pair with
    // the look-ups done (ids, lengths, where the shared original sits in both super-reads); deduceOverlap, the lines' lengths,
    // their places (a scan) and their text run on the device, in walk order.  Whatever the reference would stop at makes the
    // device report instead of write; the host form below then runs and diagnoses.
    bool emit_on_device(hc_fno_output& out) {
        const uint64_t n = cands_.size();
        if (n == 0 || n >= 0x7FFFFFF0ull) return false;
        const bool timing = getenv("HC_FNO_TIMING") != nullptr;
        auto now = [] { return std::chrono::steady_clock::now(); };
        const auto t0 = now();
        std::unique_ptr<hc::FnoItem[]> items(new hc::FnoItem[n]);
        std::atomic<bool> missing{false};
        parallel_chunks(n, threads_, [&](uint64_t b, uint64_t e, unsigned) {
            for (uint64_t i = b; i < e; ++i) {
                const Cand& c = cands_[i];
                const hc_fno_read &A = in_.srs[c.a], &B = in_.srs[c.b];
                const hc_fno_original *oa = find_original(c.a, c.original_id), *ob = find_original(c.b, c.original_id);
                hc::FnoItem& it = items[i];
                memset(&it, 0, sizeof it);
                if (!oa || !ob) {  // originals.at(original_id) would throw: the host form reports
                    missing = true;
                    continue;
                }
                it.ida = A.id;
                it.idb = B.id;
                it.v[0] = (int32_t)oa->index1;
                it.v[1] = (int32_t)oa->index2;
                it.v[2] = (int32_t)ob->index1;
                it.v[3] = (int32_t)ob->index2;
                it.v[4] = (int32_t)A.len1;
                it.v[5] = (int32_t)A.len2;
                it.v[6] = (int32_t)B.len1;
                it.v[7] = (int32_t)B.len2;
                it.kind = 4;
                it.a_paired = A.paired != 0;
                it.b_paired = B.paired != 0;
            }
        });
        if (missing) return false;
        const auto t1 = now();
        uint64_t n_lines = 0;
        double seconds[2] = {0, 0};
        auto text_of = [&](uint64_t bytes) { return sized_text(out, bytes); };
        try {
            if (!hc::fno3_lines_on_device(items.get(), n, (in_.flags & HC_FNO_NO_INCLUSIONS) != 0, text_of, &n_lines, seconds)) return false;
        } catch (const FatalError& e) {
            if (e.status != HC_ERR_NOMEM) throw;
            return false;
        }
        memset(&out.counters, 0, sizeof out.counters);
        out.counters.n_lines = n_lines;
        out.counters.candidates = n;
        out.on_device = true;
        if (timing)
            fprintf(stderr, "hc_fno3_run (device): look-ups on the host %.3f s, copy + deduceOverlap + scan %.3f s, text %.3f s\n",
                    std::chrono::duration<double>(t1 - t0).count(), seconds[0], seconds[1]);
        return true;
    }

    const hc_fno_original* find_original(uint32_t sr, uint64_t id) const {
        const hc_fno_original* b = by_id_.data() + in_.orig_off[sr];
        const hc_fno_original* e = by_id_.data() + in_.orig_off[sr + 1];
        const hc_fno_original* it = std::lower_bound(b, e, id, [](const hc_fno_original& x, uint64_t v) { return x.original_id < v; });
        return (it == e || it->original_id != id) ? nullptr : it;
    }

    void emit(hc_fno_output& out) {
        constexpr size_t kMaxLine = 160;
        const uint64_t n = cands_.size();
        if (hc::fno_device_wanted(n) && emit_on_device(out)) return;
        const unsigned T = (unsigned)std::min<uint64_t>(threads_, n ? n : 1);
        std::vector<std::unique_ptr<char[]>> arena(T);
        std::vector<size_t> used(T, 0), nlines(T, 0);
        parallel_chunks(n, T, [&](uint64_t b, uint64_t e, unsigned t) {
            arena[t].reset(new char[(e - b) * kMaxLine]);
            char* p = arena[t].get();
            size_t written = 0;
            for (uint64_t i = b; i < e; ++i) {
                char* q = deduce(cands_[i], p);
                if (!q) continue;
                p = q;
                ++written;
            }
            nlines[t] = written;
            used[t] = (size_t)(p - arena[t].get());
        });
        size_t total = 0;
        for (size_t u : used) total += u;
        out.text.clear();
        out.text.reserve(total);
        memset(&out.counters, 0, sizeof out.counters);
        for (unsigned t = 0; t < T; ++t) {
            out.text.insert(out.text.end(), arena[t].get(), arena[t].get() + used[t]);
            out.counters.n_lines += nlines[t];
        }
        out.counters.candidates = n;
    }
};

template <typename F>
int guarded_fno(const char* where, F&& f) {
    try {
        f();
        return HC_OK;
    } catch (const FatalError& e) {
        return hc::set_last_error(e.status, std::string(where) + ": " + e.what);
    } catch (const std::bad_alloc&) {
        return hc::set_last_error(HC_ERR_NOMEM, std::string(where) + ": out of memory");
    } catch (const std::exception& e) {
        return hc::set_last_error(HC_ERR_FORMAT, std::string(where) + ": " + e.what());
    }
}

}  // namespace

extern "C" {

// (flags this build does not know: refused, hcfno.h)
static int refuse_unbuilt_flags(const char* who, uint32_t flags) {
    if (flags & ~HC_FNO_KNOWN_FLAGS) return hc::set_last_error(HC_ERR_ARG, std::string(who) + ": unknown bit in flags");
    return HC_OK;
}

int hc_fno1_run(const hc_fno1_input* in, hc_fno_output** out) {
    if (!in || !out) return hc::set_last_error(HC_ERR_ARG, "hc_fno1_run: null argument");
    *out = nullptr;
    if (int rc = refuse_unbuilt_flags("hc_fno1_run", in->flags)) return rc;
    return guarded_fno("hc_fno1_run", [&] {
        std::unique_ptr<hc_fno_output> o(new hc_fno_output());
        Fno1(*in).run(*o);
        *out = o.release();
    });
}

int hc_fno3_run(const hc_fno3_input* in, hc_fno_output** out) {
    if (!in || !out) return hc::set_last_error(HC_ERR_ARG, "hc_fno3_run: null argument");
    *out = nullptr;
    if (int rc = refuse_unbuilt_flags("hc_fno3_run", in->flags)) return rc;
    return guarded_fno("hc_fno3_run", [&] {
        std::unique_ptr<hc_fno_output> o(new hc_fno_output());
        Fno3(*in).run(*o);
        *out = o.release();
    });
}

int hc_fno_output_text(const hc_fno_output* o, const char** text, uint64_t* n_bytes) {
    if (!o || !text || !n_bytes) return hc::set_last_error(HC_ERR_ARG, "hc_fno_output_text: null argument");
    *text = o->text.data();
    *n_bytes = o->text.size();
    return HC_OK;
}

int hc_fno_output_counters(const hc_fno_output* o, hc_fno_counters* c) {
    if (!o || !c) return hc::set_last_error(HC_ERR_ARG, "hc_fno_output_counters: null argument");
    *c = o->counters;
    return HC_OK;
}

int hc_fno_output_write(const hc_fno_output* o, const char* path) {
    if (!o || !path) return hc::set_last_error(HC_ERR_ARG, "hc_fno_output_write: null argument");
    FILE* f = fopen(path, "wb");
    if (!f) return hc::set_last_error(HC_ERR_IO, std::string("hc_fno_output_write: cannot open ") + path);
    const size_t w = o->text.empty() ? 0 : fwrite(o->text.data(), 1, o->text.size(), f);
    const int rc = fclose(f);
    if (w != o->text.size() || rc != 0) return hc::set_last_error(HC_ERR_IO, std::string("hc_fno_output_write: short write to ") + path);
    return HC_OK;
}

void hc_fno_output_free(hc_fno_output* o) { delete o; }
int hc_fno_output_on_device(const hc_fno_output* o) { return o && o->on_device ? (o->walk_on_device ? 2 : 1) : 0; }

int hc_fno_compute_overlap_data(const hc_fno_read* sr1, const hc_fno_read* sr2, const int32_t idx[4], const hc_fno_edge* edge, int32_t* ok,
                                int32_t out9[9]) {
    if (!sr1 || !sr2 || !idx || !edge || !ok || !out9) return hc::set_last_error(HC_ERR_ARG, "hc_fno_compute_overlap_data: null argument");
    return guarded_fno("hc_fno_compute_overlap_data", [&] {
        Induced o;
        memset(&o, 0, sizeof o);
        *ok = induced_overlap(facts(*sr1), facts(*sr2), idx[0], idx[1], idx[2], idx[3], *edge, o) ? 1 : 0;
        const int32_t v[9] = {o.pos1, o.pos2, o.ord1, o.ord2, o.type1, o.type2, o.perc, o.len1, o.len2};
        memcpy(out9, v, sizeof v);
    });
}

}  // extern "C"
