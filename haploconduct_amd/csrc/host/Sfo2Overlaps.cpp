// Sfo2Overlaps.cpp — SFO ingest: rust-overlaps' 8-column SFO lines
//   idA idB ori(N|I) OHA OHB OLA OLB K
// to SAVAGE's 13-column overlaps file, with the semantics of the reference's
// scripts/sfo2overlaps.py (cited per function), so that the text no longer round-trips through
// Python 2, `sort` and `uniq`.  Quirks that are kept on purpose because they shape the output:
//   * the overlap percentage uses Python 2's round() (halves away from zero), :189;
//   * match_candidates receives the read types of the line that CLOSES a group, not of the group, :94;
//   * the last group of paired candidates in the file is never matched (no flush after the loop), :63-103;
//   * ties of the four numeric sort keys are ordered by the whole line, bytewise (sort under LC_ALL=C).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include <sys/mman.h>

#include "../../../include/hcedge_host.h"
#include "../hc_sfo_items.h"
#include "DefaultInit.h"
#include "Types.h"

namespace hc {

namespace {

struct SfoRec {
    long id[2];        // original read ids (after get_original_id), id[0] <= id[1]
    long sfo[2];       // SFO ids in the order of the stored line
    char ori;          // 'N' / 'I' (anything else is carried through like the script does)
    long oha, ohb, ola, olb;
    // the line as the script writes it to its temporary file (sort tie-break, uniq): a slice of one arena
    size_t text_off;
    uint32_t text_len;
};



bool parse_long(const char* p, size_t n, long& v) {  // Python int(): optional sign, digits, surrounding blanks were split off
    if (n == 0 || n > 30) return false;
    char buf[32];
    memcpy(buf, p, n);
    buf[n] = 0;
    char* end = nullptr;
    v = strtol(buf, &end, 10);
    return end == buf + n;
}

long original_id(long sfo_id, long ns, long np) {  // :136-147
    if (np == 0) return sfo_id;
    if (!(sfo_id >= 0 && sfo_id < ns + 2 * np)) throw FatalError{HC_ERR_FORMAT, "SFO read id out of range for --num_singles/--num_pairs"};
    return sfo_id < ns + np ? sfo_id : sfo_id - np;
}

bool is_paired(long id, long ns, long np) {  // :124-134
    if (np == 0) return false;
    if (!(id >= 0 && id < ns + np)) throw FatalError{HC_ERR_FORMAT, "read id out of range for --num_singles/--num_pairs"};
    return id >= ns;
}

inline char* put_long(char* p, long v) {  // what "%ld" prints
    unsigned long u = v < 0 ? 0ul - (unsigned long)v : (unsigned long)v;
    if (v < 0) *p++ = '-';
    char tmp[24];
    int n = 0;
    do {
        tmp[n++] = (char)('0' + u % 10);
        u /= 10;
    } while (u);
    while (n) *p++ = tmp[--n];
    return p;
}

unsigned worker_count(size_t items) {
    unsigned t = std::thread::hardware_concurrency();
    if (t == 0) t = 1;
    if (t > 16) t = 16;
    const size_t by_size = items / 50000 + 1;  // not worth a thread below ~50 k items each
    return (unsigned)(by_size < t ? by_size : t);
}

template <typename It, typename Cmp>
void parallel_sort(It begin, It end, Cmp cmp) {  // sorted chunks, then pairwise merges; any total order
    const size_t n = (size_t)(end - begin);
    const unsigned T = worker_count(n);
    if (T <= 1) {
        std::sort(begin, end, cmp);
        return;
    }
    std::vector<size_t> cut(T + 1);
    for (unsigned t = 0; t <= T; t++) cut[t] = n * t / T;
    {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < T; t++) th.emplace_back([&, t] { std::sort(begin + cut[t], begin + cut[t + 1], cmp); });
        for (auto& x : th) x.join();
    }
    while (cut.size() > 2) {
        std::vector<size_t> next{0};
        std::vector<std::thread> th;
        for (size_t k = 0; k + 2 < cut.size(); k += 2) {
            th.emplace_back([&, k] { std::inplace_merge(begin + cut[k], begin + cut[k + 1], begin + cut[k + 2], cmp); });
            next.push_back(cut[k + 2]);
        }
        for (auto& x : th) x.join();
        if (next.back() != cut.back()) next.push_back(cut.back());
        cut.swap(next);
    }
}

struct Ingest {  // the records of one SFO input and the text arena behind them
    std::vector<SfoRec> recs;
    std::string arena;
    long ns, np;

    // one SFO record: ids, orientation text, the four numbers, the last column, and the line verbatim
    void add(long ida, long idb, const char* ori, size_t ori_len, long oha, long ohb, long ola, long olb, const char* k, size_t k_len,
             const char* line, size_t line_len) {
        SfoRec r;
        const long na = original_id(ida, ns, np), nb = original_id(idb, ns, np);
        r.ori = ori_len == 1 ? ori[0] : '?';
        r.oha = oha; r.ohb = ohb; r.ola = ola; r.olb = olb;
        r.text_off = arena.size();
        char buf[320];
        if (na > nb) {  // flip_N / flip_I, :112-122
            r.id[0] = nb; r.id[1] = na;
            r.sfo[0] = idb; r.sfo[1] = ida;
            if (ori_len == 1 && ori[0] == 'I') std::swap(r.oha, r.ohb);
            else { r.oha = -r.oha; r.ohb = -r.ohb; }
            std::swap(r.ola, r.olb);
            if (ori_len + k_len > 128) throw FatalError{HC_ERR_FORMAT, "SFO line with an oversized field"};
            char* q = buf;
            const long head[4] = {nb, na, idb, ida};
            for (long v : head) {
                q = put_long(q, v);
                *q++ = '\t';
            }
            memcpy(q, ori, ori_len);
            q += ori_len;
            const long nums[4] = {r.oha, r.ohb, r.ola, r.olb};
            for (long v : nums) {
                *q++ = '\t';
                q = put_long(q, v);
            }
            *q++ = '\t';
            memcpy(q, k, k_len);
            q += k_len;
            *q++ = '\n';
            arena.append(buf, (size_t)(q - buf));
        } else {
            r.id[0] = na; r.id[1] = nb;
            r.sfo[0] = ida; r.sfo[1] = idb;
            char* q = put_long(buf, na);
            *q++ = '\t';
            q = put_long(q, nb);
            *q++ = '\t';
            arena.append(buf, (size_t)(q - buf));
            arena.append(line, line_len);  // the original line verbatim (its own separators), :48
            arena.push_back('\n');
        }
        r.text_len = (uint32_t)(arena.size() - r.text_off);
        recs.push_back(r);
    }
    std::string finish(uint64_t& n_lines);
};

struct SS {  // one single-single overlap in SAVAGE columns, :150-200
    long id1, id2, pos1, perc, len;
    char ori1, ori2;
};

SS s_s_overlap(const SfoRec& r) {
    SS o;
    const char ori = r.ori == 'N' ? '+' : '-';
    const long ovlen = std::min(r.ola, r.olb);
    long la, lb;
    if (r.oha >= 0) {  // read A is first
        if (r.ohb >= 0) { la = r.ola + r.oha; lb = r.olb + r.ohb; } else { la = r.ola + r.oha - r.ohb; lb = r.olb; }
        o.id1 = r.id[0]; o.id2 = r.id[1]; o.pos1 = r.oha; o.ori1 = '+'; o.ori2 = ori;
    } else {           // read B is first
        if (r.ohb >= 0) { la = r.ola; lb = -r.oha + r.olb + r.ohb; } else { la = r.ola - r.ohb; lb = -r.oha + r.olb; }
        o.id1 = r.id[1]; o.id2 = r.id[0]; o.pos1 = -r.oha; o.ori1 = ori; o.ori2 = '+';
    }
    const long minlen = std::min(la, lb);
    if (minlen == 0) throw FatalError{HC_ERR_FORMAT, "SFO line with an empty read (division by zero in the reference)"};
    const double perc = std::min(std::round(100.0 * (double)ovlen / (double)minlen), 100.0);  // Python 2 round(): C round()
    if (!(minlen > 0)) throw FatalError{HC_ERR_FORMAT, "SFO line with a negative read length (assert in the reference)"};
    o.perc = (long)perc;
    o.len = ovlen;
    return o;
}

void put_ss(std::string& out, const SS& o) {
    char buf[160];
    const int k = snprintf(buf, sizeof buf, "%ld\t%ld\t%ld\t-\t-\t%c\t%c\t%ld\t-\t%ld\t-\ts\ts\n", o.id1, o.id2, o.pos1, o.ori1, o.ori2,
                           o.perc, o.len);
    out.append(buf, (size_t)k);
}

// find_paired_overlap + merge_overlaps, :221-329.  Returns false when the two candidates do not combine.
bool paired_overlap(const SfoRec& c1, const SfoRec& c2, bool type_a, bool type_b, std::string& out) {
    if (c1.ori != c2.ori) return false;
    const long a1 = c1.sfo[0], b1 = c1.sfo[1], a2 = c2.sfo[0], b2 = c2.sfo[1];
    const bool normal = c1.ori == 'N', inv = c1.ori == 'I';
    int first = 0;  // which candidate provides overlap1
    if (type_a && type_b) {
        if (normal) first = (a1 < a2 && b1 < b2) ? 1 : ((a1 > a2 && b1 > b2) ? 2 : 0);
        else if (inv) first = (a1 < a2 && b1 > b2) ? 1 : ((a1 > a2 && b1 < b2) ? 2 : 0);
    } else {
        const long p1 = c1.oha, p2 = c2.oha;
        const long k1 = (type_a && !type_b) ? a1 : b1, k2 = (type_a && !type_b) ? a2 : b2;
        if (normal) first = (k1 < k2 && p1 < p2) ? 1 : ((k1 > k2 && p1 > p2) ? 2 : 0);
        else if (inv) first = (k1 < k2 && p1 > p2) ? 2 : ((k1 > k2 && p1 < p2) ? 1 : 0);
    }
    if (!first) return false;
    const SS o1 = s_s_overlap(first == 1 ? c1 : c2), o2 = s_s_overlap(first == 1 ? c2 : c1);
    char t1, t2;
    if (o1.id1 == c1.id[0]) {
        if (o1.id2 != c1.id[1]) throw FatalError{HC_ERR_FORMAT, "inconsistent paired candidates (assert in the reference)"};
        t1 = type_a ? 'p' : 's';
        t2 = type_b ? 'p' : 's';
    } else {
        if (!(o1.id2 == c1.id[0] && o1.id1 == c1.id[1])) throw FatalError{HC_ERR_FORMAT, "inconsistent paired candidates (assert in the reference)"};
        t1 = type_b ? 'p' : 's';
        t2 = type_a ? 'p' : 's';
    }
    char ord = '-';
    if (t1 == 'p' && t2 == 'p') {
        if (o1.id1 != o2.id1) {
            if (o1.id1 != o2.id2) throw FatalError{HC_ERR_FORMAT, "inconsistent paired candidates (assert in the reference)"};
            ord = '2';
        } else {
            ord = '1';
        }
    }
    char buf[200];
    const int k = snprintf(buf, sizeof buf, "%ld\t%ld\t%ld\t%ld\t%c\t%c\t%c\t%ld\t%ld\t%ld\t%ld\t%c\t%c\n", o1.id1, o1.id2, o1.pos1, o2.pos1,
                           ord, o1.ori1, o1.ori2, o1.perc, o2.perc, o1.len, o2.len, t1, t2);
    out.assign(buf, (size_t)k);
    return true;
}

std::string Ingest::finish(uint64_t& n_lines) {
    // sort -k1,1n -k2,2n -k3,3n -k4,4n | uniq   (:53), LC_ALL=C — on indices, the records stay where they are
    std::vector<uint32_t> order(recs.size());
    for (size_t i = 0; i < order.size(); i++) order[i] = (uint32_t)i;
    const char* A = arena.data();
    auto text_cmp = [&](const SfoRec& x, const SfoRec& y) {
        const int c = memcmp(A + x.text_off, A + y.text_off, x.text_len < y.text_len ? x.text_len : y.text_len);
        return c ? c : (x.text_len < y.text_len ? -1 : (x.text_len > y.text_len ? 1 : 0));
    };
    bool small_ids = true;  // all four numeric keys fit 32 bits: sort compact keys instead of chasing the records
    for (const SfoRec& r : recs)
        if ((unsigned long)r.id[0] >> 32 || (unsigned long)r.id[1] >> 32 || (unsigned long)r.sfo[0] >> 32 || (unsigned long)r.sfo[1] >> 32) {
            small_ids = false;
            break;
        }
    if (small_ids) {
        struct Key {
            uint64_t ids, sfos;
            uint32_t idx;
        };
        std::vector<Key> keys(recs.size());
        for (size_t i = 0; i < recs.size(); i++)
            keys[i] = Key{((uint64_t)recs[i].id[0] << 32) | (uint64_t)recs[i].id[1], ((uint64_t)recs[i].sfo[0] << 32) | (uint64_t)recs[i].sfo[1],
                          (uint32_t)i};
        parallel_sort(keys.begin(), keys.end(), [&](const Key& x, const Key& y) {
            if (x.ids != y.ids) return x.ids < y.ids;
            if (x.sfos != y.sfos) return x.sfos < y.sfos;
            return text_cmp(recs[x.idx], recs[y.idx]) < 0;
        });
        for (size_t i = 0; i < keys.size(); i++) order[i] = keys[i].idx;
    } else {
        parallel_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
            const SfoRec &x = recs[a], &y = recs[b];
            if (x.id[0] != y.id[0]) return x.id[0] < y.id[0];
            if (x.id[1] != y.id[1]) return x.id[1] < y.id[1];
            if (x.sfo[0] != y.sfo[0]) return x.sfo[0] < y.sfo[0];
            if (x.sfo[1] != y.sfo[1]) return x.sfo[1] < y.sfo[1];
            return text_cmp(x, y) < 0;
        });
    }
    std::string out, last_line, cur;
    n_lines = 0;
    auto emit = [&](const std::string& l) {  // the final `uniq`, :107
        if (n_lines && l == last_line) return;
        out += l;
        last_line = l;
        n_lines++;
    };
    std::vector<const SfoRec*> cands;
    for (size_t i = 0; i < order.size(); i++) {
        const SfoRec& r = recs[order[i]];
        if (i && text_cmp(r, recs[order[i - 1]]) == 0) continue;  // uniq
        if (r.id[0] == r.id[1]) continue;  // self-overlap, :69-70
        const bool pa = is_paired(r.id[0], ns, np), pb = is_paired(r.id[1], ns, np);
        if (!pa && !pb) {  // :79-85
            cur.clear();
            put_ss(cur, s_s_overlap(r));
            emit(cur);
            continue;
        }
        if (!cands.empty() && (cands[0]->id[0] != r.id[0] || cands[0]->id[1] != r.id[1])) {  // :89-102
            if (cands.size() >= 2)
                for (size_t a = 0; a < cands.size(); a++)
                    for (size_t b = a + 1; b < cands.size(); b++)
                        if (paired_overlap(*cands[a], *cands[b], pa, pb, cur)) emit(cur);
            cands.clear();
        }
        cands.push_back(&r);
    }
    return out;
}

}  // namespace

std::string sfo_records_to_overlaps(const hc_sfo_rec* recs, uint64_t n, long ns, long np, uint64_t& n_lines);

namespace {

// One field of a CANONICAL line: "0" or [1-9][0-9]* (optionally after one '-'), ended by `end`.
inline bool canonical_int(const char*& p, const char* e, char end, bool allow_negative, int64_t lo, int64_t hi, int64_t& v) {
    bool neg = false;
    if (p < e && *p == '-') {
        if (!allow_negative) return false;
        neg = true;
        p++;
    }
    const char* b = p;
    uint64_t u = 0;
    while (p < e && (unsigned)(*p - '0') <= 9u && p - b < 11) u = u * 10 + (uint64_t)(*p++ - '0');
    const size_t d = (size_t)(p - b);
    if (d == 0 || d > 10 || (b[0] == '0' && (d > 1 || neg))) return false;  // no leading zeros, no "-0"
    v = neg ? -(int64_t)u : (int64_t)u;
    if (v < lo || v > hi) return false;
    if (end == '\n') return p == e;
    if (p >= e || *p != end) return false;
    p++;
    return true;
}

// A file as hc_host_write_sfo (and a tab-separated rust-overlaps run) writes it: eight fields, single tabs, canonical
// decimal numbers in the ranges of hc_sfo_rec, `N` or `I`.  For such a file the line the script keeps for its sort and
// uniq is exactly the ten-number line the records path reasons about, so the whole ingest can run there.  Returns false
// at the first line of any other shape (the general path below then handles — and diagnoses — the file).
template <class RecVec>  // std::vector<hc_sfo_rec>, with or without an allocator that leaves resize()'s new elements uninitialised
bool parse_canonical_sfo(const char* text, size_t N, RecVec& recs) {
    if (N == 0) return false;
    unsigned T = std::thread::hardware_concurrency();
    if (T == 0) T = 1;
    if (T > 32) T = 32;
    if (N / (1u << 20) + 1 < T) T = (unsigned)(N / (1u << 20) + 1);
    std::vector<size_t> cut(T + 1, N);
    cut[0] = 0;
    for (unsigned t = 1; t < T; t++) {
        size_t c = std::max(cut[t - 1], N * t / T);
        const char* nl = c < N ? (const char*)memchr(text + c, '\n', N - c) : nullptr;
        cut[t] = nl ? (size_t)(nl - text) + 1 : N;
    }
    std::vector<uint64_t> lines(T, 0);
    auto run = [&](const std::function<void(unsigned)>& body) {
        std::vector<std::thread> th;
        for (unsigned t = 1; t < T; t++) th.emplace_back(body, t);
        body(0);
        for (auto& x : th) x.join();
    };
    run([&](unsigned t) {
        const char* b = text + cut[t];
        const char* e = text + cut[t + 1];
        uint64_t k = (uint64_t)std::count(b, e, '\n');
        if (e > b && e[-1] != '\n') k++;
        lines[t] = k;
    });
    std::vector<uint64_t> at(T + 1, 0);
    for (unsigned t = 0; t < T; t++) at[t + 1] = at[t] + lines[t];
    recs.resize(at[T]);
    std::vector<uint8_t> ok(T, 1);
    run([&](unsigned t) {
        const char* p = text + cut[t];
        const char* const end = text + cut[t + 1];
        hc_sfo_rec* out = recs.data() + at[t];
        while (p < end) {
            const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
            const char* e = nl ? nl : end;
            int64_t a, b, oha, ohb, ola, olb, k;
            const char* q = p;
            bool good = canonical_int(q, e, '\t', false, 0, 0xFFFFFFFFll, a) && canonical_int(q, e, '\t', false, 0, 0xFFFFFFFFll, b);
            char ori = 0;
            if (good) {
                good = q + 1 < e && (q[0] == 'N' || q[0] == 'I') && q[1] == '\t';
                ori = q[0];
                q += 2;
            }
            good = good && canonical_int(q, e, '\t', true, -2147483647ll, 2147483647ll, oha) && canonical_int(q, e, '\t', true, -2147483647ll, 2147483647ll, ohb) &&
                   canonical_int(q, e, '\t', false, 0, 0xFFFFFFFFll, ola) && canonical_int(q, e, '\t', false, 0, 0xFFFFFFFFll, olb) &&
                   canonical_int(q, e, '\n', false, 0, 0xFFFFFFFFll, k);
            if (!good) {
                ok[t] = 0;
                return;
            }
            out->idA = (uint32_t)a; out->idB = (uint32_t)b;
            out->OHA = (int32_t)oha; out->OHB = (int32_t)ohb;
            out->OLA = (uint32_t)ola; out->OLB = (uint32_t)olb;
            out->K = (uint32_t)k;
            out->inverted = ori == 'I';
            out++;
            p = nl ? nl + 1 : end;
        }
    });
    for (uint8_t x : ok)
        if (!x) return false;
    return true;
}

}  // namespace

// The records of a CANONICAL SFO text (what rust-overlaps / hc_host_write_sfo write), or false: the general path owns such a file.
// (the records are written in full by the parsing threads: not zero-filled first — 2 GB at config 3's size, a third of a second on one thread)
bool sfo_text_to_records(const char* sfo_text, size_t sfo_bytes, std::vector<hc_sfo_rec, DefaultInitAllocator<hc_sfo_rec>>& recs) {
    if (getenv("HC_SFO_TEXT_GENERAL")) return false;  // (test knob: always the general path)
    return parse_canonical_sfo(sfo_text, sfo_bytes, recs);
}

// Returns the output text; n_lines receives the number of lines.
std::string sfo_to_overlaps(const char* sfo_text, size_t sfo_bytes, long ns, long np, uint64_t& n_lines) {
    if (!getenv("HC_SFO_TEXT_GENERAL")) {  // (test knob: always take the general path)
        std::vector<hc_sfo_rec> recs;
        if (parse_canonical_sfo(sfo_text, sfo_bytes, recs)) return sfo_records_to_overlaps(recs.data(), recs.size(), ns, np, n_lines);
    }
    Ingest in;
    in.ns = ns;
    in.np = np;
    in.arena.reserve(sfo_bytes + sfo_bytes / 2);
    size_t pos = 0;
    const size_t N = sfo_bytes;
    while (pos < N) {  // :31-50
        const char* nl = (const char*)memchr(sfo_text + pos, '\n', N - pos);
        const size_t end = nl ? (size_t)(nl - sfo_text) : N;
        const char* line = sfo_text + pos;
        const size_t len = end - pos;
        // line.strip('\n').split(): any run of whitespace separates
        const char* f[9];
        size_t fl[9];
        int nf = 0;
        size_t i = 0;
        while (i < len) {
            while (i < len && isspace((unsigned char)line[i])) i++;
            if (i >= len) break;
            const size_t b = i;
            while (i < len && !isspace((unsigned char)line[i])) i++;
            if (nf < 9) { f[nf] = line + b; fl[nf] = i - b; }
            nf++;
        }
        if (nf != 8) throw FatalError{HC_ERR_FORMAT, "SFO line does not have 8 fields (assert in the reference)"};
        long ida, idb, oha, ohb, ola, olb;
        if (!parse_long(f[0], fl[0], ida) || !parse_long(f[1], fl[1], idb) || !parse_long(f[3], fl[3], oha) ||
            !parse_long(f[4], fl[4], ohb) || !parse_long(f[5], fl[5], ola) || !parse_long(f[6], fl[6], olb))
            throw FatalError{HC_ERR_FORMAT, "SFO line with a non-integer field"};
        in.add(ida, idb, f[2], fl[2], oha, ohb, ola, olb, f[7], fl[7], line, len);
        pos = nl ? end + 1 : N;
    }
    return in.finish(n_lines);
}

// ---------------------------------------------------------------------------------------------------------------
// The same from binary records (hc_find_overlaps): exactly what the text path yields for the file hc_host_write_sfo
// writes for them, without writing, parsing or even building that text.  With binary input the line of the script's
// temporary file is a function of ten numbers — id0 id1 sfo0 sfo1 ori OHA OHB OLA OLB K after the flip of :112-122 —
// so `sort -k1,1n -k2,2n -k3,3n -k4,4n` with the whole line as last resort (LC_ALL=C) is: the four numbers, then `ori`,
// then the remaining five as DECIMAL STRINGS (a tab ends the shorter one and sorts below every digit and '-'), and
// `uniq` is equality of the ten.  Records are partitioned by id0 into one bucket per thread (sampled splitters),
// buckets are sorted and matched independently, and the one thing that crosses a bucket border — the group of paired
// candidates still open at its end, which the script matches when the NEXT non-single line arrives, with that line's
// read types (:89-102) — is stitched in afterwards.
namespace {

struct BRec {
    uint32_t id0, id1, s0, s1;
    int64_t oha, ohb;
    uint32_t ola, olb, k;
    char ori;
};

inline int cmp_decimal(long a, long b) {  // order of the "%ld" texts inside a tab-separated line
    if (a == b) return 0;
    char x[24], y[24];
    const size_t nx = (size_t)(put_long(x, a) - x), ny = (size_t)(put_long(y, b) - y);
    const int c = memcmp(x, y, nx < ny ? nx : ny);
    if (c) return c;
    return nx < ny ? -1 : 1;  // the shorter one is followed by '\t' / '\n'
}

inline bool brec_less(const BRec& x, const BRec& y) {
    if (x.id0 != y.id0) return x.id0 < y.id0;
    if (x.id1 != y.id1) return x.id1 < y.id1;
    if (x.s0 != y.s0) return x.s0 < y.s0;
    if (x.s1 != y.s1) return x.s1 < y.s1;
    if (x.ori != y.ori) return (unsigned char)x.ori < (unsigned char)y.ori;
    int c;
    if ((c = cmp_decimal((long)x.oha, (long)y.oha))) return c < 0;
    if ((c = cmp_decimal((long)x.ohb, (long)y.ohb))) return c < 0;
    if ((c = cmp_decimal((long)x.ola, (long)y.ola))) return c < 0;
    if ((c = cmp_decimal((long)x.olb, (long)y.olb))) return c < 0;
    return cmp_decimal((long)x.k, (long)y.k) < 0;
}

inline bool brec_same(const BRec& x, const BRec& y) {
    return x.id0 == y.id0 && x.id1 == y.id1 && x.s0 == y.s0 && x.s1 == y.s1 && x.ori == y.ori && x.oha == y.oha && x.ohb == y.ohb &&
           x.ola == y.ola && x.olb == y.olb && x.k == y.k;
}

inline SfoRec to_sfo(const BRec& r) {
    SfoRec o;
    o.id[0] = r.id0; o.id[1] = r.id1;
    o.sfo[0] = r.s0; o.sfo[1] = r.s1;
    o.ori = r.ori;
    o.oha = (long)r.oha; o.ohb = (long)r.ohb; o.ola = r.ola; o.olb = r.olb;
    o.text_off = 0;
    o.text_len = 0;
    return o;
}

// Sorts one bucket by brec_less.  The order is decided by (id0, id1) for all but the few records of one pair of reads, so
// what is sorted are 16-byte (key, index) entries — one integer comparison, the full one only between records of the same
// pair — and the 48-byte records are moved once, at the end.
struct SortEntry {
    uint64_t key;
    uint32_t idx, s0;
};
struct SortScratch {  // one per worker thread, grown to its largest bucket: no allocation (and no page faults) per bucket
    std::vector<SortEntry> entries;
    std::vector<BRec> records;
};
void sort_bucket(BRec* r, size_t n, SortScratch& scratch) {
    if (n < 2) return;
    if (n >= (1ull << 32)) {
        std::sort(r, r + n, brec_less);
        return;
    }
    if (scratch.entries.size() < n) {
        scratch.entries.resize(n);
        scratch.records.resize(n);
    }
    SortEntry* e = scratch.entries.data();
    for (size_t i = 0; i < n; i++) e[i] = SortEntry{(uint64_t)r[i].id0 << 32 | r[i].id1, (uint32_t)i, r[i].s0};
    std::sort(e, e + n, [r](const SortEntry& x, const SortEntry& y) {
        if (x.key != y.key) return x.key < y.key;
        if (x.s0 != y.s0) return x.s0 < y.s0;
        return brec_less(r[x.idx], r[y.idx]);
    });
    BRec* tmp = scratch.records.data();
    for (size_t i = 0; i < n; i++) tmp[i] = r[e[i].idx];
    memcpy(r, tmp, n * sizeof(BRec));
}

struct RecArray {  // n records, not initialised: the workers touch (and so place) their own stretches first
    BRec* p = nullptr;
    explicit RecArray(size_t n) {
        if (!n) return;
        const size_t bytes = (n * sizeof(BRec) + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
        if (posix_memalign((void**)&p, (size_t)2 << 20, bytes) != 0) throw FatalError{HC_ERR_NOMEM, "sfo_records_to_overlaps: out of memory"};
        madvise(p, bytes, MADV_HUGEPAGE);  // a hint: 2 MiB pages where the system grants them (fewer first-touch faults)
    }
    ~RecArray() { free(p); }
    RecArray(const RecArray&) = delete;
    RecArray& operator=(const RecArray&) = delete;
    void release() {
        free(p);
        p = nullptr;
    }
};

struct Emitter {  // the output lines of one stretch, with the final `uniq` (:107) applied inside it
    std::string text, last, cur;
    uint64_t n_lines = 0;
    void line(const std::string& l) {
        if (n_lines && l == last) return;
        text += l;
        last = l;
        n_lines++;
    }
    void match_group(const std::vector<BRec>& cands, bool pa, bool pb) {  // match_candidates, :89-102 / :203-219
        if (cands.size() < 2) return;
        std::vector<SfoRec> c;
        c.reserve(cands.size());
        for (const BRec& r : cands) c.push_back(to_sfo(r));
        for (size_t a = 0; a < c.size(); a++)
            for (size_t b = a + 1; b < c.size(); b++)
                if (paired_overlap(c[a], c[b], pa, pb, cur)) line(cur);
    }
};

struct BucketResult {
    Emitter before, after;      // output up to / from the first non-single record of the bucket
    bool has_paired = false;    // the bucket holds a record that involves a paired read ...
    bool pa = false, pb = false;  // ... and these are the types of the first such record
    std::vector<BRec> open;     // the group still open at the end of the bucket
    FatalError error{0, ""};
};

template <typename Get>  // get(i): the i-th record of the bucket, in sorted order
void match_bucket(Get get, size_t n, long ns, long np, BucketResult& out) {
    std::vector<BRec> cands;
    Emitter* em = &out.before;
    BRec prev{};
    for (size_t i = 0; i < n; i++) {
        const BRec r = get(i);
        const bool repeated = i && brec_same(r, prev);
        prev = r;
        if (repeated) continue;       // uniq
        if (r.id0 == r.id1) continue;  // self-overlap, :69-70
        const bool pa = is_paired(r.id0, ns, np), pb = is_paired(r.id1, ns, np);
        if (!pa && !pb) {  // :79-85
            em->cur.clear();
            put_ss(em->cur, s_s_overlap(to_sfo(r)));
            em->line(em->cur);
            continue;
        }
        if (!out.has_paired) {  // whatever group is open from earlier buckets is matched here, by the stitching pass
            out.has_paired = true;
            out.pa = pa;
            out.pb = pb;
            em = &out.after;
        }
        if (!cands.empty() && (cands[0].id0 != r.id0 || cands[0].id1 != r.id1)) {
            em->match_group(cands, pa, pb);
            cands.clear();
        }
        cands.push_back(r);
    }
    out.open.swap(cands);
}

// the open group of a bucket travels to the next bucket that holds a non-single record and is matched there
std::string stitch(std::vector<BucketResult>& res, uint64_t& n_lines) {
    std::string out;
    n_lines = 0;
    {
        size_t bytes = 0;
        for (const BucketResult& r : res) bytes += r.before.text.size() + r.after.text.size();
        out.reserve(bytes + bytes / 16);
    }
    std::vector<BRec> open;
    for (BucketResult& r : res) {
        if (r.error.status) throw r.error;
        out += r.before.text;
        n_lines += r.before.n_lines;
        if (r.has_paired) {
            Emitter em;
            em.match_group(open, r.pa, r.pb);
            out += em.text;
            n_lines += em.n_lines;
            open.swap(r.open);
        }
        out += r.after.text;
        n_lines += r.after.n_lines;
    }  // the group open at the very end is never matched (:63-103)
    return out;
}

void run_workers(unsigned count, const std::function<void(unsigned)>& body) {
    if (count <= 1) {
        body(0);
        return;
    }
    std::vector<std::thread> th;
    for (unsigned t = 1; t < count; t++) th.emplace_back(body, t);
    body(0);
    for (auto& x : th) x.join();
}

}  // namespace

std::string sfo_records_to_overlaps(const hc_sfo_rec* recs, uint64_t n, long ns, long np, uint64_t& n_lines) {
    const auto t0 = std::chrono::steady_clock::now();
    unsigned T = std::thread::hardware_concurrency();
    if (T == 0) T = 1;
    if (T > 64) T = 64;
    // buckets: several per thread (short sorts that stay in the caches, an even finish), none below 20 000 records
    uint64_t NB = std::min<uint64_t>((uint64_t)T * 8, n / 20000 + 1);
    if (const char* e = getenv("HC_SFO_BUCKETS")) NB = (uint64_t)std::max(1, atoi(e));  // test knob: many buckets on small inputs
    if (NB > 4096) NB = 4096;
    if (NB < T) T = (unsigned)NB;
    auto workers = [&](unsigned count, const std::function<void(unsigned)>& body) {
        if (count <= 1) {
            body(0);
            return;
        }
        std::vector<std::thread> th;
        for (unsigned t = 1; t < count; t++) th.emplace_back(body, t);
        body(0);
        for (auto& x : th) x.join();
    };
    // the flip of :112-122 (smaller original id first); id0_of = the id it puts first
    auto id0_of = [&](const hc_sfo_rec& r) {
        const long na = original_id((long)r.idA, ns, np), nb = original_id((long)r.idB, ns, np);
        return (uint32_t)(na > nb ? nb : na);
    };
    auto flip_into = [&](const hc_sfo_rec& r, BRec& o) {
        const long na = original_id((long)r.idA, ns, np), nb = original_id((long)r.idB, ns, np);
        o.ori = r.inverted ? 'I' : 'N';
        o.k = r.K;
        if (na > nb) {
            o.id0 = (uint32_t)nb; o.id1 = (uint32_t)na;
            o.s0 = r.idB; o.s1 = r.idA;
            if (r.inverted) { o.oha = r.OHB; o.ohb = r.OHA; }
            else { o.oha = -(int64_t)r.OHA; o.ohb = -(int64_t)r.OHB; }
            o.ola = r.OLB; o.olb = r.OLA;
        } else {
            o.id0 = (uint32_t)na; o.id1 = (uint32_t)nb;
            o.s0 = r.idA; o.s1 = r.idB;
            o.oha = r.OHA; o.ohb = r.OHB;
            o.ola = r.OLA; o.olb = r.OLB;
        }
    };
    // 1. buckets of id0 ranges: splitters from a sample, a counting pass that notes every record's bucket
    std::vector<FatalError> errs(T, FatalError{0, ""});
    std::vector<uint32_t> split;  // bucket b holds id0 in [split[b-1], split[b])
    if (NB > 1) {
        std::vector<uint32_t> sample;
        const uint64_t step = std::max<uint64_t>(1, n / (NB * 64));
        for (uint64_t i = 0; i < n; i += step) sample.push_back(id0_of(recs[i]));
        std::sort(sample.begin(), sample.end());
        for (uint64_t b = 1; b < NB; b++) {
            const uint32_t v = sample[sample.size() * b / NB];
            if (split.empty() || v > split.back()) split.push_back(v);
        }
    }
    const unsigned B = (unsigned)split.size() + 1;
    auto bucket_of = [&](uint32_t id0) { return (uint16_t)(std::upper_bound(split.begin(), split.end(), id0) - split.begin()); };
    std::vector<std::vector<uint64_t>> counts(T, std::vector<uint64_t>(B, 0));
    std::unique_ptr<uint16_t[]> bucket(new uint16_t[n ? n : 1]);  // not initialised: written by the thread that reads it back
    workers(T, [&](unsigned t) {
        try {
            std::vector<uint64_t>& c = counts[t];
            for (uint64_t i = n * t / T; i < n * (t + 1) / T; i++) c[bucket[i] = bucket_of(id0_of(recs[i]))]++;
        } catch (const FatalError& e) {
            errs[t] = e;
        }
    });
    for (const FatalError& e : errs)
        if (e.status) throw e;
    std::vector<uint64_t> start(B + 1, 0);
    for (unsigned b = 0; b < B; b++) {
        uint64_t m = 0;
        for (unsigned t = 0; t < T; t++) {
            const uint64_t c = counts[t][b];
            counts[t][b] = start[b] + m;  // where thread t writes its records of bucket b
            m += c;
        }
        start[b + 1] = start[b] + m;
    }
    // 2. flip + scatter in one pass: the flipped form of a record is written once, into its bucket
    RecArray sorted_mem(n);
    BRec* sorted = sorted_mem.p;
    workers(T, [&](unsigned t) {
        std::vector<uint64_t> at = counts[t];
        for (uint64_t i = n * t / T; i < n * (t + 1) / T; i++) flip_into(recs[i], sorted[at[bucket[i]]++]);
    });
    bucket.reset();
    const auto t1 = std::chrono::steady_clock::now();
    // 3. sort and match every bucket on its own
    // (HC_SFO_VIA_MATCHER=<records per chunk>, test knob: only sort here, then hand the sorted run to the chunk-fed matcher
    // of hc_found_to_overlaps, which otherwise needs a device to be reached)
    const char* via_env = getenv("HC_SFO_VIA_MATCHER");
    const bool via_matcher = via_env != nullptr;
    std::vector<BucketResult> res(B);
    std::atomic<uint64_t> sort_ns{0}, match_ns{0};  // summed over the threads (HC_SFO_TIMING)
    {
        std::vector<unsigned> order(B);  // largest buckets first: they bound the makespan
        for (unsigned b = 0; b < B; b++) order[b] = b;
        std::sort(order.begin(), order.end(), [&](unsigned x, unsigned y) { return start[x + 1] - start[x] > start[y + 1] - start[y]; });
        std::atomic<unsigned> next{0};
        workers(std::min(T, B), [&](unsigned) {
            SortScratch scratch;
            for (;;) {
                const unsigned k = next.fetch_add(1);
                if (k >= B) return;
                const unsigned b = order[k];
                try {
                    const auto ta = std::chrono::steady_clock::now();
                    sort_bucket(sorted + start[b], start[b + 1] - start[b], scratch);
                    const auto tb = std::chrono::steady_clock::now();
                    if (!via_matcher) {
                        const BRec* base = sorted + start[b];
                        match_bucket([base](size_t i) -> const BRec& { return base[i]; }, start[b + 1] - start[b], ns, np, res[b]);
                    }
                    sort_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(tb - ta).count();
                    match_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - tb).count();
                } catch (const FatalError& e) {
                    res[b].error = e;
                }
            }
        });
    }
    if (via_matcher) {
        for (const BucketResult& r : res)
            if (r.error.status) throw r.error;
        const uint64_t chunk = std::max<uint64_t>(1, strtoull(via_env, nullptr, 10));
        SfoSortedMatcher m(ns, np);
        std::vector<SfoFlipped> buf;
        for (uint64_t at = 0; at < n; at += chunk) {
            const uint64_t c = std::min(chunk, n - at);
            buf.resize(c);
            for (uint64_t i = 0; i < c; i++) {
                const BRec& r = sorted[at + i];
                buf[i] = SfoFlipped{r.s0, r.s1, (int32_t)r.oha, (int32_t)r.ohb, r.ola, r.olb, r.k, r.ori == 'I' ? 1u : 0u};
            }
            m.feed(buf.data(), c);
        }
        return m.finish(n_lines);
    }
    // 4. stitch: the open group travels to the next bucket that holds a non-single record and is matched there
    const auto t2 = std::chrono::steady_clock::now();
    std::string out = stitch(res, n_lines);
    if (getenv("HC_SFO_TIMING"))
        fprintf(stderr, "sfo_records_to_overlaps: flip + partition %.3f s, sort + match %.3f s (thread-seconds: sort %.3f, match %.3f), stitch %.3f s (%u buckets, %u threads)\n",
                std::chrono::duration<double>(t1 - t0).count(), std::chrono::duration<double>(t2 - t1).count(), sort_ns.load() * 1e-9, match_ns.load() * 1e-9,
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t2).count(), B, T);
    return out;
}

// The same from records the device has flipped and sorted already (hc_found_to_overlaps): only the matching is left.
// The sorted run arrives in chunks; a chunk (behind what the previous one left over) is cut where the pair (id0, id1)
// changes — a group of paired candidates and a run of repeated records never straddle a cut —, the pieces are matched on
// their own by the threads, and the results are stitched at the end as above.  What follows a chunk's last cut waits for
// the next chunk.
struct SfoSortedMatcher::Impl {
    long ns, np;
    unsigned threads;
    std::vector<SfoFlipped> carry;
    std::vector<BucketResult> res;
    double match_s = 0;

    BRec brec(const SfoFlipped& f) const {
        BRec r;
        r.id0 = (uint32_t)original_id((long)f.s0, ns, np);
        r.id1 = (uint32_t)original_id((long)f.s1, ns, np);
        r.s0 = f.s0; r.s1 = f.s1;
        r.oha = f.oha; r.ohb = f.ohb;
        r.ola = f.ola; r.olb = f.olb; r.k = f.k;
        r.ori = f.inverted ? 'I' : 'N';
        return r;
    }
    bool same_pair(const SfoFlipped& x, const SfoFlipped& y) const {
        return original_id((long)x.s0, ns, np) == original_id((long)y.s0, ns, np) && original_id((long)x.s1, ns, np) == original_id((long)y.s1, ns, np);
    }
    // the records [0, total) of carry followed by recs, cut into pieces and matched
    void match(const SfoFlipped* recs, uint64_t total) {
        if (!total) return;
        const auto t0 = std::chrono::steady_clock::now();
        const uint64_t nc = carry.size();
        auto at = [&](uint64_t i) -> const SfoFlipped& { return i < nc ? carry[i] : recs[i - nc]; };
        uint64_t P = std::min<uint64_t>((uint64_t)threads * 4, total / 20000 + 1);
        if (const char* e = getenv("HC_SFO_BUCKETS")) P = (uint64_t)std::max(1, atoi(e));
        if (P > 4096) P = 4096;
        std::vector<uint64_t> start(P + 1, total);
        start[0] = 0;
        for (uint64_t b = 1; b < P; b++) {
            uint64_t cut = std::max(total * b / P, start[b - 1]);
            while (cut > 0 && cut < total && same_pair(at(cut), at(cut - 1))) cut++;
            start[b] = cut;
        }
        const size_t first = res.size();
        res.resize(first + P);
        std::atomic<uint64_t> next{0};
        run_workers((unsigned)std::min<uint64_t>(threads, P), [&](unsigned) {
            for (;;) {
                const uint64_t b = next.fetch_add(1);
                if (b >= P) return;
                try {
                    const uint64_t base = start[b];
                    match_bucket([&, base](size_t i) { return brec(at(base + i)); }, start[b + 1] - start[b], ns, np, res[first + b]);
                } catch (const FatalError& e) {
                    res[first + b].error = e;
                }
            }
        });
        match_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
};

SfoSortedMatcher::SfoSortedMatcher(long num_singles, long num_pairs) : impl_(new Impl) {
    impl_->ns = num_singles;
    impl_->np = num_pairs;
    unsigned T = std::thread::hardware_concurrency();
    if (T == 0) T = 1;
    if (T > 32) T = 32;
    impl_->threads = T;
}
SfoSortedMatcher::~SfoSortedMatcher() = default;

void SfoSortedMatcher::feed(const SfoFlipped* recs, uint64_t n) {
    Impl& m = *impl_;
    if (!n) return;
    // the last cut of this chunk: everything behind it belongs to a pair the next chunk may continue
    uint64_t e = n;
    while (e > 0 && m.same_pair(recs[e - 1], recs[n - 1])) e--;
    if (e == 0 && !m.carry.empty() && !m.same_pair(m.carry.back(), recs[0])) {
        m.match(recs, m.carry.size());  // the carry ends where this chunk starts a new pair
        m.carry.clear();
    }
    if (e > 0) {
        m.match(recs, m.carry.size() + e);
        m.carry.clear();
    }
    m.carry.insert(m.carry.end(), recs + e, recs + n);
}

std::string SfoSortedMatcher::finish(uint64_t& n_lines) {
    Impl& m = *impl_;
    m.match(nullptr, m.carry.size());
    m.carry.clear();
    const auto t0 = std::chrono::steady_clock::now();
    std::string out = stitch(m.res, n_lines);
    if (getenv("HC_SFO_TIMING"))
        fprintf(stderr, "sorted SFO records: matching %.3f s (%zu pieces, %u threads), stitch %.3f s\n", m.match_s, m.res.size(), m.threads,
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    return out;
}

std::string sfo_sorted_to_overlaps(const SfoFlipped* recs, uint64_t n, long ns, long np, uint64_t& n_lines) {
    SfoSortedMatcher m(ns, np);
    m.feed(recs, n);
    return m.finish(n_lines);
}

}  // namespace hc
