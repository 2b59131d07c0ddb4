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
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/hcedge_host.h"
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

// Returns the output text; n_lines receives the number of lines.
std::string sfo_to_overlaps(const std::string& sfo_text, long ns, long np, uint64_t& n_lines) {
    Ingest in;
    in.ns = ns;
    in.np = np;
    in.arena.reserve(sfo_text.size() + sfo_text.size() / 2);
    size_t pos = 0;
    const size_t N = sfo_text.size();
    while (pos < N) {  // :31-50
        const char* nl = (const char*)memchr(sfo_text.data() + pos, '\n', N - pos);
        const size_t end = nl ? (size_t)(nl - sfo_text.data()) : N;
        const char* line = sfo_text.data() + pos;
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

// The same from binary records (hc_find_overlaps): exactly what the text path yields for the file hc_host_write_sfo
// writes for them, without writing or parsing it.
std::string sfo_records_to_overlaps(const hc_sfo_rec* recs, uint64_t n, long ns, long np, uint64_t& n_lines) {
    const auto t0 = std::chrono::steady_clock::now();
    Ingest in;
    in.ns = ns;
    in.np = np;
    const unsigned T = worker_count(n);
    std::vector<Ingest> part(T);
    std::vector<FatalError> errs(T, FatalError{0, ""});
    auto build = [&](unsigned t) {
      try {
        Ingest& in = part[t];
        in.ns = ns;
        in.np = np;
        const uint64_t b = n * t / T, e = n * (t + 1) / T;
        in.recs.reserve(e - b);
        in.arena.reserve((e - b) * 48);
        char line[160], kbuf[16];
        for (uint64_t i = b; i < e; i++) {
        const hc_sfo_rec& r = recs[i];
        const char ori = r.inverted ? 'I' : 'N';
        const size_t kl = (size_t)(put_long(kbuf, (long)r.K) - kbuf);
        char* q = put_long(line, (long)r.idA);  // the line hc_host_write_sfo writes
        *q++ = '\t';
        q = put_long(q, (long)r.idB);
        *q++ = '\t';
        *q++ = ori;
        const long nums[5] = {r.OHA, r.OHB, (long)r.OLA, (long)r.OLB, (long)r.K};
        for (long v : nums) {
            *q++ = '\t';
            q = put_long(q, v);
        }
        in.add((long)r.idA, (long)r.idB, &ori, 1, r.OHA, r.OHB, (long)r.OLA, (long)r.OLB, kbuf, kl, line, (size_t)(q - line));
        }
      } catch (const FatalError& e) {
        errs[t] = e;
      }
    };
    if (T == 1) {
        build(0);
    } else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < T; t++) th.emplace_back(build, t);
        for (auto& x : th) x.join();
    }
    for (const FatalError& e : errs)
        if (e.status) throw e;
    {  // one arena, one record array: the pieces in order, text offsets rebased
        size_t n_recs = 0, n_text = 0;
        for (const Ingest& p : part) {
            n_recs += p.recs.size();
            n_text += p.arena.size();
        }
        in.recs.reserve(n_recs);
        in.arena.reserve(n_text);
        for (Ingest& p : part) {
            const size_t base = in.arena.size();
            in.arena += p.arena;
            for (SfoRec r : p.recs) {
                r.text_off += base;
                in.recs.push_back(r);
            }
            p = Ingest();
        }
    }
    const auto t1 = std::chrono::steady_clock::now();
    std::string out = in.finish(n_lines);
    if (getenv("HC_SFO_TIMING"))
        fprintf(stderr, "sfo_records_to_overlaps: build %.3f s, sort+match %.3f s\n", std::chrono::duration<double>(t1 - t0).count(),
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count());
    return out;
}

}  // namespace hc
