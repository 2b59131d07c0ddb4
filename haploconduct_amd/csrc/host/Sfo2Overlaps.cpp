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
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
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
    std::string k;     // last column, passed through untouched
    std::string text;  // the line as the script writes it to its temporary file (sort tie-break, uniq)
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

}  // namespace

// Returns the output text; n_lines receives the number of lines.
std::string sfo_to_overlaps(const std::string& sfo_text, long ns, long np, uint64_t& n_lines) {
    std::vector<SfoRec> recs;
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
        SfoRec r;
        long ida, idb;
        if (!parse_long(f[0], fl[0], ida) || !parse_long(f[1], fl[1], idb) || !parse_long(f[3], fl[3], r.oha) ||
            !parse_long(f[4], fl[4], r.ohb) || !parse_long(f[5], fl[5], r.ola) || !parse_long(f[6], fl[6], r.olb))
            throw FatalError{HC_ERR_FORMAT, "SFO line with a non-integer field"};
        const long na = original_id(ida, ns, np), nb = original_id(idb, ns, np);
        const std::string ori(f[2], fl[2]);
        r.ori = ori.size() == 1 ? ori[0] : '?';
        r.k.assign(f[7], fl[7]);
        char head[64];
        if (na > nb) {  // flip_N / flip_I, :112-122
            r.id[0] = nb; r.id[1] = na;
            r.sfo[0] = idb; r.sfo[1] = ida;
            if (ori == "I") std::swap(r.oha, r.ohb);
            else { r.oha = -r.oha; r.ohb = -r.ohb; }
            std::swap(r.ola, r.olb);
            char buf[256];
            const int k = snprintf(buf, sizeof buf, "%ld\t%ld\t%ld\t%ld\t%s\t%ld\t%ld\t%ld\t%ld\t%s\n", nb, na, idb, ida, ori.c_str(), r.oha,
                                   r.ohb, r.ola, r.olb, r.k.c_str());
            r.text.assign(buf, (size_t)k);
        } else {
            r.id[0] = na; r.id[1] = nb;
            r.sfo[0] = ida; r.sfo[1] = idb;
            const int k = snprintf(head, sizeof head, "%ld\t%ld\t", na, nb);
            r.text.assign(head, (size_t)k);
            r.text.append(line, len);  // the original line verbatim (its own separators), :48
            r.text.push_back('\n');
        }
        recs.push_back(std::move(r));
        pos = nl ? end + 1 : N;
    }
    // sort -k1,1n -k2,2n -k3,3n -k4,4n | uniq   (:53), LC_ALL=C
    std::sort(recs.begin(), recs.end(), [](const SfoRec& x, const SfoRec& y) {
        if (x.id[0] != y.id[0]) return x.id[0] < y.id[0];
        if (x.id[1] != y.id[1]) return x.id[1] < y.id[1];
        if (x.sfo[0] != y.sfo[0]) return x.sfo[0] < y.sfo[0];
        if (x.sfo[1] != y.sfo[1]) return x.sfo[1] < y.sfo[1];
        return x.text < y.text;
    });
    std::string out, last_line, cur;
    n_lines = 0;
    auto emit = [&](const std::string& l) {  // the final `uniq`, :107
        if (n_lines && l == last_line) return;
        out += l;
        last_line = l;
        n_lines++;
    };
    std::vector<const SfoRec*> cands;
    for (size_t i = 0; i < recs.size(); i++) {
        if (i && recs[i].text == recs[i - 1].text) continue;  // uniq
        const SfoRec& r = recs[i];
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

}  // namespace hc
