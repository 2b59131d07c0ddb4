// hc_sfo_kernels.hip — flip (scripts/sfo2overlaps.py:112-122) and sort keys of SFO records, one lane per record.
// Order of the script's temporary file: id0, id1, sfo0, sfo1 as numbers, then the line bytewise — ori ('I' < 'N'), then
// OHA OHB OLA OLB K as decimal texts, a tab ending the shorter one below '-' and every digit.
#include "hc_fno_device.h"
#include "hc_sfo_device.h"
#include "hc_text.h"

namespace hc {
namespace {
constexpr int kBlock = 256;

template <int D>
__device__ __forceinline__ uint64_t text_key_unsigned(uint32_t v) {  // base 11: digit d -> d + 1, nothing -> 0
    uint8_t dg[10];
    int nd = 0;
    do {
        dg[nd++] = (uint8_t)(v % 10);
        v /= 10;
    } while (v);
    uint64_t key = 0;
#pragma unroll
    for (int i = 0; i < D; i++) key = key * 11 + (i < nd ? (uint64_t)dg[nd - 1 - i] + 1 : 0);
    return key;
}
template <int D>
__device__ __forceinline__ uint64_t text_key_signed(int32_t x) {  // base 12: '-' -> 1, digit d -> d + 2, nothing -> 0
    uint32_t v = x < 0 ? 0u - (uint32_t)x : (uint32_t)x;
    uint8_t dg[10], ch[11];
    int nd = 0, n = 0;
    do {
        dg[nd++] = (uint8_t)(v % 10);
        v /= 10;
    } while (v);
    if (x < 0) ch[n++] = 1;
    for (int i = nd - 1; i >= 0; i--) ch[n++] = (uint8_t)(dg[i] + 2);
    uint64_t key = 0;
#pragma unroll
    for (int i = 0; i < D; i++) key = key * 12 + (i < n ? ch[i] : 0);
    return key;
}

__global__ __launch_bounds__(kBlock) void sfo_flip_kernel(const hc_sfo_rec* __restrict__ in, uint64_t n, uint64_t ns, uint64_t np,
                                                          SfoFlipped* __restrict__ out, uint64_t* __restrict__ k0, uint64_t* __restrict__ k1,
                                                          uint64_t* __restrict__ k2, uint32_t* __restrict__ iota,
                                                          unsigned long long* __restrict__ status) {
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const hc_sfo_rec r = in[i];
    unsigned long long bad = 0;
    // original_id, :136-147
    uint64_t na = r.idA, nb = r.idB;
    if (np) {
        if (na >= ns + 2 * np || nb >= ns + 2 * np) bad |= kSfoStatusId;
        na = na < ns + np ? na : na - np;
        nb = nb < ns + np ? nb : nb - np;
    }
    SfoFlipped o;
    o.k = r.K;
    o.inverted = r.inverted ? 1u : 0u;
    uint64_t id0, id1;
    if (na > nb) {  // flip_N / flip_I
        id0 = nb;
        id1 = na;
        o.s0 = r.idB;
        o.s1 = r.idA;
        if (r.inverted) {
            o.oha = r.OHB;
            o.ohb = r.OHA;
        } else {
            if (r.OHA == INT32_MIN || r.OHB == INT32_MIN) bad |= kSfoStatusRange;
            o.oha = -r.OHA;
            o.ohb = -r.OHB;
        }
        o.ola = r.OLB;
        o.olb = r.OLA;
    } else {
        id0 = na;
        id1 = nb;
        o.s0 = r.idA;
        o.s1 = r.idB;
        o.oha = r.OHA;
        o.ohb = r.OHB;
        o.ola = r.OLA;
        o.olb = r.OLB;
    }
    const int32_t lim = 9999999;
    if (o.oha > lim || o.oha < -lim || o.ohb > lim || o.ohb < -lim || o.ola > (uint32_t)lim || o.olb > (uint32_t)lim || o.k > 9999u)
        bad |= kSfoStatusRange;
    out[i] = o;
    iota[i] = (uint32_t)i;
    k2[i] = id0 << 32 | id1;
    const uint64_t sb0 = o.s0 != id0, sb1 = o.s1 != id1, ori = o.inverted ? 0 : 1;  // 'I' < 'N'
    k1[i] = sb0 << 60 | sb1 << 59 | ori << 58 | text_key_signed<8>(o.oha) << 29 | text_key_signed<8>(o.ohb);  // 12^8 < 2^29
    k0[i] = text_key_unsigned<7>(o.ola) << 39 | text_key_unsigned<7>(o.olb) << 14 | text_key_unsigned<4>(o.k);  // 11^7 < 2^25, 11^4 < 2^14
    if (bad) atomicOr(status, bad);
}

__global__ __launch_bounds__(kBlock) void sfo_gather_kernel(const SfoFlipped* __restrict__ in, const uint32_t* __restrict__ perm, uint64_t n,
                                                            SfoFlipped* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const uint4* src = reinterpret_cast<const uint4*>(in + perm[i]);
    uint4* dst = reinterpret_cast<uint4*>(out + i);
    dst[0] = src[0];
    dst[1] = src[1];
}
// ---- which sorted records the matching can see -------------------------------------------------------------------------------
// scripts/sfo2overlaps.py:63-103 after its `uniq`: a line whose two original ids are equal is skipped; a line between two
// unpaired reads is an output line; the others collect in groups of one pair of reads, and a group is matched (every two of
// its lines, :221-310) when the NEXT such line arrives, with that line's read types (:94).  A group of one line yields nothing
// whoever closes it, so of the grouped lines only those of groups of two and more, and the line that closes such a group, have
// to reach the host's matcher: on overlaps of paired reads at one error rate that is a few per cent of the records.
__device__ __forceinline__ uint32_t sfo_original(uint32_t sfo, uint64_t ns, uint64_t np) {  // :136-147
    return (np == 0 || sfo < ns + np) ? sfo : (uint32_t)(sfo - np);
}
__device__ __forceinline__ bool sfo_same_pair(const SfoFlipped& a, const SfoFlipped& b, uint64_t ns, uint64_t np) {
    return sfo_original(a.s0, ns, np) == sfo_original(b.s0, ns, np) && sfo_original(a.s1, ns, np) == sfo_original(b.s1, ns, np);
}

__global__ __launch_bounds__(kBlock) void sfo_classify_kernel(const SfoFlipped* __restrict__ sorted, uint64_t n, uint64_t ns, uint64_t np,
                                                              uint8_t* __restrict__ grouped, uint8_t* __restrict__ keep) {
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const SfoFlipped r = sorted[i];
    bool dup = false;  // `uniq`: equal to the line in front of it
    if (i) {
        const SfoFlipped p = sorted[i - 1];
        dup = p.s0 == r.s0 && p.s1 == r.s1 && p.oha == r.oha && p.ohb == r.ohb && p.ola == r.ola && p.olb == r.olb && p.k == r.k &&
              p.inverted == r.inverted;
    }
    const uint32_t id0 = sfo_original(r.s0, ns, np), id1 = sfo_original(r.s1, ns, np);
    const bool seen = !dup && id0 != id1;                                  // :66-67
    const bool single = np == 0 || (id0 < ns && id1 < ns);                // is_paired, :124-134
    grouped[i] = seen && !single;
    keep[i] = seen && single;  // an output line by itself; the grouped ones are decided below
}

// j-th grouped record (idx[j] = its place among the sorted ones): kept when its group has two lines or more, or when it closes one
__global__ __launch_bounds__(kBlock) void sfo_groups_kernel(const SfoFlipped* __restrict__ sorted, const uint32_t* __restrict__ idx, uint64_t m,
                                                            uint64_t ns, uint64_t np, uint8_t* __restrict__ keep) {
    const uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= m) return;
    const SfoFlipped r = sorted[idx[j]];
    const bool same_prev = j > 0 && sfo_same_pair(sorted[idx[j - 1]], r, ns, np);
    const bool same_next = j + 1 < m && sfo_same_pair(r, sorted[idx[j + 1]], ns, np);
    bool closes = false;  // the group in front of it has two lines or more
    if (j > 1 && !same_prev) closes = sfo_same_pair(sorted[idx[j - 2]], sorted[idx[j - 1]], ns, np);
    keep[idx[j]] = (same_prev || same_next || closes) ? 1 : 0;
}

__global__ __launch_bounds__(kBlock) void sfo_gather_kept_kernel(const SfoFlipped* __restrict__ sorted, const uint32_t* __restrict__ idx, uint64_t k,
                                                                 SfoFlipped* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < k) out[i] = sorted[idx[i]];
}
// ---- the script's matching on the device (round 6: the device-resident stage a) ---------------------------------------------------
// What host/Sfo2Overlaps.cpp does on the host threads — scripts/sfo2overlaps.py:63-103 (the loop), :150-200 (get_s_s_overlap), :203-219
// (match_candidates), :221-310 (find_paired_overlap), :312-329 (merge_overlaps) — over the sorted records where they are, writing the
// overlap LINES as records (hc_line_rec: what the stage's parser would read from the script's text) in the order the script writes them.
// Asserts and the division by zero of the script are reported through `status` (the caller falls back to the host's matcher, which
// raises what the script raises).
enum : unsigned long long { kSfoStatusText = 8 };   // a line of the SFO text that is not canonical (or more lines than counted): the host's general path
enum : unsigned long long { kSfoStatusMatch = 4 };  // an assert of the script's matching (or its division by zero), or a group beyond kSfoMaxGroup
constexpr uint32_t kSfoMaxGroup = 2048;             // lines of one pair of reads a lane matches (every two of them: 2 * 10^6 pairs)

struct SfoSS {  // get_s_s_overlap, :150-200
    uint32_t id1, id2, pos1, perc, len;
    uint8_t ori1, ori2;
    bool bad;
};
__device__ __forceinline__ SfoSS sfo_ss(const SfoFlipped& r, uint32_t id0, uint32_t id1) {
    SfoSS o;
    const uint8_t ori = r.inverted ? '-' : '+';
    const long ola = (long)r.ola, olb = (long)r.olb, oha = r.oha, ohb = r.ohb;
    const long ovlen = ola < olb ? ola : olb;
    long la, lb;
    if (oha >= 0) {  // read A is first
        if (ohb >= 0) {
            la = ola + oha;
            lb = olb + ohb;
        } else {
            la = ola + oha - ohb;
            lb = olb;
        }
        o.id1 = id0;
        o.id2 = id1;
        o.pos1 = (uint32_t)oha;
        o.ori1 = '+';
        o.ori2 = ori;
    } else {  // read B is first
        if (ohb >= 0) {
            la = ola;
            lb = -oha + olb + ohb;
        } else {
            la = ola - ohb;
            lb = -oha + olb;
        }
        o.id1 = id1;
        o.id2 = id0;
        o.pos1 = (uint32_t)(-oha);
        o.ori1 = ori;
        o.ori2 = '+';
    }
    const long minlen = la < lb ? la : lb;
    o.bad = minlen <= 0;  // division by zero / `assert minreadlen > 0`
    // min(round(100 * ovlen / minreadlen), 100) under `from __future__ import division`: the correctly rounded quotient of two exact
    // doubles, Python 2's round() = C round() (half away from zero)
    const double q = o.bad ? 0.0 : round(100.0 * (double)ovlen / (double)minlen);
    o.perc = (uint32_t)(q < 100.0 ? q : 100.0);
    o.len = (uint32_t)ovlen;
    return o;
}

// find_paired_overlap + merge_overlaps for candidates c1, c2 (in the group's order) with the CLOSING line's read types; false: no line
__device__ __forceinline__ bool sfo_paired_line(const SfoFlipped& c1, const SfoFlipped& c2, uint32_t id0, uint32_t id1, bool type_a, bool type_b,
                                                hc_line_rec& out, bool& bad) {
    if (c1.inverted != c2.inverted) return false;
    const long a1 = c1.s0, b1 = c1.s1, a2 = c2.s0, b2 = c2.s1;
    const bool normal = !c1.inverted;
    int first = 0;  // which candidate provides overlap1
    if (type_a && type_b) {
        if (normal) first = (a1 < a2 && b1 < b2) ? 1 : ((a1 > a2 && b1 > b2) ? 2 : 0);
        else first = (a1 < a2 && b1 > b2) ? 1 : ((a1 > a2 && b1 < b2) ? 2 : 0);
    } else {
        const long p1 = c1.oha, p2 = c2.oha;
        const long k1 = (type_a && !type_b) ? a1 : b1, k2 = (type_a && !type_b) ? a2 : b2;
        if (normal) first = (k1 < k2 && p1 < p2) ? 1 : ((k1 > k2 && p1 > p2) ? 2 : 0);
        else first = (k1 < k2 && p1 > p2) ? 2 : ((k1 > k2 && p1 < p2) ? 1 : 0);
    }
    if (!first) return false;
    const SfoSS o1 = sfo_ss(first == 1 ? c1 : c2, id0, id1), o2 = sfo_ss(first == 1 ? c2 : c1, id0, id1);
    if (o1.bad || o2.bad) {
        bad = true;
        return false;
    }
    uint8_t t1, t2;
    if (o1.id1 == id0) {  // (both candidates name the group's pair of reads: o1.id2 == id1 follows)
        t1 = type_a ? 'p' : 's';
        t2 = type_b ? 'p' : 's';
    } else {
        t1 = type_b ? 'p' : 's';
        t2 = type_a ? 'p' : 's';
    }
    uint8_t ord = '-';
    if (t1 == 'p' && t2 == 'p') ord = o1.id1 != o2.id1 ? '2' : '1';  // (o1.id1 != o2.id1 implies o1.id1 == o2.id2: the same two reads)
    out.id1 = o1.id1;
    out.id2 = o1.id2;
    out.pos1 = o1.pos1;
    out.pos2 = o2.pos1;
    out.perc1 = o1.perc;
    out.perc2 = o2.perc;
    out.len1 = o1.len;
    out.len2 = o2.len;
    out.ord = ord;
    out.ori1 = o1.ori1;
    out.ori2 = o1.ori2;
    out.type1 = t1;
    out.type2 = t2;
    out.pad[0] = out.pad[1] = out.pad[2] = 0;
    return true;
}

// the j-th grouped record opens a group
__global__ __launch_bounds__(kBlock) void sfo_group_starts_kernel(const SfoFlipped* __restrict__ sorted, const uint32_t* __restrict__ idx, uint64_t m,
                                                                  uint64_t ns, uint64_t np, uint8_t* __restrict__ start) {
    const uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= m) return;
    start[j] = (j == 0 || !sfo_same_pair(sorted[idx[j - 1]], sorted[idx[j]], ns, np)) ? 1 : 0;
}

// Group g (1 <= g < G) is matched when the line that opens group g — the TRIGGER, whose read types the script passes on (:94) —
// arrives: every two lines of group g - 1, in the script's order.  WRITE = false: emit[trigger] = the number of lines that come out;
// WRITE = true: the lines themselves at off[trigger]....  One lane per group (groups of two lines are the rule).
template <bool WRITE>
__global__ __launch_bounds__(kBlock) void sfo_match_groups_kernel(const SfoFlipped* __restrict__ sorted, const uint32_t* __restrict__ idx,
                                                                  const uint32_t* __restrict__ starts, uint64_t G, uint64_t ns, uint64_t np,
                                                                  uint32_t* __restrict__ emit, const uint32_t* __restrict__ off,
                                                                  hc_line_rec* __restrict__ lines, unsigned long long* __restrict__ status) {
    const uint64_t g = (uint64_t)blockIdx.x * kBlock + threadIdx.x + 1;
    if (g >= G) return;
    const uint32_t b = starts[g - 1], e = starts[g];  // the group in front of the trigger: grouped records [b, e)
    const uint32_t trig = idx[e];
    if (e - b < 2) {
        if (!WRITE) emit[trig] = 0;
        return;
    }
    if (e - b > kSfoMaxGroup) {  // thousands of lines for ONE pair of reads (millions of pairs to try): not a lane's work — the host's matcher takes the input
        if (!WRITE) emit[trig] = 0;
        atomicOr(status, (unsigned long long)kSfoStatusMatch);
        return;
    }
    const SfoFlipped t = sorted[trig];
    const uint32_t tid0 = sfo_original(t.s0, ns, np), tid1 = sfo_original(t.s1, ns, np);
    const bool type_a = np != 0 && tid0 >= ns, type_b = np != 0 && tid1 >= ns;  // is_paired of the CLOSING line's reads
    const SfoFlipped c0 = sorted[idx[b]];
    const uint32_t id0 = sfo_original(c0.s0, ns, np), id1 = sfo_original(c0.s1, ns, np);
    uint32_t count = 0;
    bool bad = false;
    uint32_t at = WRITE ? off[trig] : 0u;
    for (uint32_t x = b; x < e; x++) {
        const SfoFlipped cx = x == b ? c0 : sorted[idx[x]];
        for (uint32_t y = x + 1; y < e; y++) {
            hc_line_rec line;
            if (sfo_paired_line(cx, sorted[idx[y]], id0, id1, type_a, type_b, line, bad)) {
                if (WRITE) lines[at + count] = line;
                count++;
            }
        }
    }
    if (!WRITE) emit[trig] = count;
    if (bad) atomicOr(status, (unsigned long long)kSfoStatusMatch);
}

// a line between two unpaired reads is an output line by itself (:79-85)
__global__ __launch_bounds__(kBlock) void sfo_single_lines_kernel(const SfoFlipped* __restrict__ sorted, const uint8_t* __restrict__ keep, uint64_t n,
                                                                  uint64_t ns, uint64_t np, uint32_t* __restrict__ emit, const uint32_t* __restrict__ off,
                                                                  hc_line_rec* __restrict__ lines, unsigned long long* __restrict__ status, int write) {
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n || !keep[i]) return;
    if (!write) {
        emit[i] = 1;
        return;
    }
    const SfoFlipped r = sorted[i];
    const SfoSS o = sfo_ss(r, sfo_original(r.s0, ns, np), sfo_original(r.s1, ns, np));
    if (o.bad) atomicOr(status, (unsigned long long)kSfoStatusMatch);
    hc_line_rec line;
    line.id1 = o.id1;
    line.id2 = o.id2;
    line.pos1 = o.pos1;
    line.pos2 = 0;
    line.perc1 = o.perc;
    line.perc2 = 0;
    line.len1 = o.len;
    line.len2 = 0;
    line.ord = '-';
    line.ori1 = o.ori1;
    line.ori2 = o.ori2;
    line.type1 = 's';
    line.type2 = 's';
    line.pad[0] = line.pad[1] = line.pad[2] = 0;
    lines[off[i]] = line;
}

// the script's last `uniq` (:107): a line equal to the one in front of it goes (keep[k] = 0)
__global__ __launch_bounds__(kBlock) void sfo_uniq_lines_kernel(const hc_line_rec* __restrict__ lines, uint64_t n, uint8_t* __restrict__ keep,
                                                                unsigned long long* __restrict__ n_dup) {
    const uint64_t k = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (k >= n) return;
    bool dup = false;
    if (k) {
        const uint4* a = reinterpret_cast<const uint4*>(lines + k);
        const uint4* b = reinterpret_cast<const uint4*>(lines + k - 1);
        dup = true;
#pragma unroll
        for (int w = 0; w < 3; w++) dup = dup && a[w].x == b[w].x && a[w].y == b[w].y && a[w].z == b[w].z && a[w].w == b[w].w;
    }
    keep[k] = dup ? 0 : 1;
    if (dup) atomicAdd(n_dup, 1ull);
}

__global__ __launch_bounds__(kBlock) void sfo_gather_lines_kernel(const hc_line_rec* __restrict__ in, const uint32_t* __restrict__ idx, uint64_t k,
                                                                  hc_line_rec* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= k) return;
    const uint4* src = reinterpret_cast<const uint4*>(in + idx[i]);
    uint4* dst = reinterpret_cast<uint4*>(out + i);
    dst[0] = src[0];
    dst[1] = src[1];
    dst[2] = src[2];
}
}  // namespace

// ---- the SFO file's text read on the device (round 6) -----------------------------------------------------------------------------
// One lane per line of a chunk of the file (line starts: the overlaps file's own kernels, hc_text_kernels.hip): a CANONICAL line — what
// rust-overlaps and hc_host_write_sfo write: eight fields, single tabs, "0" or [1-9][0-9]* (one '-' allowed in front of the two overhangs),
// `N` or `I` — becomes an hc_sfo_rec at out[*lines_before + i]; any other line raises `status` and the whole file goes to the host's general
// path (host/Sfo2Overlaps.cpp: parse_canonical_sfo is this kernel's twin on the host threads, same acceptance rules).
struct SfoTextCursor {
    const char* text;
    uint32_t at, e;
    uint32_t w;
    __device__ __forceinline__ void load() { w = *(const uint32_t*)(text + (at & ~3u)); }
    __device__ __forceinline__ uint32_t cur() const { return (w >> (8u * (at & 3u))) & 0xFFu; }
    __device__ __forceinline__ void next() {
        at++;
        if ((at & 3u) == 0) w = *(const uint32_t*)(text + at);
    }
};
// one canonical number followed by `end` ('\t', or 0 = the line's end); lo <= value <= hi
__device__ __forceinline__ bool sfo_take_int(SfoTextCursor& c, bool allow_negative, bool last, long long lo, long long hi, long long& v) {
    bool neg = false;
    if (c.at < c.e && c.cur() == '-') {
        if (!allow_negative) return false;
        neg = true;
        c.next();
    }
    const uint32_t b = c.at;
    const uint32_t first = c.at < c.e ? c.cur() : 0u;
    unsigned long long u = 0;
    while (c.at < c.e && (c.cur() - '0') <= 9u && c.at - b < 11u) {
        u = u * 10ull + (unsigned long long)(c.cur() - '0');
        c.next();
    }
    const uint32_t d = c.at - b;
    if (d == 0 || d > 10 || (first == '0' && (d > 1 || neg))) return false;  // no leading zeros, no "-0"
    v = neg ? -(long long)u : (long long)u;
    if (v < lo || v > hi) return false;
    if (last) return c.at == c.e;
    if (c.at >= c.e || c.cur() != '\t') return false;
    c.next();
    return true;
}

__global__ __launch_bounds__(kBlock) void sfo_parse_text_kernel(const char* __restrict__ text, const uint32_t* __restrict__ line_start,
                                                                const unsigned long long* __restrict__ counters /* hc_text.h */,
                                                                const unsigned long long* __restrict__ lines_before, hc_sfo_rec* __restrict__ out,
                                                                uint64_t out_cap, unsigned long long* __restrict__ status) {
    if (counters[kTextOverflow]) {
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(status, (unsigned long long)kSfoStatusText);
        return;
    }
    const uint32_t n = (uint32_t)counters[kTextLines];
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const uint64_t at_out = *lines_before + i;
    SfoTextCursor c{text, line_start[i], line_start[i + 1] - 1u, 0u};
    c.load();
    long long a, b, oha, ohb, ola, olb, k;
    bool good = sfo_take_int(c, false, false, 0, 0xFFFFFFFFll, a) && sfo_take_int(c, false, false, 0, 0xFFFFFFFFll, b);
    uint32_t ori = 0;
    if (good) {
        ori = c.at < c.e ? c.cur() : 0u;
        good = ori == 'N' || ori == 'I';
        c.next();
        good = good && c.at < c.e && c.cur() == '\t';
        c.next();
    }
    good = good && sfo_take_int(c, true, false, -2147483647ll, 2147483647ll, oha) && sfo_take_int(c, true, false, -2147483647ll, 2147483647ll, ohb) &&
           sfo_take_int(c, false, false, 0, 0xFFFFFFFFll, ola) && sfo_take_int(c, false, false, 0, 0xFFFFFFFFll, olb) &&
           sfo_take_int(c, false, true, 0, 0xFFFFFFFFll, k);
    if (!good || at_out >= out_cap) {
        atomicOr(status, (unsigned long long)kSfoStatusText);
        return;
    }
    hc_sfo_rec r;
    r.idA = (uint32_t)a;
    r.idB = (uint32_t)b;
    r.OHA = (int32_t)oha;
    r.OHB = (int32_t)ohb;
    r.OLA = (uint32_t)ola;
    r.OLB = (uint32_t)olb;
    r.K = (uint32_t)k;
    r.inverted = ori == 'I' ? 1u : 0u;
    out[at_out] = r;
}

#define HC_SFO_GRID(n) dim3((unsigned)(((n) + kBlock - 1) / kBlock)), dim3(kBlock)
hipError_t sfo_parse_text(const char* text, const uint32_t* line_start, uint32_t max_lines, const unsigned long long* counters,
                          const unsigned long long* lines_before, hc_sfo_rec* out, uint64_t out_cap, unsigned long long* status, hipStream_t s) {
    if (!max_lines) return hipSuccess;
    hipLaunchKernelGGL(sfo_parse_text_kernel, HC_SFO_GRID(max_lines), 0, s, text, line_start, counters, lines_before, out, out_cap, status);
    return hipGetLastError();
}
hipError_t sfo_group_starts(const SfoFlipped* sorted, const uint32_t* idx, uint64_t m, uint64_t ns, uint64_t np, uint8_t* start, hipStream_t s) {
    if (!m) return hipSuccess;
    hipLaunchKernelGGL(sfo_group_starts_kernel, HC_SFO_GRID(m), 0, s, sorted, idx, m, ns, np, start);
    return hipGetLastError();
}
hipError_t sfo_match_groups(bool write, const SfoFlipped* sorted, const uint32_t* idx, const uint32_t* starts, uint64_t G, uint64_t ns, uint64_t np,
                            uint32_t* emit, const uint32_t* off, hc_line_rec* lines, unsigned long long* status, hipStream_t s) {
    if (G < 2) return hipSuccess;
    if (write) hipLaunchKernelGGL((sfo_match_groups_kernel<true>), HC_SFO_GRID(G - 1), 0, s, sorted, idx, starts, G, ns, np, emit, off, lines, status);
    else hipLaunchKernelGGL((sfo_match_groups_kernel<false>), HC_SFO_GRID(G - 1), 0, s, sorted, idx, starts, G, ns, np, emit, off, lines, status);
    return hipGetLastError();
}
hipError_t sfo_single_lines(bool write, const SfoFlipped* sorted, const uint8_t* keep, uint64_t n, uint64_t ns, uint64_t np, uint32_t* emit,
                            const uint32_t* off, hc_line_rec* lines, unsigned long long* status, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(sfo_single_lines_kernel, HC_SFO_GRID(n), 0, s, sorted, keep, n, ns, np, emit, off, lines, status, write ? 1 : 0);
    return hipGetLastError();
}
hipError_t sfo_uniq_lines(const hc_line_rec* lines, uint64_t n, uint8_t* keep, unsigned long long* n_dup, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(sfo_uniq_lines_kernel, HC_SFO_GRID(n), 0, s, lines, n, keep, n_dup);
    return hipGetLastError();
}
hipError_t sfo_gather_lines(const hc_line_rec* in, const uint32_t* idx, uint64_t k, hc_line_rec* out, hipStream_t s) {
    if (!k) return hipSuccess;
    hipLaunchKernelGGL(sfo_gather_lines_kernel, HC_SFO_GRID(k), 0, s, in, idx, k, out);
    return hipGetLastError();
}

hipError_t sfo_flip(const hc_sfo_rec* in, uint64_t n, uint64_t ns, uint64_t np, SfoFlipped* out, uint64_t* k0, uint64_t* k1, uint64_t* k2,
                    uint32_t* iota, unsigned long long* status, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(sfo_flip_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, in, n, ns, np, out, k0, k1, k2, iota, status);
    return hipGetLastError();
}
hipError_t sfo_gather(const SfoFlipped* in, const uint32_t* perm, uint64_t n, SfoFlipped* out, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(sfo_gather_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, in, perm, n, out);
    return hipGetLastError();
}

hipError_t sfo_classify(const SfoFlipped* sorted, uint64_t n, uint64_t ns, uint64_t np, uint8_t* grouped, uint8_t* keep, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(sfo_classify_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, sorted, n, ns, np, grouped, keep);
    return hipGetLastError();
}
hipError_t sfo_groups(const SfoFlipped* sorted, const uint32_t* idx, uint64_t m, uint64_t ns, uint64_t np, uint8_t* keep, hipStream_t s) {
    if (!m) return hipSuccess;
    hipLaunchKernelGGL(sfo_groups_kernel, dim3((unsigned)((m + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, sorted, idx, m, ns, np, keep);
    return hipGetLastError();
}
hipError_t sfo_gather_kept(const SfoFlipped* sorted, const uint32_t* idx, uint64_t k, SfoFlipped* out, hipStream_t s) {
    if (!k) return hipSuccess;
    hipLaunchKernelGGL(sfo_gather_kept_kernel, dim3((unsigned)((k + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, sorted, idx, k, out);
    return hipGetLastError();
}

}  // namespace hc
