// hc_fno_kernels.hip — find-next-overlaps on the device (hc_fno_device.h): one lane per combination deduces its line's
// columns (computeOverlapData, src/FindNextOverlaps.cpp:351-565) and the line's 256-bit sort key; after four stable
// radix sorts (least significant 64 bits first) one lane per sorted place drops repeated lines and measures the text;
// after a scan one lane per line writes it.  Integer work, one float division per percentage (correctly rounded on
// both sides, -ffp-contract=off).
#include "hc_fno_device.h"

namespace hc {
namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }

// (int)floor(std::max(ov/float(la), ov/float(lb))*100), src/FindNextOverlaps.cpp:375,429,487,549
__device__ __forceinline__ int perc_max(int ov, int la, int lb) {
    const float a = (float)ov / (float)la, b = (float)ov / (float)lb;
    const float m = (a < b ? b : a) * 100.0f;
    return (int)floorf(m);
}

struct Induced {
    int pos1, pos2, perc, len1, len2;
    uint8_t ord1, ord2, type1, type2;
};

// 0 = a line, 1 = no line ("failure": trimmed away / should have been merged), 2 = the reference would stop
__device__ int induced_overlap(const FnoItem& it, Induced& o) {
    const int e_pos1 = it.v[0], e_pos2 = it.v[1], i1l = it.v[2], i1r = it.v[3], i2l = it.v[4], i2r = it.v[5];
    const int a1 = it.v[6], a2 = it.v[7], b1 = it.v[8], b2 = it.v[9];
    const bool ap = it.a_paired != 0, bp = it.b_paired != 0;
    const int shift1 = (e_pos1 + i1l) - i2l;
    o.ord1 = shift1 < 0 ? '2' : '1';
    o.pos1 = shift1 < 0 ? -shift1 : shift1;
    o.type1 = ap ? 'p' : 's';
    o.type2 = bp ? 'p' : 's';
    if (!ap && !bp) {  // :358-385
        if (!(a1 > 0 && b1 > 0)) return 2;
        const int len = shift1 < 0 ? b1 : a1;
        o.len1 = imin(imin(len - o.pos1, a1), b1);
        o.len2 = 0;
        o.perc = perc_max(o.len1, a1, b1);
        o.ord2 = '-';
        o.pos2 = 0;
        if (o.pos1 >= len) return 1;
    } else if (ap != bp) {  // P-S :387-440, S-P :442-486
        const int P1 = ap ? a1 : b1, P2 = ap ? a2 : b2, S1 = ap ? b1 : a1;
        if (!(P1 + P2 > 0 && S1 > 0)) return 2;
        const bool starts_in_pair = ap ? (shift1 >= 0) : (shift1 < 0);
        if (o.pos1 >= (starts_in_pair ? P1 : S1)) return 1;
        o.len1 = starts_in_pair ? P1 - o.pos1 : P1;
        if (ap) o.pos2 = it.e_ord == '1' ? i2r - (i1r + e_pos2) : (e_pos2 + i2r) - i1r;
        else o.pos2 = it.e_ord == '2' ? i1r - (e_pos2 + i2r) : i1r + e_pos2 - i2r;
        if (o.pos2 >= S1 || o.pos2 < 0) return 1;
        o.ord2 = '-';
        o.len2 = imin(S1 - o.pos2, P2);
        const int total = o.len1 + o.len2;
        o.perc = imin(ap ? perc_max(total, P1 + P2, S1) : perc_max(total, S1, P1 + P2), 100);
    } else {  // P-P :488-547
        if (o.pos1 >= (shift1 < 0 ? b1 : a1)) return 1;
        o.len1 = shift1 < 0 ? imin(a1, b1 - o.pos1) : imin(a1 - o.pos1, b1);
        const int shift2 = it.e_ord == '1' ? (e_pos2 + i1r) - i2r : i1r - (e_pos2 + i2r);
        if (shift2 < 0) {
            o.ord2 = o.ord1 == '1' ? '2' : '1';
            o.pos2 = -shift2;
            if (o.pos2 >= b2) return 1;
            o.len2 = imin(a2, b2 - o.pos2);
        } else {
            o.ord2 = o.ord1 == '1' ? '1' : '2';
            o.pos2 = shift2;
            if (o.pos2 >= a2) return 1;
            o.len2 = imin(a2 - o.pos2, b2);
        }
        o.perc = imin(perc_max(o.len1 + o.len2, a1 + a2, b1 + b2), 100);
    }
    if (!(o.perc >= 0 && o.perc <= 100)) return 2;  // :562
    return 0;
}

// ---- decimal text <-> order-preserving integers ----------------------------------------------------------------
// the text of v followed by a tab, read as a number in base 11 with D places: digit d -> d + 1, nothing -> 0
template <int D>
__device__ __forceinline__ uint64_t dec_key_unsigned(uint64_t v) {
    uint8_t dg[20];
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
// the same with a sign: base 12, '-' -> 1, digit d -> d + 2, 11 places ("-2147483648")
__device__ __forceinline__ uint64_t dec_key_signed(int32_t x) {
    uint8_t ch[11];
    int n = 0;
    uint32_t v = x < 0 ? 0u - (uint32_t)x : (uint32_t)x;
    uint8_t dg[10];
    int nd = 0;
    do {
        dg[nd++] = (uint8_t)(v % 10);
        v /= 10;
    } while (v);
    if (x < 0) ch[n++] = 1;
    for (int i = nd - 1; i >= 0; i--) ch[n++] = (uint8_t)(dg[i] + 2);
    uint64_t key = 0;
#pragma unroll
    for (int i = 0; i < 11; i++) key = key * 12 + (i < n ? ch[i] : 0);
    return key;
}
__device__ __forceinline__ uint32_t digits_u64(uint64_t v) {
    uint32_t n = 1;
    while (v >= 10) {
        v /= 10;
        n++;
    }
    return n;
}
__device__ __forceinline__ uint32_t chars_i32(int32_t x) {
    const uint32_t v = x < 0 ? 0u - (uint32_t)x : (uint32_t)x;
    return digits_u64(v) + (x < 0 ? 1u : 0u);
}
__device__ __forceinline__ char* put_u64(char* p, uint64_t v) {
    char tmp[20];
    int n = 0;
    do {
        tmp[n++] = (char)('0' + v % 10);
        v /= 10;
    } while (v);
    while (n) *p++ = tmp[--n];
    return p;
}
__device__ __forceinline__ char* put_i32(char* p, int32_t x) {  // std::to_string(int)
    if (x < 0) *p++ = '-';
    return put_u64(p, x < 0 ? 0u - (uint32_t)x : (uint32_t)x);
}

__global__ __launch_bounds__(kBlock) void fno_deduce_kernel(const FnoItem* __restrict__ items, uint64_t n, uint32_t no_inclusions,
                                                            FnoRec* __restrict__ rec, uint64_t* __restrict__ k0, uint64_t* __restrict__ k1,
                                                            uint64_t* __restrict__ k2, uint64_t* __restrict__ k3, uint32_t* __restrict__ iota,
                                                            unsigned long long* __restrict__ counters) {
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    __shared__ unsigned int tally[4];
    __shared__ unsigned int status;
    if (threadIdx.x < 4) tally[threadIdx.x] = 0;
    if (threadIdx.x == 4) status = 0;
    __syncthreads();
    if (i < n) {
        const FnoItem it = items[i];
        FnoRec r;
        r.pad = 0;
        r.kind = it.kind;
        r.ori1 = it.ori1;
        r.ori2 = it.ori2;
        unsigned bad = 0;
        bool line = true;
        if (it.kind == 0) {  // :44-68
            r.id1 = it.ida;
            r.id2 = it.idb;
            r.pos1 = it.v[0];
            r.pos2 = it.v[1];
            r.perc = it.v[2];
            r.len1 = it.v[3];
            r.len2 = it.v[4];
            r.ord2 = it.e_ord;
            r.type1 = it.a_paired ? 'p' : 's';
            r.type2 = it.b_paired ? 'p' : 's';
        } else {
            Induced o;
            const int what = induced_overlap(it, o);
            if (what == 2) bad |= (unsigned)kFnoStatusRequire;
            line = what == 0;
            const bool fwd = o.ord1 == '1';
            r.id1 = fwd ? it.ida : it.idb;
            r.id2 = fwd ? it.idb : it.ida;
            r.type1 = fwd ? o.type1 : o.type2;
            r.type2 = fwd ? o.type2 : o.type1;
            r.pos1 = o.pos1;
            r.pos2 = o.pos2;
            r.perc = o.perc;
            r.len1 = o.len1;
            r.len2 = o.len2;
            r.ord2 = o.ord2;
        }
        if (line && !(r.ord2 == '-' || r.ord2 == '1' || r.ord2 == '2')) bad |= (unsigned)kFnoStatusRequire;
        if (line && no_inclusions && r.perc == 100) line = false;
        if (line && (r.id1 >= 10000000000ull || r.id2 >= 10000000000ull || r.perc < 0 || r.perc > 999)) bad |= (unsigned)kFnoStatusRange;
        if (bad) line = false;
        r.valid = line ? 1 : 0;
        rec[i] = r;
        iota[i] = (uint32_t)i;
        if (line) {
            atomicAdd(&tally[it.kind & 3], 1u);
            const uint64_t kid1 = dec_key_unsigned<10>(r.id1), kid2 = dec_key_unsigned<10>(r.id2);  // 35 bits each
            const uint64_t kp1 = dec_key_signed(r.pos1), kp2 = dec_key_signed(r.pos2);             // 40 bits each
            const uint64_t kl1 = dec_key_signed(r.len1), kl2 = dec_key_signed(r.len2);
            const uint64_t kpc = dec_key_unsigned<3>((uint64_t)r.perc);                            // 11 bits
            const uint64_t ord = r.ord2 == '-' ? 0 : (r.ord2 == '1' ? 1 : 2);                       // '-' < '1' < '2'
            const uint64_t o1 = r.ori1 == '+' ? 0 : 1, o2 = r.ori2 == '+' ? 0 : 1;                  // '+' < '-'
            const uint64_t t1 = r.type1 == 'p' ? 0 : 1, t2 = r.type2 == 'p' ? 0 : 1;                // 'p' < 's'
            k3[i] = kid1 << 29 | kid2 >> 6;
            k2[i] = (kid2 & 63) << 58 | kp1 << 18 | kp2 >> 22;
            k1[i] = (kp2 & ((1ull << 22) - 1)) << 42 | ord << 40 | o1 << 39 | o2 << 38 | kpc << 27 | kl1 >> 13;
            k0[i] = (kl1 & 8191) << 51 | kl2 << 11 | t1 << 10 | t2 << 9;
        } else {  // behind every line
            k3[i] = ~0ull;
            k2[i] = ~0ull;
            k1[i] = ~0ull;
            k0[i] = ~0ull;
            if (bad) atomicOr(&status, bad);
        }
    }
    __syncthreads();
    if (threadIdx.x < 4 && tally[threadIdx.x]) atomicAdd(&counters[threadIdx.x], (unsigned long long)tally[threadIdx.x]);
    if (threadIdx.x == 4 && status) atomicOr(&counters[4], (unsigned long long)status);
}

__global__ __launch_bounds__(kBlock) void fno_gather_kernel(const uint64_t* __restrict__ key, const uint32_t* __restrict__ perm, uint64_t n,
                                                            uint64_t* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) out[i] = key[perm[i]];
}

__device__ __forceinline__ bool same_line(const FnoRec& a, const FnoRec& b) {
    return a.id1 == b.id1 && a.id2 == b.id2 && a.pos1 == b.pos1 && a.pos2 == b.pos2 && a.perc == b.perc && a.len1 == b.len1 && a.len2 == b.len2 &&
           a.ord2 == b.ord2 && a.ori1 == b.ori1 && a.ori2 == b.ori2 && a.type1 == b.type1 && a.type2 == b.type2;
}

__global__ __launch_bounds__(kBlock) void fno_mark_kernel(const FnoRec* __restrict__ rec, const uint32_t* __restrict__ perm, uint64_t n,
                                                          uint64_t* __restrict__ len, unsigned long long* __restrict__ counters) {
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    __shared__ unsigned int lines;
    if (threadIdx.x == 0) lines = 0;
    __syncthreads();
    if (i <= n) {
        uint64_t bytes = 0;
        if (i < n) {
            const FnoRec r = rec[perm[i]];
            if (r.valid && (i == 0 || !same_line(r, rec[perm[i - 1]]))) {
                // 12 tabs + newline + ord2, ori1, ori2, "0", type1, type2
                bytes = 19 + digits_u64(r.id1) + digits_u64(r.id2) + chars_i32(r.pos1) + chars_i32(r.pos2) + chars_i32(r.perc) + chars_i32(r.len1) +
                        chars_i32(r.len2);
                atomicAdd(&lines, 1u);
            }
        }
        len[i] = bytes;
    }
    __syncthreads();
    if (threadIdx.x == 0 && lines) atomicAdd(&counters[5], (unsigned long long)lines);
}

__global__ __launch_bounds__(kBlock) void fno_format_kernel(const FnoRec* __restrict__ rec, const uint32_t* __restrict__ perm,
                                                            const uint64_t* __restrict__ len, const uint64_t* __restrict__ off, uint64_t n,
                                                            char* __restrict__ text) {
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n || len[i] == 0) return;
    const FnoRec r = rec[perm[i]];
    char line[96];  // 2 x 20 + 5 x 11 would be 95 + 19: ids are below 10^10 here, so 2 x 10 + 5 x 11 + 19 = 94
    char* p = line;
    p = put_u64(p, r.id1);
    *p++ = '\t';
    p = put_u64(p, r.id2);
    *p++ = '\t';
    p = put_i32(p, r.pos1);
    *p++ = '\t';
    p = put_i32(p, r.pos2);
    *p++ = '\t';
    *p++ = (char)r.ord2;
    *p++ = '\t';
    *p++ = (char)r.ori1;
    *p++ = '\t';
    *p++ = (char)r.ori2;
    *p++ = '\t';
    p = put_i32(p, r.perc);
    *p++ = '\t';
    *p++ = '0';
    *p++ = '\t';
    p = put_i32(p, r.len1);
    *p++ = '\t';
    p = put_i32(p, r.len2);
    *p++ = '\t';
    *p++ = (char)r.type1;
    *p++ = '\t';
    *p++ = (char)r.type2;
    *p++ = '\n';
    char* dst = text + off[i];
    const int m = (int)(p - line);
    for (int k = 0; k < m; k++) dst[k] = line[k];
}

// ---- FNO=3 ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int ratio100(int a, int b) { return (int)floorf((float)a / (float)b * 100.0f); }  // FindNextOverlaps3.cpp:259

// deduceOverlap, src/FindNextOverlaps3.cpp:180-406, and the two tests of :149-157 — statement for statement the host form
// (host/FindNextOverlaps.cpp, Fno3::deduce).  0 = a line, 1 = no line, 2 = the reference would stop.
__device__ int deduce3(const FnoItem& it, uint32_t no_inclusions, Fno3Rec& o) {
    const int a_l = it.v[0], a_r = it.v[1], b_l = it.v[2], b_r = it.v[3];
    const int A1 = it.v[4], A2 = it.v[5], B1 = it.v[6], B2 = it.v[7];
    const bool ap = it.a_paired != 0, bp = it.b_paired != 0;
    const bool a_first = a_l - b_l >= 0;
    o.id1 = a_first ? it.ida : it.idb;
    o.id2 = a_first ? it.idb : it.ida;
    int pos1 = a_first ? a_l - b_l : b_l - a_l, pos2 = 0, len1, len2 = 0;
    int perc1, perc2 = 0;
    uint8_t ord = '-', t1, t2;
    if (!ap && !bp) {  // :204-237
        if (A1 <= 0 || B1 <= 0) return 2;  // a division by zero in the reference
        if (pos1 > (a_first ? A1 : B1)) return 1;
        len1 = a_first ? imin(A1 - pos1, B1) : imin(A1, B1 - pos1);
        perc1 = perc_max(len1, A1, B1);
        t1 = t2 = 's';
    } else if (ap != bp) {  // :238-281 (P-S), :282-324 (S-P)
        const int P1 = ap ? A1 : B1, P2 = ap ? A2 : B2, S = ap ? B1 : A1;
        const bool pair_first = ap == a_first;
        len1 = pair_first ? P1 - pos1 : imin(P1, S - pos1);
        if (len1 <= 0) return 1;
        t1 = pair_first ? 'p' : 's';
        t2 = pair_first ? 's' : 'p';
        if (P1 <= 0) return 2;
        perc1 = ratio100(len1, P1);
        pos2 = ap ? b_r - a_r : a_r - b_r;
        len2 = imin(P2, S - pos2);
        if (len2 <= 0 || pos2 < 0) return 1;
        if (P2 <= 0) return 2;
        perc2 = ratio100(len2, P2);
    } else {  // :325-399
        len1 = a_first ? imin(A1 - pos1, B1) : imin(A1, B1 - pos1);
        const bool back = a_r - b_r >= 0;
        pos2 = back ? a_r - b_r : b_r - a_r;
        len2 = back ? imin(A2 - pos2, B2) : imin(A2, B2 - pos2);
        if (len1 <= 0 || len2 <= 0) return 1;
        if (A1 <= 0 || B1 <= 0 || A2 <= 0 || B2 <= 0) return 2;
        perc1 = perc_max(len1, A1, B1);
        perc2 = perc_max(len2, A2, B2);
        if (!((unsigned)perc1 <= 100u && (unsigned)perc2 <= 100u)) return 2;
        ord = a_first == back ? '1' : '2';
        t1 = t2 = 'p';
    }
    if (perc1 < 0 || perc1 > 100 || perc2 < 0 || perc2 > 100) return 2;  // the Overlap constructor's checks, src/Overlap.h:88-102
    if (len1 < 0 || len2 < 0) return 2;
    const unsigned perc = perc2 > 0 ? (unsigned)(0.5 * (double)(perc1 + perc2)) : (unsigned)perc1;  // Overlap::get_perc
    if (no_inclusions && perc == 100u) return 1;
    if (len1 <= 0) return 1;
    o.pos1 = pos1;
    o.pos2 = pos2;
    o.perc1 = perc1;
    o.perc2 = perc2;
    o.len1 = len1;
    o.len2 = len2;
    o.ord = ord;
    o.type1 = t1;
    o.type2 = t2;
    return 0;
}

__global__ __launch_bounds__(kBlock) void fno3_deduce_kernel(const FnoItem* __restrict__ items, uint64_t n, uint32_t no_inclusions, Fno3Rec* __restrict__ rec,
                                                             uint64_t* __restrict__ len, unsigned long long* __restrict__ counters) {
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    __shared__ unsigned int lines, status;
    if (threadIdx.x == 0) {
        lines = 0;
        status = 0;
    }
    __syncthreads();
    if (i <= n) {
        uint64_t bytes = 0;
        if (i < n) {
            Fno3Rec r;
            const FnoItem it = items[i];
            const int what = it.ida >= 10000000000ull || it.idb >= 10000000000ull ? 3 : deduce3(it, no_inclusions, r);
            if (what == 0) {
                // 12 tabs + newline + ord, "+", "+", type1, type2
                bytes = 18 + digits_u64(r.id1) + digits_u64(r.id2) + chars_i32(r.pos1) + chars_i32(r.pos2) + chars_i32(r.perc1) + chars_i32(r.perc2) +
                        chars_i32(r.len1) + chars_i32(r.len2);
                rec[i] = r;
                atomicAdd(&lines, 1u);
            } else if (what >= 2) {
                atomicOr(&status, what == 2 ? (unsigned)kFnoStatusRequire : (unsigned)kFnoStatusRange);
            }
        }
        len[i] = bytes;
    }
    __syncthreads();
    if (threadIdx.x == 0 && lines) atomicAdd(&counters[5], (unsigned long long)lines);
    if (threadIdx.x == 0 && status) atomicOr(&counters[4], (unsigned long long)status);
}

__global__ __launch_bounds__(kBlock) void fno3_format_kernel(const Fno3Rec* __restrict__ rec, const uint64_t* __restrict__ len, const uint64_t* __restrict__ off,
                                                             uint64_t n, char* __restrict__ text) {
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n || len[i] == 0) return;
    const Fno3Rec r = rec[i];
    char line[104];  // 2 x 10 + 6 x 11 + 18
    char* p = line;
    p = put_u64(p, r.id1);
    *p++ = '\t';
    p = put_u64(p, r.id2);
    *p++ = '\t';
    p = put_i32(p, r.pos1);
    *p++ = '\t';
    p = put_i32(p, r.pos2);
    *p++ = '\t';
    *p++ = (char)r.ord;
    *p++ = '\t';
    *p++ = '+';
    *p++ = '\t';
    *p++ = '+';
    *p++ = '\t';
    p = put_i32(p, r.perc1);
    *p++ = '\t';
    p = put_i32(p, r.perc2);
    *p++ = '\t';
    p = put_i32(p, r.len1);
    *p++ = '\t';
    p = put_i32(p, r.len2);
    *p++ = '\t';
    *p++ = (char)r.type1;
    *p++ = '\t';
    *p++ = (char)r.type2;
    *p++ = '\n';
    char* dst = text + off[i];
    const int m = (int)(p - line);
    for (int k = 0; k < m; k++) dst[k] = line[k];
}

// ---- the walk of FNO=1 (src/FindNextOverlaps.cpp:25-349, 612-630, 891-935) -----------------------------------------------------
// updateOverlap's case analysis per edge in walk order: an edge between two unmerged vertices is copied (one "direct" item); an
// edge with a merged end stands for one combination per super-read holding that end (both ends merged: per pair of super-reads),
// and of the combinations of one unordered pair of new ids the reference keeps the FIRST it meets (overlaps_found).  Here:
// count per edge, scan, one lane per combination writes the pair as a sort key — combinations are numbered in walk order, so a
// STABLE sort by pair puts the first met in front of its run — the heads of the runs are the kept ones.
__device__ __forceinline__ uint64_t n2s_count(const FnoWalkInput& w, uint64_t v) { return w.n2s_off[v + 1] - w.n2s_off[v]; }

__global__ __launch_bounds__(kBlock) void fno_walk_count_kernel(FnoWalkInput w, uint64_t* __restrict__ cnt_comb, uint64_t* __restrict__ cnt_direct,
                                                                unsigned long long* __restrict__ counters) {
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i > w.n_edges) return;
    uint64_t comb = 0, direct = 0;
    if (i < w.n_edges) {
        const uint64_t u = w.edges[i].v1, v = w.edges[i].v2;
        if (u >= w.n_nodes || v >= w.n_nodes) {
            atomicOr(&counters[4], (unsigned long long)kFnoStatusRequire);  // "edge vertex out of range"
        } else {
            const bool vu = w.nodes[u].visited != 0, vv = w.nodes[v].visited != 0;
            if (!vu && !vv) direct = 1;
            else if (!vu) comb = n2s_count(w, v);
            else if (!vv) comb = n2s_count(w, u);
            else {
                comb = n2s_count(w, u) * n2s_count(w, v);
                if (comb >= 0xFFFFFFFFull) {  // the host path's "too many super-reads share one vertex"
                    atomicOr(&counters[4], (unsigned long long)kFnoStatusRange);
                    comb = 0;
                }
            }
        }
    }
    cnt_comb[i] = comb;  // [n_edges] = 0: the scans' totals
    cnt_direct[i] = direct;
}

// the edge a combination belongs to: the last i with off[i] <= c (off is the exclusive scan of the counts, off[n] = total)
__device__ __forceinline__ uint64_t edge_of(const uint64_t* __restrict__ off, uint64_t n, uint64_t c) {
    uint64_t lo = 0, hi = n;  // off[lo] <= c < off[hi]
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) >> 1;
        if (off[mid] <= c) lo = mid;
        else hi = mid;
    }
    return lo;
}

// ---- what the walk needs first, built where the walk runs ---------------------------------------------------------------------
// OverlapGraph::adj_out as offsets into graph_edges, which the caller hands over vertex by vertex (sorted by v1): lane i writes the
// offsets of the vertices in (v1[i - 1], v1[i]]; lane G those up to n_nodes.  Unsorted edges are the host form's business.
__global__ __launch_bounds__(kBlock) void fno_adj_offsets_kernel(const hc_fno_edge* __restrict__ ge, uint64_t G, uint64_t n_nodes, uint64_t* __restrict__ off,
                                                                 unsigned long long* __restrict__ counters) {
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i > G) return;
    if (i < G && (ge[i].v1 >= n_nodes || ge[i].v2 >= n_nodes)) {
        atomicOr(&counters[4], (unsigned long long)kFnoStatusRequire);  // "edge vertex out of range"
        return;
    }
    if (i && i < G && ge[i - 1].v1 > ge[i].v1) {
        atomicOr(&counters[4], (unsigned long long)kFnoStatusUnsorted);
        return;
    }
    const uint64_t from = i ? ge[i - 1].v1 + 1 : 0, to = i < G ? ge[i].v1 : n_nodes;
    if (i && ge[i - 1].v1 >= n_nodes) return;  // reported by lane i - 1
    for (uint64_t v = from; v <= to; ++v) off[v] = i;
}

// OverlapGraph::checkEdge(v, w, true), src/OverlapGraph.cpp:233-259: the score of the first edge v -> w, else of the first w -> v, else -1
__device__ __forceinline__ double check_edge(const hc_fno_edge* __restrict__ ge, const uint64_t* __restrict__ off, uint64_t v, uint64_t w) {
    for (uint64_t k = off[v]; k < off[v + 1]; ++k)
        if (ge[k].v2 == w) return ge[k].score;
    for (uint64_t k = off[w]; k < off[w + 1]; ++k)
        if (ge[k].v2 == v) return ge[k].score;
    return -1.0;
}

// the stored non-edges updateOverlap is called on (:635-813): those no edge of the graph stands for already (:702)
__global__ __launch_bounds__(kBlock) void fno_nonedge_filter_kernel(const hc_fno_edge* __restrict__ nonedges, uint64_t n, const hc_fno_edge* __restrict__ ge,
                                                                    const uint64_t* __restrict__ off, uint64_t n_nodes, uint8_t* __restrict__ keep,
                                                                    unsigned long long* __restrict__ counters) {
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const hc_fno_edge e = nonedges[i];
    uint8_t k = 0;
    if (e.score != 0) atomicOr(&counters[4], (unsigned long long)kFnoStatusRange);  // not a stored non-edge: the host form says so
    else if (!(e.len1 > 0 && e.len2 >= 0) || e.v1 >= n_nodes || e.v2 >= n_nodes) atomicOr(&counters[4], (unsigned long long)kFnoStatusRequire);
    else k = check_edge(ge, off, e.v1, e.v2) > 0 ? 0 : 1;
    keep[i] = k;
}
__global__ __launch_bounds__(kBlock) void fno_gather_edges_kernel(const hc_fno_edge* __restrict__ in, const uint32_t* __restrict__ idx, uint64_t n,
                                                                  hc_fno_edge* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) out[i] = in[idx[i]];
}
__global__ __launch_bounds__(kBlock) void fno_gather_mirrored_kernel(const hc_fno_edge* __restrict__ in, const uint32_t* __restrict__ idx, uint64_t n,
                                                                     const hc_fno_read* __restrict__ nodes, uint64_t half, hc_fno_edge* __restrict__ out,
                                                                     unsigned long long* __restrict__ counters) {
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const hc_fno_edge e = in[idx[i]];  // (its vertices are below 2 * half: the filter looked them up)
    hc_fno_edge o;
    if (fno_mirror_nonedge(e, nodes[e.v1], nodes[e.v2], half, &o) != 0) {
        atomicOr(&counters[4], (unsigned long long)kFnoStatusRequire);
        o = e;
    }
    out[2 * i] = e;
    out[2 * i + 1] = o;
}

// nodes_to_SR (:893-906): (vertex, super-read) for every member of every clique, in super-read order — a stable sort by vertex
// leaves every vertex's super-reads in the order the reference pushes them — then the offsets as for adj_out
__global__ __launch_bounds__(kBlock) void fno_clique_pairs_kernel(const uint64_t* __restrict__ clique_nodes, const uint64_t* __restrict__ clique_off,
                                                                  uint64_t n_srs, uint64_t total, uint64_t n_nodes, uint64_t* __restrict__ key,
                                                                  uint32_t* __restrict__ sr, unsigned long long* __restrict__ counters) {
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= total) return;
    const uint64_t node = clique_nodes[i];
    if (node >= n_nodes) atomicOr(&counters[4], (unsigned long long)kFnoStatusRequire);  // nodes_to_SR.at(node)
    key[i] = node;
    sr[i] = (uint32_t)edge_of(clique_off, n_srs, i);  // the super-read entry i belongs to
}
__global__ __launch_bounds__(kBlock) void fno_offsets_kernel(const uint64_t* __restrict__ sorted, uint64_t n, uint64_t n_nodes, uint64_t* __restrict__ off) {
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i > n) return;
    const uint64_t from = i ? sorted[i - 1] + 1 : 0, to = i < n ? sorted[i] : n_nodes;
    if ((i && sorted[i - 1] >= n_nodes) || (i < n && sorted[i] >= n_nodes)) return;  // reported by fno_clique_pairs_kernel
    for (uint64_t v = from; v <= to; ++v) off[v] = i;
}

struct Combination {
    uint64_t edge;
    uint32_t sr1, sr2;
    uint64_t id1, id2;
    uint8_t kind;  // 1 u2sr, 2 v2sr, 3 sr2sr
};
__device__ __forceinline__ Combination combination(const FnoWalkInput& w, const uint64_t* __restrict__ off_comb, uint64_t c) {
    Combination k;
    k.edge = edge_of(off_comb, w.n_edges, c);
    const uint64_t nth = c - off_comb[k.edge];
    const uint64_t u = w.edges[k.edge].v1, v = w.edges[k.edge].v2;
    const bool vu = w.nodes[u].visited != 0;
    const bool vv = w.nodes[v].visited != 0;
    k.sr1 = k.sr2 = 0;
    if (!vu) {  // :69-150
        k.kind = 1;
        k.sr2 = w.n2s[w.n2s_off[v] + nth];
        k.id1 = w.nodes[u].id;
        k.id2 = w.srs[k.sr2].id;
    } else if (!vv) {  // :151-230
        k.kind = 2;
        k.sr1 = w.n2s[w.n2s_off[u] + nth];
        k.id1 = w.nodes[v].id;
        k.id2 = w.srs[k.sr1].id;
    } else {  // :231-349
        k.kind = 3;
        const uint64_t n2 = n2s_count(w, v);
        k.sr1 = w.n2s[w.n2s_off[u] + nth / n2];
        k.sr2 = w.n2s[w.n2s_off[v] + nth % n2];
        k.id1 = w.srs[k.sr1].id;
        k.id2 = w.srs[k.sr2].id;
    }
    return k;
}

__global__ __launch_bounds__(kBlock) void fno_walk_expand_kernel(FnoWalkInput w, const uint64_t* __restrict__ off_comb, uint64_t n_comb,
                                                                 uint64_t* __restrict__ key, uint32_t* __restrict__ iota,
                                                                 unsigned long long* __restrict__ counters) {
    const uint64_t c = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (c >= n_comb) return;
    const Combination k = combination(w, off_comb, c);
    uint64_t out = ~0ull;  // sr2sr of one super-read with itself: skipped (:246), sorts behind every pair
    if (k.id1 == k.id2) {
        if (k.kind != 3) atomicOr(&counters[4], (unsigned long long)kFnoStatusRequire);  // assert(id1 != id2)
    } else {
        const uint64_t lo = k.id1 < k.id2 ? k.id1 : k.id2, hi = k.id1 < k.id2 ? k.id2 : k.id1;
        if (hi >= w.new_read_count) atomicOr(&counters[4], (unsigned long long)kFnoStatusRequire);  // overlaps_found.at(id)
        else out = lo << w.id_bits | hi;
    }
    key[c] = out;
    iota[c] = (uint32_t)c;
}

__global__ __launch_bounds__(kBlock) void fno_walk_heads_kernel(const uint64_t* __restrict__ key_sorted, uint64_t n_comb, uint8_t* __restrict__ flag) {
    const uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= n_comb) return;
    const uint64_t k = key_sorted[j];
    flag[j] = (k != ~0ull && (j == 0 || key_sorted[j - 1] != k)) ? 1 : 0;
}

__global__ __launch_bounds__(kBlock) void fno_walk_direct_kernel(const uint64_t* __restrict__ off_direct, uint64_t n_edges, uint32_t* __restrict__ direct_edge) {
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_edges) return;
    if (off_direct[i + 1] != off_direct[i]) direct_edge[off_direct[i]] = (uint32_t)i;
}

// the pair of findCliqueIndex calls of :95-109 etc.: start offsets of `node` inside super-read `sr` (its subreads sorted by node)
__device__ __forceinline__ bool clique_indices(const FnoWalkInput& w, uint64_t node, uint32_t sr, bool read_paired, int& left, int& right) {
    uint64_t lo = w.subread_off[sr], hi = w.subread_off[sr + 1];
    if (lo == hi) return false;  // assert(!subreadMap.empty())
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) >> 1;
        if (w.subreads[mid].node <= node) lo = mid;
        else hi = mid;
    }
    const hc_fno_subread s = w.subreads[lo];
    if (s.node != node) return false;  // subreadMap.at(node)
    if (!(s.index1 >= 0 && s.startpos1 >= 0) || (s.index1 > 0 && s.startpos1 > 0)) return false;
    left = s.index1 - s.startpos1;
    const bool sr_paired = w.srs[sr].paired != 0;
    if (!sr_paired && !read_paired) {
        right = left;
        return true;
    }
    if (!(s.index2 >= 0 && s.startpos2 >= 0)) return false;
    if (sr_paired && s.index2 > 0 && s.startpos2 > 0) return false;
    right = s.index2 - s.startpos2;
    return true;
}

// One item per kept combination, everything computeOverlapData needs looked up (the host form's build_item):
// items [0, n_direct) are the copied edges, the others the heads of the sorted runs.
__global__ __launch_bounds__(kBlock) void fno_walk_items_kernel(FnoWalkInput w, const uint64_t* __restrict__ off_comb, const uint32_t* __restrict__ direct_edge,
                                                                uint64_t n_direct, const uint32_t* __restrict__ val_sorted,
                                                                const uint32_t* __restrict__ head_pos, uint64_t n_heads, FnoItem* __restrict__ items,
                                                                unsigned long long* __restrict__ counters) {
    const uint64_t t = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= n_direct + n_heads) return;
    FnoItem o;
    for (int k = 0; k < 10; ++k) o.v[k] = 0;
    o.pad[0] = o.pad[1] = 0;
    bool ok = true;
    Combination c;
    if (t < n_direct) {
        c.edge = direct_edge[t];
        c.kind = 0;
        c.sr1 = c.sr2 = 0;
    } else {
        c = combination(w, off_comb, val_sorted[head_pos[t - n_direct]]);
    }
    const hc_fno_edge e = w.edges[c.edge];
    const hc_fno_read n1 = w.nodes[e.v1], n2 = w.nodes[e.v2];
    o.ori1 = o.ori2 = '+';
    if (w.resolve_orientations && e.score == 0) {  // :35-38
        o.ori1 = ((e.ori1 != 0) == (n1.orientation != 0)) ? '+' : '-';
        o.ori2 = ((e.ori2 != 0) == (n2.orientation != 0)) ? '+' : '-';
    }
    o.kind = c.kind;
    o.e_ord = e.ord;
    if (c.kind == 0) {  // :44-68
        ok = e.perc >= 0;
        o.ida = n1.id;
        o.idb = n2.id;
        o.v[0] = e.pos1; o.v[1] = e.pos2; o.v[2] = e.perc; o.v[3] = e.len1; o.v[4] = e.len2;
        o.a_paired = n1.paired != 0;
        o.b_paired = n2.paired != 0;
    } else {
        int i1l = 0, i1r = 0, i2l = 0, i2r = 0;
        hc_fno_read a, b;
        if (c.kind == 1) {
            a = n1;
        } else {
            a = w.srs[c.sr1];
            ok = clique_indices(w, e.v1, c.sr1, n1.paired != 0, i1l, i1r) && ok;
        }
        if (c.kind == 2) {
            b = n2;
        } else {
            b = w.srs[c.sr2];
            ok = clique_indices(w, e.v2, c.sr2, n2.paired != 0, i2l, i2r) && ok;
        }
        o.ida = a.id;
        o.idb = b.id;
        o.v[0] = e.pos1; o.v[1] = e.pos2;
        o.v[2] = i1l; o.v[3] = i1r; o.v[4] = i2l; o.v[5] = i2r;
        o.v[6] = (int)a.len1; o.v[7] = (int)a.len2; o.v[8] = (int)b.len1; o.v[9] = (int)b.len2;
        o.a_paired = a.paired != 0;
        o.b_paired = b.paired != 0;
    }
    if (!ok) atomicOr(&counters[4], (unsigned long long)kFnoStatusRequire);
    items[t] = o;
}

inline dim3 grid_for(uint64_t n) { return dim3((unsigned)((n + kBlock - 1) / kBlock)); }

}  // namespace

hipError_t fno_deduce(const FnoItem* items, uint64_t n, uint32_t no_inclusions, FnoRec* rec, uint64_t* k0, uint64_t* k1, uint64_t* k2,
                      uint64_t* k3, uint32_t* iota, unsigned long long* counters, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(fno_deduce_kernel, grid_for(n), dim3(kBlock), 0, s, items, n, no_inclusions, rec, k0, k1, k2, k3, iota, counters);
    return hipGetLastError();
}
hipError_t fno_gather_keys(const uint64_t* key, const uint32_t* perm, uint64_t n, uint64_t* out, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(fno_gather_kernel, grid_for(n), dim3(kBlock), 0, s, key, perm, n, out);
    return hipGetLastError();
}
hipError_t fno_mark_lines(const FnoRec* rec, const uint32_t* perm, uint64_t n, uint64_t* len, unsigned long long* counters, hipStream_t s) {
    hipLaunchKernelGGL(fno_mark_kernel, grid_for(n + 1), dim3(kBlock), 0, s, rec, perm, n, len, counters);
    return hipGetLastError();
}
hipError_t fno_format(const FnoRec* rec, const uint32_t* perm, const uint64_t* len, const uint64_t* off, uint64_t n, char* text, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(fno_format_kernel, grid_for(n), dim3(kBlock), 0, s, rec, perm, len, off, n, text);
    return hipGetLastError();
}

hipError_t fno_adj_offsets(const hc_fno_edge* ge, uint64_t G, uint64_t n_nodes, uint64_t* off, unsigned long long* counters, hipStream_t s) {
    hipLaunchKernelGGL(fno_adj_offsets_kernel, grid_for(G + 1), dim3(kBlock), 0, s, ge, G, n_nodes, off, counters);
    return hipGetLastError();
}
hipError_t fno_nonedge_filter(const hc_fno_edge* nonedges, uint64_t n, const hc_fno_edge* ge, const uint64_t* off, uint64_t n_nodes, uint8_t* keep,
                              unsigned long long* counters, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(fno_nonedge_filter_kernel, grid_for(n), dim3(kBlock), 0, s, nonedges, n, ge, off, n_nodes, keep, counters);
    return hipGetLastError();
}
hipError_t fno_gather_edges(const hc_fno_edge* in, const uint32_t* idx, uint64_t n, hc_fno_edge* out, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(fno_gather_edges_kernel, grid_for(n), dim3(kBlock), 0, s, in, idx, n, out);
    return hipGetLastError();
}
hipError_t fno_gather_mirrored(const hc_fno_edge* in, const uint32_t* idx, uint64_t n, const hc_fno_read* nodes, uint64_t half, hc_fno_edge* out,
                               unsigned long long* counters, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(fno_gather_mirrored_kernel, grid_for(n), dim3(kBlock), 0, s, in, idx, n, nodes, half, out, counters);
    return hipGetLastError();
}
hipError_t fno_clique_pairs(const uint64_t* clique_nodes, const uint64_t* clique_off, uint64_t n_srs, uint64_t total, uint64_t n_nodes, uint64_t* key,
                            uint32_t* sr, unsigned long long* counters, hipStream_t s) {
    if (!total) return hipSuccess;
    hipLaunchKernelGGL(fno_clique_pairs_kernel, grid_for(total), dim3(kBlock), 0, s, clique_nodes, clique_off, n_srs, total, n_nodes, key, sr, counters);
    return hipGetLastError();
}
hipError_t fno_offsets(const uint64_t* sorted, uint64_t n, uint64_t n_nodes, uint64_t* off, hipStream_t s) {
    hipLaunchKernelGGL(fno_offsets_kernel, grid_for(n + 1), dim3(kBlock), 0, s, sorted, n, n_nodes, off);
    return hipGetLastError();
}
hipError_t fno_walk_count(const FnoWalkInput& w, uint64_t* cnt_comb, uint64_t* cnt_direct, unsigned long long* counters, hipStream_t s) {
    hipLaunchKernelGGL(fno_walk_count_kernel, grid_for(w.n_edges + 1), dim3(kBlock), 0, s, w, cnt_comb, cnt_direct, counters);
    return hipGetLastError();
}
hipError_t fno_walk_expand(const FnoWalkInput& w, const uint64_t* off_comb, uint64_t n_comb, uint64_t* key, uint32_t* iota, unsigned long long* counters,
                           hipStream_t s) {
    if (!n_comb) return hipSuccess;
    hipLaunchKernelGGL(fno_walk_expand_kernel, grid_for(n_comb), dim3(kBlock), 0, s, w, off_comb, n_comb, key, iota, counters);
    return hipGetLastError();
}
hipError_t fno_walk_heads(const uint64_t* key_sorted, uint64_t n_comb, uint8_t* flag, hipStream_t s) {
    if (!n_comb) return hipSuccess;
    hipLaunchKernelGGL(fno_walk_heads_kernel, grid_for(n_comb), dim3(kBlock), 0, s, key_sorted, n_comb, flag);
    return hipGetLastError();
}
hipError_t fno_walk_direct(const uint64_t* off_direct, uint64_t n_edges, uint32_t* direct_edge, hipStream_t s) {
    if (!n_edges) return hipSuccess;
    hipLaunchKernelGGL(fno_walk_direct_kernel, grid_for(n_edges), dim3(kBlock), 0, s, off_direct, n_edges, direct_edge);
    return hipGetLastError();
}
hipError_t fno_walk_items(const FnoWalkInput& w, const uint64_t* off_comb, const uint32_t* direct_edge, uint64_t n_direct, const uint32_t* val_sorted,
                          const uint32_t* head_pos, uint64_t n_heads, FnoItem* items, unsigned long long* counters, hipStream_t s) {
    if (!(n_direct + n_heads)) return hipSuccess;
    hipLaunchKernelGGL(fno_walk_items_kernel, grid_for(n_direct + n_heads), dim3(kBlock), 0, s, w, off_comb, direct_edge, n_direct, val_sorted, head_pos,
                       n_heads, items, counters);
    return hipGetLastError();
}

hipError_t fno3_deduce(const FnoItem* items, uint64_t n, uint32_t no_inclusions, Fno3Rec* rec, uint64_t* len, unsigned long long* counters, hipStream_t s) {
    hipLaunchKernelGGL(fno3_deduce_kernel, grid_for(n + 1), dim3(kBlock), 0, s, items, n, no_inclusions, rec, len, counters);
    return hipGetLastError();
}
hipError_t fno3_format(const Fno3Rec* rec, const uint64_t* len, const uint64_t* off, uint64_t n, char* text, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(fno3_format_kernel, grid_for(n), dim3(kBlock), 0, s, rec, len, off, n, text);
    return hipGetLastError();
}

}  // namespace hc
