// hc_kernels.hip — hand-written HIP kernels for gfx950 (MI355X, CDNA4, wave64).
//
// Hot op: the body of the `omp for` in EdgeCalculator::process_overlaps
// (reference src/EdgeCalculator.cpp:400-414): for one candidate overlap pick the
// oriented sequences (compute_overlap, :143-385), walk the overlapped positions
// (overlap_score, :67-139; score, :26-56) and take the 3-way admission decision.
//
// Bit-exactness contract (DESIGN.md "Numerics"):
//   * log p(Q1,Q2,match?) comes from a table built on the HOST with the host libm
//     by the reference's own expressions, so every term equals the reference's.
//   * terms are added in increasing position order into one fp64 accumulator per
//     candidate (one lane owns one candidate), exactly the order of :106-128;
//     an N position adds +0.0 (S + 0.0 == S); a `p < --mismatch` term is +inf (poison).
//   * x = (1.0/total_len) * total_score uses IEEE fp64 divide and multiply; the
//     file is compiled with -ffp-contract=off so nothing is fused.
//   * exp() is NOT taken on the device: thresholds are inverted into x-space on the
//     host through the host libm (guard band => HC_CLS_AMBIG, host decides).
// No MFMA: this is byte gathering + table lookup + a serial fp64 add chain.
//
// Mapping (DESIGN.md "Kernel"): one lane = one candidate.  Per 16 positions a lane
// loads 16 symbols of each read (16 B per load for 8-bit symbols), derives N / mismatch
// masks and counts with packed-byte integer ops (4 positions per VALU op), builds the 16
// LDS addresses of the log table with one v_perm_b32 + one shift each, issues the 16
// ds_read_b64 back to back, and only then runs the 16 dependent v_add_f64.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <hipcub/hipcub.hpp>

#include "../../include/hcedge.h"
#include "hc_device.h"

namespace hc {

// ---------------------------------------------------------------------------
// Store encoding: raw ASCII bases + quality bytes -> symbol slots (both orientations).
// One wave per sequence; lanes stride over positions (coalesced reads and writes).
template <typename SymT, bool WIDE>
__global__ __launch_bounds__(256) void encode_store_kernel(const uint8_t* __restrict__ bases,
                                                           const uint8_t* __restrict__ quals,
                                                           const uint64_t* __restrict__ raw_off,  // [n_seq+1]
                                                           const uint64_t* __restrict__ seq_off,  // [n_seq] symbols
                                                           const uint8_t* __restrict__ qmap,  // [256] byte -> qidx, 255 = invalid
                                                           uint32_t n_seq, uint32_t K, SymT* __restrict__ sym,
                                                           uint8_t* __restrict__ seq_bad) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    const SymT nsym = WIDE ? (SymT)(kWideN << 2) : (SymT)((K << 3) | kCodeN);
    for (uint32_t q = wave; q < n_seq; q += n_waves) {
        const uint64_t r0 = raw_off[q];
        const uint32_t len = (uint32_t)(raw_off[q + 1] - r0);
        const uint64_t f0 = seq_off[q];
        const uint64_t stride = slot_stride(len, sizeof(SymT));
        uint32_t bad = 0;
        for (uint32_t i = lane; i < (uint32_t)stride; i += 64) {
            if (i < len) {
                const uint8_t b = bases[r0 + i];
                const uint8_t qi = qmap[quals[r0 + i]];
                uint32_t code;
                switch (b) {
                    case 'A': code = 0; break;
                    case 'C': code = 1; break;
                    case 'G': code = 2; break;
                    case 'T': code = 3; break;
                    case 'N': code = kCodeN; break;
                    default: code = kCodeBadBase; bad = 1; break;
                }
                uint32_t qx = qi;
                if (WIDE) {
                    // sym = qidx << 2 | base2; N / invalid are reserved quality indices with base bits 0
                    uint32_t b2 = code < 4 ? code : 0u;
                    if (code == kCodeN) qx = kWideN;
                    if (code == kCodeBadBase) qx = kWideBadBase;
                    if (qi == 255) { qx = kWideBadQual; b2 = 0; }  // quality outside [33,127]: always fatal inside an overlap
                    const uint32_t rb2 = qx < kWideN ? 3u - b2 : 0u;
                    sym[f0 + i] = (SymT)((qx << 2) | b2);
                    sym[f0 + stride + (len - 1 - i)] = (SymT)((qx << 2) | rb2);
                } else {
                    if (code == kCodeN) qx = K;            // zero row of the log table
                    if (code == kCodeBadBase) qx = K + 1;  // NaN row
                    if (qi == 255) {                       // quality outside [33,127]: always fatal inside an overlap
                        code = kCodeBadQual;
                        qx = K + 1;
                    }
                    const uint32_t rcode = code < 4 ? 3 - code : code;
                    sym[f0 + i] = (SymT)((qx << 3) | code);
                    sym[f0 + stride + (len - 1 - i)] = (SymT)((qx << 3) | rcode);
                }
            } else {
                sym[f0 + i] = nsym;
                sym[f0 + stride + i] = nsym;
            }
        }
        const unsigned long long any_bad = __ballot(bad != 0);
        if (lane == 0) seq_bad[q] = any_bad ? 1 : 0;
    }
}

__global__ __launch_bounds__(256) void build_read_desc_kernel(const uint32_t* __restrict__ read_first_seq,
                                                              const uint64_t* __restrict__ seq_off,
                                                              const uint64_t* __restrict__ raw_off,
                                                              const uint8_t* __restrict__ seq_bad, uint32_t n_reads,
                                                              ReadDesc* __restrict__ out) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const uint32_t f = read_first_seq[r];
    const bool paired = (read_first_seq[r + 1] - f) == 2;
    ReadDesc d;
    d.off1 = seq_off[f];
    d.len1 = (uint32_t)(raw_off[f + 1] - raw_off[f]);
    d.flags = (paired ? kReadPaired : 0u) | (seq_bad[f] ? kReadBadBase1 : 0u);
    d.off2 = 0;
    d.len2 = 0;
    if (paired) {
        d.off2 = seq_off[f + 1];
        d.len2 = (uint32_t)(raw_off[f + 2] - raw_off[f + 1]);
        d.flags |= seq_bad[f + 1] ? kReadBadBase2 : 0u;
    }
    d.pad = 0;
    out[r] = d;
}

// ---------------------------------------------------------------------------
// Candidate -> sub-overlap descriptors (reference compute_overlap, :197-380).
struct View {
    uint64_t off;  // symbol offset of the oriented sequence
    uint32_t len;
    uint32_t fatal;  // reverse-complementing a sequence that holds an invalid base: build_rev_comp exits
};

__device__ __forceinline__ ReadDesc load_desc(const ReadDesc* p) {
    const uint4* q = (const uint4*)p;
    const uint4 a = q[0], b = q[1];
    ReadDesc d;
    d.off1 = ((uint64_t)a.y << 32) | a.x;
    d.off2 = ((uint64_t)a.w << 32) | a.z;
    d.len1 = b.x;
    d.len2 = b.y;
    d.flags = b.z;
    d.pad = 0;
    return d;
}

// mate: 0 = /1 (or the single sequence), 1 = /2
template <int SB>
__device__ __forceinline__ View make_view(const ReadDesc& d, uint32_t mate, uint32_t fwd) {
    View v;
    const uint64_t off = mate ? d.off2 : d.off1;
    v.len = mate ? d.len2 : d.len1;
    // slot_stride() with a compile-time symbol size (no 64-bit division)
    const uint32_t stride = SB == 1 ? (((v.len + 15u) & ~15u) + 32u) : ((((2u * v.len + 15u) & ~15u) + 32u) >> 1);
    v.off = off + (fwd ? 0u : stride);
    v.fatal = (!fwd && (d.flags & (mate ? kReadBadBase2 : kReadBadBase1))) ? 1u : 0u;
    return v;
}

struct Sub {
    uint64_t offA, offB;
    uint32_t lenA, lenB, pos, fatal;
};

__device__ __forceinline__ Sub make_sub(const View& A, const View& B, uint32_t pos) {
    Sub s;
    s.offA = A.off;
    s.offB = B.off;
    s.lenA = A.len;
    s.lenB = B.len;
    s.pos = pos;
    s.fatal = A.fatal | B.fatal;
    return s;
}

// Returns the number of sub-overlaps (1 or 2); 0 = malformed record.
template <int SB>
__device__ __forceinline__ int resolve(const StoreView& st, const hc_overlap_rec& r, Sub& s0, Sub& s1) {
    if (r.read1 >= st.n_reads || r.read2 >= st.n_reads || r.read1 == r.read2) return 0;
    const ReadDesc d1 = load_desc(st.reads + r.read1);
    const ReadDesc d2 = load_desc(st.reads + r.read2);
    const uint32_t p1 = d1.flags & kReadPaired, p2 = d2.flags & kReadPaired;
    const uint32_t o1 = r.ori1 ? 1u : 0u, o2 = r.ori2 ? 1u : 0u;
    // single: S(R,o); paired: F(R,o) = o ? /1 : rc(/2) ("front"), K(R,o) = o ? /2 : rc(/1) ("back")
    const View F1 = make_view<SB>(d1, p1 ? (o1 ? 0u : 1u) : 0u, o1);
    const View F2 = make_view<SB>(d2, p2 ? (o2 ? 0u : 1u) : 0u, o2);
    s0 = make_sub(F1, F2, r.pos1);  // every type: (front1, front2, pos1)
    if (!p1 && !p2) return 1;       // s-s :199-233
    const View K1 = make_view<SB>(d1, p1 ? (o1 ? 1u : 0u) : 0u, o1);
    const View K2 = make_view<SB>(d2, p2 ? (o2 ? 1u : 0u) : 0u, o2);
    if (!p1) {  // s-p :234-271: (S1, K2, pos2)
        s1 = make_sub(F1, K2, r.pos2);
    } else if (!p2) {  // p-s :272-309: (S2, K1, pos2)
        s1 = make_sub(F2, K1, r.pos2);
    } else {  // p-p :312-380
        if (r.ord == '1') s1 = make_sub(K1, K2, r.pos2);
        else if (r.ord == '2') s1 = make_sub(K2, K1, r.pos2);
        else return 0;
    }
    return 2;
}

__device__ __forceinline__ uint32_t sub_positions(const Sub& s, uint32_t min_read_len) {
    if (s.pos >= s.lenA) return 0;                                  // :76-79
    if (s.lenA < min_read_len || s.lenB < min_read_len) return 0;  // :82-84
    const uint32_t rem = s.lenA - s.pos;                            // :88
    return rem < s.lenB ? rem : s.lenB;
}

struct SubScore {
    double x;  // (1.0/total_len)*total_score, or -inf
    uint32_t mm, n;
    uint32_t err;
};

// ---------------------------------------------------------------------------
// Symbol-width traits.  A "word" is 32 bits = 4 (uint8) or 2 (uint16) symbols; a chunk is 16 symbols.
template <typename SymT>
struct Tr;
template <>
struct Tr<uint8_t> {
    static constexpr int kSymsPerWord = 4;
    static constexpr int kWords = 4;  // per 16-symbol chunk
    static constexpr uint32_t kLow1 = 0x01010101u;
    static constexpr uint32_t kQMask = 0xF8F8F8F8u;
    static constexpr int kSymBits = 8;
};
template <>
struct Tr<uint16_t> {
    static constexpr int kSymsPerWord = 2;
    static constexpr int kWords = 8;
    static constexpr uint32_t kLow1 = 0x00010001u;
    static constexpr uint32_t kQMask = 0xFFF8FFF8u;
    static constexpr int kSymBits = 16;
};

__device__ __forceinline__ double lds_f64(const char* lut, uint32_t byte_addr) {
    return *(const double*)(lut + byte_addr);
}

// Exact per-position re-scan (rare: only when the fast sum came out NaN, i.e. the window
// holds an invalid base / quality byte).  Mirrors the reference's order of checks:
// all quality bytes of the window first (:92-101), then position by position (:106-128).
template <typename SymT>
__device__ __noinline__ SubScore score_sub_slow(const SymT* __restrict__ a, const SymT* __restrict__ b, uint32_t L,
                                                const char* lut, uint32_t Kp) {
    const uint32_t lg = lut_lg(Kp - 2u);
    SubScore r;
    r.x = -__builtin_inf();
    r.mm = 1;
    r.n = 1;
    r.err = 0;
    const bool wide = sizeof(SymT) == 1 && lg == 6;
    // kind of a symbol: 0 = base, 1 = N, 2 = invalid quality, 3 = invalid base
    auto kind = [&](uint32_t sy) -> uint32_t {
        if (wide) {
            const uint32_t q = sy >> 2;
            return q == kWideN ? 1u : (q == kWideBadQual ? 2u : (q == kWideBadBase ? 3u : 0u));
        }
        const uint32_t c = sy & 7u;
        return c == kCodeN ? 1u : (c == kCodeBadQual ? 2u : (c == kCodeBadBase ? 3u : 0u));
    };
    for (uint32_t i = 0; i < L; ++i)
        if (kind(a[i]) == 2u || kind(b[i]) == 2u) {
            r.err = 1;
            return r;
        }
    double S = 0.0;
    uint32_t cn = 0, cm = 0;
    for (uint32_t i = 0; i < L; ++i) {
        const uint32_t sa = a[i], sb = b[i];
        const uint32_t ka = kind(sa), kb = kind(sb);
        if (ka == 3u || kb == 3u) {  // :29-30 assert
            r.err = 1;
            return r;
        }
        if (ka == 1u || kb == 1u) continue;  // N: :35-39, :122-124
        const uint32_t m = wide ? ((sa & 3u) != (sb & 3u)) : ((sa & 7u) != (sb & 7u));
        const uint32_t qa = wide ? sa >> 2 : sa >> 3, qb = wide ? sb >> 2 : sb >> 3;
        const uint32_t addr = sizeof(SymT) == 1 ? lut_addr_u8(lg, qa, qb, m) : lut_addr_u16(Kp, qa, qb, m);
        const double t = lds_f64(lut, addr);
        if (t == __builtin_inf()) return r;  // :125-127
        S += t;
        cn += 1;
        cm += m;
    }
    if (cn == 0) return r;
    r.x = (1.0 / (double)cn) * S;
    r.mm = cm;
    r.n = cn;
    return r;
}

// Tail masks: kMaskTab[r][j] keeps the first r symbols of a 16-symbol chunk, as four (uint8) or
// eight (uint16) 32-bit words.  Lives in LDS right behind the log table (one ds_read_b128 per
// chunk instead of ~35 VALU instructions of shift/compare/select).
template <typename SymT>
__device__ __forceinline__ void fill_mask_table(uint32_t* tab /* 17 * kWords */, uint32_t tid, uint32_t nthreads) {
    using T = Tr<SymT>;
    for (uint32_t i = tid; i < 17u * T::kWords; i += nthreads) {
        const int r = (int)(i / T::kWords), j = (int)(i % T::kWords);
        const int left = r - j * T::kSymsPerWord;
        tab[i] = left >= T::kSymsPerWord ? 0xFFFFFFFFu : (left <= 0 ? 0u : ((1u << (left * T::kSymBits)) - 1u));
    }
}

// The table reads of one half-chunk (8 positions): packed-byte N / mismatch masks and counters, then
// one LDS address per position.  aw/bw/keep point at the half's 32-bit words.
template <typename SymT, int LG>
__device__ __forceinline__ void half_chunk_terms(const uint32_t* wa, const uint32_t* wb, const uint32_t* keep,
                                                 uint32_t nsym_word, const char* lut, uint32_t Kp, double (&t)[8],
                                                 uint32_t& skipped, uint32_t& cm) {
    using T = Tr<SymT>;
#pragma unroll
    for (int jj = 0; jj < T::kWords / 2; ++jj) {
        // symbols at or beyond L become N: they add 0.0 and count as skipped
        const uint32_t aw = (wa[jj] & keep[jj]) | (nsym_word & ~keep[jj]);
        const uint32_t bw = wb[jj];
        const uint32_t e = aw ^ bw;
        if (sizeof(SymT) == 1 && LG == 6) {
            // wide 8-bit encoding: byte = qidx << 2 | base2; qidx >= 48 (both top bits set) is N / invalid.
            // address = m << 15 | qa << 9 | ((qb ^ qa) & 63) << 3
            const uint32_t nm = ((aw & (aw << 1)) | (bw & (bw << 1))) & 0x80808080u;
            const uint32_t mk = ((e << 7) | (e << 6)) & 0x80808080u & ~nm;  // base bits differ, neither is N
            skipped += __builtin_popcount(nm);
            cm += __builtin_popcount(mk);
            const uint32_t lo = (e << 1) & 0xF8F8F8F8u;
            const uint32_t hi = ((e >> 7) & 0x01010101u) | ((aw >> 1) & 0x7E7E7E7Eu) | mk;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                t[jj * 4 + k] = lds_f64(lut, __builtin_amdgcn_perm(hi, lo, 0x0C0C0000u | ((4u + k) << 8) | (uint32_t)k));
            continue;
        }
        const uint32_t x = aw | bw;
        const uint32_t nm = x & (T::kLow1 << 2);                             // code bit 2 on either side: N (or invalid)
        const uint32_t mk = ((e << 1) | (e << 2)) & (T::kLow1 << 2) & ~x;  // bases differ, neither is N
        skipped += __builtin_popcount(nm);
        cm += __builtin_popcount(mk);
        if (sizeof(SymT) == 1) {
            // Two bytes per position whose concatenation IS the table's byte address
            //   m * (8 << 2LG) + qa * (8 << LG) + (qb ^ qa) * 8        (hc_device.h)
            // symbol byte = qidx << 3 | code, mk holds the mismatch flag at bit 2 of each byte, e = aw ^ bw.
            uint32_t lo, hi;
            if (LG == 5) {         // bits 3-7: qb^qa | bits 8-12: qa, bit 13: m
                lo = e & 0xF8F8F8F8u;
                hi = ((aw >> 3) & 0x1F1F1F1Fu) | (mk << 3);
            } else if (LG == 4) {  // bits 3-6: qb^qa, bit 7: qa0 | bits 8-10: qa>>1, bit 11: m
                lo = (e & 0x78787878u) | ((aw << 4) & 0x80808080u);
                hi = ((aw >> 4) & 0x07070707u) | (mk << 1);
            } else {               // bits 3-5: qb^qa, bits 6-7: qa&3 | bit 8: qa>>2, bit 9: m
                lo = (e & 0x38383838u) | ((aw << 3) & 0xC0C0C0C0u);
                hi = ((aw >> 5) & 0x01010101u) | (mk >> 1);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                t[jj * 4 + k] = lds_f64(lut, __builtin_amdgcn_perm(hi, lo, 0x0C0C0000u | ((4u + k) << 8) | (uint32_t)k));
        } else {
            // two positions per packed 16-bit op: entry = m*Kp*Kp + qa*Kp + qb (< 2*97*97, fits 16 bits)
            typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
            const u16x2 qa2 = __builtin_bit_cast(u16x2, aw) >> (unsigned short)3;  // the code bits fall off each lane
            const u16x2 qb2 = __builtin_bit_cast(u16x2, bw) >> (unsigned short)3;
            const u16x2 m2 = __builtin_bit_cast(u16x2, mk >> 2);                    // 0 / 1 per lane
            const u16x2 kp2 = {(unsigned short)Kp, (unsigned short)Kp};
            const u16x2 pl2 = {(unsigned short)(Kp * Kp), (unsigned short)(Kp * Kp)};
            const uint32_t e2 = __builtin_bit_cast(uint32_t, (u16x2)(m2 * pl2 + (qa2 * kp2 + qb2)));
            t[jj * 2 + 0] = lds_f64(lut, (e2 << 3) & 0x7FFF8u);
            t[jj * 2 + 1] = lds_f64(lut, (e2 >> 13) & 0x7FFF8u);
        }
    }
}

template <typename SymT>
struct ChunkData {
    uint32_t a[Tr<SymT>::kWords], b[Tr<SymT>::kWords];
};

// overlap_score (:67-139) for NS sub-overlaps of one candidate, one lane, interleaved:
// NS independent fp64 accumulators (each summed strictly in position order) so that the
// dependent v_add_f64 chains of the sub-overlaps overlap, and the next chunk of every stream is
// loaded while the current one is scored.
template <typename SymT, int NS, bool PREFETCH, int LG>
__device__ __forceinline__ void score_subs(const SymT* __restrict__ sym, const Sub* s, const char* lut,
                                           const uint32_t* masktab, uint32_t Kp, uint32_t nsym_word,
                                           uint32_t min_read_len, SubScore* out) {
    using T = Tr<SymT>;
    uint32_t L[NS], nch[NS];
    const SymT* a[NS];
    const SymT* b[NS];
    double S[NS];
    uint32_t skipped[NS], cm[NS];
    uint32_t nmax = 0;
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        out[u].x = -__builtin_inf();
        out[u].mm = 1;
        out[u].n = 1;
        out[u].err = s[u].fatal;
        L[u] = sub_positions(s[u], min_read_len);
        nch[u] = (L[u] + 15u) >> 4;
        nmax = nch[u] > nmax ? nch[u] : nmax;
        // an exhausted / empty sub keeps loading its (valid) first chunk: masked to N, adds 0.0
        a[u] = sym + s[u].offA + (L[u] ? s[u].pos : 0u);
        b[u] = sym + s[u].offB;
        S[u] = 0.0;
        skipped[u] = 0;
        cm[u] = 0;
    }
    if (nmax == 0) return;
    // Loads are predicated, never clamped: a lane that has finished its sub-overlap issues no more
    // memory requests while the longest lane of the wave is still running (the vector-memory front
    // end is this kernel's bottleneck; its cost is per lane request).
    ChunkData<SymT> nxt[NS] = {};
    if (PREFETCH) {
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            if (nch[u]) {
                __builtin_memcpy(nxt[u].a, a[u], sizeof(nxt[u].a));  // unaligned (pos is arbitrary)
                __builtin_memcpy(nxt[u].b, b[u], sizeof(nxt[u].b));
            }
        }
    }
    for (uint32_t c = 0; c < nmax; ++c) {
        ChunkData<SymT> cur[NS] = {};
        uint32_t keep[NS][T::kWords];
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            if (PREFETCH) {
                cur[u] = nxt[u];
                if (c + 1u < nch[u]) {
                    __builtin_memcpy(nxt[u].a, a[u] + 16u * (c + 1u), sizeof(nxt[u].a));
                    __builtin_memcpy(nxt[u].b, b[u] + 16u * (c + 1u), sizeof(nxt[u].b));
                }
            } else if (c < nch[u]) {
                __builtin_memcpy(cur[u].a, a[u] + 16u * c, sizeof(cur[u].a));
                __builtin_memcpy(cur[u].b, b[u] + 16u * c, sizeof(cur[u].b));
            }
            const int rem = (int)L[u] - (int)(16u * c);
            const uint32_t r = rem >= 16 ? 16u : (rem <= 0 ? 0u : (uint32_t)rem);
            __builtin_memcpy(keep[u], masktab + r * T::kWords, sizeof(keep[u]));
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {  // two half-chunks of 8 positions: bounds the registers held by table reads
            double t[NS][8];
#pragma unroll
            for (int u = 0; u < NS; ++u)
                half_chunk_terms<SymT, LG>(cur[u].a + h * (T::kWords / 2), cur[u].b + h * (T::kWords / 2),
                                       keep[u] + h * (T::kWords / 2), nsym_word, lut, Kp, t[u], skipped[u], cm[u]);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
#pragma unroll
                for (int u = 0; u < NS; ++u) S[u] += t[u][k];  // :119, strictly in position order per sub-overlap
            }
        }
    }
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        if (L[u] == 0) continue;
        if (S[u] != S[u]) {  // an invalid symbol inside the window
            const SubScore e = score_sub_slow<SymT>(a[u], b[u], L[u], lut, Kp);
            out[u].x = e.x;
            out[u].mm = e.mm;
            out[u].n = e.n;
            out[u].err |= e.err;
            continue;
        }
        if (S[u] == __builtin_inf()) continue;  // a position fell below --mismatch: :125-127
        const uint32_t cn = 16u * nmax - skipped[u];
        if (cn == 0) continue;                  // :129-131
        out[u].x = (1.0 / (double)cn) * S[u];   // :137
        out[u].mm = cm[u];
        out[u].n = cn;
    }
}

// One sub-overlap with 64-symbol fetch groups: a lane pulls four consecutive 16-byte pieces of each
// stream back to back (one whole 64-byte line of an aligned stream), so a line is fetched into L1 once
// and consumed at once instead of being re-requested by four separate loop iterations that other
// waves' lines evict in between.  The next group is prefetched while the current one is scored.
template <typename SymT, int LG, int G /* 16-symbol chunks per group */>
__device__ __forceinline__ void score_sub_wide(const SymT* __restrict__ sym, const Sub& s, const char* lut,
                                               const uint32_t* masktab, uint32_t Kp, uint32_t nsym_word,
                                               uint32_t min_read_len, SubScore& out) {
    using T = Tr<SymT>;
    out.x = -__builtin_inf();
    out.mm = 1;
    out.n = 1;
    out.err = s.fatal;
    const uint32_t L = sub_positions(s, min_read_len);
    if (L == 0) return;
    const uint32_t nch = (L + 15u) >> 4;
    const SymT* a = sym + s.offA + s.pos;
    const SymT* b = sym + s.offB;
    double S = 0.0;
    uint32_t skipped = 0, cm = 0;
    uint32_t na[G][T::kWords] = {}, nb[G][T::kWords] = {};
#pragma unroll
    for (int q = 0; q < G; ++q)
        if ((uint32_t)q < nch) {
            __builtin_memcpy(na[q], a + 16u * q, sizeof(na[q]));
            __builtin_memcpy(nb[q], b + 16u * q, sizeof(nb[q]));
        }
    for (uint32_t c0 = 0; c0 < nch; c0 += G) {
        uint32_t ca[G][T::kWords], cb[G][T::kWords];
#pragma unroll
        for (int q = 0; q < G; ++q) {
#pragma unroll
            for (int w = 0; w < T::kWords; ++w) {
                ca[q][w] = na[q][w];
                cb[q][w] = nb[q][w];
            }
        }
#pragma unroll
        for (int q = 0; q < G; ++q)
            if (c0 + G + q < nch) {
                __builtin_memcpy(na[q], a + 16u * (c0 + G + q), sizeof(na[q]));
                __builtin_memcpy(nb[q], b + 16u * (c0 + G + q), sizeof(nb[q]));
            }
#pragma unroll
        for (int q = 0; q < G; ++q) {
            const uint32_t c = c0 + q;
            if (c < nch) {
                uint32_t keep[T::kWords];
                const uint32_t rem = L - 16u * c;
                __builtin_memcpy(keep, masktab + (rem >= 16u ? 16u : rem) * T::kWords, sizeof(keep));
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    double t[8];
                    half_chunk_terms<SymT, LG>(ca[q] + h * (T::kWords / 2), cb[q] + h * (T::kWords / 2), keep + h * (T::kWords / 2),
                                               nsym_word, lut, Kp, t, skipped, cm);
#pragma unroll
                    for (int k = 0; k < 8; ++k) S += t[k];  // :119, strictly in position order
                }
            }
        }
    }
    if (S != S) {  // an invalid symbol inside the window
        const SubScore e = score_sub_slow<SymT>(a, b, L, lut, Kp);
        out.x = e.x;
        out.mm = e.mm;
        out.n = e.n;
        out.err |= e.err;
        return;
    }
    if (S == __builtin_inf()) return;
    const uint32_t cn = 16u * nch - skipped;
    if (cn == 0) return;
    out.x = (1.0 / (double)cn) * S;
    out.mm = cm;
    out.n = cn;
}

// exp(x) > T  in x-space: 1 pass, 0 fail, 2 ambiguous
__device__ __forceinline__ uint32_t band_test(double x, const Band& b) { return x > b.hi ? 1u : (x <= b.lo ? 0u : 2u); }

// The tail of compute_overlap + the 3-way class of process_overlaps for one candidate whose sub-overlap
// results are known: mismatch_rate = max of the two (:254), class in x-space (:404-413), result record.
__device__ __forceinline__ hc_result_rec classify_and_store(const ScoreParams& prm, int ns, const SubScore& s1, const SubScore& s2,
                                                            uint64_t i, hc_result_rec* __restrict__ out) {
    hc_result_rec res;
    // mismatch_rate = float(mismatch_count)/total_len (:132); std::max over the two (:254)
    const double m1 = (double)(float)s1.mm / (double)s1.n;
    uint32_t mm = s1.mm, nn = s1.n;
    double mrate = m1;
    if (ns == 2) {
        const double m2 = (double)(float)s2.mm / (double)s2.n;
        if (m1 < m2) {
            mrate = m2;
            mm = s2.mm;
            nn = s2.n;
        }
    }
    // :404-413 in x-space; flags bit0 / bit1: threshold < 0, every score (0 included) passes
    const bool e_all = prm.flags & 1u, o_all = prm.flags & 2u;
    uint32_t e = e_all ? 1u : band_test(s1.x, prm.edge);
    uint32_t o = o_all ? 1u : band_test(s1.x, prm.ov);
    if (ns == 2) {
        const uint32_t e2 = e_all ? 1u : band_test(s2.x, prm.edge), o2 = o_all ? 1u : band_test(s2.x, prm.ov);
        e = (e == 0 || e2 == 0) ? 0u : ((e == 1 && e2 == 1) ? 1u : 2u);
        o = (o == 0 || o2 == 0) ? 0u : ((o == 1 && o2 == 1) ? 1u : 2u);
    }
    uint32_t cls;
    if (s1.err | s2.err) cls = HC_CLS_ERROR;
    else if (e == 1) cls = HC_CLS_EDGE;
    else if (e == 2) cls = HC_CLS_AMBIG;
    else if (mrate <= prm.merge_contigs) cls = HC_CLS_EDGE_MC;
    else if (o == 1) cls = HC_CLS_NONEDGE;
    else if (o == 2) cls = HC_CLS_AMBIG;
    else cls = HC_CLS_DROP;
    res.x1 = s1.x;
    res.x2 = s2.x;
    res.mm = mm;
    res.n_cls = (nn & 0x0FFFFFFFu) | (cls << 28);
    out[i] = res;
    return res;
}

// One candidate, one lane: score its sub-overlaps and write the result record.
template <typename SymT, int VAR, int LG>
__device__ __forceinline__ hc_result_rec score_candidate(const ScoreParams& prm, const SymT* __restrict__ sym, const char* lut,
                                                         const uint32_t* masktab, uint32_t Kp, uint32_t nsym_word, int ns,
                                                         const Sub& sub0, const Sub& sub1, uint64_t i,
                                                         hc_result_rec* __restrict__ out) {
    if (ns == 0) {
        hc_result_rec res;
        res.x1 = -__builtin_inf();
        res.x2 = __builtin_nan("");
        res.mm = 1;
        res.n_cls = 1u | (HC_CLS_ERROR << 28);
        out[i] = res;
        return res;
    }
    SubScore s1, s2;
    s2.x = __builtin_nan("");
    s2.mm = 0;
    s2.n = 1;
    s2.err = 0;
    constexpr bool kPre = (VAR & 2) != 0;
    if (VAR & 4) {
        constexpr int kG = (VAR & 3) == 0 ? 4 : ((VAR & 3) == 1 ? 2 : ((VAR & 3) == 2 ? 8 : 3));
        score_sub_wide<SymT, LG, kG>(sym, sub0, lut, masktab, Kp, nsym_word, prm.min_read_len, s1);
        if (ns == 2) score_sub_wide<SymT, LG, kG>(sym, sub1, lut, masktab, Kp, nsym_word, prm.min_read_len, s2);
    } else if (ns == 2) {
        if (VAR & 1) {
            const Sub subs[2] = {sub0, sub1};
            SubScore r[2];
            score_subs<SymT, 2, kPre, LG>(sym, subs, lut, masktab, Kp, nsym_word, prm.min_read_len, r);
            s1 = r[0];
            s2 = r[1];
        } else {
            score_subs<SymT, 1, kPre, LG>(sym, &sub0, lut, masktab, Kp, nsym_word, prm.min_read_len, &s1);
            score_subs<SymT, 1, kPre, LG>(sym, &sub1, lut, masktab, Kp, nsym_word, prm.min_read_len, &s2);
        }
    } else {
        score_subs<SymT, 1, kPre, LG>(sym, &sub0, lut, masktab, Kp, nsym_word, prm.min_read_len, &s1);
    }

    return classify_and_store(prm, ns, s1, s2, i, out);
}

// What the multi-GPU collection needs of a batch, produced by the scoring kernel itself (score_kernel_rows): every
// record that is not dropped is appended, tagged with its global candidate index, to a payload whose row 0 counts
// them.  One atomic per wave and iteration; the rows arrive in no particular order (they carry their index).
struct RowSink {
    hc_gather_row* payload;  // cap + 1 rows; payload[0].index = number of appended rows (may exceed cap: overflow)
    uint64_t cap;
    uint64_t base_index;
};

// Called by ALL lanes of the workgroup (uniform control flow; `valid` = this lane scored candidate i).  One atomic per
// workgroup and iteration: same-address atomics serialise in L2, and one per wave cost 0.1 ms per 2 M candidates.
__device__ __forceinline__ void append_rows_block(const RowSink& sink, bool valid, const hc_result_rec& res, uint64_t i, uint32_t* lds4) {
    const bool keep = valid && (res.n_cls >> 28) != HC_CLS_DROP;
    const uint64_t m = __ballot(keep);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    if (lane == 0) lds4[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (uint32_t w = 0; w < n_waves; w++) {
            const uint32_t c = lds4[w];
            lds4[w] = total;  // exclusive offsets of the waves
            total += c;
        }
        unsigned long long base = 0;
        if (total) base = atomicAdd((unsigned long long*)&sink.payload[0].index, (unsigned long long)total);
        lds4[16] = (uint32_t)base;
        lds4[17] = (uint32_t)(base >> 32);
    }
    __syncthreads();
    if (keep) {
        const uint64_t base = ((uint64_t)lds4[17] << 32) | lds4[16];
        const uint64_t pos = base + lds4[wave] + (uint64_t)__popcll(m & ((1ull << lane) - 1ull));
        if (pos < sink.cap) {
            hc_gather_row r;
            r.index = sink.base_index + i;
            r.x1 = res.x1;
            r.x2 = res.x2;
            r.mm = res.mm;
            r.n_cls = res.n_cls;
            sink.payload[1 + pos] = r;
        }
    }
    __syncthreads();  // lds4 is reused by the next iteration
}

// VAR bit0: interleave the two sub-overlaps of a candidate; bit1: software prefetch of the next
// chunk; bit2: 64/32/128/48-symbol fetch groups (score_sub_wide).  LG: log2 of the 8-bit-symbol table
// dimension (3..6; ignored for 16-bit symbols).  BAL: block-local length balancing (below).
template <typename SymT, int VAR, int LG, bool BAL, bool ROWS = false>
__device__ __forceinline__ void score_kernel_body(const StoreView& st, const ScoreParams& prm, const double* __restrict__ lut_g,
                                                  const hc_overlap_rec* __restrict__ in, uint64_t n,
                                                  hc_result_rec* __restrict__ out, const uint32_t* __restrict__ perm,
                                                  const RowSink* sink = nullptr) {
    extern __shared__ __attribute__((aligned(16))) double lut_s[];
    const uint32_t lut_n = st.lut_bytes >> 3;
    for (uint32_t i = threadIdx.x; i < lut_n; i += blockDim.x) lut_s[i] = lut_g[i];
    uint32_t* masktab = (uint32_t*)(lut_s + lut_n);
    fill_mask_table<SymT>(masktab, threadIdx.x, blockDim.x);
    __syncthreads();
    const char* lut = (const char*)lut_s;

    const SymT* sym = (const SymT*)st.sym;
    const uint32_t Kp = st.K + 2u;
    const uint32_t nsym = (sizeof(SymT) == 1 && LG == 6) ? (kWideN << 2) : ((st.K << 3) | kCodeN);
    const uint32_t nsym_word = sizeof(SymT) == 1 ? nsym * 0x01010101u : nsym * 0x00010001u;
    // with a permutation, slot s scores candidate perm[s] (neighbouring lanes share reads) and writes its
    // record back to the candidate's own position: out[i] <-> in[i] always holds
    if (ROWS) {  // workgroup-uniform loop: every lane reaches the row append (it synchronises the workgroup)
        uint32_t* lds4 = masktab + 17 * Tr<SymT>::kWords;  // 18 words of the 392 behind the mask table
        const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
        for (uint64_t block_base = (uint64_t)blockIdx.x * blockDim.x; block_base < n; block_base += stride) {
            const uint64_t i = block_base + threadIdx.x;
            hc_result_rec res;
            res.n_cls = 0;
            if (i < n) {
                hc_overlap_rec rec;
                const uint4* p = (const uint4*)(in + i);
                const uint4 a = p[0], b = p[1];
                __builtin_memcpy(&rec, &a, 16);
                __builtin_memcpy((char*)&rec + 16, &b, 16);
                Sub sub0, sub1;
                const int ns = resolve<(int)sizeof(SymT)>(st, rec, sub0, sub1);
                res = score_candidate<SymT, VAR, LG>(prm, sym, lut, masktab, Kp, nsym_word, ns, sub0, sub1, i, out);
            }
            append_rows_block(*sink, i < n, res, i, lds4);
        }
        return;
    }
    if (!BAL) {
        const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
        for (uint64_t slot = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; slot < n; slot += stride) {
            const uint64_t i = perm ? (uint64_t)perm[slot] : slot;
            hc_overlap_rec rec;
            {
                const uint4* p = (const uint4*)(in + i);
                const uint4 a = p[0], b = p[1];
                __builtin_memcpy(&rec, &a, 16);
                __builtin_memcpy((char*)&rec + 16, &b, 16);
            }
            Sub sub0, sub1;
            const int ns = resolve<(int)sizeof(SymT)>(st, rec, sub0, sub1);
            score_candidate<SymT, VAR, LG>(prm, sym, lut, masktab, Kp, nsym_word, ns, sub0, sub1, i, out);
        }
        return;
    }
    // Block-local length balancing (read sets with mixed sequence lengths).  A wave runs as long as its
    // longest lane; with mixed-length contigs (BASELINE config 5) the mean lane is busy 29 % of that time.
    // When the overlap lengths of the 256 candidates of this workgroup differ widely, they are
    // redistributed over the lanes by length (LDS counting sort over quarter-octave length classes, longest
    // first), so every wave gets similar work; the candidates stay inside their workgroup, which keeps the
    // read-sharing locality (reordering over larger windows measured slower).
    uint32_t* bal = masktab + 17 * Tr<SymT>::kWords;  // [0..127] class histogram / offsets, [128..383] order, [384..391] reduce
    const uint32_t tid = threadIdx.x;
    for (uint64_t block_base = (uint64_t)blockIdx.x * blockDim.x; block_base < n; block_base += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t slot = block_base + tid;
        // phase 1: the length class of the candidate in this lane's own slot (its state is dead before phase 2)
        uint32_t chunks = 0;
        if (slot < n) {
            const hc_overlap_rec rec = in[perm ? (uint64_t)perm[slot] : slot];
            Sub s0, s1;
            const int ns = resolve<(int)sizeof(SymT)>(st, rec, s0, s1);
            if (ns >= 1) chunks = (sub_positions(s0, prm.min_read_len) + 15u) >> 4;
            if (ns == 2) {
                const uint32_t c1 = (sub_positions(s1, prm.min_read_len) + 15u) >> 4;
                chunks = c1 > chunks ? c1 : chunks;
            }
        }
        uint32_t wmax = chunks, wsum = chunks;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const uint32_t m = (uint32_t)__shfl_xor((int)wmax, o, 64);
            wmax = m > wmax ? m : wmax;
            wsum += (uint32_t)__shfl_xor((int)wsum, o, 64);
        }
        if ((tid & 63u) == 0) {
            bal[384 + (tid >> 6)] = wmax;
            bal[388 + (tid >> 6)] = wsum;
        }
        if (tid < 128) bal[tid] = 0;
        __syncthreads();
        uint32_t bmax = 0, bsum = 0;
        for (uint32_t w = 0; w < (blockDim.x >> 6); ++w) {
            bmax = bal[384 + w] > bmax ? bal[384 + w] : bmax;
            bsum += bal[388 + w];
        }
        // worth it when the longest overlap is at least twice the block's mean and there is real work to balance
        if (bmax >= 16u && (uint64_t)bmax * blockDim.x > 2ull * bsum) {
            uint32_t cls = 0;  // quarter-octave class of the chunk count
            if (chunks > 1) {
                const uint32_t lg = 31u - (uint32_t)__builtin_clz(chunks);
                cls = lg * 4u + (lg >= 2 ? (chunks >> (lg - 2)) & 3u : 0u);
            }
            cls = cls > 127u ? 127u : cls;
            atomicAdd(&bal[cls], 1u);
            __syncthreads();
            if (tid < 64) {  // exclusive scan over the classes, longest class first
                const uint32_t c0 = bal[127 - 2 * tid], c1 = bal[126 - 2 * tid];
                uint32_t incl = c0 + c1;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const uint32_t up = (uint32_t)__shfl_up((int)incl, o, 64);
                    if ((int)tid >= o) incl += up;
                }
                const uint32_t excl = incl - (c0 + c1);
                bal[127 - 2 * tid] = excl;
                bal[126 - 2 * tid] = excl + c0;
            }
            __syncthreads();
            const uint32_t at = atomicAdd(&bal[cls], 1u);
            bal[128 + at] = tid;
            __syncthreads();
            slot = block_base + bal[128 + tid];
        }
        __syncthreads();  // bal is reused by the next iteration
        // phase 2: score the candidate of the (possibly reassigned) slot
        if (slot < n) {
            const uint64_t i = perm ? (uint64_t)perm[slot] : slot;
            hc_overlap_rec rec;
            {
                const uint4* p = (const uint4*)(in + i);
                const uint4 a = p[0], b = p[1];
                __builtin_memcpy(&rec, &a, 16);
                __builtin_memcpy((char*)&rec + 16, &b, 16);
            }
            Sub sub0, sub1;
            const int ns = resolve<(int)sizeof(SymT)>(st, rec, sub0, sub1);
            score_candidate<SymT, VAR, LG>(prm, sym, lut, masktab, Kp, nsym_word, ns, sub0, sub1, i, out);
        }
    }
}

template <typename SymT, int VAR, int LG, bool BAL>
__global__ __launch_bounds__(256) void score_kernel(StoreView st, ScoreParams prm, const double* __restrict__ lut_g,
                                                    const hc_overlap_rec* __restrict__ in, uint64_t n,
                                                    hc_result_rec* __restrict__ out,
                                                    const uint32_t* __restrict__ perm) {
    score_kernel_body<SymT, VAR, LG, BAL>(st, prm, lut_g, in, n, out, perm);
}

// The scoring kernel that also feeds the multi-GPU collection (RowSink above); read sets without length balancing.
template <typename SymT, int VAR, int LG>
__global__ __launch_bounds__(256) void score_kernel_rows(StoreView st, ScoreParams prm, const double* __restrict__ lut_g,
                                                         const hc_overlap_rec* __restrict__ in, uint64_t n,
                                                         hc_result_rec* __restrict__ out, const uint32_t* __restrict__ perm,
                                                         RowSink sink) {
    score_kernel_body<SymT, VAR, LG, false, true>(st, prm, lut_g, in, n, out, perm, &sink);
}

// The same kernel for workgroups of up to 1 024 lanes.  A large quality alphabet means a large log table in LDS
// (64 KiB for the wide 8-bit symbols, up to 150 KiB for 16-bit symbols), and with one table per 256-lane
// workgroup only 2 (or 1) workgroups fit a CU: 8 (4) waves, too few to hide the gather latency.  One table
// shared by 512 lanes restores 16 waves per CU (used for the wide 8-bit symbols; see launch_score).
template <typename SymT, int VAR, int LG>
__global__ __launch_bounds__(1024) void score_kernel_wide_wg(StoreView st, ScoreParams prm, const double* __restrict__ lut_g,
                                                             const hc_overlap_rec* __restrict__ in, uint64_t n,
                                                             hc_result_rec* __restrict__ out,
                                                             const uint32_t* __restrict__ perm) {
    score_kernel_body<SymT, VAR, LG, false>(st, prm, lut_g, in, n, out, perm);
}

// ---------------------------------------------------------------------------
// Row-staged variant.  The per-lane gather of the kernel above makes every load instruction touch
// up to 64 different cache lines, and the vector-memory front end (TA) is what saturates first
// (profiles/traffic_c2.json: TA busy 87 %).  Here a wave first copies the windows of its 64
// candidates into wave-private LDS with ROW-COALESCED loads — 6 consecutive lanes fetch the 6
// 16-byte pieces of one 96-byte row, so an instruction touches ~20 lines instead of ~64+ and every
// fetched line is used at once — and then every lane scores its own two rows out of LDS
// (row stride 112 B: conflict-free for ds_read_b128).  Same arithmetic, same order, same results.
constexpr int kStageRowBytes = 112;   // 96 data bytes + 16: 28-word stride, conflict-free b128 reads
constexpr int kStagePieces = 6;       // 16-byte pieces per row and round
constexpr int kStageDescBytes = 32;   // per-lane {offA, offB, L}
constexpr int kStageWaveBytes = 2 * 64 * kStageRowBytes + 64 * kStageDescBytes;
typedef uint4 __attribute__((aligned(1))) uint4_unaligned;

template <typename SymT, int LG>
__global__ __launch_bounds__(256) void score_kernel_staged(StoreView st, ScoreParams prm,
                                                           const double* __restrict__ lut_g,
                                                           const hc_overlap_rec* __restrict__ in, uint64_t n,
                                                           hc_result_rec* __restrict__ out,
                                                           const uint32_t* __restrict__ perm) {
    using T = Tr<SymT>;
    constexpr int SB = (int)sizeof(SymT);
    constexpr uint32_t RSYM = 96 / SB;    // symbols per row and round
    constexpr int CH = (int)RSYM / 16;    // 16-symbol chunks per round
    constexpr uint32_t PSYM = 16 / SB;    // symbols per 16-byte piece
    extern __shared__ __attribute__((aligned(16))) double lut_s[];
    const uint32_t lut_n = st.lut_bytes >> 3;
    for (uint32_t i = threadIdx.x; i < lut_n; i += blockDim.x) lut_s[i] = lut_g[i];
    uint32_t* masktab = (uint32_t*)(lut_s + lut_n);
    fill_mask_table<SymT>(masktab, threadIdx.x, blockDim.x);
    __syncthreads();
    const char* lut = (const char*)lut_s;
    const uint32_t lane = threadIdx.x & 63u, wib = threadIdx.x >> 6;
    char* stage = (char*)(masktab + 17 * T::kWords);
    stage += (16 - ((uintptr_t)stage & 15)) & 15;
    char* rowsA = stage + wib * kStageWaveBytes;
    char* rowsB = rowsA + 64 * kStageRowBytes;
    char* desc = rowsB + 64 * kStageRowBytes;

    const SymT* sym = (const SymT*)st.sym;
    const uint32_t Kp = st.K + 2u;
    const uint32_t nsym = (SB == 1 && LG == 6) ? (kWideN << 2) : ((st.K << 3) | kCodeN);
    const uint32_t nsym_word = SB == 1 ? nsym * 0x01010101u : nsym * 0x00010001u;
    const uint64_t wave_stride = (uint64_t)gridDim.x * (blockDim.x >> 6) * 64u;
    for (uint64_t base = ((uint64_t)blockIdx.x * (blockDim.x >> 6) + wib) * 64u; base < n; base += wave_stride) {
        const uint64_t slot = base + lane;
        const bool active = slot < n;
        uint64_t i = 0;
        hc_overlap_rec rec;
        Sub sub0, sub1;
        int ns = 0;
        if (active) {
            i = perm ? (uint64_t)perm[slot] : slot;
            const uint4* p = (const uint4*)(in + i);
            const uint4 a = p[0], b = p[1];
            __builtin_memcpy(&rec, &a, 16);
            __builtin_memcpy((char*)&rec + 16, &b, 16);
            ns = resolve<SB>(st, rec, sub0, sub1);
        }
        // one sub-overlap of every lane's candidate: wave-cooperative staging, lane-private scoring
        auto run_sub = [&](const Sub& sb, bool have, SubScore& r) {
            uint32_t L = 0;
            uint64_t offA = 0, offB = 0;
            if (have) {
                r.x = -__builtin_inf();
                r.mm = 1;
                r.n = 1;
                r.err = sb.fatal;
                L = sub_positions(sb, prm.min_read_len);
                offA = sb.offA + sb.pos;
                offB = sb.offB;
            }
            // publish this lane's row descriptor to the wave
            *(uint2*)(desc + lane * kStageDescBytes) = make_uint2((uint32_t)offA, (uint32_t)(offA >> 32));
            *(uint2*)(desc + lane * kStageDescBytes + 8) = make_uint2((uint32_t)offB, (uint32_t)(offB >> 32));
            *(uint32_t*)(desc + lane * kStageDescBytes + 16) = L;
            uint32_t Lmax = L;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const uint32_t other = (uint32_t)__shfl_xor((int)Lmax, o, 64);
                Lmax = other > Lmax ? other : Lmax;
            }
            Lmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)Lmax);
            __builtin_amdgcn_wave_barrier();
            double S = 0.0;
            uint32_t skipped = 0, cm = 0;
            for (uint32_t rb = 0; rb < Lmax; rb += RSYM) {
                // ---- stage: row-coalesced global loads -> wave-private LDS rows
#pragma unroll
                for (int k = 0; k < kStagePieces; ++k) {
                    const uint32_t f = (uint32_t)k * 64u + lane;
                    const uint32_t row = f / kStagePieces, piece = f - row * kStagePieces;
                    const uint4 d0 = *(const uint4*)(desc + row * kStageDescBytes);
                    const uint32_t rowL = *(const uint32_t*)(desc + row * kStageDescBytes + 16);
                    const uint32_t so = rb + piece * PSYM;
                    if (so < rowL) {
                        const uint64_t oa = (((uint64_t)d0.y << 32) | d0.x) + so;
                        const uint64_t ob = (((uint64_t)d0.w << 32) | d0.z) + so;
                        // unaligned: the A window starts at an arbitrary symbol
                        const uint4 va = *(const uint4_unaligned*)(sym + oa);
                        const uint4 vb = *(const uint4_unaligned*)(sym + ob);
                        const uint32_t dst = row * kStageRowBytes + piece * 16u;
                        *(uint4*)(rowsA + dst) = va;
                        *(uint4*)(rowsB + dst) = vb;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                // ---- score: every lane walks its own two rows
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    const uint32_t p0 = rb + 16u * (uint32_t)c;
                    if (p0 < L) {
                        uint32_t wa[T::kWords], wb[T::kWords], keep[T::kWords];
                        __builtin_memcpy(wa, rowsA + lane * kStageRowBytes + c * 16 * SB, sizeof(wa));
                        __builtin_memcpy(wb, rowsB + lane * kStageRowBytes + c * 16 * SB, sizeof(wb));
                        const uint32_t rem = L - p0;
                        __builtin_memcpy(keep, masktab + (rem >= 16u ? 16u : rem) * T::kWords, sizeof(keep));
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            double t[8];
                            half_chunk_terms<SymT, LG>(wa + h * (T::kWords / 2), wb + h * (T::kWords / 2), keep + h * (T::kWords / 2),
                                                   nsym_word, lut, Kp, t, skipped, cm);
#pragma unroll
                            for (int k = 0; k < 8; ++k) S += t[k];  // :119, strictly in position order
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
            if (L == 0) return;
            if (S != S) {  // an invalid symbol inside the window
                const SubScore e = score_sub_slow<SymT>(sym + offA, sym + offB, L, lut, Kp);
                r.x = e.x;
                r.mm = e.mm;
                r.n = e.n;
                r.err |= e.err;
                return;
            }
            if (S == __builtin_inf()) return;
            const uint32_t cn = 16u * ((L + 15u) >> 4) - skipped;
            if (cn == 0) return;
            r.x = (1.0 / (double)cn) * S;
            r.mm = cm;
            r.n = cn;
        };
        SubScore s1, s2;
        s1.x = -__builtin_inf();
        s1.mm = 1;
        s1.n = 1;
        s1.err = 0;
        s2.x = __builtin_nan("");
        s2.mm = 0;
        s2.n = 1;
        s2.err = 0;
        run_sub(sub0, active && ns >= 1, s1);
        const unsigned long long any2 = __ballot(active && ns == 2);
        if (any2) run_sub(sub1, active && ns == 2, s2);
        if (!active) continue;
        hc_result_rec res;
        if (ns == 0) {
            res.x1 = -__builtin_inf();
            res.x2 = __builtin_nan("");
            res.mm = 1;
            res.n_cls = 1u | (HC_CLS_ERROR << 28);
            out[i] = res;
            continue;
        }
        const double m1 = (double)(float)s1.mm / (double)s1.n;
        uint32_t mm = s1.mm, nn = s1.n;
        double mrate = m1;
        if (ns == 2) {
            const double m2 = (double)(float)s2.mm / (double)s2.n;
            if (m1 < m2) {
                mrate = m2;
                mm = s2.mm;
                nn = s2.n;
            }
        }
        const bool e_all = prm.flags & 1u, o_all = prm.flags & 2u;
        uint32_t e = e_all ? 1u : band_test(s1.x, prm.edge);
        uint32_t o = o_all ? 1u : band_test(s1.x, prm.ov);
        if (ns == 2) {
            const uint32_t e2 = e_all ? 1u : band_test(s2.x, prm.edge), o2 = o_all ? 1u : band_test(s2.x, prm.ov);
            e = (e == 0 || e2 == 0) ? 0u : ((e == 1 && e2 == 1) ? 1u : 2u);
            o = (o == 0 || o2 == 0) ? 0u : ((o == 1 && o2 == 1) ? 1u : 2u);
        }
        uint32_t cls;
        if (s1.err | s2.err) cls = HC_CLS_ERROR;
        else if (e == 1) cls = HC_CLS_EDGE;
        else if (e == 2) cls = HC_CLS_AMBIG;
        else if (mrate <= prm.merge_contigs) cls = HC_CLS_EDGE_MC;
        else if (o == 1) cls = HC_CLS_NONEDGE;
        else if (o == 2) cls = HC_CLS_AMBIG;
        else cls = HC_CLS_DROP;
        res.x1 = s1.x;
        res.x2 = s2.x;
        res.mm = mm;
        res.n_cls = (nn & 0x0FFFFFFFu) | (cls << 28);
        out[i] = res;
    }
}

// Sum of overlapped positions / sub-overlaps over a batch (algorithmic-bytes multiplier).
__global__ __launch_bounds__(256) void count_positions_kernel(StoreView st, uint32_t min_read_len,
                                                              const hc_overlap_rec* __restrict__ in, uint64_t n,
                                                              unsigned long long* __restrict__ totals) {
    unsigned long long pos = 0, subs = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const hc_overlap_rec rec = in[i];
        Sub s0, s1;
        const int ns = st.symbytes == 1 ? resolve<1>(st, rec, s0, s1) : resolve<2>(st, rec, s0, s1);
        if (ns >= 1) pos += sub_positions(s0, min_read_len);
        if (ns == 2) pos += sub_positions(s1, min_read_len);
        subs += (unsigned long long)ns;
    }
    for (int off = 32; off > 0; off >>= 1) {
        pos += __shfl_down(pos, off, 64);
        subs += __shfl_down(subs, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&totals[0], pos);
        atomicAdd(&totals[1], subs);
    }
}

// ---------------------------------------------------------------------------
// Candidate reorder for locality: key = the smaller read index of the pair (the grouping real
// overlap files have, scripts/sfo2overlaps.py:53); a stable LSD radix sort of (key, index) pairs
// gives the permutation the scoring kernel walks.  hipCUB is used as a utility here; the hot op
// stays the hand-written kernel above.
// key = [length bucket : 6 bits][smaller read index : 26 bits].  The bucket is floor(log2) of the
// candidate's overlapped positions in quarter-octave steps, so that the lanes of a wave run a
// similar number of chunks (mixed-length contigs, BASELINE config 5) while candidates of one read
// stay together inside a bucket.  bucket_shift = 32 disables bucketing (key = read index only).
template <int SB>
__global__ __launch_bounds__(256) void make_keys_kernel(StoreView st, uint32_t min_read_len,
                                                        const hc_overlap_rec* __restrict__ in, uint32_t n,
                                                        uint32_t use_buckets, uint32_t* __restrict__ keys,
                                                        uint32_t* __restrict__ idx) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const hc_overlap_rec rec = in[i];
    uint32_t key = rec.read1 < rec.read2 ? rec.read1 : rec.read2;
    if (use_buckets) {
        Sub s0, s1;
        const int ns = resolve<SB>(st, rec, s0, s1);
        uint32_t L = 0;
        if (ns >= 1) L = sub_positions(s0, min_read_len);
        if (ns == 2) {
            const uint32_t L1 = sub_positions(s1, min_read_len);
            L = L1 > L ? L1 : L;
        }
        const uint32_t chunks = (L + 15u) >> 4;
        uint32_t b = 0;
        if (chunks > 1) {
            const uint32_t lg = 31u - __builtin_clz(chunks);          // floor(log2)
            const uint32_t frac = lg >= 2 ? (chunks >> (lg - 2)) & 3u : 0u;  // quarter-octave
            b = lg * 4u + frac;
        }
        key = (key & 0x03FFFFFFu) | ((63u - (b > 63u ? 63u : b)) << 26);  // longest first
    }
    keys[i] = key;
    idx[i] = i;
}

size_t reorder_temp_bytes(uint32_t n) {
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                             (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)n);
    return bytes;
}

// keys_in/idx_in are scratch (n each); perm_out receives the permutation.
hipError_t launch_reorder(const StoreView& st, uint32_t min_read_len, const hc_overlap_rec* in, uint32_t n,
                          bool use_buckets, uint32_t* keys_in, uint32_t* keys_out, uint32_t* idx_in, uint32_t* perm_out,
                          void* temp, size_t temp_bytes, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const uint32_t ub = use_buckets && st.n_reads < (1u << 26) ? 1u : 0u;
    if (st.symbytes == 1)
        hipLaunchKernelGGL(make_keys_kernel<1>, dim3((n + 255) / 256), dim3(256), 0, stream, st, min_read_len, in, n, ub,
                           keys_in, idx_in);
    else
        hipLaunchKernelGGL(make_keys_kernel<2>, dim3((n + 255) / 256), dim3(256), 0, stream, st, min_read_len, in, n, ub,
                           keys_in, idx_in);
    int end_bit = 32;
    if (!ub) {
        end_bit = 1;
        while (end_bit < 32 && (st.n_reads >> end_bit)) end_bit++;  // keys < n_reads
    }
    return hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys_in, keys_out, idx_in, perm_out, (int)n, 0, end_bit, stream);
}

// ---------------------------------------------------------------------------
// Compaction of the records the host / the gather still need (class != DROP), in sequence order.
struct NotDropped {
    const hc_result_rec* res;
    __device__ __forceinline__ bool operator()(const uint32_t& i) const { return (res[i].n_cls >> 28) != HC_CLS_DROP; }
};

size_t compact_temp_bytes(uint32_t n) {
    size_t bytes = 0;
    hipcub::CountingInputIterator<uint32_t> it(0);
    (void)hipcub::DeviceSelect::If(nullptr, bytes, it, (uint32_t*)nullptr, (unsigned long long*)nullptr, (int)n,
                                   NotDropped{nullptr});
    return bytes;
}

hipError_t launch_compact(const hc_result_rec* res, uint32_t n, uint32_t* idx_out, unsigned long long* count_out, void* temp,
                          size_t temp_bytes, hipStream_t stream) {
    hipcub::CountingInputIterator<uint32_t> it(0);
    return hipcub::DeviceSelect::If(temp, temp_bytes, it, idx_out, count_out, (int)n, NotDropped{res}, stream);
}

__global__ __launch_bounds__(256) void gather_results_kernel(const hc_result_rec* __restrict__ res,
                                                             const uint32_t* __restrict__ idx,
                                                             const unsigned long long* __restrict__ count,
                                                             hc_result_rec* __restrict__ out) {
    const unsigned long long k = *count;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < k;
         i += (unsigned long long)gridDim.x * blockDim.x)
        out[i] = res[idx[i]];
}

hipError_t launch_gather_results(const hc_result_rec* res, const uint32_t* idx, const unsigned long long* count,
                                 hc_result_rec* out, uint32_t n_cu, hipStream_t stream) {
    hipLaunchKernelGGL(gather_results_kernel, dim3(n_cu * 4), dim3(256), 0, stream, res, idx, count, out);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void pack_rows_kernel(const hc_result_rec* __restrict__ res, const uint32_t* __restrict__ idx,
                                                        const unsigned long long* __restrict__ count, unsigned long long cap,
                                                        unsigned long long base, hc_gather_row* __restrict__ rows) {
    unsigned long long k = *count;
    k = k < cap ? k : cap;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < k;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const uint32_t j = idx[i];
        const hc_result_rec r = res[j];
        hc_gather_row o;
        o.index = base + j;
        o.x1 = r.x1;
        o.x2 = r.x2;
        o.mm = r.mm;
        o.n_cls = r.n_cls;
        rows[i] = o;
    }
}

__global__ void pack_header_kernel(const unsigned long long* __restrict__ count, hc_gather_row* __restrict__ header) {
    hc_gather_row h;
    h.index = *count;
    h.x1 = 0;
    h.x2 = 0;
    h.mm = 0;
    h.n_cls = 0;
    *header = h;
}

hipError_t launch_pack_header(const unsigned long long* count, hc_gather_row* header, hipStream_t stream) {
    hipLaunchKernelGGL(pack_header_kernel, dim3(1), dim3(1), 0, stream, count, header);
    return hipGetLastError();
}

hipError_t launch_pack_rows(const hc_result_rec* res, const uint32_t* idx, const unsigned long long* count, uint64_t cap, uint64_t base,
                            hc_gather_row* rows, uint32_t n_cu, hipStream_t stream) {
    hipLaunchKernelGGL(pack_rows_kernel, dim3(n_cu * 4), dim3(256), 0, stream, res, idx, count, (unsigned long long)cap,
                       (unsigned long long)base, rows);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Launch wrappers (called from hc_api.cpp).
hipError_t launch_encode(uint32_t symbytes, const uint8_t* bases, const uint8_t* quals, const uint64_t* raw_off,
                         const uint64_t* seq_off, const uint8_t* qmap, uint32_t n_seq, uint32_t K, void* sym,
                         uint8_t* seq_bad, const uint32_t* read_first_seq, uint32_t n_reads, ReadDesc* descs,
                         hipStream_t stream) {
    if (n_seq == 0) return hipSuccess;
    const uint32_t waves_per_block = 4;
    uint32_t blocks = (n_seq + waves_per_block - 1) / waves_per_block;
    if (blocks > 65536) blocks = 65536;
    if (symbytes == 1 && lut_lg(K) == 6)
        hipLaunchKernelGGL((encode_store_kernel<uint8_t, true>), dim3(blocks), dim3(256), 0, stream, bases, quals, raw_off,
                           seq_off, qmap, n_seq, K, (uint8_t*)sym, seq_bad);
    else if (symbytes == 1)
        hipLaunchKernelGGL((encode_store_kernel<uint8_t, false>), dim3(blocks), dim3(256), 0, stream, bases, quals, raw_off,
                           seq_off, qmap, n_seq, K, (uint8_t*)sym, seq_bad);
    else
        hipLaunchKernelGGL((encode_store_kernel<uint16_t, false>), dim3(blocks), dim3(256), 0, stream, bases, quals, raw_off,
                           seq_off, qmap, n_seq, K, (uint16_t*)sym, seq_bad);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (n_reads)
        hipLaunchKernelGGL(build_read_desc_kernel, dim3((n_reads + 255) / 256), dim3(256), 0, stream, read_first_seq,
                           seq_off, raw_off, seq_bad, n_reads, descs);
    return hipGetLastError();
}

template <typename SymT, int VAR, int LG>
static void launch_score_one(const StoreView& st, const ScoreParams& prm, const double* lut_g, const hc_overlap_rec* in,
                             uint64_t n, hc_result_rec* out, const uint32_t* perm, uint32_t blocks, size_t lds,
                             hipStream_t stream, uint32_t block = 256) {
    if (block > 256) {
        if constexpr (LG >= 5 && (VAR == 4 || VAR == 5))  // the instantiations set_reads can select for large tables
            hipLaunchKernelGGL((score_kernel_wide_wg<SymT, VAR, LG>), dim3(blocks), dim3(block), lds, stream, st, prm, lut_g, in, n,
                               out, perm);
        return;
    }
    if (st.balance)
        hipLaunchKernelGGL((score_kernel<SymT, VAR, LG, true>), dim3(blocks), dim3(256), lds, stream, st, prm, lut_g, in, n, out,
                           perm);
    else
        hipLaunchKernelGGL((score_kernel<SymT, VAR, LG, false>), dim3(blocks), dim3(256), lds, stream, st, prm, lut_g, in, n, out,
                           perm);
}

template <typename SymT, int LG>
static hipError_t launch_score_lg(int var, const StoreView& st, const ScoreParams& prm, const double* lut_g,
                                  const hc_overlap_rec* in, uint64_t n, hc_result_rec* out, const uint32_t* perm,
                                  uint32_t blocks, size_t lds, hipStream_t stream, uint32_t block = 256) {
    switch (var & 7) {
        case 4: launch_score_one<SymT, 4, LG>(st, prm, lut_g, in, n, out, perm, blocks, lds, stream, block); return hipGetLastError();
        case 5: launch_score_one<SymT, 5, LG>(st, prm, lut_g, in, n, out, perm, blocks, lds, stream, block); return hipGetLastError();
        case 6: launch_score_one<SymT, 6, LG>(st, prm, lut_g, in, n, out, perm, blocks, lds, stream); return hipGetLastError();
        case 7: launch_score_one<SymT, 7, LG>(st, prm, lut_g, in, n, out, perm, blocks, lds, stream); return hipGetLastError();
        default: break;
    }
    switch (var & 3) {
        case 0: launch_score_one<SymT, 0, LG>(st, prm, lut_g, in, n, out, perm, blocks, lds, stream); break;
        case 1: launch_score_one<SymT, 1, LG>(st, prm, lut_g, in, n, out, perm, blocks, lds, stream); break;
        case 2: launch_score_one<SymT, 2, LG>(st, prm, lut_g, in, n, out, perm, blocks, lds, stream); break;
        default: launch_score_one<SymT, 3, LG>(st, prm, lut_g, in, n, out, perm, blocks, lds, stream); break;
    }
    return hipGetLastError();
}

template <typename SymT, int LG>
static hipError_t launch_staged_lg(const StoreView& st, const ScoreParams& prm, const double* lut_g, const hc_overlap_rec* in,
                                   uint64_t n, hc_result_rec* out, const uint32_t* perm, uint32_t blocks, size_t lds,
                                   hipStream_t stream) {
    hipLaunchKernelGGL((score_kernel_staged<SymT, LG>), dim3(blocks), dim3(256), lds, stream, st, prm, lut_g, in, n, out, perm);
    return hipGetLastError();
}

hipError_t launch_score(const StoreView& st, const ScoreParams& prm, const double* lut_g, const hc_overlap_rec* in,
                        uint64_t n, hc_result_rec* out, const uint32_t* perm, uint32_t n_cu, int variant,
                        hipStream_t stream) {
    if (n == 0) return hipSuccess;
    static const size_t lds_pad = getenv("HC_LDS_PAD") ? (size_t)atol(getenv("HC_LDS_PAD")) : 0;  // occupancy experiments
    const size_t lds = st.lut_bytes + (17 * (st.symbytes == 1 ? 4 : 8) + 392) * sizeof(uint32_t) + lds_pad;
    const uint32_t block = 256;
    const uint32_t lg = lut_lg(st.K);
    const size_t lds_staged = lds + 16 + 4 * kStageWaveBytes;
    if ((variant & 8) && lds_staged <= 160 * 1024) {  // experimental row-staged variant
        uint32_t bpc = (uint32_t)((160 * 1024) / lds_staged);
        if (bpc > 8) bpc = 8;
        uint64_t blocks = (n + block - 1) / block;
        const uint64_t cap = (uint64_t)n_cu * bpc * 2;
        if (blocks > cap) blocks = cap;
        const uint32_t nb = (uint32_t)blocks;
        if (st.symbytes == 2) return launch_staged_lg<uint16_t, 5>(st, prm, lut_g, in, n, out, perm, nb, lds_staged, stream);
        if (lg == 3) return launch_staged_lg<uint8_t, 3>(st, prm, lut_g, in, n, out, perm, nb, lds_staged, stream);
        if (lg == 4) return launch_staged_lg<uint8_t, 4>(st, prm, lut_g, in, n, out, perm, nb, lds_staged, stream);
        if (lg == 5) return launch_staged_lg<uint8_t, 5>(st, prm, lut_g, in, n, out, perm, nb, lds_staged, stream);
        return launch_staged_lg<uint8_t, 6>(st, prm, lut_g, in, n, out, perm, nb, lds_staged, stream);
    }
    // Fill the chip: enough 256-thread blocks for 8 waves per SIMD, bounded by LDS.
    uint32_t blocks_per_cu = 8;
    if (lds > 0) {
        const uint32_t by_lds = (uint32_t)((160 * 1024) / lds);
        if (by_lds < blocks_per_cu) blocks_per_cu = by_lds < 1 ? 1 : by_lds;
    }
    // large table, few workgroups per CU: let more lanes share each table (score_kernel_wide_wg)
    static const long wg_env = getenv("HC_WG") ? atol(getenv("HC_WG")) : 0;  // experiment knob: 256 / 512 / 1024
    // measured (C4, 35 quality values, 64 KiB table): 256 lanes 0.169 ms, 512 lanes 0.152 ms, 1 024 lanes 0.174 ms;
    // the 16-bit-symbol instantiations need more than the 128 registers a 1 024-lane bound leaves and get slower
    uint32_t wg = 256;
    if (!st.balance && (variant & 7) >= 4 && (variant & 7) <= 5 && !(st.symbytes == 1 && lg < 5)) {
        if (st.symbytes == 1 && blocks_per_cu <= 2) wg = 512;
        if (wg_env == 256 || wg_env == 512 || wg_env == 1024) wg = (uint32_t)wg_env;
    }
    if (wg > 256) {
        uint64_t wblocks = (n + wg - 1) / wg;
        const uint64_t wcap = (uint64_t)n_cu * blocks_per_cu * 4;
        if (wblocks > wcap) wblocks = wcap;
        const uint32_t wnb = (uint32_t)wblocks;
        if (st.symbytes == 2) return launch_score_lg<uint16_t, 5>(variant, st, prm, lut_g, in, n, out, perm, wnb, lds, stream, wg);
        if (lg == 5) return launch_score_lg<uint8_t, 5>(variant, st, prm, lut_g, in, n, out, perm, wnb, lds, stream, wg);
        return launch_score_lg<uint8_t, 6>(variant, st, prm, lut_g, in, n, out, perm, wnb, lds, stream, wg);
    }
    uint64_t blocks = (n + block - 1) / block;
    const uint64_t cap = (uint64_t)n_cu * blocks_per_cu * 4;  // grid-stride beyond this
    if (blocks > cap) blocks = cap;
    const uint32_t nb = (uint32_t)blocks;
    if (st.symbytes == 2) return launch_score_lg<uint16_t, 5>(variant, st, prm, lut_g, in, n, out, perm, nb, lds, stream);
    if (lg == 3) return launch_score_lg<uint8_t, 3>(variant, st, prm, lut_g, in, n, out, perm, nb, lds, stream);
    if (lg == 4) return launch_score_lg<uint8_t, 4>(variant, st, prm, lut_g, in, n, out, perm, nb, lds, stream);
    if (lg == 5) return launch_score_lg<uint8_t, 5>(variant, st, prm, lut_g, in, n, out, perm, nb, lds, stream);
    return launch_score_lg<uint8_t, 6>(variant, st, prm, lut_g, in, n, out, perm, nb, lds, stream);
}

// Scoring + row append in one kernel.  Returns hipErrorNotSupported when the read set needs an instantiation that has
// no row-appending twin (length balancing, 16-bit symbols, experimental variants): the caller then compacts separately.
hipError_t launch_score_rows(const StoreView& st, const ScoreParams& prm, const double* lut_g, const hc_overlap_rec* in, uint64_t n,
                             hc_result_rec* out, uint32_t n_cu, int variant, hc_gather_row* payload, uint64_t cap, uint64_t base_index,
                             hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const int v = variant & 7;
    if (st.balance || st.symbytes != 1 || (variant & 8) || (v != 4 && v != 5)) return hipErrorNotSupported;
    const size_t lds = st.lut_bytes + (17 * 4 + 392) * sizeof(uint32_t);
    uint32_t blocks_per_cu = 8;
    const uint32_t by_lds = (uint32_t)((160 * 1024) / lds);
    if (by_lds < blocks_per_cu) blocks_per_cu = by_lds < 1 ? 1 : by_lds;
    uint64_t blocks = (n + 255) / 256;
    const uint64_t cap_blocks = (uint64_t)n_cu * blocks_per_cu * 4;
    if (blocks > cap_blocks) blocks = cap_blocks;
    const uint32_t nb = (uint32_t)blocks;
    const RowSink sink{payload, cap, base_index};
    const uint32_t lg = lut_lg(st.K);
#define HC_ROWS(VARV, LGV) \
    hipLaunchKernelGGL((score_kernel_rows<uint8_t, VARV, LGV>), dim3(nb), dim3(256), lds, stream, st, prm, lut_g, in, n, out, \
                       (const uint32_t*)nullptr, sink)
    if (v == 4) {
        if (lg == 3) HC_ROWS(4, 3);
        else if (lg == 4) HC_ROWS(4, 4);
        else if (lg == 5) HC_ROWS(4, 5);
        else HC_ROWS(4, 6);
    } else {
        if (lg == 3) HC_ROWS(5, 3);
        else if (lg == 4) HC_ROWS(5, 4);
        else if (lg == 5) HC_ROWS(5, 5);
        else HC_ROWS(5, 6);
    }
#undef HC_ROWS
    return hipGetLastError();
}

hipError_t launch_count_positions(const StoreView& st, uint32_t min_read_len, const hc_overlap_rec* in, uint64_t n,
                                  unsigned long long* totals, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    uint64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(count_positions_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, st, min_read_len, in, n,
                       totals);
    return hipGetLastError();
}

template <typename SymT, int LG>
static hipError_t set_lds_limit_lg() {
    const int kMax = 160 * 1024;  // allow the full 160 KiB of LDS for large quality alphabets
    hipError_t e;
    if ((e = hipFuncSetAttribute((const void*)score_kernel<SymT, 0, LG, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kMax)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)score_kernel<SymT, 0, LG, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kMax)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)score_kernel<SymT, 1, LG, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kMax)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)score_kernel<SymT, 1, LG, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kMax)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)score_kernel<SymT, 2, LG, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kMax)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)score_kernel<SymT, 2, LG, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kMax)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)score_kernel<SymT, 3, LG, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kMax)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)score_kernel<SymT, 3, LG, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kMax)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)score_kernel<SymT, 4, LG, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kMax)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)score_kernel<SymT, 4, LG, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kMax)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)score_kernel<SymT, 5, LG, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kMax)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)score_kernel<SymT, 5, LG, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kMax)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)score_kernel<SymT, 6, LG, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kMax)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)score_kernel<SymT, 6, LG, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kMax)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)score_kernel<SymT, 7, LG, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kMax)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)score_kernel<SymT, 7, LG, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kMax)) != hipSuccess) return e;
    if constexpr (sizeof(SymT) == 1) {
        if ((e = hipFuncSetAttribute((const void*)score_kernel_rows<SymT, 4, LG>, hipFuncAttributeMaxDynamicSharedMemorySize, kMax)) != hipSuccess) return e;
        if ((e = hipFuncSetAttribute((const void*)score_kernel_rows<SymT, 5, LG>, hipFuncAttributeMaxDynamicSharedMemorySize, kMax)) != hipSuccess) return e;
    }
    if constexpr (LG >= 5) {
        if ((e = hipFuncSetAttribute((const void*)score_kernel_wide_wg<SymT, 4, LG>, hipFuncAttributeMaxDynamicSharedMemorySize, kMax)) != hipSuccess) return e;
        if ((e = hipFuncSetAttribute((const void*)score_kernel_wide_wg<SymT, 5, LG>, hipFuncAttributeMaxDynamicSharedMemorySize, kMax)) != hipSuccess) return e;
    }
    return hipFuncSetAttribute((const void*)score_kernel_staged<SymT, LG>, hipFuncAttributeMaxDynamicSharedMemorySize, kMax);
}
hipError_t set_score_kernel_lds_limit() {
    hipError_t e;
    if ((e = set_lds_limit_lg<uint8_t, 3>()) != hipSuccess) return e;
    if ((e = set_lds_limit_lg<uint8_t, 4>()) != hipSuccess) return e;
    if ((e = set_lds_limit_lg<uint8_t, 5>()) != hipSuccess) return e;
    if ((e = set_lds_limit_lg<uint8_t, 6>()) != hipSuccess) return e;
    return set_lds_limit_lg<uint16_t, 5>();
}

}  // namespace hc
