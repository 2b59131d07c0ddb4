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
// Mapping: one lane OWNS one candidate — it derives N / mismatch masks and counts with packed-byte
// integer ops (4 positions per VALU op), builds the 16 LDS addresses of the log table of a 16-position chunk with one
// v_perm_b32 each, issues the 16 ds_read_b64 back to back and runs the 16 dependent v_add_f64 — and the lanes of a wave
// FETCH together: quads read 64-byte rows that reach the owner through a per-wave LDS image (score_kernel_coop).  The
// one-lane-one-fetch kernel (score_kernel) remains for contig-length read sets and stores of 4 GiB and more.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>
#include <type_traits>
#include <cstdio>
#include <string>

#include "../../include/hcedge.h"
#include "hc_device.h"
#include "hc_resolve.h"

// cache policy bits of the cooperative fetch's row loads (experiment: 2 = non-temporal)
#ifndef HC_COOP_AUX_A
#define HC_COOP_AUX_A 0
#endif
#ifndef HC_COOP_AUX_B
#define HC_COOP_AUX_B 0
#endif

// Experiment builds only (tools/experiments/ablate.sh): HC_ABLATE is a bit mask of parts of the cooperative kernel that are
// cut out to see what the others cost — results are garbage.  1: no table reads (the LDS look-up of every position becomes a
// register move), 2: no row loads from memory, 4: no passage of the rows through the LDS image; round 6, LDS-DMA form: 8: no A rows
// are fetched, 16: only B rows, TWO steps in flight (profiles/r06_second_step.md).  The shipped library is built without it.
#ifndef HC_ABLATE
#define HC_ABLATE 0
#endif

namespace hc {

// ---------------------------------------------------------------------------
// Store encoding: raw ASCII bases + quality bytes -> symbol slots (both orientations).
// One wave per sequence; lanes stride over positions (coalesced reads and writes).
template <typename SymT, bool WIDE>
__global__ __launch_bounds__(256) void encode_store_kernel(const uint8_t* __restrict__ bases,
                                                           const uint8_t* __restrict__ quals,
                                                           const uint64_t* __restrict__ raw_off,  // [n_seq+1]
                                                           const uint64_t* __restrict__ seq_off,  // [n_seq] symbols
                                                           const uint32_t* __restrict__ rc_delta,  // [n_seq] fwd slot -> rc slot
                                                           const uint8_t* __restrict__ qmap,  // [256] byte -> qidx, 255 = invalid
                                                           uint32_t n_seq, uint32_t K, SymT* __restrict__ sym,
                                                           uint8_t* __restrict__ seq_bad, uint32_t slot_align) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    // smallest 8-bit table (lut_lg == 3): the index's low two bits once more in bits 6-7 of the symbol (hc_device.h)
    const bool dup = sizeof(SymT) == 1 && !WIDE && lut_lg(K) == 3;
    const SymT nsym = WIDE ? (SymT)(kWideN << 2) : (SymT)((K << 3) | kCodeN | (dup ? (K & 3u) << 6 : 0u));
    for (uint32_t q = wave; q < n_seq; q += n_waves) {
        const uint64_t r0 = raw_off[q];
        const uint32_t len = (uint32_t)(raw_off[q + 1] - r0);
        const uint64_t f0 = seq_off[q];
        const uint64_t stride = slot_stride(len, sizeof(SymT), slot_align);  // symbols of one slot (sequence + N padding)
        const uint64_t rc0 = f0 + rc_delta[q];
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
                    const uint32_t rb2 = qx >= wide_first(lut_lg(K)) ? 3u - b2 : 0u;
                    sym[f0 + i] = (SymT)((qx << 2) | b2);
                    sym[rc0 + (len - 1 - i)] = (SymT)((qx << 2) | rb2);
                } else {
                    if (code == kCodeN) qx = K;            // zero row of the log table
                    if (code == kCodeBadBase) qx = K + 1;  // NaN row
                    if (qi == 255) {                       // quality outside [33,127]: always fatal inside an overlap
                        code = kCodeBadQual;
                        qx = K + 1;
                    }
                    const uint32_t rcode = code < 4 ? 3 - code : code;
                    const uint32_t qbits = (qx << 3) | (dup ? (qx & 3u) << 6 : 0u);
                    sym[f0 + i] = (SymT)(qbits | code);
                    sym[rc0 + (len - 1 - i)] = (SymT)(qbits | rcode);
                }
            } else {
                sym[f0 + i] = nsym;
                sym[rc0 + i] = nsym;
            }
        }
        const unsigned long long any_bad = __ballot(bad != 0);
        if (lane == 0) seq_bad[q] = any_bad ? 1 : 0;
    }
}

__global__ __launch_bounds__(256) void build_read_desc_kernel(const uint32_t* __restrict__ read_first_seq,
                                                              const uint64_t* __restrict__ seq_off,
                                                              const uint64_t* __restrict__ raw_off,
                                                              const uint8_t* __restrict__ seq_bad, const uint32_t* __restrict__ rc_delta,
                                                              uint32_t n_reads, ReadDesc* __restrict__ out) {
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
    d.rc_delta = rc_delta[f];
    out[r] = d;
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
    static constexpr int kSymBits = 8;
};
template <>
struct Tr<uint16_t> {
    static constexpr int kSymsPerWord = 2;
    static constexpr int kWords = 8;
    static constexpr uint32_t kLow1 = 0x00010001u;
    static constexpr int kSymBits = 16;
};

// The log table starts at LDS address 0 (the kernel has no other LDS than its one dynamic array; checked at entry), so a
// table byte address IS the LDS address: no per-position base add in front of the ds_read_b64.
typedef const __attribute__((address_space(3))) double lds_cdouble;
__device__ __forceinline__ double lds_f64(uint32_t byte_addr) {
#if HC_ABLATE & 1
    return __hiloint2double(0x3C000000 | (int)(byte_addr & 0xFFFFu), (int)byte_addr);  // a tiny positive number made from the address
#else
    return *(lds_cdouble*)(uintptr_t)byte_addr;
#endif
}

// Exact per-position re-scan (rare: only when the fast sum came out NaN, i.e. the window
// holds an invalid base / quality byte).  Mirrors the reference's order of checks:
// all quality bytes of the window first (:92-101), then position by position (:106-128).
template <typename SymT>
__device__ __noinline__ SubScore score_sub_slow(const SymT* __restrict__ a, const SymT* __restrict__ b, uint32_t L, uint32_t Kp) {
    const uint32_t lg = lut_lg(Kp - 2u);
    SubScore r;
    r.x = -__builtin_inf();
    r.mm = 1;
    r.n = 1;
    r.err = 0;
    const bool wide = sizeof(SymT) == 1 && lg >= 6;
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
        // 8-bit symbols of the smallest table carry a copy of the index's low bits above it (hc_device.h)
        const uint32_t qmask = sizeof(SymT) == 1 ? (1u << lg) - 1u : 0xFFFFu;
        const uint32_t qa = wide ? sa >> 2 : (sa >> 3) & qmask, qb = wide ? sb >> 2 : (sb >> 3) & qmask;
        const uint32_t addr = sizeof(SymT) == 1 ? lut_addr_u8(lg, qa, qb, m) : lut_addr_u16(Kp, qa, qb, m);
        const double t = lds_f64(addr);
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

// The table reads of one half-chunk (8 positions): packed-byte N / mismatch masks and counters, then one LDS address per
// position.  wa/wb point at the half's 32-bit words.  Symbols at or beyond the end of the overlap need no masking: there
// at least one of the two streams is in the N padding behind its sequence (L = min(lenA - pos, lenB); every slot is
// padded with N symbols to a multiple of 16 plus 32 bytes), and a position with an N adds 0.0 and counts as skipped.
// cn4 / cm4: per-byte counters in steps of 4 (the flag sits at bit 2 of its byte); the caller folds them at least every
// 63 words.  The wide encoding keeps plain popcounts (its flags sit at bit 7).
template <typename SymT, int LG>
__device__ __forceinline__ void half_chunk_terms(const uint32_t* wa, const uint32_t* wb, uint32_t Kp, double (&t)[8], uint32_t& cn4,
                                                 uint32_t& cm4) {
    using T = Tr<SymT>;
#pragma unroll
    for (int jj = 0; jj < T::kWords / 2; ++jj) {
        const uint32_t aw = wa[jj];
        const uint32_t bw = wb[jj];
        const uint32_t e = aw ^ bw;
        if (sizeof(SymT) == 1 && LG >= 6) {
            // wide 8-bit encoding: byte = qidx << 2 | base2; qidx < 16 (both top bits clear; LG = 7: qidx < 4, all four top bits clear) is N / invalid.
            // address (hc_device.h: lut_addr_u8) = m << 15 | qa << 9 | x5 << 8 | ((x & 31) ^ (qa >> 1)) << 3, x = qa ^ qb.
            // 23 VALU ops per four positions (round 5's form: 28): the three-input ops written out, the counters kept in one v_bcnt each.
            constexpr uint32_t k80 = 0x80808080u;
            uint32_t ua = (aw << 1) | aw, ub = (bw << 1) | bw;                          // bit 7 of a byte: the symbol is a base
            if (LG == 7) {  // (quality values in indices 4..63: two more of the top bits count)
                ua |= ua << 2;
                ub |= ub << 2;
            }
            const uint32_t nm = __builtin_amdgcn_bitop3_b32(ua, k80, ub, 0x4C);         // ~(ua & ub) & k80
            const uint32_t mk = __builtin_amdgcn_bitop3_b32((e << 7) | (e << 6), k80, nm, 0x40);  // base bits differ, neither is N
            asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(cn4) : "v"(nm));
            asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(cm4) : "v"(mk));
            const uint32_t lo = __builtin_amdgcn_bitop3_b32(e << 1, 0xF8F8F8F8u, aw, 0x48);   // (e << 1 ^ aw) & 0xF8
            const uint32_t z = __builtin_amdgcn_bitop3_b32(e >> 7, 0x01010101u, mk, 0xEA);     // (e >> 7 & 0x01) | mk
            const uint32_t hi = __builtin_amdgcn_bitop3_b32(aw >> 1, 0x7E7E7E7Eu, z, 0xE2);    // bits 6..1 = qa, the others z's
#pragma unroll
            for (int k = 0; k < 4; ++k)
                t[jj * 4 + k] = lds_f64(__builtin_amdgcn_perm(hi, lo, 0x0C0C0000u | ((4u + k) << 8) | (uint32_t)k));
            continue;
        }
        // N flag of either symbol; mismatch flag = (e0 | e1) moved to bit 2 and cleared where the N flag is set.  Two three-input bit ops
        // (v_bitop3_b32, written out: left to itself the compiler forms aw | bw as a value of its own — one more VALU op per four positions)
        constexpr uint32_t kFlag = T::kLow1 << 2;
        const uint32_t nm = __builtin_amdgcn_bitop3_b32(aw, kFlag, bw, 0xC8);                        // (aw | bw) & kFlag
        const uint32_t mk = __builtin_amdgcn_bitop3_b32((e << 1) | (e << 2), kFlag, nm, 0x40);      // u & kFlag & ~nm
        cn4 += nm;
        cm4 += mk;
        if (sizeof(SymT) == 1) {
            // Two bytes per position whose concatenation IS the table's byte address (hc_device.h: lut_addr_u8);
            // symbol byte = qidx << 3 | code, mk holds the mismatch flag at bit 2 of each byte, e = aw ^ bw.
            uint32_t lo, hi;
            if (LG == 5) {         // bits 3-7: qb^qa | bits 8-12: qa, bit 13: m
                lo = e & 0xF8F8F8F8u;
                hi = ((aw >> 3) & 0x1F1F1F1Fu) | (mk << 3);
            } else if (LG == 4) {  // bits 3-6: qb^qa, bit 7: qa0 | bits 8-10: qa>>1, bit 11: m
                lo = (e & 0x78787878u) | ((aw << 4) & 0x80808080u);
                hi = ((aw >> 4) & 0x07070707u) | (mk << 1);
            } else {               // sparse layout, nothing to shift: bits 3-5: qb^qa, bits 6-7: qa&3 (the symbol's own
                                   // bits 6-7) | bit 10: m, bits 11-13: qa
                lo = (e & 0x38383838u) | (aw & 0xC0C0C0C0u);
                hi = (aw & 0x38383838u) | mk;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                t[jj * 4 + k] = lds_f64(__builtin_amdgcn_perm(hi, lo, 0x0C0C0000u | ((4u + k) << 8) | (uint32_t)k));
        } else {
            // two positions per packed 16-bit op: entry = m*T + hi*(hi+1)/2 + lo of the triangular planes (hc_device.h;
            // < 2*4753, fits 16 bits; hi*(hi+1) <= 96*97 does too)
            typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
            const u16x2 qa2 = __builtin_bit_cast(u16x2, aw) >> (unsigned short)3;  // the code bits fall off each lane
            const u16x2 qb2 = __builtin_bit_cast(u16x2, bw) >> (unsigned short)3;
            const u16x2 m2 = __builtin_bit_cast(u16x2, mk >> 2);                    // 0 / 1 per lane
            const u16x2 hi2 = __builtin_elementwise_max(qa2, qb2), lo2 = __builtin_elementwise_min(qa2, qb2);
            const u16x2 one2 = {1, 1};
            const u16x2 tri2 = {(unsigned short)lut_tri(Kp), (unsigned short)lut_tri(Kp)};
            const uint32_t e2 = __builtin_bit_cast(uint32_t, (u16x2)(m2 * tri2 + (((hi2 * (hi2 + one2)) >> (unsigned short)1) + lo2)));
            t[jj * 2 + 0] = lds_f64((e2 << 3) & 0x7FFF8u);
            t[jj * 2 + 1] = lds_f64((e2 >> 13) & 0x7FFF8u);
        }
    }
}

// One sub-overlap with 64-symbol fetch groups: a lane pulls four consecutive 16-byte pieces of each stream back to back
// (one whole 64-byte line of an aligned stream), so a line is fetched into L1 once and consumed at once instead of being
// re-requested by four separate loop iterations that other waves' lines evict in between.  The next group is in flight
// while the current one is scored.
template <typename SymT, int LG, int G /* 16-symbol chunks per group */>
__device__ __forceinline__ void score_sub_wide(const SymT* __restrict__ a, const SymT* __restrict__ b, uint32_t L, uint32_t fatal,
                                               uint32_t Kp, SubScore& out) {
    using T = Tr<SymT>;
    constexpr bool kPacked = !(sizeof(SymT) == 1 && LG >= 6);  // per-byte counters (half_chunk_terms)
    out.x = -__builtin_inf();
    out.mm = 1;
    out.n = 1;
    out.err = fatal;
    if (L == 0) return;
    const uint32_t nch = (L + 15u) >> 4;
    double S = 0.0;
    uint32_t skipped = 0, cm = 0;
    // two register sets (x / y) take turns: while one group is scored the next one is in flight into the other set
    uint32_t xa[G][T::kWords], xb[G][T::kWords], ya[G][T::kWords], yb[G][T::kWords];
    auto fetch = [&](uint32_t (&ra)[G][T::kWords], uint32_t (&rb)[G][T::kWords], uint32_t cbase) {
#pragma unroll
        for (int q = 0; q < G; ++q)
            if (cbase + q < nch) {
                __builtin_memcpy(ra[q], a + 16u * (cbase + q), sizeof(ra[q]));
                __builtin_memcpy(rb[q], b + 16u * (cbase + q), sizeof(rb[q]));
            }
    };
    auto score_group = [&](const uint32_t (&ra)[G][T::kWords], const uint32_t (&rb)[G][T::kWords], uint32_t cbase) {
        uint32_t cn4 = 0, cm4 = 0;
#pragma unroll
        for (int q = 0; q < G; ++q)
            if (cbase + q < nch) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    double t[8];
                    half_chunk_terms<SymT, LG>(ra[q] + h * (T::kWords / 2), rb[q] + h * (T::kWords / 2), Kp, t, cn4, cm4);
#pragma unroll
                    for (int k = 0; k < 8; ++k) S += t[k];  // :119, strictly in position order
                }
            }
        if (kPacked) {  // a byte (16-bit lane) of cn4 / cm4 holds at most 4 * 4 * G <= 64 here
            skipped = __builtin_amdgcn_sad_u8(cn4, 0u, skipped);
            cm = __builtin_amdgcn_sad_u8(cm4, 0u, cm);
        } else {
            skipped += cn4;
            cm += cm4;
        }
    };
    // `arrived`: loads complete in order, so a (pretended) use of the set's last register makes the compiler wait for the
    // set HERE, before the other set's loads are issued — otherwise it waits at the first real use, behind those loads,
    // and (the loads being conditional) for them too.
    auto arrived = [](uint32_t (&rb)[G][T::kWords]) { asm volatile("" : "+v"(rb[G - 1][T::kWords - 1])); };
    fetch(xa, xb, 0);
    for (uint32_t c0 = 0; c0 < nch; c0 += 2 * G) {
        arrived(xb);
        fetch(ya, yb, c0 + G);
        score_group(xa, xb, c0);
        arrived(yb);
        fetch(xa, xb, c0 + 2 * G);
        score_group(ya, yb, c0 + G);
    }
    if (kPacked) {
        skipped >>= 2;
        cm >>= 2;
    }
    if (S != S) {  // an invalid symbol inside the window (or next to it, in the last chunk)
        const SubScore e = score_sub_slow<SymT>(a, b, L, Kp);
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

// ---------------------------------------------------------------------------
// Cooperative fetch (the memory side of the kernel, measured alone: tools/experiments/gather_shapes.hip — one lane
// fetching its own candidate's windows 16 bytes at a time is bound by the L1's tag look-ups, one per lane and load; four
// lanes fetching ONE candidate's 64 contiguous bytes need a quarter of them: 1.67x the candidates per second).
//
// The 64 candidates of a wave advance together in 64-byte rows (64 symbols of 8 bits, 32 of 16).  Lane l OWNS candidate
// l (it adds that candidate's terms, in position order, as before) and is also a LOADER: in load j (0..3) it fetches
// the 16-byte piece (l & 3) ^ ((l >> 4) & 3) of the current row of candidate 16 j + (l >> 2), so a quad reads one
// candidate's 64 contiguous bytes.  The pieces go through a per-wave LDS image of 64 rows x 64 bytes — written lane-linear
// (1 KiB per instruction), read by the owner one row per lane; the XOR of the piece number with (row >> 2) & 3 puts the 16
// rows a 128-bit LDS read serves in one pass into 16 different bank groups.  The A rows and then the B rows of a step pass
// through the same image (the owner keeps its A row in registers); the next step's pieces are in flight into registers
// meanwhile.  Loads go through a buffer descriptor of the store: a piece that is not needed (beyond the candidate's
// window) gets the first out-of-range offset, which the range check drops without a memory access — no divergent control flow
// around the loads, so the compiler counts them exactly (s_waitcnt vmcnt(N)).  Wave-synchronous: no workgroup barrier;
// LDS executes one wave's instructions in order.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
__device__ __forceinline__ void lds_store128(uint32_t addr, u32x4 v) { *(lds_u32x4*)(uintptr_t)addr = v; }
__device__ __forceinline__ u32x4 lds_load128(uint32_t addr) { return *(const lds_u32x4*)(uintptr_t)addr; }
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) u32x2 lds_u32x2;
__device__ __forceinline__ void lds_store64(uint32_t addr, uint32_t a, uint32_t b) { *(lds_u32x2*)(uintptr_t)addr = u32x2{a, b}; }
__device__ __forceinline__ void lds_load64(uint32_t addr, uint32_t& a, uint32_t& b) {
    const u32x2 v = *(const lds_u32x2*)(uintptr_t)addr;
    a = v[0];
    b = v[1];
}
__device__ __forceinline__ uint32_t lds_load32(uint32_t addr) { return *(const lds_u32*)(uintptr_t)addr; }
__device__ __forceinline__ uint32_t lds_add_rtn(uint32_t addr, uint32_t v) {  // ds_add_rtn_u32
    return __hip_atomic_fetch_add((lds_u32*)(uintptr_t)addr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void wave_lds_order() {  // keeps the compiler from reordering the image's stores and loads
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

constexpr uint32_t kStageBytesPerWave = 64u * 64u;
// LDS of the cooperative kernel: [table][scratch: 32 words][pad to 1 KiB][one image per wave]
// The wide 8-bit table's LDS-DMA form (768 lanes: 64 KiB + 12 x 8 KiB = all 160 KiB) keeps the scratch words INSIDE the table, in a row no
// symbol addresses (quality index 3 is never encoded, hc_device.h: row 3 of either plane is a hole of 512 bytes; LG = 6 leaves 3..15 free).
constexpr uint32_t kWideDmaLanes = 768, kWideDmaScratch = 3u << 9;  // row 3 of the match plane (index 3: dealt by neither wide encoding)
__host__ __device__ inline uint32_t coop_stage_base(uint32_t lut_bytes, uint32_t wg) {
    return (lut_bytes == 65536u && wg == kWideDmaLanes) ? 65536u : (lut_bytes + 128u + 1023u) & ~1023u;
}

// What a sub-overlap's walk leaves behind -> its result (x = (1/n) S, mismatches, n), or the exact re-scan when the sum came
// out NaN (an invalid symbol inside the window, or next to it in the last chunk).
// inv_n: StoreView::inv_n — 1.0 / n for every n a sub-overlap of this read set can count (built on the host: the same IEEE quotient the
// device's division gives, :137; round 5: the division was ~15 VALU instructions per sub-overlap)
template <typename SymT>
__device__ __forceinline__ void finish_sub(const SymT* __restrict__ sym, uint32_t offA, uint32_t offB, uint32_t L, uint32_t fatal, uint32_t Kp, double S,
                                           uint32_t skipped, uint32_t cm, bool packed, const double* __restrict__ inv_n, SubScore& out) {
    out.x = -__builtin_inf();
    out.mm = 1;
    out.n = 1;
    out.err = fatal;
    if (L == 0) return;
    if (packed) {
        skipped >>= 2;
        cm >>= 2;
    }
    if (S != S) {
        const SubScore e = score_sub_slow<SymT>((const SymT*)((const char*)sym + offA), (const SymT*)((const char*)sym + offB), L, Kp);
        out.x = e.x;
        out.mm = e.mm;
        out.n = e.n;
        out.err |= e.err;
        return;
    }
    if (S == __builtin_inf()) return;
    const uint32_t cn = 16u * ((L + 15u) >> 4) - skipped;
    if (cn == 0) return;
    out.x = inv_n[cn] * S;  // cn <= L + 15 <= the longest sequence + 15 < the table's length
    out.mm = cm;
    out.n = cn;
}

// One sub-overlap of each of the wave's 64 candidates.  Called by all 64 lanes; a lane without one passes L = 0.
// offA / offB: byte offsets of the window starts in the store; L: positions (sub_positions()).
template <typename SymT, int LG, int DEPTH = 1>
__device__ __forceinline__ void score_sub_coop(__amdgpu_buffer_rsrc_t rsrc, uint32_t oob, const SymT* __restrict__ sym, uint32_t stage, uint32_t offA,
                                               uint32_t offB, uint32_t L, uint32_t fatal, uint32_t Kp, const double* __restrict__ inv_n, SubScore& out) {
    using T = Tr<SymT>;
    constexpr uint32_t kSymB = sizeof(SymT);
    constexpr int kChunks = 4 / kSymB;            // 16-symbol chunks per 64-byte row
    constexpr uint32_t kChunkB = 16u * kSymB;     // bytes per chunk
    constexpr bool kPacked = !(sizeof(SymT) == 1 && LG >= 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t piece = (lane & 3u) ^ ((lane >> 4) & 3u);   // of the row, as a loader
    const uint32_t Lb = L * kSymB;
    uint32_t la[4], lb[4], lim[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int src = 16 * j + (int)(lane >> 2);
        la[j] = (uint32_t)__shfl((int)offA, src, 64) + 16u * piece;
        lb[j] = (uint32_t)__shfl((int)offB, src, 64) + 16u * piece;
        // a piece is wanted while the chunk it belongs to starts inside the window
        const uint32_t end = (uint32_t)__shfl((int)Lb, src, 64), cs = (16u * piece) & ~(kChunkB - 1u);
        lim[j] = end > cs ? end - cs : 0u;
    }
    const uint32_t wr = stage + lane * 16u;                                  // + j KiB: lane-linear
    const uint32_t rd = stage + lane * 64u + (((lane >> 2) & 3u) << 4);    // ^ piece << 4
    if constexpr (DEPTH == 0) {
        // LDS-DMA form (buffer_load_dwordx4 ... lds, gfx950): the pieces go from memory straight into the wave's image — A rows
        // into its first 4 KiB, B rows into the second — with no destination registers and no ds_write_b128; the image is
        // lane-linear per instruction (M0 = wave-uniform base, lane l writes 16 bytes at base + 16 l), the XOR swizzle sits on
        // the SOURCE side (which piece a lane fetches), exactly as in the register-staged form.  A step: wait for its pieces,
        // every owner copies its two rows into registers, the next step's pieces are requested into the same image, the rows
        // are scored from the registers.
        double S = 0.0;
        uint32_t skipped = 0, cm = 0;
        const uint32_t rdA = stage + lane * 64u + (((lane >> 2) & 3u) << 4), rdB = rdA + 4096u;
        auto fetch_dma = [&](uint32_t at) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool on = at < lim[j];
                // (the step's offset `at` is wave-uniform: it travels as the instruction's scalar offset, the lane's part is la / lb or the
                // first out-of-range offset — no per-lane add per load; the range check takes the scalar offset into account)
#if !(HC_ABLATE & 8)  // (ablation 8: no A rows are fetched — the image keeps what it held)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(uintptr_t)(stage + 1024u * j), 16,
                                                         on ? la[j] : oob, at, 0, HC_COOP_AUX_A);
#endif
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(uintptr_t)(stage + 4096u + 1024u * j), 16,
                                                         on ? lb[j] : oob, at, 0, HC_COOP_AUX_B);
            }
        };
#if HC_ABLATE & 16
        // (ablation, garbage results: what would a SECOND step of partner rows in flight buy?  Only the B rows are fetched — as if the A
        // side came from a wave-shared copy of the common read — and the two 4 KiB halves of the wave's image take turns: the rows of step
        // k + 2 are requested as soon as step k's have been copied into registers)
        if (__ballot(Lb != 0u) != 0ull) {
            auto fetch_b = [&](uint32_t at, uint32_t half) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool on = at < lim[j];
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(uintptr_t)(stage + 4096u * half + 1024u * j), 16,
                                                             on ? lb[j] : oob, at, 0, HC_COOP_AUX_B);
                }
            };
            fetch_b(0, 0);
            fetch_b(64u, 1);
            uint32_t half = 0;
            for (uint32_t at = 0;; at += 64u, half ^= 1u) {
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // the older of the two steps in flight has landed
                wave_lds_order();
                u32x4 xa[4], xb[4];
                const uint32_t rdH = rdA + 4096u * half;
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    xb[p] = lds_load128(rdH ^ ((uint32_t)p << 4));
                    xa[p] = xb[p];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xa[0]), "+v"(xa[1]), "+v"(xa[2]), "+v"(xa[3]), "+v"(xb[0]), "+v"(xb[1]), "+v"(xb[2]), "+v"(xb[3])::"memory");
                wave_lds_order();
                const bool more = __ballot(at + 64u < Lb) != 0ull;
                fetch_b(at + 128u, half);  // (beyond every window: dropped by the range check, still counted by vmcnt)
                uint32_t cn4 = 0, cm4 = 0;
#pragma unroll
                for (int q = 0; q < kChunks; ++q)
                    if (at + kChunkB * q < Lb) {
                        uint32_t wa[T::kWords], wb[T::kWords];
#pragma unroll
                        for (int sw = 0; sw < (int)kSymB; ++sw) {
                            const int p = q * (int)kSymB + sw;
#pragma unroll
                            for (int w = 0; w < 4; ++w) {
                                wa[4 * sw + w] = xa[p][w] ^ 0x08080808u;
                                wb[4 * sw + w] = xb[p][w];
                            }
                        }
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            double t[8];
                            half_chunk_terms<SymT, LG>(wa + h * (T::kWords / 2), wb + h * (T::kWords / 2), Kp, t, cn4, cm4);
#pragma unroll
                            for (int k = 0; k < 8; ++k) S += t[k];
                        }
                    }
                skipped = __builtin_amdgcn_sad_u8(cn4, 0u, skipped);
                cm = __builtin_amdgcn_sad_u8(cm4, 0u, cm);
                if (!more) break;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        finish_sub<SymT>(sym, offA, offB, L, fatal, Kp, S == S ? S : 0.0, skipped, cm, kPacked, inv_n, out);
        return;
#endif
        if (__ballot(Lb != 0u) != 0ull) {  // wave-uniform
            fetch_dma(0);
            for (uint32_t at = 0;; at += 64u) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this step's pieces have landed
                wave_lds_order();
                u32x4 xa[4], xb[4];
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    xb[p] = lds_load128(rdB ^ ((uint32_t)p << 4));
#if HC_ABLATE & 8
                    xa[p] = xb[p] ^ 0x08080808u;  // (valid symbols without an A row)
#else
                    xa[p] = lds_load128(rdA ^ ((uint32_t)p << 4));
#endif
                }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xa[0]), "+v"(xa[1]), "+v"(xa[2]), "+v"(xa[3]), "+v"(xb[0]), "+v"(xb[1]), "+v"(xb[2]), "+v"(xb[3])::"memory");
                wave_lds_order();
                const bool more = __ballot(at + 64u < Lb) != 0ull;
                if (more) fetch_dma(at + 64u);  // into the image every owner has just emptied
                uint32_t cn4 = 0, cm4 = 0;
#pragma unroll
                for (int q = 0; q < kChunks; ++q)
                    if (at + kChunkB * q < Lb) {
                        uint32_t wa[T::kWords], wb[T::kWords];
#pragma unroll
                        for (int sw = 0; sw < (int)kSymB; ++sw) {
                            const int p = q * (int)kSymB + sw;
#pragma unroll
                            for (int w = 0; w < 4; ++w) {
                                wa[4 * sw + w] = xa[p][w];
                                wb[4 * sw + w] = xb[p][w];
                            }
                        }
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            double t[8];
                            half_chunk_terms<SymT, LG>(wa + h * (T::kWords / 2), wb + h * (T::kWords / 2), Kp, t, cn4, cm4);
#pragma unroll
                            for (int k = 0; k < 8; ++k) S += t[k];  // :119, strictly in position order
                        }
                    }
                if (kPacked) {
                    skipped = __builtin_amdgcn_sad_u8(cn4, 0u, skipped);
                    cm = __builtin_amdgcn_sad_u8(cm4, 0u, cm);
                } else {
                    skipped += cn4;
                    cm += cm4;
                }
                if (!more) break;
            }
        }
        finish_sub<SymT>(sym, offA, offB, L, fatal, Kp, S, skipped, cm, kPacked, inv_n, out);
        return;
    }
    // DEPTH register sets of pieces in flight: set s holds the pieces of the step it is consumed in and is refilled, right
    // after its pieces went to LDS, with those of DEPTH steps further on
    u32x4 nA[DEPTH == 0 ? 1 : DEPTH][4], nB[DEPTH == 0 ? 1 : DEPTH][4];
    auto fetch = [&](u32x4 (&sA)[4], u32x4 (&sB)[4], uint32_t at) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool on = at < lim[j];
#if HC_ABLATE & 2
            sA[j] = u32x4{0x01010101u + (on ? (la[j] + at) & 0x02020202u : 0u), 0x01010101u, 0x09090909u, 0x11111111u};  // valid symbols only
            sB[j] = u32x4{0x09090909u + (on ? (lb[j] + at) & 0x02020202u : 0u), 0x09090909u, 0x01010101u, 0x11111111u};
#else
            sA[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, on ? la[j] : oob, at, HC_COOP_AUX_A);
            sB[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, on ? lb[j] : oob, at, HC_COOP_AUX_B);
#endif
        }
    };
    double S = 0.0;
    uint32_t skipped = 0, cm = 0;
    auto step = [&](u32x4 (&sA)[4], u32x4 (&sB)[4], uint32_t at) {
        u32x4 xa[4];
#if HC_ABLATE & 4
        u32x4 xb_[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            xa[p] = sA[p];
            xb_[p] = sB[p];
        }
#else
#pragma unroll
        for (int j = 0; j < 4; ++j) lds_store128(wr + 1024u * j, sA[j]);
        wave_lds_order();
#pragma unroll
        for (int p = 0; p < 4; ++p) xa[p] = lds_load128(rd ^ ((uint32_t)p << 4));
        wave_lds_order();
#pragma unroll
        for (int j = 0; j < 4; ++j) lds_store128(wr + 1024u * j, sB[j]);
        wave_lds_order();
#endif
        fetch(sA, sB, at + 64u * DEPTH);
        uint32_t cn4 = 0, cm4 = 0;
#pragma unroll
        for (int q = 0; q < kChunks; ++q)
            if (at + kChunkB * q < Lb) {
                uint32_t wa[T::kWords], wb[T::kWords];
#pragma unroll
                for (int s = 0; s < (int)kSymB; ++s) {
                    const int p = q * (int)kSymB + s;
#if HC_ABLATE & 4
                    const u32x4 vb = xb_[p];
#else
                    const u32x4 vb = lds_load128(rd ^ ((uint32_t)p << 4));
#endif
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        wa[4 * s + w] = xa[p][w];
                        wb[4 * s + w] = vb[w];
                    }
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    double t[8];
                    half_chunk_terms<SymT, LG>(wa + h * (T::kWords / 2), wb + h * (T::kWords / 2), Kp, t, cn4, cm4);
#pragma unroll
                    for (int k = 0; k < 8; ++k) S += t[k];  // :119, strictly in position order
                }
            }
        if (kPacked) {
            skipped = __builtin_amdgcn_sad_u8(cn4, 0u, skipped);
            cm = __builtin_amdgcn_sad_u8(cm4, 0u, cm);
        } else {
            skipped += cn4;
            cm += cm4;
        }
        wave_lds_order();  // the owners' reads of this step's B rows stay in front of the next step's stores
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) fetch(nA[d], nB[d], 64u * d);
    for (uint32_t at = 0; __ballot(at < Lb) != 0ull; at += 64u * DEPTH) {  // wave-uniform
        step(nA[0], nB[0], at);
        if (DEPTH == 2) {
            if (__ballot(at + 64u < Lb) == 0ull) break;
            step(nA[DEPTH - 1], nB[DEPTH - 1], at + 64u);
        }
    }
    finish_sub<SymT>(sym, offA, offB, L, fatal, Kp, S, skipped, cm, kPacked, inv_n, out);
}

// Result records are written once and read by nobody on the device: streamed past the caches, so they do not push the
// read store out of L2 / the Infinity Cache (the candidate records are read the same way, hc_resolve.h).
// (STREAM = false: the bucketed launch writes a tile's records in rank order, i.e. scattered over the tile's 96 KiB — plain stores
// let the L2 put the lines together before they leave)
template <bool STREAM = true>
__device__ __forceinline__ void store_result(hc_result_rec* __restrict__ out, uint64_t i, const hc_result_rec& r) {
    unsigned long long* p = (unsigned long long*)(out + i);
    if (STREAM) {
        __builtin_nontemporal_store((unsigned long long)__double_as_longlong(r.x1), p);
        __builtin_nontemporal_store((unsigned long long)__double_as_longlong(r.x2), p + 1);
        __builtin_nontemporal_store(((unsigned long long)r.n_cls << 32) | r.mm, p + 2);
    } else {
        p[0] = (unsigned long long)__double_as_longlong(r.x1);
        p[1] = (unsigned long long)__double_as_longlong(r.x2);
        p[2] = ((unsigned long long)r.n_cls << 32) | r.mm;
    }
}

// exp(x) > T  in x-space: 1 pass, 0 fail, 2 ambiguous
__device__ __forceinline__ uint32_t band_test(double x, const Band& b) { return x > b.hi ? 1u : (x <= b.lo ? 0u : 2u); }

// The tail of compute_overlap + the 3-way class of process_overlaps for one candidate whose sub-overlap
// results are known: mismatch_rate = max of the two (:254), class in x-space (:404-413), result record.
template <bool STREAM = true>
__device__ __forceinline__ hc_result_rec classify_and_store(const ScoreParams& prm, int ns, const SubScore& s1, const SubScore& s2,
                                                            uint64_t i, hc_result_rec* __restrict__ out) {
    hc_result_rec res;
    // mismatch_rate = float(mismatch_count)/total_len (:132); std::max over the two (:254): the sub-overlap with the larger
    // rate supplies (mm, n).  Which one that is needs no division: for mm < 2^24 (exact as a float) and n < 2^26 two
    // different fractions mm/n <= 1 lie more than 2^-52 apart — two ulps at least — so their correctly rounded quotients
    // compare as the fractions do, i.e. as the cross products; equal fractions give equal quotients.
    uint32_t mm = s1.mm, nn = s1.n;
    if (ns == 2) {
        bool second;
        if ((s1.mm | s2.mm) < (1u << 24) && (s1.n | s2.n) < (1u << 26)) second = (uint64_t)s1.mm * s2.n < (uint64_t)s2.mm * s1.n;
        else second = (double)(float)s1.mm / (double)s1.n < (double)(float)s2.mm / (double)s2.n;
        if (second) {
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
    // mismatch_rate <= --merge_contigs; against the default 0 that is "no mismatch" (the rate is never negative)
    else if (prm.merge_contigs == 0.0 ? mm == 0u : (double)(float)mm / (double)nn <= prm.merge_contigs) cls = HC_CLS_EDGE_MC;
    else if (o == 1) cls = HC_CLS_NONEDGE;
    else if (o == 2) cls = HC_CLS_AMBIG;
    else cls = HC_CLS_DROP;
    res.x1 = s1.x;
    res.x2 = s2.x;
    res.mm = mm;
    res.n_cls = (nn & 0x0FFFFFFFu) | (cls << 28);
    store_result<STREAM>(out, i, res);
    return res;
}

// One candidate, one lane: score its sub-overlaps and write the result record.
// G = 16-symbol chunks per fetch group (4: 64-symbol groups for short reads; 2: 32-symbol groups for contigs).
template <typename SymT, int G, int LG>
__device__ __forceinline__ hc_result_rec score_candidate(const ScoreParams& prm, const SymT* __restrict__ sym, uint32_t Kp, int ns,
                                                         const Sub& sub0, const Sub& sub1, uint64_t i,
                                                         hc_result_rec* __restrict__ out) {
    if (ns <= 0) {  // malformed record: an error; "skip" record (a line dropped before scoring): a dropped result
        hc_result_rec res;
        res.x1 = -__builtin_inf();
        res.x2 = __builtin_nan("");
        res.mm = 1;
        res.n_cls = 1u | ((ns == 0 ? HC_CLS_ERROR : HC_CLS_DROP) << 28);
        store_result(out, i, res);
        return res;
    }
    SubScore s1, s2;
    s2.x = __builtin_nan("");
    s2.mm = 0;
    s2.n = 1;
    s2.err = 0;
    // the second sub-overlap waits as two pointers and a length while the first is scored
    const SymT* a1 = sym + sub1.offA + sub1.pos;
    const SymT* b1 = sym + sub1.offB;
    const uint32_t L1 = ns == 2 ? sub_positions(sub1, prm.min_read_len) | (sub1.fatal << 31) : 0u;
    score_sub_wide<SymT, LG, G>(sym + sub0.offA + sub0.pos, sym + sub0.offB, sub_positions(sub0, prm.min_read_len), sub0.fatal, Kp, s1);
    if (ns == 2) score_sub_wide<SymT, LG, G>(a1, b1, L1 & 0x7FFFFFFFu, L1 >> 31, Kp, s2);
    return classify_and_store(prm, ns, s1, s2, i, out);
}

// What the stage and the multi-GPU collection need of a batch, produced by the scoring kernel itself: every record that
// is not dropped is appended, tagged with base_index + its position in the batch, to `rows`; *count is the number of
// appended rows (it may exceed cap: overflow, nothing is written beyond cap).  One atomic per workgroup and iteration
// (same-address atomics serialise in L2: one per wave cost 0.1 ms per 2 M candidates); the rows arrive in no
// particular order (they carry their index).  `rows` may be page-locked host memory mapped into the device's address
// space (the stage: the few per cent of records that survive land where the host reads them, with no copy in
// between); the counter always lives in device memory.
struct RowSink {
    hc_gather_row* rows;  // nullptr: no collection
    unsigned long long* count;
    uint64_t cap;
    uint64_t base_index;
    // the device parser's stage: the parsed line of every appended row travels with it (lines_out[pos] = lines_in[i])
    const hc_line_rec* lines_in;
    hc_line_rec* lines_out;
    // Segment mode (the cooperative kernel's plain launches; hc_score_pack_device): every workgroup appends into a segment of its
    // own — seg_rows rows at seg_buf + blockIdx.x * seg_rows, its place taken from a counter in LDS, so the waves neither meet at a
    // barrier nor share a global atomic — and leaves its count in seg_count[blockIdx.x]; sink_compact_kernel then moves the
    // segments into `rows` back to back and writes *count.  nullptr: rows are appended directly (one global atomic per workgroup
    // and iteration behind two barriers, or one per wave in the bucketed launch).
    // A row that finds its workgroup's segment full is not lost (round 4, ADVICE: one workgroup's share of the kept rows is not bounded
    // by any multiple of the mean — bucketed launches pull pieces at their own pace, and a file may keep all its rows in one stretch):
    // it goes to the SPILL area behind the segments (spill_cap rows, its place from one global counter per wave — the slow path, taken
    // by skewed launches only), and sink_compact_kernel appends the spilled rows behind the segments' rows.  *count is then exactly
    // the number of kept rows, above cap only when there were more kept rows than cap.
    hc_gather_row* seg_buf;
    uint32_t* seg_count;
    uint32_t seg_rows;
    uint32_t spill_cap;
    hc_gather_row* spill_buf;
    uint32_t* spill_count;  // zero when the launch starts (sink_compact_kernel re-arms the counter of the launch after the next)
    // hc_comm_gate_device (the multi-GPU step): every workgroup of a cooperative launch adds one here when it starts, so that a gate
    // kernel on the exchange's stream can hold the collective back until this launch's workgroups sit on their CUs.  nullptr: not counted.
    unsigned long long* started;
};

// Called by ALL lanes of the workgroup (uniform control flow; `valid` = this lane scored candidate i).
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
        if (total) base = atomicAdd(sink.count, (unsigned long long)total);
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
            sink.rows[pos] = r;
            if (sink.lines_in) {  // 48 bytes as three 16-byte pieces
                const uint4* a = (const uint4*)(sink.lines_in + i);
                uint4* b = (uint4*)(sink.lines_out + pos);
                b[0] = a[0];
                b[1] = a[1];
                b[2] = a[2];
            }
        }
    }
    __syncthreads();  // lds4 is reused by the next iteration
}

// The log table into LDS.  The sparse layout of the smallest table (hc_device.h) has 128 live entries in 16 KiB of address
// space: only those are copied (a workgroup that scores one block of 256 candidates would otherwise move 16 KiB for them).
template <typename SymT, int LG>
__device__ __forceinline__ void load_log_table(double* lut_s, const double* __restrict__ lut_g, uint32_t lut_n, uint32_t tid, uint32_t n_threads) {
    if (sizeof(SymT) == 1 && LG == 3) {
        for (uint32_t e = tid; e < 128u; e += n_threads) {
            const uint32_t at = lut_addr_u8(3, e >> 4, (e >> 1) & 7u, e & 1u) >> 3;
            lut_s[at] = lut_g[at];
        }
    } else {
        for (uint32_t i = tid; i < lut_n; i += n_threads) lut_s[i] = lut_g[i];
    }
}

// Block-local length balancing (read sets with mixed sequence lengths: contigs next to reads).  A wave runs as long as
// its longest lane, a workgroup iteration as long as its longest wave; with the log-uniform 150..6 000 bp contigs of
// BASELINE config 5 the mean lane is busy 29 % of that time.  One iteration of a workgroup takes kBalItems x blockDim.x
// candidates, ranks them by overlap length (LDS counting sort over quarter-octave classes, longest first) and deals them
// like cards, there and back: lane t gets ranks t, 2W-1-t, 2W+t, ... (W = blockDim.x) and scores them one after the
// other.  Neighbouring lanes then hold overlaps of neighbouring rank at any moment (little divergence inside a wave); with
// several items per lane every lane's items also add up to about the same length — but the wider window costs more in
// lost read sharing than the even finish gains (kBalItems below), so it is one item per lane: the candidates stay inside
// their 256-candidate block.
// Called by every lane of the workgroup (barriers inside).  bal: 128 + kBalItems * blockDim.x + 32 words of LDS scratch
// ([0..127] class histogram / offsets, then the order array, then 2 x 16 words of reduction).
constexpr int kBalItems = 1;  // measured on C5 (2 M contig overlaps): 1 -> 0.975 ms, 2 -> 1.036, 4 -> 1.167: the wider the window, the less neighbouring lanes share reads
template <typename SymT>
__device__ __forceinline__ void balanced_slots(const StoreView& st, const ScoreParams& prm, const void* __restrict__ in, uint64_t n,
                                               const uint32_t* __restrict__ perm, uint32_t fmt, uint64_t block_base, uint32_t* bal,
                                               uint64_t (&slots)[kBalItems]) {
    const uint32_t tid = threadIdx.x, W = blockDim.x;
    const uint32_t red = 128u + kBalItems * W;  // behind the order array
    // phase 1: the length classes of the candidates in this lane's own slots (their state is dead before phase 2)
    uint32_t chunks[kBalItems];
    uint32_t wmax = 0, wsum = 0;
#pragma unroll
    for (int k = 0; k < kBalItems; ++k) {
        const uint64_t slot = block_base + (uint64_t)k * W + tid;
        slots[k] = slot;
        uint32_t c = 0;
        if (slot < n) {
            const Cand rec = load_cand(in, perm ? (uint64_t)perm[slot] : slot, fmt);
            Sub s0, s1;
            const int ns = resolve<(int)sizeof(SymT)>(st, rec, s0, s1);
            if (ns >= 1) c = (sub_positions(s0, prm.min_read_len) + 15u) >> 4;
            if (ns == 2) c += (sub_positions(s1, prm.min_read_len) + 15u) >> 4;  // the lane scores both, one after the other
        }
        chunks[k] = c;
        wmax = c > wmax ? c : wmax;
        wsum += c;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t m = (uint32_t)__shfl_xor((int)wmax, o, 64);
        wmax = m > wmax ? m : wmax;
        wsum += (uint32_t)__shfl_xor((int)wsum, o, 64);
    }
    if ((tid & 63u) == 0) {
        bal[red + (tid >> 6)] = wmax;
        bal[red + 16 + (tid >> 6)] = wsum;
    }
    if (tid < 128) bal[tid] = 0;
    __syncthreads();
    uint32_t bmax = 0, bsum = 0;
    for (uint32_t w = 0; w < (W >> 6); ++w) {
        bmax = bal[red + w] > bmax ? bal[red + w] : bmax;
        bsum += bal[red + 16 + w];
    }
    // worth it when the longest overlap is at least twice the block's mean and there is real work to balance
    if (bmax >= 16u && (uint64_t)bmax * kBalItems * W > 2ull * bsum) {
        uint32_t cls[kBalItems];
#pragma unroll
        for (int k = 0; k < kBalItems; ++k) {
            uint32_t c = 0;  // quarter-octave class of the chunk count
            if (chunks[k] > 1) {
                const uint32_t lg = 31u - (uint32_t)__builtin_clz(chunks[k]);
                c = lg * 4u + (lg >= 2 ? (chunks[k] >> (lg - 2)) & 3u : 0u);
            }
            cls[k] = c > 127u ? 127u : c;
            atomicAdd(&bal[cls[k]], 1u);
        }
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
#pragma unroll
        for (int k = 0; k < kBalItems; ++k) bal[128 + atomicAdd(&bal[cls[k]], 1u)] = (uint32_t)k * W + tid;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kBalItems; ++k) {
            const uint32_t rank = (k & 1) ? (uint32_t)(k + 1) * W - 1u - tid : (uint32_t)k * W + tid;  // there and back
            slots[k] = block_base + bal[128 + rank];
        }
    }
    __syncthreads();  // bal is reused below and by the next iteration
}

// The scoring kernel.  LG: log2 of the 8-bit-symbol table dimension (3..6; ignored for 16-bit symbols).  BAL:
// block-local length balancing (balanced_slots).  Workgroup-uniform loop: an iteration of a workgroup handles the
// candidates [block_base, block_base + blockDim.x) — kBalItems times as many with BAL.
template <typename SymT, int G, int LG, bool BAL>
__device__ __forceinline__ void score_kernel_body(const StoreView& st, const ScoreParams& prm, const double* __restrict__ lut_g,
                                                  const void* __restrict__ in, uint64_t n, hc_result_rec* __restrict__ out,
                                                  const uint32_t* __restrict__ perm, const RowSink& sink) {
    if (prm.n_dev) {  // the number of records is only known on the device (lines of a block of text): at most n of them
        const uint64_t nd = *prm.n_dev;
        n = nd < n ? nd : n;
    }
    extern __shared__ __attribute__((aligned(16))) double lut_s[];
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lut_s != 0u) __builtin_trap();  // lds_f64 relies on it
    const uint32_t lut_n = st.lut_bytes >> 3;
    load_log_table<SymT, LG>(lut_s, lut_g, lut_n, threadIdx.x, blockDim.x);
    __syncthreads();

    const SymT* sym = (const SymT*)st.sym;
    const uint32_t Kp = st.K + 2u;
    const uint32_t fmt = prm.rec_fmt;
    // scratch behind the table: length balancing (balanced_slot) / [0..17] wave offsets of the row append (the two never
    // overlap in time: barriers in between)
    uint32_t* bal = (uint32_t*)(lut_s + lut_n);
    const uint32_t tid = threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    // with a permutation, slot s scores candidate perm[s] (neighbouring lanes share reads) and writes its
    // record back to the candidate's own position: out[i] <-> in[i] always holds
    constexpr int kItems = BAL ? kBalItems : 1;  // candidates per lane and iteration
    for (uint64_t block_base = (uint64_t)blockIdx.x * blockDim.x * kItems; block_base < n; block_base += stride * kItems) {
        uint64_t slots[kBalItems];
        slots[0] = block_base + tid;
        if (BAL) balanced_slots<SymT>(st, prm, in, n, perm, fmt, block_base, bal, slots);
#pragma unroll 1
        for (int k = 0; k < kItems; ++k) {
            const uint64_t slot = slots[k];
            // score the candidate of the (possibly reassigned) slot
            hc_result_rec res;
            res.n_cls = 0;
            uint64_t i = 0;
            if (slot < n) {
                i = perm ? (uint64_t)perm[slot] : slot;
                const Cand rec = load_cand(in, i, fmt);
                Sub sub0{}, sub1{};
                const int ns = resolve<(int)sizeof(SymT)>(st, rec, sub0, sub1);
                res = score_candidate<SymT, G, LG>(prm, sym, Kp, ns, sub0, sub1, i, out);
            }
            if (sink.rows) append_rows_block(sink, slot < n, res, i, bal);  // kernel-argument-uniform branch
        }
    }
}

// The row append of ONE wave (the length-bucketed launch: its waves take their work from a queue, each at its own pace, so
// there is no workgroup-wide moment to share an atomic).  Called by all 64 lanes.
__device__ __forceinline__ void append_rows_wave(const RowSink& sink, bool valid, const hc_result_rec& res, uint64_t i) {
    const bool keep = valid && (res.n_cls >> 28) != HC_CLS_DROP;
    const uint64_t m = __ballot(keep);
    if (m == 0ull) return;  // wave-uniform
    const uint32_t lane = threadIdx.x & 63u;
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(sink.count, (unsigned long long)__popcll(m));
    base = __shfl(base, 0, 64);
    if (keep) {
        const uint64_t pos = base + (uint64_t)__popcll(m & ((1ull << lane) - 1ull));
        if (pos < sink.cap) {
            hc_gather_row r;
            r.index = sink.base_index + i;
            r.x1 = res.x1;
            r.x2 = res.x2;
            r.mm = res.mm;
            r.n_cls = res.n_cls;
            sink.rows[pos] = r;
            if (sink.lines_in) {
                const uint4* a = (const uint4*)(sink.lines_in + i);
                uint4* b = (uint4*)(sink.lines_out + pos);
                b[0] = a[0];
                b[1] = a[1];
                b[2] = a[2];
            }
        }
    }
}

// Segment mode of the row sink: called by all 64 lanes of a wave, at its own pace.  lds_counter: LDS byte address of the
// workgroup's row counter.
__device__ __forceinline__ void append_rows_segment(const RowSink& sink, bool valid, const hc_result_rec& res, uint64_t i, uint32_t lds_counter) {
    const bool keep = valid && (res.n_cls >> 28) != HC_CLS_DROP;
    const uint64_t m = __ballot(keep);
    if (m == 0ull) return;  // wave-uniform
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t base = 0;
    if (lane == 0) base = lds_add_rtn(lds_counter, (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, 0, 64);
    if (keep) {
        const uint32_t pos = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (pos < sink.seg_rows) {
            hc_gather_row r;
            r.index = sink.base_index + i;
            r.x1 = res.x1;
            r.x2 = res.x2;
            r.mm = res.mm;
            r.n_cls = res.n_cls;
            sink.seg_buf[(uint64_t)blockIdx.x * sink.seg_rows + pos] = r;
        }
    }
    if (base + (uint32_t)__popcll(m) > sink.seg_rows) {  // wave-uniform: some of this wave's rows found the segment full
        const bool over = keep && base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)) >= sink.seg_rows;
        const uint64_t mo = __ballot(over);
        uint32_t sbase = 0;
        if (lane == 0) sbase = atomicAdd(sink.spill_count, (uint32_t)__popcll(mo));
        sbase = (uint32_t)__shfl((int)sbase, 0, 64);
        if (over) {
            const uint32_t pos = sbase + (uint32_t)__popcll(mo & ((1ull << lane) - 1ull));
            if (pos < sink.spill_cap) {
                hc_gather_row r;
                r.index = sink.base_index + i;
                r.x1 = res.x1;
                r.x2 = res.x2;
                r.mm = res.mm;
                r.n_cls = res.n_cls;
                sink.spill_buf[pos] = r;
            }
        }
    }
}

// The segments of a launch back to back into the payload, the spilled rows behind them: workgroup g adds up the counts in front of
// it and all of them (at most 4 096), copies its rows as 16-byte pieces and a share of the spilled ones, and the last one writes the
// total = the number of kept rows (nothing is written beyond `cap`; a total above cap tells the caller rows were lost:
// hc_score_pack_device's contract).  spill_next: the counter the launch after this one spills through, zeroed here.
__global__ __launch_bounds__(256) void sink_compact_kernel(const hc_gather_row* __restrict__ seg_buf, const uint32_t* __restrict__ seg_count, uint32_t seg_rows,
                                                           uint32_t G, hc_gather_row* __restrict__ rows, unsigned long long cap,
                                                           unsigned long long* __restrict__ count, const hc_gather_row* __restrict__ spill_buf,
                                                           const uint32_t* __restrict__ spill_count, uint32_t spill_cap, uint32_t* __restrict__ spill_next) {
    __shared__ unsigned long long part[8];
    const uint32_t g = blockIdx.x, tid = threadIdx.x;
    unsigned long long before = 0, all = 0;
    for (uint32_t k = tid; k < G; k += 256) {
        const uint32_t c = seg_count[k] < seg_rows ? seg_count[k] : seg_rows;
        if (k < g) before += c;
        all += c;
    }
    for (int o = 32; o > 0; o >>= 1) {
        before += __shfl_down(before, o, 64);
        all += __shfl_down(all, o, 64);
    }
    if ((tid & 63u) == 0) {
        part[tid >> 6] = before;
        part[4 + (tid >> 6)] = all;
    }
    __syncthreads();
    before = part[0] + part[1] + part[2] + part[3];
    all = part[4] + part[5] + part[6] + part[7];
    const uint32_t mine = seg_count[g] < seg_rows ? seg_count[g] : seg_rows;
    const uint4* src = (const uint4*)(seg_buf + (uint64_t)g * seg_rows);
    uint4* dst = (uint4*)(rows + before);
    const unsigned long long room = before < cap ? cap - before : 0ull;
    const uint32_t n_copy = mine < room ? mine : (uint32_t)room;
    for (uint32_t k = tid; k < 2u * n_copy; k += 256) dst[k] = src[k];  // 32-byte rows as two 16-byte pieces
    const uint32_t spilled = *spill_count;
    if (spilled) {  // the slow path's rows, dealt to the workgroups row by row
        const uint32_t have = spilled < spill_cap ? spilled : spill_cap;
        for (uint64_t k = (uint64_t)g * 256 + tid; k < have; k += (uint64_t)G * 256) {
            if (all + k < cap) {
                ((uint4*)(rows + all + k))[0] = ((const uint4*)(spill_buf + k))[0];
                ((uint4*)(rows + all + k))[1] = ((const uint4*)(spill_buf + k))[1];
            }
        }
    }
    if (g == G - 1 && tid == 0) {
        // the payload's whole header row: { count, 0, 0, 0 } (rows == the row behind `count`: hc_score_pack_device's layout — round 6: the
        // fill in front of every launch that used to zero it is gone from the segmented launches)
        count[0] = all + spilled;
        count[1] = 0;
        count[2] = 0;
        count[3] = 0;
        *spill_next = 0;
    }
}

__device__ __forceinline__ uint32_t length_class(uint32_t chunks) {  // 0..15 exact, then quarter octaves; < 128
    if (chunks < 16u) return chunks;
    const uint32_t lg = 31u - (uint32_t)__builtin_clz(chunks);
    const uint32_t c = 16u + (lg - 4u) * 4u + ((chunks >> (lg - 2u)) & 3u);
    return c > 127u ? 127u : c;
}

// ---------------------------------------------------------------------------
// Length bucketing (read sets of mixed sequence length: contigs next to reads, BASELINE config 5).  A wave advances its 64
// candidates row by row and runs as long as its longest one: with overlaps of log-uniform length 100..6 000 the mean lane
// is busy 28 % of that time.  Candidates are therefore bucketed by (tile, length class): inside every tile of kBucketTile
// consecutive candidates — consecutive candidates share reads, so a tile keeps the file's locality — this kernel ranks the
// candidates by the length class of their overlap (LDS counting sort, longest first) and writes the ranking as a
// permutation; the scoring kernel then takes GROUPS of 64 consecutive ranks (one wave each: lengths within a quarter octave of each other, lane efficiency 0.95 on
// config 5) from a queue, group g of every tile before group g + 1 of any — the longest groups first, so the waves that
// finish last are finishing short ones.  Results go back to the candidate's own place (out[i] <-> in[i]).
// One 1 024-lane workgroup per tile, 4 slots per lane.  perm_in: an earlier permutation to compose with (hc_set_reorder) or nullptr.
constexpr uint32_t kBucketTile = 4096;
template <int SB>
__global__ __launch_bounds__(1024) void bucket_perm_kernel(StoreView st, uint32_t min_read_len, uint32_t fmt, const void* __restrict__ in, uint64_t n,
                                                           const unsigned long long* __restrict__ n_dev, const uint32_t* __restrict__ perm_in,
                                                           uint32_t* __restrict__ perm_out, uint32_t* __restrict__ queue) {
    if (n_dev) {
        const uint64_t nd = *n_dev;
        n = nd < n ? nd : n;
    }
    __shared__ uint32_t hist[129];  // class + 1 of a candidate (1..128); 0 = no candidate in the slot (those rank last)
    const uint32_t tid = threadIdx.x;
    if (blockIdx.x == 0 && tid == 0) *queue = 0;  // the scoring kernel behind this one in the stream starts its queue at 0
    if (tid < 129) hist[tid] = 0;
    __syncthreads();
    const uint64_t tile_base = (uint64_t)blockIdx.x * kBucketTile;
    uint32_t cls[4], idx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint64_t slot = tile_base + (uint32_t)k * 1024u + tid;
        cls[k] = 0;
        idx[k] = 0;
        if (slot < n) {
            idx[k] = perm_in ? perm_in[slot] : (uint32_t)slot;
            const Cand rec = load_cand(in, idx[k], fmt);
            Sub s0, s1;
            const int ns = resolve<SB>(st, rec, s0, s1);
            uint32_t c = 0;
            if (ns >= 1) c = (sub_positions(s0, min_read_len) + 15u) >> 4;
            if (ns == 2) c += (sub_positions(s1, min_read_len) + 15u) >> 4;
            cls[k] = length_class(c) + 1u;
        }
        atomicAdd(&hist[cls[k]], 1u);
    }
    __syncthreads();
    if (tid < 64) {  // exclusive scan, longest class first: lane t owns classes 128 - 2t and 127 - 2t; class 0 follows them all
        const uint32_t hi = 128u - 2u * tid, lo = hi - 1u;
        const uint32_t c0 = hist[hi], c1 = hist[lo];
        uint32_t incl = c0 + c1;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, o, 64);
            if ((int)tid >= o) incl += up;
        }
        const uint32_t excl = incl - (c0 + c1);
        hist[hi] = excl;
        hist[lo] = excl + c0;
        if (tid == 63) hist[0] = incl;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t rank = atomicAdd(&hist[cls[k]], 1u);
        if (cls[k]) perm_out[tile_base + rank] = idx[k];  // ranks of the candidates are < the number of candidates in the tile
    }
}

// The scoring kernel with the cooperative fetch (score_sub_coop), for stores below 4 GiB (32-bit byte offsets).  Same
// results, records and row sink as score_kernel.  WG: lanes per workgroup — one log table per workgroup, so a large table
// (wide 8-bit symbols: 64 KiB; 16-bit symbols: up to 74 KiB) is shared by 1 024 lanes to keep 16 waves on a CU.
// LDS: coop_stage_base().
// SORT: the 128 sub-overlaps of a wave's 64 candidates (two per candidate at most) are dealt to its lanes by length —
// ranks from one pair of ballots per length class, longest first; lane t scores rank t and then rank 127 - t — so the 64
// sub-overlaps the wave steps through together are the longer half, then the shorter half: a wave runs as long as its
// longest lane, and with windows of 75..150 symbols next to each other a lane is busy 76 % of that time; sorted, the second
// pass usually needs one 64-byte step less.  Parameters and results change lanes through the wave's own image space: no
// workgroup barrier (a workgroup-wide sort saved more work and lost it again waiting at its seven barriers).
// DEPTH: steps of pieces in flight (2: the launch for contig-length read sets, which keeps 8 waves per CU and so has the
// registers for a second set, StoreView::long_rows).
// WQ (round 4, the LDS-DMA form): one resident workgroup per CU; the candidates are dealt to the workgroups in equal contiguous ranges and
// every WAVE takes its items — 64 consecutive candidates each — from a ticket counter in LDS (ds_add_rtn; the first item is the wave's own
// number, the next ticket is asked for before the current item is scored).  The waves of a CU then finish within one item of each other,
// and no global atomic is involved.  Measured against the static grid (a wave strides over its workgroup's iterations) and against a
// global queue (eight counters in memory, items of 512 candidates): C2 0.191 / 0.210 / 0.165 ms, C3-lite 1.525 / 1.531 / 1.482, C3
// 7.04 / 6.67 / 6.55 (profiles/r04_wave_queue.txt, r04_wq_local.txt); the CUs hold 15.3 of their 16 waves on average instead of 12.7.
template <typename SymT, int LG, int WG, bool SORT, bool DYN, int DEPTH = 1, bool WQ = false>
__global__ __launch_bounds__(WG, DEPTH == 2 ? 2 : (WG == 256 ? 4 : (WG == 512 ? 2 : 1))) void score_kernel_coop(StoreView st, ScoreParams prm, const double* __restrict__ lut_g,
                                                            const void* __restrict__ in, uint64_t n, hc_result_rec* __restrict__ out,
                                                            const uint32_t* __restrict__ perm, RowSink sink, uint32_t* __restrict__ queue) {
    // DYN (length-bucketed launches, bucket_perm_kernel): the workgroup takes WG consecutive ranks of a tile from `queue`;
    // n_hint = the launch's n, which the queue's geometry was laid out for (the device may know fewer records)
    const uint64_t n_hint = n;
    if (prm.n_dev) {
        const uint64_t nd = *prm.n_dev;
        n = nd < n ? nd : n;
    }
    extern __shared__ __attribute__((aligned(16))) double lut_s[];
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lut_s != 0u) __builtin_trap();  // lds_f64 relies on it
    const uint32_t lut_n = st.lut_bytes >> 3;
    load_log_table<SymT, LG>(lut_s, lut_g, lut_n, threadIdx.x, WG);
    constexpr bool kScratchInTable = sizeof(SymT) == 1 && LG >= 6 && WG == (int)kWideDmaLanes;  // (coop_stage_base)
    uint32_t* scratch = kScratchInTable ? (uint32_t*)((char*)lut_s + kWideDmaScratch) : (uint32_t*)(lut_s + lut_n);  // row append: [0..17]; segment mode: [24] the workgroup's row counter
    if (kScratchInTable) __syncthreads();  // the table's loaders have written the hole before its words are set
    const uint32_t seg_counter = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(scratch + 24);
    if (threadIdx.x == 0) {
        scratch[24] = 0;
        scratch[29] = 0;  // WQ: the workgroup's ticket counter
        if (DYN && !WQ) scratch[26] = atomicAdd(queue, 1u);  // the workgroup's first queue entry; [26], [27]: this iteration's and the next one's
        if (sink.started) atomicAdd(sink.started, 1ull);     // kernel-argument-uniform branch
    }
    __syncthreads();
    // this wave's image (LDS byte address); the images start at a multiple of 1 KiB (the XOR swizzle needs whole 64-byte rows)
    const uint32_t stage_base = coop_stage_base(st.lut_bytes, WG);
    // (wave-uniform, and told so: the LDS-DMA form's M0 values and every image address + constant are then scalar work)
    const uint32_t stage = (uint32_t)__builtin_amdgcn_readfirstlane(
        (int)(stage_base + (threadIdx.x >> 6) * (DEPTH == 0 ? 2u * kStageBytesPerWave : kStageBytesPerWave)));  // DEPTH 0: LDS-DMA, A and B images
    const SymT* sym = (const SymT*)st.sym;
    const uint32_t Kp = st.K + 2u;
    const uint32_t fmt = prm.rec_fmt;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)st.sym, 0, (uint32_t)st.store_bytes, 0x00020000);
    const uint32_t oob = (uint32_t)st.store_bytes;  // the first offset the descriptor's range check rejects (no wrap-around at +16)
    const uint32_t tid = threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * WG;
    // DYN: queue entry q = piece (q / n_tiles) of tile (q % n_tiles), a piece being WG consecutive ranks — WG / 64 groups of
    // neighbouring length, one per wave: piece p of every tile before piece p + 1 of any, i.e. the longest candidates of the launch
    // first.  One atomic per workgroup and iteration, asked for before the current piece is scored and published in LDS behind it;
    // the barrier that orders the two is the only one of the loop, and the waves reach it together because their groups are alike.
    // (Round 3's first form queued groups per wave: 1.6 * 10^6 atomics on one address at C3's size, 2.5 x the plain launch on reads of
    // 100..400 bp.)
    const uint32_t n_tiles = DYN ? (uint32_t)((n_hint + kBucketTile - 1) / kBucketTile) : 0u;
    const uint32_t n_pieces = n_tiles * (kBucketTile / WG);
    uint32_t q_ahead = 0, iter = 0;
    // WQ: this wave's item, the one it has asked for already, where it stands inside the item
    const uint32_t wq_steps = WQ ? ((prm.pad >> 8) & 0xFu) : 0u;        // 64-candidate steps per item (1 unless HC_WAVE_QUEUE_STEPS says otherwise)
    // DYN with WQ (bucketed launches by ticket): the workgroup owns the pieces q = blockIdx, blockIdx + G, ... of the queue's order (every
    // workgroup a like mix of long and short ones) and its waves take (piece, group of 64 ranks) pairs by ticket: no global atomic, no barrier
    const uint32_t dq_pieces = (DYN && WQ && blockIdx.x < n_pieces) ? (n_pieces - blockIdx.x + gridDim.x - 1) / gridDim.x : 0u;
    const uint32_t wq_items = !WQ ? 0u : (DYN ? dq_pieces * (uint32_t)(WG / 64) : (uint32_t)((n + 64ull * wq_steps - 1) / (64ull * wq_steps)));
    const uint32_t wq_per = !WQ ? 0u : (DYN ? wq_items : (wq_items + gridDim.x - 1) / gridDim.x);  // items of this workgroup: [wq_first, wq_end)
    const uint32_t wq_first = DYN ? 0u : blockIdx.x * wq_per, wq_end = wq_first + wq_per < wq_items ? wq_first + wq_per : wq_items;
    uint32_t wq_item = 0, wq_next_raw = 0, wq_k = 0;
    auto wq_pull = [&]() -> uint32_t {  // issued by lane 0 (the other lanes hold 0); the answer is taken when the item is entered
        uint32_t t = 0;
        if ((tid & 63u) == 0) t = lds_add_rtn(seg_counter + 20u, 1u);
        return t;
    };
    if (WQ) {
        wq_item = wq_first + (tid >> 6);  // the first item of a wave: its own number (the tickets start behind the WG / 64 of them)
        if (wq_item < wq_end) wq_next_raw = wq_pull();
    }
    for (uint64_t block_base = (uint64_t)blockIdx.x * WG;; block_base += stride, ++iter) {
        uint64_t slot;
        if (WQ) {
            if (wq_k == wq_steps) {
                wq_item = wq_first + (uint32_t)(WG / 64) + (uint32_t)__builtin_amdgcn_readfirstlane((int)wq_next_raw);
                wq_k = 0;
                if (wq_item < wq_end) wq_next_raw = wq_pull();
            }
            if (wq_item >= wq_end) break;  // wave-uniform
            if (DYN) {
                const uint32_t k = wq_item / (uint32_t)(WG / 64), w = wq_item - k * (uint32_t)(WG / 64);
                const uint32_t q = blockIdx.x + k * gridDim.x;
                const uint32_t piece = q / n_tiles, tile = q - piece * n_tiles;
                slot = (uint64_t)tile * kBucketTile + piece * WG + w * 64u + (tid & 63u);
            } else {
                slot = ((uint64_t)wq_item * wq_steps + wq_k) * 64u + (tid & 63u);
            }
            ++wq_k;
        } else if (DYN) {
            const uint32_t q = scratch[26 + (iter & 1u)];
            if (q >= n_pieces) break;  // workgroup-uniform
            if (tid == 0) q_ahead = atomicAdd(queue, 1u);
            const uint32_t piece = q / n_tiles, tile = q - piece * n_tiles;
            slot = (uint64_t)tile * kBucketTile + piece * WG + tid;  // (behind the end of the last tile: lanes without a candidate)
        } else {
            if (block_base >= n) break;
            slot = block_base + tid;
        }
        uint64_t i = 0;
        int ns = -2;  // no candidate in this lane
        // a sub-overlap as it travels: window starts as byte offsets into the store, positions | fatal << 31
        uint32_t a0 = 0, b0 = 0, l0 = 0, a1 = 0, b1 = 0, l1 = 0;
        if (slot < n) {
            i = perm ? (uint64_t)perm[slot] : slot;
            const Cand rec = load_cand(in, i, fmt);
            if (st.regular) {  // kernel-argument-uniform: descriptors are arithmetic, 32-bit offsets, no `fatal` (hc_resolve.h)
                ns = resolve_regular32<(int)sizeof(SymT)>(st, prm.min_read_len, rec, a0, b0, l0, a1, b1, l1);
            } else {
                Sub sub0{}, sub1{};
                ns = resolve<(int)sizeof(SymT)>(st, rec, sub0, sub1);
                if (ns >= 1) {
                    a0 = (uint32_t)((sub0.offA + sub0.pos) * sizeof(SymT));
                    b0 = (uint32_t)(sub0.offB * sizeof(SymT));
                    l0 = sub_positions(sub0, prm.min_read_len) | (sub0.fatal << 31);
                }
                if (ns == 2) {
                    a1 = (uint32_t)((sub1.offA + sub1.pos) * sizeof(SymT));
                    b1 = (uint32_t)(sub1.offB * sizeof(SymT));
                    l1 = sub_positions(sub1, prm.min_read_len) | (sub1.fatal << 31);
                }
            }
        }
        const uint32_t L0 = l0 & 0x7FFFFFFFu, L1 = l1 & 0x7FFFFFFFu;
        SubScore s1, s2;
        if (SORT) {
            // rank of the wave's sub-overlap 2 lane + s, longest class first, by a counting sort over the 128 length classes in the
            // upper half of the wave's own image: every sub-overlap takes a ticket in its class (ds_add_rtn: the tickets of a class
            // are its members in some order), a 64-lane scan over the bins turns the counts into class starts.  (Round 2 ranked
            // with one pair of ballots per class between the wave's longest and shortest: 24 VALU instructions per class, 150 - 200
            // per candidate on 2 x 150 bp reads, more on mixed lengths; this is 40 whatever the lengths.)
            const uint32_t lane = tid & 63u;
            const uint32_t c0 = length_class((L0 + 15u) >> 4), c1 = length_class((L1 + 15u) >> 4);
            const uint32_t hb = stage + 2048u;  // 128 bins of 4 bytes; bin 127 - c holds class c
            lds_store64(hb + 8u * lane, 0u, 0u);
            wave_lds_order();
            const uint32_t bin0 = hb + 4u * (127u - c0), bin1 = hb + 4u * (127u - c1);
            uint32_t r0 = lds_add_rtn(bin0, 1u);
            uint32_t r1 = lds_add_rtn(bin1, 1u);
            wave_lds_order();
            {
                uint32_t h0, h1;
                lds_load64(hb + 8u * lane, h0, h1);
                uint32_t incl = h0 + h1;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const uint32_t up = (uint32_t)__shfl_up((int)incl, o, 64);
                    if ((int)lane >= o) incl += up;
                }
                const uint32_t excl = incl - (h0 + h1);
                wave_lds_order();
                lds_store64(hb + 8u * lane, excl, excl + h0);
            }
            wave_lds_order();
            r0 += lds_load32(bin0);
            r1 += lds_load32(bin1);
            wave_lds_order();
            // through the wave's own image space (wave-synchronous): what to score, and where its result belongs
            const uint32_t x = stage;  // LDS byte address; 16 bytes per rank
            lds_store128(x + 16u * r0, u32x4{a0, b0, l0, 2u * lane});
            lds_store128(x + 16u * r1, u32x4{a1, b1, l1, 2u * lane + 1u});
            wave_lds_order();
            const u32x4 p0 = lds_load128(x + 16u * lane), p1 = lds_load128(x + 16u * (127u - lane));  // rank `lane`, then the other half mirrored
            wave_lds_order();
            SubScore r[2];
            score_sub_coop<SymT, LG, DEPTH>(rsrc, oob, sym, stage, p0[0], p0[1], p0[2] & 0x7FFFFFFFu, p0[2] >> 31, Kp, st.inv_n, r[0]);
            r[1].x = -__builtin_inf();
            r[1].mm = 1;
            r[1].n = 1;
            r[1].err = p1[2] >> 31;
            if (__ballot((p1[2] & 0x7FFFFFFFu) != 0u) != 0ull)
                score_sub_coop<SymT, LG, DEPTH>(rsrc, oob, sym, stage, p1[0], p1[1], p1[2] & 0x7FFFFFFFu, p1[2] >> 31, Kp, st.inv_n, r[1]);
            wave_lds_order();
            lds_store128(x + 16u * p0[3], u32x4{(uint32_t)__double2loint(r[0].x), (uint32_t)__double2hiint(r[0].x), r[0].mm, r[0].n | (r[0].err << 31)});
            lds_store128(x + 16u * p1[3], u32x4{(uint32_t)__double2loint(r[1].x), (uint32_t)__double2hiint(r[1].x), r[1].mm, r[1].n | (r[1].err << 31)});
            wave_lds_order();
            const u32x4 q0 = lds_load128(x + 32u * lane), q1 = lds_load128(x + 32u * lane + 16u);
            wave_lds_order();
            s1.x = __hiloint2double((int)q0[1], (int)q0[0]);
            s1.mm = q0[2];
            s1.n = q0[3] & 0x7FFFFFFFu;
            s1.err = q0[3] >> 31;
            s2.x = __hiloint2double((int)q1[1], (int)q1[0]);
            s2.mm = q1[2];
            s2.n = q1[3] & 0x7FFFFFFFu;
            s2.err = q1[3] >> 31;
        } else {
            score_sub_coop<SymT, LG, DEPTH>(rsrc, oob, sym, stage, a0, b0, L0, l0 >> 31, Kp, st.inv_n, s1);
            if (__ballot(ns == 2) != 0ull) score_sub_coop<SymT, LG, DEPTH>(rsrc, oob, sym, stage, a1, b1, L1, l1 >> 31, Kp, st.inv_n, s2);
        }
        if (ns != 2) {  // what a candidate without a second sub-overlap reports (compute_overlap, s-s)
            s2.x = __builtin_nan("");
            s2.mm = 0;
            s2.n = 1;
            s2.err = 0;
        }
        hc_result_rec res;
        res.n_cls = 0;
        if (slot < n) {
            if (ns <= 0) {  // malformed record: an error; "skip" record: a dropped result
                res.x1 = -__builtin_inf();
                res.x2 = __builtin_nan("");
                res.mm = 1;
                res.n_cls = 1u | ((ns == 0 ? HC_CLS_ERROR : HC_CLS_DROP) << 28);
                store_result<!DYN>(out, i, res);
            } else {
                res = classify_and_store<!DYN>(prm, ns, s1, s2, i, out);
            }
        }
        if (sink.rows) {  // kernel-argument-uniform branch
            if (sink.seg_count) append_rows_segment(sink, slot < n, res, i, seg_counter);
            else if (DYN) append_rows_wave(sink, slot < n, res, i);
            else append_rows_block(sink, slot < n, res, i, scratch);
        }
        if (DYN && !WQ) {
            if (tid == 0) scratch[26 + ((iter + 1u) & 1u)] = q_ahead;
            __syncthreads();
        }
    }
    if (sink.rows && sink.seg_count) {  // kernel-argument-uniform: every wave of the workgroup leaves its loop and arrives here
        __syncthreads();
        if (threadIdx.x == 0) sink.seg_count[blockIdx.x] = scratch[24];
    }
}

// (second launch bound: at least 4 waves per SIMD, i.e. at most 128 registers — at 130 the kernel drops to 3 waves per SIMD
// and loses a quarter of the loads in flight; the 16-bit-symbol instantiations need 132 registers: 3 waves per SIMD — 4 with spills measured slower)
template <typename SymT, int G, int LG, bool BAL>
__global__ __launch_bounds__(256, sizeof(SymT) == 1 ? 4 : 3) void score_kernel(StoreView st, ScoreParams prm, const double* __restrict__ lut_g,
                                                    const void* __restrict__ in, uint64_t n, hc_result_rec* __restrict__ out,
                                                    const uint32_t* __restrict__ perm, RowSink sink) {
    score_kernel_body<SymT, G, LG, BAL>(st, prm, lut_g, in, n, out, perm, sink);
}

// The same kernel for 512-lane workgroups.  A large quality alphabet means a large log table in LDS (64 KiB for the
// wide 8-bit symbols), and with one table per 256-lane workgroup only 2 workgroups fit a CU: 8 waves, too few to hide
// the gather latency.  One table shared by 512 lanes restores 16 waves per CU (see launch_score).
template <typename SymT, int G, int LG>
__global__ __launch_bounds__(512, 4) void score_kernel_wide_wg(StoreView st, ScoreParams prm, const double* __restrict__ lut_g,
                                                            const void* __restrict__ in, uint64_t n,
                                                            hc_result_rec* __restrict__ out, const uint32_t* __restrict__ perm,
                                                            RowSink sink) {
    score_kernel_body<SymT, G, LG, false>(st, prm, lut_g, in, n, out, perm, sink);
}

// ---------------------------------------------------------------------------
// Launch wrappers (called from hc_api.cpp).
hipError_t launch_encode(uint32_t symbytes, const uint8_t* bases, const uint8_t* quals, const uint64_t* raw_off,
                         const uint64_t* seq_off, const uint32_t* rc_delta, const uint8_t* qmap, uint32_t n_seq, uint32_t K, void* sym,
                         uint8_t* seq_bad, const uint32_t* read_first_seq, uint32_t n_reads, ReadDesc* descs, uint32_t slot_align,
                         hipStream_t stream) {
    if (n_seq == 0) return hipSuccess;
    const uint32_t waves_per_block = 4;
    uint32_t blocks = (n_seq + waves_per_block - 1) / waves_per_block;
    if (blocks > 65536) blocks = 65536;
    if (symbytes == 1 && lut_lg(K) >= 6)
        hipLaunchKernelGGL((encode_store_kernel<uint8_t, true>), dim3(blocks), dim3(256), 0, stream, bases, quals, raw_off,
                           seq_off, rc_delta, qmap, n_seq, K, (uint8_t*)sym, seq_bad, slot_align);
    else if (symbytes == 1)
        hipLaunchKernelGGL((encode_store_kernel<uint8_t, false>), dim3(blocks), dim3(256), 0, stream, bases, quals, raw_off,
                           seq_off, rc_delta, qmap, n_seq, K, (uint8_t*)sym, seq_bad, slot_align);
    else
        hipLaunchKernelGGL((encode_store_kernel<uint16_t, false>), dim3(blocks), dim3(256), 0, stream, bases, quals, raw_off,
                           seq_off, rc_delta, qmap, n_seq, K, (uint16_t*)sym, seq_bad, slot_align);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (n_reads)
        hipLaunchKernelGGL(build_read_desc_kernel, dim3((n_reads + 255) / 256), dim3(256), 0, stream, read_first_seq,
                           seq_off, raw_off, seq_bad, rc_delta, n_reads, descs);
    return hipGetLastError();
}

namespace {
struct ScoreLaunch {
    const StoreView& st;
    const ScoreParams& prm;
    const double* lut_g;
    const void* in;
    uint64_t n;
    hc_result_rec* out;
    const uint32_t* perm;
    RowSink sink;
    uint32_t blocks, wg;
    size_t lds;
    hipStream_t stream;
};

template <typename SymT, int G, int LG>
void launch_one(const ScoreLaunch& a) {
    if (a.wg == 512) {
        if constexpr (sizeof(SymT) == 1 && LG >= 6)
            hipLaunchKernelGGL((score_kernel_wide_wg<SymT, G, LG>), dim3(a.blocks), dim3(512), a.lds, a.stream, a.st, a.prm, a.lut_g,
                               a.in, a.n, a.out, a.perm, a.sink);
        return;
    }
    if (a.st.balance)
        hipLaunchKernelGGL((score_kernel<SymT, G, LG, true>), dim3(a.blocks), dim3(256), a.lds, a.stream, a.st, a.prm, a.lut_g, a.in,
                           a.n, a.out, a.perm, a.sink);
    else
        hipLaunchKernelGGL((score_kernel<SymT, G, LG, false>), dim3(a.blocks), dim3(256), a.lds, a.stream, a.st, a.prm, a.lut_g, a.in,
                           a.n, a.out, a.perm, a.sink);
}

template <typename SymT, int LG>
void launch_lg(int group, const ScoreLaunch& a) {
    if constexpr (sizeof(SymT) == 2) {  // 16-bit symbols: 32-symbol fetch groups only (64-symbol groups need 196 registers)
        launch_one<SymT, 2, LG>(a);
    } else {
        if (group == 2) launch_one<SymT, 2, LG>(a);
        else launch_one<SymT, 4, LG>(a);
    }
}
}  // namespace

// The LDS-DMA form of the cooperative fetch (score_sub_coop, DEPTH = 0) runs 1 024-lane workgroups, one per CU, whose waves take their items
// by ticket: launches of fewer candidates than this keep the 256-lane register-staged form, which fills the chip with four times as many
// workgroups.  2 * 10^5 candidates: 0.0224 against 0.0253 ms, 4 * 10^5: 0.0441 / 0.0482, 10^5: 0.0251 / 0.0198 (profiles/r04_tickets_reg.txt;
// round 3, static grid: from 5 * 10^5 on).  HC_COOP_DMA=0 turns it off (a tuning knob).
constexpr uint64_t kDmaMinCandidates = 150000;
static bool coop_dma_wanted() {
    static const bool on = !(getenv("HC_COOP_DMA") && atoi(getenv("HC_COOP_DMA")) == 0);
    return on;
}
// HC_WIDE_DMA=0: the wide 8-bit table (64 KiB) keeps the register-staged form at every size (round 6's A/B knob; default: its own LDS-DMA form)
static bool wide_dma_wanted() {  // (read at every launch: the tests switch it inside one process)
    const char* e = getenv("HC_WIDE_DMA");
    return !(e && atoi(e) == 0);
}
// HC_WAVE_QUEUE=0: the LDS-DMA form on the static grid (round 3's launch; an A/B knob); HC_WAVE_QUEUE_STEPS: 64-candidate steps per item (1)
static bool wave_queue_on(uint32_t* steps_out) {
    static const bool on = !(getenv("HC_WAVE_QUEUE") && atoi(getenv("HC_WAVE_QUEUE")) == 0);
    static const uint32_t steps = getenv("HC_WAVE_QUEUE_STEPS") ? (uint32_t)std::min(8, std::max(1, atoi(getenv("HC_WAVE_QUEUE_STEPS")))) : 1u;
    if (steps_out) *steps_out = steps;
    return on;
}
static uint64_t coop_dma_min() {  // HC_COOP_DMA_MIN: test knob — the LDS-DMA form for launches of that many candidates and more
    const char* e = getenv("HC_COOP_DMA_MIN");
    return e ? strtoull(e, nullptr, 10) : kDmaMinCandidates;
}

// fetch_group: 0 = cooperative fetch (falls back to lane_fetch_group for stores of 4 GiB and more); per lane: 4 = 64-symbol
// fetch groups (short reads), 2 = 32-symbol groups (contig-length sequences); chosen per
// read set by hc_set_reads.  rows == nullptr: plain scoring; otherwise every non-dropped record is also appended to
// rows (RowSink above).
hipError_t launch_score(const StoreView& st, const ScoreParams& prm, const double* lut_g, const void* in, uint64_t n,
                        hc_result_rec* out, const uint32_t* perm, uint32_t n_cu, int fetch_group, int lane_fetch_group, hc_gather_row* rows,
                        unsigned long long* row_count, uint64_t cap, uint64_t base_index, hipStream_t stream,
                        const hc_line_rec* lines_in, hc_line_rec* lines_out, uint32_t* bucket_perm, uint32_t* bucket_queue, hc_gather_row* seg_buf,
                        uint32_t* seg_count, uint64_t seg_total_rows, uint32_t* spill_turn, unsigned long long* started, uint32_t* started_groups) {
    if (started_groups) *started_groups = 0;
    if (n == 0) return hipSuccess;
    const uint32_t lg = lut_lg(st.K);
    // rows != nullptr is hc_score_pack_device's layout: row_count = the first word of the 32-byte header row in front of `rows`.  A launch
    // that appends through the counter itself (no segments) needs it zeroed first; a segmented launch leaves the header to its compaction.
    auto zero_header = [&]() -> hipError_t { return rows ? hipMemsetAsync(row_count, 0, sizeof(hc_gather_row), stream) : hipSuccess; };
    if (fetch_group == 0) {
        // a 4 KiB image per wave next to the table: 256-lane workgroups while four of them fit a CU, else one table for 1 024 lanes
        const bool coop = st.store_bytes < 0xFFFF0000ull;
        const size_t lds_256 = coop_stage_base(st.lut_bytes, 256) + 4 * kStageBytesPerWave;
        const uint32_t wg_c = 4 * lds_256 <= 160 * 1024 ? 256u : 1024u;
        const size_t lds_c = coop_stage_base(st.lut_bytes, wg_c) + (wg_c / 64) * kStageBytesPerWave;
        if (coop && lds_c <= 160 * 1024) {
            uint32_t per_cu = (uint32_t)((160 * 1024) / lds_c);
            per_cu = per_cu * (wg_c / 64) > 32 ? 32 / (wg_c / 64) : per_cu;
            size_t lds_launch = lds_c;
            static const int wg_per_cu = getenv("HC_COOP_WG_PER_CU") ? atoi(getenv("HC_COOP_WG_PER_CU")) : 0;  // experiment knob: fewer resident workgroups
            // 8 waves per CU: contig-length sequences (StoreView::long_rows) and every length-bucketed launch (measured on reads of
            // 100..400, 150..1 500 and 150..6 000 bp: 0.233 / 0.348 / 0.629 ms against 0.260 / 0.435 / 0.841 with 16)
            const uint32_t want_per_cu = wg_per_cu > 0 ? (uint32_t)wg_per_cu : ((st.long_rows || st.balance) && wg_c == 256 ? 2u : 0u);
            if (want_per_cu > 0 && want_per_cu < per_cu) {
                per_cu = want_per_cu;
                lds_launch = std::max(lds_c, (size_t)(160 * 1024) / (per_cu + 1) + 1024);  // LDS no other workgroup fits beside
            }
            // mixed sequence lengths: bucket the candidates by (tile, length class) first; the waves then take groups of 64
            // ranks from a queue (bucket_perm_kernel)
            const bool bucketed = st.balance && bucket_perm && bucket_queue && n < (1ull << 32);
            uint64_t blocks_c = (n + wg_c - 1) / wg_c;
            static const int grid_mult = getenv("HC_GRID_MULT") ? std::max(1, atoi(getenv("HC_GRID_MULT"))) : 4;  // experiment knob
            const uint64_t cap_c = (uint64_t)n_cu * per_cu * (bucketed ? 1 : grid_mult);  // a queue needs resident workgroups only
            if (blocks_c > cap_c) blocks_c = cap_c;
            RowSink sink{rows, row_count, cap, base_index, lines_in, lines_out, nullptr, nullptr, 0u, 0u, nullptr, nullptr, started};
            // the cooperative launches collect their rows in per-workgroup segments (RowSink); G = the launch's workgroups.  Of the
            // seg_total_rows rows of scratch the last `cap` are the spill area, the others are dealt to the workgroups.
            const bool segmented = rows && seg_buf && seg_count && spill_turn && !lines_in && cap < 0xFFFFFFFFull && seg_total_rows > cap;
            bool seg_on = false;
            auto use_segments = [&](uint64_t G) {
                seg_on = segmented && G <= kSinkMaxGroups;  // seg_count holds that many counters (a larger grid only under HC_GRID_MULT)
                if (!seg_on) return;
                const uint64_t per = (seg_total_rows - cap) / G;
                sink.seg_buf = seg_buf;
                sink.seg_count = seg_count;
                sink.seg_rows = (uint32_t)std::min<uint64_t>(per, 0xFFFFFFFFull);
                sink.spill_buf = seg_buf + (seg_total_rows - cap);
                sink.spill_cap = (uint32_t)cap;
                sink.spill_count = seg_count + kSinkMaxGroups + (*spill_turn & 1u);
            };
            auto compact_segments = [&](uint64_t G) {
                if (seg_on)
                    hipLaunchKernelGGL(sink_compact_kernel, dim3((uint32_t)G), dim3(256), 0, stream, (const hc_gather_row*)seg_buf, (const uint32_t*)seg_count,
                                       sink.seg_rows, (uint32_t)G, rows, (unsigned long long)cap, row_count, (const hc_gather_row*)sink.spill_buf,
                                       (const uint32_t*)sink.spill_count, sink.spill_cap, seg_count + kSinkMaxGroups + ((*spill_turn + 1u) & 1u));
                if (seg_on) ++*spill_turn;  // the next segmented launch spills through the counter this one has just zeroed
            };
            static const int deep_env = getenv("HC_COOP_DEPTH") ? atoi(getenv("HC_COOP_DEPTH")) : 0;  // experiment knob: 1 = one step in flight always
            const bool deep = bucketed && per_cu <= 2 && wg_c == 256 && deep_env != 1;
            if (bucketed) {
                const uint32_t tiles = (uint32_t)((n + kBucketTile - 1) / kBucketTile);
                if (st.symbytes == 2)
                    hipLaunchKernelGGL((bucket_perm_kernel<2>), dim3(tiles), dim3(1024), 0, stream, st, prm.min_read_len, prm.rec_fmt, in, n, prm.n_dev,
                                       perm, bucket_perm, bucket_queue);
                else
                    hipLaunchKernelGGL((bucket_perm_kernel<1>), dim3(tiles), dim3(1024), 0, stream, st, prm.min_read_len, prm.rec_fmt, in, n, prm.n_dev,
                                       perm, bucket_perm, bucket_queue);
                perm = bucket_perm;
            }
            // LDS-DMA form (score_sub_coop, DEPTH = 0): 8 KiB of image per wave, so one 1 024-lane workgroup with one table per CU;
            // 8-bit symbols with a table of at most 16 KiB
            const size_t lds_dma = coop_stage_base(st.lut_bytes, 1024) + 16 * 2 * kStageBytesPerWave;
            {
                // the wide 8-bit table (64 KiB) in the LDS-DMA form: 12 waves per CU instead of the register-staged form's 16 (round 6;
                // HC_WIDE_DMA=0: the register-staged form, the A/B knob)
                uint32_t steps = 1;
                if (wide_dma_wanted() && coop_dma_wanted() && n >= coop_dma_min() && !bucketed && st.symbytes == 1 && lg >= 6 && st.lut_bytes == 65536u &&
                    wave_queue_on(&steps) && !(rows && !segmented)) {
                    const uint64_t blocks_w = std::min<uint64_t>((n + kWideDmaLanes - 1) / kWideDmaLanes, n_cu);
                    ScoreParams pq = prm;
                    pq.pad = (prm.pad & 0xFFu) | (steps << 8);
                    use_segments(blocks_w);
                    if (!seg_on) {
                        const hipError_t ze = zero_header();
                        if (ze != hipSuccess) return ze;
                    }
                    if (lg == 6)
                        hipLaunchKernelGGL((score_kernel_coop<uint8_t, 6, (int)kWideDmaLanes, true, false, 0, true>), dim3((uint32_t)blocks_w),
                                           dim3(kWideDmaLanes), 160 * 1024, stream, st, pq, lut_g, in, n, out, perm, sink, nullptr);
                    else
                        hipLaunchKernelGGL((score_kernel_coop<uint8_t, 7, (int)kWideDmaLanes, true, false, 0, true>), dim3((uint32_t)blocks_w),
                                           dim3(kWideDmaLanes), 160 * 1024, stream, st, pq, lut_g, in, n, out, perm, sink, nullptr);
                    compact_segments(blocks_w);
                    if (started_groups) *started_groups = (uint32_t)blocks_w;
                    return hipGetLastError();
                }
            }
            if (coop_dma_wanted() && n >= coop_dma_min() && !bucketed && st.symbytes == 1 && lg <= 5 && lds_dma <= 160 * 1024) {
                uint64_t blocks_d = (n + 1023) / 1024;
                // one workgroup is resident per CU; 16 queued per CU even out what the CUs finish at different times (C3: 1 per CU 7.32 ms,
                // 4: 7.10, 16: 6.92, 64: 6.88, 256: 7.32, one per 1 024 candidates 7.63; profiles/r03_dma_grid.txt)
                static const int grid_mult_d = getenv("HC_GRID_MULT") ? std::max(1, atoi(getenv("HC_GRID_MULT"))) : 16;
                const uint64_t cap_d = (uint64_t)n_cu * grid_mult_d;
                if (blocks_d > cap_d) blocks_d = cap_d;
                uint32_t steps = 1;
                // (rows collected without segments — a payload of 2^32 rows and more — go through append_rows_block, whose barriers need every
                // wave of a workgroup in the same iteration: that launch keeps the static grid)
                if (wave_queue_on(&steps) && !(rows && !segmented)) {
                    // one resident workgroup per CU; its waves take their items from a ticket counter in LDS (score_kernel_coop: WQ)
                    blocks_d = std::min<uint64_t>((n + 1023) / 1024, n_cu);
                    ScoreParams pq = prm;
                    pq.pad = (prm.pad & 0xFFu) | (steps << 8);
                    use_segments(blocks_d);
                    if (!seg_on) {
                        const hipError_t ze = zero_header();
                        if (ze != hipSuccess) return ze;
                    }
#define HC_COOP_WQ_LAUNCH(LG_)                                                                                                                  \
    hipLaunchKernelGGL((score_kernel_coop<uint8_t, LG_, 1024, true, false, 0, true>), dim3((uint32_t)blocks_d), dim3(1024), lds_dma, stream, st, \
                       pq, lut_g, in, n, out, perm, sink, nullptr)
                    if (lg == 3) HC_COOP_WQ_LAUNCH(3);
                    else if (lg == 4) HC_COOP_WQ_LAUNCH(4);
                    else HC_COOP_WQ_LAUNCH(5);
#undef HC_COOP_WQ_LAUNCH
                    compact_segments(blocks_d);
                    if (started_groups) *started_groups = (uint32_t)blocks_d;
                    return hipGetLastError();
                }
                use_segments(blocks_d);
                if (!seg_on) {
                    const hipError_t ze = zero_header();
                    if (ze != hipSuccess) return ze;
                }
#define HC_COOP_DMA_LAUNCH(LG_)                                                                                                       \
    hipLaunchKernelGGL((score_kernel_coop<uint8_t, LG_, 1024, true, false, 0>), dim3((uint32_t)blocks_d), dim3(1024), lds_dma, stream, st, prm, \
                       lut_g, in, n, out, perm, sink, nullptr)
                if (lg == 3) HC_COOP_DMA_LAUNCH(3);
                else if (lg == 4) HC_COOP_DMA_LAUNCH(4);
                else HC_COOP_DMA_LAUNCH(5);
#undef HC_COOP_DMA_LAUNCH
                compact_segments(blocks_d);
                if (started_groups) *started_groups = (uint32_t)blocks_d;
                return hipGetLastError();
            }
            using W256 = std::integral_constant<int, 256>;
            using W1024 = std::integral_constant<int, 1024>;
            // the plain register-staged launches of 1 024-lane workgroups (wide 8-bit and 16-bit symbol tables: one workgroup per CU) take their
            // items by ticket too (WQ): C4 with 35 quality values 0.153 -> 0.122 ms, with 60 (16-bit symbols) 0.208 -> 0.179; 256-lane workgroups
            // (four per CU, four waves each) gain nothing by it (0.0195 / 0.0198, 0.0505 / 0.0482 ms) and keep the static grid
            // (profiles/r04_tickets_reg.txt)
            uint32_t steps_c = 1;
            const bool tickets = !bucketed && wg_c == 1024 && wave_queue_on(&steps_c) && !(rows && !segmented);  // (append_rows_block has barriers)
            ScoreParams pq = prm;
            if (tickets) {
                pq.pad = (prm.pad & 0xFFu) | (steps_c << 8);
                blocks_c = std::min<uint64_t>((n + wg_c - 1) / wg_c, (uint64_t)n_cu * per_cu);
            }
            // bucketed launches go by ticket as well: the workgroup owns the pieces blockIdx, blockIdx + G, ... of the longest-first order and
            // its waves take (piece, group of 64 ranks) pairs from the LDS counter — no global queue atomic, no barrier per piece (round 3's
            // workgroup queue: C5 0.627 -> 0.620 ms, singles of 150..1 500 bp 0.342 -> 0.324, 120..900 bp 0.283 -> 0.268; profiles/r04_bucket_tickets.txt)
            if (bucketed) pq.pad = (prm.pad & 0xFFu) | (1u << 8);
            // The instantiations a read set can reach: 8-bit symbols with a table of at most 16 KiB (LG 3..5) always fit four 256-lane
            // workgroups per CU; the wide 8-bit encoding (64 KiB table) always shares one table among 1 024 lanes; 16-bit symbols take
            // either, by table size.  Nothing else is compiled (round 3 carried 45 scoring kernels, a third of them unreachable).
            auto launch_coop = [&](auto sym_tag, auto lg_tag, auto wg_tag) {
                using T_ = decltype(sym_tag);
                constexpr int LG_ = decltype(lg_tag)::value;
                constexpr int WG_ = decltype(wg_tag)::value;
                if (bucketed && deep) {
                    if constexpr (WG_ == 256)
                        hipLaunchKernelGGL((score_kernel_coop<T_, LG_, 256, true, true, 2, true>), dim3((uint32_t)blocks_c), dim3(256), lds_launch, stream, st, pq,
                                           lut_g, in, n, out, perm, sink, bucket_queue);
                } else if (bucketed) {
                    hipLaunchKernelGGL((score_kernel_coop<T_, LG_, WG_, true, true, 1, true>), dim3((uint32_t)blocks_c), dim3(WG_), lds_launch, stream, st, pq, lut_g,
                                       in, n, out, perm, sink, bucket_queue);
                } else if (tickets) {
                    if constexpr (WG_ == 1024)
                        hipLaunchKernelGGL((score_kernel_coop<T_, LG_, 1024, true, false, 1, true>), dim3((uint32_t)blocks_c), dim3(1024), lds_launch, stream, st,
                                           pq, lut_g, in, n, out, perm, sink, nullptr);
                } else {
                    hipLaunchKernelGGL((score_kernel_coop<T_, LG_, WG_, true, false>), dim3((uint32_t)blocks_c), dim3(WG_), lds_launch, stream, st, prm, lut_g,
                                       in, n, out, perm, sink, nullptr);
                }
            };
            use_segments(blocks_c);
            if (!seg_on) {
                const hipError_t ze = zero_header();
                if (ze != hipSuccess) return ze;
            }
            if (st.symbytes == 2) {
                if (wg_c == 256) launch_coop(uint16_t{}, std::integral_constant<int, 5>{}, W256{});
                else launch_coop(uint16_t{}, std::integral_constant<int, 5>{}, W1024{});
            } else if (lg >= 6) {
                if (wg_c != 1024) return hipErrorInvalidConfiguration;  // (a 64 KiB table never leaves room for four workgroups)
                if (lg == 6) launch_coop(uint8_t{}, std::integral_constant<int, 6>{}, W1024{});
                else launch_coop(uint8_t{}, std::integral_constant<int, 7>{}, W1024{});
            } else {
                if (wg_c != 256) return hipErrorInvalidConfiguration;   // (tables of at most 16 KiB always do)
                if (lg == 3) launch_coop(uint8_t{}, std::integral_constant<int, 3>{}, W256{});
                else if (lg == 4) launch_coop(uint8_t{}, std::integral_constant<int, 4>{}, W256{});
                else launch_coop(uint8_t{}, std::integral_constant<int, 5>{}, W256{});
            }
            compact_segments(blocks_c);
            if (started_groups) *started_groups = (uint32_t)blocks_c;
            return hipGetLastError();
        }
        fetch_group = lane_fetch_group;
    }
    auto lds_for = [&](uint32_t lanes) { return st.lut_bytes + (128 + (st.balance ? kBalItems : 1) * lanes + 32) * sizeof(uint32_t); };
    // Fill the chip: enough 256-thread blocks for 8 waves per SIMD, bounded by LDS.
    uint32_t blocks_per_cu = 8;
    const uint32_t by_lds = (uint32_t)((160 * 1024) / lds_for(256));
    if (by_lds < blocks_per_cu) blocks_per_cu = by_lds < 1 ? 1 : by_lds;
    // large table, few workgroups per CU: let 512 lanes share each table (measured, C4, 35 quality values, 64 KiB
    // table: 256 lanes 0.169 ms, 512 lanes 0.152 ms, 1 024 lanes 0.174 ms)
    const uint32_t wg = (!st.balance && st.symbytes == 1 && lg >= 6 && blocks_per_cu <= 2) ? 512u : 256u;
    const size_t lds = lds_for(wg);
    const uint64_t per_wg = (uint64_t)wg * (st.balance ? kBalItems : 1);  // candidates per workgroup iteration
    uint64_t blocks = (n + per_wg - 1) / per_wg;
    const uint64_t grid_cap = (uint64_t)n_cu * blocks_per_cu * 4;  // grid-stride beyond this
    if (blocks > grid_cap) blocks = grid_cap;
    {
        const hipError_t ze = zero_header();
        if (ze != hipSuccess) return ze;
    }
    const ScoreLaunch a{st, prm, lut_g, in, n, out, perm, RowSink{rows, row_count, cap, base_index, lines_in, lines_out, nullptr, nullptr, 0u, 0u, nullptr, nullptr}, (uint32_t)blocks, wg, lds, stream};
    if (st.symbytes == 2) launch_lg<uint16_t, 5>(fetch_group, a);
    else if (lg == 3) launch_lg<uint8_t, 3>(fetch_group, a);
    else if (lg == 4) launch_lg<uint8_t, 4>(fetch_group, a);
    else if (lg == 5) launch_lg<uint8_t, 5>(fetch_group, a);
    else if (lg == 6) launch_lg<uint8_t, 6>(fetch_group, a);
    else launch_lg<uint8_t, 7>(fetch_group, a);
    return hipGetLastError();
}

// Which kernel launch_score picks for this store, as text (hc_get_kernel_info: tests and bench.py name the measured kernel with it).
std::string describe_score_kernel(const StoreView& st, int fetch_group, int lane_fetch_group, uint32_t n_cu, uint64_t n) {
    const uint32_t lg = lut_lg(st.K);
    const std::string sym = st.symbytes == 2 ? "uint16_t" : "uint8_t";
    const std::string enc = st.symbytes == 2 ? "u16" : (lg >= 6 ? "wide8" : "packed8");
    char buf[640];
    if (fetch_group == 0) {
        const bool coop = st.store_bytes < 0xFFFF0000ull;
        const size_t lds_256 = coop_stage_base(st.lut_bytes, 256) + 4 * kStageBytesPerWave;
        const uint32_t wg_c = 4 * lds_256 <= 160 * 1024 ? 256u : 1024u;
        const size_t lds_c = coop_stage_base(st.lut_bytes, wg_c) + (wg_c / 64) * kStageBytesPerWave;
        if (coop && lds_c <= 160 * 1024) {
            uint32_t per_cu = (uint32_t)((160 * 1024) / lds_c);
            per_cu = per_cu * (wg_c / 64) > 32 ? 32 / (wg_c / 64) : per_cu;
            if ((st.long_rows || st.balance) && wg_c == 256 && per_cu > 2) per_cu = 2;
            const bool deep = st.balance && per_cu <= 2 && wg_c == 256;
            const uint32_t lgt = st.symbytes == 2 ? 5u : lg;
            char small[128];
            snprintf(small, sizeof small, "hc::score_kernel_coop<%s, %u, %u, true, %s, %d%s>", sym.c_str(), lgt, wg_c, st.balance ? "true" : "false", deep ? 2 : 1,
                     (st.balance || (wg_c == 1024 && wave_queue_on(nullptr))) ? ", true" : "");
            const size_t lds_dma = coop_stage_base(st.lut_bytes, 1024) + 16 * 2 * kStageBytesPerWave;
            // the wide 8-bit table: its own LDS-DMA form, 768 lanes (launch_score)
            if (!st.balance && st.symbytes == 1 && lg >= 6 && st.lut_bytes == 65536u && wide_dma_wanted() && coop_dma_wanted() && wave_queue_on(nullptr) &&
                (n == 0 || n >= coop_dma_min())) {
                snprintf(buf, sizeof buf, "hc::score_kernel_coop<%s, %u, %u, true, false, 0, true> encoding=%s table_bytes=%u lds_bytes=%u waves_per_cu=%u "
                                          "LDS-DMA fetch for launches of %llu candidates and more (one workgroup per CU, the waves take their items from a "
                                          "ticket counter in LDS; the scratch words sit in an unaddressed row of the table); smaller launches: %s",
                         sym.c_str(), lg, kWideDmaLanes, enc.c_str(), st.lut_bytes, 160u * 1024u, kWideDmaLanes / 64u, (unsigned long long)coop_dma_min(), small);
                return buf;
            }
            const bool dma = !st.balance && st.symbytes == 1 && lg <= 5 && lds_dma <= 160 * 1024 && coop_dma_wanted();
            // n != 0: the form a launch of n candidates takes; n == 0: the read set's forms in general
            if (dma && (n == 0 || n >= coop_dma_min())) {
                const bool wq = wave_queue_on(nullptr);
                (void)n_cu;
                snprintf(buf, sizeof buf, "hc::score_kernel_coop<%s, %u, 1024, true, false, 0%s> encoding=%s table_bytes=%u lds_bytes=%zu waves_per_cu=16 "
                                          "LDS-DMA fetch for launches of %llu candidates and more (one workgroup per CU, the waves take their items from a "
                                          "ticket counter in LDS); smaller launches: %s",
                         sym.c_str(), lgt, wq ? ", true" : "", enc.c_str(), st.lut_bytes, lds_dma, (unsigned long long)coop_dma_min(), small);
            } else
                snprintf(buf, sizeof buf, "%s encoding=%s table_bytes=%u lds_bytes=%zu waves_per_cu=%u%s", small, enc.c_str(), st.lut_bytes, lds_c,
                         per_cu * (wg_c / 64), st.balance ? " length-bucketed (hc::bucket_perm_kernel, items by ticket)" : "");
            return buf;
        }
        fetch_group = lane_fetch_group;
    }
    const int g = st.symbytes == 2 ? 2 : (fetch_group == 2 ? 2 : 4);
    snprintf(buf, sizeof buf, "hc::score_kernel<%s, %d, %u, %s> encoding=%s table_bytes=%u (one lane, one fetch)", sym.c_str(), g, st.symbytes == 2 ? 5u : lg,
             st.balance ? "true" : "false", enc.c_str(), st.lut_bytes);
    return buf;
}

namespace {
template <typename SymT, int LG>
hipError_t set_lds_limit_lg() {
    const int kMax = 160 * 1024;  // allow the full 160 KiB of LDS for large quality alphabets
    hipError_t e;
    if constexpr (sizeof(SymT) == 1) {
        if ((e = hipFuncSetAttribute((const void*)score_kernel<SymT, 4, LG, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kMax)) != hipSuccess) return e;
        if ((e = hipFuncSetAttribute((const void*)score_kernel<SymT, 4, LG, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kMax)) != hipSuccess) return e;
    }
    if ((e = hipFuncSetAttribute((const void*)score_kernel<SymT, 2, LG, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kMax)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)score_kernel<SymT, 2, LG, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kMax)) != hipSuccess) return e;
    if constexpr (sizeof(SymT) == 1 && LG >= 6) {
        if ((e = hipFuncSetAttribute((const void*)score_kernel_wide_wg<SymT, 4, LG>, hipFuncAttributeMaxDynamicSharedMemorySize, kMax)) != hipSuccess) return e;
        if ((e = hipFuncSetAttribute((const void*)score_kernel_wide_wg<SymT, 2, LG>, hipFuncAttributeMaxDynamicSharedMemorySize, kMax)) != hipSuccess) return e;
    }
    return hipSuccess;
}
}  // namespace

hipError_t set_score_kernel_lds_limit() {
    hipError_t e;
    const int kMax = 160 * 1024;
#define HC_COOP_ATTR(T_, LG_, WG_)                                                                                                                         \
    if ((e = hipFuncSetAttribute((const void*)score_kernel_coop<T_, LG_, WG_, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kMax)) != hipSuccess) return e; \
    if ((e = hipFuncSetAttribute((const void*)score_kernel_coop<T_, LG_, WG_, true, true, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kMax)) != hipSuccess) return e;
#define HC_COOP_ATTR_DEEP(T_, LG_) \
    if ((e = hipFuncSetAttribute((const void*)score_kernel_coop<T_, LG_, 256, true, true, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kMax)) != hipSuccess) return e;
#define HC_COOP_ATTR_DMA(LG_)                                                                                                                                          \
    if ((e = hipFuncSetAttribute((const void*)score_kernel_coop<uint8_t, LG_, 1024, true, false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, kMax)) != hipSuccess) return e; \
    if ((e = hipFuncSetAttribute((const void*)score_kernel_coop<uint8_t, LG_, 1024, true, false, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kMax)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)score_kernel_coop<uint8_t, 6, (int)kWideDmaLanes, true, false, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kMax)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)score_kernel_coop<uint8_t, 7, (int)kWideDmaLanes, true, false, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kMax)) != hipSuccess) return e;
    HC_COOP_ATTR_DMA(3)
    HC_COOP_ATTR_DMA(4)
    HC_COOP_ATTR_DMA(5)
    HC_COOP_ATTR(uint8_t, 3, 256)
    HC_COOP_ATTR(uint8_t, 4, 256)
    HC_COOP_ATTR(uint8_t, 5, 256)
    HC_COOP_ATTR(uint8_t, 6, 1024)
    HC_COOP_ATTR(uint8_t, 7, 1024)
    HC_COOP_ATTR(uint16_t, 5, 256)
    HC_COOP_ATTR(uint16_t, 5, 1024)
    if ((e = hipFuncSetAttribute((const void*)score_kernel_coop<uint8_t, 6, 1024, true, false, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kMax)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)score_kernel_coop<uint8_t, 7, 1024, true, false, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kMax)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)score_kernel_coop<uint16_t, 5, 1024, true, false, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kMax)) != hipSuccess) return e;
    HC_COOP_ATTR_DEEP(uint8_t, 3)
    HC_COOP_ATTR_DEEP(uint8_t, 4)
    HC_COOP_ATTR_DEEP(uint8_t, 5)
    HC_COOP_ATTR_DEEP(uint16_t, 5)
#undef HC_COOP_ATTR
#undef HC_COOP_ATTR_DEEP
#undef HC_COOP_ATTR_DMA
    if ((e = set_lds_limit_lg<uint8_t, 3>()) != hipSuccess) return e;
    if ((e = set_lds_limit_lg<uint8_t, 4>()) != hipSuccess) return e;
    if ((e = set_lds_limit_lg<uint8_t, 5>()) != hipSuccess) return e;
    if ((e = set_lds_limit_lg<uint8_t, 6>()) != hipSuccess) return e;
    if ((e = set_lds_limit_lg<uint8_t, 7>()) != hipSuccess) return e;
    return set_lds_limit_lg<uint16_t, 5>();
}

}  // namespace hc
