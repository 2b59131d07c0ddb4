// hc_resolve.h — device-side: candidate record -> sub-overlap descriptors (reference compute_overlap,
// src/EdgeCalculator.cpp:197-380).  Shared by the scoring kernel, the position counter and the
// candidate reorder.  Two record formats enter here (include/hcedge.h): hc_overlap_rec (32 bytes, the parser's full
// record) and hc_cand_rec (16 bytes: only what the device reads, the form that crosses PCIe in the stage).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hcedge.h"
#include "hc_device.h"

namespace hc {

// What resolve() reads of a candidate, whichever record format it arrived in.
struct Cand {
    uint32_t read1, read2, pos1, pos2;
    uint32_t ori1, ori2;  // 1 = "+"
    uint32_t ord;         // '-', '1', '2'; anything else is malformed for a p-p overlap
    uint32_t skip;        // compact records only: not a candidate (a line the device parser dropped before scoring)
};

// fmt: HC_REC_FULL (hc_overlap_rec) or HC_REC_COMPACT (hc_cand_rec); wave-uniform
__device__ __forceinline__ Cand load_cand(const void* __restrict__ in, uint64_t i, uint32_t fmt) {
    Cand c;
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    if (fmt == HC_REC_COMPACT) {
        const u32x4_t a = __builtin_nontemporal_load((const u32x4_t*)in + i);  // read once: streamed
        c.read1 = a.x;
        c.read2 = a.y;
        c.pos1 = a.z & HC_CAND_POS_MASK;
        c.pos2 = a.w & HC_CAND_POS_MASK;
        c.skip = a.w & HC_CAND_SKIP;
        c.ori1 = (a.z >> 28) & 1u;
        c.ori2 = (a.z >> 29) & 1u;
        const uint32_t oc = a.z >> 30;
        c.ord = oc == 1u ? (uint32_t)'1' : (oc == 2u ? (uint32_t)'2' : (oc == 0u ? (uint32_t)'-' : 0u));
    } else {
        const u32x4_t a = __builtin_nontemporal_load((const u32x4_t*)in + 2 * i);  // the second half (len1, len2, perc) is host-only
        c.read1 = a.x;
        c.read2 = a.y;
        c.pos1 = a.z;
        c.pos2 = a.w;
        const uint32_t w = ((const uint32_t*)in)[8 * i + 4];
        c.ori1 = (w & 0xFFu) ? 1u : 0u;
        c.ori2 = ((w >> 8) & 0xFFu) ? 1u : 0u;
        c.ord = (w >> 16) & 0xFFu;
        c.skip = 0;
    }
    return c;
}

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
    d.rc_delta = b.w;
    return d;
}

// the descriptor of read r in a regular store (StoreView::regular): sequences are laid out back to back
__device__ __forceinline__ ReadDesc regular_desc(const StoreView& st, uint32_t r) {
    const bool paired = r >= st.n_single;
    const uint32_t seq = paired ? st.n_single + 2u * (r - st.n_single) : r;
    ReadDesc d;
    const uint32_t slot = st.seq_syms >> 1;
    d.off1 = (uint64_t)seq * st.seq_syms;
    d.off2 = paired ? d.off1 + slot : 0u;  // [/1 fwd][/2 fwd][/1 rc][/2 rc]
    d.len1 = st.ulen;
    d.len2 = paired ? st.ulen : 0u;
    d.flags = paired ? kReadPaired : 0u;
    d.rc_delta = paired ? 2u * slot : slot;
    return d;
}

// mate: 0 = /1 (or the single sequence), 1 = /2
template <int SB>
__device__ __forceinline__ View make_view(const ReadDesc& d, uint32_t mate, uint32_t fwd) {
    View v;
    const uint64_t off = mate ? d.off2 : d.off1;
    v.len = mate ? d.len2 : d.len1;
    v.off = off + (fwd ? 0u : d.rc_delta);
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

// Returns the number of sub-overlaps (1 or 2); 0 = malformed record; -1 = a "skip" record (nothing to score).
template <int SB>
__device__ __forceinline__ int resolve(const StoreView& st, const Cand& r, Sub& s0, Sub& s1) {
    if (r.skip) return -1;
    if (r.read1 >= st.n_reads || r.read2 >= st.n_reads || r.read1 == r.read2) return 0;
    const ReadDesc d1 = st.regular ? regular_desc(st, r.read1) : load_desc(st.reads + r.read1);
    const ReadDesc d2 = st.regular ? regular_desc(st, r.read2) : load_desc(st.reads + r.read2);
    const uint32_t p1 = d1.flags & kReadPaired, p2 = d2.flags & kReadPaired;
    const uint32_t o1 = r.ori1, o2 = r.ori2;
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

// resolve() + sub_positions() for a REGULAR store below 4 GiB (StoreView::regular: one sequence length, singles before pairs, no base
// outside ACGTN — so no descriptor look-up, no `fatal`), as the cooperative kernel wants a sub-overlap: window starts as 32-bit BYTE offsets
// into the store and the number of positions.  Same decisions as resolve() (the tests run both kernels on the same sets); what is arithmetic
// there is folded here: with slot = seq_syms / 2 a single's oriented sequence S(R, o) starts at base + (o ? 0 : slot), a pair's front
// F(R, o) = o ? /1 : rc(/2) at base + (o ? 0 : 3 slot) and its back K(R, o) = o ? /2 : rc(/1) at base + (o ? slot : 2 slot)
// ([/1 fwd][/2 fwd][/1 rc][/2 rc]); every length is `ulen`, so min(lenA - pos, lenB) = ulen - pos.  Round 5: the generic path cost ~215
// VALU instructions per candidate of the kernel's 1 856 (profiles/r05_valu_attribution.txt).
template <int SB>
__device__ __forceinline__ int resolve_regular32(const StoreView& st, uint32_t min_read_len, const Cand& r, uint32_t& a0, uint32_t& b0, uint32_t& L0,
                                                 uint32_t& a1, uint32_t& b1, uint32_t& L1) {
    a0 = b0 = L0 = a1 = b1 = L1 = 0u;
    if (r.skip) return -1;
    if (r.read1 >= st.n_reads || r.read2 >= st.n_reads || r.read1 == r.read2) return 0;
    const uint32_t slot = st.seq_syms >> 1;
    const bool p1 = r.read1 >= st.n_single, p2 = r.read2 >= st.n_single;
    const uint32_t base1 = (p1 ? 2u * r.read1 - st.n_single : r.read1) * st.seq_syms;  // n_single + 2 (r - n_single)
    const uint32_t base2 = (p2 ? 2u * r.read2 - st.n_single : r.read2) * st.seq_syms;
    const uint32_t F1 = base1 + (r.ori1 ? 0u : (p1 ? 3u * slot : slot));
    const uint32_t F2 = base2 + (r.ori2 ? 0u : (p2 ? 3u * slot : slot));
    const uint32_t ok = st.ulen >= min_read_len ? st.ulen : 0u;  // :82-84 (wave-uniform)
    a0 = (F1 + r.pos1) * SB;
    b0 = F2 * SB;
    L0 = r.pos1 < ok ? ok - r.pos1 : 0u;  // :76-79, :88
    if (!p1 && !p2) return 1;
    const uint32_t K1 = p1 ? base1 + (r.ori1 ? slot : 2u * slot) : F1;
    const uint32_t K2 = p2 ? base2 + (r.ori2 ? slot : 2u * slot) : F2;
    uint32_t A, B;
    if (!p1) {
        A = F1;
        B = K2;
    } else if (!p2) {
        A = F2;
        B = K1;
    } else if (r.ord == '1') {
        A = K1;
        B = K2;
    } else if (r.ord == '2') {
        A = K2;
        B = K1;
    } else {
        return 0;
    }
    a1 = (A + r.pos2) * SB;
    b1 = B * SB;
    L1 = r.pos2 < ok ? ok - r.pos2 : 0u;
    return 2;
}

__device__ __forceinline__ uint32_t sub_positions(const Sub& s, uint32_t min_read_len) {
    if (s.pos >= s.lenA) return 0;                                  // :76-79
    if (s.lenA < min_read_len || s.lenB < min_read_len) return 0;  // :82-84
    const uint32_t rem = s.lenA - s.pos;                            // :88
    return rem < s.lenB ? rem : s.lenB;
}

}  // namespace hc
