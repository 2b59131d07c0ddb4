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
//     an N position adds nothing; a `p < --mismatch` term is +inf (poison).
//   * x = (1.0/total_len) * total_score uses IEEE fp64 divide and multiply; the
//     file is compiled with -ffp-contract=off so nothing is fused.
//   * exp() is NOT taken on the device: thresholds are inverted into x-space on the
//     host through the host libm exp (guard band => HC_CLS_AMBIG, host decides).
// No MFMA: this is byte gathering + table lookup + a serial fp64 add chain.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hcedge.h"
#include "hc_device.h"

namespace hc {

// ---------------------------------------------------------------------------
// Store encoding: raw ASCII bases + quality bytes -> symbol slots (both orientations).
// One wave per sequence; lanes stride over positions (coalesced reads and writes).
template <typename SymT>
__global__ __launch_bounds__(256) void encode_store_kernel(const uint8_t* __restrict__ bases,
                                                           const uint8_t* __restrict__ quals,
                                                           const uint64_t* __restrict__ raw_off,  // [n_seq+1]
                                                           const uint64_t* __restrict__ seq_off,  // [n_seq] symbols
                                                           const uint8_t* __restrict__ qmap,      // [256] byte -> qidx, 255 = invalid
                                                           uint32_t n_seq, SymT* __restrict__ sym,
                                                           uint8_t* __restrict__ seq_flags) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t q = wave; q < n_seq; q += n_waves) {
        const uint64_t r0 = raw_off[q];
        const uint32_t len = (uint32_t)(raw_off[q + 1] - r0);
        const uint64_t f0 = seq_off[q];
        const uint64_t stride = slot_stride(len, sizeof(SymT));
        uint32_t bad = 0;
        for (uint32_t i = lane; i < (uint32_t)stride; i += 64) {
            SymT sf = 0, sr = 0;
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
                uint32_t q3 = (uint32_t)qi << 3;
                if (qi == 255) { code = kCodeBadQual; q3 = 0; }
                sf = (SymT)(q3 | code);
                const uint32_t rcode = code < 4 ? 3 - code : code;
                sr = (SymT)(q3 | rcode);
                sym[f0 + i] = sf;
                sym[f0 + stride + (len - 1 - i)] = sr;
            } else {
                // padding of both slots
                sym[f0 + i] = 0;
                sym[f0 + stride + i] = 0;
            }
        }
        const unsigned long long any_bad = __ballot(bad != 0);
        if (lane == 0) seq_flags[q] = any_bad ? kSeqFlagBadBase : 0;
    }
}

// ---------------------------------------------------------------------------
// Candidate -> sub-overlap descriptors (reference compute_overlap, :197-380).
struct Sub {
    uint64_t offA, offB;  // symbol offsets of the oriented views
    uint32_t lenA, lenB, pos;
    uint32_t fatal;  // reverse-complement of a sequence holding an invalid base: build_rev_comp exits
};

struct SeqRef {
    uint32_t q;
    uint32_t fwd;
};

__device__ __forceinline__ void make_sub(const StoreView& st, SeqRef A, SeqRef B, uint32_t pos, Sub& s) {
    const uint32_t la = st.seq_len[A.q], lb = st.seq_len[B.q];
    s.lenA = la;
    s.lenB = lb;
    s.pos = pos;
    s.offA = st.seq_off[A.q] + (A.fwd ? 0 : slot_stride(la, st.symbytes));
    s.offB = st.seq_off[B.q] + (B.fwd ? 0 : slot_stride(lb, st.symbytes));
    s.fatal = ((!A.fwd) & (st.seq_flags[A.q] & kSeqFlagBadBase)) | ((!B.fwd) & (st.seq_flags[B.q] & kSeqFlagBadBase));
}

// Returns the number of sub-overlaps (1 or 2); 0 = malformed record.
__device__ __forceinline__ int resolve(const StoreView& st, const hc_overlap_rec& r, Sub (&sub)[2]) {
    if (r.read1 >= st.n_reads || r.read2 >= st.n_reads || r.read1 == r.read2) return 0;
    const uint32_t f1 = st.read_first_seq[r.read1], f2 = st.read_first_seq[r.read2];
    const bool p1 = (st.read_first_seq[r.read1 + 1] - f1) == 2;
    const bool p2 = (st.read_first_seq[r.read2 + 1] - f2) == 2;
    const uint32_t o1 = r.ori1 ? 1u : 0u, o2 = r.ori2 ? 1u : 0u;
    // single: S(R,o); paired: F(R,o) = o ? /1 : rc(/2), K(R,o) = o ? /2 : rc(/1)
    const SeqRef F1 = {p1 ? (o1 ? f1 : f1 + 1) : f1, o1};
    const SeqRef K1 = {p1 ? (o1 ? f1 + 1 : f1) : f1, o1};
    const SeqRef F2 = {p2 ? (o2 ? f2 : f2 + 1) : f2, o2};
    const SeqRef K2 = {p2 ? (o2 ? f2 + 1 : f2) : f2, o2};
    if (!p1 && !p2) {  // s-s :199-233
        make_sub(st, F1, F2, r.pos1, sub[0]);
        return 1;
    }
    if (!p1 && p2) {  // s-p :234-271
        make_sub(st, F1, F2, r.pos1, sub[0]);
        make_sub(st, F1, K2, r.pos2, sub[1]);
        return 2;
    }
    if (p1 && !p2) {  // p-s :272-309
        make_sub(st, F1, F2, r.pos1, sub[0]);
        make_sub(st, F2, K1, r.pos2, sub[1]);
        return 2;
    }
    // p-p :312-380
    make_sub(st, F1, F2, r.pos1, sub[0]);
    if (r.ord == '1') make_sub(st, K1, K2, r.pos2, sub[1]);
    else if (r.ord == '2') make_sub(st, K2, K1, r.pos2, sub[1]);
    else return 0;
    return 2;
}

__device__ __forceinline__ uint32_t sub_positions(const Sub& s, uint32_t min_read_len) {
    if (s.pos >= s.lenA) return 0;
    if (s.lenA < min_read_len || s.lenB < min_read_len) return 0;
    const uint32_t rem = s.lenA - s.pos;
    return rem < s.lenB ? rem : s.lenB;
}

// ---------------------------------------------------------------------------
// 8 symbols per chunk: 8 bytes (uint8 symbols) or 16 bytes (uint16 symbols).
template <typename SymT>
struct Chunk;
template <>
struct Chunk<uint8_t> {
    uint64_t w;
    __device__ __forceinline__ void load(const uint8_t* p) { __builtin_memcpy(&w, p, 8); }
    __device__ __forceinline__ uint32_t get(int k) const { return (uint32_t)(w >> (8 * k)) & 0xFFu; }
};
template <>
struct Chunk<uint16_t> {
    uint64_t w0, w1;
    __device__ __forceinline__ void load(const uint16_t* p) {
        __builtin_memcpy(&w0, p, 8);
        __builtin_memcpy(&w1, p + 4, 8);
    }
    __device__ __forceinline__ uint32_t get(int k) const {
        return (uint32_t)((k < 4 ? w0 : w1) >> (16 * (k & 3))) & 0xFFFFu;
    }
};

struct SubScore {
    double x;  // (1.0/total_len)*total_score, or -inf
    uint32_t mm, n;
    uint32_t err;
};

// overlap_score (:67-139) for one sub-overlap, one lane.
template <typename SymT>
__device__ __forceinline__ SubScore score_sub(const SymT* __restrict__ sym, const Sub& s, const double* lut, uint32_t K,
                                              uint32_t min_read_len) {
    SubScore r;
    r.x = -__builtin_inf();
    r.mm = 1;
    r.n = 1;
    r.err = s.fatal;
    const uint32_t L = sub_positions(s, min_read_len);  // :76-88
    if (L == 0) return r;
    const SymT* a = sym + s.offA + s.pos;
    const SymT* b = sym + s.offB;
    double S = 0.0;
    uint32_t cn = 0, cm = 0, bad = 0;
    for (uint32_t i = 0; i < L; i += 8) {
        Chunk<SymT> ca, cb;
        ca.load(a + i);
        cb.load(b + i);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (i + k < L) {
                const uint32_t sa = ca.get(k), sb = cb.get(k);
                const uint32_t ba = sa & 7u, bb = sb & 7u;
                const uint32_t skip = (ba | bb) & 4u;  // N (or an invalid symbol) on either side: :35-39
                const uint32_t mmf = (ba != bb) ? 1u : 0u;
                bad |= (ba == kCodeBadQual) | (bb == kCodeBadQual);             // :97-98, before any scoring
                bad |= ((ba == kCodeBadBase) | (bb == kCodeBadBase)) & (S < __builtin_inf());  // :29-30, unless already rejected
                const double t = lut[((sa >> 3) * K + (sb >> 3)) * 2u + mmf];
                if (!skip) {
                    S += t;  // :119, in position order
                    cn += 1;
                    cm += mmf;
                }
            }
        }
    }
    r.err |= bad;
    if (S == __builtin_inf()) return r;  // a position fell below --mismatch: :125-127
    if (cn == 0) return r;               // :129-131
    r.x = (1.0 / (double)cn) * S;        // :137
    r.mm = cm;
    r.n = cn;
    return r;
}

// exp(x) > T  in x-space: 1 pass, 0 fail, 2 ambiguous
__device__ __forceinline__ uint32_t band_test(double x, const Band& b) { return x > b.hi ? 1u : (x <= b.lo ? 0u : 2u); }

template <typename SymT>
__global__ __launch_bounds__(256) void score_kernel(StoreView st, ScoreParams prm, const double* __restrict__ lut_g,
                                                    const hc_overlap_rec* __restrict__ in, uint64_t n,
                                                    hc_result_rec* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) double lut[];
    const uint32_t lut_n = st.K * st.K * 2u;
    for (uint32_t i = threadIdx.x; i < lut_n; i += blockDim.x) lut[i] = lut_g[i];
    __syncthreads();

    const SymT* sym = (const SymT*)st.sym;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        hc_overlap_rec rec;
        {
            const uint4* p = (const uint4*)(in + i);
            uint4 a = p[0], b = p[1];
            __builtin_memcpy(&rec, &a, 16);
            __builtin_memcpy((char*)&rec + 16, &b, 16);
        }
        Sub sub[2];
        const int ns = resolve(st, rec, sub);
        hc_result_rec res;
        if (ns == 0) {
            res.x1 = -__builtin_inf();
            res.x2 = __builtin_nan("");
            res.mm = 1;
            res.n_cls = 1u | (HC_CLS_ERROR << 28);
            out[i] = res;
            continue;
        }
        const SubScore s1 = score_sub<SymT>(sym, sub[0], lut, st.K, prm.min_read_len);
        SubScore s2;
        s2.x = __builtin_nan("");
        s2.mm = 0;
        s2.n = 1;
        s2.err = 0;
        if (ns == 2) s2 = score_sub<SymT>(sym, sub[1], lut, st.K, prm.min_read_len);

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
        // :404-413 in x-space
        // flags bit0 / bit1: threshold < 0, every score (0 included) passes
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
        Sub sub[2];
        const int ns = resolve(st, rec, sub);
        for (int k = 0; k < ns; ++k) pos += sub_positions(sub[k], min_read_len);
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
// Launch wrappers (called from hc_api.cpp).
hipError_t launch_encode(uint32_t symbytes, const uint8_t* bases, const uint8_t* quals, const uint64_t* raw_off,
                         const uint64_t* seq_off, const uint8_t* qmap, uint32_t n_seq, void* sym, uint8_t* seq_flags,
                         hipStream_t stream) {
    if (n_seq == 0) return hipSuccess;
    const uint32_t waves_per_block = 4;
    uint32_t blocks = (n_seq + waves_per_block - 1) / waves_per_block;
    if (blocks > 65536) blocks = 65536;
    if (symbytes == 1)
        hipLaunchKernelGGL(encode_store_kernel<uint8_t>, dim3(blocks), dim3(256), 0, stream, bases, quals, raw_off, seq_off,
                           qmap, n_seq, (uint8_t*)sym, seq_flags);
    else
        hipLaunchKernelGGL(encode_store_kernel<uint16_t>, dim3(blocks), dim3(256), 0, stream, bases, quals, raw_off,
                           seq_off, qmap, n_seq, (uint16_t*)sym, seq_flags);
    return hipGetLastError();
}

hipError_t launch_score(const StoreView& st, const ScoreParams& prm, const double* lut_g, const hc_overlap_rec* in,
                        uint64_t n, hc_result_rec* out, uint32_t n_cu, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const size_t lds = (size_t)st.K * st.K * 2 * sizeof(double);
    const uint32_t block = 256;
    // Fill the chip: enough 256-thread blocks for 8 waves per SIMD, bounded by LDS.
    uint32_t blocks_per_cu = 8;
    if (lds > 0) {
        const uint32_t by_lds = (uint32_t)((160 * 1024) / lds);
        if (by_lds < blocks_per_cu) blocks_per_cu = by_lds < 1 ? 1 : by_lds;
    }
    uint64_t blocks = (n + block - 1) / block;
    const uint64_t cap = (uint64_t)n_cu * blocks_per_cu * 4;  // grid-stride beyond this
    if (blocks > cap) blocks = cap;
    if (st.symbytes == 1)
        hipLaunchKernelGGL(score_kernel<uint8_t>, dim3((uint32_t)blocks), dim3(block), lds, stream, st, prm, lut_g, in, n,
                           out);
    else
        hipLaunchKernelGGL(score_kernel<uint16_t>, dim3((uint32_t)blocks), dim3(block), lds, stream, st, prm, lut_g, in, n,
                           out);
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

hipError_t set_score_kernel_lds_limit() {
    // allow the full 160 KiB of LDS for large quality alphabets
    hipError_t e = hipFuncSetAttribute((const void*)score_kernel<uint8_t>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)score_kernel<uint16_t>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               160 * 1024);
}

}  // namespace hc
