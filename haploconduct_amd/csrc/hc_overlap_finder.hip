// hc_overlap_finder.hip — candidate generation on the device: all suffix–prefix overlaps
// (and inclusions) between the sequences of the read store, under a Hamming error rate — the job the pipelines
// give to the external `rust-overlaps` tool (savage.py:664,713; polyte.py:514,542) — reported as SFO records
// (idA idB N|I OHA OHB OLA OLB K, scripts/sfo2overlaps.py:36).
//
// Exact, not heuristic.  An overlap of length L >= T with K <= floor(e*L) mismatches contains an error-free window
// of at least w(L) = floor((L - K) / (K + 1)) positions; with w = min over L in [T, max length] and seeds of k
// symbols taken at every s-th position of a sequence, k + s - 1 <= w guarantees that one seed lies inside that
// window.  So:
//   index    every k-mer of every forward sequence (2 bits per base, a k-mer with a non-ACGT symbol never
//            matches) -> radix sort by k-mer                                             [hc_prims]
//   seeds    k-mers at positions 0, s, 2s, ... of every sequence B, forward and (with reversals) reverse
//            complement — the store holds both orientations — binary-searched in the index
//   expand   every hit (A, q) with id(A) < id(B) gives a diagonal d = q - p: key (A, B, orientation, d); done in
//            batches of seed sequences so that the number of hits in flight stays bounded whatever the coverage
//   unique   radix sort + unique of the keys (many seeds find the same diagonal)          [hc_prims]
//   verify   one lane per candidate: overlap region, length >= T, mismatches <= floor(e*L) (N matches nothing);
//            writes 8 bytes per candidate (mismatch count, flag), not a record
//   emit     exclusive scan of the flags [hc_prims], then the records of the verified candidates in key order
// Every unordered pair is examined once (the lower id is the indexed side), so no record appears twice.
#include <hip/hip_runtime.h>
#include <stdint.h>


#include "../../include/hcedge.h"
#include "hc_device.h"
#include "hc_prims.h"
#include "hc_overlap_finder.h"

namespace hc {

// 0..3 = A,C,G,T; 4 = anything else (N, invalid): matches nothing
template <int SB, bool WIDE>
__device__ __forceinline__ uint32_t base_at(const void* __restrict__ sym, uint64_t i) {
    if (SB == 1) {
        const uint32_t s = ((const uint8_t*)sym)[i];
        if (WIDE) return (s >> 2) < kWide7First ? 4u : (s & 3u);  // (quality indices 0..2 are N / invalid, 3 is never dealt: both wide encodings)
        const uint32_t c = s & 7u;
        return c < 4u ? c : 4u;
    }
    const uint32_t c = ((const uint16_t*)sym)[i] & 7u;
    return c < 4u ? c : 4u;
}

constexpr uint64_t kNoKmer = ~(uint64_t)0;

template <int SB, bool WIDE>
__device__ __forceinline__ uint64_t kmer_at(const void* __restrict__ sym, uint64_t at, uint32_t k) {
    uint64_t code = 0;
    for (uint32_t i = 0; i < k; i++) {
        const uint32_t b = base_at<SB, WIDE>(sym, at + i);
        if (b > 3u) return kNoKmer;
        code = (code << 2) | b;
    }
    return code;
}

// one wave per sequence, lanes stride over its positions
template <int SB, bool WIDE>
__global__ __launch_bounds__(256) void finder_index_kernel(const void* __restrict__ sym, const SeqRef* __restrict__ seqs,
                                                           const uint64_t* __restrict__ pos_start, uint32_t n_seq, uint32_t k,
                                                           uint64_t* __restrict__ keys, uint64_t* __restrict__ vals) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t q = wave; q < n_seq; q += n_waves) {
        const SeqRef r = seqs[q];
        const uint64_t base = pos_start[q];
        for (uint32_t p = lane; p < r.len; p += 64u) {
            keys[base + p] = p + k <= r.len ? kmer_at<SB, WIDE>(sym, r.off + p, k) : kNoKmer;
            vals[base + p] = ((uint64_t)q << 32) | p;
        }
    }
}

__device__ __forceinline__ uint64_t lower_bound_u64(const uint64_t* __restrict__ a, uint64_t n, uint64_t v) {
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// seeds of sequence q: orientation o in [0, n_ori), t-th seed at position t*s; flat id = seed_start[q] + o*nt + t
template <int SB, bool WIDE>
__global__ __launch_bounds__(256) void finder_seed_kernel(const void* __restrict__ sym, const SeqRef* __restrict__ seqs,
                                                          const uint64_t* __restrict__ seed_start, uint32_t n_seq, uint32_t k,
                                                          uint32_t s, uint32_t n_ori, uint32_t symbytes,
                                                          const uint64_t* __restrict__ keys, uint64_t n_keys,
                                                          uint64_t* __restrict__ seed_lo, uint64_t* __restrict__ seed_cnt) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t q = wave; q < n_seq; q += n_waves) {
        const SeqRef r = seqs[q];
        if (r.len < k) continue;
        const uint32_t nt = (r.len - k) / s + 1;
        const uint64_t rc_off = r.off + r.rc_delta;
        for (uint32_t j = lane; j < nt * n_ori; j += 64u) {
            const uint32_t o = j / nt, t = j - o * nt;
            const uint64_t code = kmer_at<SB, WIDE>(sym, (o ? rc_off : r.off) + (uint64_t)t * s, k);
            uint64_t lo = 0, cnt = 0;
            if (code != kNoKmer) {
                lo = lower_bound_u64(keys, n_keys, code);
                cnt = lower_bound_u64(keys, n_keys, code + 1) - lo;
            }
            seed_lo[seed_start[q] + j] = lo;
            seed_cnt[seed_start[q] + j] = cnt;
        }
    }
}

constexpr uint64_t kNoKey = ~(uint64_t)0;
constexpr int kDiagBias = 1 << 14;  // diagonals in (-2^14, 2^14)

__device__ __forceinline__ uint64_t pack_key(uint32_t idA, uint32_t idB, uint32_t o, int d) {
    return ((uint64_t)idA << 40) | ((uint64_t)idB << 16) | ((uint64_t)o << 15) | (uint64_t)(uint32_t)(d + kDiagBias);
}

constexpr uint64_t kFewHits = 24;  // up to this many hits a seed is handled by one lane, above by a whole wave

// Whether a diagonal can be reported at all — a function of the two lengths alone, asked of every seed hit BEFORE it becomes a
// key (a third of the hits at 150 bp and T = 90 lie on diagonals of shorter overlaps) and again by the verification:
// B[0] aligns with A[d]; the overlap is [max(0, d), min(la, d + lb)).
__device__ __forceinline__ bool diagonal_wanted(int d, int la, int lb, uint32_t min_overlap, uint32_t flags) {
    const int start = d > 0 ? d : 0, end = la < d + lb ? la : d + lb;
    const bool inclusion = (d >= 0 && d + lb <= la) || (d <= 0 && d + lb >= la);
    return end - start >= (int)min_overlap && (!inclusion || (flags & HC_FIND_INCLUSIONS));
}

// Number of hits of every seed that will become a candidate: the indexed sequence has the lower id (every unordered pair
// once) and the diagonal can be reported.  Replaces the raw hit counts before the scan, so that only those keys are written and sorted.
__global__ __launch_bounds__(256) void finder_count_valid_kernel(const SeqRef* __restrict__ seqs, const uint2* __restrict__ idlen,
                                                                 const uint64_t* __restrict__ seed_start,
                                                                 uint32_t n_seq, uint32_t k, uint32_t s, uint32_t n_ori,
                                                                 const uint64_t* __restrict__ vals, const uint64_t* __restrict__ seed_lo,
                                                                 const uint64_t* __restrict__ seed_cnt, uint32_t min_overlap, uint32_t flags,
                                                                 uint64_t* __restrict__ seed_valid) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t q = wave; q < n_seq; q += n_waves) {
        const SeqRef r = seqs[q];
        if (r.len < k) continue;
        const uint32_t nt = (r.len - k) / s + 1;
        // seeds with few hits: one lane each; repeat-rich seeds: the whole wave per seed
        for (uint32_t j = lane; j < nt * n_ori; j += 64u) {
            const uint64_t sid = seed_start[q] + j;
            const uint64_t cnt = seed_cnt[sid], lo = seed_lo[sid];
            if (cnt > kFewHits) continue;
            const int p = (int)((j % nt) * s);
            uint32_t mine = 0;
            for (uint64_t h = 0; h < cnt; h++) {
                const uint64_t v = vals[lo + h];
                const uint2 a = idlen[(uint32_t)(v >> 32)];  // (sfo id, length) of the indexed sequence: 8 bytes a hit, not a 24-byte SeqRef
                mine += a.x < r.sfo_id && diagonal_wanted((int)(uint32_t)v - p, (int)a.y, (int)r.len, min_overlap, flags);
            }
            seed_valid[sid] = mine;
        }
        for (uint32_t j = 0; j < nt * n_ori; j++) {
            const uint64_t sid = seed_start[q] + j;
            const uint64_t cnt = seed_cnt[sid], lo = seed_lo[sid];
            if (cnt <= kFewHits) continue;
            const int p = (int)((j % nt) * s);
            uint32_t mine = 0;
            for (uint64_t h = lane; h < cnt; h += 64u) {
                const uint64_t v = vals[lo + h];
                const uint2 a = idlen[(uint32_t)(v >> 32)];  // (sfo id, length) of the indexed sequence: 8 bytes a hit, not a 24-byte SeqRef
                mine += a.x < r.sfo_id && diagonal_wanted((int)(uint32_t)v - p, (int)a.y, (int)r.len, min_overlap, flags);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor((int)mine, o, 64);
            if (lane == 0) seed_valid[sid] = mine;
        }
    }
}

__global__ __launch_bounds__(256) void finder_expand_kernel(const SeqRef* __restrict__ seqs, const uint2* __restrict__ idlen,
                                                            const uint64_t* __restrict__ seed_start,
                                                            uint32_t q_begin, uint32_t n_seq, uint64_t out_base, uint32_t k, uint32_t s,
                                                            uint32_t n_ori,
                                                            const uint64_t* __restrict__ vals, const uint64_t* __restrict__ seed_lo,
                                                            const uint64_t* __restrict__ seed_cnt,
                                                            const uint64_t* __restrict__ seed_out, uint32_t min_overlap, uint32_t flags,
                                                            uint64_t* __restrict__ out_keys) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t q = q_begin + wave; q < n_seq; q += n_waves) {  // the seed sequences [q_begin, n_seq) of this batch
        const SeqRef r = seqs[q];
        if (r.len < k) continue;
        const uint32_t nt = (r.len - k) / s + 1;
        // seeds with few hits: one lane each
        for (uint32_t j = lane; j < nt * n_ori; j += 64u) {
            const uint64_t sid = seed_start[q] + j;
            const uint64_t cnt = seed_cnt[sid];
            if (cnt == 0 || cnt > kFewHits) continue;
            const uint32_t o = j / nt, t = j - o * nt;
            const int p = (int)(t * s);
            const uint64_t lo = seed_lo[sid];
            uint64_t at = seed_out[sid] - out_base;
            for (uint64_t h = 0; h < cnt; h++) {
                const uint64_t v = vals[lo + h];
                const uint2 a = idlen[(uint32_t)(v >> 32)];  // (sfo id, length) of the indexed sequence: 8 bytes a hit, not a 24-byte SeqRef
                const int d = (int)(uint32_t)v - p;
                if (a.x < r.sfo_id && diagonal_wanted(d, (int)a.y, (int)r.len, min_overlap, flags)) out_keys[at++] = pack_key(a.x, r.sfo_id, o, d);
            }
        }
        // repeat-rich seeds: the hits of one seed are spread over the lanes, so that it does not stall a single lane
        for (uint32_t j = 0; j < nt * n_ori; j++) {
            const uint64_t sid = seed_start[q] + j;
            const uint64_t cnt = seed_cnt[sid];
            if (cnt <= kFewHits) continue;
            const uint32_t o = j / nt, t = j - o * nt;
            const int p = (int)(t * s);
            const uint64_t lo = seed_lo[sid];
            uint64_t at = seed_out[sid] - out_base;  // where this seed's candidates go (exclusive scan of the valid counts)
            for (uint64_t h0 = 0; h0 < cnt; h0 += 64u) {
                const uint64_t h = h0 + lane;
                uint64_t key = kNoKey;
                if (h < cnt) {
                    const uint64_t v = vals[lo + h];
                    const uint2 a = idlen[(uint32_t)(v >> 32)];  // (sfo id, length) of the indexed sequence: 8 bytes a hit, not a 24-byte SeqRef
                    const int d = (int)(uint32_t)v - p;
                    if (a.x < r.sfo_id && diagonal_wanted(d, (int)a.y, (int)r.len, min_overlap, flags)) key = pack_key(a.x, r.sfo_id, o, d);
                }
                const uint64_t m = __ballot(key != kNoKey);
                if (key != kNoKey) out_keys[at + (uint64_t)__popcll(m & ((1ull << lane) - 1ull))] = key;
                at += (uint64_t)__popcll(m);
            }
        }
    }
}

// Mismatches among the first `nbytes` (1..8) symbol bytes of two 8-byte words of 8-bit symbols: a position counts when
// the bases differ or either symbol is not A,C,G,T.
template <bool WIDE>
__device__ __forceinline__ uint32_t mismatches8(uint64_t a, uint64_t b, int nbytes) {
    uint64_t m;
    if (WIDE) {  // base in bits 0-1; a quality index below 4 (the four top bits clear) marks N / invalid in both wide encodings (hc_device.h)
        uint64_t va = a | (a << 1), vb = b | (b << 1);
        va |= va << 2;
        vb |= vb << 2;
        m = ((a ^ b) & 0x0303030303030303ull) | (~(va & vb) & 0x8080808080808080ull);
    } else {  // code in bits 0-2: 0..3 bases, 4 N, 6/7 invalid
        m = ((a ^ b) | (a & 0x0404040404040404ull)) & 0x0707070707070707ull;
    }
    if (nbytes < 8) m &= ~(uint64_t)0 >> (8 * (8 - nbytes));
    const uint64_t nz = (((m & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | m) & 0x8080808080808080ull;
    return (uint32_t)__popcll(nz);
}

// Verifies candidate i: kout[i] = number of mismatches if it is a reportable overlap, flag[i] = 1; else flag[i] = 0.
template <int SB, bool WIDE>
__global__ __launch_bounds__(256) void finder_verify_kernel(const void* __restrict__ sym, const SeqRef* __restrict__ by_sfo,
                                                            uint32_t symbytes, const uint64_t* __restrict__ keys, uint64_t n,
                                                            double err_rate, uint32_t min_overlap, uint32_t flags,
                                                            uint32_t* __restrict__ kout, uint32_t* __restrict__ flag) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t key = keys[i];
        uint32_t ok = 0, mm = 0;
        if (key != kNoKey) {
            const uint32_t ida = (uint32_t)(key >> 40), idb = (uint32_t)(key >> 16) & 0xFFFFFFu, o = (uint32_t)(key >> 15) & 1u;
            const int d = (int)(uint32_t)(key & 0x7FFFu) - kDiagBias;
            const SeqRef A = by_sfo[ida], B = by_sfo[idb];
            const int la = (int)A.len, lb = (int)B.len;
            const int start = d > 0 ? d : 0, end = la < d + lb ? la : d + lb;
            const int L = end - start;
            if (diagonal_wanted(d, la, lb, min_overlap, flags)) {
                const uint32_t kmax = (uint32_t)(err_rate * (double)L);
                const uint64_t offb = o ? B.off + B.rc_delta : B.off;
                if (SB == 1) {  // 32 symbols per step, all eight loads issued before the first is looked at: the give-up test between
                                // steps makes every step a round trip to memory, and a window of 150 symbols now takes 5 of them, not 19
                                // (slots are padded by 32 bytes and more: reading past the end of the window is safe)
                    const uint8_t* pa = (const uint8_t*)sym + A.off + (uint64_t)start;
                    const uint8_t* pb = (const uint8_t*)sym + offb + (uint64_t)(start - d);
                    for (int x = 0; x < L && mm <= kmax; x += 32) {
                        uint64_t a[4], b[4];
                        __builtin_memcpy(a, pa + x, 32);
                        __builtin_memcpy(b, pb + x, 32);
#pragma unroll
                        for (int w = 0; w < 4; ++w) {
                            const int nb = L - x - 8 * w;
                            if (nb > 0) mm += mismatches8<WIDE>(a[w], b[w], nb < 8 ? nb : 8);
                        }
                    }
                } else {
                    for (int x = start; x < end && mm <= kmax; x++) {
                        const uint32_t a = base_at<SB, WIDE>(sym, A.off + (uint64_t)x);
                        const uint32_t b = base_at<SB, WIDE>(sym, offb + (uint64_t)(x - d));
                        mm += (a != b) | (a > 3u);
                    }
                }
                ok = mm <= kmax;
            }
        }
        kout[i] = mm;
        flag[i] = ok;
    }
}

// The records of the verified candidates, in candidate (= key) order: pos[] is the exclusive scan of flag[].
__global__ __launch_bounds__(256) void finder_emit_kernel(const SeqRef* __restrict__ by_sfo, const uint64_t* __restrict__ keys,
                                                          const uint32_t* __restrict__ kout, const uint32_t* __restrict__ flag,
                                                          const uint32_t* __restrict__ pos, uint64_t n, hc_sfo_rec* __restrict__ out) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        if (!flag[i]) continue;
        const uint64_t key = keys[i];
        const uint32_t ida = (uint32_t)(key >> 40), idb = (uint32_t)(key >> 16) & 0xFFFFFFu, o = (uint32_t)(key >> 15) & 1u;
        const int d = (int)(uint32_t)(key & 0x7FFFu) - kDiagBias;
        const int la = (int)by_sfo[ida].len, lb = (int)by_sfo[idb].len;
        const int start = d > 0 ? d : 0, end = la < d + lb ? la : d + lb;
        hc_sfo_rec rec;
        rec.idA = ida;
        rec.idB = idb;
        rec.OHA = d;
        rec.OHB = d + lb - la;
        rec.OLA = rec.OLB = (uint32_t)(end - start);
        rec.K = kout[i];
        rec.inverted = o;
        out[pos[i]] = rec;
    }
}

// After a batched run: the records of the batches, each sorted, are brought into one global order.
__global__ __launch_bounds__(256) void finder_rekey_kernel(const hc_sfo_rec* __restrict__ recs, uint64_t n, uint64_t* __restrict__ keys,
                                                           uint64_t* __restrict__ idx) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const hc_sfo_rec r = recs[i];
        keys[i] = pack_key(r.idA, r.idB, r.inverted, r.OHA);
        idx[i] = i;
    }
}
__global__ __launch_bounds__(256) void finder_gather_kernel(const hc_sfo_rec* __restrict__ recs, const uint64_t* __restrict__ idx, uint64_t n,
                                                            hc_sfo_rec* __restrict__ out) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) out[i] = recs[idx[i]];
}

// out[q] = off[seed_start[q]]: the running total of candidate hits at every sequence boundary (what the host needs
// to cut batches), instead of copying the whole per-seed scan to the host.
__global__ __launch_bounds__(256) void finder_boundaries_kernel(const uint64_t* __restrict__ off, const uint64_t* __restrict__ seed_start,
                                                                uint32_t n, uint64_t* __restrict__ out) {
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < n; q += gridDim.x * blockDim.x) out[q] = off[seed_start[q]];
}

// ---- launch wrappers -----------------------------------------------------------------------------------------
static uint32_t wave_grid(uint32_t n_seq) {
    uint64_t blocks = ((uint64_t)n_seq + 3) / 4;  // 4 waves per 256-thread block
    if (blocks > 65536) blocks = 65536;
    return blocks ? (uint32_t)blocks : 1u;
}

#define HC_FINDER_DISPATCH(KERNEL, GRID, ...)                                                                              \
    do {                                                                                                                   \
        if (symbytes == 2) hipLaunchKernelGGL((KERNEL<2, false>), dim3(GRID), dim3(256), 0, stream, __VA_ARGS__);          \
        else if (wide) hipLaunchKernelGGL((KERNEL<1, true>), dim3(GRID), dim3(256), 0, stream, __VA_ARGS__);               \
        else hipLaunchKernelGGL((KERNEL<1, false>), dim3(GRID), dim3(256), 0, stream, __VA_ARGS__);                        \
    } while (0)

hipError_t finder_index(const void* sym, uint32_t symbytes, bool wide, const SeqRef* seqs, const uint64_t* pos_start, uint32_t n_seq,
                        uint32_t k, uint64_t* keys, uint64_t* vals, hipStream_t stream) {
    HC_FINDER_DISPATCH(finder_index_kernel, wave_grid(n_seq), sym, seqs, pos_start, n_seq, k, keys, vals);
    return hipGetLastError();
}

hipError_t finder_seeds(const void* sym, uint32_t symbytes, bool wide, const SeqRef* seqs, const uint64_t* seed_start, uint32_t n_seq,
                        uint32_t k, uint32_t s, uint32_t n_ori, const uint64_t* keys, uint64_t n_keys, uint64_t* seed_lo,
                        uint64_t* seed_cnt, hipStream_t stream) {
    HC_FINDER_DISPATCH(finder_seed_kernel, wave_grid(n_seq), sym, seqs, seed_start, n_seq, k, s, n_ori, symbytes, keys, n_keys, seed_lo,
                       seed_cnt);
    return hipGetLastError();
}

hipError_t finder_count_valid(const SeqRef* seqs, const uint2* idlen, const uint64_t* seed_start, uint32_t n_seq, uint32_t k, uint32_t s, uint32_t n_ori,
                              const uint64_t* vals, const uint64_t* seed_lo, const uint64_t* seed_cnt, uint32_t min_overlap, uint32_t flags,
                              uint64_t* seed_valid, hipStream_t stream) {
    hipLaunchKernelGGL(finder_count_valid_kernel, dim3(wave_grid(n_seq)), dim3(256), 0, stream, seqs, idlen, seed_start, n_seq, k, s, n_ori, vals,
                       seed_lo, seed_cnt, min_overlap, flags, seed_valid);
    return hipGetLastError();
}

hipError_t finder_expand(const SeqRef* seqs, const uint2* idlen, const uint64_t* seed_start, uint32_t q_begin, uint32_t q_end, uint64_t out_base, uint32_t k,
                         uint32_t s, uint32_t n_ori, const uint64_t* vals, const uint64_t* seed_lo, const uint64_t* seed_cnt,
                         const uint64_t* seed_out, uint32_t min_overlap, uint32_t flags, uint64_t* out_keys, hipStream_t stream) {
    hipLaunchKernelGGL(finder_expand_kernel, dim3(wave_grid(q_end - q_begin)), dim3(256), 0, stream, seqs, idlen, seed_start, q_begin, q_end,
                       out_base, k, s, n_ori, vals, seed_lo, seed_cnt, seed_out, min_overlap, flags, out_keys);
    return hipGetLastError();
}

hipError_t finder_verify(const void* sym, uint32_t symbytes, bool wide, const SeqRef* by_sfo, const uint64_t* keys, uint64_t n,
                         double err_rate, uint32_t min_overlap, uint32_t flags, uint32_t* kout, uint32_t* flag, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    uint64_t blocks = (n + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    HC_FINDER_DISPATCH(finder_verify_kernel, (uint32_t)blocks, sym, by_sfo, symbytes, keys, n, err_rate, min_overlap, flags, kout, flag);
    return hipGetLastError();
}

hipError_t finder_emit(const SeqRef* by_sfo, const uint64_t* keys, const uint32_t* kout, const uint32_t* flag, const uint32_t* pos, uint64_t n,
                       hc_sfo_rec* out, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    uint64_t blocks = (n + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(finder_emit_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, by_sfo, keys, kout, flag, pos, n, out);
    return hipGetLastError();
}

hipError_t finder_boundaries(const uint64_t* off, const uint64_t* seed_start, uint32_t n, uint64_t* out, hipStream_t stream) {
    uint32_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(finder_boundaries_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, stream, off, seed_start, n, out);
    return hipGetLastError();
}
hipError_t finder_rekey(const hc_sfo_rec* recs, uint64_t n, uint64_t* keys, uint64_t* idx, hipStream_t stream) {
    uint64_t blocks = (n + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(finder_rekey_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, recs, n, keys, idx);
    return hipGetLastError();
}
hipError_t finder_gather(const hc_sfo_rec* recs, const uint64_t* idx, uint64_t n, hc_sfo_rec* out, hipStream_t stream) {
    uint64_t blocks = (n + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(finder_gather_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, recs, idx, n, out);
    return hipGetLastError();
}

// sorting, scans and unique (hc_prims.hip); temp == nullptr returns the scratch size in temp_bytes
hipError_t finder_sort_pairs(void* temp, size_t& temp_bytes, const uint64_t* k_in, uint64_t* k_out, const uint64_t* v_in, uint64_t* v_out,
                             uint64_t n, int end_bit, hipStream_t stream) {
    if (!temp) {
        temp_bytes = prims::sort_temp_bytes(n, sizeof(uint64_t), sizeof(uint64_t));
        return hipSuccess;
    }
    return prims::sort_pairs(temp, temp_bytes, k_in, k_out, v_in, v_out, n, 0, end_bit, stream);
}
hipError_t finder_sort_keys(void* temp, size_t& temp_bytes, const uint64_t* k_in, uint64_t* k_out, uint64_t n, hipStream_t stream) {
    if (!temp) {
        temp_bytes = prims::sort_temp_bytes(n, sizeof(uint64_t), 0);
        return hipSuccess;
    }
    return prims::sort_keys(temp, temp_bytes, k_in, k_out, n, 0, 64, stream);
}
hipError_t finder_scan(void* temp, size_t& temp_bytes, const uint64_t* in, uint64_t* out, uint64_t n, hipStream_t stream) {
    if (!temp) {
        temp_bytes = prims::scan_temp_bytes(n, sizeof(uint64_t));
        return hipSuccess;
    }
    return prims::exclusive_sum(temp, temp_bytes, in, out, n, stream);
}
hipError_t finder_unique(void* temp, size_t& temp_bytes, const uint64_t* in, uint64_t* out, unsigned long long* n_out, uint64_t n,
                         hipStream_t stream) {
    if (!temp) {
        temp_bytes = prims::select_temp_bytes(n);
        return hipSuccess;
    }
    return prims::unique(temp, temp_bytes, in, out, n_out, n, stream);
}
hipError_t finder_scan32(void* temp, size_t& temp_bytes, const uint32_t* in, uint32_t* out, uint64_t n, hipStream_t stream) {
    if (!temp) {
        temp_bytes = prims::scan_temp_bytes(n, sizeof(uint32_t));
        return hipSuccess;
    }
    return prims::exclusive_sum(temp, temp_bytes, in, out, n, stream);
}

}  // namespace hc
