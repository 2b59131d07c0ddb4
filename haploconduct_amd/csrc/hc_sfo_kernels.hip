// hc_sfo_kernels.hip — flip (scripts/sfo2overlaps.py:112-122) and sort keys of SFO records, one lane per record.
// Order of the script's temporary file: id0, id1, sfo0, sfo1 as numbers, then the line bytewise — ori ('I' < 'N'), then
// OHA OHB OLA OLB K as decimal texts, a tab ending the shorter one below '-' and every digit.
#include "hc_fno_device.h"
#include "hc_sfo_device.h"

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
}  // namespace

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
