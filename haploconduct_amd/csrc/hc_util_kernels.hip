// hc_util_kernels.hip — the small kernels around the scoring kernel (gfx950): position counting for the
// algorithmic-bytes figure, the optional candidate reorder, ordered stream compaction of the non-dropped records
// and the packing of multi-GPU collection rows.  hipCUB (radix sort, select) is used as a utility here; the hot op
// is the hand-written kernel in hc_kernels.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <hipcub/hipcub.hpp>

#include "../../include/hcedge.h"
#include "hc_device.h"
#include "hc_resolve.h"

namespace hc {

// Sum of overlapped positions / sub-overlaps over a batch (algorithmic-bytes multiplier).
__global__ __launch_bounds__(256) void count_positions_kernel(StoreView st, uint32_t min_read_len, uint32_t fmt,
                                                              const void* __restrict__ in, uint64_t n,
                                                              unsigned long long* __restrict__ totals) {
    unsigned long long pos = 0, subs = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const Cand rec = load_cand(in, i, fmt);
        Sub s0, s1;
        const int ns = st.symbytes == 1 ? resolve<1>(st, rec, s0, s1) : resolve<2>(st, rec, s0, s1);
        if (ns >= 1) pos += sub_positions(s0, min_read_len);
        if (ns == 2) pos += sub_positions(s1, min_read_len);
        if (ns > 0) subs += (unsigned long long)ns;
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
__global__ __launch_bounds__(256) void make_keys_kernel(uint32_t fmt, const void* __restrict__ in, uint32_t n,
                                                        uint32_t* __restrict__ keys, uint32_t* __restrict__ idx) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Cand rec = load_cand(in, i, fmt);
    keys[i] = rec.read1 < rec.read2 ? rec.read1 : rec.read2;
    idx[i] = i;
}

size_t reorder_temp_bytes(uint32_t n) {
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                             (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)n);
    return bytes;
}

// keys_in/idx_in are scratch (n each); perm_out receives the permutation.
hipError_t launch_reorder(uint32_t n_reads, uint32_t fmt, const void* in, uint32_t n, uint32_t* keys_in, uint32_t* keys_out,
                          uint32_t* idx_in, uint32_t* perm_out, void* temp, size_t temp_bytes, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(make_keys_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, fmt, in, n, keys_in, idx_in);
    int end_bit = 1;
    while (end_bit < 32 && (n_reads >> end_bit)) end_bit++;  // keys < n_reads; a malformed record's key may exceed that: the result is a permutation either way
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

hipError_t launch_count_positions(const StoreView& st, uint32_t min_read_len, uint32_t fmt, const void* in, uint64_t n,
                                  unsigned long long* totals, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    uint64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(count_positions_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, st, min_read_len, fmt, in, n, totals);
    return hipGetLastError();
}

}  // namespace hc
