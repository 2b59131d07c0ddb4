// hc_util_kernels.hip — the small kernels around the scoring kernel (gfx950): position counting for the
// algorithmic-bytes figure, the optional candidate reorder, ordered stream compaction of the non-dropped records
// and the packing of multi-GPU collection rows.  Sorting and selection: hc_prims.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hcedge.h"
#include "hc_device.h"
#include "hc_prims.h"
#include "hc_resolve.h"
#include "hc_text.h"

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
// gives the permutation the scoring kernel walks.
__global__ __launch_bounds__(256) void make_keys_kernel(uint32_t fmt, const void* __restrict__ in, uint32_t n,
                                                        uint32_t* __restrict__ keys, uint32_t* __restrict__ idx) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Cand rec = load_cand(in, i, fmt);
    keys[i] = rec.read1 < rec.read2 ? rec.read1 : rec.read2;
    idx[i] = i;
}

size_t reorder_temp_bytes(uint32_t n) { return prims::sort_temp_bytes(n, sizeof(uint32_t), sizeof(uint32_t)); }

// keys_in/idx_in are scratch (n each); perm_out receives the permutation.
hipError_t launch_reorder(uint32_t n_reads, uint32_t fmt, const void* in, uint32_t n, uint32_t* keys_in, uint32_t* keys_out,
                          uint32_t* idx_in, uint32_t* perm_out, void* temp, size_t temp_bytes, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(make_keys_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, fmt, in, n, keys_in, idx_in);
    int end_bit = 1;
    while (end_bit < 32 && (n_reads >> end_bit)) end_bit++;  // keys < n_reads; a malformed record's key may exceed that: the result is a permutation either way
    return prims::sort_pairs(temp, temp_bytes, keys_in, keys_out, idx_in, perm_out, n, 0, end_bit, stream);
}

// ---------------------------------------------------------------------------
// Compaction of the records the host / the gather still need (class != DROP), in sequence order.
size_t compact_temp_bytes(uint32_t n) { return prims::select_temp_bytes(n); }

hipError_t launch_compact(const hc_result_rec* res, uint32_t n, uint32_t* idx_out, unsigned long long* count_out, void* temp,
                          size_t temp_bytes, hipStream_t stream) {
    return prims::select_not_dropped(temp, temp_bytes, res, n, idx_out, count_out, stream);
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

// The 24-byte form of a collection payload (the multi-GPU exchange moves nothing else: a quarter fewer bytes over every xGMI link).
// in: (cap + 1) rows of 32 bytes as hc_score_pack_device / hc_compact_pack_device write them (row 0 = the count).  out: (cap + 1) rows of
// three 64-bit words: row 0 = { count, rows that did not fit, 0 }, row k = { x1 bits, x2 bits, index | mm << 32 | n << 46 | class << 60 }.
// A row fits when index < 2^32 and mm, n < 2^14 (reads of up to 16 383 overlapped positions); the callers choose the form per read set.
__global__ __launch_bounds__(256) void narrow_payload_kernel(const unsigned long long* __restrict__ in, unsigned long long cap,
                                                             unsigned long long* __restrict__ out) {
    __shared__ unsigned int bad_s;
    if (threadIdx.x == 0) bad_s = 0;
    __syncthreads();
    const unsigned long long count = in[0];
    const unsigned long long have = count < cap ? count : cap;
    unsigned int bad = 0;
    for (unsigned long long k = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; k < have; k += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned long long* r = in + 4 * (k + 1);
        const unsigned long long index = r[0], w = r[3];
        const unsigned long long mm = w & 0xFFFFFFFFull, n = (w >> 32) & 0x0FFFFFFFull, cls = w >> 60;
        bad += (index >> 32) != 0 || (mm >> 14) != 0 || (n >> 14) != 0;
        unsigned long long* o = out + 3 * (k + 1);
        o[0] = r[1];
        o[1] = r[2];
        o[2] = (index & 0xFFFFFFFFull) | ((mm & 0x3FFFull) << 32) | ((n & 0x3FFFull) << 46) | (cls << 60);
    }
    if (bad) atomicAdd(&bad_s, bad);
    __syncthreads();
    if (threadIdx.x == 0 && bad_s) atomicAdd(out + 1, (unsigned long long)bad_s);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        out[0] = count;
        out[2] = 0;
    }
}

hipError_t launch_narrow_payload(const void* in32, uint64_t cap, void* out24, uint32_t n_cu, hipStream_t stream) {
    hipError_t e = hipMemsetAsync((char*)out24 + 8, 0, 8, stream);  // the misfit counter (row 0, word 1)
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(narrow_payload_kernel, dim3(n_cu * 2), dim3(256), 0, stream, (const unsigned long long*)in32, (unsigned long long)cap,
                       (unsigned long long*)out24);
    return hipGetLastError();
}

hipError_t launch_pack_rows(const hc_result_rec* res, const uint32_t* idx, const unsigned long long* count, uint64_t cap, uint64_t base,
                            hc_gather_row* rows, uint32_t n_cu, hipStream_t stream) {
    hipLaunchKernelGGL(pack_rows_kernel, dim3(n_cu * 4), dim3(256), 0, stream, res, idx, count, (unsigned long long)cap,
                       (unsigned long long)base, rows);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// The non-dropped records of a scored block IN SEQUENCE ORDER, as rows tagged with their index — what the stage consumes.
// (The scoring kernel's own row append is unordered: fine for the multi-GPU payload, whose consumer sorts once per
// batch; the stage would sort every block on a host thread, which costs several times the block's PCIe time.)
// Three small passes over the 24-byte result records, which the scoring kernel has just written (L2 / Infinity Cache):
// non-dropped records per 1 024-record tile, exclusive scan of the tile counts, ordered scatter.
constexpr uint32_t kKeptTile = 1024;

__device__ __forceinline__ uint64_t records_of(uint64_t n, const unsigned long long* n_dev) {
    if (!n_dev) return n;
    const uint64_t nd = *n_dev;
    return nd < n ? nd : n;
}

__global__ __launch_bounds__(256) void kept_count_kernel(const hc_result_rec* __restrict__ res, uint64_t n,
                                                         const unsigned long long* __restrict__ n_dev, uint32_t* __restrict__ tile_cnt) {
    n = records_of(n, n_dev);
    const uint64_t i0 = (uint64_t)blockIdx.x * kKeptTile + threadIdx.x * 4u;
    uint32_t c = 0;
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (i0 + j < n) c += (res[i0 + j].n_cls >> 28) != HC_CLS_DROP;
    for (int o = 32; o > 0; o >>= 1) c += (uint32_t)__shfl_down((int)c, o, 64);
    __shared__ uint32_t part[4];
    if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// tile_off[t] = sum of tile_cnt[0..t); *total = the sum of all.  One workgroup.  For a text block it also adds up what the parse
// kernel's workgroups tallied (text_counters[0..6] += sum over the workgroups that had lines of tally[g][0..6]): a kernel of its own
// until round 4, one more launch per block for seven sums.
__global__ __launch_bounds__(1024) void scan_tiles_kernel(const uint32_t* __restrict__ tile_cnt, uint32_t n_tiles, uint32_t* __restrict__ tile_off,
                                                          unsigned long long* __restrict__ total, const uint32_t* __restrict__ tally,
                                                          unsigned long long* __restrict__ text_counters) {
    __shared__ uint32_t wave_sum[16];
    __shared__ unsigned long long tally_part[16][8];
    if (tally && !text_counters[kTextOverflow]) {
        const uint32_t n_wg = (uint32_t)((text_counters[kTextLines] + 255u) / 256u);
        unsigned long long sum[7] = {0, 0, 0, 0, 0, 0, 0};
        for (uint32_t g = threadIdx.x; g < n_wg; g += 1024u)
#pragma unroll
            for (int k = 0; k < 7; k++) sum[k] += tally[g * 8u + k];
#pragma unroll
        for (int k = 0; k < 7; k++) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) sum[k] += (unsigned long long)__shfl_xor((long long)sum[k], o, 64);
            if ((threadIdx.x & 63u) == 0) tally_part[threadIdx.x >> 6][k] = sum[k];
        }
        __syncthreads();
        if (threadIdx.x < 7) {
            unsigned long long t = 0;
            for (int w = 0; w < 16; w++) t += tally_part[w][threadIdx.x];
            text_counters[threadIdx.x] += t;
        }
    }
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n_tiles; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < n_tiles ? tile_cnt[i] : 0u;
        uint32_t incl = v;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, o, 64);
            if ((int)(threadIdx.x & 63u) >= o) incl += up;
        }
        if ((threadIdx.x & 63u) == 63u) wave_sum[threadIdx.x >> 6] = incl;
        __syncthreads();
        uint32_t before = carry;
        for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) before += wave_sum[w];
        if (i < n_tiles) tile_off[i] = before + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}

__global__ __launch_bounds__(256) void kept_scatter_kernel(const hc_result_rec* __restrict__ res, uint64_t n,
                                                           const unsigned long long* __restrict__ n_dev, const uint32_t* __restrict__ tile_off,
                                                           uint64_t base_index, hc_gather_row* __restrict__ rows, uint64_t cap,
                                                           const hc_line_rec* __restrict__ lines_in, hc_line_rec* __restrict__ lines_out) {
    n = records_of(n, n_dev);
    const uint64_t i0 = (uint64_t)blockIdx.x * kKeptTile + threadIdx.x * 4u;
    hc_result_rec r[4];
    uint32_t keep[4], c = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        keep[j] = 0;
        if (i0 + j < n) {
            r[j] = res[i0 + j];
            keep[j] = (r[j].n_cls >> 28) != HC_CLS_DROP;
        }
        c += keep[j];
    }
    uint32_t incl = c;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, o, 64);
        if ((int)(threadIdx.x & 63u) >= o) incl += up;
    }
    __shared__ uint32_t wave_sum[4];
    if ((threadIdx.x & 63u) == 63u) wave_sum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint64_t at = (uint64_t)tile_off[blockIdx.x] + incl - c;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) at += wave_sum[w];
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (keep[j]) {
            if (at < cap) {
                hc_gather_row o;
                o.index = base_index + i0 + j;
                o.x1 = r[j].x1;
                o.x2 = r[j].x2;
                o.mm = r[j].mm;
                o.n_cls = r[j].n_cls;
                rows[at] = o;
                if (lines_in) {
                    const uint4* a = (const uint4*)(lines_in + i0 + j);
                    uint4* b = (uint4*)(lines_out + at);
                    b[0] = a[0];
                    b[1] = a[1];
                    b[2] = a[2];
                }
            }
            at++;
        }
}

// rows / lines_out: device buffers of cap records; *count = number of non-dropped records (may exceed cap).
// tile_cnt / tile_off: scratch of (n + 1023) / 1024 + 1 entries each.
hipError_t launch_kept_rows(const hc_result_rec* res, uint64_t n, const unsigned long long* n_dev, uint64_t base_index, uint32_t* tile_cnt,
                            uint32_t* tile_off, hc_gather_row* rows, uint64_t cap, unsigned long long* count, const hc_line_rec* lines_in,
                            hc_line_rec* lines_out, hipStream_t stream, const uint32_t* text_tally, unsigned long long* text_counters) {
    const uint32_t n_tiles = (uint32_t)((n + kKeptTile - 1) / kKeptTile);
    if (n_tiles == 0) return hipMemsetAsync(count, 0, sizeof(unsigned long long), stream);
    hipLaunchKernelGGL(kept_count_kernel, dim3(n_tiles), dim3(256), 0, stream, res, n, n_dev, tile_cnt);
    hipLaunchKernelGGL(scan_tiles_kernel, dim3(1), dim3(1024), 0, stream, tile_cnt, n_tiles, tile_off, count, text_tally, text_counters);
    hipLaunchKernelGGL(kept_scatter_kernel, dim3(n_tiles), dim3(256), 0, stream, res, n, n_dev, tile_off, base_index, rows, cap, lines_in, lines_out);
    return hipGetLastError();
}

// Rows in a DEVICE buffer -> page-locked host memory mapped into the device's address
// space, as one coalesced stream of 16-byte pieces (consecutive lanes write consecutive pieces: full-size PCIe writes;
// scattered 32-byte stores cross PCIe at a fraction of the rate).
__global__ __launch_bounds__(256) void flush_rows_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst,
                                                         const unsigned long long* __restrict__ count, unsigned long long cap,
                                                         uint32_t pieces_per_row) {
    unsigned long long k = *count;
    k = (k < cap ? k : cap) * pieces_per_row;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < k; i += (unsigned long long)gridDim.x * blockDim.x)
        dst[i] = src[i];
}

hipError_t launch_flush_rows(const void* src, void* dst_mapped, const unsigned long long* count, uint64_t cap, uint32_t row_bytes, uint32_t n_cu,
                             hipStream_t stream) {
    hipLaunchKernelGGL(flush_rows_kernel, dim3(n_cu), dim3(256), 0, stream, (const uint4*)src, (uint4*)dst_mapped, count,
                       (unsigned long long)cap, row_bytes / 16u);
    return hipGetLastError();
}

// A text block's two row arrays in one launch, and its counters with them: the first workgroup copies the 16 counters — final by
// now — into the block's page-locked words (a hipMemcpyAsync of 128 bytes per block until round 4).
__global__ __launch_bounds__(256) void flush_text_rows_kernel(const uint4* __restrict__ src_a, uint4* __restrict__ dst_a, uint32_t pieces_a,
                                                              const uint4* __restrict__ src_b, uint4* __restrict__ dst_b, uint32_t pieces_b,
                                                              const unsigned long long* __restrict__ count, unsigned long long cap,
                                                              const unsigned long long* __restrict__ counters,
                                                              unsigned long long* __restrict__ counters_host) {
    unsigned long long k = *count;
    k = k < cap ? k : cap;
    const unsigned long long ka = k * pieces_a, kb = k * pieces_b, stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < ka; i += stride) dst_a[i] = src_a[i];
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < kb; i += stride) dst_b[i] = src_b[i];
    if (blockIdx.x == 0 && threadIdx.x < kTextCounters) counters_host[threadIdx.x] = counters[threadIdx.x];
}

hipError_t launch_flush_text_rows(const void* rows, void* rows_mapped, const void* lines, void* lines_mapped, const unsigned long long* count,
                                  uint64_t cap, const unsigned long long* counters, unsigned long long* counters_mapped, uint32_t n_cu,
                                  hipStream_t stream) {
    hipLaunchKernelGGL(flush_text_rows_kernel, dim3(n_cu), dim3(256), 0, stream, (const uint4*)rows, (uint4*)rows_mapped,
                       (uint32_t)(sizeof(hc_gather_row) / 16u), (const uint4*)lines, (uint4*)lines_mapped, (uint32_t)(sizeof(hc_line_rec) / 16u), count,
                       (unsigned long long)cap, counters, counters_mapped);
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

// hc_comm_gate_device: one lane that leaves when `target` workgroups of the context's cooperative scoring launches have started (or
// after timeout_us).  Put in front of a collective on the exchange's stream, it makes the collective library's kernels arrive when the
// scoring kernel that runs beside them already sits on its CUs (all but the hc_set_comm_reserve'd ones), so they take the free ones —
// arriving first, their workgroups would be spread over CUs the scoring kernel's one-per-CU workgroups then cannot share.  A few
// registers, no LDS: fits on any CU beside anything.  The timeout makes a wait for a launch that never comes end (the exchange is
// then simply not gated).
__global__ __launch_bounds__(64) void comm_gate_kernel(const unsigned long long* __restrict__ started, unsigned long long target, unsigned long long timeout_ticks) {
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(64);
        if (wall_clock64() - t0 > timeout_ticks) break;
    }
}

hipError_t launch_comm_gate(const unsigned long long* started, unsigned long long target, uint32_t timeout_us, hipStream_t stream) {
    int khz = 100000;  // wall_clock64(): the constant-rate counter, 100 MHz on gfx9
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev);
    if (khz <= 0) khz = 100000;
    hipLaunchKernelGGL(comm_gate_kernel, dim3(1), dim3(64), 0, stream, started, target, (unsigned long long)timeout_us * (unsigned long long)khz / 1000ull);
    return hipGetLastError();
}

}  // namespace hc
