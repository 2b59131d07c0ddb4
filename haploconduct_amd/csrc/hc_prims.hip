// hc_prims.hip — device-wide primitives for gfx950 (MI355X, wave64), hand-written: stable LSD radix sort, exclusive prefix
// sum, ordered selection, unique.  They serve the kernels around the scoring kernel — duplicate resolution and adjacency
// (hc_graph_kernels.hip), the overlap finder, the SFO ingest and find-next-overlaps — all HBM-bound integer work.
//
// Radix sort, one pass (8 bits) = three launches, no spinning between workgroups (nothing here can hang):
//   upsweep    workgroup b counts the digits of its contiguous range of tiles   -> counts[b][digit]
//   groups     one workgroup per 32 consecutive ranges: counts[b][digit] becomes the prefix inside the group, the group's
//              sums go to sums[group][digit] (no single workgroup walks the whole table: with 1 024 ranges that took
//              0.4 ms a pass, five times the pass itself for 10^7 items)
//   downsweep  workgroup b first adds up what is in front of it — the digit's total over all groups below the digit, the
//              sums of the groups in front of its own, its prefix inside the group: at most 32 + 1 rows of 1 KiB — then
//              walks its tiles in order; per tile: every wave ranks its items by digit with eight ballots (lanes holding the
//              same digit find each other; the lowest of them bumps the wave's digit counter in LDS once for all), a
//              256-lane scan over the digits places the tile in LDS sorted by digit, and the tile leaves as runs of
//              consecutive addresses.  Items keep their order within a digit (stable): wave-striped loads make index
//              order = (wave, item, lane) order, which is the order the ranks are dealt in.
// Bytes per item and pass: key read twice, written once; value read and written once.
#include "hc_prims.h"

#include <algorithm>

namespace hc {
namespace prims {

namespace {

constexpr int kThreads = 256;
constexpr int kIpt = 8;                       // items per thread and tile
constexpr uint32_t kTile = kThreads * kIpt;   // 2 048
constexpr uint32_t kMaxBlocks = 1024;         // workgroups of a pass (each walks a contiguous range of tiles)

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, uint32_t lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)v, o, 64);
        if ((int)lane >= o) v += up;
    }
    return v;
}

// ---------------------------------------------------------------------------------------------------------------
// In-place exclusive scan of M entries by ONE workgroup: every lane sums a contiguous chunk, the 1 024 chunk sums are
// scanned in LDS, every lane rewrites its chunk.  *total (may be null) receives the sum of all.
template <typename T>
__global__ __launch_bounds__(1024) void spine_scan_kernel(T* __restrict__ data, uint64_t M, unsigned long long* __restrict__ total) {
    __shared__ T wave_sum[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint64_t chunk = (M + 1023) / 1024;
    const uint64_t lo = (uint64_t)tid * chunk, hi = lo + chunk < M ? lo + chunk : M;
    T sum = 0;
    for (uint64_t i = lo; i < hi; ++i) sum += data[i];
    T incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const T up = __shfl_up(incl, o, 64);
        if ((int)lane >= o) incl += up;
    }
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    T before = 0;
    for (uint32_t w = 0; w < wave; ++w) before += wave_sum[w];
    T run = before + incl - sum;
    for (uint64_t i = lo; i < hi; ++i) {
        const T v = data[i];
        data[i] = run;
        run += v;
    }
    if (total && tid == 1023) *total = (unsigned long long)(before + incl);
}

// ---------------------------------------------------------------------------------------------------------------
// Radix sort
template <typename K>
__global__ __launch_bounds__(kThreads) void radix_upsweep_kernel(const K* __restrict__ keys, uint64_t n, int shift, uint32_t mask,
                                                                 uint32_t tiles_per_block, uint32_t G, uint32_t* __restrict__ counts) {
    __shared__ uint32_t h[4][256];
    const uint32_t tid = threadIdx.x, wave = tid >> 6;
#pragma unroll
    for (int w = 0; w < 4; ++w) h[w][tid] = 0;
    __syncthreads();
    const uint64_t lo = (uint64_t)blockIdx.x * tiles_per_block * kTile;
    uint64_t hi = lo + (uint64_t)tiles_per_block * kTile;
    hi = hi < n ? hi : n;
    uint64_t i = lo + tid;
    for (; i + 3 * kThreads < hi; i += 4 * kThreads) {  // four loads in flight
        const K k0 = keys[i], k1 = keys[i + kThreads], k2 = keys[i + 2 * kThreads], k3 = keys[i + 3 * kThreads];
        atomicAdd(&h[wave][(uint32_t)(k0 >> shift) & mask], 1u);
        atomicAdd(&h[wave][(uint32_t)(k1 >> shift) & mask], 1u);
        atomicAdd(&h[wave][(uint32_t)(k2 >> shift) & mask], 1u);
        atomicAdd(&h[wave][(uint32_t)(k3 >> shift) & mask], 1u);
    }
    for (; i < hi; i += kThreads) atomicAdd(&h[wave][(uint32_t)(keys[i] >> shift) & mask], 1u);
    __syncthreads();
    counts[(uint64_t)blockIdx.x * 256 + tid] = h[0][tid] + h[1][tid] + h[2][tid] + h[3][tid];
}

constexpr uint32_t kGroup = 32;  // ranges per group of the offset table
__global__ __launch_bounds__(256) void radix_groups_kernel(uint32_t* __restrict__ counts, uint32_t G, uint32_t* __restrict__ sums) {
    const uint32_t d = threadIdx.x, b0 = blockIdx.x * kGroup;
    uint32_t v[kGroup];
#pragma unroll
    for (uint32_t j = 0; j < kGroup; ++j) v[j] = b0 + j < G ? counts[(uint64_t)(b0 + j) * 256 + d] : 0u;
    uint32_t run = 0;
#pragma unroll
    for (uint32_t j = 0; j < kGroup; ++j) {
        if (b0 + j < G) counts[(uint64_t)(b0 + j) * 256 + d] = run;
        run += v[j];
    }
    sums[(uint64_t)blockIdx.x * 256 + d] = run;
}

template <typename K, typename V, bool HAS_V>
__global__ __launch_bounds__(kThreads) void radix_downsweep_kernel(const K* __restrict__ k_in, K* __restrict__ k_out, const V* __restrict__ v_in,
                                                                   V* __restrict__ v_out, uint64_t n, int shift, uint32_t mask, uint32_t tiles_per_block,
                                                                   uint32_t G, const uint32_t* __restrict__ offsets /* prefix inside the group */,
                                                                   const uint32_t* __restrict__ sums /* [groups][256] */) {
    __shared__ K skeys[kTile];
    __shared__ V svals[HAS_V ? kTile : 1];
    __shared__ uint32_t wcount[4][256];  // per wave: running count of a digit while ranking, then the wave's offset inside the digit
    __shared__ uint32_t tstart[256];     // where a digit starts in the tile's sorted image
    __shared__ uint32_t delta[256];      // global start of the digit for this tile minus tstart (mod 2^32)
    __shared__ uint32_t gbase[256];      // global position of this workgroup's next item of a digit
    __shared__ uint32_t wsum[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    {  // where this workgroup's first item of digit tid goes
        const uint32_t groups = (G + kGroup - 1) / kGroup, mine = blockIdx.x / kGroup;
        uint32_t total = 0, before = 0;
        for (uint32_t g = 0; g < groups; ++g) {
            const uint32_t v = sums[(uint64_t)g * 256 + tid];
            total += v;
            before += g < mine ? v : 0u;
        }
        const uint32_t incl = wave_incl_scan(total, lane);
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t lower = incl - total;  // items of smaller digits
        for (uint32_t w = 0; w < wave; ++w) lower += wsum[w];
        gbase[tid] = lower + before + offsets[(uint64_t)blockIdx.x * 256 + tid];
        __syncthreads();  // wsum is used again per tile
    }
    const uint64_t lo = (uint64_t)blockIdx.x * tiles_per_block * kTile;
    uint64_t hi = lo + (uint64_t)tiles_per_block * kTile;
    hi = hi < n ? hi : n;
    const uint64_t below = (1ull << lane) - 1ull;
    for (uint64_t tile_lo = lo; tile_lo < hi; tile_lo += kTile) {
        const uint32_t tile_n = hi - tile_lo < kTile ? (uint32_t)(hi - tile_lo) : kTile;
#pragma unroll
        for (int w = 0; w < 4; ++w) wcount[w][tid] = 0;
        K key[kIpt];
        V val[HAS_V ? kIpt : 1];
        uint32_t rank[kIpt];
        const uint32_t first = wave * 64u * kIpt + lane;  // index in the tile of this lane's item 0; item j is 64 j further
#pragma unroll
        for (int j = 0; j < kIpt; ++j) {
            const uint32_t at = first + 64u * j;
            // beyond the end: all ones, i.e. the largest digit, and — being last in index order — behind every real item of it
            key[j] = at < tile_n ? k_in[tile_lo + at] : (K)~(K)0;
            if (HAS_V) val[j] = at < tile_n ? v_in[tile_lo + at] : (V)0;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kIpt; ++j) {
            const uint32_t d = (uint32_t)(key[j] >> shift) & mask;
            uint64_t peers = ~0ull;  // the lanes of this wave whose item j has the same digit
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const bool bit = (d >> b) & 1u;
                const uint64_t bal = __ballot(bit);
                peers &= bit ? bal : ~bal;
            }
            const uint32_t before = (uint32_t)__popcll(peers & below);
            uint32_t old = 0;
            if (before == 0) old = atomicAdd(&wcount[wave][d], (uint32_t)__popcll(peers));  // the lowest lane of the group, once for all
            old = (uint32_t)__shfl((int)old, __ffsll((unsigned long long)peers) - 1, 64);
            rank[j] = old + before;
        }
        __syncthreads();
        {  // lane tid owns digit tid: offsets of the waves inside the digit, the digit's place in the tile, its place in the output
            const uint32_t c0 = wcount[0][tid], c1 = wcount[1][tid], c2 = wcount[2][tid], c3 = wcount[3][tid];
            const uint32_t total = c0 + c1 + c2 + c3;
            wcount[0][tid] = 0;
            wcount[1][tid] = c0;
            wcount[2][tid] = c0 + c1;
            wcount[3][tid] = c0 + c1 + c2;
            const uint32_t incl = wave_incl_scan(total, lane);
            if (lane == 63) wsum[wave] = incl;
            __syncthreads();
            uint32_t prior = 0;
            for (uint32_t w = 0; w < wave; ++w) prior += wsum[w];
            const uint32_t excl = prior + incl - total;
            tstart[tid] = excl;
            delta[tid] = gbase[tid] - excl;
            gbase[tid] += total;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kIpt; ++j) {
            const uint32_t d = (uint32_t)(key[j] >> shift) & mask;
            const uint32_t p = tstart[d] + wcount[wave][d] + rank[j];
            skeys[p] = key[j];
            if (HAS_V) svals[p] = val[j];
        }
        __syncthreads();
        for (uint32_t i = tid; i < tile_n; i += kThreads) {
            const K k = skeys[i];
            const uint32_t d = (uint32_t)(k >> shift) & mask;
            const uint64_t g = (uint64_t)(uint32_t)(i + delta[d]);
            k_out[g] = k;
            if (HAS_V) v_out[g] = svals[i];
        }
        __syncthreads();  // the image and the counters are reused by the next tile
    }
}

struct SortPlan {
    uint32_t tiles_per_block, G;
    size_t counts_bytes, keys_off, vals_off, total;
};
SortPlan plan_sort(uint64_t n, size_t kb, size_t vb) {
    SortPlan p;
    const uint64_t tiles = (n + kTile - 1) / kTile;
    p.tiles_per_block = (uint32_t)std::max<uint64_t>(1, (tiles + kMaxBlocks - 1) / kMaxBlocks);
    p.G = (uint32_t)std::max<uint64_t>(1, (tiles + p.tiles_per_block - 1) / p.tiles_per_block);
    p.counts_bytes = (((size_t)256 * (p.G + (p.G + kGroup - 1) / kGroup) * sizeof(uint32_t)) + 255) & ~(size_t)255;  // counts, then the groups' sums
    p.keys_off = p.counts_bytes;
    p.vals_off = p.keys_off + ((n * kb + 255) & ~(size_t)255);
    p.total = p.vals_off + ((n * vb + 255) & ~(size_t)255);
    return p;
}

template <typename K, typename V, bool HAS_V>
hipError_t radix_sort(void* temp, size_t temp_bytes, const K* k_in, K* k_out, const V* v_in, V* v_out, uint64_t n, int begin_bit, int end_bit,
                      hipStream_t s) {
    if (n == 0) return hipSuccess;
    if (n >= (1ull << 32) || begin_bit < 0 || end_bit > (int)(8 * sizeof(K)) || end_bit < begin_bit) return hipErrorInvalidValue;
    const SortPlan p = plan_sort(n, sizeof(K), HAS_V ? sizeof(V) : 0);
    if (!temp || temp_bytes < p.total) return hipErrorInvalidValue;
    const int passes = (end_bit - begin_bit + 7) / 8;
    hipError_t e;
    if (passes == 0) {
        if ((e = hipMemcpyAsync(k_out, k_in, n * sizeof(K), hipMemcpyDeviceToDevice, s)) != hipSuccess) return e;
        if (HAS_V && (e = hipMemcpyAsync(v_out, v_in, n * sizeof(V), hipMemcpyDeviceToDevice, s)) != hipSuccess) return e;
        return hipSuccess;
    }
    uint32_t* counts = (uint32_t*)temp;
    uint32_t* sums = counts + (size_t)256 * p.G;
    K* tk = (K*)((char*)temp + p.keys_off);
    V* tv = (V*)((char*)temp + p.vals_off);
    const K* sk = k_in;
    const V* sv = v_in;
    for (int pass = 0; pass < passes; ++pass) {
        const int shift = begin_bit + 8 * pass;
        const int bits = std::min(8, end_bit - shift);
        const uint32_t mask = (1u << bits) - 1u;
        const bool to_out = ((passes - 1 - pass) & 1) == 0;  // the last pass lands in the caller's arrays
        K* dk = to_out ? k_out : tk;
        V* dv = to_out ? v_out : tv;
        hipLaunchKernelGGL((radix_upsweep_kernel<K>), dim3(p.G), dim3(kThreads), 0, s, sk, n, shift, mask, p.tiles_per_block, p.G, counts);
        hipLaunchKernelGGL(radix_groups_kernel, dim3((p.G + kGroup - 1) / kGroup), dim3(256), 0, s, counts, p.G, sums);
        hipLaunchKernelGGL((radix_downsweep_kernel<K, V, HAS_V>), dim3(p.G), dim3(kThreads), 0, s, sk, dk, sv, dv, n, shift, mask, p.tiles_per_block,
                           p.G, (const uint32_t*)counts, (const uint32_t*)sums);
        sk = dk;
        sv = dv;
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// Exclusive sum: tile sums -> spine -> tile-local scan + the tile's start.  Blocked arrangement: lane t owns 8 neighbours.
template <typename T>
__global__ __launch_bounds__(kThreads) void scan_tile_sums_kernel(const T* __restrict__ in, uint64_t n, T* __restrict__ tile_sum) {
    __shared__ T part[4];
    const uint64_t i0 = (uint64_t)blockIdx.x * kTile + threadIdx.x * kIpt;
    T s = 0;
#pragma unroll
    for (int j = 0; j < kIpt; ++j)
        if (i0 + j < n) s += in[i0 + j];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

template <typename T>
__global__ __launch_bounds__(kThreads) void scan_apply_kernel(const T* __restrict__ in, T* __restrict__ out, uint64_t n, const T* __restrict__ tile_start) {
    __shared__ T wsum[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint64_t i0 = (uint64_t)blockIdx.x * kTile + tid * kIpt;
    T v[kIpt];
    T s = 0;
#pragma unroll
    for (int j = 0; j < kIpt; ++j) {
        v[j] = i0 + j < n ? in[i0 + j] : (T)0;
        s += v[j];
    }
    T incl = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const T up = __shfl_up(incl, o, 64);
        if ((int)lane >= o) incl += up;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    T run = tile_start[blockIdx.x] + incl - s;
    for (uint32_t w = 0; w < wave; ++w) run += wsum[w];
#pragma unroll
    for (int j = 0; j < kIpt; ++j) {
        if (i0 + j < n) out[i0 + j] = run;
        run += v[j];
    }
}

// In-place exclusive scan of M entries (tile sums / tile counts), *total = their sum: one workgroup while that is quick
// (M <= 8 192: eight entries per lane), else the three-launch form once more with its own small table behind `spare`.
constexpr uint64_t kSpineMax = 8192;
template <typename T>
size_t spine_spare_bytes(uint64_t M) { return M > kSpineMax ? ((M + kTile - 1) / kTile + 1) * sizeof(T) : 0; }
template <typename T>
hipError_t scan_inplace(T* data, uint64_t M, unsigned long long* total, void* spare, hipStream_t s) {
    if (M <= kSpineMax) {
        hipLaunchKernelGGL((spine_scan_kernel<T>), dim3(1), dim3(1024), 0, s, data, M, total);
        return hipGetLastError();
    }
    const uint64_t tiles = (M + kTile - 1) / kTile;
    if (tiles > kSpineMax) return hipErrorInvalidValue;
    T* tile = (T*)spare;
    hipLaunchKernelGGL((scan_tile_sums_kernel<T>), dim3((uint32_t)tiles), dim3(kThreads), 0, s, (const T*)data, M, tile);
    hipLaunchKernelGGL((spine_scan_kernel<T>), dim3(1), dim3(1024), 0, s, tile, tiles, total);
    hipLaunchKernelGGL((scan_apply_kernel<T>), dim3((uint32_t)tiles), dim3(kThreads), 0, s, (const T*)data, data, M, (const T*)tile);
    return hipGetLastError();
}

template <typename T>
hipError_t exclusive_sum_t(void* temp, size_t temp_bytes, const T* in, T* out, uint64_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const uint64_t tiles = (n + kTile - 1) / kTile;
    if (tiles >= (1ull << 31) || !temp || temp_bytes < tiles * sizeof(T) + spine_spare_bytes<T>(tiles)) return hipErrorInvalidValue;
    T* tile = (T*)temp;
    hipLaunchKernelGGL((scan_tile_sums_kernel<T>), dim3((uint32_t)tiles), dim3(kThreads), 0, s, in, n, tile);
    hipError_t e = scan_inplace<T>(tile, tiles, nullptr, tile + tiles, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((scan_apply_kernel<T>), dim3((uint32_t)tiles), dim3(kThreads), 0, s, in, out, n, (const T*)tile);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// Ordered selection: count per tile -> spine (also the total) -> scatter.  The predicate is evaluated twice.
struct FlagPred {
    const uint8_t* f;
    __device__ __forceinline__ bool operator()(uint64_t i) const { return f[i] != 0; }
};
struct NotDroppedPred {
    const hc_result_rec* res;
    __device__ __forceinline__ bool operator()(uint64_t i) const { return (res[i].n_cls >> 28) != HC_CLS_DROP; }
};
struct RunHeadPred {
    const uint64_t* in;
    __device__ __forceinline__ bool operator()(uint64_t i) const { return i == 0 || in[i] != in[i - 1]; }
};
struct EmitIndex {
    uint32_t* out;
    __device__ __forceinline__ void operator()(uint64_t at, uint64_t i) const { out[at] = (uint32_t)i; }
};
struct EmitValue {
    const uint64_t* in;
    uint64_t* out;
    __device__ __forceinline__ void operator()(uint64_t at, uint64_t i) const { out[at] = in[i]; }
};

template <typename Pred>
__global__ __launch_bounds__(kThreads) void select_count_kernel(Pred pred, uint64_t n, uint32_t* __restrict__ tile_cnt) {
    __shared__ uint32_t part[4];
    const uint64_t i0 = (uint64_t)blockIdx.x * kTile + threadIdx.x * kIpt;
    uint32_t c = 0;
#pragma unroll
    for (int j = 0; j < kIpt; ++j)
        if (i0 + j < n) c += pred(i0 + j) ? 1u : 0u;
    for (int o = 32; o > 0; o >>= 1) c += (uint32_t)__shfl_down((int)c, o, 64);
    if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

template <typename Pred, typename Emit>
__global__ __launch_bounds__(kThreads) void select_scatter_kernel(Pred pred, Emit emit, uint64_t n, const uint32_t* __restrict__ tile_off) {
    __shared__ uint32_t wsum[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint64_t i0 = (uint64_t)blockIdx.x * kTile + tid * kIpt;
    uint32_t keep = 0, c = 0;
#pragma unroll
    for (int j = 0; j < kIpt; ++j)
        if (i0 + j < n && pred(i0 + j)) {
            keep |= 1u << j;
            c++;
        }
    const uint32_t incl = wave_incl_scan(c, lane);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint64_t at = (uint64_t)tile_off[blockIdx.x] + incl - c;
    for (uint32_t w = 0; w < wave; ++w) at += wsum[w];
#pragma unroll
    for (int j = 0; j < kIpt; ++j)
        if (keep & (1u << j)) emit(at++, i0 + j);
}

template <typename Pred, typename Emit>
hipError_t select_t(void* temp, size_t temp_bytes, Pred pred, Emit emit, uint64_t n, unsigned long long* count, hipStream_t s) {
    if (n == 0) return hipMemsetAsync(count, 0, sizeof(unsigned long long), s);
    const uint64_t tiles = (n + kTile - 1) / kTile;
    if (n >= (1ull << 32) || !temp || temp_bytes < tiles * sizeof(uint32_t) + spine_spare_bytes<uint32_t>(tiles)) return hipErrorInvalidValue;
    uint32_t* tile = (uint32_t*)temp;
    hipLaunchKernelGGL((select_count_kernel<Pred>), dim3((uint32_t)tiles), dim3(kThreads), 0, s, pred, n, tile);
    hipError_t e = scan_inplace<uint32_t>(tile, tiles, count, tile + tiles, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((select_scatter_kernel<Pred, Emit>), dim3((uint32_t)tiles), dim3(kThreads), 0, s, pred, emit, n, (const uint32_t*)tile);
    return hipGetLastError();
}

}  // namespace

size_t sort_temp_bytes(uint64_t n, size_t key_bytes, size_t val_bytes) { return plan_sort(n ? n : 1, key_bytes, val_bytes).total; }

hipError_t sort_pairs(void* temp, size_t temp_bytes, const uint32_t* k_in, uint32_t* k_out, const uint32_t* v_in, uint32_t* v_out, uint64_t n,
                      int begin_bit, int end_bit, hipStream_t stream) {
    return radix_sort<uint32_t, uint32_t, true>(temp, temp_bytes, k_in, k_out, v_in, v_out, n, begin_bit, end_bit, stream);
}
hipError_t sort_pairs(void* temp, size_t temp_bytes, const uint64_t* k_in, uint64_t* k_out, const uint32_t* v_in, uint32_t* v_out, uint64_t n,
                      int begin_bit, int end_bit, hipStream_t stream) {
    return radix_sort<uint64_t, uint32_t, true>(temp, temp_bytes, k_in, k_out, v_in, v_out, n, begin_bit, end_bit, stream);
}
hipError_t sort_pairs(void* temp, size_t temp_bytes, const uint64_t* k_in, uint64_t* k_out, const uint64_t* v_in, uint64_t* v_out, uint64_t n,
                      int begin_bit, int end_bit, hipStream_t stream) {
    return radix_sort<uint64_t, uint64_t, true>(temp, temp_bytes, k_in, k_out, v_in, v_out, n, begin_bit, end_bit, stream);
}
hipError_t sort_keys(void* temp, size_t temp_bytes, const uint64_t* k_in, uint64_t* k_out, uint64_t n, int begin_bit, int end_bit,
                     hipStream_t stream) {
    return radix_sort<uint64_t, uint32_t, false>(temp, temp_bytes, k_in, k_out, (const uint32_t*)nullptr, (uint32_t*)nullptr, n, begin_bit, end_bit,
                                                 stream);
}

size_t scan_temp_bytes(uint64_t n, size_t elem_bytes) {
    const uint64_t tiles = (n + kTile - 1) / kTile + 1;
    return (tiles + (tiles + kTile - 1) / kTile + 2) * elem_bytes;
}
hipError_t exclusive_sum(void* temp, size_t temp_bytes, const uint32_t* in, uint32_t* out, uint64_t n, hipStream_t stream) {
    return exclusive_sum_t<uint32_t>(temp, temp_bytes, in, out, n, stream);
}
hipError_t exclusive_sum(void* temp, size_t temp_bytes, const uint64_t* in, uint64_t* out, uint64_t n, hipStream_t stream) {
    return exclusive_sum_t<uint64_t>(temp, temp_bytes, in, out, n, stream);
}

size_t select_temp_bytes(uint64_t n) { return scan_temp_bytes(n, sizeof(uint32_t)); }
hipError_t select_flagged(void* temp, size_t temp_bytes, const uint8_t* flags, uint64_t n, uint32_t* idx_out, unsigned long long* count,
                          hipStream_t stream) {
    return select_t(temp, temp_bytes, FlagPred{flags}, EmitIndex{idx_out}, n, count, stream);
}
hipError_t select_not_dropped(void* temp, size_t temp_bytes, const hc_result_rec* res, uint64_t n, uint32_t* idx_out, unsigned long long* count,
                              hipStream_t stream) {
    return select_t(temp, temp_bytes, NotDroppedPred{res}, EmitIndex{idx_out}, n, count, stream);
}
hipError_t unique(void* temp, size_t temp_bytes, const uint64_t* in, uint64_t* out, unsigned long long* count, uint64_t n, hipStream_t stream) {
    return select_t(temp, temp_bytes, RunHeadPred{in}, EmitValue{in, out}, n, count, stream);
}

}  // namespace prims
}  // namespace hc
