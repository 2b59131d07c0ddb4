// hc_graph_kernels.hip — the serial half of process_overlaps on the device (gfx950), for a whole overlaps file at
// once into an empty graph (reference src/EdgeCalculator.cpp:431-545, src/OverlapGraph.cpp:94-101,
// :722-764):
//   admitted candidates in sequence order
//     -> edge_build_kernel      the tail of compute_overlap (pos3/pos4, lengths, vertices; :219-232, :254-270,
//                               :292-308, :353-379), the normalisation of :443-448, the slot key
//     -> stable radix sort      by slot key = (smaller vertex, larger vertex, ori1 == ori2): sequence order survives
//                               inside a slot
//     -> slot_replay_kernel     one lane per slot replays the replace / keep decisions in sequence order (score,
//                               then the tie-break chain :470-521); marks the survivor, counts `doubles`, marks
//                               OverlapGraph::inclusions from the record inserted FIRST (:459-468)
//     -> select + stable sorts  survivors in sequence order -> adjacency lists (CSR) in the order the reference's
//                               addEdge calls leave them, or in the order sortEdges() would re-order them to
// All of it is HBM-bound integer work on a few per cent of the candidates; the radix sorts and selections are
// hc_prims.hip's.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "../../include/hcedge.h"
#include "hc_device.h"
#include "hc_graph.h"
#include "hc_prims.h"
#include "hc_fno_device.h"

namespace hc {

__device__ __forceinline__ ReadDesc g_load_desc(const ReadDesc* p) {
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

// counters: [0] inclusion_count, [1] dup_count, [2] slots, [3] tied lists, [4] first bad record (min index)
__global__ __launch_bounds__(256) void edge_build_kernel(GraphParams gp, const hc_admit_rec* __restrict__ A, uint32_t m,
                                                         hc_edge_rec* __restrict__ E, uint64_t* __restrict__ key,
                                                         uint32_t* __restrict__ idx, unsigned long long* __restrict__ counters) {
    const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t incl = 0;
    if (a < m) {
        const hc_admit_rec o = A[a];
        hc_edge_rec e;
        bool bad = o.read1 >= gp.n_reads || o.read2 >= gp.n_reads;
        uint64_t v1 = 0, v2 = 0;
        int pos3 = 0, pos4 = 0;
        bool ss = true;
        if (!bad) {
            const ReadDesc d1 = g_load_desc(gp.reads + o.read1), d2 = g_load_desc(gp.reads + o.read2);
            const bool p1 = d1.flags & kReadPaired, p2 = d2.flags & kReadPaired;
            ss = !p1 && !p2;
            const int pos1 = (int)o.pos1, pos2 = (int)o.pos2;
            const int a1 = (int)d1.len1, b1 = (int)d1.len2, a2 = (int)d2.len1, b2 = (int)d2.len2;
            if (!p1 && !p2) {
                pos3 = a1 - pos1 - a2;  // :222
            } else if (!p1 && p2) {
                pos3 = a1 - pos2 - b2;  // :262
                pos4 = a1 - pos1 - a2;  // :263
            } else if (p1 && !p2) {
                pos3 = b1 + pos2 - a2;  // :300
                pos4 = a2 + pos1 - a1;  // :301
            } else {
                pos3 = o.ord == '1' ? b1 - pos2 - b2   // :363
                                    : b1 + pos2 - b2;  // :370
                pos4 = a1 - pos1 - a2;                 // :372
            }
            v1 = gp.vtx ? gp.vtx[o.read1] : o.read1;
            v2 = gp.vtx ? gp.vtx[o.read2] : o.read2;
            bad = v1 >= gp.n_vertices || v2 >= gp.n_vertices;
        }
        const int len1 = (int)o.len1, len2 = ss ? 0 : (int)o.len2;  // :227 / :268
        if (!(len1 > 0) || !(len2 >= 0)) bad = true;                // Edge::set_len, src/Edge.h:211-218
        if (!(o.score == 0 || o.score == -1 || o.score > 0)) bad = true;  // Edge's constructor, src/Edge.h:43-57
        e.score = o.score;
        e.mismatch_rate = (double)(float)o.mm / (double)o.n;  // :132
        e.pos1 = (int)o.pos1;
        e.pos2 = (int)o.pos2;
        e.pos3 = pos3;
        e.pos4 = pos4;
        e.ori1 = o.ori1 ? 1 : 0;
        e.ori2 = o.ori2 ? 1 : 0;
        e.ord = o.ord;
        e.pad = 0;
        e.read1 = o.read1;
        e.read2 = o.read2;
        e.v1 = v1;
        e.v2 = v2;
        e.perc = (int)o.perc;
        e.len0 = len1 + len2;
        e.len1 = len1;
        e.len2 = len2;
        if (e.pos1 == 0 && e.v1 > e.v2) {  // :443-448, Edge::swap_reads (src/Edge.h:74-88)
            const uint32_t r = e.read1; e.read1 = e.read2; e.read2 = r;
            const uint64_t v = e.v1; e.v1 = e.v2; e.v2 = v;
            const uint8_t t = e.ori1; e.ori1 = e.ori2; e.ori2 = t;
            if (e.ord == '1') e.ord = '2';
            else if (e.ord == '2') e.ord = '1';
            e.pos3 = -e.pos3;
            e.pos4 = -e.pos4;
        }
        if (e.perc == 100) incl = 1;  // :449-451, before de-duplication
        E[a] = e;
        const uint64_t lo = e.v1 < e.v2 ? e.v1 : e.v2, hi = e.v1 < e.v2 ? e.v2 : e.v1;
        key[a] = (lo << 33) | (hi << 1) | (uint64_t)(e.ori1 == e.ori2);
        idx[a] = a;
        if (bad) atomicMin(&counters[4], (unsigned long long)a);
    }
    const unsigned long long n_incl = __popcll(__ballot(incl != 0));
    if ((threadIdx.x & 63u) == 0 && n_incl) atomicAdd(&counters[0], n_incl);
}

// true when the reference keeps the existing edge although the new one scores as high (:474-521; all equal: replace)
__device__ __forceinline__ bool chain_keeps_existing(const hc_edge_rec& ex, const hc_edge_rec& e) {
    if (ex.len0 != e.len0) return ex.len0 > e.len0;
    if (ex.mismatch_rate != e.mismatch_rate) return ex.mismatch_rate < e.mismatch_rate;
    if (ex.v1 != e.v1) return ex.v1 < e.v1;
    if (ex.ori1 != e.ori1) return ex.ori1 != 0;
    if (ex.ori2 != e.ori2) return ex.ori2 != 0;
    if (ex.pos1 != e.pos1) return ex.pos1 < e.pos1;
    if (ex.pos2 != e.pos2) return ex.pos2 < e.pos2;
    return false;
}

// One lane per sorted position; the lane at the head of a slot's run walks the run (runs are short: a slot holds the
// duplicates of one read pair in one orientation class).
__global__ __launch_bounds__(256) void slot_replay_kernel(GraphParams gp, const hc_edge_rec* __restrict__ E,
                                                          const uint64_t* __restrict__ key_s, const uint32_t* __restrict__ idx_s,
                                                          uint32_t m, uint8_t* __restrict__ keep, uint8_t* __restrict__ inclusions,
                                                          unsigned long long* __restrict__ counters) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long dups = 0;
    uint32_t head = 0;
    if (p < m) {
        const uint64_t k = key_s[p];
        if (p == 0 || key_s[p - 1] != k) {
            head = 1;
            uint32_t cur_i = idx_s[p];
            hc_edge_rec cur = E[cur_i];
            // the only record that is inserted into an empty slot: :455-469
            if (gp.ignore_inclusions && cur.perc == 100 && cur.mismatch_rate < 0.000001 && cur.mismatch_rate >= 0) {
                if (cur.pos3 < 0) {
                    if (cur.pos1 == 0) inclusions[cur.v1] = 1;  // otherwise only an effect of rounding the percentage
                } else {
                    inclusions[cur.v2] = 1;
                }
            }
            for (uint32_t q = p + 1; q < m && key_s[q] == k; q++) {  // the later records of the slot, in sequence order
                dups++;
                const uint32_t i = idx_s[q];
                const hc_edge_rec e = E[i];
                if (!(e.score >= cur.score)) continue;                                 // :535-538
                if (e.score == cur.score && chain_keeps_existing(cur, e)) continue;   // :474-521
                cur = e;                                                               // :523-530
                cur_i = i;
            }
            keep[cur_i] = 1;
        }
    }
    // block-level tallies: two atomics per wave at most
    const unsigned long long heads = __popcll(__ballot(head != 0));
    for (int off = 32; off > 0; off >>= 1) dups += __shfl_down(dups, off, 64);
    if ((threadIdx.x & 63u) == 0) {
        if (dups) atomicAdd(&counters[1], dups);
        if (heads) atomicAdd(&counters[2], heads);
    }
}

// Keys of the adjacency orders.  S: survivors (admitted indices) in the order to be refined.
//   mode 0: key32 = vertex1                       (out-lists, insertion order)
//   mode 1: key32 = vertex2                       (in-lists; first pass of the sortEdges order)
//   mode 2: key64 = vertex1 << 32 | non-overlap   (second pass of the sortEdges order; src/Edge.h:58-63: unsigned)
__global__ __launch_bounds__(256) void order_keys_kernel(GraphParams gp, const hc_edge_rec* __restrict__ E,
                                                         const uint32_t* __restrict__ S, uint32_t n, int mode,
                                                         uint32_t* __restrict__ key32, uint64_t* __restrict__ key64) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const hc_edge_rec& e = E[S[k]];
    if (mode == 0) key32[k] = (uint32_t)e.v1;
    else if (mode == 1) key32[k] = (uint32_t)e.v2;
    else {
        const ReadDesc d1 = g_load_desc(gp.reads + e.read1), d2 = g_load_desc(gp.reads + e.read2);
        const uint32_t l1 = d1.len1 + ((d1.flags & kReadPaired) ? d1.len2 : 0u);  // Read::get_len, src/Read.h:203-212
        const uint32_t l2 = d2.len1 + ((d2.flags & kReadPaired) ? d2.len2 : 0u);
        const uint32_t nonoverlap = l1 + l2 - 2u * (uint32_t)e.len0;
        key64[k] = ((uint64_t)(uint32_t)e.v1 << 32) | nonoverlap;
    }
}

// off[v] = number of sorted keys < v, for v in [0, V]: the CSR offsets of a sorted key column
template <typename K>
__global__ __launch_bounds__(256) void offsets_kernel(const K* __restrict__ keys, uint32_t n, int shift, uint32_t V,
                                                      unsigned long long* __restrict__ off) {
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v > V) return;
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if ((uint64_t)(keys[mid] >> shift) < (uint64_t)v) lo = mid + 1;
        else hi = mid;
    }
    off[v] = lo;
}

__global__ __launch_bounds__(256) void gather_edges_kernel(const hc_edge_rec* __restrict__ E, const uint32_t* __restrict__ O,
                                                           uint32_t n, hc_edge_rec* __restrict__ out) {
    // 80-byte records as five 16-byte pieces: consecutive lanes write consecutive pieces
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t k = t / 5, piece = t - 5 * k;
    if (k >= n) return;
    ((uint4*)out)[t] = ((const uint4*)(E + O[k]))[piece];
}

__global__ __launch_bounds__(256) void in_nodes_kernel(const hc_edge_rec* __restrict__ E, const uint32_t* __restrict__ O,
                                                       uint32_t n, uint32_t* __restrict__ in_nodes) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) in_nodes[k] = (uint32_t)E[O[k]].v1;
}

// sortEdges order: an out-list longer than 16 (std::sort leaves the insertion sort it uses below that length, which
// is stable) holding two edges that compare equal (same non-overlap length, same vertex2) — the reference's order of
// those depends on its introsort and on the insertion order.  Such lists are reported, not guessed.
__global__ __launch_bounds__(256) void tied_lists_kernel(const uint64_t* __restrict__ key64_s, const uint32_t* __restrict__ v2_s,
                                                         uint32_t n, const unsigned long long* __restrict__ out_off,
                                                         uint8_t* __restrict__ tied) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k == 0 || k >= n) return;
    if (key64_s[k] != key64_s[k - 1] || v2_s[k] != v2_s[k - 1]) return;
    const uint32_t v = (uint32_t)(key64_s[k] >> 32);
    if (out_off[v + 1] - out_off[v] > 16) tied[v] = 1;
}

__global__ __launch_bounds__(256) void gather_u32_kernel(const hc_edge_rec* __restrict__ E, const uint32_t* __restrict__ O, uint32_t n,
                                                         uint32_t* __restrict__ v2) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) v2[k] = (uint32_t)E[O[k]].v2;
}

// ---------------------------------------------------------------------------------------------------------------
// host-side launchers (hc_api.cpp owns the buffers); sorting and selection: hc_prims.hip

// the (64-bit key, 32-bit value) radix sort for other translation units (find-next-overlaps); temp == nullptr: the scratch size
hipError_t sort_pairs_u64_u32(void* temp, size_t& temp_bytes, const uint64_t* k_in, uint64_t* k_out, const uint32_t* v_in, uint32_t* v_out,
                              uint32_t n, int end_bit, hipStream_t s) {
    if (!temp) {
        temp_bytes = prims::sort_temp_bytes(n, sizeof(uint64_t), sizeof(uint32_t));
        return hipSuccess;
    }
    return prims::sort_pairs(temp, temp_bytes, k_in, k_out, v_in, v_out, n, 0, end_bit, s);
}

size_t graph_temp_bytes(uint32_t m, uint32_t V) {
    size_t best = prims::sort_temp_bytes(m, sizeof(uint64_t), sizeof(uint32_t));
    best = std::max(best, prims::sort_temp_bytes(m, sizeof(uint32_t), sizeof(uint32_t)));
    best = std::max(best, prims::select_temp_bytes(m > V ? m : V));
    return best;
}

hipError_t graph_build_and_replay(const GraphParams& gp, const hc_admit_rec* A, uint32_t m, hc_edge_rec* E, uint64_t* key0,
                                  uint64_t* key1, uint32_t* idx0, uint32_t* idx1, uint8_t* keep, uint8_t* inclusions,
                                  unsigned long long* counters, uint32_t* survivors, unsigned long long* d_count, void* temp,
                                  size_t temp_bytes, hipStream_t s) {
    if (m == 0) return hipSuccess;
    const dim3 grid((m + 255) / 256), block(256);
    hipLaunchKernelGGL(edge_build_kernel, grid, block, 0, s, gp, A, m, E, key0, idx0, counters);
    int vbits = 1;
    while (vbits < 31 && (gp.n_vertices >> vbits)) vbits++;
    hipError_t e = prims::sort_pairs(temp, temp_bytes, key0, key1, idx0, idx1, m, 0, 33 + vbits, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(slot_replay_kernel, grid, block, 0, s, gp, E, key1, idx1, m, keep, inclusions, counters);
    return prims::select_flagged(temp, temp_bytes, keep, m, survivors, d_count, s);
}

// survivors (sequence order, n of them) -> O_out (CSR order of adj_out), out_off, O_in (CSR order of adj_in), in_off
hipError_t graph_orders(const GraphParams& gp, const hc_edge_rec* E, const uint32_t* survivors, uint32_t n, uint32_t order,
                        uint32_t* k32a, uint32_t* k32b, uint64_t* k64a, uint64_t* k64b, uint32_t* tmp_idx, uint32_t* O_out,
                        uint32_t* O_in, unsigned long long* out_off, unsigned long long* in_off, uint8_t* tied, void* temp,
                        size_t temp_bytes, hipStream_t s) {
    const uint32_t V = gp.n_vertices;
    const dim3 vgrid((V + 1 + 255) / 256), block(256);
    if (n == 0) {
        hipError_t e = hipMemsetAsync(out_off, 0, (size_t)(V + 1) * 8, s);
        if (e != hipSuccess) return e;
        return hipMemsetAsync(in_off, 0, (size_t)(V + 1) * 8, s);
    }
    const dim3 grid((n + 255) / 256);
    int vbits = 1;
    while (vbits < 32 && ((uint64_t)V >> vbits)) vbits++;
    hipError_t e;
    if (order == HC_GRAPH_INSERTION_ORDER) {
        // adj_out[v]: survivors with vertex1 == v in sequence order; adj_in[w]: vertex1 of survivors with vertex2 == w
        // in sequence order (addEdge appends to both lists, OverlapGraph.cpp:94-101)
        hipLaunchKernelGGL(order_keys_kernel, grid, block, 0, s, gp, E, survivors, n, 0, k32a, (uint64_t*)nullptr);
        if ((e = prims::sort_pairs(temp, temp_bytes, k32a, k32b, survivors, O_out, n, 0, vbits, s)) != hipSuccess) return e;
        hipLaunchKernelGGL(offsets_kernel<uint32_t>, vgrid, block, 0, s, k32b, n, 0, V, out_off);
        hipLaunchKernelGGL(order_keys_kernel, grid, block, 0, s, gp, E, survivors, n, 1, k32a, (uint64_t*)nullptr);
        if ((e = prims::sort_pairs(temp, temp_bytes, k32a, k32b, survivors, O_in, n, 0, vbits, s)) != hipSuccess) return e;
        hipLaunchKernelGGL(offsets_kernel<uint32_t>, vgrid, block, 0, s, k32b, n, 0, V, in_off);
        return hipGetLastError();
    }
    // sortEdges: out-lists by (non-overlap length, vertex2), ties in sequence order (two stable LSD passes); adj_in
    // rebuilt by walking the sorted out-lists in vertex order (:751-762) = a stable sort of that order by vertex2
    hipLaunchKernelGGL(order_keys_kernel, grid, block, 0, s, gp, E, survivors, n, 1, k32a, (uint64_t*)nullptr);
    if ((e = prims::sort_pairs(temp, temp_bytes, k32a, k32b, survivors, tmp_idx, n, 0, vbits, s)) != hipSuccess) return e;
    hipLaunchKernelGGL(order_keys_kernel, grid, block, 0, s, gp, E, tmp_idx, n, 2, (uint32_t*)nullptr, k64a);
    if ((e = prims::sort_pairs(temp, temp_bytes, k64a, k64b, tmp_idx, O_out, n, 0, 32 + vbits, s)) != hipSuccess) return e;
    hipLaunchKernelGGL(offsets_kernel<uint64_t>, vgrid, block, 0, s, k64b, n, 32, V, out_off);
    hipLaunchKernelGGL(gather_u32_kernel, grid, block, 0, s, E, O_out, n, k32a);  // vertex2 in sorted out-order
    hipLaunchKernelGGL(tied_lists_kernel, grid, block, 0, s, k64b, k32a, n, out_off, tied);
    if ((e = prims::sort_pairs(temp, temp_bytes, k32a, k32b, O_out, O_in, n, 0, vbits, s)) != hipSuccess) return e;
    hipLaunchKernelGGL(offsets_kernel<uint32_t>, vgrid, block, 0, s, k32b, n, 0, V, in_off);
    return hipGetLastError();
}

hipError_t graph_select_tied(const uint8_t* tied, uint32_t V, uint32_t* out, unsigned long long* d_count, void* temp, size_t temp_bytes,
                             hipStream_t s) {
    return prims::select_flagged(temp, temp_bytes, tied, V, out, d_count, s);
}

hipError_t graph_gather(const hc_edge_rec* E, const uint32_t* O_out, const uint32_t* O_in, uint32_t n, hc_edge_rec* edges_out,
                        uint32_t* in_nodes, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const uint64_t pieces = (uint64_t)n * 5;
    hipLaunchKernelGGL(gather_edges_kernel, dim3((uint32_t)((pieces + 255) / 256)), dim3(256), 0, s, E, O_out, n, edges_out);
    hipLaunchKernelGGL(in_nodes_kernel, dim3((n + 255) / 256), dim3(256), 0, s, E, O_in, n, in_nodes);
    return hipGetLastError();
}

}  // namespace hc
