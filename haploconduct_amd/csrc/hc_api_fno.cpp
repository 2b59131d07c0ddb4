// hc_api_fno.cpp — host side of find-next-overlaps' device form (hc_fno_items.h): buffers, the launch sequence of
// hc_fno_kernels.hip, the four stable radix sorts, the copy of the text.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <thread>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/hcedge.h"
#include "hc_fno_device.h"
#include "hc_hostcopy.h"
#include "hc_overlap_finder.h"
#include "hc_prims.h"
#include "host/Types.h"

namespace hc {
namespace {

struct DeviceBuffers {  // freed on every way out
    std::vector<void*> all;
    ~DeviceBuffers() {
        for (void* p : all)
            if (p) (void)hipFree(p);
    }
    void release(void* p) {  // early, for buffers that are spent
        if (!p) return;
        for (void*& q : all)
            if (q == p) {
                (void)hipFree(p);
                q = nullptr;
                return;
            }
    }
    template <typename T>
    T* get(uint64_t count) {
        void* p = nullptr;
        const hipError_t e = hipMalloc(&p, std::max<uint64_t>(count, 1) * sizeof(T));
        if (e != hipSuccess) throw FatalError{HC_ERR_NOMEM, std::string("find-next-overlaps on the device: hipMalloc: ") + hipGetErrorString(e)};
        all.push_back(p);
        return (T*)p;
    }
};
void hip_check(hipError_t e, const char* what) {
    if (e != hipSuccess) throw FatalError{HC_ERR_HIP, std::string("find-next-overlaps on the device: ") + what + ": " + hipGetErrorString(e)};
}


}  // namespace

bool fno_device_wanted(uint64_t n_items) {
    const char* e = getenv("HC_FNO");
    if (e && strcmp(e, "host") == 0) return false;
    const bool forced = e && strcmp(e, "device") == 0;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count < 1) {
        (void)hipGetLastError();
        if (forced) throw FatalError{HC_ERR_NO_DEVICE, "HC_FNO=device: no HIP device"};
        return false;
    }
    return forced || n_items >= 200000;
}

namespace {

// the second half from items that already sit on the device: computeOverlapData, the four stable sorts, unique, scan, text
bool lines_from_device_items(DeviceBuffers& d, const FnoItem* d_items, uint64_t n, bool no_inclusions, const std::function<char*(uint64_t)>& text_of,
                             uint64_t counters[5], double* seconds, std::chrono::steady_clock::time_point t0) {
    auto now = [] { return std::chrono::steady_clock::now(); };
    hipStream_t st = nullptr;  // the default stream of the calling thread's device
    FnoRec* d_rec = d.get<FnoRec>(n);
    uint64_t* d_k[4];
    for (auto& k : d_k) k = d.get<uint64_t>(n);
    uint64_t *d_ka = d.get<uint64_t>(n + 1), *d_kb = d.get<uint64_t>(n + 1);
    uint32_t *d_pa = d.get<uint32_t>(n), *d_pb = d.get<uint32_t>(n);
    unsigned long long* d_counters = d.get<unsigned long long>(kFnoCounters);
    size_t tmp_bytes = 0, scan_bytes = 0;
    hip_check(sort_pairs_u64_u32(nullptr, tmp_bytes, d_ka, d_kb, d_pa, d_pb, (uint32_t)n, 64, st), "sort size");
    hip_check(finder_scan(nullptr, scan_bytes, d_ka, d_kb, n + 1, st), "scan size");
    tmp_bytes = std::max(tmp_bytes, scan_bytes);
    void* d_tmp = d.get<char>(tmp_bytes);
    hip_check(hipMemsetAsync(d_counters, 0, kFnoCounters * sizeof(unsigned long long), st), "memset");
    hip_check(fno_deduce(d_items, n, no_inclusions ? 1u : 0u, d_rec, d_k[0], d_k[1], d_k[2], d_k[3], d_pa, d_counters, st), "deduce");
    // least significant 64 bits first; every sort is stable
    uint32_t *perm = d_pa, *perm_next = d_pb;
    for (int c = 0; c < 4; c++) {
        const uint64_t* keys = d_k[c];
        if (c) {
            hip_check(fno_gather_keys(d_k[c], perm, n, d_ka, st), "gather");
            keys = d_ka;
        }
        size_t b = tmp_bytes;
        hip_check(sort_pairs_u64_u32(d_tmp, b, keys, d_kb, perm, perm_next, (uint32_t)n, 64, st), "sort");
        std::swap(perm, perm_next);
    }
    hip_check(fno_mark_lines(d_rec, perm, n, d_ka, d_counters, st), "mark");
    {
        size_t b = tmp_bytes;
        hip_check(finder_scan(d_tmp, b, d_ka, d_kb, n + 1, st), "scan");
    }
    unsigned long long h_counters[kFnoCounters];
    uint64_t total_bytes = 0;
    hip_check(hipMemcpyAsync(h_counters, d_counters, sizeof h_counters, hipMemcpyDeviceToHost, st), "counters");
    hip_check(hipMemcpyAsync(&total_bytes, d_kb + n, 8, hipMemcpyDeviceToHost, st), "size of the text");
    hip_check(hipStreamSynchronize(st), "synchronize");
    if (h_counters[4]) return false;
    const auto t1 = now();
    for (auto& k : d_k) d.release(k);  // the keys are spent: room for the text
    char* d_text = d.get<char>(total_bytes);
    hip_check(fno_format(d_rec, perm, d_ka, d_kb, n, d_text, st), "format");
    char* h_text = text_of(total_bytes);
    if (total_bytes) hip_check(copy_to_pageable_host(h_text, d_text, total_bytes), "copy of the text");
    for (int k = 0; k < 4; k++) counters[k] = h_counters[k];
    counters[4] = h_counters[5];
    if (seconds) {
        seconds[0] = std::chrono::duration<double>(t1 - t0).count();
        seconds[1] = std::chrono::duration<double>(now() - t1).count();
    }
    return true;
}

}  // namespace

bool fno_lines_on_device(const FnoItem* items, uint64_t n, bool no_inclusions, const std::function<char*(uint64_t)>& text_of,
                         uint64_t counters[5], double* seconds) {
    const auto t0 = std::chrono::steady_clock::now();
    DeviceBuffers d;
    FnoItem* d_items = d.get<FnoItem>(n);
    hip_check(hipMemcpyAsync(d_items, items, n * sizeof(FnoItem), hipMemcpyHostToDevice, nullptr), "copy of the combinations");
    return lines_from_device_items(d, d_items, n, no_inclusions, text_of, counters, seconds, t0);
}

// FNO=1 whole: the walk (updateOverlap's case analysis, the first combination per pair of new ids), the look-ups, then the second
// half above.  false: the device met something the host path has to report or to handle; nothing was written.
bool fno1_walk_on_device(const FnoWalkHost& h, const std::function<char*(uint64_t)>& text_of, uint64_t counters[5], uint64_t* n_items_out,
                         double* seconds) {
    auto now = [] { return std::chrono::steady_clock::now(); };
    const auto t0 = now();
    hipStream_t st = nullptr;
    const bool debug = getenv("HC_FNO_DEBUG") != nullptr;
    auto left = [&](const char* where, unsigned long long status) {  // why the host form takes over (HC_FNO_DEBUG)
        if (debug) fprintf(stderr, "[hc fno] the device route ends at %s, status %llu\n", where, status);
        return false;
    };
    const uint64_t G = h.graph.n, B = h.branching.n, N = h.nonedges.n, I = h.induced.n;
    const uint64_t per_nonedge = h.dup_half ? 2 : 1;  // --add_duplicates: every kept stored non-edge and its opposite
    if (G + B + N + I == 0 || G + B + per_nonedge * N + I >= 0x7FFFFFF0ull || h.n_srs >= 0xFFFFFFFFull || h.n_nodes >= 0xFFFFFFFFull) return false;
    uint32_t id_bits = 1;
    while (id_bits < 64 && (h.new_read_count - 1) >> id_bits) id_bits++;
    if (h.new_read_count == 0 || 2 * id_bits > 62) return false;
    DeviceBuffers d;
    FnoWalkInput w{};
    unsigned long long* d_status = d.get<unsigned long long>(kFnoCounters);
    hip_check(hipMemsetAsync(d_status, 0, kFnoCounters * sizeof(unsigned long long), st), "memset");
    unsigned long long h_status[kFnoCounters];
    auto upload = [&](auto* dst_tag, const void* src, uint64_t count, const char* what) {
        using T = std::remove_pointer_t<decltype(dst_tag)>;
        T* p = d.get<T>(count);
        if (count) hip_check(hipMemcpyAsync(p, src, count * sizeof(T), hipMemcpyHostToDevice, st), what);
        return (const T*)p;
    };
    // the edges in walk order: [adj_out][branching][stored non-edges that pass :702][inclusion-induced]
    hc_fno_edge* d_edges = d.get<hc_fno_edge>(G + B + per_nonedge * N + I);
    w.nodes = upload((hc_fno_read*)nullptr, h.nodes, h.n_nodes, "copy of the vertices");
    w.n_nodes = h.n_nodes;
    if (G) hip_check(hipMemcpyAsync(d_edges, h.graph.p, G * sizeof(hc_fno_edge), hipMemcpyHostToDevice, st), "copy of the edges");
    if (B) hip_check(hipMemcpyAsync(d_edges + G, h.branching.p, B * sizeof(hc_fno_edge), hipMemcpyHostToDevice, st), "copy of the edges");
    uint64_t kept = 0;
    if (N) {
        const hc_fno_edge* d_non = upload((hc_fno_edge*)nullptr, h.nonedges.p, N, "copy of the stored non-edges");
        uint64_t* d_adj = d.get<uint64_t>(h.n_nodes + 1);
        uint8_t* d_keep = d.get<uint8_t>(N);
        uint32_t* d_idx = d.get<uint32_t>(N);
        const size_t sel_bytes = prims::select_temp_bytes(N);
        void* d_sel = d.get<char>(sel_bytes);
        hip_check(fno_adj_offsets(d_edges, G, h.n_nodes, d_adj, d_status, st), "adj_out");
        hip_check(hipMemcpyAsync(h_status, d_status, sizeof h_status, hipMemcpyDeviceToHost, st), "status");
        hip_check(hipStreamSynchronize(st), "synchronize");
        if (h_status[4]) return left("adj_out", h_status[4]);  // unsorted or out of range: the host form
        hip_check(fno_nonedge_filter(d_non, N, d_edges, d_adj, h.n_nodes, d_keep, d_status, st), "stored non-edges");
        hip_check(prims::select_flagged(d_sel, sel_bytes, d_keep, N, d_idx, d_status + 5, st), "select");
        hip_check(hipMemcpyAsync(h_status, d_status, sizeof h_status, hipMemcpyDeviceToHost, st), "status");
        hip_check(hipStreamSynchronize(st), "synchronize");
        if (h_status[4]) return left("the stored non-edges", h_status[4]);
        kept = h_status[5];
        if (h.dup_half) {
            hip_check(fno_gather_mirrored(d_non, d_idx, kept, w.nodes, h.dup_half, d_edges + G + B, d_status, st), "gather");
            hip_check(hipMemcpyAsync(h_status, d_status, sizeof h_status, hipMemcpyDeviceToHost, st), "status");
            hip_check(hipStreamSynchronize(st), "synchronize");
            if (h_status[4]) return left("the opposite overlaps of the stored non-edges", h_status[4]);
            kept *= 2;
        } else {
            hip_check(fno_gather_edges(d_non, d_idx, kept, d_edges + G + B, st), "gather");
            hip_check(hipStreamSynchronize(st), "synchronize");
        }
        d.release((void*)d_non);
        d.release(d_adj);
        d.release(d_keep);
        d.release(d_idx);
        d.release(d_sel);
    }
    if (I) hip_check(hipMemcpyAsync(d_edges + G + B + kept, h.induced.p, I * sizeof(hc_fno_edge), hipMemcpyHostToDevice, st), "copy of the edges");
    const uint64_t E = G + B + kept + I;
    if (E == 0) return false;
    w.edges = d_edges;
    w.n_edges = E;
    w.srs = upload((hc_fno_read*)nullptr, h.srs, h.n_srs, "copy of the super-reads");
    w.n_srs = h.n_srs;
    {  // nodes_to_SR: (vertex, super-read) per clique member, a stable sort by vertex, the offsets
        const uint64_t total = h.n_srs ? h.clique_off[h.n_srs] : 0;
        if (total >= 0x7FFFFFF0ull) return false;
        const uint64_t* d_cn = upload((uint64_t*)nullptr, h.clique_nodes, total, "copy of the cliques");
        const uint64_t* d_co = upload((uint64_t*)nullptr, h.clique_off, h.n_srs + 1, "copy of the cliques");
        uint64_t *d_key = d.get<uint64_t>(total), *d_key_sorted = d.get<uint64_t>(total);
        uint32_t *d_sr = d.get<uint32_t>(total), *d_n2s = d.get<uint32_t>(total);
        uint64_t* d_n2s_off = d.get<uint64_t>(h.n_nodes + 1);
        hip_check(fno_clique_pairs(d_cn, d_co, h.n_srs, total, h.n_nodes, d_key, d_sr, d_status, st), "nodes_to_SR");
        if (total) {
            size_t sort_bytes = 0;
            hip_check(sort_pairs_u64_u32(nullptr, sort_bytes, d_key, d_key_sorted, d_sr, d_n2s, (uint32_t)total, 64, st), "sort size");
            void* d_tmp = d.get<char>(sort_bytes);
            uint32_t node_bits = 1;
            while (node_bits < 64 && (h.n_nodes - 1) >> node_bits) node_bits++;
            hip_check(sort_pairs_u64_u32(d_tmp, sort_bytes, d_key, d_key_sorted, d_sr, d_n2s, (uint32_t)total, (int)node_bits, st), "sort");
            hip_check(hipMemcpyAsync(h_status, d_status, sizeof h_status, hipMemcpyDeviceToHost, st), "status");
            hip_check(hipStreamSynchronize(st), "synchronize");
            if (h_status[4]) return left("nodes_to_SR", h_status[4]);  // a clique names a vertex the graph does not have
            d.release(d_tmp);
        }
        hip_check(fno_offsets(d_key_sorted, total, h.n_nodes, d_n2s_off, st), "nodes_to_SR offsets");
        hip_check(hipStreamSynchronize(st), "synchronize");
        d.release((void*)d_cn);
        d.release((void*)d_co);
        d.release(d_key);
        d.release(d_key_sorted);
        d.release(d_sr);
        w.n2s = d_n2s;
        w.n2s_off = d_n2s_off;
    }
    w.subread_off = upload((uint64_t*)nullptr, h.subread_off, h.n_srs + 1, "copy of the subread maps");
    w.subreads = upload((hc_fno_subread*)nullptr, h.subreads, h.n_srs ? h.subread_off[h.n_srs] : 0, "copy of the subread maps");
    w.new_read_count = h.new_read_count;
    w.id_bits = id_bits;
    w.resolve_orientations = h.resolve_orientations ? 1u : 0u;

    uint64_t *d_off_comb = d.get<uint64_t>(E + 1), *d_off_direct = d.get<uint64_t>(E + 1);
    hip_check(fno_walk_count(w, d_off_comb, d_off_direct, d_status, st), "count");
    size_t scan_bytes = 0;
    hip_check(finder_scan(nullptr, scan_bytes, d_off_comb, d_off_comb, E + 1, st), "scan size");
    void* d_scan_tmp = d.get<char>(scan_bytes);
    {
        size_t b = scan_bytes;
        hip_check(finder_scan(d_scan_tmp, b, d_off_comb, d_off_comb, E + 1, st), "scan");
        b = scan_bytes;
        hip_check(finder_scan(d_scan_tmp, b, d_off_direct, d_off_direct, E + 1, st), "scan");
    }
    uint64_t n_comb = 0, n_direct = 0;
    hip_check(hipMemcpyAsync(&n_comb, d_off_comb + E, 8, hipMemcpyDeviceToHost, st), "count of the combinations");
    hip_check(hipMemcpyAsync(&n_direct, d_off_direct + E, 8, hipMemcpyDeviceToHost, st), "count of the copied edges");
    hip_check(hipMemcpyAsync(h_status, d_status, sizeof h_status, hipMemcpyDeviceToHost, st), "status");
    hip_check(hipStreamSynchronize(st), "synchronize");
    if (h_status[4] || n_comb >= 0x7FFFFFF0ull || n_direct + n_comb >= 0x7FFFFFF0ull) return left("the count of the combinations", h_status[4]);

    uint32_t* d_val_sorted = nullptr;
    uint32_t* d_head_pos = nullptr;
    uint64_t n_heads = 0;
    if (n_comb) {
        uint64_t *d_key = d.get<uint64_t>(n_comb), *d_key_sorted = d.get<uint64_t>(n_comb);
        uint32_t* d_iota = d.get<uint32_t>(n_comb);
        d_val_sorted = d.get<uint32_t>(n_comb);
        hip_check(fno_walk_expand(w, d_off_comb, n_comb, d_key, d_iota, d_status, st), "expand");
        size_t sort_bytes = 0;
        hip_check(sort_pairs_u64_u32(nullptr, sort_bytes, d_key, d_key_sorted, d_iota, d_val_sorted, (uint32_t)n_comb, 64, st), "sort size");
        const size_t sel_bytes = prims::select_temp_bytes(n_comb);
        void* d_tmp = d.get<char>(std::max(sort_bytes, sel_bytes));
        size_t b = sort_bytes;
        hip_check(sort_pairs_u64_u32(d_tmp, b, d_key, d_key_sorted, d_iota, d_val_sorted, (uint32_t)n_comb, (int)(2 * id_bits), st), "sort");
        uint8_t* d_flag = d.get<uint8_t>(n_comb);
        hip_check(fno_walk_heads(d_key_sorted, n_comb, d_flag, st), "heads");
        d_head_pos = d.get<uint32_t>(n_comb);
        hip_check(prims::select_flagged(d_tmp, sel_bytes, d_flag, n_comb, d_head_pos, d_status + 5, st), "select");
        hip_check(hipMemcpyAsync(h_status, d_status, sizeof h_status, hipMemcpyDeviceToHost, st), "status");
        hip_check(hipStreamSynchronize(st), "synchronize");
        if (h_status[4]) return left("the pairs of new ids", h_status[4]);
        n_heads = h_status[5];
        d.release(d_key);
        d.release(d_key_sorted);
        d.release(d_iota);
        d.release(d_flag);
        d.release(d_tmp);
    }
    const uint64_t n_items = n_direct + n_heads;
    if (n_items == 0) return left("no combination kept", 0);  // (nothing to write: the host form says so in its own way)
    uint32_t* d_direct_edge = d.get<uint32_t>(n_direct);
    hip_check(fno_walk_direct(d_off_direct, E, d_direct_edge, st), "copied edges");
    FnoItem* d_items = d.get<FnoItem>(n_items);
    hip_check(fno_walk_items(w, d_off_comb, d_direct_edge, n_direct, d_val_sorted, d_head_pos, n_heads, d_items, d_status, st), "items");
    hip_check(hipMemcpyAsync(h_status, d_status, sizeof h_status, hipMemcpyDeviceToHost, st), "status");
    hip_check(hipStreamSynchronize(st), "synchronize");
    if (h_status[4]) return left("the look-ups", h_status[4]);
    // the walk's inputs are spent
    d.release((void*)w.edges);
    d.release(d_off_comb);
    d.release(d_off_direct);
    d.release(d_val_sorted);
    d.release(d_head_pos);
    d.release(d_direct_edge);
    if (n_items_out) *n_items_out = n_items;
    const auto t1 = now();
    double sec2[2] = {0, 0};
    const bool ok = lines_from_device_items(d, d_items, n_items, h.no_inclusions, text_of, counters, sec2, t1);
    if (!ok) left("computeOverlapData", 0);
    if (seconds) {
        seconds[0] = std::chrono::duration<double>(t1 - t0).count();
        seconds[1] = sec2[0];
        seconds[2] = sec2[1];
    }
    return ok;
}

bool fno3_lines_on_device(const FnoItem* items, uint64_t n, bool no_inclusions, const std::function<char*(uint64_t)>& text_of, uint64_t* n_lines,
                          double* seconds) {
    auto now = [] { return std::chrono::steady_clock::now(); };
    const auto t0 = now();
    hipStream_t st = nullptr;
    DeviceBuffers d;
    FnoItem* d_items = d.get<FnoItem>(n);
    Fno3Rec* d_rec = d.get<Fno3Rec>(n);
    uint64_t *d_len = d.get<uint64_t>(n + 1), *d_off = d.get<uint64_t>(n + 1);
    unsigned long long* d_counters = d.get<unsigned long long>(kFnoCounters);
    size_t scan_bytes = 0;
    hip_check(finder_scan(nullptr, scan_bytes, d_len, d_off, n + 1, st), "scan size");
    void* d_tmp = d.get<char>(scan_bytes);
    hip_check(hipMemcpyAsync(d_items, items, n * sizeof(FnoItem), hipMemcpyHostToDevice, st), "copy of the candidate pairs");
    hip_check(hipMemsetAsync(d_counters, 0, kFnoCounters * sizeof(unsigned long long), st), "memset");
    hip_check(fno3_deduce(d_items, n, no_inclusions ? 1u : 0u, d_rec, d_len, d_counters, st), "deduce");
    hip_check(finder_scan(d_tmp, scan_bytes, d_len, d_off, n + 1, st), "scan");
    unsigned long long h_counters[kFnoCounters];
    uint64_t total_bytes = 0;
    hip_check(hipMemcpyAsync(h_counters, d_counters, sizeof h_counters, hipMemcpyDeviceToHost, st), "counters");
    hip_check(hipMemcpyAsync(&total_bytes, d_off + n, 8, hipMemcpyDeviceToHost, st), "size of the text");
    hip_check(hipStreamSynchronize(st), "synchronize");
    if (h_counters[4]) return false;
    const auto t1 = now();
    char* d_text = d.get<char>(total_bytes);
    hip_check(fno3_format(d_rec, d_len, d_off, n, d_text, st), "format");
    char* h_text = text_of(total_bytes);
    if (total_bytes) hip_check(copy_to_pageable_host(h_text, d_text, total_bytes), "copy of the text");
    if (n_lines) *n_lines = h_counters[5];
    if (seconds) {
        seconds[0] = std::chrono::duration<double>(t1 - t0).count();
        seconds[1] = std::chrono::duration<double>(now() - t1).count();
    }
    return true;
}

}  // namespace hc
