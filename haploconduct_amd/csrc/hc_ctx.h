// hc_ctx.h — the context behind the C ABI (include/hcedge.h), shared by the hc_api*.cpp translation units, and the
// launch interface of the kernel translation units.  Internal: nothing here is part of the ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <string>
#include <vector>

#include "../../include/hcedge.h"
#include "hc_device.h"
#include "hc_overlap_finder.h"

namespace hc {
// hc_kernels.hip
hipError_t launch_encode(uint32_t symbytes, const uint8_t* bases, const uint8_t* quals, const uint64_t* raw_off,
                         const uint64_t* seq_off, const uint32_t* rc_delta, const uint8_t* qmap, uint32_t n_seq, uint32_t K, void* sym,
                         uint8_t* seq_bad, const uint32_t* read_first_seq, uint32_t n_reads, ReadDesc* descs, uint32_t slot_align,
                         hipStream_t stream);
hipError_t launch_score(const StoreView& st, const ScoreParams& prm, const double* lut_g, const void* in, uint64_t n,
                        hc_result_rec* out, const uint32_t* perm, uint32_t n_cu, int fetch_group, int lane_fetch_group, hc_gather_row* rows,
                        unsigned long long* row_count, uint64_t cap, uint64_t base_index, hipStream_t stream,
                        const hc_line_rec* lines_in = nullptr, hc_line_rec* lines_out = nullptr, uint32_t* bucket_perm = nullptr,
                        uint32_t* bucket_queue = nullptr, hc_gather_row* seg_buf = nullptr, uint32_t* seg_count = nullptr, uint64_t seg_total_rows = 0,
                        uint32_t* spill_turn = nullptr, unsigned long long* started = nullptr, uint32_t* started_groups = nullptr);
// started / started_groups: hc_comm_gate_device — the cooperative launch's workgroups add one each to *started as they start;
// *started_groups = how many will (0: the launch took a kernel that does not count)
// seg_buf (seg_total_rows rows: segments, then `cap` rows of spill area) / seg_count (kSinkMaxGroups counters + 2 spill counters, all
// zero before the first launch) / spill_turn (host: which spill counter the next launch uses; advanced by a launch that used segments):
// scratch of a launch that collects its rows in per-workgroup segments
// bucket_perm (n uint32) / bucket_queue (one uint32): scratch of the length-bucketed launch (read sets of mixed sequence
// length, StoreView::balance): without them such a set is scored in the order given
hipError_t set_score_kernel_lds_limit();
std::string describe_score_kernel(const StoreView& st, int fetch_group, int lane_fetch_group, uint32_t n_cu = 256, uint64_t n = 0);  // n: the form a launch of n candidates takes (0: in general)
// hc_util_kernels.hip
size_t compact_temp_bytes(uint32_t n);
hipError_t launch_compact(const hc_result_rec* res, uint32_t n, uint32_t* idx_out, unsigned long long* count_out, void* temp,
                          size_t temp_bytes, hipStream_t stream);
hipError_t launch_narrow_payload(const void* in32, uint64_t cap, void* out24, uint32_t n_cu, hipStream_t stream);
hipError_t launch_pack_rows(const hc_result_rec* res, const uint32_t* idx, const unsigned long long* count, uint64_t cap, uint64_t base,
                            hc_gather_row* rows, uint32_t n_cu, hipStream_t stream);
hipError_t launch_pack_header(const unsigned long long* count, hc_gather_row* header, hipStream_t stream);
hipError_t launch_gather_results(const hc_result_rec* res, const uint32_t* idx, const unsigned long long* count,
                                 hc_result_rec* out, uint32_t n_cu, hipStream_t stream);
size_t reorder_temp_bytes(uint32_t n);
hipError_t launch_reorder(uint32_t n_reads, uint32_t fmt, const void* in, uint32_t n, uint32_t* keys_in, uint32_t* keys_out,
                          uint32_t* idx_in, uint32_t* perm_out, void* temp, size_t temp_bytes, hipStream_t stream);
hipError_t launch_count_positions(const StoreView& st, uint32_t min_read_len, uint32_t fmt, const void* in, uint64_t n,
                                  unsigned long long* totals, hipStream_t stream);
hipError_t launch_kept_rows(const hc_result_rec* res, uint64_t n, const unsigned long long* n_dev, uint64_t base_index, uint32_t* tile_cnt,
                            uint32_t* tile_off, hc_gather_row* rows, uint64_t cap, unsigned long long* count, const hc_line_rec* lines_in,
                            hc_line_rec* lines_out, hipStream_t stream, const uint32_t* text_tally = nullptr,
                            unsigned long long* text_counters = nullptr);  // (a text block: the parse kernel's tallies are summed on the way)
hipError_t launch_flush_text_rows(const void* rows, void* rows_mapped, const void* lines, void* lines_mapped, const unsigned long long* count,
                                  uint64_t cap, const unsigned long long* counters, unsigned long long* counters_mapped, uint32_t n_cu,
                                  hipStream_t stream);
hipError_t launch_comm_gate(const unsigned long long* started, unsigned long long target, uint32_t timeout_us, hipStream_t stream);
hipError_t launch_flush_rows(const void* src, void* dst_mapped, const unsigned long long* count, uint64_t cap, uint32_t row_bytes, uint32_t n_cu,
                             hipStream_t stream);

int set_last_error(int status, const std::string& what);  // thread-local text behind hc_last_error()
}  // namespace hc

#define HC_HIP(call)                                                                                   \
    do {                                                                                               \
        hipError_t e__ = (call);                                                                       \
        if (e__ != hipSuccess)                                                                         \
            return hc::set_last_error(HC_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); \
    } while (0)

// A grow-only device (or page-locked host) block owned by the context: allocating and freeing per call costs more
// than most of the kernels here.
struct hc_scratch {
    void* p = nullptr;
    size_t cap = 0;
    bool host = false;  // hipHostMalloc (mapped) instead of hipMalloc
    int ensure(size_t bytes) {  // contents are not kept
        if (bytes <= cap) return HC_OK;
        release();
        const size_t want = bytes + bytes / 8;
        if (host) HC_HIP(hipHostMalloc(&p, want, hipHostMallocMapped));
        else HC_HIP(hipMalloc(&p, want));
        cap = want;
        return HC_OK;
    }
    void release() {
        if (p) (void)(host ? hipHostFree(p) : hipFree(p));
        p = nullptr;
        cap = 0;
    }
    template <typename T>
    T* as() const { return (T*)p; }
    hc_scratch() = default;
    hc_scratch(const hc_scratch&) = delete;
    hc_scratch& operator=(const hc_scratch&) = delete;
    ~hc_scratch() { release(); }
};

// Scratch of one length-bucketed scoring launch (hc::bucket_perm_kernel): the queue counter, then the permutation.  One per
// thing that launches on its own stream (the context, every hc_block / hc_textblock); grow-only.
struct hc_bucket_ws {
    hc_scratch mem;
    int ensure(uint64_t n) { return mem.ensure((n + 16) * sizeof(uint32_t)); }
    uint32_t* queue() const { return mem.as<uint32_t>(); }
    uint32_t* perm() const { return mem.as<uint32_t>() + 16; }
};

struct hc_ctx {
    hc_settings settings;
    int device = 0;
    uint32_t n_cu = 256;
    bool coop_fetch = true;  // the scoring kernel fetches symbols cooperatively (quads read 64-byte rows; stores below 4 GiB)
    int fetch_group = 4;     // otherwise per lane, in groups of 4 (short reads) or 2 (contigs, 16-bit symbols) 16-symbol chunks;
                             // chosen per read set in hc_set_reads (HC_FETCH_GROUP=coop|4|2 overrides: a tuning knob only)
    hipStream_t stream = nullptr;
    hipStream_t text_copy_stream[4] = {nullptr, nullptr, nullptr, nullptr};  // the text blocks' host-to-device copies, in turn (hc_api_text.cpp)
    std::atomic<uint32_t> text_copy_next{0};  // (the streams themselves are created by the one thread that submits: hcedge.h)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // read store
    bool have_reads = false;
    void* d_sym = nullptr;
    hc::ReadDesc* d_reads = nullptr;
    double* d_lut = nullptr;
    double* d_inv_n = nullptr;  // StoreView::inv_n
    uint32_t len_p5 = 0, len_p95 = 0;  // 5th / 95th percentile of the sequence lengths: what the kernel dispatch calls "mixed"
    uint64_t store_bytes = 0;
    hc::StoreView view{};
    hc::ScoreParams params{};
    // what the overlap finder needs of the sequences (host copies, filled by hc_set_reads)
    std::vector<hc::SeqRef> seq_refs;  // by store sequence index
    bool singles_first = true;
    // result of the last hc_find_overlaps, kept on the device so that the usual "ask for the count, then fetch"
    // pair of calls computes once
    struct Scratch {  // grow-only device scratch of the finder, one slot per buffer, freed with the store
        void* p = nullptr;
        size_t cap = 0;
    } finder_scratch[24], ingest_scratch[12];  // the second set: hc_found_to_overlaps
    void* h_ingest[2] = {nullptr, nullptr};  // page-locked ring hc_found_to_overlaps copies the sorted records through
    size_t h_ingest_cap = 0;                  // bytes of each
    void* h_sfo_text[3] = {nullptr, nullptr, nullptr};  // page-locked stations of hc_set_found_from_sfo_text (32 MiB + 64 each), kept
    hc_sfo_rec* d_found = nullptr;  // grow-only (round 6): room for found_cap records, n_found of them valid
    uint64_t found_cap = 0;
    uint64_t n_found = 0;
    hc_line_rec* d_found_lines = nullptr;  // hc_found_to_lines_device: the overlap lines of the found records, kept until the store is replaced
    uint64_t found_lines_cap = 0;
    double found_err = -1;
    uint32_t found_min = 0, found_flags = 0;
    bool found_valid = false;
    // grow-only workspace for the host-buffer entry points
    void* d_in = nullptr;
    void* d_out = nullptr;
    uint64_t ws_cap = 0;
    unsigned long long* d_totals = nullptr;
    // candidate reorder (HC_REORDER_*): scratch for the (key, index) radix sort, grow-only
    int reorder_mode = HC_REORDER_AUTO;
    uint32_t* d_sort = nullptr;  // 4 arrays of sort_cap uint32: keys_in, keys_out, idx_in, perm
    void* d_sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    uint64_t sort_cap = 0;
    hc_bucket_ws bucket;  // length-bucketed launches on the context's own entry points
    hc_scratch sink_rows, sink_counts;  // hc_score_pack_device: the row sink's per-workgroup segments (hc_kernels.hip: RowSink)
    uint32_t sink_turn = 0;             // which of the two spill counters the next segmented launch uses
    bool sink_dirty = false;            // a segmented launch failed at enqueue: the spill counters are re-zeroed before the next one
    // the multi-GPU step (hc_set_comm_reserve / hc_comm_gate_device): CUs left free for the collective library's kernels, and the
    // count of workgroups that have started, which the gate kernel on the exchange's stream waits for
    uint32_t comm_reserve = 0;
    unsigned long long* d_started = nullptr;
    unsigned long long started_target = 0;  // workgroups launched so far with the counter (host side)

    // The context's own scratch (reorder workspace, `bucket`, the sink's segments) serves one launch at a time: a launch that uses
    // any of it on another stream than the last such launch waits for that one (hc_ctx_score)
    hipEvent_t scratch_done = nullptr;
    hipStream_t scratch_stream = nullptr;
    bool scratch_used = false;
    // compaction scratch, grow-only
    void* d_compact_tmp = nullptr;
    size_t compact_tmp_bytes = 0;
    uint32_t* d_compact_idx = nullptr;
    hc_result_rec* d_compact_res = nullptr;
    uint64_t compact_cap = 0;
    // hc_text_set_ids (hc_api_text.cpp): FastqStorage::m_ID_to_index on the device
    hc_scratch id_table, id_keys;
    uint64_t id_size = 0;
    int id_shift = 0, id_direct = 1;
    bool have_ids = false;
    // hc_graph_resolve / hc_graph_fetch (hc_api_stage.cpp): device-resident result of the last resolve
    struct Graph {
        hc_scratch adm, E, key0, key1, idx0, idx1, keep, incl, tied, counters, surv, k32a, k32b, k64a, k64b, tmp_idx, o_out, o_in,
            out_off, in_off, edges_out, in_nodes, vtx, temp, tied_list;
        uint64_t n_vertices = 0, n_edges = 0, n_tied = 0;
        uint64_t n_appended = 0;  // records hc_graph_append has put into adm
        bool valid = false;
        // hc_graph_append copies through two page-locked buffers on the context's stream and does not wait for the copy
        void* h_stage[2] = {nullptr, nullptr};
        size_t stage_cap[2] = {0, 0};
        hipEvent_t stage_free[2] = {nullptr, nullptr};
        int stage_turn = 0;
    } graph;
};

int hc_ctx_score(hc_ctx* c, uint32_t fmt, const void* d_in, uint64_t n, void* d_out, hipStream_t s, bool reorder,
                 hc_gather_row* rows, unsigned long long* row_count, uint64_t cap, uint64_t base_index,
                 const unsigned long long* n_dev = nullptr, const hc_line_rec* lines_in = nullptr, hc_line_rec* lines_out = nullptr,
                 hc_bucket_ws* bucket = nullptr);  // bucket: the caller's own scratch (launches beside the context's stream), else the context's

// hc_found_to_overlaps with the overlaps file's text left in memory (hc_api_finder.cpp)
int hc_found_to_overlaps_text(hc_ctx* c, uint64_t num_singles, uint64_t num_pairs, std::string& text, uint64_t* n_lines);
