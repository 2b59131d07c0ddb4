// hc_api_text.cpp — hc_text_* / hc_textblock_* (include/hcedge.h): a block of the overlaps file's text goes to the
// device, is split into lines, parsed, prefiltered and scored there (kernels: hc_text_kernels.hip + the scoring kernel),
// and comes back as the few per cent of records the host still has to look at.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/hcedge.h"
#include "hc_ctx.h"
#include "hc_text.h"

static int fail(int status, const std::string& what) { return hc::set_last_error(status, what); }

struct hc_textblock {
    hc_ctx* ctx = nullptr;
    uint64_t max_bytes = 0;
    uint32_t max_lines = 0;
    uint32_t row_cap = 0;  // rows / rejects the mapped host buffers hold; a block that needs more goes to the host
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    hipEvent_t lines_known = nullptr;          // chained submits: this block's entry of the line chain is written
    hipEvent_t copied = nullptr;               // the text has arrived (the copies of all blocks share the context's copy stream)
    char* h_text = nullptr;                    // page-locked, allocated on first use (hc_textblock_buffer): the caller reads the file into it
    char* d_slab = nullptr;                    // ONE device allocation holding d_text ... d_kept_tiles
    char *d_rowslab = nullptr, *h_rowslab = nullptr;  // the row buffers (they may grow): d_rows + d_row_lines; h_rows + h_row_lines + h_rejects, page-locked and mapped
    char* d_text = nullptr;                    // max_bytes + 64
    uint32_t *d_tile_cnt = nullptr, *d_tile_off = nullptr, *d_line_start = nullptr;
    uint32_t* d_tally = nullptr;               // the parse kernel's per-workgroup tallies
    hc_cand_rec* d_cands = nullptr;            // max_lines
    hc_line_rec* d_lines = nullptr;            // max_lines
    hc_result_rec* d_out = nullptr;            // max_lines
    unsigned long long* d_counters = nullptr;  // hc::kTextCounters
    uint32_t* d_kept_tiles = nullptr;          // scratch of launch_kept_rows
    hc_gather_row* d_rows = nullptr;           // row_cap rows: the non-dropped records in file order ...
    hc_line_rec* d_row_lines = nullptr;        // ... each with the parsed line it came from
    hc_gather_row* h_rows = nullptr;           // page-locked, mapped: both streamed out by copy kernels behind the scoring kernel
    hc_line_rec* h_row_lines = nullptr;
    hc_text_reject* h_rejects = nullptr;       // page-locked, mapped: written by the parse kernel
    unsigned long long* h_counters = nullptr;  // page-locked
    hc_text_nonplain* h_nonplain = nullptr;    // page-locked, mapped: the parse kernel lists the lines it does not read (hc_textblock_list_nonplain)
    uint32_t nonplain_cap = 0;
    std::vector<hc_text_row> rows;             // what hc_textblock_wait hands out
    hc_bucket_ws bucket;                       // scratch of a length-bucketed scoring launch (read sets of mixed sequence length)
    // the last submit, kept so that hc_textblock_wait can redo its device half with larger row buffers
    uint64_t sub_bytes = 0, sub_first_line = 0, sub_base_index = 0;
    const unsigned long long* sub_first_line_ptr = nullptr;
    const hc_line_rec* sub_src_lines = nullptr;  // hc_textblock_submit_lines: the block's lines come parsed, from device memory
    uint32_t sub_n_lines = 0;
    uint64_t n_regrown = 0;                    // how often that happened
    std::vector<void*> old_device, old_host;   // row buffers hc_textblock_reserve_rows replaced: freed with the block (a free waits for the device)
    bool in_flight = false;
};

extern "C" {

// FastqStorage::m_ID_to_index (a std::map built by insert(): the first occurrence of an id wins, FastqStorage.h:83-96) as
// a table the parse kernel reads: direct when the ids are dense, else open addressing (the host parser's IdIndex, restated)
int hc_text_set_ids(hc_ctx* c, const uint64_t* read_ids, uint32_t n_reads) {
    if (!c || (n_reads && !read_ids)) return fail(HC_ERR_ARG, "hc_text_set_ids: null argument");
    HC_HIP(hipSetDevice(c->device));
    c->have_ids = false;
    uint64_t max_id = 0;
    for (uint32_t i = 0; i < n_reads; i++) max_id = read_ids[i] > max_id ? read_ids[i] : max_id;
    std::vector<uint32_t> table;
    std::vector<uint64_t> keys;
    if (n_reads == 0 || max_id < 8ull * n_reads + 1024) {
        c->id_direct = 1;
        table.assign(n_reads ? (size_t)max_id + 1 : 1, 0xFFFFFFFFu);
        for (uint32_t i = 0; i < n_reads; i++)
            if (table[read_ids[i]] == 0xFFFFFFFFu) table[read_ids[i]] = i;
        c->id_size = n_reads ? max_id + 1 : 0;
        c->id_shift = 0;
    } else {
        c->id_direct = 0;
        size_t cap = 16;
        int bits = 4;
        while (cap < 2 * (size_t)n_reads) {
            cap <<= 1;
            bits++;
        }
        c->id_shift = 64 - bits;
        c->id_size = cap;
        table.assign(cap, 0xFFFFFFFFu);
        keys.assign(cap, 0);
        for (uint32_t i = 0; i < n_reads; i++) {
            const uint64_t id = read_ids[i];
            uint64_t h = (id * 0x9E3779B97F4A7C15ull) >> c->id_shift;
            while (table[h] != 0xFFFFFFFFu && keys[h] != id) h = (h + 1) & (cap - 1);
            if (table[h] == 0xFFFFFFFFu) {
                table[h] = i;
                keys[h] = id;
            }
        }
    }
    int rc;
    if ((rc = c->id_table.ensure(table.size() * 4)) != HC_OK) return rc;
    HC_HIP(hipMemcpy(c->id_table.p, table.data(), table.size() * 4, hipMemcpyHostToDevice));
    if (!keys.empty()) {
        if ((rc = c->id_keys.ensure(keys.size() * 8)) != HC_OK) return rc;
        HC_HIP(hipMemcpy(c->id_keys.p, keys.data(), keys.size() * 8, hipMemcpyHostToDevice));
    }
    c->have_ids = true;
    return HC_OK;
}

static int textblock_row_buffers(hc_textblock* b, uint64_t cap, bool defer_free);

int hc_textblock_create(hc_ctx* c, uint64_t max_bytes, hc_textblock** out) {
    if (!c || !out || max_bytes < 64 || max_bytes >= (1ull << 31)) return fail(HC_ERR_ARG, "hc_textblock_create: bad argument");
    *out = nullptr;
    HC_HIP(hipSetDevice(c->device));
    hc_textblock* b = new (std::nothrow) hc_textblock();
    if (!b) return fail(HC_ERR_NOMEM, "hc_textblock_create: host allocation failed");
    b->ctx = c;
    b->max_bytes = max_bytes;
    b->max_lines = (uint32_t)(max_bytes / 24 + 16);  // a plain line has at least 26 bytes with its newline; more lines: the host's block
    const uint32_t n_tiles = (uint32_t)((max_bytes + 4095) / 4096);
    const size_t L = b->max_lines;
    b->row_cap = b->max_lines / 8 + 4096;  // a few per cent of the lines survive scoring in real files
    const size_t RC = b->row_cap;
    hipError_t e = hipSuccess;
    auto ok = [&](hipError_t r) {
        if (e == hipSuccess) e = r;
    };
    ok(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
    ok(hipEventCreateWithFlags(&b->done, hipEventDisableTiming));
    ok(hipEventCreateWithFlags(&b->lines_known, hipEventDisableTiming));
    ok(hipEventCreateWithFlags(&b->copied, hipEventDisableTiming));
    // the arrays of fixed size in ONE device allocation (ten of them took 1 ms each: a block cost 17 ms to make, the stage's ten 35 ms
    // beside the read store's upload, the SAVAGE example's process a sixth of its constructor); the row buffers, which may grow, apart
    {
        size_t at = 0;
        auto place = [&](size_t bytes) {
            const size_t here = at;
            at += (bytes + 255) & ~(size_t)255;
            return here;
        };
        const size_t o_text = place(max_bytes + 64), o_tile_cnt = place((size_t)(n_tiles + 1) * 4), o_tile_off = place((size_t)(n_tiles + 1) * 4),
                     o_tally = place(((size_t)b->max_lines / 256 + 2) * 8 * 4), o_line_start = place((L + 2) * 4),
                     o_cands = place((L + 256) * sizeof(hc_cand_rec)), o_lines = place(L * sizeof(hc_line_rec)), o_out = place(L * sizeof(hc_result_rec)),
                     o_counters = place(hc::kTextCounters * sizeof(unsigned long long)), o_kept = place(2 * (L / 1024 + 2) * sizeof(uint32_t));
        ok(hipMalloc((void**)&b->d_slab, at));
        if (b->d_slab) {
            b->d_text = b->d_slab + o_text;
            b->d_tile_cnt = (uint32_t*)(b->d_slab + o_tile_cnt);
            b->d_tile_off = (uint32_t*)(b->d_slab + o_tile_off);
            b->d_tally = (uint32_t*)(b->d_slab + o_tally);
            b->d_line_start = (uint32_t*)(b->d_slab + o_line_start);
            b->d_cands = (hc_cand_rec*)(b->d_slab + o_cands);
            b->d_lines = (hc_line_rec*)(b->d_slab + o_lines);
            b->d_out = (hc_result_rec*)(b->d_slab + o_out);
            b->d_counters = (unsigned long long*)(b->d_slab + o_counters);
            b->d_kept_tiles = (uint32_t*)(b->d_slab + o_kept);
        }
    }
    ok(hipHostMalloc((void**)&b->h_counters, hc::kTextCounters * sizeof(unsigned long long), hipHostMallocMapped));  // the last launch of a block writes them
    if (e != hipSuccess) {
        hc_textblock_destroy(b);
        return fail(HC_ERR_HIP, std::string("hc_textblock_create: ") + hipGetErrorString(e));
    }
    if (const int rc = textblock_row_buffers(b, RC, false)) {  // the row buffers: one device and one page-locked allocation
        const std::string why = hc_last_error();
        hc_textblock_destroy(b);
        return fail(rc, why);
    }
    *out = b;
    return HC_OK;
}

char* hc_textblock_buffer(hc_textblock* b) {
    if (!b) return nullptr;
    if (!b->h_text) {  // only callers that fill the block themselves pay for the page-locked buffer
        (void)hipSetDevice(b->ctx->device);
        // HC_TEXT_BUFFER=wc: write-combined — pread fills it 1.4x as fast, but the CPU must then never read it (the stage
        // with HC_TEXT_SOURCE=pread-chain does not: line numbers come from the device side)
        const char* kind = getenv("HC_TEXT_BUFFER");
        const unsigned flags = (kind && !strcmp(kind, "wc")) ? hipHostMallocWriteCombined : hipHostMallocDefault;
        if (hipHostMalloc((void**)&b->h_text, b->max_bytes + 64, flags) != hipSuccess) b->h_text = nullptr;
    }
    return b->h_text;
}

// A chain of line counts shared by the blocks of one file, in page-locked memory every device and the host can read:
// entry k = lines in front of block k.
struct hc_linechain {
    unsigned long long* h = nullptr;
    uint64_t n = 0;
};

int hc_linechain_create(hc_ctx* c, uint64_t n_blocks, hc_linechain** out) {
    if (!c || !out) return fail(HC_ERR_ARG, "hc_linechain_create: null argument");
    HC_HIP(hipSetDevice(c->device));
    hc_linechain* ch = new (std::nothrow) hc_linechain();
    if (!ch) return fail(HC_ERR_NOMEM, "hc_linechain_create: host allocation failed");
    ch->n = n_blocks + 2;
    hipError_t e = hipHostMalloc((void**)&ch->h, ch->n * sizeof(unsigned long long), hipHostMallocMapped | hipHostMallocPortable);
    if (e != hipSuccess) {
        delete ch;
        return fail(HC_ERR_HIP, std::string("hc_linechain_create: ") + hipGetErrorString(e));
    }
    memset(ch->h, 0, ch->n * sizeof(unsigned long long));
    *out = ch;
    return HC_OK;
}

int hc_linechain_destroy(hc_linechain* ch) {
    if (!ch) return HC_OK;
    if (ch->h) (void)hipHostFree(ch->h);
    delete ch;
    return HC_OK;
}

int hc_textblock_destroy(hc_textblock* b) {
    if (!b) return HC_OK;
    (void)hipSetDevice(b->ctx->device);
    if (b->stream) (void)hipStreamSynchronize(b->stream);
    for (void* p : {(void*)b->d_slab, (void*)b->d_rowslab})  // (d_text ... d_kept_tiles lie in the slab, d_rows / d_row_lines in the row slab)
        if (p) (void)hipFree(p);
    for (void* p : {(void*)b->h_text, (void*)b->h_rowslab, (void*)b->h_counters, (void*)b->h_nonplain})
        if (p) (void)hipHostFree(p);
    for (void* p : b->old_device) (void)hipFree(p);
    for (void* p : b->old_host) (void)hipHostFree(p);
    if (b->done) (void)hipEventDestroy(b->done);
    if (b->lines_known) (void)hipEventDestroy(b->lines_known);
    if (b->copied) (void)hipEventDestroy(b->copied);
    if (b->stream) (void)hipStreamDestroy(b->stream);
    delete b;
    return HC_OK;
}

// Parse -> score -> surviving rows in file order -> the mapped host buffers, on the block's stream: everything of a submit
// behind the line starts.  Reads the block's text, line starts and counters[kTextLines / kTextOverflow] as they are on the device.
static int textblock_device_half(hc_textblock* b) {
    hc_ctx* c = b->ctx;
    hipStream_t s = b->stream;
    hc::TextParams prm;
    prm.n_bytes = b->sub_bytes;
    prm.first_line_no = b->sub_first_line;
    prm.first_line_ptr = b->sub_first_line_ptr;
    prm.max_overlaps = c->settings.max_overlaps;
    prm.max_lines = b->max_lines;
    prm.min_overlap_len = c->settings.min_overlap_len;
    prm.min_overlap_perc = c->settings.min_overlap_perc;
    prm.relax_pe = (c->settings.flags & HC_FLAG_RELAX_PE_EDGES) ? 1u : 0u;
    prm.reject_cap = b->row_cap;
    prm.nonplain_cap = b->h_nonplain ? b->nonplain_cap : 0u;
    hc::IdTable ids;
    ids.table = c->id_table.as<uint32_t>();
    ids.keys = c->id_keys.as<uint64_t>();
    ids.size = c->id_size;
    ids.shift = c->id_shift;
    ids.direct = c->id_direct;
    void *d_rejects = nullptr, *d_rows = nullptr, *d_row_lines = nullptr;
    HC_HIP(hipHostGetDevicePointer(&d_rejects, b->h_rejects, 0));
    HC_HIP(hipHostGetDevicePointer(&d_rows, b->h_rows, 0));
    HC_HIP(hipHostGetDevicePointer(&d_row_lines, b->h_row_lines, 0));
    void* d_nonplain = nullptr;
    if (b->h_nonplain) HC_HIP(hipHostGetDevicePointer(&d_nonplain, b->h_nonplain, 0));
    if (b->sub_src_lines)
        HC_HIP(hc::launch_lines_accept(prm, b->sub_src_lines, b->sub_n_lines, ids, b->d_cands, b->d_lines, (hc_text_reject*)d_rejects, b->d_counters,
                                       b->d_tally, s));
    else
        HC_HIP(hc::launch_text_parse(prm, b->d_text, b->d_line_start, ids, b->d_cands, b->d_lines, (hc_text_reject*)d_rejects, b->d_counters, b->d_tally,
                                     (hc_text_nonplain*)d_nonplain, s));
    // the scoring kernel on the records the parse kernel left behind; how many there are is only known on the device
    int rc = hc_ctx_score(c, HC_REC_COMPACT, b->d_cands, b->max_lines, b->d_out, s, false, nullptr, nullptr, 0, 0,
                          b->d_counters + hc::kTextLines, nullptr, nullptr, &b->bucket);
    if (rc) return rc;
    HC_HIP(hc::launch_kept_rows(b->d_out, b->max_lines, b->d_counters + hc::kTextLines, b->sub_base_index, b->d_kept_tiles,
                                b->d_kept_tiles + (b->max_lines / 1024 + 2), b->d_rows, b->row_cap, b->d_counters + hc::kTextRows, b->d_lines,
                                b->d_row_lines, s, b->d_tally, b->d_counters));
    // both row arrays and the counters -> the block's page-locked words, one launch
    void* d_counters_host = nullptr;
    HC_HIP(hipHostGetDevicePointer(&d_counters_host, b->h_counters, 0));
    HC_HIP(hc::launch_flush_text_rows(b->d_rows, d_rows, b->d_row_lines, d_row_lines, b->d_counters + hc::kTextRows, b->row_cap, b->d_counters,
                                      (unsigned long long*)d_counters_host, c->n_cu, s));
    return HC_OK;
}

// More surviving rows (or prefilter rejects) than the block's row buffers hold: real files can keep more than the eighth
// of the lines the buffers start with (overlaps found at a low error rate; later iterations).  The text, its line starts and
// the line count are still on the device and the device's parse is good: the buffers grow to what this block needs (and stay
// that size) and the device half runs again — the block does not fall back to the host's tokeniser.
// the row buffers (device rows, their mapped host twins, the rejects) for `cap` rows; defer_free: the old ones are kept for
// hc_textblock_destroy instead of being freed now (hipFree / hipHostFree wait for the device: not next to another launch sequence)
static int textblock_row_buffers(hc_textblock* b, uint64_t cap, bool defer_free) {
    if (b->d_rowslab) {
        if (defer_free) b->old_device.push_back(b->d_rowslab);
        else (void)hipFree(b->d_rowslab);
    }
    if (b->h_rowslab) {
        if (defer_free) b->old_host.push_back(b->h_rowslab);
        else (void)hipHostFree(b->h_rowslab);
    }
    b->d_rowslab = b->h_rowslab = nullptr;
    b->d_rows = nullptr;
    b->d_row_lines = nullptr;
    b->h_rows = nullptr;
    b->h_row_lines = nullptr;
    b->h_rejects = nullptr;
    b->row_cap = (uint32_t)cap;
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t rows_b = up(cap * sizeof(hc_gather_row)), lines_b = up(cap * sizeof(hc_line_rec)), rej_b = up(cap * sizeof(hc_text_reject));
    HC_HIP(hipMalloc((void**)&b->d_rowslab, rows_b + lines_b));
    HC_HIP(hipHostMalloc((void**)&b->h_rowslab, rows_b + lines_b + rej_b, hipHostMallocMapped));
    b->d_rows = (hc_gather_row*)b->d_rowslab;
    b->d_row_lines = (hc_line_rec*)(b->d_rowslab + rows_b);
    b->h_rows = (hc_gather_row*)b->h_rowslab;
    b->h_row_lines = (hc_line_rec*)(b->h_rowslab + rows_b);
    b->h_rejects = (hc_text_reject*)(b->h_rowslab + rows_b + lines_b);
    return HC_OK;
}

static int textblock_regrow(hc_textblock* b, uint64_t need) {
    hc_ctx* c = b->ctx;
    uint64_t cap = need + need / 8 + 1024;
    if (cap > b->max_lines) cap = b->max_lines;
    if (cap < need) return fail(HC_ERR_STATE, "hc_textblock_wait: more rows than lines");
    HC_HIP(hipStreamSynchronize(b->stream));
    {
        const int rc = textblock_row_buffers(b, cap, false);
        if (rc) return rc;
    }
    // what the device half adds to: the parse kernel's tallies and slots, the row count (lines / overflow stay)
    HC_HIP(hipMemsetAsync(b->d_counters, 0, hc::kTextLines * sizeof(unsigned long long), b->stream));
    HC_HIP(hipMemsetAsync(b->d_counters + hc::kTextRows, 0, (hc::kTextCounters - hc::kTextRows) * sizeof(unsigned long long), b->stream));
    int rc = textblock_device_half(b);  // (its last launch leaves the counters in h_counters)
    if (rc) return rc;
    HC_HIP(hipStreamSynchronize(b->stream));
    b->n_regrown++;
    (void)c;
    return HC_OK;
}

static int textblock_submit(hc_textblock* b, const void* src, uint64_t n_bytes, uint64_t first_line_no, hc_linechain* chain, uint64_t k,
                            hc_textblock* prev, uint64_t base_index);

int hc_textblock_submit(hc_textblock* b, uint64_t n_bytes, uint64_t first_line_no, uint64_t base_index) {
    if (!b) return fail(HC_ERR_ARG, "hc_textblock_submit: null block");
    if (!b->h_text) return fail(HC_ERR_STATE, "hc_textblock_submit: hc_textblock_buffer was never filled");
    return textblock_submit(b, b->h_text, n_bytes, first_line_no, nullptr, 0, nullptr, base_index);
}

int hc_textblock_submit_from(hc_textblock* b, const void* text, uint64_t n_bytes, hc_linechain* chain, uint64_t k, hc_textblock* prev,
                             uint64_t base_index) {
    if (!b || (n_bytes && !text) || !chain) return fail(HC_ERR_ARG, "hc_textblock_submit_from: null argument");
    if (k + 1 >= chain->n) return fail(HC_ERR_ARG, "hc_textblock_submit_from: the line chain is shorter than that");
    return textblock_submit(b, text, n_bytes, 0, chain, k, prev, base_index);
}

static int textblock_submit(hc_textblock* b, const void* src, uint64_t n_bytes, uint64_t first_line_no, hc_linechain* chain, uint64_t k,
                            hc_textblock* prev, uint64_t base_index) {
    hc_ctx* c = b->ctx;
    if (!c->have_reads || !c->have_ids) return fail(HC_ERR_STATE, "hc_textblock_submit: hc_set_reads and hc_text_set_ids come first");
    if (b->in_flight) return fail(HC_ERR_STATE, "hc_textblock_submit: the block is still in flight (hc_textblock_wait first)");
    if (n_bytes > b->max_bytes) return fail(HC_ERR_ARG, "hc_textblock_submit: more text than the block was created for");
    HC_HIP(hipSetDevice(c->device));
    hipStream_t s = b->stream;
    unsigned long long* d_chain = nullptr;
    if (chain) HC_HIP(hipHostGetDevicePointer((void**)&d_chain, chain->h, 0));
    if (n_bytes) {
        // the text as it is (page-locked or pageable: a file mapping goes to the device without a copy by the caller).  The 16-byte
        // pieces of the last tile reach past the text: the kernels that look for newlines ignore what lies there.
        // The blocks' copies take turns on TWO streams of the context and a block's own stream waits for its copy.  On the ten
        // blocks' own streams the runtime opened a further copy queue whenever a copy met others in flight: 7 - 8 ms inside
        // hipMemcpyAsync, five or six times during the first file of a process (a per-block trace of the submits, round 3) — C3's first construct_edges of a
        // process 0.18 - 0.20 s against 0.13 - 0.14 s with two; one stream alone does not keep the link busy (later files 0.13 - 0.17 s
        // against 0.12; a round-3 knob, gone: the measurement stands).
        {
            hipStream_t& cs = c->text_copy_stream[c->text_copy_next++ % 2u];
            if (!cs) HC_HIP(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
            HC_HIP(hipMemcpyAsync(b->d_text, src, n_bytes, hipMemcpyHostToDevice, cs));
            HC_HIP(hipEventRecord(b->copied, cs));
            HC_HIP(hipStreamWaitEvent(s, b->copied, 0));
        }
        HC_HIP(hc::launch_text_count(b->d_text, n_bytes, b->d_tile_cnt, s));
    }
    // The one-workgroup scan zeroes the block's counters, sets the line count and hands it down the chain: entry k is written by
    // block k - 1's scan (another stream, maybe another device), entry k + 1 by this one.  An empty block runs it too (over no tiles).
    if (chain && prev) HC_HIP(hipStreamWaitEvent(s, prev->lines_known, 0));
    HC_HIP(hc::launch_text_scan(b->d_text, n_bytes, b->d_tile_cnt, b->d_tile_off, b->max_lines, b->d_line_start, b->d_counters,
                                chain ? d_chain + k : nullptr, chain ? d_chain + k + 1 : nullptr, s));
    if (chain) HC_HIP(hipEventRecord(b->lines_known, s));
    if (n_bytes) HC_HIP(hc::launch_text_line_starts(b->d_text, n_bytes, b->d_tile_off, b->max_lines, b->d_line_start, s));
    b->sub_bytes = n_bytes;
    b->sub_first_line = first_line_no;
    b->sub_first_line_ptr = chain ? d_chain + k : nullptr;
    b->sub_base_index = base_index;
    b->sub_src_lines = nullptr;
    b->sub_n_lines = 0;
    if (n_bytes) {
        int rc = textblock_device_half(b);  // (its last launch leaves the counters in h_counters)
        if (rc) return rc;
    } else {
        HC_HIP(hipMemcpyAsync(b->h_counters, b->d_counters, hc::kTextCounters * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    }
    HC_HIP(hipEventRecord(b->done, s));
    b->in_flight = true;
    return HC_OK;
}

int hc_textblock_submit_lines(hc_textblock* b, const hc_line_rec* d_lines, uint64_t n_lines, uint64_t first_line_no, uint64_t base_index) {
    if (!b || (n_lines && !d_lines)) return fail(HC_ERR_ARG, "hc_textblock_submit_lines: null argument");
    hc_ctx* c = b->ctx;
    if (!c->have_reads || !c->have_ids) return fail(HC_ERR_STATE, "hc_textblock_submit_lines: hc_set_reads and hc_text_set_ids come first");
    if (b->in_flight) return fail(HC_ERR_STATE, "hc_textblock_submit_lines: the block is still in flight (hc_textblock_wait first)");
    if (n_lines > b->max_lines) return fail(HC_ERR_ARG, "hc_textblock_submit_lines: more lines than the block has room for (hc_textblock_max_lines)");
    HC_HIP(hipSetDevice(c->device));
    b->sub_bytes = 0;
    b->sub_first_line = first_line_no;
    b->sub_first_line_ptr = nullptr;
    b->sub_base_index = base_index;
    b->sub_src_lines = d_lines;
    b->sub_n_lines = (uint32_t)n_lines;
    int rc = textblock_device_half(b);  // (its last launch leaves the counters in h_counters)
    if (rc) return rc;
    HC_HIP(hipEventRecord(b->done, b->stream));
    b->in_flight = true;
    return HC_OK;
}

uint64_t hc_textblock_max_lines(hc_textblock* b) { return b ? b->max_lines : 0; }

uint64_t hc_textblock_regrown(hc_textblock* b) { return b ? b->n_regrown : 0; }

int hc_textblock_reserve_rows(hc_textblock* b, uint64_t rows) {
    if (!b) return fail(HC_ERR_ARG, "hc_textblock_reserve_rows: null block");
    if (b->in_flight) return fail(HC_ERR_STATE, "hc_textblock_reserve_rows: the block is still in flight (hc_textblock_wait first)");
    if (rows > b->max_lines) rows = b->max_lines;
    if (rows <= b->row_cap) return HC_OK;
    HC_HIP(hipSetDevice(b->ctx->device));
    return textblock_row_buffers(b, rows, true);
}

int hc_textblock_wait(hc_textblock* b, hc_text_result* out) {
    if (!b || !out) return fail(HC_ERR_ARG, "hc_textblock_wait: null argument");
    memset(out, 0, sizeof *out);
    if (!b->in_flight) return fail(HC_ERR_STATE, "hc_textblock_wait: nothing was submitted");
    HC_HIP(hipSetDevice(b->ctx->device));
    HC_HIP(hipEventSynchronize(b->done));
    b->in_flight = false;
    const unsigned long long* k = b->h_counters;
    // lines the device did not read send the block to the host unless they are all in the list (hc_textblock_list_nonplain)
    auto nonplain_blocks = [&](const unsigned long long* kk) {
        return kk[hc::kTextNonPlain] != 0 && !(b->h_nonplain && kk[hc::kTextNonPlain] <= b->nonplain_cap && kk[hc::kTextNonPlainSlots] == kk[hc::kTextNonPlain]);
    };
    if (!(k[hc::kTextOverflow] || nonplain_blocks(k) || k[hc::kTextUnknownId]) &&
        (k[hc::kTextRows] > b->row_cap || k[hc::kTextRejectSlots] > b->row_cap)) {
        const uint64_t need = k[hc::kTextRows] > k[hc::kTextRejectSlots] ? k[hc::kTextRows] : k[hc::kTextRejectSlots];
        int rc = textblock_regrow(b, need);
        if (rc) return rc;
        k = b->h_counters;
    }
    out->n_lines = k[hc::kTextLines];
    out->lines_read = k[hc::kTextRead];
    out->n_nonplain = k[hc::kTextNonPlain];
    out->n_unknown_id = k[hc::kTextUnknownId];
    out->needs_host = (k[hc::kTextOverflow] || nonplain_blocks(k) || k[hc::kTextUnknownId] || k[hc::kTextRows] > b->row_cap ||
                       k[hc::kTextRejectSlots] > b->row_cap)
                          ? 1
                          : 0;
    if (out->needs_host) return HC_OK;
    out->self_overlaps = k[hc::kTextSelf];
    out->silently_dropped = k[hc::kTextSilent];
    out->prefilter_rejected = k[hc::kTextReject];
    out->scored = k[hc::kTextPass];
    const uint64_t n_rows = k[hc::kTextRows], n_rej = k[hc::kTextRejectSlots];
    // rows: in file order as they are (launch_kept_rows); the parse kernel's rejects are appended in no particular order
    b->rows.resize(n_rows);
    for (uint64_t i = 0; i < n_rows; i++) {
        b->rows[i].row = b->h_rows[i];
        b->rows[i].line = b->h_row_lines[i];
    }
    std::sort(b->h_rejects, b->h_rejects + n_rej, [](const hc_text_reject& x, const hc_text_reject& y) { return x.line_index < y.line_index; });
    out->rows = b->rows.data();
    out->n_rows = n_rows;
    out->rejected = b->h_rejects;
    out->n_rejected = n_rej;
    if (k[hc::kTextNonPlain]) {  // (all of them are listed, else needs_host)
        std::sort(b->h_nonplain, b->h_nonplain + k[hc::kTextNonPlain], [](const hc_text_nonplain& x, const hc_text_nonplain& y) { return x.line_index < y.line_index; });
        out->nonplain = b->h_nonplain;
        out->n_nonplain_listed = k[hc::kTextNonPlain];
    }
    return HC_OK;
}

int hc_textblock_list_nonplain(hc_textblock* b, uint32_t max_lines) {
    if (!b) return fail(HC_ERR_ARG, "hc_textblock_list_nonplain: null block");
    if (b->in_flight) return fail(HC_ERR_STATE, "hc_textblock_list_nonplain: the block is still in flight (hc_textblock_wait first)");
    HC_HIP(hipSetDevice(b->ctx->device));
    if (max_lines > b->nonplain_cap || (max_lines == 0 && b->h_nonplain)) {
        if (b->h_nonplain) HC_HIP(hipHostFree(b->h_nonplain));
        b->h_nonplain = nullptr;
        if (max_lines) HC_HIP(hipHostMalloc((void**)&b->h_nonplain, (size_t)max_lines * sizeof(hc_text_nonplain), hipHostMallocMapped));
    }
    b->nonplain_cap = max_lines;
    return HC_OK;
}

}  // extern "C"
