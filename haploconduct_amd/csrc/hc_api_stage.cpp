// hc_api_stage.cpp — what the stage (construct_edges) needs of the device beyond plain scoring (include/hcedge.h):
//   hc_block_*  : one block of candidates in flight — H2D of the compact records, the scoring kernel appending the
//                 non-dropped records straight into page-locked host memory, one event to wait on;
//   hc_graph_*  : duplicate resolution + adjacency lists on the device (kernels: hc_graph_kernels.hip).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/hcedge.h"
#include "hc_ctx.h"
#include "hc_graph.h"

static int fail(int status, const std::string& what) { return hc::set_last_error(status, what); }

struct hc_block {
    hc_ctx* ctx = nullptr;
    uint64_t cap = 0;  // candidates
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    void* d_in = nullptr;                  // cap hc_cand_rec
    void* d_out = nullptr;                 // cap hc_result_rec
    unsigned long long* d_count = nullptr; // rows appended by the kernel
    hc_gather_row* d_rows = nullptr;       // cap rows: the non-dropped records in sequence order (launch_kept_rows)
    uint32_t* d_tiles = nullptr;           // its scratch: two arrays of cap / 1024 + 2 counters
    hc_gather_row* h_rows = nullptr;       // page-locked, mapped: the rows, streamed out by a copy kernel behind it
    unsigned long long* h_count = nullptr; // page-locked
    uint64_t n = 0, base_index = 0;
    hc_bucket_ws bucket;                   // scratch of a length-bucketed scoring launch (read sets of mixed sequence length)
    bool in_flight = false;
};

extern "C" {

int hc_block_create(hc_ctx* c, uint64_t max_candidates, hc_block** out) {
    if (!c || !out || max_candidates == 0 || max_candidates >= (1ull << 31)) return fail(HC_ERR_ARG, "hc_block_create: bad argument");
    *out = nullptr;
    HC_HIP(hipSetDevice(c->device));
    hc_block* b = new (std::nothrow) hc_block();
    if (!b) return fail(HC_ERR_NOMEM, "hc_block_create: host allocation failed");
    b->ctx = c;
    b->cap = max_candidates;
    auto cleanup = [&](hipError_t e, const char* what) {
        hc_block_destroy(b);
        return fail(HC_ERR_HIP, std::string("hc_block_create: ") + what + ": " + hipGetErrorString(e));
    };
    hipError_t e;
    if ((e = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking)) != hipSuccess) return cleanup(e, "stream");
    if ((e = hipEventCreateWithFlags(&b->done, hipEventDisableTiming)) != hipSuccess) return cleanup(e, "event");
    if ((e = hipMalloc(&b->d_in, max_candidates * sizeof(hc_cand_rec))) != hipSuccess) return cleanup(e, "candidates");
    if ((e = hipMalloc(&b->d_out, max_candidates * sizeof(hc_result_rec))) != hipSuccess) return cleanup(e, "results");
    if ((e = hipMalloc((void**)&b->d_count, sizeof(unsigned long long))) != hipSuccess) return cleanup(e, "count");
    if ((e = hipMalloc((void**)&b->d_rows, max_candidates * sizeof(hc_gather_row))) != hipSuccess) return cleanup(e, "rows");
    if ((e = hipMalloc((void**)&b->d_tiles, 2 * (max_candidates / 1024 + 2) * sizeof(uint32_t))) != hipSuccess) return cleanup(e, "tiles");
    if ((e = hipHostMalloc((void**)&b->h_rows, max_candidates * sizeof(hc_gather_row), hipHostMallocMapped)) != hipSuccess)
        return cleanup(e, "row buffer");
    if ((e = hipHostMalloc((void**)&b->h_count, sizeof(unsigned long long), hipHostMallocDefault)) != hipSuccess) return cleanup(e, "count buffer");
    *out = b;
    return HC_OK;
}

int hc_block_destroy(hc_block* b) {
    if (!b) return HC_OK;
    (void)hipSetDevice(b->ctx->device);
    if (b->stream) (void)hipStreamSynchronize(b->stream);
    if (b->d_in) (void)hipFree(b->d_in);
    if (b->d_out) (void)hipFree(b->d_out);
    if (b->d_count) (void)hipFree(b->d_count);
    if (b->d_rows) (void)hipFree(b->d_rows);
    if (b->d_tiles) (void)hipFree(b->d_tiles);
    if (b->h_rows) (void)hipHostFree(b->h_rows);
    if (b->h_count) (void)hipHostFree(b->h_count);
    if (b->done) (void)hipEventDestroy(b->done);
    if (b->stream) (void)hipStreamDestroy(b->stream);
    delete b;
    return HC_OK;
}

int hc_block_submit(hc_block* b, const hc_cand_rec* cands, uint64_t n, uint64_t base_index) {
    if (!b) return fail(HC_ERR_ARG, "hc_block_submit: null block");
    hc_ctx* c = b->ctx;
    if (!c->have_reads) return fail(HC_ERR_STATE, "hc_block_submit: hc_set_reads has not been called");
    if (b->in_flight) return fail(HC_ERR_STATE, "hc_block_submit: the block is still in flight (hc_block_wait first)");
    if (n > b->cap) return fail(HC_ERR_ARG, "hc_block_submit: more candidates than the block was created for");
    if (n && !cands) return fail(HC_ERR_ARG, "hc_block_submit: null records");
    HC_HIP(hipSetDevice(c->device));
    b->n = n;
    b->base_index = base_index;
    *b->h_count = 0;
    if (n) {
        void* d_rows = nullptr;
        HC_HIP(hipHostGetDevicePointer(&d_rows, b->h_rows, 0));
        HC_HIP(hipMemcpyAsync(b->d_in, cands, n * sizeof(hc_cand_rec), hipMemcpyHostToDevice, b->stream));
        // as given: the stage's blocks come from files in sfo2overlaps / FNO order; an unordered file still scores
        // correctly, only slower (hc_set_reorder(HC_REORDER_ALWAYS) sorts every block first)
        int rc = hc_ctx_score(c, HC_REC_COMPACT, b->d_in, n, b->d_out, b->stream, false, nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, &b->bucket);
        if (rc) return rc;
        HC_HIP(hc::launch_kept_rows((const hc_result_rec*)b->d_out, n, nullptr, base_index, b->d_tiles, b->d_tiles + (b->cap / 1024 + 2), b->d_rows,
                                    b->cap, b->d_count, nullptr, nullptr, b->stream));
        HC_HIP(hc::launch_flush_rows(b->d_rows, d_rows, b->d_count, b->cap, sizeof(hc_gather_row), c->n_cu, b->stream));
        HC_HIP(hipMemcpyAsync(b->h_count, b->d_count, sizeof(unsigned long long), hipMemcpyDeviceToHost, b->stream));
    }
    HC_HIP(hipEventRecord(b->done, b->stream));
    b->in_flight = true;
    return HC_OK;
}

int hc_block_wait(hc_block* b, const hc_gather_row** rows, uint64_t* n_rows) {
    if (!b || !rows || !n_rows) return fail(HC_ERR_ARG, "hc_block_wait: null argument");
    *rows = nullptr;
    *n_rows = 0;
    if (!b->in_flight) return fail(HC_ERR_STATE, "hc_block_wait: nothing was submitted");
    HC_HIP(hipSetDevice(b->ctx->device));
    HC_HIP(hipEventSynchronize(b->done));
    b->in_flight = false;
    const uint64_t k = *b->h_count;
    if (k > b->cap) return fail(HC_ERR_STATE, "hc_block_wait: row count beyond the block's capacity");
    *rows = b->h_rows;  // in sequence order as they are (launch_kept_rows)
    *n_rows = k;
    return HC_OK;
}

// ---------------------------------------------------------------------------------------------------------------
int hc_graph_begin(hc_ctx* c) {
    if (!c) return fail(HC_ERR_ARG, "hc_graph_begin: null context");
    c->graph.n_appended = 0;
    c->graph.valid = false;
    return HC_OK;
}

int hc_graph_append(hc_ctx* c, const hc_admit_rec* admitted, uint64_t n) {
    if (!c || (n && !admitted)) return fail(HC_ERR_ARG, "hc_graph_append: null argument");
    if (n == 0) return HC_OK;
    HC_HIP(hipSetDevice(c->device));
    hc_ctx::Graph& g = c->graph;
    g.valid = false;
    const uint64_t have = g.n_appended, want = have + n;
    if (want >= (1ull << 31)) return fail(HC_ERR_ARG, "hc_graph_append: more than 2^31-1 records");
    if (want * sizeof(hc_admit_rec) > g.adm.cap) {  // grow, keeping what is there
        size_t cap = g.adm.cap ? g.adm.cap : ((size_t)1 << 24);
        while (cap < want * sizeof(hc_admit_rec)) cap *= 2;
        void* bigger = nullptr;
        HC_HIP(hipStreamSynchronize(c->stream));  // appends in flight land in the old buffer first
        HC_HIP(hipMalloc(&bigger, cap));
        if (have) {
            const hipError_t e = hipMemcpy(bigger, g.adm.p, have * sizeof(hc_admit_rec), hipMemcpyDeviceToDevice);
            if (e != hipSuccess) {
                (void)hipFree(bigger);
                return fail(HC_ERR_HIP, std::string("hc_graph_append: ") + hipGetErrorString(e));
            }
        }
        g.adm.release();
        g.adm.p = bigger;
        g.adm.cap = cap;
    }
    // through a page-locked buffer, asynchronously on the context's stream (the blocks score on their own streams;
    // hc_graph_resolve runs on this one, behind the copies): the caller — the stage's in-order half — does not wait
    const int t = g.stage_turn;
    g.stage_turn ^= 1;
    const size_t bytes = n * sizeof(hc_admit_rec);
    if (!g.stage_free[t]) HC_HIP(hipEventCreateWithFlags(&g.stage_free[t], hipEventDisableTiming));
    else HC_HIP(hipEventSynchronize(g.stage_free[t]));  // its previous copy (two appends ago) has left the buffer
    if (g.stage_cap[t] < bytes) {
        if (g.h_stage[t]) (void)hipHostFree(g.h_stage[t]);
        g.h_stage[t] = nullptr;
        g.stage_cap[t] = 0;
        size_t cap = (size_t)1 << 20;
        while (cap < bytes) cap *= 2;
        HC_HIP(hipHostMalloc(&g.h_stage[t], cap, hipHostMallocDefault));
        g.stage_cap[t] = cap;
    }
    memcpy(g.h_stage[t], admitted, bytes);
    HC_HIP(hipMemcpyAsync((char*)g.adm.p + have * sizeof(hc_admit_rec), g.h_stage[t], bytes, hipMemcpyHostToDevice, c->stream));
    HC_HIP(hipEventRecord(g.stage_free[t], c->stream));
    g.n_appended = want;
    return HC_OK;
}

int hc_graph_resolve(hc_ctx* c, const hc_admit_rec* admitted, uint64_t n, uint64_t n_vertices, const uint32_t* vertex_of_read,
                     uint32_t order, hc_graph_counts* counts) {
    if (!c || !counts) return fail(HC_ERR_ARG, "hc_graph_resolve: null argument");
    memset(counts, 0, sizeof *counts);
    counts->first_bad = -1;
    if (!c->have_reads) return fail(HC_ERR_STATE, "hc_graph_resolve: hc_set_reads has not been called");
    const bool appended = admitted == nullptr && n != 0;
    if (appended && n != c->graph.n_appended) return fail(HC_ERR_ARG, "hc_graph_resolve: n differs from the number of appended records");
    if (order != HC_GRAPH_INSERTION_ORDER && order != HC_GRAPH_SORTED) return fail(HC_ERR_ARG, "hc_graph_resolve: unknown order");
    if (n >= (1ull << 31) || n_vertices >= (1ull << 31)) return fail(HC_ERR_ARG, "hc_graph_resolve: more than 2^31-1 records or vertices");
    HC_HIP(hipSetDevice(c->device));
    hc_ctx::Graph& g = c->graph;
    g.valid = false;
    const uint32_t m = (uint32_t)n, V = (uint32_t)n_vertices;
    hipStream_t s = c->stream;
    const size_t m1 = m ? m : 1;
    int rc;
#define ENS(buf, bytes)                                 \
    if ((rc = g.buf.ensure(bytes)) != HC_OK) return rc
    if (!appended) ENS(adm, m1 * sizeof(hc_admit_rec));
    ENS(E, m1 * sizeof(hc_edge_rec));
    ENS(key0, m1 * 8);
    ENS(key1, m1 * 8);
    ENS(idx0, m1 * 4);
    ENS(idx1, m1 * 4);
    ENS(keep, m1);
    ENS(incl, (size_t)V + 1);
    ENS(tied, (size_t)V + 1);
    ENS(counters, 8 * sizeof(unsigned long long));
    ENS(surv, m1 * 4);
    ENS(k32a, m1 * 4);
    ENS(k32b, m1 * 4);
    ENS(k64a, m1 * 8);
    ENS(k64b, m1 * 8);
    ENS(tmp_idx, m1 * 4);
    ENS(o_out, m1 * 4);
    ENS(o_in, m1 * 4);
    ENS(out_off, ((size_t)V + 1) * 8);
    ENS(in_off, ((size_t)V + 1) * 8);
    ENS(in_nodes, m1 * 4);
    ENS(tied_list, ((size_t)V + 1) * 4);
    ENS(temp, hc::graph_temp_bytes(m ? m : 1, V ? V : 1));
    hc::GraphParams gp;
    gp.reads = c->d_reads;
    gp.n_reads = c->view.n_reads;
    gp.vtx = nullptr;
    gp.n_vertices = V;
    gp.ignore_inclusions = (c->settings.flags & HC_FLAG_IGNORE_INCLUSIONS) ? 1u : 0u;
    if (vertex_of_read) {
        ENS(vtx, (size_t)(c->view.n_reads ? c->view.n_reads : 1) * 4);
        HC_HIP(hipMemcpyAsync(g.vtx.p, vertex_of_read, (size_t)c->view.n_reads * 4, hipMemcpyHostToDevice, s));
        gp.vtx = g.vtx.as<uint32_t>();
    }
    unsigned long long init[8] = {0, 0, 0, 0, ~0ull, 0, 0, 0};
    HC_HIP(hipMemcpyAsync(g.counters.p, init, sizeof init, hipMemcpyHostToDevice, s));
    HC_HIP(hipMemsetAsync(g.keep.p, 0, m1, s));
    HC_HIP(hipMemsetAsync(g.incl.p, 0, (size_t)V + 1, s));
    HC_HIP(hipMemsetAsync(g.tied.p, 0, (size_t)V + 1, s));
    unsigned long long* d_count = g.counters.as<unsigned long long>() + 5;
    if (m) {
        if (!appended) {
            HC_HIP(hipMemcpyAsync(g.adm.p, admitted, (size_t)m * sizeof(hc_admit_rec), hipMemcpyHostToDevice, s));
            g.n_appended = 0;
        }
        HC_HIP(hc::graph_build_and_replay(gp, g.adm.as<hc_admit_rec>(), m, g.E.as<hc_edge_rec>(), g.key0.as<uint64_t>(), g.key1.as<uint64_t>(),
                                          g.idx0.as<uint32_t>(), g.idx1.as<uint32_t>(), g.keep.as<uint8_t>(), g.incl.as<uint8_t>(),
                                          g.counters.as<unsigned long long>(), g.surv.as<uint32_t>(), d_count, g.temp.p, g.temp.cap, s));
    }
    unsigned long long h[8];
    HC_HIP(hipMemcpyAsync(h, g.counters.p, sizeof h, hipMemcpyDeviceToHost, s));
    HC_HIP(hipStreamSynchronize(s));
    const uint32_t n_edges = m ? (uint32_t)h[5] : 0u;
    counts->n_admitted = m;
    counts->n_edges = n_edges;
    counts->inclusion_count = h[0];
    counts->dup_count = h[1];
    counts->first_bad = h[4] == ~0ull ? -1 : (int64_t)h[4];
    if (counts->first_bad >= 0) return HC_OK;  // the caller reports it; nothing to fetch
    if (m && h[2] != n_edges) return fail(HC_ERR_STATE, "hc_graph_resolve: slots and survivors disagree");
    ENS(edges_out, (size_t)(n_edges ? n_edges : 1) * sizeof(hc_edge_rec));
    HC_HIP(hc::graph_orders(gp, g.E.as<hc_edge_rec>(), g.surv.as<uint32_t>(), n_edges, order, g.k32a.as<uint32_t>(), g.k32b.as<uint32_t>(),
                            g.k64a.as<uint64_t>(), g.k64b.as<uint64_t>(), g.tmp_idx.as<uint32_t>(), g.o_out.as<uint32_t>(), g.o_in.as<uint32_t>(),
                            g.out_off.as<unsigned long long>(), g.in_off.as<unsigned long long>(), g.tied.as<uint8_t>(), g.temp.p, g.temp.cap, s));
    HC_HIP(hc::graph_gather(g.E.as<hc_edge_rec>(), g.o_out.as<uint32_t>(), g.o_in.as<uint32_t>(), n_edges, g.edges_out.as<hc_edge_rec>(),
                            g.in_nodes.as<uint32_t>(), s));
    uint64_t n_tied = 0;
    if (order == HC_GRAPH_SORTED && n_edges && V) {
        unsigned long long* d_tied_count = g.counters.as<unsigned long long>() + 6;
        HC_HIP(hc::graph_select_tied(g.tied.as<uint8_t>(), V, g.tied_list.as<uint32_t>(), d_tied_count, g.temp.p, g.temp.cap, s));
        unsigned long long t = 0;
        HC_HIP(hipMemcpyAsync(&t, d_tied_count, sizeof t, hipMemcpyDeviceToHost, s));
        HC_HIP(hipStreamSynchronize(s));
        n_tied = t;
    }
#undef ENS
    counts->n_tied_lists = n_tied;
    g.n_vertices = V;
    g.n_edges = n_edges;
    g.n_tied = n_tied;
    g.valid = true;
    return HC_OK;
}

int hc_graph_fetch(hc_ctx* c, hc_edge_rec* edges, uint64_t* out_off, uint32_t* in_nodes, uint64_t* in_off, uint32_t* seq, uint8_t* inclusions,
                   uint32_t* tied_vertices) {
    if (!c) return fail(HC_ERR_ARG, "hc_graph_fetch: null context");
    hc_ctx::Graph& g = c->graph;
    if (!g.valid) return fail(HC_ERR_STATE, "hc_graph_fetch: no resolved graph on the device");
    HC_HIP(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const size_t E = g.n_edges, V = g.n_vertices;
    static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "offsets are copied as they are");
    if (edges && E) HC_HIP(hipMemcpyAsync(edges, g.edges_out.p, E * sizeof(hc_edge_rec), hipMemcpyDeviceToHost, s));
    if (out_off) HC_HIP(hipMemcpyAsync(out_off, g.out_off.p, (V + 1) * 8, hipMemcpyDeviceToHost, s));
    if (in_nodes && E) HC_HIP(hipMemcpyAsync(in_nodes, g.in_nodes.p, E * 4, hipMemcpyDeviceToHost, s));
    if (in_off) HC_HIP(hipMemcpyAsync(in_off, g.in_off.p, (V + 1) * 8, hipMemcpyDeviceToHost, s));
    if (seq && E) HC_HIP(hipMemcpyAsync(seq, g.o_out.p, E * 4, hipMemcpyDeviceToHost, s));
    if (inclusions && V) HC_HIP(hipMemcpyAsync(inclusions, g.incl.p, V, hipMemcpyDeviceToHost, s));
    if (tied_vertices && g.n_tied) HC_HIP(hipMemcpyAsync(tied_vertices, g.tied_list.p, g.n_tied * 4, hipMemcpyDeviceToHost, s));
    HC_HIP(hipStreamSynchronize(s));
    return HC_OK;
}

int hc_graph_fetch_edges(hc_ctx* c, uint64_t first, uint64_t count, hc_edge_rec* dst) {
    if (!c) return fail(HC_ERR_ARG, "hc_graph_fetch_edges: null context");
    hc_ctx::Graph& g = c->graph;
    if (!g.valid) return fail(HC_ERR_STATE, "hc_graph_fetch_edges: no resolved graph on the device");
    if (first > g.n_edges || count > g.n_edges - first) return fail(HC_ERR_ARG, "hc_graph_fetch_edges: range beyond the edges");
    if (count == 0) return HC_OK;
    if (!dst) return fail(HC_ERR_ARG, "hc_graph_fetch_edges: null destination");
    HC_HIP(hipSetDevice(c->device));
    HC_HIP(hipMemcpyAsync(dst, g.edges_out.as<hc_edge_rec>() + first, count * sizeof(hc_edge_rec), hipMemcpyDeviceToHost, c->stream));
    HC_HIP(hipStreamSynchronize(c->stream));
    return HC_OK;
}

}  // extern "C"
