// hc_api_finder.cpp — hc_find_overlaps (include/hcedge.h): candidate generation on the device against the read store
// (SURVEY.md §8(f4)); kernels in hc_overlap_finder.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/hcedge.h"
#include "hc_ctx.h"
#include "hc_fno_device.h"
#include "hc_hostcopy.h"
#include "hc_prims.h"
#include "hc_sfo_device.h"
#include "hc_text.h"
#include "host/NumaBind.h"
#include "host/Types.h"

namespace hc {
std::string sfo_records_to_overlaps(const hc_sfo_rec* recs, uint64_t n, long ns, long np, uint64_t& n_lines);
std::string sfo_sorted_to_overlaps(const SfoFlipped* recs, uint64_t n, long ns, long np, uint64_t& n_lines);
}  // namespace hc

static int fail(int status, const std::string& what) { return hc::set_last_error(status, what); }

namespace {
// A view of one of the context's grow-only scratch slots (hc_ctx::finder_scratch).  The overlap finder needs
// gigabytes of scratch per call; allocating and freeing them every call (hipMalloc/hipFree or the stream-ordered
// pool alike) costs several times its kernels, so the blocks stay with the context until the store is replaced.
struct DevBuf {
    void* p = nullptr;
    hc_ctx::Scratch* slot = nullptr;
    void* own = nullptr;  // a block that is not a slot (the result, which outlives the call)
    ~DevBuf() {
        if (own) (void)hipFree(own);
    }
    template <typename T>
    T* as() const { return (T*)p; }
};
// Device blocks for the SFO ingest: the finder's grow-only scratch is idle then and large enough for most of what is needed; what it cannot
// serve comes from a second grow-only set (hc_ctx::ingest_scratch).  (Allocating and freeing 8 GB per call costs ten times the sorts.)  Every
// block handed out is the caller's until the pool goes; a context that ingests inputs of growing size regrows its smallest idle block when
// the second set is full.
struct IngestScratch {
    hc_ctx* c;
    std::vector<hc_ctx::Scratch*> idle;
    std::vector<void*> mine;  // freed with the pool
    unsigned n_own = 0;
    IngestScratch(const IngestScratch&) = delete;
    IngestScratch& operator=(const IngestScratch&) = delete;
    ~IngestScratch() {
        for (void* q : mine) (void)hipFree(q);
    }
    explicit IngestScratch(hc_ctx* ctx) : c(ctx) {
        for (auto& sl : c->finder_scratch)
            if (sl.p) idle.push_back(&sl);
        for (auto& sl : c->ingest_scratch)
            if (sl.p) idle.push_back(&sl);
    }
    hipError_t operator()(size_t bytes, void** p) {
        const hipError_t e = take(bytes, p);
        if (getenv("HC_SCRATCH_TRACE")) fprintf(stderr, "[hc scratch] %zu bytes: %s\n", bytes, how);
        return e;
    }
    const char* how = "";
    hipError_t take(size_t bytes, void** p) {
        const size_t need = bytes ? bytes : 16;
        int best = -1;
        for (size_t i = 0; i < idle.size(); i++)
            if (idle[i] && idle[i]->cap >= need && (best < 0 || idle[i]->cap < idle[(size_t)best]->cap)) best = (int)i;
        if (best >= 0) {
            *p = idle[(size_t)best]->p;
            idle[(size_t)best] = nullptr;
            how = "an idle block";
            return hipSuccess;
        }
        while (n_own < 12 && c->ingest_scratch[n_own].p) n_own++;  // an empty slot of the second set
        hc_ctx::Scratch* sl = nullptr;
        how = "a new block of the second set";
        if (n_own < 12) {
            sl = &c->ingest_scratch[n_own];
        } else {  // every slot holds a block that is too small: the smallest idle one OF THE SECOND SET grows (a finder slot given another
                  // size here would be freed and allocated again by the finder's next call, and again here: 0.6 s per call at config 3's size)
            int small = -1;
            for (size_t i = 0; i < idle.size(); i++)
                if (idle[i] && idle[i] >= &c->ingest_scratch[0] && idle[i] <= &c->ingest_scratch[11] &&
                    (small < 0 || idle[i]->cap < idle[(size_t)small]->cap))
                    small = (int)i;
            if (small < 0) {  // (every block of the second set is handed out: a block of this call's own)
                void* q = nullptr;
                const hipError_t e = hipMalloc(&q, need);
                if (e != hipSuccess) return e;
                mine.push_back(q);
                *p = q;
                how = "a block of this call's own (hipMalloc + hipFree)";
                return hipSuccess;
            }
            sl = idle[(size_t)small];
            idle[(size_t)small] = nullptr;
            how = "the smallest idle block of the second set regrown (hipFree + hipMalloc)";
            (void)hipFree(sl->p);
            sl->p = nullptr;
            sl->cap = 0;
        }
        const hipError_t e = hipMalloc(&sl->p, need);
        if (e != hipSuccess) {
            sl->p = nullptr;
            return e;
        }
        sl->cap = need;
        *p = sl->p;
        return hipSuccess;
    }
};
}  // namespace

#define HC_ALLOC(buf, bytes)                                                                  \
    do {                                                                                      \
        hc_ctx::Scratch& sl__ = c->finder_scratch[n_slots++];                                 \
        const size_t need__ = (bytes) ? (size_t)(bytes) : 16;                                 \
        if (sl__.cap < need__) {                                                              \
            if (sl__.p) (void)hipFree(sl__.p);                                                \
            sl__.p = nullptr;                                                                 \
            sl__.cap = 0;                                                                     \
            HC_HIP(hipMalloc(&sl__.p, need__ + need__ / 8));                                  \
            sl__.cap = need__ + need__ / 8;                                                   \
        }                                                                                     \
        (buf).slot = &sl__;                                                                   \
        (buf).p = sl__.p;                                                                     \
    } while (0)

// the temporary storage of the library calls: the slot keeps what earlier calls grew it to
#define HC_GROW_TMP(bytes)                                        \
    do {                                                          \
        if ((bytes) > d_tmp.slot->cap) {                          \
            HC_HIP(hipStreamSynchronize(st));                     \
            (void)hipFree(d_tmp.slot->p);                         \
            d_tmp.slot->p = nullptr;                              \
            d_tmp.slot->cap = 0;                                  \
            HC_HIP(hipMalloc(&d_tmp.slot->p, (bytes)));           \
            d_tmp.slot->cap = (bytes);                            \
            d_tmp.p = d_tmp.slot->p;                              \
        }                                                         \
        tmp_bytes = d_tmp.slot->cap;                              \
    } while (0)

extern "C" {

// ---- candidate generation ------------------------------------------------------------------------------------
int hc_find_overlaps(hc_ctx* c, double err_rate, uint32_t min_overlap, uint32_t flags, hc_sfo_rec* out, uint64_t cap, uint64_t* n_out) {
    if (!c || !n_out) return fail(HC_ERR_ARG, "hc_find_overlaps: null argument");
    *n_out = 0;
    if (!c->have_reads) return fail(HC_ERR_STATE, "hc_find_overlaps: hc_set_reads has not been called");
    if (cap && !out) return fail(HC_ERR_ARG, "hc_find_overlaps: null output buffer");
    if (!(err_rate >= 0.0) || err_rate >= 1.0 || min_overlap == 0) return fail(HC_ERR_ARG, "hc_find_overlaps: need 0 <= err_rate < 1, min_overlap > 0");
    if (!c->singles_first) return fail(HC_ERR_ARG, "hc_find_overlaps: the read set must list single-end reads before pairs (SFO ids)");
    const uint32_t n_seq = (uint32_t)c->seq_refs.size();
    if (n_seq >= (1u << 24) - 1) return fail(HC_ERR_ARG, "hc_find_overlaps: more than 2^24-2 sequences");
    uint32_t max_len = 0;
    for (const hc::SeqRef& r : c->seq_refs) max_len = r.len > max_len ? r.len : max_len;
    if (max_len >= (1u << 14)) return fail(HC_ERR_ARG, "hc_find_overlaps: sequences of 16384 symbols or more are not supported");
    if (n_seq < 2 || max_len < min_overlap) {  // nothing can overlap: an empty result, remembered like any other
        c->n_found = 0;
        c->found_err = err_rate;
        c->found_min = min_overlap;
        c->found_flags = flags & ~HC_FIND_RECOMPUTE;
        c->found_valid = true;
        return HC_OK;
    }
    HC_HIP(hipSetDevice(c->device));
    const bool timing = getenv("HC_FIND_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tmark = now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        (void)hipStreamSynchronize(c->stream);
        const double t = now();
        fprintf(stderr, "hc_find_overlaps: %-28s %.4f s\n", what, t - tmark);
        tmark = t;
    };
    const bool recompute = flags & HC_FIND_RECOMPUTE;
    flags &= ~HC_FIND_RECOMPUTE;
    if (!recompute && c->found_valid && c->found_err == err_rate && c->found_min == min_overlap && c->found_flags == flags) {
        *n_out = c->n_found;
        const uint64_t take = c->n_found < cap ? c->n_found : cap;
        if (take) HC_HIP(hipMemcpy(out, c->d_found, take * sizeof(hc_sfo_rec), hipMemcpyDeviceToHost));
        return HC_OK;
    }
    c->n_found = 0;  // (the previous result's buffer stays: grow-only — asking the driver for 2 GB per call cost 0.3 - 0.4 s from the second call on)
    c->found_valid = false;
    lap("free previous result");
    auto remember = [&](hc_sfo_rec* d, uint64_t n) {
        if (d) c->d_found = d;
        c->n_found = n;
        c->found_err = err_rate;
        c->found_min = min_overlap;
        c->found_flags = flags;
        c->found_valid = true;
    };
    // the longest stretch without a mismatch that every reportable overlap is guaranteed to contain
    uint32_t w = 0xFFFFFFFFu;
    for (uint32_t L = min_overlap; L <= max_len; L++) {
        const uint32_t K = (uint32_t)(err_rate * (double)L);
        const uint32_t wl = (L - K) / (K + 1);
        w = wl < w ? wl : w;
    }
    if (w < 12)
        return fail(HC_ERR_ARG, min_overlap < 12 ? "hc_find_overlaps: min_overlap below 12 is not supported by the seed filter"
                                                 : "hc_find_overlaps: err_rate too high for this min_overlap: an overlap need not contain 12 error-free positions in a row");
    const uint32_t k = w < 31 ? w : 31, s = w - k + 1;
    const uint32_t n_ori = (flags & HC_FIND_REVERSALS) ? 2u : 1u;
    const bool wide = c->view.symbytes == 1 && hc::lut_lg(c->view.K) >= 6;
    hipStream_t st = c->stream;
    unsigned n_slots = 0;  // HC_ALLOC takes the context's scratch slots in order

    // host-side layout of the index and of the seeds
    std::vector<uint64_t> pos_start(n_seq + 1, 0), seed_start(n_seq + 1, 0);
    std::vector<hc::SeqRef> by_sfo(n_seq);
    for (uint32_t q = 0; q < n_seq; q++) {
        const hc::SeqRef& r = c->seq_refs[q];
        pos_start[q + 1] = pos_start[q] + r.len;
        seed_start[q + 1] = seed_start[q] + (r.len >= k ? (uint64_t)((r.len - k) / s + 1) * n_ori : 0);
        by_sfo[r.sfo_id] = r;
    }
    const uint64_t P = pos_start[n_seq], S = seed_start[n_seq];
    if (P >= (1ull << 31) || S >= (1ull << 31)) return fail(HC_ERR_ARG, "hc_find_overlaps: read set too large for one call (2^31 positions)");

    DevBuf d_idlen, d_seqs, d_by_sfo, d_pos_start, d_seed_start, d_k0, d_k1, d_v0, d_v1, d_tmp, d_lo, d_cnt, d_off, d_count;
    HC_ALLOC(d_seqs, n_seq * sizeof(hc::SeqRef));
    HC_ALLOC(d_by_sfo, n_seq * sizeof(hc::SeqRef));
    HC_ALLOC(d_idlen, n_seq * sizeof(uint2));
    HC_ALLOC(d_pos_start, (n_seq + 1) * sizeof(uint64_t));
    HC_ALLOC(d_seed_start, (n_seq + 1) * sizeof(uint64_t));
    HC_ALLOC(d_count, sizeof(unsigned long long));
    HC_HIP(hipMemcpyAsync(d_seqs.p, c->seq_refs.data(), n_seq * sizeof(hc::SeqRef), hipMemcpyHostToDevice, st));
    HC_HIP(hipMemcpyAsync(d_by_sfo.p, by_sfo.data(), n_seq * sizeof(hc::SeqRef), hipMemcpyHostToDevice, st));
    std::vector<uint2> idlen(n_seq);  // what a seed hit asks of the indexed sequence: 8 bytes instead of a SeqRef
    for (uint32_t q = 0; q < n_seq; q++) idlen[q] = make_uint2(c->seq_refs[q].sfo_id, c->seq_refs[q].len);
    HC_HIP(hipMemcpyAsync(d_idlen.p, idlen.data(), n_seq * sizeof(uint2), hipMemcpyHostToDevice, st));
    HC_HIP(hipMemcpyAsync(d_pos_start.p, pos_start.data(), (n_seq + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    HC_HIP(hipMemcpyAsync(d_seed_start.p, seed_start.data(), (n_seq + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st));

    // 1. index: (k-mer, sequence|position) of every forward position, sorted by k-mer
    HC_ALLOC(d_k0, P * 8);
    HC_ALLOC(d_k1, P * 8);
    HC_ALLOC(d_v0, P * 8);
    HC_ALLOC(d_v1, P * 8);
    HC_HIP(hc::finder_index(c->d_sym, c->view.symbytes, wide, d_seqs.as<hc::SeqRef>(), d_pos_start.as<uint64_t>(), n_seq, k, d_k0.as<uint64_t>(),
                            d_v0.as<uint64_t>(), st));
    size_t tmp_bytes = 0;
    HC_HIP(hc::finder_sort_pairs(nullptr, tmp_bytes, d_k0.as<uint64_t>(), d_k1.as<uint64_t>(), d_v0.as<uint64_t>(), d_v1.as<uint64_t>(), P, 64, st));
    HC_ALLOC(d_tmp, tmp_bytes);
    tmp_bytes = d_tmp.slot->cap;
    HC_HIP(hc::finder_sort_pairs(d_tmp.p, tmp_bytes, d_k0.as<uint64_t>(), d_k1.as<uint64_t>(), d_v0.as<uint64_t>(), d_v1.as<uint64_t>(), P, 64, st));
    lap("index + sort");
    // 2. seeds: range of every seed k-mer in the index
    HC_ALLOC(d_lo, S * 8);
    HC_ALLOC(d_cnt, (S + 1) * 8);
    HC_ALLOC(d_off, (S + 1) * 8);
    HC_HIP(hipMemsetAsync(d_cnt.p, 0, (S + 1) * 8, st));
    HC_HIP(hc::finder_seeds(c->d_sym, c->view.symbytes, wide, d_seqs.as<hc::SeqRef>(), d_seed_start.as<uint64_t>(), n_seq, k, s, n_ori,
                            d_k1.as<uint64_t>(), P, d_lo.as<uint64_t>(), d_cnt.as<uint64_t>(), st));
    // only hits whose indexed sequence has the lower id become candidates: count those, and lay the keys out by them
    DevBuf d_val;
    HC_ALLOC(d_val, (S + 1) * 8);
    HC_HIP(hipMemsetAsync(d_val.p, 0, (S + 1) * 8, st));
    HC_HIP(hc::finder_count_valid(d_seqs.as<hc::SeqRef>(), d_idlen.as<uint2>(), d_seed_start.as<uint64_t>(), n_seq, k, s, n_ori, d_v1.as<uint64_t>(),
                                  d_lo.as<uint64_t>(), d_cnt.as<uint64_t>(), min_overlap, flags, d_val.as<uint64_t>(), st));
    {
        size_t b = 0;
        HC_HIP(hc::finder_scan(nullptr, b, d_val.as<uint64_t>(), d_off.as<uint64_t>(), S + 1, st));
        HC_GROW_TMP(b);
        HC_HIP(hc::finder_scan(d_tmp.p, b, d_val.as<uint64_t>(), d_off.as<uint64_t>(), S + 1, st));
    }
    uint64_t H = 0;  // number of candidate hits = last element of the exclusive scan over S + 1 counts (the extra one is 0)
    HC_HIP(hipMemcpyAsync(&H, d_off.as<uint64_t>() + S, 8, hipMemcpyDeviceToHost, st));
    HC_HIP(hipStreamSynchronize(st));
    lap("seeds + scan");
    if (H == 0) {
        remember(nullptr, 0);
        return HC_OK;
    }
    // 3./4. in batches of seed sequences, so that the hits in flight stay bounded whatever the coverage of the data:
    //   one key per hit -> sort -> unique (the candidate diagonals) -> verify (8 bytes out per candidate) -> scan of the
    //   flags -> emit the records of the verified ones behind those of the batches before.
    // Keys start with the ids, batches are id ranges of the seed side: the concatenation is still sorted and unique.
    std::vector<uint64_t> h_bound(n_seq + 1);  // candidate hits before sequence q = off[seed_start[q]]
    {
        DevBuf d_bound;
        HC_ALLOC(d_bound, (n_seq + 1) * 8);
        HC_HIP(hc::finder_boundaries(d_off.as<uint64_t>(), d_seed_start.as<uint64_t>(), n_seq + 1, d_bound.as<uint64_t>(), st));
        HC_HIP(hipMemcpyAsync(h_bound.data(), d_bound.p, (n_seq + 1) * 8, hipMemcpyDeviceToHost, st));
        HC_HIP(hipStreamSynchronize(st));
    }
    // 2^27 seed hits in flight (28 bytes of scratch each): with 2^29 the first call of a context allocated 15 GB for the batches alone,
    // which costs 0.2 - 0.9 s on some hosts (the call itself takes 0.12 s); four batches instead of one cost 6 % once the scratch exists
    uint64_t batch_hits = 1ull << 27;
    if (const char* e = getenv("HC_FIND_BATCH_HITS")) batch_hits = strtoull(e, nullptr, 10);
    if (batch_hits < 1024) batch_hits = 1024;
    if (batch_hits > (1ull << 31)) batch_hits = 1ull << 31;  // the sorts (hc_prims.hip) index with 32 bits
    struct Batch {
        uint32_t q0, q1;
        uint64_t base, hits;
    };
    std::vector<Batch> batches;
    uint64_t Hmax = 0;
    for (uint32_t q0 = 0; q0 < n_seq;) {
        const uint64_t base = h_bound[q0];
        uint32_t q1 = q0 + 1;
        while (q1 < n_seq && h_bound[q1 + 1] - base <= batch_hits) q1++;
        const uint64_t hits = h_bound[q1] - base;
        if (hits >= (1ull << 31)) return fail(HC_ERR_ARG, "hc_find_overlaps: one sequence alone has more than 2^31 seed hits");
        if (hits) {
            batches.push_back(Batch{q0, q1, base, hits});
            Hmax = hits > Hmax ? hits : Hmax;
        }
        q0 = q1;
    }
    DevBuf d_h0, d_h1, d_kout, d_flag, d_pos, d_r1;
    HC_ALLOC(d_h0, Hmax * 8);
    HC_ALLOC(d_h1, Hmax * 8);
    HC_ALLOC(d_kout, Hmax * 4);
    HC_ALLOC(d_flag, (Hmax + 1) * 4);
    HC_ALLOC(d_pos, (Hmax + 1) * 4);
    {
        size_t b = 0, b2 = 0, b3 = 0;
        HC_HIP(hc::finder_sort_keys(nullptr, b, d_h0.as<uint64_t>(), d_h1.as<uint64_t>(), Hmax, st));
        HC_HIP(hc::finder_unique(nullptr, b2, d_h1.as<uint64_t>(), d_h0.as<uint64_t>(), d_count.as<unsigned long long>(), Hmax, st));
        HC_HIP(hc::finder_scan32(nullptr, b3, d_flag.as<uint32_t>(), d_pos.as<uint32_t>(), Hmax + 1, st));
        b = b2 > b ? b2 : b;
        b = b3 > b ? b3 : b;
        HC_GROW_TMP(b);
    }
    HC_HIP(hipStreamSynchronize(st));
    lap("scratch for the batches");
    unsigned long long R = 0;
    // the records of the batches collect in one more grow-only slot (growing keeps what is in it); the result that
    // outlives the call is allocated once, at the end, at its size: one hipMalloc and (for the previous result) one hipFree
    // per call — allocating and freeing gigabytes per batch and for the final sort cost more than the kernels on some hosts
    hc_ctx::Scratch& acc = c->finder_scratch[n_slots++];
    size_t res_cap = acc.cap / sizeof(hc_sfo_rec);  // records
    for (const Batch& bt : batches) {
        const uint64_t Hb = bt.hits;
        HC_HIP(hc::finder_expand(d_seqs.as<hc::SeqRef>(), d_idlen.as<uint2>(), d_seed_start.as<uint64_t>(), bt.q0, bt.q1, bt.base, k, s, n_ori, d_v1.as<uint64_t>(),
                                 d_lo.as<uint64_t>(), d_cnt.as<uint64_t>(), d_off.as<uint64_t>(), min_overlap, flags, d_h0.as<uint64_t>(), st));
        size_t bs = tmp_bytes;
        HC_HIP(hc::finder_sort_keys(d_tmp.p, bs, d_h0.as<uint64_t>(), d_h1.as<uint64_t>(), Hb, st));
        bs = tmp_bytes;
        HC_HIP(hc::finder_unique(d_tmp.p, bs, d_h1.as<uint64_t>(), d_h0.as<uint64_t>(), d_count.as<unsigned long long>(), Hb, st));
        unsigned long long M = 0;
        HC_HIP(hipMemcpyAsync(&M, d_count.p, sizeof M, hipMemcpyDeviceToHost, st));
        HC_HIP(hipStreamSynchronize(st));
        if (M == 0) continue;
        HC_HIP(hipMemsetAsync(d_flag.as<uint32_t>() + M, 0, 4, st));
        HC_HIP(hc::finder_verify(c->d_sym, c->view.symbytes, wide, d_by_sfo.as<hc::SeqRef>(), d_h0.as<uint64_t>(), M, err_rate, min_overlap, flags,
                                 d_kout.as<uint32_t>(), d_flag.as<uint32_t>(), st));
        bs = tmp_bytes;
        HC_HIP(hc::finder_scan32(d_tmp.p, bs, d_flag.as<uint32_t>(), d_pos.as<uint32_t>(), M + 1, st));
        uint32_t Rb = 0;
        HC_HIP(hipMemcpyAsync(&Rb, d_pos.as<uint32_t>() + M, 4, hipMemcpyDeviceToHost, st));
        HC_HIP(hipStreamSynchronize(st));
        if (Rb == 0) continue;
        if (R + Rb > res_cap) {
            size_t want = res_cap ? res_cap * 2 : (size_t)Rb;
            if (want < R + Rb) want = R + Rb;
            if (batches.size() == 1) want = Rb;
            void* bigger = nullptr;
            HC_HIP(hipMalloc(&bigger, want * sizeof(hc_sfo_rec)));
            if (R) HC_HIP(hipMemcpyAsync(bigger, acc.p, R * sizeof(hc_sfo_rec), hipMemcpyDeviceToDevice, st));
            HC_HIP(hipStreamSynchronize(st));
            if (acc.p) (void)hipFree(acc.p);
            acc.p = bigger;
            acc.cap = want * sizeof(hc_sfo_rec);
            res_cap = want;
        }
        HC_HIP(hc::finder_emit(d_by_sfo.as<hc::SeqRef>(), d_h0.as<uint64_t>(), d_kout.as<uint32_t>(), d_flag.as<uint32_t>(), d_pos.as<uint32_t>(), M,
                               (hc_sfo_rec*)acc.p + R, st));
        R += Rb;
    }
    HC_HIP(hipStreamSynchronize(st));
    lap("expand/sort/unique/verify/emit");
    if (R == 0) {
        remember(nullptr, 0);
        return HC_OK;
    }
    if (R >= (1ull << 31)) return fail(HC_ERR_ARG, "hc_find_overlaps: more than 2^31 overlaps");
    if (!c->d_found || c->found_cap < R) {  // the result: the context's until the next call (grow-only)
        if (c->d_found) (void)hipFree(c->d_found);
        c->d_found = nullptr;
        c->found_cap = 0;
        HC_HIP(hipMalloc((void**)&c->d_found, (R + R / 8) * sizeof(hc_sfo_rec)));
        c->found_cap = R + R / 8;
    }
    d_r1.p = c->d_found;
    if (batches.size() > 1) {  // every batch is sorted; one more sort (key, position) + gather for the global order
        // the batches' buffers are idle now: they hold the keys and positions of this sort when they are large enough
        DevBuf own[4];  // what they cannot hold; freed on every return path
        auto room = [&](const DevBuf& idle, size_t bytes, DevBuf& fallback, void** p) -> hipError_t {
            if (idle.slot && idle.slot->cap >= bytes) {
                *p = idle.slot->p;
                return hipSuccess;
            }
            const hipError_t e = hipMalloc(&fallback.own, bytes ? bytes : 16);
            *p = fallback.own;
            return e;
        };
        void *sk0 = nullptr, *sk1 = nullptr, *si0 = nullptr, *si1 = nullptr, *stmp = nullptr;
        HC_HIP(room(d_h0, R * 8, own[0], &sk0));
        HC_HIP(room(d_h1, R * 8, own[1], &sk1));
        HC_HIP(room(d_kout, R * 8, own[2], &si0));
        HC_HIP(room(d_flag, R * 8, own[3], &si1));
        HC_HIP(hc::finder_rekey((const hc_sfo_rec*)acc.p, R, (uint64_t*)sk0, (uint64_t*)si0, st));
        size_t b = 0;
        HC_HIP(hc::finder_sort_pairs(nullptr, b, (uint64_t*)sk0, (uint64_t*)sk1, (uint64_t*)si0, (uint64_t*)si1, R, 64, st));
        if (d_pos.slot && d_pos.slot->cap >= b) {
            stmp = d_pos.slot->p;
        } else {  // (a gigabyte at config 3's size: one more grow-only slot, not a block of this call's own)
            DevBuf d_stmp;
            HC_ALLOC(d_stmp, b);
            stmp = d_stmp.p;
        }
        HC_HIP(hc::finder_sort_pairs(stmp, b, (uint64_t*)sk0, (uint64_t*)sk1, (uint64_t*)si0, (uint64_t*)si1, R, 64, st));
        HC_HIP(hc::finder_gather((const hc_sfo_rec*)acc.p, (const uint64_t*)si1, R, (hc_sfo_rec*)d_r1.p, st));
        HC_HIP(hipStreamSynchronize(st));
        lap("global order of the batches");
    } else {
        HC_HIP(hipMemcpyAsync(d_r1.p, acc.p, R * sizeof(hc_sfo_rec), hipMemcpyDeviceToDevice, st));
        HC_HIP(hipStreamSynchronize(st));
    }
    *n_out = R;
    const uint64_t take = R < cap ? R : cap;
    if (take) HC_HIP(hc::copy_to_pageable_host(out, d_r1.p, take * sizeof(hc_sfo_rec)));
    lap("copy to host");
    remember((hc_sfo_rec*)d_r1.p, R);
    return HC_OK;
}

// ---- the SFO ingest straight from the records of hc_find_overlaps ------------------------------------------------------
}  // extern "C"

// The ingest's result as text in memory (hc_ctx.h): what hc_found_to_overlaps writes to its file, and what the stage's
// reads -> graph call hands to its own text blocks without a file in between.
int hc_found_to_overlaps_text(hc_ctx* c, uint64_t num_singles, uint64_t num_pairs, std::string& text, uint64_t* n_lines) {
    if (!c) return fail(HC_ERR_ARG, "hc_found_to_overlaps: null argument");
    if (!c->found_valid) return fail(HC_ERR_STATE, "hc_found_to_overlaps: hc_find_overlaps has not been called on this read set");
    const uint64_t n = c->n_found;
    const bool timing = getenv("HC_SFO_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    struct Freed {
        void* host = nullptr;
        ~Freed() { free(host); }
    } mem;
    IngestScratch dmalloc(c);
    try {
        HC_HIP(hipSetDevice(c->device));
        hc::BoundForNow bound(hc::cpus_near_device(c->device));  // the matching threads read the page-locked ring: next to the device
        hipStream_t st = c->stream;
        uint64_t k = 0;
        text.clear();
        const double t0 = now();
        bool sorted_on_device = false;
        if (n && n < 0x7FFFFFF0ull) {
            hc::SfoFlipped *d_flip = nullptr, *d_sorted = nullptr;
            uint64_t *d_k[3] = {nullptr, nullptr, nullptr}, *d_ka = nullptr, *d_kb = nullptr;
            uint32_t *d_pa = nullptr, *d_pb = nullptr;
            unsigned long long* d_status = nullptr;
            void* d_tmp = nullptr;
            HC_HIP(dmalloc(n * sizeof(hc::SfoFlipped), (void**)&d_flip));
            HC_HIP(dmalloc(n * sizeof(hc::SfoFlipped), (void**)&d_sorted));
            for (auto& kk : d_k) HC_HIP(dmalloc(n * 8, (void**)&kk));
            HC_HIP(dmalloc(n * 8, (void**)&d_ka));
            HC_HIP(dmalloc(n * 8, (void**)&d_kb));
            HC_HIP(dmalloc(n * 4, (void**)&d_pa));
            HC_HIP(dmalloc(n * 4, (void**)&d_pb));
            HC_HIP(dmalloc(8, (void**)&d_status));
            size_t tmp_bytes = 0;
            HC_HIP(hc::sort_pairs_u64_u32(nullptr, tmp_bytes, d_ka, d_kb, d_pa, d_pb, (uint32_t)n, 64, st));
            HC_HIP(dmalloc(tmp_bytes, &d_tmp));
            HC_HIP(hipMemsetAsync(d_status, 0, 8, st));
            HC_HIP(hipStreamSynchronize(st));
            const double t_alloc = now();
            HC_HIP(hc::sfo_flip(c->d_found, n, num_singles, num_pairs, d_flip, d_k[0], d_k[1], d_k[2], d_pa, d_status, st));
            uint32_t *perm = d_pa, *perm_next = d_pb;
            for (int ch = 0; ch < 3; ch++) {  // least significant 64 bits first; every sort is stable
                const uint64_t* keys = d_k[ch];
                if (ch) {
                    HC_HIP(hc::fno_gather_keys(d_k[ch], perm, n, d_ka, st));
                    keys = d_ka;
                }
                size_t b = tmp_bytes;
                HC_HIP(hc::sort_pairs_u64_u32(d_tmp, b, keys, d_kb, perm, perm_next, (uint32_t)n, 64, st));
                std::swap(perm, perm_next);
            }
            HC_HIP(hc::sfo_gather(d_flip, perm, n, d_sorted, st));
            // Only the records the matching can see anything of leave the device (hc_sfo_kernels.hip: lines between unpaired reads,
            // groups of two lines and more, the lines that close them): d_flip, spent, takes them; the keys' buffers the flags
            // and the places.  HC_SFO_FILTER=0: all of them, as before.
            uint64_t n_out = n;
            const hc::SfoFlipped* d_send = d_sorted;
            const bool filter = !(getenv("HC_SFO_FILTER") && atoi(getenv("HC_SFO_FILTER")) == 0);
            if (filter) {
                uint8_t *d_grouped = (uint8_t*)d_k[0], *d_keep = (uint8_t*)d_k[1];
                uint32_t* d_idx = (uint32_t*)d_k[2];
                unsigned long long* d_cnt = (unsigned long long*)d_ka;  // [0] grouped records, [1] kept records
                const size_t sel_bytes = hc::prims::select_temp_bytes(n);
                if (sel_bytes > tmp_bytes) return fail(HC_ERR_STATE, "hc_found_to_overlaps: scratch smaller than the selection needs");
                HC_HIP(hc::sfo_classify(d_sorted, n, num_singles, num_pairs, d_grouped, d_keep, st));
                HC_HIP(hc::prims::select_flagged(d_tmp, tmp_bytes, d_grouped, n, d_idx, d_cnt, st));
                unsigned long long m = 0;
                HC_HIP(hipMemcpyAsync(&m, d_cnt, 8, hipMemcpyDeviceToHost, st));
                HC_HIP(hipStreamSynchronize(st));
                HC_HIP(hc::sfo_groups(d_sorted, d_idx, m, num_singles, num_pairs, d_keep, st));
                HC_HIP(hc::prims::select_flagged(d_tmp, tmp_bytes, d_keep, n, d_idx, d_cnt + 1, st));
                unsigned long long kept = 0;
                HC_HIP(hipMemcpyAsync(&kept, d_cnt + 1, 8, hipMemcpyDeviceToHost, st));
                HC_HIP(hipStreamSynchronize(st));
                HC_HIP(hc::sfo_gather_kept(d_sorted, d_idx, kept, d_flip, st));
                n_out = kept;
                d_send = d_flip;
            }
            unsigned long long status = 0;
            HC_HIP(hipMemcpyAsync(&status, d_status, 8, hipMemcpyDeviceToHost, st));
            // the page-locked ring the sorted records come back through (kept with the context)
            const uint64_t ring = 2u << 20;  // records per buffer of the ring: 64 MiB
            uint64_t chunk = ring;            // records per copy (HC_SFO_CHUNK: test knob, small chunks on small inputs)
            if (const char* e = getenv("HC_SFO_CHUNK")) chunk = std::min<uint64_t>(ring, std::max<uint64_t>(1, strtoull(e, nullptr, 10)));
            if (!c->h_ingest[0]) {
                for (int t = 0; t < 2; t++) HC_HIP(hipHostMalloc(&c->h_ingest[t], ring * sizeof(hc::SfoFlipped), hipHostMallocDefault));
                c->h_ingest_cap = ring * sizeof(hc::SfoFlipped);
            }
            HC_HIP(hipStreamSynchronize(st));
            const double t1 = now();
            if (!status) {
                // copy of chunk j + 1 beside the matching of chunk j
                hc::SfoSortedMatcher matcher((long)num_singles, (long)num_pairs);
                const uint64_t n_chunks = (n_out + chunk - 1) / chunk;
                auto count_of = [&](uint64_t j) { return std::min(chunk, n_out - j * chunk); };
                if (n_out) HC_HIP(hipMemcpyAsync(c->h_ingest[0], d_send, count_of(0) * sizeof(hc::SfoFlipped), hipMemcpyDeviceToHost, st));
                for (uint64_t j = 0; j < n_chunks; j++) {
                    HC_HIP(hipStreamSynchronize(st));  // chunk j has arrived
                    if (j + 1 < n_chunks)
                        HC_HIP(hipMemcpyAsync(c->h_ingest[(j + 1) & 1], d_send + (j + 1) * chunk, count_of(j + 1) * sizeof(hc::SfoFlipped),
                                              hipMemcpyDeviceToHost, st));
                    matcher.feed((const hc::SfoFlipped*)c->h_ingest[j & 1], count_of(j));
                }
                const double t2 = now();
                text = matcher.finish(k);
                sorted_on_device = true;
                if (timing)
                    fprintf(stderr, "hc_found_to_overlaps: device blocks %.3f s, flip + 3 sorts + gather + filter on the device %.3f s (%llu of %llu records leave it), copy + match %.3f s, stitch %.3f s\n",
                            t_alloc - t0, t1 - t_alloc, (unsigned long long)n_out, (unsigned long long)n, t2 - t1, now() - t2);
            }
        }
        if (!sorted_on_device) {  // nothing found, an id or a number the device's keys do not hold: the host path sorts, and reports
            mem.host = malloc(n ? n * sizeof(hc_sfo_rec) : 16);
            if (!mem.host) return fail(HC_ERR_NOMEM, "hc_found_to_overlaps: out of host memory");
            if (n) HC_HIP(hipMemcpy(mem.host, c->d_found, n * sizeof(hc_sfo_rec), hipMemcpyDeviceToHost));
            text = hc::sfo_records_to_overlaps((const hc_sfo_rec*)mem.host, n, (long)num_singles, (long)num_pairs, k);
        }
        if (n_lines) *n_lines = k;
        return HC_OK;
    } catch (const hc::FatalError& e) {
        return fail(e.status, e.what);
    } catch (const std::bad_alloc&) {
        return fail(HC_ERR_NOMEM, "hc_found_to_overlaps: out of host memory");
    }
}

extern "C" {

// The ingest WITHOUT text or host (round 6; SURVEY.md 8(f4): "would remove ... the text file altogether"): flip, the script's sort, its
// matching and both its `uniq`s on the device; what comes out is the overlaps file's lines as hc_line_rec in device memory, in file
// order — what the stage's text kernels would parse from the script's output.  *d_lines stays valid until the next call or hc_set_reads.
// HC_ERR_NOT_ON_DEVICE: an SFO id / number the device keys do not hold, an assert of the script's matching, a group of thousands of
// lines — the caller takes hc_found_to_overlaps' route, which raises what the script raises.
int hc_found_to_lines_device(hc_ctx* c, uint64_t num_singles, uint64_t num_pairs, const hc_line_rec** d_lines, uint64_t* n_lines) {
    if (!c || !d_lines || !n_lines) return fail(HC_ERR_ARG, "hc_found_to_lines_device: null argument");
    *d_lines = nullptr;
    *n_lines = 0;
    if (!c->found_valid) return fail(HC_ERR_STATE, "hc_found_to_lines_device: hc_find_overlaps has not been called on this read set");
    const uint64_t n = c->n_found;
    if (n == 0) return HC_OK;
    if (n >= 0x7FFFFFF0ull) return fail(HC_ERR_NOT_ON_DEVICE, "hc_found_to_lines_device: not on the device (2^31 records and more)");
    const bool timing = getenv("HC_SFO_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    IngestScratch dmalloc(c);
    HC_HIP(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const double t0 = now();
    hc::SfoFlipped *d_flip = nullptr, *d_sorted = nullptr;
    uint64_t *d_k[3] = {nullptr, nullptr, nullptr}, *d_ka = nullptr, *d_kb = nullptr;
    uint32_t *d_pa = nullptr, *d_pb = nullptr;
    unsigned long long* d_status = nullptr;  // [0] status, [1] grouped records, [2] groups, [3] lines, [4] equal neighbours, [5] lines kept
    void* d_tmp = nullptr;
    HC_HIP(dmalloc(n * sizeof(hc::SfoFlipped), (void**)&d_flip));
    HC_HIP(dmalloc(n * sizeof(hc::SfoFlipped), (void**)&d_sorted));
    for (auto& kk : d_k) HC_HIP(dmalloc(n * 8, (void**)&kk));
    HC_HIP(dmalloc(n * 8, (void**)&d_ka));
    HC_HIP(dmalloc(n * 8, (void**)&d_kb));
    HC_HIP(dmalloc(n * 4, (void**)&d_pa));
    HC_HIP(dmalloc(n * 4, (void**)&d_pb));
    HC_HIP(dmalloc(64, (void**)&d_status));
    size_t tmp_bytes = 0;
    HC_HIP(hc::sort_pairs_u64_u32(nullptr, tmp_bytes, d_ka, d_kb, d_pa, d_pb, (uint32_t)n, 64, st));
    tmp_bytes = std::max(tmp_bytes, std::max(hc::prims::select_temp_bytes(n), hc::prims::scan_temp_bytes(n + 1, 4)));
    HC_HIP(dmalloc(tmp_bytes, &d_tmp));
    HC_HIP(hipMemsetAsync(d_status, 0, 64, st));
    // flip + the script's sort (three stable radix sorts over the 192-bit key), as hc_found_to_overlaps does
    HC_HIP(hc::sfo_flip(c->d_found, n, num_singles, num_pairs, d_flip, d_k[0], d_k[1], d_k[2], d_pa, d_status, st));
    uint32_t *perm = d_pa, *perm_next = d_pb;
    for (int ch = 0; ch < 3; ch++) {
        const uint64_t* keys = d_k[ch];
        if (ch) {
            HC_HIP(hc::fno_gather_keys(d_k[ch], perm, n, d_ka, st));
            keys = d_ka;
        }
        size_t b = tmp_bytes;
        HC_HIP(hc::sort_pairs_u64_u32(d_tmp, b, keys, d_kb, perm, perm_next, (uint32_t)n, 64, st));
        std::swap(perm, perm_next);
    }
    HC_HIP(hc::sfo_gather(d_flip, perm, n, d_sorted, st));
    // which sorted records the matching sees (its first `uniq`, self overlaps, lines between unpaired reads, groups)
    uint8_t *d_grouped = (uint8_t*)d_k[0], *d_single = (uint8_t*)d_k[1];
    uint32_t* d_idx = (uint32_t*)d_k[2];                 // the grouped records' places, in order
    uint8_t* d_start = (uint8_t*)d_ka;                   // per grouped record: opens a group
    uint32_t* d_starts = (uint32_t*)d_kb;                // the grouped records that open a group, in order
    uint32_t* d_emit = (uint32_t*)d_flip;                // per sorted record: lines it makes the script write ([n + 1]: d_flip holds 8 n words)
    uint32_t* d_off = d_emit + (n + 1);
    HC_HIP(hc::sfo_classify(d_sorted, n, num_singles, num_pairs, d_grouped, d_single, st));
    HC_HIP(hc::prims::select_flagged(d_tmp, tmp_bytes, d_grouped, n, d_idx, d_status + 1, st));
    unsigned long long host[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    HC_HIP(hipMemcpyAsync(host, d_status, 16, hipMemcpyDeviceToHost, st));
    HC_HIP(hipStreamSynchronize(st));
    if (host[0]) return fail(HC_ERR_NOT_ON_DEVICE, "hc_found_to_lines_device: not on the device (an SFO id or number the device keys do not hold)");
    const uint64_t m = host[1];
    HC_HIP(hc::sfo_group_starts(d_sorted, d_idx, m, num_singles, num_pairs, d_start, st));
    HC_HIP(hc::prims::select_flagged(d_tmp, tmp_bytes, d_start, m, d_starts, d_status + 2, st));
    HC_HIP(hipMemsetAsync(d_emit, 0, (n + 1) * 4, st));
    HC_HIP(hipMemcpyAsync(host + 2, d_status + 2, 8, hipMemcpyDeviceToHost, st));
    HC_HIP(hipStreamSynchronize(st));
    const uint64_t G = host[2];
    // lines per record that makes the script write, their places by a scan, the lines
    HC_HIP(hc::sfo_single_lines(false, d_sorted, d_single, n, num_singles, num_pairs, d_emit, nullptr, nullptr, d_status, st));
    HC_HIP(hc::sfo_match_groups(false, d_sorted, d_idx, d_starts, G, num_singles, num_pairs, d_emit, nullptr, nullptr, d_status, st));
    HC_HIP(hc::prims::exclusive_sum(d_tmp, tmp_bytes, d_emit, d_off, n + 1, st));
    uint32_t total32 = 0;
    HC_HIP(hipMemcpyAsync(&total32, d_off + n, 4, hipMemcpyDeviceToHost, st));
    HC_HIP(hipMemcpyAsync(host, d_status, 8, hipMemcpyDeviceToHost, st));
    HC_HIP(hipStreamSynchronize(st));
    if (host[0]) return fail(HC_ERR_NOT_ON_DEVICE, "hc_found_to_lines_device: not on the device (an assert of the script's matching, or thousands of lines for one pair of reads)");
    const uint64_t total = total32;
    if (total > c->found_lines_cap) {
        if (c->d_found_lines) (void)hipFree(c->d_found_lines);
        c->d_found_lines = nullptr;
        c->found_lines_cap = 0;
        HC_HIP(hipMalloc((void**)&c->d_found_lines, (total + total / 16 + 1024) * sizeof(hc_line_rec)));
        c->found_lines_cap = total + total / 16 + 1024;
    }
    if (total == 0) return HC_OK;
    // the lines go to scratch first (the sorted records' keys are spent), the script's last `uniq` decides what stays
    hc_line_rec* d_raw = nullptr;
    HC_HIP(dmalloc(total * sizeof(hc_line_rec), (void**)&d_raw));
    HC_HIP(hc::sfo_single_lines(true, d_sorted, d_single, n, num_singles, num_pairs, d_emit, d_off, d_raw, d_status, st));
    HC_HIP(hc::sfo_match_groups(true, d_sorted, d_idx, d_starts, G, num_singles, num_pairs, d_emit, d_off, d_raw, d_status, st));
    uint8_t* d_keep = (uint8_t*)d_k[0];
    if (total > n * 8) return fail(HC_ERR_NOT_ON_DEVICE, "hc_found_to_lines_device: not on the device (more lines than the flags have room for)");
    HC_HIP(hc::sfo_uniq_lines(d_raw, total, d_keep, d_status + 4, st));
    HC_HIP(hipMemcpyAsync(host, d_status, 40, hipMemcpyDeviceToHost, st));
    HC_HIP(hipStreamSynchronize(st));
    if (host[0]) return fail(HC_ERR_NOT_ON_DEVICE, "hc_found_to_lines_device: not on the device (an assert of the script's matching)");
    uint64_t kept = total;
    if (host[4]) {  // equal neighbours (rare): the kept lines, in order
        if (total * 4 > n * 8 || hc::prims::select_temp_bytes(total) > tmp_bytes)
            return fail(HC_ERR_NOT_ON_DEVICE, "hc_found_to_lines_device: not on the device (more lines than the index has room for)");
        uint32_t* d_kidx = (uint32_t*)d_k[1];
        HC_HIP(hc::prims::select_flagged(d_tmp, tmp_bytes, d_keep, total, d_kidx, d_status + 5, st));
        HC_HIP(hipMemcpyAsync(host + 5, d_status + 5, 8, hipMemcpyDeviceToHost, st));
        HC_HIP(hipStreamSynchronize(st));
        kept = host[5];
        HC_HIP(hc::sfo_gather_lines(d_raw, d_kidx, kept, c->d_found_lines, st));
    } else {
        HC_HIP(hipMemcpyAsync(c->d_found_lines, d_raw, total * sizeof(hc_line_rec), hipMemcpyDeviceToDevice, st));
    }
    HC_HIP(hipStreamSynchronize(st));
    if (timing)
        fprintf(stderr, "hc_found_to_lines_device: %llu SFO records -> %llu grouped, %llu groups -> %llu lines (%llu equal neighbours dropped) in %.3f s, all on the device\n",
                (unsigned long long)n, (unsigned long long)m, (unsigned long long)G, (unsigned long long)kept, (unsigned long long)(total - kept), now() - t0);
    *d_lines = c->d_found_lines;
    *n_lines = kept;
    return HC_OK;
}

// SFO records from elsewhere (a rust-overlaps file parsed by the caller; the tests' hand-made records) take the place of the finder's: the
// ingest entry points (hc_found_to_overlaps, hc_found_to_lines_device) then run on them.  Needs a read store (the ids are its sequences').
int hc_set_found_records(hc_ctx* c, const hc_sfo_rec* recs, uint64_t n) {
    if (!c || (n && !recs)) return fail(HC_ERR_ARG, "hc_set_found_records: null argument");
    if (!c->have_reads) return fail(HC_ERR_STATE, "hc_set_found_records: hc_set_reads has not been called");
    HC_HIP(hipSetDevice(c->device));
    c->n_found = 0;
    c->found_valid = false;
    if (n) {
        if (!c->d_found || c->found_cap < n) {
            if (c->d_found) (void)hipFree(c->d_found);
            c->d_found = nullptr;
            c->found_cap = 0;
            HC_HIP(hipMalloc((void**)&c->d_found, n * sizeof(hc_sfo_rec)));
            c->found_cap = n;
        }
        HC_HIP(hipMemcpy(c->d_found, recs, n * sizeof(hc_sfo_rec), hipMemcpyHostToDevice));
    }
    c->n_found = n;
    c->found_err = -1;  // (no finder arguments describe these records: the next hc_find_overlaps computes)
    c->found_min = 0;
    c->found_flags = 0;
    c->found_valid = true;
    return HC_OK;
}

// The SFO FILE's text in the finder's place, read on the device (round 6): 32 MiB chunks of the text (cut behind a newline) go to the device as
// they are — three stations, the copies of chunks k + 1, k + 2 beside the kernels of chunk k — and every chunk's lines (the overlaps file's own line-start
// kernels, hc_text_kernels.hip) are read by one lane each (sfo_parse_text_kernel) into the context's found records at the place a chain of
// line counters assigns.  A canonical file only — eight fields, single tabs, plain decimal numbers: what rust-overlaps writes; anything else:
// HC_ERR_NOT_ON_DEVICE, and the host's general path (hc_sfo2overlaps' code) takes the file and owns its errors.
int hc_set_found_from_sfo_text(hc_ctx* c, const char* text, uint64_t n_bytes, uint64_t* n_records) {
    if (!c || (n_bytes && !text)) return fail(HC_ERR_ARG, "hc_set_found_from_sfo_text: null argument");
    if (!c->have_reads) return fail(HC_ERR_STATE, "hc_set_found_from_sfo_text: hc_set_reads has not been called");
    if (n_records) *n_records = 0;
    HC_HIP(hipSetDevice(c->device));
    c->n_found = 0;
    c->found_valid = false;
    // room for the records: a canonical line has 16 bytes and more ("0\t0\tN\t0\t0\t0\t0\t0\n"), so n_bytes / 16 + 1 bounds their number — a file
    // with more lines is not canonical and the parse kernel says so (out_cap).  The lines themselves are counted on the device, chunk by
    // chunk (the line chain).  (Round 6's first form counted the newlines on the host's threads first: 15 ms at config 3's size, in
    // front of the first copy.)
    const uint64_t lines = n_bytes / 16 + 1;
    if (lines >= 0x7FFFFFF0ull) return fail(HC_ERR_NOT_ON_DEVICE, "hc_set_found_from_sfo_text: not on the device (34 GB of text and more)");
    if (n_bytes == 0) {
        c->found_err = -1;
        c->found_min = 0;
        c->found_flags = 0;
        c->found_valid = true;
        return HC_OK;
    }
    if (!c->d_found || c->found_cap < lines) {
        if (c->d_found) (void)hipFree(c->d_found);
        c->d_found = nullptr;
        c->found_cap = 0;
        HC_HIP(hipMalloc((void**)&c->d_found, lines * sizeof(hc_sfo_rec)));
        c->found_cap = lines;
    }
    // Chunks of 32 MiB, three stations (a page-locked host buffer the context keeps, a device buffer, line-start arrays, counters, events):
    // the host's threads copy chunk k out of the caller's (pageable, usually mapped-file) memory into station k % 3's page-locked buffer —
    // in parallel slices: one thread through the runtime's own staging moved 27 GB/s at config 3's size —, the copy stream takes it to the
    // device, the kernels of chunk k follow on the context's stream; chunk k + 1 is being copied out meanwhile.
    const uint64_t C = 32ull << 20;            // bytes per chunk
    constexpr int kStations = 3;
    const uint32_t max_lines = (uint32_t)(C / 16 + 2);  // (a canonical line has 16 bytes and more; a chunk with more lines is not canonical: overflow -> status)
    const uint32_t n_tiles = (uint32_t)(C / 4096 + 2);
    const uint64_t n_chunks_max = n_bytes / (C / 2) + 2;
    for (int k = 0; k < kStations; k++)
        if (!c->h_sfo_text[k]) HC_HIP(hipHostMalloc(&c->h_sfo_text[k], C + 64, hipHostMallocDefault));
    struct Bufs {
        char* text[kStations] = {};
        uint32_t *tile_cnt[kStations] = {}, *tile_off[kStations] = {}, *line_start[kStations] = {};
        unsigned long long *counters[kStations] = {}, *chain = nullptr, *status = nullptr;
        hipEvent_t copied[kStations] = {}, parsed[kStations] = {};
        hipStream_t copy = nullptr;
        ~Bufs() {
            for (int k = 0; k < kStations; k++) {
                if (text[k]) (void)hipFree(text[k]);
                if (tile_cnt[k]) (void)hipFree(tile_cnt[k]);
                if (tile_off[k]) (void)hipFree(tile_off[k]);
                if (line_start[k]) (void)hipFree(line_start[k]);
                if (counters[k]) (void)hipFree(counters[k]);
                if (copied[k]) (void)hipEventDestroy(copied[k]);
                if (parsed[k]) (void)hipEventDestroy(parsed[k]);
            }
            if (chain) (void)hipFree(chain);
            if (status) (void)hipFree(status);
            if (copy) (void)hipStreamDestroy(copy);
        }
    } b;
    for (int k = 0; k < kStations; k++) {
        HC_HIP(hipMalloc((void**)&b.text[k], C + 64));
        HC_HIP(hipMalloc((void**)&b.tile_cnt[k], (size_t)n_tiles * 4));
        HC_HIP(hipMalloc((void**)&b.tile_off[k], (size_t)n_tiles * 4));
        HC_HIP(hipMalloc((void**)&b.line_start[k], ((size_t)max_lines + 2) * 4));
        HC_HIP(hipMalloc((void**)&b.counters[k], hc::kTextCounters * sizeof(unsigned long long)));
        HC_HIP(hipEventCreateWithFlags(&b.copied[k], hipEventDisableTiming));
        HC_HIP(hipEventCreateWithFlags(&b.parsed[k], hipEventDisableTiming));
    }
    HC_HIP(hipMalloc((void**)&b.chain, (n_chunks_max + 1) * sizeof(unsigned long long)));
    HC_HIP(hipMalloc((void**)&b.status, sizeof(unsigned long long)));
    HC_HIP(hipStreamCreateWithFlags(&b.copy, hipStreamNonBlocking));
    hipStream_t st = c->stream;
    HC_HIP(hipMemsetAsync(b.chain, 0, sizeof(unsigned long long), st));
    HC_HIP(hipMemsetAsync(b.status, 0, sizeof(unsigned long long), st));
    HC_HIP(hipStreamSynchronize(st));
    // the copiers: T - 1 threads beside the caller's, one slice of the chunk each (HC_SFO_COPY_THREADS; 1: the caller's thread alone)
    struct Copiers {
        std::vector<std::thread> th;
        std::mutex m;
        std::condition_variable go, done;
        const char* src = nullptr;
        char* dst = nullptr;
        size_t len = 0;
        unsigned T = 1, gen = 0, pending = 0;
        bool stop = false;
        static void slice(const char* src, char* dst, size_t len, unsigned t, unsigned T) {
            const size_t a = (len * t / T) & ~(size_t)63, e = t + 1 == T ? len : (len * (t + 1) / T) & ~(size_t)63;
            if (e > a) memcpy(dst + a, src + a, e - a);
        }
        void start(unsigned threads) {
            T = threads < 1 ? 1 : threads;
            for (unsigned t = 1; t < T; t++)
                th.emplace_back([this, t] {
                    unsigned seen = 0;
                    for (;;) {
                        std::unique_lock<std::mutex> lk(m);
                        go.wait(lk, [&] { return stop || gen != seen; });
                        if (stop) return;
                        seen = gen;
                        const char* s_ = src;
                        char* d_ = dst;
                        const size_t n_ = len;
                        lk.unlock();
                        slice(s_, d_, n_, t, T);
                        lk.lock();
                        if (--pending == 0) done.notify_one();
                    }
                });
        }
        void copy(const char* s_, char* d_, size_t n_) {
            if (T == 1 || n_ < (1u << 20)) {
                memcpy(d_, s_, n_);
                return;
            }
            {
                std::lock_guard<std::mutex> lk(m);
                src = s_;
                dst = d_;
                len = n_;
                pending = T - 1;
                gen++;
            }
            go.notify_all();
            slice(s_, d_, n_, 0, T);
            std::unique_lock<std::mutex> lk(m);
            done.wait(lk, [&] { return pending == 0; });
        }
        ~Copiers() {
            {
                std::lock_guard<std::mutex> lk(m);
                stop = true;
            }
            go.notify_all();
            for (auto& x : th) x.join();
        }
    } copiers;
    {
        unsigned T = std::thread::hardware_concurrency();
        T = T == 0 ? 1 : (T > 16 ? 16 : T);
        if (const char* e = getenv("HC_SFO_COPY_THREADS")) T = (unsigned)std::max(1, std::min(64, atoi(e)));
        if (n_bytes < (8u << 20)) T = 1;
        copiers.start(T);
    }
    uint64_t pos = 0, k = 0;
    while (pos < n_bytes) {
        uint64_t len = n_bytes - pos < C ? n_bytes - pos : C;
        if (pos + len < n_bytes) {  // cut behind the last newline of the stretch
            const void* nl = memrchr(text + pos, '\n', (size_t)len);
            if (!nl) return fail(HC_ERR_NOT_ON_DEVICE, "hc_set_found_from_sfo_text: not on the device (a line of 32 MiB and more)");
            len = (uint64_t)((const char*)nl - (text + pos)) + 1;
        }
        if (k >= n_chunks_max) return fail(HC_ERR_NOT_ON_DEVICE, "hc_set_found_from_sfo_text: not on the device (more chunks than planned)");
        const int j = (int)(k % kStations);
        if (k >= (uint64_t)kStations) HC_HIP(hipEventSynchronize(b.copied[j]));  // the station's host buffer has left for the device
        copiers.copy(text + pos, (char*)c->h_sfo_text[j], (size_t)len);
        if (k >= (uint64_t)kStations) HC_HIP(hipStreamWaitEvent(b.copy, b.parsed[j], 0));  // the station's device buffer has been read
        HC_HIP(hipMemcpyAsync(b.text[j], c->h_sfo_text[j], len, hipMemcpyHostToDevice, b.copy));
        HC_HIP(hipEventRecord(b.copied[j], b.copy));
        HC_HIP(hipStreamWaitEvent(st, b.copied[j], 0));
        HC_HIP(hc::launch_text_count(b.text[j], len, b.tile_cnt[j], st));
        HC_HIP(hc::launch_text_scan(b.text[j], len, b.tile_cnt[j], b.tile_off[j], max_lines, b.line_start[j], b.counters[j], b.chain + k, b.chain + k + 1, st));
        HC_HIP(hc::launch_text_line_starts(b.text[j], len, b.tile_off[j], max_lines, b.line_start[j], st));
        HC_HIP(hc::sfo_parse_text(b.text[j], b.line_start[j], max_lines, b.counters[j], b.chain + k, c->d_found, lines, b.status, st));
        HC_HIP(hipEventRecord(b.parsed[j], st));
        pos += len;
        k++;
    }
    unsigned long long status = 0, total = 0;
    HC_HIP(hipMemcpyAsync(&status, b.status, sizeof status, hipMemcpyDeviceToHost, st));
    HC_HIP(hipMemcpyAsync(&total, b.chain + k, sizeof total, hipMemcpyDeviceToHost, st));
    HC_HIP(hipStreamSynchronize(st));
    HC_HIP(hipStreamSynchronize(b.copy));
    if (status || total > lines) return fail(HC_ERR_NOT_ON_DEVICE, "hc_set_found_from_sfo_text: not on the device (a line that is not canonical)");
    c->n_found = total;
    c->found_err = -1;
    c->found_min = 0;
    c->found_flags = 0;
    c->found_valid = true;
    if (n_records) *n_records = total;
    return HC_OK;
}

// the lines hc_found_to_lines_device left on the device, copied to the host (tests, tools)
int hc_found_lines_fetch(hc_ctx* c, const hc_line_rec* d_lines, uint64_t n, hc_line_rec* out) {
    if (!c || (n && (!d_lines || !out))) return fail(HC_ERR_ARG, "hc_found_lines_fetch: null argument");
    HC_HIP(hipSetDevice(c->device));
    if (n) HC_HIP(hipMemcpy(out, d_lines, n * sizeof(hc_line_rec), hipMemcpyDeviceToHost));
    return HC_OK;
}

int hc_found_to_overlaps(hc_ctx* c, const char* out_path, uint64_t num_singles, uint64_t num_pairs, uint64_t* n_lines) {
    if (!c || !out_path) return fail(HC_ERR_ARG, "hc_found_to_overlaps: null argument");
    std::string text;
    const int rc = hc_found_to_overlaps_text(c, num_singles, num_pairs, text, n_lines);
    if (rc) return rc;
    FILE* o = fopen(out_path, "wb");
    if (!o) return fail(HC_ERR_IO, std::string("cannot write ") + out_path);
    const size_t w = text.empty() ? 0 : fwrite(text.data(), 1, text.size(), o);
    if (fclose(o) != 0 || w != text.size()) return fail(HC_ERR_IO, std::string("short write to ") + out_path);
    return HC_OK;
}

}  // extern "C"
