// hc_api.cpp — C-ABI glue of libhcedge.so (include/hcedge.h): context, HBM read
// store, host-built log-probability table, threshold inversion, launches.
// Compiled with hipcc for gfx950.  There is NO CPU fallback in this library: without
// a HIP device hc_create fails with HC_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <array>
#include <limits>
#include <new>
#include <string>
#include <vector>

#include "../../include/hcedge.h"
#include "hc_ctx.h"

static thread_local std::string g_last_error;

static int fail(int status, const std::string& what) {
    g_last_error = what;
    return status;
}
namespace hc {
int set_last_error(int status, const std::string& what) { return fail(status, what); }
}

// --------------------------------------------------------------------------
// Threshold inversion: exp(x) > T decided in x-space.
// glibc's exp/log are accurate to < 1 ULP, and the true exp is monotone, so
//   true exp(x) > T + 1.5 ulp(T)  =>  fl(exp(x)) > T,   true exp(x) < T - 1.5 ulp(T)  =>  fl(exp(x)) <= T.
// In x that is |x - ln T| > 1.5 * 2^-52 (+ the error of log()).  Everything inside
// [ln T - d, ln T + d], d = 2^-49 * max(1, |ln T|), is handed to the host libm
// (HC_CLS_AMBIG); the band covers ~1e-13 of the x values that occur.
static hc::Band make_band(double T) {
    hc::Band b;
    const double inf = std::numeric_limits<double>::infinity();
    if (T != T) {  // NaN: `score > T` is never true
        b.lo = inf;
        b.hi = inf;
    } else if (T < 0) {  // every score (>= 0, including 0 = exp(-inf)) passes
        b.lo = -inf;
        b.hi = -inf;
    } else if (T == 0) {  // exp(x) > 0 unless it underflows to 0 (x < ~ -745.13)
        b.lo = -760.0;
        b.hi = -740.0;
    } else if (T >= 1) {  // x <= 0 always, exp(x) <= 1 <= T
        b.lo = inf;
        b.hi = inf;
    } else {
        const double lt = std::log(T);
        const double d = std::ldexp(1.0, -49) * std::fmax(1.0, std::fabs(lt));
        b.lo = lt - d;
        b.hi = lt + d;
    }
    return b;
}
// x = -inf (score 0): -inf > hi is false unless hi = -inf is meant as "always": handle T < 0 on the device by
// comparing with >= when hi == -inf?  Simpler: for T < 0 we use hi = -inf and the device test `x > hi`
// fails only for x = -inf; score 0 > T (T<0) is true in the reference.  hc_finalize and the device both
// special-case this through ScoreParams.flags bits below.
static const uint32_t kParamEdgeAlways = 1u;  // edge_threshold < 0
static const uint32_t kParamOvAlways = 2u;    // ov_threshold < 0

extern "C" {

const char* hc_version(void) { return "hcedge 0.1.0 (gfx950)"; }

const char* hc_strerror(int status) {
    switch (status) {
        case HC_OK: return "ok";
        case HC_ERR_ARG: return "invalid argument";
        case HC_ERR_NOMEM: return "out of memory";
        case HC_ERR_HIP: return "HIP runtime error";
        case HC_ERR_NO_DEVICE: return "no HIP device (libhcedge has no CPU fallback)";
        case HC_ERR_STATE: return "call order violated";
        case HC_ERR_BAD_READ: return "malformed read set";
        case HC_ERR_BAD_OVERLAP: return "malformed overlap record";
        case HC_ERR_IO: return "I/O error";
        case HC_ERR_FORMAT: return "input format the reference rejects";
        case HC_ERR_DATA: return "overlap touches a base/quality the reference asserts on";
        case HC_ERR_NOT_ON_DEVICE: return "not decided on the device (the host's route takes the input)";
        default: return "unknown status";
    }
}

const char* hc_last_error(void) { return g_last_error.c_str(); }

int hc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// what of a context follows from the settings alone (hc_create, hc_reset)
static void apply_settings(hc_ctx* c, const hc_settings* settings) {
    c->settings = *settings;
    c->params.edge = make_band(settings->edge_threshold);
    c->params.ov = make_band(settings->ov_threshold);
    c->params.merge_contigs = settings->merge_contigs;
    c->params.min_read_len = settings->min_read_len;
    c->params.flags = (settings->edge_threshold < 0 ? kParamEdgeAlways : 0u) | (settings->ov_threshold < 0 ? kParamOvAlways : 0u);
    c->params.rec_fmt = HC_REC_FULL;
    c->params.pad = 0;
    c->params.n_dev = nullptr;
}

static int create_ctx(hc_ctx* c, const hc_settings* settings) {
    c->settings = *settings;
    c->device = settings->device;
    HC_HIP(hipSetDevice(c->device));
    hipDeviceProp_t prop;
    HC_HIP(hipGetDeviceProperties(&prop, c->device));
    c->n_cu = prop.multiProcessorCount > 0 ? (uint32_t)prop.multiProcessorCount : 256u;
    HC_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HC_HIP(hipEventCreate(&c->ev0));
    HC_HIP(hipEventCreate(&c->ev1));
    HC_HIP(hipMalloc((void**)&c->d_totals, 2 * sizeof(unsigned long long)));
    HC_HIP(hc::set_score_kernel_lds_limit());
    apply_settings(c, settings);
    return HC_OK;
}

int hc_create(hc_ctx** out, const hc_settings* settings) {
    if (!out || !settings) return fail(HC_ERR_ARG, "hc_create: null argument");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(HC_ERR_NO_DEVICE, "hc_create: no HIP device visible");
    if (settings->device < 0 || settings->device >= n) return fail(HC_ERR_ARG, "hc_create: device ordinal out of range");
    hc_ctx* c = new (std::nothrow) hc_ctx();
    if (!c) return fail(HC_ERR_NOMEM, "hc_create: host allocation failed");
    const int rc = create_ctx(c, settings);
    if (rc != HC_OK) {  // nothing of a half-built context stays behind
        const std::string why = g_last_error;
        hc_destroy(c);
        return fail(rc, why);
    }
    *out = c;
    return HC_OK;
}

// scratch_too = false (hc_reset: the resident process, a caller's stage after stage on parked devices): the grow-only scratch of the finder and
// of the SFO ingest — gigabytes at config 3's size — stays with the context for the next read set.  Giving it back and asking for it again
// every stage stalled the next stage's first kernels by 0.3 - 0.4 s on this pool (round 6: profiles/r06_stage_a_from_store.md).
static void free_store(hc_ctx* c, bool scratch_too = true) {
    if (scratch_too) {
        for (auto& sl : c->finder_scratch) {
            if (sl.p) (void)hipFree(sl.p);
            sl.p = nullptr;
            sl.cap = 0;
        }
        for (auto& sl : c->ingest_scratch) {
            if (sl.p) (void)hipFree(sl.p);
            sl.p = nullptr;
            sl.cap = 0;
        }
        if (c->d_found_lines) (void)hipFree(c->d_found_lines);
        c->d_found_lines = nullptr;
        c->found_lines_cap = 0;
        if (c->d_found) (void)hipFree(c->d_found);
        c->d_found = nullptr;
        c->found_cap = 0;
    }
    c->n_found = 0;
    c->found_valid = false;
    if (c->d_sym) (void)hipFree(c->d_sym);
    if (c->d_reads) (void)hipFree(c->d_reads);
    if (c->d_lut) (void)hipFree(c->d_lut);
    if (c->d_inv_n) (void)hipFree(c->d_inv_n);
    c->d_inv_n = nullptr;
    c->d_sym = nullptr;
    c->d_reads = nullptr;
    c->d_lut = nullptr;
    c->have_reads = false;
    c->store_bytes = 0;
}

int hc_destroy(hc_ctx* c) {
    if (!c) return HC_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    free_store(c);
    if (c->d_in) (void)hipFree(c->d_in);
    if (c->d_out) (void)hipFree(c->d_out);
    if (c->d_totals) (void)hipFree(c->d_totals);
    if (c->d_started) (void)hipFree(c->d_started);
    if (c->d_sort) (void)hipFree(c->d_sort);
    if (c->d_sort_tmp) (void)hipFree(c->d_sort_tmp);
    if (c->d_compact_tmp) (void)hipFree(c->d_compact_tmp);
    if (c->d_compact_idx) (void)hipFree(c->d_compact_idx);
    if (c->d_compact_res) (void)hipFree(c->d_compact_res);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->scratch_done) (void)hipEventDestroy(c->scratch_done);
    for (int t = 0; t < 2; t++) {
        if (c->graph.h_stage[t]) (void)hipHostFree(c->graph.h_stage[t]);
        if (c->h_ingest[t]) (void)hipHostFree(c->h_ingest[t]);
        if (c->graph.stage_free[t]) (void)hipEventDestroy(c->graph.stage_free[t]);
    }
    for (void* h : c->h_sfo_text)
        if (h) (void)hipHostFree(h);
    for (hipStream_t s : c->text_copy_stream)
        if (s) (void)hipStreamDestroy(s);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return HC_OK;
}

// The log-probability table.  Built with the HOST libm by the reference's own expressions
// (EdgeCalculator.cpp:41,44,52,60) so that every term the device adds is bit-identical to the
// reference's log(p).  Dimension Kp = K + 2: index K is the N row/column (0.0: the position is
// skipped, :35-39), index K+1 the invalid-symbol row/column (NaN poison).  Layouts: hc_device.h.
// `wide_rows` (the wide 8-bit encoding only, 64 entries): what each quality index means there — a Phred value, -1 = N, -2 = not a value.
static bool build_lut(const std::vector<int>& phred, const std::vector<int>& wide_rows, double mismatch_setting, uint32_t symbytes,
                      std::vector<double>& lut) {
    const size_t K = phred.size(), Kp = K + 2;
    const double inf = std::numeric_limits<double>::infinity();
    const double nan = std::numeric_limits<double>::quiet_NaN();
    const uint32_t lg = hc::lut_lg((uint32_t)K);
    const bool wide = symbytes == 1 && lg >= 6;
    const size_t dim = symbytes == 1 ? (wide ? (size_t)64 : ((size_t)1 << lg)) : Kp;  // rows / columns that can be addressed
    lut.assign(symbytes == 1 ? (size_t)hc::lut_doubles_u8(lg) : (size_t)hc::lut_tri((uint32_t)Kp) * 2, nan);
    bool symmetric = true;  // the 16-bit layout keeps one triangle: every (a, b) must equal (b, a) bit for bit
    for (size_t a = 0; a < dim; a++) {
        for (size_t b = 0; b < dim; b++) {
            // which index means N / invalid depends on the encoding (hc_device.h)
            const bool bad = wide ? (wide_rows[a] == -2 || wide_rows[b] == -2) : (a >= K + 1 || b >= K + 1);
            const bool isn = wide ? (wide_rows[a] == -1 || wide_rows[b] == -1) : (a == K || b == K);
            double vm, vx;
            if (bad) {
                vm = vx = nan;
            } else if (isn) {
                vm = vx = 0.0;
            } else {
                const int ph1 = wide ? wide_rows[a] : phred[a], ph2 = wide ? wide_rows[b] : phred[b];
                const double p1 = pow(10, -ph1 / 10.0);  // phred_to_prob, :59-63
                const double p2 = pow(10, -ph2 / 10.0);
                const double pm = (1 - p1) * (1 - p2) + (p1 * p2) / 3.0;                               // :41
                const double px = p1 * (1 - p2) / 3.0 + p2 * (1 - p1) / 3.0 + (2 / 9.0) * p1 * p2;  // :44
                vm = (pm < mismatch_setting) ? inf : log(pm);  // :49-52
                vx = (px < mismatch_setting) ? inf : log(px);
            }
            if (symbytes == 1) {
                lut[hc::lut_addr_u8(lg, (uint32_t)a, (uint32_t)b, 0) / 8] = vm;
                lut[hc::lut_addr_u8(lg, (uint32_t)a, (uint32_t)b, 1) / 8] = vx;
            } else {
                double& sm = lut[hc::lut_addr_u16((uint32_t)Kp, (uint32_t)a, (uint32_t)b, 0) / 8];
                double& sx = lut[hc::lut_addr_u16((uint32_t)Kp, (uint32_t)a, (uint32_t)b, 1) / 8];
                if (a > b && (memcmp(&sm, &vm, 8) != 0 || memcmp(&sx, &vx, 8) != 0)) symmetric = false;  // (b, a) was stored first
                sm = vm;
                sx = vx;
            }
        }
    }
    return symmetric;
}

int hc_set_reads(hc_ctx* c, const uint8_t* bases, const uint8_t* quals, const uint64_t* seq_off,
                 const uint32_t* read_first_seq, uint32_t n_reads) {
    if (!c || !seq_off || !read_first_seq) return fail(HC_ERR_ARG, "hc_set_reads: null argument");
    HC_HIP(hipSetDevice(c->device));
    const uint32_t n_seq = read_first_seq[n_reads];
    if (read_first_seq[0] != 0) return fail(HC_ERR_BAD_READ, "hc_set_reads: read_first_seq[0] != 0");
    for (uint32_t r = 0; r < n_reads; r++) {
        const uint32_t k = read_first_seq[r + 1] - read_first_seq[r];
        if (k != 1 && k != 2) return fail(HC_ERR_BAD_READ, "hc_set_reads: a read must own 1 or 2 sequences");
    }
    const uint64_t total = seq_off[n_seq];
    if (total > 0 && (!bases || !quals)) return fail(HC_ERR_ARG, "hc_set_reads: null bases/quals");
    if (seq_off[0] != 0) return fail(HC_ERR_BAD_READ, "hc_set_reads: seq_off[0] != 0");
    std::vector<uint32_t> seq_len(n_seq ? n_seq : 1);
    for (uint32_t q = 0; q < n_seq; q++) {
        if (seq_off[q + 1] <= seq_off[q])  // FastqStorage.cpp:143-146,218-221: empty sequence => exit(1)
            return fail(HC_ERR_BAD_READ, "hc_set_reads: empty sequence");
        const uint64_t len = seq_off[q + 1] - seq_off[q];
        if (len >= (1ull << 28)) return fail(HC_ERR_BAD_READ, "hc_set_reads: sequence longer than 2^28-1");
        seq_len[q] = (uint32_t)len;
    }
    // dense quality alphabet over the bytes the reference accepts: Q = byte-33 >= 0 as a signed char
    uint64_t hist[256] = {0}, base_hist[256] = {0};
    {  // byte histograms of the two arrays (300 MB each at C3): a few threads, one partial pair each
        const unsigned T = total < (1u << 22) ? 1u : std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
        std::vector<std::array<uint64_t, 512>> part(T);
        auto work = [&](unsigned t) {
            std::array<uint64_t, 512>& h = part[t];
            h.fill(0);
            const uint64_t a = total * t / T, b = total * (t + 1) / T;
            for (uint64_t i = a; i < b; i++) {
                h[quals[i]]++;
                h[256 + bases[i]]++;
            }
        };
        std::vector<std::thread> th;
        for (unsigned t = 1; t < T; t++) th.emplace_back(work, t);
        work(0);
        for (auto& x : th) x.join();
        for (unsigned t = 0; t < T; t++)
            for (int b = 0; b < 256; b++) {
                hist[b] += part[t][(size_t)b];
                base_hist[b] += part[t][256 + (size_t)b];
            }
    }
    bool any_bad_base = false;  // a base outside ACGTN somewhere (the encoder flags the sequence; the store is then not "regular")
    for (int b = 0; b < 256; b++)
        if (base_hist[b] && b != 'A' && b != 'C' && b != 'G' && b != 'T' && b != 'N') any_bad_base = true;
    std::vector<int> phred, wide_rows;
    uint8_t qmap[256];
    memset(qmap, 255, sizeof qmap);
    {
        std::vector<int> present;
        for (int b = 33; b <= 127; b++)
            if (hist[b]) present.push_back(b);
        const uint32_t Kq = (uint32_t)present.size();
        std::vector<uint32_t> index_of(Kq);
        for (uint32_t k = 0; k < Kq; k++) index_of[k] = k;  // by byte value
        // The smallest table (K + 2 <= 8 indices, hc_device.h: sparse layout): the LDS bank of an entry is (qa & 3, (qa ^ qb) & 7), so entries
        // (qa, qb) and (qa ^ 4, qb ^ 4) share a bank at different addresses — the only structural conflict of that table.  The indices are
        // therefore dealt by FREQUENCY (round 5): the most frequent quality values take the indices whose partner i ^ 4 is the N index, the
        // invalid index or no index at all (entries that are next to never read), the others are paired hottest with coldest.  Any
        // assignment gives the same results (table and symbols are built from the same map); HC_QIDX_ORDER=value keeps byte order (A/B knob).
        const char* qo = getenv("HC_QIDX_ORDER");
        if (Kq >= 5 && Kq <= 6 && !(qo && strcmp(qo, "value") == 0)) {
            std::vector<uint32_t> by_freq(Kq);
            for (uint32_t k = 0; k < Kq; k++) by_freq[k] = k;
            std::stable_sort(by_freq.begin(), by_freq.end(), [&](uint32_t a, uint32_t b) { return hist[present[a]] > hist[present[b]]; });
            std::vector<uint32_t> cold, pair_lo;  // indices with a cold partner; the lower index of an occupied pair (i, i ^ 4)
            for (uint32_t i = 0; i < Kq; i++) {
                if ((i ^ 4u) >= Kq) cold.push_back(i);
                else if (i < (i ^ 4u)) pair_lo.push_back(i);
            }
            uint32_t next = 0, last = Kq;
            for (uint32_t i : cold) index_of[by_freq[next++]] = i;
            for (uint32_t i : pair_lo) {  // hottest remaining with coldest remaining
                index_of[by_freq[next++]] = i;
                index_of[by_freq[--last]] = i ^ 4u;
            }
        }
        phred.assign(Kq, 0);
        if (hc::sym_bytes_for(Kq) == 1 && hc::lut_lg(Kq) >= 6) {
            // the wide 8-bit encodings (hc_device.h): indices 16..63 (up to 48 values) or 4..63 (up to 60), the r-th most frequent value takes
            // kWideRankLabel[r] / kWide7RankLabel[r] — the table's rows are addressed by them, nothing else depends on the assignment
            // (HC_QIDX_ORDER=value: the first index + the value's rank by byte)
            const bool seven = hc::lut_lg(Kq) == 7;
            std::vector<uint32_t> by_freq(Kq);
            for (uint32_t k = 0; k < Kq; k++) by_freq[k] = k;
            if (!(qo && strcmp(qo, "value") == 0))
                std::stable_sort(by_freq.begin(), by_freq.end(), [&](uint32_t a, uint32_t b) { return hist[present[a]] > hist[present[b]]; });
            wide_rows.assign(64, -2);
            wide_rows[hc::kWideN] = -1;
            for (uint32_t r = 0; r < Kq; r++) {
                const uint32_t label = (qo && strcmp(qo, "value") == 0) ? hc::wide_first(seven ? 7u : 6u) + r
                                                                        : (seven ? hc::kWide7RankLabel[r] : hc::kWideRankLabel[r]);
                qmap[present[by_freq[r]]] = (uint8_t)label;
                wide_rows[label] = present[by_freq[r]] - 33;
                phred[by_freq[r]] = present[by_freq[r]] - 33;  // (K entries; the table is built from wide_rows)
            }
        } else {
            for (uint32_t k = 0; k < Kq; k++) {
                qmap[present[k]] = (uint8_t)index_of[k];
                phred[index_of[k]] = present[k] - 33;
            }
        }
    }
    if (phred.empty()) phred.push_back(0);
    const uint32_t K = (uint32_t)phred.size();
    const uint32_t symbytes = hc::sym_bytes_for(K);

    // store layout (hc_device.h): a single read [fwd][rc]; a pair [/1 fwd][/2 fwd][/1 rc][/2 rc]
    std::vector<uint64_t> sym_off(n_seq ? n_seq : 1);
    std::vector<uint32_t> rc_delta(n_seq ? n_seq : 1);
    // slots on 128-byte lines while the store stays inside the Infinity Cache (256 MB), else packed (hc_device.h: slot_stride)
    uint32_t slot_align = 128;
    {
        uint64_t aligned_bytes = 0;
        for (uint32_t q = 0; q < n_seq; q++) aligned_bytes += 2 * hc::slot_stride(seq_len[q], symbytes, 128) * symbytes;
        if (aligned_bytes > (256ull << 20)) slot_align = 16;
        if (const char* v = getenv("HC_SLOT_ALIGN")) {  // tuning knob: 16, 32, 64, 128, 256
            const int a = atoi(v);
            if (a >= 16 && a <= 4096 && (a & (a - 1)) == 0) slot_align = (uint32_t)a;
        }
    }
    uint64_t nsym = 0;
    for (uint32_t r = 0; r < n_reads; r++) {
        const uint32_t q = read_first_seq[r];
        const uint64_t s1 = hc::slot_stride(seq_len[q], symbytes, slot_align);
        if (read_first_seq[r + 1] - q == 2) {
            const uint64_t s2 = hc::slot_stride(seq_len[q + 1], symbytes, slot_align);
            sym_off[q] = nsym;
            sym_off[q + 1] = nsym + s1;
            rc_delta[q] = rc_delta[q + 1] = (uint32_t)(s1 + s2);
            nsym += 2 * (s1 + s2);
        } else {
            sym_off[q] = nsym;
            rc_delta[q] = (uint32_t)s1;
            nsym += 2 * s1;
        }
    }
    std::vector<double> lut;
    if (!build_lut(phred, wide_rows, c->settings.mismatch, symbytes, lut))  // (never seen: the reference's expressions commute for every Phred pair)
        return fail(HC_ERR_STATE, "hc_set_reads: the log-probability table is not symmetric in its two qualities (--mismatch within one ulp of a term?)");

    // (the grow-only scratch of the finder and of the SFO ingest stays: it is capacity, not state.  A context that takes read set after read
    // set — a pipeline's stages on parked devices — gave gigabytes back here and asked for them again in the next hc_find_overlaps, and one
    // hipMalloc in three then took 0.75 - 0.9 s on this pool: tools/experiments/r06_find_stall_trace.sh, round 6.  hc_destroy returns it.)
    free_store(c, false);
    struct Tmp {  // freed on every return path
        void* p = nullptr;
        ~Tmp() {
            if (p) (void)hipFree(p);
        }
    } t_bases, t_quals, t_qmap, t_seq_bad, t_raw_off, t_seq_off, t_first, t_rc_delta;
    const uint64_t sym_bytes_total = (nsym ? nsym : 1) * symbytes;
    HC_HIP(hipMalloc(&c->d_sym, sym_bytes_total));
    HC_HIP(hipMalloc((void**)&c->d_reads, sizeof(hc::ReadDesc) * (n_reads ? n_reads : 1)));
    HC_HIP(hipMalloc((void**)&c->d_lut, sizeof(double) * lut.size()));
    HC_HIP(hipMalloc(&t_seq_off.p, sizeof(uint64_t) * (n_seq ? n_seq : 1)));
    HC_HIP(hipMalloc(&t_seq_bad.p, (n_seq ? n_seq : 1)));
    HC_HIP(hipMalloc(&t_first.p, sizeof(uint32_t) * (n_reads + 1)));
    HC_HIP(hipMalloc(&t_bases.p, total ? total : 1));
    HC_HIP(hipMalloc(&t_quals.p, total ? total : 1));
    HC_HIP(hipMalloc(&t_raw_off.p, sizeof(uint64_t) * (n_seq + 1)));
    HC_HIP(hipMalloc(&t_qmap.p, 256));
    HC_HIP(hipMalloc(&t_rc_delta.p, sizeof(uint32_t) * (n_seq ? n_seq : 1)));
    uint8_t *d_bases = (uint8_t*)t_bases.p, *d_quals = (uint8_t*)t_quals.p, *d_qmap = (uint8_t*)t_qmap.p, *d_seq_bad = (uint8_t*)t_seq_bad.p;
    uint64_t *d_raw_off = (uint64_t*)t_raw_off.p, *d_seq_off = (uint64_t*)t_seq_off.p;
    uint32_t* d_first = (uint32_t*)t_first.p;
    if (total) {
        HC_HIP(hipMemcpyAsync(d_bases, bases, total, hipMemcpyHostToDevice, c->stream));
        HC_HIP(hipMemcpyAsync(d_quals, quals, total, hipMemcpyHostToDevice, c->stream));
    }
    HC_HIP(hipMemcpyAsync(d_raw_off, seq_off, sizeof(uint64_t) * (n_seq + 1), hipMemcpyHostToDevice, c->stream));
    HC_HIP(hipMemcpyAsync(d_qmap, qmap, 256, hipMemcpyHostToDevice, c->stream));
    HC_HIP(hipMemcpyAsync(d_seq_off, sym_off.data(), sizeof(uint64_t) * n_seq, hipMemcpyHostToDevice, c->stream));
    HC_HIP(hipMemcpyAsync(t_rc_delta.p, rc_delta.data(), sizeof(uint32_t) * n_seq, hipMemcpyHostToDevice, c->stream));
    HC_HIP(hipMemcpyAsync(d_first, read_first_seq, sizeof(uint32_t) * (n_reads + 1), hipMemcpyHostToDevice, c->stream));
    HC_HIP(hipMemcpyAsync(c->d_lut, lut.data(), sizeof(double) * lut.size(), hipMemcpyHostToDevice, c->stream));
    // 1.0 / n for every count a sub-overlap can reach (n <= positions rounded up to 16 <= the longest sequence + 15): the reference's
    // `1.0/total_len` (:137) as the host's IEEE division — the same quotient the device's gives — read by the kernel instead of divided
    uint32_t longest = 0;
    for (uint32_t q = 0; q < n_seq; q++) longest = std::max(longest, seq_len[q]);
    std::vector<double> inv_n((size_t)longest + 17);
    inv_n[0] = std::numeric_limits<double>::infinity();
    for (size_t k = 1; k < inv_n.size(); k++) inv_n[k] = 1.0 / (double)k;
    HC_HIP(hipMalloc((void**)&c->d_inv_n, sizeof(double) * inv_n.size()));
    HC_HIP(hipMemcpyAsync(c->d_inv_n, inv_n.data(), sizeof(double) * inv_n.size(), hipMemcpyHostToDevice, c->stream));
    HC_HIP(hc::launch_encode(symbytes, d_bases, d_quals, d_raw_off, d_seq_off, (const uint32_t*)t_rc_delta.p, d_qmap, n_seq, K, c->d_sym, d_seq_bad,
                             d_first, n_reads, c->d_reads, slot_align, c->stream));
    HC_HIP(hipStreamSynchronize(c->stream));

    {  // SFO ids: singles, then every /1 mate, then every /2 mate (s_p1_p2.fasta, savage.py:643-664)
        uint32_t n_single = 0, n_pairs = 0;
        c->singles_first = true;
        for (uint32_t r = 0; r < n_reads; r++) {
            const bool paired = read_first_seq[r + 1] - read_first_seq[r] == 2;
            if (paired) n_pairs++;
            else {
                if (n_pairs) c->singles_first = false;
                n_single++;
            }
        }
        c->seq_refs.assign(n_seq, hc::SeqRef{0, 0, 0, 0, 0});
        uint32_t pair_no = 0;
        for (uint32_t r = 0; r < n_reads; r++) {
            const uint32_t q = read_first_seq[r];
            if (read_first_seq[r + 1] - q == 2) {
                c->seq_refs[q] = hc::SeqRef{sym_off[q], seq_len[q], n_single + pair_no, rc_delta[q], 0};
                c->seq_refs[q + 1] = hc::SeqRef{sym_off[q + 1], seq_len[q + 1], n_single + n_pairs + pair_no, rc_delta[q + 1], 0};
                pair_no++;
            } else {
                c->seq_refs[q] = hc::SeqRef{sym_off[q], seq_len[q], r, rc_delta[q], 0};
            }
        }
    }
    c->view.sym = c->d_sym;
    c->view.reads = c->d_reads;
    c->view.n_reads = n_reads;
    c->view.n_seq = n_seq;
    c->view.K = K;
    c->view.symbytes = symbytes;
    c->view.lut_bytes = (uint32_t)(lut.size() * sizeof(double));
    c->view.inv_n = c->d_inv_n;
    c->view.inv_len = (uint32_t)inv_n.size();
    c->store_bytes = sym_bytes_total;
    c->view.store_bytes = sym_bytes_total;
    {  // regular store (hc_device.h): descriptors by arithmetic
        bool same_len = n_seq > 0;
        for (uint32_t q = 1; q < n_seq && same_len; q++) same_len = seq_len[q] == seq_len[0];
        uint32_t n_single = 0;
        for (uint32_t r = 0; r < n_reads; r++) n_single += read_first_seq[r + 1] - read_first_seq[r] == 1;
        c->view.regular = (same_len && c->singles_first && !any_bad_base) ? 1u : 0u;
        if (const char* v = getenv("HC_REGULAR_STORE")) c->view.regular = c->view.regular && atoi(v) != 0;  // test / tuning knob: 0 forces look-ups
        c->view.ulen = same_len ? seq_len[0] : 0u;
        c->view.n_single = n_single;
        c->view.seq_syms = same_len ? (uint32_t)(2 * hc::slot_stride(seq_len[0], symbytes, slot_align)) : 0u;
    }
    c->have_reads = true;
    {
        // How mixed the lengths are, by the 5th and 95th percentile (round 5; until then by the shortest and the longest sequence: ONE short
        // read — routine after quality trimming — flipped the kernel of an otherwise uniform set)
        uint32_t lmin = 0, lmax = 0;
        if (n_seq) {
            std::vector<uint32_t> sorted_len(seq_len.begin(), seq_len.begin() + n_seq);
            const size_t k5 = (size_t)n_seq * 5 / 100, k95 = std::min<size_t>(n_seq - 1, (size_t)n_seq * 95 / 100);
            std::nth_element(sorted_len.begin(), sorted_len.begin() + k5, sorted_len.end());
            lmin = sorted_len[k5];
            std::nth_element(sorted_len.begin(), sorted_len.begin() + k95, sorted_len.end());
            lmax = sorted_len[k95];
        }
        c->len_p5 = lmin;
        c->len_p95 = lmax;
        // mixed-length read set (contigs + reads) of sequences that are not short: the launches bucket their candidates by length.
        // Mixed but short sequences keep the plain launch, whose waves deal their sub-overlaps by length themselves: the bucketing's
        // gathered records and scattered results cost more than the idle lanes there.  2 * 10^6 s-s overlaps, plain / bucketed:
        // reads of 100..400 bp (mean 216) 0.196 / 0.234 ms, 120..900 bp (mean 387) 0.307 / 0.289, 150..1 500 bp (mean 586)
        // 0.441 / 0.348, 150..6 000 bp (mean 1 586) 1.35 / 0.63 (profiles/r03_bucket_dispatch.txt)
        c->view.balance = (n_seq && lmax > 2u * lmin && total / n_seq > 350) ? 1u : 0u;
        if (const char* b = getenv("HC_BALANCE")) c->view.balance = atoi(b) != 0;
    }
    c->coop_fetch = true;
    if (const char* v = getenv("HC_FETCH_GROUP")) {
        c->coop_fetch = !strcmp(v, "coop");
        c->fetch_group = atoi(v) == 2 ? 2 : 4;
    } else {
        // Which kernel a read set takes, by its shape (2 * 10^6 candidates each, kernel ms; profiles/r04_dispatch.txt, r04_dispatch_pairs.txt):
        //   pairs 2 x 150 (C2)                        cooperative 0.186        per lane 0.25            -> cooperative (LDS-DMA rows)
        //   pairs trimmed to 60..150 per mate         cooperative 0.098        per lane 0.097           -> cooperative
        //   singles 400..500 (the SAVAGE example)     cooperative 0.265        per lane 0.304           -> cooperative
        //   singles 250, 35 quality values (C4)       cooperative 0.147        per lane 0.139 - 0.147   -> cooperative (within the noise)
        //   singles 100..400, mean 216                cooperative 0.186        per lane (G = 2) 0.169   -> PER LANE: a wave's step pays the
        //                                                                        row exchange for its longest lane while most lanes are done
        //   singles 120..900, mean 387                bucketed 0.288           per lane 0.284 - 0.287   -> bucketed cooperative (level)
        //   singles 150..1 500, mean 586              bucketed 0.347           per lane 0.388           -> bucketed cooperative
        // 64-symbol fetch groups for short-read sets, 32-symbol groups when the sequences are long (contigs) or the symbols 16 bits wide:
        // the per-lane kernel's, for the sets that take it (below) and for stores of 4 GiB and more
        const uint64_t mean_len = n_seq ? total / n_seq : 0;
        c->fetch_group = (mean_len > 600 || symbytes == 2) ? 2 : 4;
        const uint32_t lmin = c->len_p5, lmax = c->len_p95;  // 5th / 95th percentile of the sequence lengths (above)
        uint32_t n_pairs = 0;
        for (uint32_t r = 0; r < n_reads; r++) n_pairs += read_first_seq[r + 1] - read_first_seq[r] == 2;
        const bool short_mixed_singles = n_seq && n_pairs == 0 && lmax > 2u * lmin && mean_len <= 300 && !c->view.balance;  // (HC_BALANCE=1 forces the bucketed launch)
        c->coop_fetch = !short_mixed_singles;
        if (short_mixed_singles) c->fetch_group = 2;
    }
    c->view.long_rows = (n_seq && total / n_seq > 600) ? 1u : 0u;
    return HC_OK;
}

// Sampled locality probe on a host copy of a batch: in overlap files as sfo2overlaps / FNO write them,
// consecutive lines share a read almost always; if fewer than half of the sampled neighbours do, the
// batch is worth reordering on the device.
static bool host_batch_is_ordered(const void* in, size_t rec_bytes, uint64_t n) {
    if (n < 2) return true;
    const uint64_t samples = n - 1 < 4096 ? n - 1 : 4096;
    const uint64_t step = (n - 2) / samples > 0 ? (n - 2) / samples : 1;  // k * step + 1 <= n - 1
    uint64_t share = 0;
    for (uint64_t k = 0; k < samples; k++) {
        const uint32_t* a = (const uint32_t*)((const char*)in + k * step * rec_bytes);  // both formats start with read1, read2
        const uint32_t* b = (const uint32_t*)((const char*)in + (k * step + 1) * rec_bytes);
        share += (a[0] == b[0]) | (a[0] == b[1]) | (a[1] == b[0]) | (a[1] == b[1]);
    }
    return share * 2 >= samples;
}

static int ensure_sort_workspace(hc_ctx* c, uint64_t n) {
    if (n <= c->sort_cap) return HC_OK;
    if (c->d_sort) (void)hipFree(c->d_sort);
    if (c->d_sort_tmp) (void)hipFree(c->d_sort_tmp);
    c->d_sort = nullptr;
    c->d_sort_tmp = nullptr;
    c->sort_cap = 0;
    c->sort_tmp_bytes = hc::reorder_temp_bytes((uint32_t)n);
    HC_HIP(hipMalloc((void**)&c->d_sort, 4 * n * sizeof(uint32_t)));
    HC_HIP(hipMalloc(&c->d_sort_tmp, c->sort_tmp_bytes ? c->sort_tmp_bytes : 16));
    c->sort_cap = n;
    return HC_OK;
}

}  // extern "C"

int hc_ctx_score(hc_ctx* c, uint32_t fmt, const void* d_in, uint64_t n, void* d_out, hipStream_t s, bool reorder,
                 hc_gather_row* rows, unsigned long long* row_count, uint64_t cap, uint64_t base_index, const unsigned long long* n_dev,
                 const hc_line_rec* lines_in, hc_line_rec* lines_out, hc_bucket_ws* bucket) {
    // Which of the context's own scratch this launch will use (blocks bring their own and take none of it)
    const bool want_perm = reorder && n > 1 && n < (1ull << 31);
    const bool want_bucket = c->view.balance && c->coop_fetch && n < (1ull << 32);
    const bool want_segments = rows && !lines_in && c->coop_fetch;
    const bool ctx_scratch = want_perm || (want_bucket && !bucket) || want_segments;
    if (ctx_scratch) {
        // one launch at a time on that scratch: a launch on another stream than the last one waits for it on the device
        if (!c->scratch_done) HC_HIP(hipEventCreateWithFlags(&c->scratch_done, hipEventDisableTiming));
        if (c->scratch_used && c->scratch_stream != s) HC_HIP(hipStreamWaitEvent(s, c->scratch_done, 0));
    }
    const uint32_t* perm = nullptr;
    if (want_perm) {
        int rc = ensure_sort_workspace(c, n);
        if (rc) return rc;
        uint32_t* keys_in = c->d_sort;
        uint32_t* keys_out = c->d_sort + c->sort_cap;
        uint32_t* idx_in = c->d_sort + 2 * c->sort_cap;
        uint32_t* perm_out = c->d_sort + 3 * c->sort_cap;
        HC_HIP(hc::launch_reorder(c->view.n_reads, fmt, d_in, (uint32_t)n, keys_in, keys_out, idx_in, perm_out, c->d_sort_tmp,
                                  c->sort_tmp_bytes, s));
        perm = perm_out;
    }
    hc::ScoreParams prm = c->params;
    prm.rec_fmt = fmt;
    prm.pad = 0;
    prm.n_dev = n_dev;
    uint32_t *bperm = nullptr, *bqueue = nullptr;
    if (want_bucket) {  // mixed sequence lengths: the launch buckets its candidates by length first
        hc_bucket_ws* ws = bucket ? bucket : &c->bucket;
        int rc = ws->ensure(n);
        if (rc) return rc;
        bperm = ws->perm();
        bqueue = ws->queue();
    }
    hc_gather_row* seg_buf = nullptr;
    uint32_t* seg_count = nullptr;
    uint64_t seg_total = 0;
    if (want_segments) {  // the cooperative launches collect their rows in per-workgroup segments, spilling into `cap` rows behind them
        seg_total = 2 * cap + hc::kSinkMaxGroups * 512;  // the expected share per workgroup and room for the small ones, then the spill area
        // (a grown buffer is a new one: the launches in flight on the old one are behind scratch_done, which this stream has waited for
        // or is itself ordered behind — but the runtime frees at once, so wait for them on the host before letting go of it)
        if (seg_total * sizeof(hc_gather_row) > c->sink_rows.cap && c->scratch_used) HC_HIP(hipEventSynchronize(c->scratch_done));
        int rc = c->sink_rows.ensure(seg_total * sizeof(hc_gather_row));
        if (rc) return rc;
        if (!c->sink_counts.p) {
            if ((rc = c->sink_counts.ensure((hc::kSinkMaxGroups + 2) * sizeof(uint32_t))) != HC_OK) return rc;
            HC_HIP(hipMemsetAsync(c->sink_counts.p, 0, (hc::kSinkMaxGroups + 2) * sizeof(uint32_t), s));  // the spill counters start at zero
        }
        seg_buf = c->sink_rows.as<hc_gather_row>();
        seg_count = c->sink_counts.as<uint32_t>();
    }
    if (want_segments && c->sink_dirty) {  // a segmented launch failed at enqueue: whatever it left in the spill counters goes
        HC_HIP(hipMemsetAsync(c->sink_counts.as<uint32_t>() + hc::kSinkMaxGroups, 0, 2 * sizeof(uint32_t), s));
        c->sink_dirty = false;
    }
    // the multi-GPU step: CUs left to the collective library (hc_set_comm_reserve), workgroup starts counted for hc_comm_gate_device
    const uint32_t cus = c->comm_reserve && c->comm_reserve < c->n_cu ? c->n_cu - c->comm_reserve : c->n_cu;
    unsigned long long* started = (rows && !lines_in) ? c->d_started : nullptr;
    uint32_t started_groups = 0;
    const hipError_t le = hc::launch_score(c->view, prm, c->d_lut, d_in, n, (hc_result_rec*)d_out, perm, cus, c->coop_fetch ? 0 : c->fetch_group, c->fetch_group, rows,
                                           row_count, cap, base_index, s, lines_in, lines_out, bperm, bqueue, seg_buf, seg_count, seg_total, &c->sink_turn, started,
                                           &started_groups);
    if (le != hipSuccess) {
        if (want_segments) c->sink_dirty = true;
        return hc::set_last_error(HC_ERR_HIP, std::string("launch_score: ") + hipGetErrorString(le));
    }
    if (started) c->started_target += started_groups;
    if (ctx_scratch) {
        HC_HIP(hipEventRecord(c->scratch_done, s));
        c->scratch_stream = s;
        c->scratch_used = true;
    }
    return HC_OK;
}

extern "C" {

int hc_set_reorder(hc_ctx* c, int mode) {
    if (!c || mode < HC_REORDER_NEVER || mode > HC_REORDER_AUTO) return fail(HC_ERR_ARG, "hc_set_reorder: bad argument");
    c->reorder_mode = mode;
    return HC_OK;
}

int hc_score_batch_device(hc_ctx* c, const void* d_in, uint64_t n, void* d_out, void* hip_stream) {
    if (!c) return fail(HC_ERR_ARG, "hc_score_batch_device: null context");
    if (!c->have_reads) return fail(HC_ERR_STATE, "hc_score_batch_device: hc_set_reads has not been called");
    if (n == 0) return HC_OK;
    if (!d_in || !d_out) return fail(HC_ERR_ARG, "hc_score_batch_device: null buffer");
    HC_HIP(hipSetDevice(c->device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
    // device-resident records cannot be inspected without a synchronisation: AUTO means "as given"
    return hc_ctx_score(c, HC_REC_FULL, d_in, n, d_out, s, c->reorder_mode == HC_REORDER_ALWAYS, nullptr, nullptr, 0, 0);
}

int hc_score_cands_device(hc_ctx* c, const void* d_in, uint64_t n, void* d_out, void* hip_stream) {
    if (!c) return fail(HC_ERR_ARG, "hc_score_cands_device: null context");
    if (!c->have_reads) return fail(HC_ERR_STATE, "hc_score_cands_device: hc_set_reads has not been called");
    if (n == 0) return HC_OK;
    if (!d_in || !d_out) return fail(HC_ERR_ARG, "hc_score_cands_device: null buffer");
    HC_HIP(hipSetDevice(c->device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
    return hc_ctx_score(c, HC_REC_COMPACT, d_in, n, d_out, s, c->reorder_mode == HC_REORDER_ALWAYS, nullptr, nullptr, 0, 0);
}

void hc_pack_cands(const hc_overlap_rec* in, uint64_t n, hc_cand_rec* out) {
    for (uint64_t i = 0; i < n; i++) {
        const hc_overlap_rec& r = in[i];
        const uint32_t p1 = r.pos1 < HC_CAND_POS_MASK ? r.pos1 : HC_CAND_POS_MASK;
        const uint32_t p2 = r.pos2 < HC_CAND_POS_MASK ? r.pos2 : HC_CAND_POS_MASK;
        const uint32_t oc = r.ord == '-' ? 0u : (r.ord == '1' ? 1u : (r.ord == '2' ? 2u : 3u));
        out[i].read1 = r.read1;
        out[i].read2 = r.read2;
        out[i].pos1_bits = p1 | (r.ori1 ? 1u << 28 : 0u) | (r.ori2 ? 1u << 29 : 0u) | (oc << 30);
        out[i].pos2_bits = p2;
    }
}

int hc_synchronize(hc_ctx* c) {
    if (!c) return fail(HC_ERR_ARG, "hc_synchronize: null context");
    HC_HIP(hipSetDevice(c->device));
    HC_HIP(hipStreamSynchronize(c->stream));
    return HC_OK;
}

static int ensure_workspace(hc_ctx* c, uint64_t n) {
    if (n <= c->ws_cap) return HC_OK;
    if (c->d_in) (void)hipFree(c->d_in);
    if (c->d_out) (void)hipFree(c->d_out);
    c->d_in = c->d_out = nullptr;
    c->ws_cap = 0;
    HC_HIP(hipMalloc(&c->d_in, n * sizeof(hc_overlap_rec)));
    HC_HIP(hipMalloc(&c->d_out, n * sizeof(hc_result_rec)));
    c->ws_cap = n;
    return HC_OK;
}

static int score_host(hc_ctx* c, const char* who, uint32_t fmt, const void* in, uint64_t n, hc_result_rec* out) {
    if (!c) return fail(HC_ERR_ARG, std::string(who) + ": null context");
    if (!c->have_reads) return fail(HC_ERR_STATE, std::string(who) + ": hc_set_reads has not been called");
    if (n == 0) return HC_OK;
    if (!in || !out) return fail(HC_ERR_ARG, std::string(who) + ": null buffer");
    HC_HIP(hipSetDevice(c->device));
    int rc = ensure_workspace(c, n);
    if (rc) return rc;
    const size_t rb = fmt == HC_REC_COMPACT ? sizeof(hc_cand_rec) : sizeof(hc_overlap_rec);
    HC_HIP(hipMemcpyAsync(c->d_in, in, n * rb, hipMemcpyHostToDevice, c->stream));
    bool reorder = c->reorder_mode == HC_REORDER_ALWAYS;
    if (c->reorder_mode == HC_REORDER_AUTO && n >= 4096) reorder = !host_batch_is_ordered(in, rb, n);
    rc = hc_ctx_score(c, fmt, c->d_in, n, c->d_out, c->stream, reorder, nullptr, nullptr, 0, 0);
    if (rc) return rc;
    HC_HIP(hipMemcpyAsync(out, c->d_out, n * sizeof(hc_result_rec), hipMemcpyDeviceToHost, c->stream));
    HC_HIP(hipStreamSynchronize(c->stream));
    return HC_OK;
}

int hc_score_batch(hc_ctx* c, const hc_overlap_rec* in, uint64_t n, hc_result_rec* out) {
    return score_host(c, "hc_score_batch", HC_REC_FULL, in, n, out);
}

int hc_score_cands(hc_ctx* c, const hc_cand_rec* in, uint64_t n, hc_result_rec* out) {
    return score_host(c, "hc_score_cands", HC_REC_COMPACT, in, n, out);
}

int hc_host_alloc(hc_ctx* c, void** ptr, uint64_t bytes) {
    if (!c || !ptr) return fail(HC_ERR_ARG, "hc_host_alloc: null argument");
    *ptr = nullptr;
    HC_HIP(hipSetDevice(c->device));
    HC_HIP(hipHostMalloc(ptr, bytes ? bytes : 1, hipHostMallocDefault));
    return HC_OK;
}

int hc_host_free(hc_ctx* c, void* ptr) {
    if (!c) return fail(HC_ERR_ARG, "hc_host_free: null context");
    if (ptr) HC_HIP(hipHostFree(ptr));
    return HC_OK;
}

static int ensure_compact_workspace(hc_ctx* c, uint64_t n, bool with_buffers) {
    const size_t need = hc::compact_temp_bytes((uint32_t)n);
    if (need > c->compact_tmp_bytes) {
        if (c->d_compact_tmp) (void)hipFree(c->d_compact_tmp);
        c->d_compact_tmp = nullptr;
        c->compact_tmp_bytes = 0;
        HC_HIP(hipMalloc(&c->d_compact_tmp, need));
        c->compact_tmp_bytes = need;
    }
    if (with_buffers && n > c->compact_cap) {
        if (c->d_compact_idx) (void)hipFree(c->d_compact_idx);
        if (c->d_compact_res) (void)hipFree(c->d_compact_res);
        c->d_compact_idx = nullptr;
        c->d_compact_res = nullptr;
        c->compact_cap = 0;
        HC_HIP(hipMalloc((void**)&c->d_compact_idx, n * sizeof(uint32_t)));
        HC_HIP(hipMalloc((void**)&c->d_compact_res, n * sizeof(hc_result_rec)));
        c->compact_cap = n;
    }
    return HC_OK;
}

int hc_reset(hc_ctx* c, const hc_settings* settings) {
    if (!c || !settings) return fail(HC_ERR_ARG, "hc_reset: null argument");
    if (settings->device != c->device) return fail(HC_ERR_ARG, "hc_reset: a context stays on its device");
    HC_HIP(hipSetDevice(c->device));
    HC_HIP(hipDeviceSynchronize());  // nothing of the previous stage is in flight
    free_store(c, false);            // the read store, the finder's results, the id table's validity: as after hc_create (the scratch stays)
    c->have_ids = false;
    c->reorder_mode = HC_REORDER_AUTO;
    c->graph.valid = false;
    c->graph.n_appended = 0;
    apply_settings(c, settings);
    return HC_OK;
}

int hc_set_comm_reserve(hc_ctx* c, uint32_t cus) {
    if (!c) return fail(HC_ERR_ARG, "hc_set_comm_reserve: null context");
    if (cus >= c->n_cu) return fail(HC_ERR_ARG, "hc_set_comm_reserve: more CUs than the device has");
    HC_HIP(hipSetDevice(c->device));
    if (cus && !c->d_started) {
        HC_HIP(hipMalloc((void**)&c->d_started, sizeof(unsigned long long)));
        HC_HIP(hipMemset(c->d_started, 0, sizeof(unsigned long long)));
        c->started_target = 0;
    }
    c->comm_reserve = cus;
    return HC_OK;
}

int hc_comm_gate_device(hc_ctx* c, void* hip_stream, uint32_t timeout_us) {
    if (!c) return fail(HC_ERR_ARG, "hc_comm_gate_device: null context");
    if (!c->d_started) return HC_OK;  // nothing is counted: nothing to wait for
    HC_HIP(hipSetDevice(c->device));
    HC_HIP(hc::launch_comm_gate(c->d_started, c->started_target, timeout_us ? timeout_us : 2000u, hip_stream ? (hipStream_t)hip_stream : c->stream));
    return HC_OK;
}

int hc_compact_device(hc_ctx* c, const void* d_results, uint64_t n, void* d_indices, void* d_count, void* hip_stream) {
    if (!c || !d_count) return fail(HC_ERR_ARG, "hc_compact_device: null argument");
    if (n >= (1ull << 31)) return fail(HC_ERR_ARG, "hc_compact_device: n must be < 2^31");
    HC_HIP(hipSetDevice(c->device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
    if (n == 0) {
        HC_HIP(hipMemsetAsync(d_count, 0, sizeof(unsigned long long), s));
        return HC_OK;
    }
    if (!d_results || !d_indices) return fail(HC_ERR_ARG, "hc_compact_device: null buffer");
    int rc = ensure_compact_workspace(c, n, false);
    if (rc) return rc;
    HC_HIP(hc::launch_compact((const hc_result_rec*)d_results, (uint32_t)n, (uint32_t*)d_indices, (unsigned long long*)d_count,
                              c->d_compact_tmp, c->compact_tmp_bytes, s));
    return HC_OK;
}

int hc_pack_rows_device(hc_ctx* c, const void* d_results, const void* d_indices, const void* d_count, uint64_t cap, uint64_t base_index,
                        void* d_rows, void* hip_stream) {
    if (!c || !d_count) return fail(HC_ERR_ARG, "hc_pack_rows_device: null argument");
    if (cap == 0) return HC_OK;
    if (!d_results || !d_indices || !d_rows) return fail(HC_ERR_ARG, "hc_pack_rows_device: null buffer");
    HC_HIP(hipSetDevice(c->device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
    HC_HIP(hc::launch_pack_rows((const hc_result_rec*)d_results, (const uint32_t*)d_indices, (const unsigned long long*)d_count, cap,
                                base_index, (hc_gather_row*)d_rows, c->n_cu, s));
    return HC_OK;
}

int hc_compact_pack_device(hc_ctx* c, const void* d_results, uint64_t n, void* d_indices, void* d_count, uint64_t cap, uint64_t base_index,
                           void* d_payload, void* hip_stream) {
    if (!c || !d_payload) return fail(HC_ERR_ARG, "hc_compact_pack_device: null argument");
    int rc = hc_compact_device(c, d_results, n, d_indices, d_count, hip_stream);
    if (rc) return rc;
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
    HC_HIP(hc::launch_pack_header((const unsigned long long*)d_count, (hc_gather_row*)d_payload, s));
    if (n == 0 || cap == 0) return HC_OK;
    return hc_pack_rows_device(c, d_results, d_indices, d_count, cap, base_index, (hc_gather_row*)d_payload + 1, hip_stream);
}

int hc_narrow_payload_device(hc_ctx* c, const void* d_payload, uint64_t cap, void* d_payload24, void* hip_stream) {
    if (!c || !d_payload || !d_payload24) return fail(HC_ERR_ARG, "hc_narrow_payload_device: null argument");
    HC_HIP(hipSetDevice(c->device));
    HC_HIP(hc::launch_narrow_payload(d_payload, cap, d_payload24, c->n_cu, (hipStream_t)hip_stream));
    return HC_OK;
}

int hc_score_pack_device(hc_ctx* c, uint32_t fmt, const void* d_in, uint64_t n, void* d_out, uint64_t cap, uint64_t base_index,
                         void* d_payload, void* hip_stream) {
    if (!c || !d_payload) return fail(HC_ERR_ARG, "hc_score_pack_device: null argument");
    if (fmt != HC_REC_FULL && fmt != HC_REC_COMPACT) return fail(HC_ERR_ARG, "hc_score_pack_device: unknown record format");
    if (!c->have_reads) return fail(HC_ERR_STATE, "hc_score_pack_device: hc_set_reads has not been called");
    if (n && (!d_in || !d_out)) return fail(HC_ERR_ARG, "hc_score_pack_device: null buffer");
    if (n >= (1ull << 31)) return fail(HC_ERR_ARG, "hc_score_pack_device: n must be < 2^31");
    HC_HIP(hipSetDevice(c->device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
    hc_gather_row* payload = (hc_gather_row*)d_payload;
    // row 0, the header { count, 0, 0, 0 }: written by the launch's own compaction when its rows collect in per-workgroup segments (every
    // cooperative launch: round 6 — one launch fewer in front of the scoring kernel per step), zeroed by launch_score in front of the other forms
    if (n == 0) HC_HIP(hipMemsetAsync(payload, 0, sizeof(hc_gather_row), s));
    return hc_ctx_score(c, fmt, d_in, n, d_out, s, c->reorder_mode == HC_REORDER_ALWAYS, payload + 1,
                        (unsigned long long*)&payload[0].index, cap, base_index);
}

int hc_score_batch_compact(hc_ctx* c, const hc_overlap_rec* in, uint64_t n, uint32_t* idx_out, hc_result_rec* res_out,
                           uint64_t cap, uint64_t* n_out) {
    if (!c || !n_out) return fail(HC_ERR_ARG, "hc_score_batch_compact: null argument");
    *n_out = 0;
    if (!c->have_reads) return fail(HC_ERR_STATE, "hc_score_batch_compact: hc_set_reads has not been called");
    if (n == 0) return HC_OK;
    if (!in || (cap && (!idx_out || !res_out))) return fail(HC_ERR_ARG, "hc_score_batch_compact: null buffer");
    if (n >= (1ull << 31)) return fail(HC_ERR_ARG, "hc_score_batch_compact: n must be < 2^31");
    HC_HIP(hipSetDevice(c->device));
    int rc = ensure_workspace(c, n);
    if (rc) return rc;
    rc = ensure_compact_workspace(c, n, true);
    if (rc) return rc;
    HC_HIP(hipMemcpyAsync(c->d_in, in, n * sizeof(hc_overlap_rec), hipMemcpyHostToDevice, c->stream));
    bool reorder = c->reorder_mode == HC_REORDER_ALWAYS;
    if (c->reorder_mode == HC_REORDER_AUTO && n >= 4096) reorder = !host_batch_is_ordered(in, sizeof(hc_overlap_rec), n);
    rc = hc_ctx_score(c, HC_REC_FULL, c->d_in, n, c->d_out, c->stream, reorder, nullptr, nullptr, 0, 0);
    if (rc) return rc;
    HC_HIP(hc::launch_compact((const hc_result_rec*)c->d_out, (uint32_t)n, c->d_compact_idx, c->d_totals, c->d_compact_tmp,
                              c->compact_tmp_bytes, c->stream));
    HC_HIP(hc::launch_gather_results((const hc_result_rec*)c->d_out, c->d_compact_idx, c->d_totals, c->d_compact_res, c->n_cu,
                                     c->stream));
    unsigned long long k = 0;
    HC_HIP(hipMemcpyAsync(&k, c->d_totals, sizeof k, hipMemcpyDeviceToHost, c->stream));
    HC_HIP(hipStreamSynchronize(c->stream));
    *n_out = k;
    if (k > cap) return fail(HC_ERR_ARG, "hc_score_batch_compact: output capacity too small");
    if (k) {
        HC_HIP(hipMemcpyAsync(idx_out, c->d_compact_idx, k * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HC_HIP(hipMemcpyAsync(res_out, c->d_compact_res, k * sizeof(hc_result_rec), hipMemcpyDeviceToHost, c->stream));
        HC_HIP(hipStreamSynchronize(c->stream));
    }
    return HC_OK;
}

int hc_time_score_kernel(hc_ctx* c, uint32_t fmt, const void* d_in, uint64_t n, void* d_out, int iters, float* ms_per_launch) {
    if (!c || !ms_per_launch || iters <= 0 || (fmt != HC_REC_FULL && fmt != HC_REC_COMPACT))
        return fail(HC_ERR_ARG, "hc_time_score_kernel: bad argument");
    if (!c->have_reads) return fail(HC_ERR_STATE, "hc_time_score_kernel: hc_set_reads has not been called");
    HC_HIP(hipSetDevice(c->device));
    HC_HIP(hipEventRecord(c->ev0, c->stream));
    for (int i = 0; i < iters; i++) {
        int rc = fmt == HC_REC_COMPACT ? hc_score_cands_device(c, d_in, n, d_out, c->stream) : hc_score_batch_device(c, d_in, n, d_out, c->stream);
        if (rc) return rc;
    }
    HC_HIP(hipEventRecord(c->ev1, c->stream));
    HC_HIP(hipEventSynchronize(c->ev1));
    float ms = 0;
    HC_HIP(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    *ms_per_launch = ms / (float)iters;
    return HC_OK;
}

int hc_count_positions_device(hc_ctx* c, uint32_t fmt, const void* d_in, uint64_t n, uint64_t* total_positions, uint64_t* total_subs) {
    if (!c || !total_positions || !total_subs || (fmt != HC_REC_FULL && fmt != HC_REC_COMPACT))
        return fail(HC_ERR_ARG, "hc_count_positions_device: bad argument");
    if (!c->have_reads) return fail(HC_ERR_STATE, "hc_count_positions_device: hc_set_reads has not been called");
    HC_HIP(hipSetDevice(c->device));
    HC_HIP(hipMemsetAsync(c->d_totals, 0, 2 * sizeof(unsigned long long), c->stream));
    HC_HIP(hc::launch_count_positions(c->view, c->params.min_read_len, fmt, d_in, n, c->d_totals, c->stream));
    unsigned long long h[2] = {0, 0};
    HC_HIP(hipMemcpyAsync(h, c->d_totals, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HC_HIP(hipStreamSynchronize(c->stream));
    *total_positions = h[0];
    *total_subs = h[1];
    return HC_OK;
}

int hc_finalize(const hc_settings* s, const hc_result_rec* r, double* score, double* mismatch_rate, uint32_t* cls) {
    if (!s || !r) return fail(HC_ERR_ARG, "hc_finalize: null argument");
    const uint32_t dcls = HC_RES_CLS(*r);
    if (dcls == HC_CLS_ERROR) {
        if (score) *score = 0;
        if (mismatch_rate) *mismatch_rate = -1;
        if (cls) *cls = HC_CLS_ERROR;
        return HC_ERR_DATA;
    }
    const uint32_t n = HC_RES_N(*r);
    const double mrate = (float)r->mm / (double)n;  // EdgeCalculator.cpp:132
    const double ov1 = exp(r->x1);                   // :138 (exp(-inf) == 0: every `return 0` of overlap_score)
    double sc;
    if (r->x2 != r->x2) {  // one sub-overlap (s-s)
        sc = ov1;
    } else {
        const double ov2 = exp(r->x2);
        if (ov1 > s->edge_threshold && ov2 > s->edge_threshold) sc = 0.5 * (ov1 + ov2);  // :256-258
        else sc = ov2 < ov1 ? ov2 : ov1;                                                  // std::min, :260
    }
    uint32_t k;
    if (sc > s->edge_threshold) k = HC_CLS_EDGE;             // :404
    else if (mrate <= s->merge_contigs) k = HC_CLS_EDGE_MC;  // :407 (mismatch_rate is never -1 here)
    else if (sc > s->ov_threshold) k = HC_CLS_NONEDGE;       // :410
    else k = HC_CLS_DROP;
    if (score) *score = sc;
    if (mismatch_rate) *mismatch_rate = mrate;
    if (cls) *cls = k;
    return HC_OK;
}

int hc_finalize_batch(const hc_settings* s, const hc_result_rec* r, uint64_t n, double* score, double* mismatch_rate,
                      uint32_t* cls) {
    if (!s || (!r && n)) return fail(HC_ERR_ARG, "hc_finalize_batch: null argument");
    int rc = HC_OK;
    for (uint64_t i = 0; i < n; i++) {
        double sc, mr;
        uint32_t k;
        if (hc_finalize(s, &r[i], &sc, &mr, &k) != HC_OK) rc = HC_ERR_DATA;
        if (score) score[i] = sc;
        if (mismatch_rate) mismatch_rate[i] = mr;
        if (cls) cls[i] = k;
    }
    if (rc) return fail(rc, "hc_finalize_batch: a record touched an invalid base/quality");
    return HC_OK;
}

int hc_device_bus_id(int32_t device, char* bus_id, uint32_t cap) {
    if (!bus_id || cap < 16) return fail(HC_ERR_ARG, "hc_device_bus_id: null or short buffer");
    bus_id[0] = 0;
    HC_HIP(hipDeviceGetPCIBusId(bus_id, (int)cap, device));
    return HC_OK;
}

int hc_get_info(hc_ctx* c, uint32_t* qual_alphabet, uint64_t* store_bytes, double* x_edge_lo, double* x_edge_hi,
                double* x_ov_lo, double* x_ov_hi) {
    if (!c) return fail(HC_ERR_ARG, "hc_get_info: null context");
    if (qual_alphabet) *qual_alphabet = c->have_reads ? c->view.K : 0;
    if (store_bytes) *store_bytes = c->store_bytes;
    if (x_edge_lo) *x_edge_lo = c->params.edge.lo;
    if (x_edge_hi) *x_edge_hi = c->params.edge.hi;
    if (x_ov_lo) *x_ov_lo = c->params.ov.lo;
    if (x_ov_hi) *x_ov_hi = c->params.ov.hi;
    return HC_OK;
}

int hc_get_kernel_info_for(hc_ctx* c, uint64_t n, char* buf, uint32_t cap) {
    if (!c || !buf || cap == 0) return fail(HC_ERR_ARG, "hc_get_kernel_info: null argument");
    buf[0] = 0;
    if (!c->have_reads) return fail(HC_ERR_STATE, "hc_get_kernel_info: hc_set_reads has not been called");
    const std::string d = hc::describe_score_kernel(c->view, c->coop_fetch ? 0 : c->fetch_group, c->fetch_group, c->n_cu, n);
    snprintf(buf, cap, "%s", d.c_str());
    return HC_OK;
}

int hc_get_kernel_info(hc_ctx* c, char* buf, uint32_t cap) { return hc_get_kernel_info_for(c, 0, buf, cap); }

}  // extern "C"
