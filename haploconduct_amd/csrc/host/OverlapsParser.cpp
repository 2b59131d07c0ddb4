// OverlapsParser.cpp — tokenizer / validator / prefilter of construct_edges over an mmap'ed
// overlaps file (reference src/EdgeCalculator.cpp:569-635).
#include "OverlapsParser.h"

#include "WorkerPool.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <chrono>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

namespace hc {

static inline bool is_ws(char c) { return c == '\t' || c == ' '; }

int split_overlap_line(const char* s, size_t n, bool allow_spaces, const char* field[], size_t len[], int max_fields) {
    // boost::trim_if(line, is_any_of("\t ")), src/EdgeCalculator.cpp:584
    size_t a = 0, b = n;
    while (a < b && is_ws(s[a])) a++;
    while (b > a && is_ws(s[b - 1])) b--;
    int nf = 0;
    if (allow_spaces) {
        // boost::split(..., is_any_of("\t "), token_compress_on), :587: an empty input yields one empty token
        size_t p = a;
        for (;;) {
            size_t q = p;
            while (q < b && !is_ws(s[q])) q++;
            if (nf < max_fields) { field[nf] = s + p; len[nf] = q - p; }
            nf++;
            if (q >= b) break;
            while (q < b && is_ws(s[q])) q++;
            p = q;
        }
        return nf;
    }
    // while (getline(ss, tmp, '\t')), :590-594: an empty input yields no token
    if (a == b) return 0;
    size_t p = a;
    for (;;) {
        const char* t = (const char*)memchr(s + p, '\t', b - p);
        const size_t q = t ? (size_t)(t - s) : b;
        if (nf < max_fields) { field[nf] = s + p; len[nf] = q - p; }
        nf++;
        if (!t) break;
        p = q + 1;
        if (p >= b) break;  // cannot happen after the trim; getline yields nothing for a trailing delimiter
    }
    return nf;
}

IdIndex::IdIndex(const FastqStorage& fastq) {
    // m_ID_to_index (a std::map, src/FastqStorage.h:83-96) maps an id to its FIRST occurrence in m_read_vec; the same
    // table is built from m_read_vec itself, in index order, without walking the tree
    const std::vector<Read*>& reads = fastq.m_read_vec;
    const size_t n = reads.size();
    read_id_t max_id = 0;
    for (const Read* r : reads) max_id = r->get_read_id() > max_id ? r->get_read_id() : max_id;
    if (n == 0 || max_id < 8 * n + 1024) {
        m_direct = true;
        m_table.assign(n ? (size_t)max_id + 1 : 0, kNone);
        for (size_t i = 0; i < n; i++) {
            uint32_t& slot = m_table[reads[i]->get_read_id()];
            if (slot == kNone) slot = (uint32_t)i;
        }
    } else {
        m_direct = false;
        size_t cap = 16;
        int bits = 4;
        while (cap < 2 * n) { cap <<= 1; bits++; }
        m_shift = 64 - bits;
        m_table.assign(cap, kNone);
        m_keys.assign(cap, 0);
        for (size_t i = 0; i < n; i++) {
            const read_id_t id = reads[i]->get_read_id();
            uint64_t h = (id * 0x9E3779B97F4A7C15ull) >> m_shift;
            while (m_table[h] != kNone && m_keys[h] != id) h = (h + 1) & (cap - 1);
            if (m_table[h] == kNone) {
                m_table[h] = (uint32_t)i;
                m_keys[h] = id;
            }
        }
    }
}

// One contiguous piece of the file, parsed independently by one thread.
struct OverlapsParser::Segment {
    size_t begin = 0, end = 0;      // byte range, begin at a line start
    uint64_t first_line = 0;        // index of its first line in the file
    uint64_t line_limit = 0;        // lines with index >= line_limit are not read (max_overlaps, :581)
    Overlap* out_line = nullptr;        // where this segment's passing candidates go (room for one per line + 1)
    hc_cand_rec* out_rec = nullptr;
    size_t n_pass = 0;
    std::vector<Overlap> rejected;
    std::vector<uint32_t> reject_before;  // rejected[k] precedes pass[reject_before[k]] in file order (unused: order is kept per kind)
    ParseCounters c;
    uint64_t malformed_prints = 0;
    bool failed = false;
    FatalError error{0, ""};
};

// What construct_edges does with ONE line (src/EdgeCalculator.cpp:584-635): tokenise, Overlap's constructor (which owns the reference's
// exits), the self-overlap test, the prefilter, the id look-up.  Throws what the reference exits on.  Used line by line by the segment
// parser below and, for the lines the device's parser does not read, by the stage (EdgeCalculator::score_device_parsed).
OverlapsParser::LineKind OverlapsParser::classify_line(const char* line, size_t n, Overlap& o_slot, hc_cand_rec& rec) const {
    if (!Overlap::from_plain_line(line, n, o_slot)) {  // nearly every line is plain; the rest: the reference's steps
        const char* field[14];
        size_t flen[14];
        const int nf = split_overlap_line(line, n, m_ps.allow_spaces, field, flen, 14);
        if (nf != 13) return LineKind::Malformed;  // :598-603
        o_slot = Overlap::from_fields(field, flen);
    }
    const Overlap& o = o_slot;
    if (o.m_id1 == o.m_id2) return LineKind::Self;  // :605-607
    const unsigned int perc = o.get_perc();
    const bool ss = o.m_type1 == 's' && o.m_type2 == 's';
    const bool anyp = o.m_type1 == 'p' || o.m_type2 == 'p';
    bool pass = false;
    if (o.m_len1 >= m_ps.min_overlap_len && ss) {  // :612-617
        pass = perc >= m_ps.min_overlap_perc;
    } else if (o.m_len1 >= 0.5 * m_ps.min_overlap_len && o.m_len2 >= 0.5 * m_ps.min_overlap_len && anyp) {  // :618-624
        pass = perc >= m_ps.min_overlap_perc;
    } else if (m_ps.relax_PE_edges && o.m_len1 + o.m_len2 >= m_ps.min_overlap_len && anyp) {  // :626-632
        pass = perc >= m_ps.min_overlap_perc;
    } else {  // :633-635
        return LineKind::Rejected;
    }
    if (!pass) return LineKind::Silent;
    // id -> index: std::map::at in compute_overlap, :170-171 (throws => the reference aborts)
    uint32_t r1, r2;
    if (!m_ids.find(o.m_id1, r1) || !m_ids.find(o.m_id2, r2))
        throw FatalError{HC_ERR_BAD_OVERLAP, "overlap refers to a read id that is not in the FASTQ input"};
    rec = make_cand(o, r1, r2);
    return LineKind::Pass;
}

void OverlapsParser::parse_segment(Segment& seg) const {
    size_t pos = seg.begin;
    uint64_t line_no = seg.first_line;
    try {
        while (pos < seg.end) {
            const char* nl = (const char*)memchr(m_data + pos, '\n', seg.end - pos);
            const size_t end = nl ? (size_t)(nl - m_data) : seg.end;
            const char* line = m_data + pos;
            const size_t n = end - pos;
            pos = nl ? end + 1 : seg.end;
            if (!(line_no < seg.line_limit)) break;  // `&& i < max_overlaps`, :581
            line_no++;
            seg.c.lines_read++;
            Overlap o_slot;  // copied into the block only if the line passes
            hc_cand_rec rec;
            switch (classify_line(line, n, o_slot, rec)) {
                case LineKind::Malformed: seg.c.malformed++; continue;
                case LineKind::Self: seg.c.self_overlaps++; continue;
                case LineKind::Silent: seg.c.silently_dropped++; continue;
                case LineKind::Rejected:
                    seg.rejected.push_back(o_slot);
                    seg.c.prefilter_rejected++;
                    continue;
                case LineKind::Pass: break;
            }
            seg.out_line[seg.n_pass] = o_slot;
            seg.out_rec[seg.n_pass] = rec;
            seg.n_pass++;
        }
    } catch (const FatalError& e) {
        seg.failed = true;
        seg.error = e;
    } catch (const std::exception& e) {  // vector growth: nothing may leave a pool thread
        seg.failed = true;
        seg.error = FatalError{HC_ERR_NOMEM, std::string("overlaps parser: ") + e.what()};
    }
}

OverlapsParser::OverlapsParser(const std::string& path, const ProgramSettings& ps, const FastqStorage& fastq, WorkerPool* shared_pool)
    : m_ps(ps), m_ids(fastq), m_threads(ps.n_threads ? ps.n_threads : 1) {
    m_fd = open(path.c_str(), O_RDONLY);
    if (m_fd < 0) return;
    struct stat st;
    if (fstat(m_fd, &st) != 0) { close(m_fd); m_fd = -1; return; }
    m_size = (size_t)st.st_size;
    if (m_size > 0) {
        void* p = mmap(nullptr, m_size, PROT_READ, MAP_PRIVATE, m_fd, 0);
        if (p == MAP_FAILED) { close(m_fd); m_fd = -1; return; }
        madvise(p, m_size, MADV_SEQUENTIAL);
        m_data = (const char*)p;
    }
    m_open = true;
    if (m_threads > 1) {
        if (shared_pool && shared_pool->workers() + 1 >= m_threads) {
            m_pool = shared_pool;
        } else {
            m_own_pool.reset(new WorkerPool(m_threads - 1));
            m_pool = m_own_pool.get();
        }
    }
}

OverlapsParser::OverlapsParser(std::shared_ptr<const std::string> text, const ProgramSettings& ps, const FastqStorage& fastq, WorkerPool* shared_pool)
    : m_ps(ps), m_ids(fastq), m_threads(ps.n_threads ? ps.n_threads : 1), m_text(std::move(text)) {
    if (!m_text) return;
    m_size = m_text->size();
    m_data = m_size ? m_text->data() : nullptr;
    m_open = true;
    if (m_threads > 1) {
        if (shared_pool && shared_pool->workers() + 1 >= m_threads) {
            m_pool = shared_pool;
        } else {
            m_own_pool.reset(new WorkerPool(m_threads - 1));
            m_pool = m_own_pool.get();
        }
    }
}

OverlapsParser::~OverlapsParser() {
    const bool timing = getenv("HC_STAGE_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    m_pool = nullptr;
    m_own_pool.reset();
    const double t1 = now();
    if (m_data && !m_text) munmap((void*)m_data, m_size);
    const double t2 = now();
    if (m_fd >= 0) close(m_fd);
    if (timing) fprintf(stderr, "[hc stage] overlaps parser closed: pool %.3f s, munmap %.3f s, close %.3f s\n", t1 - t0, t2 - t1, now() - t2);
}

// Parses the next block of the file with m_threads threads: the block is cut into segments at line
// starts, the lines of each segment are numbered from a parallel newline count (so that the
// max_overlaps line limit is honoured exactly), segments are parsed concurrently and concatenated in
// file order.  `max_batch` bounds the block by bytes (~64 bytes of text per accepted candidate), not
// exactly by count: batch boundaries do not influence the result (the insert is sequential anyway).
bool OverlapsParser::next_batch(ParsedBatch& batch, size_t max_batch, std::vector<Overlap>& rejected, ParseCounters& c,
                                bool print_malformed) {
    batch.clear();
    if (!m_open || m_done) return false;
    if (m_pos >= m_size || !(m_line_no < m_ps.max_overlaps)) {
        m_done = true;
        return false;
    }
    // block = up to max_batch * 40 bytes of text (a line is >= ~26 bytes), ending at a line end
    size_t block_end = m_pos + max_batch * 40;
    if (block_end >= m_size) block_end = m_size;
    else block_end = line_end_at(block_end);
    m_line_no += parse_range(m_pos, block_end, m_line_no, batch, rejected, c, print_malformed);
    m_pos = block_end;
    if (m_pos >= m_size || !(m_line_no < m_ps.max_overlaps)) m_done = true;
    return true;
}

size_t OverlapsParser::line_end_at(size_t at) const {
    if (at >= m_size) return m_size;
    const char* nl = (const char*)memchr(m_data + at, '\n', m_size - at);
    return nl ? (size_t)(nl - m_data) + 1 : m_size;
}

void OverlapsParser::copy_range(char* dst, size_t begin, size_t end, uint64_t& newlines, bool count_newlines) const {
    std::lock_guard<std::mutex> pool_guard(m_pool_mu);
    const size_t bytes = end - begin;
    unsigned int T = m_threads;
    if (bytes < (size_t)T * (1u << 18)) T = (unsigned int)(bytes >> 18) + 1;
    std::vector<uint64_t> nl(T, 0);
    std::vector<uint8_t> bad(T, 0);
    auto body = [&](unsigned int t) {
        const size_t a = begin + bytes * t / T, b = begin + bytes * (t + 1) / T;
        size_t at = a;
        if (m_fd < 0) {  // text in memory
            memcpy(dst + (a - begin), m_data + a, b - a);
            at = b;
        }
        while (at < b) {  // pread: the kernel copies out of the page cache, no page of a mapping is faulted in
            const ssize_t k = pread(m_fd, dst + (at - begin), b - at, (off_t)at);
            if (k <= 0) {
                bad[t] = 1;
                return;
            }
            at += (size_t)k;
        }
        if (count_newlines) nl[t] = (uint64_t)std::count(dst + (a - begin), dst + (b - begin), '\n');
    };
    if (T == 1 || !m_pool) {
        for (unsigned int t = 0; t < T; t++) body(t);
    } else {
        m_pool->run(T, body);
    }
    newlines = 0;
    for (unsigned int t = 0; t < T; t++) {
        if (bad[t]) throw FatalError{HC_ERR_IO, "Unable to read the overlaps file"};
        newlines += nl[t];
    }
}

// Parses bytes [m_pos, block_end) of the file with m_threads threads: the block is cut into segments at line
// starts, the lines of each segment are numbered from a parallel newline count (so that the
// max_overlaps line limit is honoured exactly), segments are parsed concurrently and concatenated in
// file order.
uint64_t OverlapsParser::parse_range(size_t range_begin, size_t block_end, uint64_t first_line, ParsedBatch& batch, std::vector<Overlap>& rejected,
                                     ParseCounters& c, bool print_malformed) {
    std::lock_guard<std::mutex> pool_guard(m_pool_mu);
    batch.clear();
    const size_t m_pos = range_begin;  // (the body below was written against the sequential reader's members)
    const uint64_t m_line_no = first_line;
    if (block_end <= m_pos) return 0;
    const size_t bytes = block_end - m_pos;
    unsigned int T = m_threads;
    if (bytes < (size_t)T * 65536) T = (unsigned int)(bytes / 65536) + 1;
    std::vector<Segment> segs(T);
    size_t cut = m_pos;
    for (unsigned int t = 0; t < T; t++) {
        segs[t].begin = cut;
        size_t e = t + 1 == T ? block_end : m_pos + bytes * (t + 1) / T;
        if (e < cut) e = cut;
        if (e < block_end) {
            const char* nl = (const char*)memchr(m_data + e, '\n', block_end - e);
            e = nl ? (size_t)(nl - m_data) + 1 : block_end;
        }
        segs[t].end = e;
        cut = e;
    }
    auto run = [&](const std::function<void(unsigned int)>& fn) {
        if (T == 1 || !m_pool) {
            for (unsigned int t = 0; t < T; t++) fn(t);
            return;
        }
        m_pool->run(T, fn);
    };
    // pass 1: lines per segment (a final piece without a trailing newline is a line too)
    std::vector<uint64_t> nlines(T, 0);
    run([&](unsigned int t) {
        // newline count, plus one for a final piece without a trailing newline
        const char* b = m_data + segs[t].begin;
        const char* e = m_data + segs[t].end;
        uint64_t k = (uint64_t)std::count(b, e, '\n');
        if (e > b && e[-1] != '\n') k++;
        nlines[t] = k;
    });
    uint64_t line = m_line_no;
    for (unsigned int t = 0; t < T; t++) {
        segs[t].first_line = line;
        segs[t].line_limit = m_ps.max_overlaps;
        line += nlines[t];
    }
    // pass 2: parse, every segment straight into the stretch of the block its line count reserves — the records
    // array is the page-locked memory the device reads from.  When every line passes (the usual case) the block is
    // complete after this pass.
    std::vector<size_t> src(T + 1, 0);
    for (unsigned int t = 0; t < T; t++) src[t + 1] = src[t] + nlines[t];
    batch.ensure(src[T]);
    run([&](unsigned int t) {
        segs[t].out_line = batch.lines.data() + src[t];
        segs[t].out_rec = batch.recs + src[t];
        parse_segment(segs[t]);
    });
    // pass 3, only when lines were left out: close the gaps.  Every segment moves towards the front and may land on
    // a neighbour's not-yet-moved entries, so the moving segments go through their workers' scratch: out, then in —
    // both by the same workers (one thread closing the gaps of a 250 000-line block costs more than parsing it on 32).
    {
        std::vector<size_t> at(T + 1, 0);
        for (unsigned int t = 0; t < T; t++) at[t + 1] = at[t] + segs[t].n_pass;
        if (at[T] != src[T]) {
            if (m_scratch.size() < T) m_scratch.resize(T);
            auto moves = [&](unsigned int t) { return segs[t].n_pass != 0 && at[t] != src[t]; };
            run([&](unsigned int t) {
                if (!moves(t)) return;
                Scratch& sc = m_scratch[t];
                if (sc.lines.size() < segs[t].n_pass) {
                    sc.lines.resize(segs[t].n_pass + segs[t].n_pass / 8);
                    sc.recs.resize(segs[t].n_pass + segs[t].n_pass / 8);
                }
                memcpy((void*)sc.lines.data(), (const void*)segs[t].out_line, segs[t].n_pass * sizeof(Overlap));
                memcpy((void*)sc.recs.data(), (const void*)segs[t].out_rec, segs[t].n_pass * sizeof(hc_cand_rec));
            });
            run([&](unsigned int t) {
                if (!moves(t)) return;
                memcpy((void*)(batch.lines.data() + at[t]), (const void*)m_scratch[t].lines.data(), segs[t].n_pass * sizeof(Overlap));
                memcpy((void*)(batch.recs + at[t]), (const void*)m_scratch[t].recs.data(), segs[t].n_pass * sizeof(hc_cand_rec));
            });
        }
        batch.n = at[T];
    }
    for (auto& sg : segs) {
        c.lines_read += sg.c.lines_read;
        c.malformed += sg.c.malformed;
        c.self_overlaps += sg.c.self_overlaps;
        c.prefilter_rejected += sg.c.prefilter_rejected;
        c.silently_dropped += sg.c.silently_dropped;
        if (print_malformed)
            for (uint64_t k = 0; k < sg.c.malformed; k++) puts("incorrect overlap; skipping");
        if (sg.failed) {  // first failing segment in file order
            batch.n = 0;
            throw sg.error;
        }
        rejected.insert(rejected.end(), sg.rejected.begin(), sg.rejected.end());
    }
    return line - m_line_no;
}

}  // namespace hc
