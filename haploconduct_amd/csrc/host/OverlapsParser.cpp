// OverlapsParser.cpp — tokenizer / validator / prefilter of construct_edges over an mmap'ed
// overlaps file (reference src/EdgeCalculator.cpp:569-635).
#include "OverlapsParser.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstring>

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

OverlapsParser::OverlapsParser(const std::string& path, const ProgramSettings& ps, const FastqStorage& fastq)
    : m_ps(ps), m_fastq(fastq) {
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
}

OverlapsParser::~OverlapsParser() {
    if (m_data) munmap((void*)m_data, m_size);
    if (m_fd >= 0) close(m_fd);
}

bool OverlapsParser::next_batch(std::vector<ParsedOverlap>& batch, size_t max_batch, std::vector<Overlap>& rejected,
                                ParseCounters& c, bool print_malformed) {
    batch.clear();
    if (!m_open || m_done) return false;
    const bool allow_spaces = m_ps.allow_spaces;
    const char* field[14];
    size_t flen[14];
    while (batch.size() < max_batch) {
        if (m_pos >= m_size) { m_done = true; break; }                     // getline fails at EOF
        const char* nl = (const char*)memchr(m_data + m_pos, '\n', m_size - m_pos);
        const size_t end = nl ? (size_t)(nl - m_data) : m_size;
        const char* line = m_data + m_pos;
        const size_t n = end - m_pos;
        m_pos = nl ? end + 1 : m_size;
        if (!(m_line_no < m_ps.max_overlaps)) { m_done = true; break; }    // `&& i < max_overlaps`, :581
        m_line_no++;
        c.lines_read++;
        const int nf = split_overlap_line(line, n, allow_spaces, field, flen, 14);
        if (nf != 13) {                                                    // :598-603
            c.malformed++;
            if (print_malformed) puts("incorrect overlap; skipping");
            continue;
        }
        ParsedOverlap po;
        po.line = Overlap::from_fields(field, flen);
        const Overlap& o = po.line;
        if (o.m_id1 == o.m_id2) { c.self_overlaps++; continue; }           // :605-607
        const unsigned int perc = o.get_perc();
        const bool ss = o.m_type1 == 's' && o.m_type2 == 's';
        const bool anyp = o.m_type1 == 'p' || o.m_type2 == 'p';
        bool pass = false;
        if (o.m_len1 >= m_ps.min_overlap_len && ss) {                      // :612-617
            pass = perc >= m_ps.min_overlap_perc;
            if (!pass) c.silently_dropped++;
        } else if (o.m_len1 >= 0.5 * m_ps.min_overlap_len && o.m_len2 >= 0.5 * m_ps.min_overlap_len && anyp) {  // :618-624
            pass = perc >= m_ps.min_overlap_perc;
            if (!pass) c.silently_dropped++;
        } else if (m_ps.relax_PE_edges && o.m_len1 + o.m_len2 >= m_ps.min_overlap_len && anyp) {                // :626-632
            pass = perc >= m_ps.min_overlap_perc;
            if (!pass) c.silently_dropped++;
        } else {                                                           // :633-635
            rejected.push_back(o);
            c.prefilter_rejected++;
        }
        if (!pass) continue;
        // id -> index: std::map::at in compute_overlap, :170-171 (throws => the reference aborts)
        auto i1 = m_fastq.m_ID_to_index.find(o.m_id1);
        auto i2 = m_fastq.m_ID_to_index.find(o.m_id2);
        if (i1 == m_fastq.m_ID_to_index.end() || i2 == m_fastq.m_ID_to_index.end())
            throw FatalError{HC_ERR_BAD_OVERLAP, "overlap refers to a read id that is not in the FASTQ input"};
        hc_overlap_rec& r = po.rec;
        r.read1 = i1->second;
        r.read2 = i2->second;
        r.pos1 = o.m_pos1;
        r.pos2 = o.m_pos2;
        r.ori1 = o.m_ori1 == '+';
        r.ori2 = o.m_ori2 == '+';
        r.ord = (uint8_t)o.m_ord;
        r.flags = (uint8_t)((o.m_type1 == 'p') | ((o.m_type2 == 'p') << 1));
        r.len1 = o.m_len1;
        r.len2 = o.m_len2;
        r.perc = perc;
        batch.push_back(po);
    }
    return !batch.empty() || !m_done;
}

}  // namespace hc
