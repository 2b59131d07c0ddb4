// host_model.cpp — Read, FastqStorage, Overlap, OverlapGraph: the host-side mirror of the
// reference's data model for the edge-calculation path.  Own implementation; every routine
// cites the reference lines whose behaviour it reproduces.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <functional>
#include <thread>

#include "../../../include/hcedge.h"
#include "Edge.h"
#include "EdgeCalculator.h"
#include "FastqStorage.h"
#include "Overlap.h"
#include "OverlapGraph.h"

namespace hc {

// ------------------------------------------------------------------ Read
node_id_t Read::get_vertex_id(bool normal) const {  // src/Read.h:111-120
    if (normal ? !m_N_set : !m_R_set) throw FatalError{HC_ERR_STATE, "Read::get_vertex_id: vertex id not set"};
    return normal ? m_vertex_N : m_vertex_R;
}

static void check_mate(bool paired, int i) {  // asserts of src/Read.h:145-149
    if (paired ? !(i == 1 || i == 2) : i != 0) throw FatalError{HC_ERR_ARG, "Read: mate index must be 0 (single) or 1/2 (paired)"};
}

unsigned int Read::get_seq_len(int i) const {
    check_mate(m_is_paired, i);
    return m_store->seq_len(m_store->seq_index(m_index, i));
}

std::string Read::get_seq(int i) const {
    check_mate(m_is_paired, i);
    const uint32_t q = m_store->seq_index(m_index, i);
    const uint8_t* p = m_store->bases().data() + m_store->seq_off()[q];
    return std::string((const char*)p, m_store->seq_len(q));
}

std::string Read::get_phred(int i) const {
    check_mate(m_is_paired, i);
    const uint32_t q = m_store->seq_index(m_index, i);
    const uint8_t* p = m_store->quals().data() + m_store->seq_off()[q];
    return std::string((const char*)p, m_store->seq_len(q));
}

std::string Read::get_rev_comp(int i) const {  // src/Read.h:172-184 + src/Types.h:109-129
    std::string s = get_seq(i);
    std::reverse(s.begin(), s.end());
    for (char& c : s) {
        switch (c) {
            case 'A': c = 'T'; break;
            case 'T': c = 'A'; break;
            case 'C': c = 'G'; break;
            case 'G': c = 'C'; break;
            case 'N': break;
            default: throw FatalError{HC_ERR_FORMAT, "Invalid sequence character. Aborting."};
        }
    }
    return s;
}

std::string Read::get_rev_phred(int i) const {  // src/Read.h:186-201
    std::string s = get_phred(i);
    std::reverse(s.begin(), s.end());
    return s;
}

unsigned int Read::get_len() const {  // src/Read.h:203-212
    return m_is_paired ? get_seq_len(1) + get_seq_len(2) : get_seq_len(0);
}

// ------------------------------------------------------------------ FastqStorage
// The lines of a file as std::getline yields them (src/FastqStorage.cpp:42-57: split at '\n' only, a final piece
// without a newline is a line, nothing after a trailing newline), at most max_lines of them, over a read-only mapping
// instead of one std::string per line.
class LineFile {
public:
    LineFile(const std::string& path, uint64_t max_lines) : m_left(max_lines) {
        m_fd = open(path.c_str(), O_RDONLY);
        if (m_fd < 0) return;
        struct stat st;
        if (fstat(m_fd, &st) != 0 || !S_ISREG(st.st_mode)) {  // not mappable (a pipe, ...): read it whole
            FILE* f = fdopen(m_fd, "rb");
            if (!f) { close(m_fd); m_fd = -1; return; }
            char buf[1 << 16];
            size_t k;
            while ((k = fread(buf, 1, sizeof buf, f)) > 0) m_owned.append(buf, k);
            fclose(f);
            m_fd = -1;
            m_p = m_owned.data();
            m_end = m_p + m_owned.size();
            m_ok = true;
            return;
        }
        m_size = (size_t)st.st_size;
        if (m_size) {
            void* p = mmap(nullptr, m_size, PROT_READ, MAP_PRIVATE, m_fd, 0);
            if (p == MAP_FAILED) { close(m_fd); m_fd = -1; return; }
            madvise(p, m_size, MADV_SEQUENTIAL);
            m_map = p;
            m_p = (const char*)p;
            m_end = m_p + m_size;
        }
        m_ok = true;
    }
    ~LineFile() {
        if (m_map) munmap(m_map, m_size);
        if (m_fd >= 0) close(m_fd);
    }
    LineFile(const LineFile&) = delete;
    LineFile& operator=(const LineFile&) = delete;
    bool is_open() const { return m_ok; }
    size_t bytes() const { return (size_t)(m_end - m_p); }
    // the next four lines, or false when fewer than four are left (an incomplete record is ignored, :109 / :170)
    bool next4(const char* line[4], size_t len[4]) {
        if (m_left < 4) return false;
        const char* p = m_p;
        for (int k = 0; k < 4; k++) {
            if (p >= m_end) return false;
            const char* nl = (const char*)memchr(p, '\n', (size_t)(m_end - p));
            line[k] = p;
            len[k] = nl ? (size_t)(nl - p) : (size_t)(m_end - p);
            p = nl ? nl + 1 : m_end;
        }
        m_p = p;
        m_left -= 4;
        return true;
    }

private:
    int m_fd = -1;
    void* m_map = nullptr;
    size_t m_size = 0;
    std::string m_owned;
    const char* m_p = nullptr;
    const char* m_end = nullptr;
    uint64_t m_left;
    bool m_ok = false;
};

// `stream >> id` on the header line without its first character: the first whitespace-delimited token
static void first_token(const char* s, size_t n, const char*& tok, size_t& tok_len) {
    size_t a = n ? 1 : 0;
    while (a < n && isspace((unsigned char)s[a])) a++;
    size_t b = a;
    while (b < n && !isspace((unsigned char)s[b])) b++;
    tok = s + a;
    tok_len = b - a;
}

unsigned long parse_id(const char* p, size_t n);

read_id_t FastqStorage::resolve_id(const char* tok, size_t n) const {  // src/FastqStorage.cpp:112-117
    if (m_have_new_ids) return resolve_id(std::string(tok, n));
    return parse_id(tok, n);  // str_to_read_id
}

read_id_t FastqStorage::resolve_id(const std::string& token) const {  // src/FastqStorage.cpp:112-117
    if (m_have_new_ids) {
        auto it = m_new_readIDs.find(token);
        if (it == m_new_readIDs.end()) throw FatalError{HC_ERR_FORMAT, "read id '" + token + "' missing from the --IDs file"};
        return str_to_read_id(it->second);
    }
    return str_to_read_id(token);
}

void FastqStorage::read_new_ids(const std::string& path) {  // src/FastqStorage.cpp:60-90
    std::ifstream f(path.c_str());
    if (!f.is_open()) throw FatalError{HC_ERR_IO, "Unable to open read-to-overlapID file"};
    std::string line;
    unsigned int max_id = 0;
    // The reference pushes every line into ONE stringstream and takes two tab-delimited fields out of it (:69-71); what a
    // line holds beyond its second field stays in the stream and is read in front of the next line's first field (:79-80
    // clear the flags, not the contents).  `pending` is that remainder.
    std::string pending;
    while (std::getline(f, line)) {
        const std::string s = pending + line;
        pending.clear();
        const size_t t1 = s.find('\t');
        const std::string new_id = s.substr(0, t1);  // getline(ss, new_read_ID, '\t')
        std::string old_id;
        if (t1 != std::string::npos) {
            const size_t t2 = s.find('\t', t1 + 1);
            old_id = s.substr(t1 + 1, t2 == std::string::npos ? std::string::npos : t2 - t1 - 1);
            if (t2 != std::string::npos) pending = s.substr(t2 + 1);
        }
        if (old_id.empty()) throw FatalError{HC_ERR_FORMAT, "--IDs line without an old id"};  // .at(0) throws in the reference (:72)
        if (old_id[0] == '>') old_id = old_id.substr(1);
        const read_id_t nid = str_to_read_id(new_id);
        if (nid > max_id) max_id = (unsigned int)nid;
        m_new_readIDs.insert(std::make_pair(old_id, new_id));
    }
    m_largest_read_id = max_id;
    m_have_new_ids = true;
}

void FastqStorage::push_sequence(const char* s, size_t ns, const char* q, size_t nq, bool upper) {
    if (ns != nq)  // the reference would index past the shorter string (.at() throws) or misalign rev_phred
        throw FatalError{HC_ERR_BAD_READ, "FASTQ record with sequence and quality strings of different length"};
    static const std::array<uint8_t, 256> up = [] {  // toupper() once per byte value instead of once per base
        std::array<uint8_t, 256> t{};
        for (int c = 0; c < 256; c++) t[(size_t)c] = (uint8_t)toupper(c);
        return t;
    }();
    const size_t o = m_bases.size();
    m_bases.insert(m_bases.end(), (const uint8_t*)s, (const uint8_t*)s + ns);
    if (upper)
        for (size_t i = 0; i < ns; i++) m_bases[o + i] = up[m_bases[o + i]];
    m_quals.insert(m_quals.end(), (const uint8_t*)q, (const uint8_t*)q + ns);
    m_seq_off.push_back(o + ns);
}

// ---- the files on several threads ------------------------------------------------------------------------------------
// A read-only mapping of one FASTQ file and the start of every record the sequential reader (LineFile::next4) would
// return: lines as std::getline yields them, at most max_lines of them, four per record, an incomplete last record ignored.
namespace {
struct MappedFastq {
    int fd = -1;
    const char* p = nullptr;
    size_t size = 0;
    std::vector<size_t> rec;  // rec[r] = offset of record r's header line; rec[n] = where the line after the last record starts (or size)
    size_t n = 0;
    ~MappedFastq() {
        if (p) munmap((void*)p, size);
        if (fd >= 0) close(fd);
    }
    // false: not a regular file (the caller falls back to LineFile, which also reads pipes)
    bool open_and_index(const std::string& path, uint64_t max_lines, unsigned T, bool& exists) {
        fd = ::open(path.c_str(), O_RDONLY);
        exists = fd >= 0;
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) return false;
        size = (size_t)st.st_size;
        if (size == 0) {
            rec.assign(1, 0);
            return true;
        }
        void* m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) return false;
        p = (const char*)m;
        T = std::max(1u, std::min<unsigned>(T, (unsigned)(size >> (getenv("HC_FASTQ_GRAIN") ? 4 : 20)) + 1));
        std::vector<uint64_t> nl(T + 1, 0);
        auto chunk = [&](unsigned t, size_t& a, size_t& b) {
            a = size * t / T;
            b = size * (t + 1) / T;
        };
        auto run = [&](const std::function<void(unsigned)>& f) {
            std::vector<std::thread> th;
            for (unsigned t = 1; t < T; t++) th.emplace_back(f, t);
            f(0);
            for (auto& x : th) x.join();
        };
        run([&](unsigned t) {
            size_t a, b;
            chunk(t, a, b);
            nl[t + 1] = (uint64_t)std::count(p + a, p + b, '\n');
        });
        for (unsigned t = 0; t < T; t++) nl[t + 1] += nl[t];  // nl[t] = newlines in front of chunk t = index of the line its first byte is in
        const uint64_t total_lines = nl[T] + (p[size - 1] != '\n' ? 1u : 0u);
        const uint64_t used = std::min<uint64_t>(total_lines, max_lines);
        n = (size_t)(used / 4);
        rec.assign(n + 1, size);
        rec[0] = 0;
        run([&](unsigned t) {
            size_t a, b;
            chunk(t, a, b);
            uint64_t line = nl[t];  // of the byte at a
            const char* q = p + a;
            const char* const e = p + b;
            while ((q = (const char*)memchr(q, '\n', (size_t)(e - q))) != nullptr) {
                line++;  // the line that starts behind this newline
                q++;
                if ((line & 3u) == 0 && line / 4 <= n) rec[(size_t)(line / 4)] = (size_t)(q - p);
            }
        });
        return true;
    }
    // the four lines of record r
    void lines(size_t r, const char* l[4], size_t len[4]) const {
        const char* q = p + rec[r];
        const char* const e = p + size;
        for (int k = 0; k < 4; k++) {
            const char* nlp = q < e ? (const char*)memchr(q, '\n', (size_t)(e - q)) : nullptr;
            l[k] = q;
            len[k] = nlp ? (size_t)(nlp - q) : (size_t)(e - q);
            q = nlp ? nlp + 1 : e;
        }
    }
};
}  // namespace

bool FastqStorage::read_mapped(const std::string& path1, const std::string* path2, unsigned long max_reads, unsigned threads) {
    const bool paired = path2 != nullptr;
    const uint64_t max_lines = 4ull * (unsigned int)max_reads;
    MappedFastq f1, f2;
    bool e1 = false, e2 = false;
    if (!f1.open_and_index(path1, max_lines, threads, e1)) {
        if (!e1) throw FatalError{HC_ERR_IO, "Unable to open fastq file " + path1};
        return false;
    }
    if (paired && !f2.open_and_index(*path2, max_lines, threads, e2)) {
        if (!e2) throw FatalError{HC_ERR_IO, "Unable to open fastq file " + *path2};
        return false;
    }
    const double tm0 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    auto tlap = [&](const char* what) {
        if (getenv("HC_STAGE_TIMING")) fprintf(stderr, "[hc stage] fastq %s at %.3f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - tm0);
    };
    const size_t n = paired ? std::min(f1.n, f2.n) : f1.n;  // records up to the shorter file, :170
    // small input: the sequential reader is as fast (HC_FASTQ_PARALLEL_MIN / HC_FASTQ_GRAIN: test knobs — bytes from which
    // this path is taken, records per thread at least)
    size_t min_bytes = (size_t)4 << 20, grain = 4096;
    if (const char* e = getenv("HC_FASTQ_PARALLEL_MIN")) min_bytes = (size_t)strtoull(e, nullptr, 10);
    if (const char* e = getenv("HC_FASTQ_GRAIN")) grain = std::max<size_t>(1, (size_t)strtoull(e, nullptr, 10));
    if (f1.size + f2.size < min_bytes) return false;
    const unsigned T = (unsigned)std::max<size_t>(1, std::min<size_t>(threads, n / grain + 1));
    const size_t per = paired ? 2 : 1;
    // pass A: every check of the sequential reader in its order, the ids, the lengths
    std::vector<read_id_t> ids(n);
    std::vector<uint32_t> len(n * per);
    struct Err {
        size_t at = (size_t)-1;
        FatalError e{0, ""};
    };
    std::vector<Err> err(T);
    std::vector<uint64_t> bytes(T + 1, 0);
    auto run = [&](const std::function<void(unsigned)>& f) {
        std::vector<std::thread> th;
        for (unsigned t = 1; t < T; t++) th.emplace_back(f, t);
        f(0);
        for (auto& x : th) x.join();
    };
    run([&](unsigned t) {
        uint64_t sum = 0;
        for (size_t r = n * t / T; r < n * (t + 1) / T; r++) {
            try {
                const char *l1[4], *l2[4];
                size_t n1[4], n2[4] = {0, 0, 0, 0};
                f1.lines(r, l1, n1);
                if (paired) f2.lines(r, l2, n2);
                if (n1[0] == 0 || l1[0][0] != '@')
                    throw FatalError{HC_ERR_FORMAT, paired ? "Read ID does not start with @. Exiting read_pairs." : "Read ID does not start with @. Exiting read_singles."};
                const char *t1, *t2;
                size_t tn1, tn2;
                first_token(l1[0], n1[0], t1, tn1);
                if (paired) {
                    first_token(l2[0], n2[0], t2, tn2);
                    if (tn1 != tn2 || memcmp(t1, t2, tn1) != 0) throw FatalError{HC_ERR_FORMAT, "Fastq files /1 /2 are not ordered identically. Exiting read_pairs."};
                }
                const read_id_t id = resolve_id(t1, tn1);
                ids[r] = id;
                if (paired) {
                    if (n1[1] == 0 || n2[1] == 0)
                        throw FatalError{HC_ERR_BAD_READ, "paired read with ID " + std::to_string(id) + " has an empty sequence... exiting."};
                } else if (n1[1] == 0) {
                    throw FatalError{HC_ERR_BAD_READ, "single read with ID " + std::to_string(id) + " has an empty sequence... exiting."};
                }
                if (n1[1] != n1[3] || (paired && n2[1] != n2[3]))
                    throw FatalError{HC_ERR_BAD_READ, "FASTQ record with sequence and quality strings of different length"};
                len[r * per] = (uint32_t)n1[1];
                sum += n1[1];
                if (paired) {
                    len[r * per + 1] = (uint32_t)n2[1];
                    sum += n2[1];
                }
            } catch (const FatalError& e) {
                err[t].at = r;
                err[t].e = e;
                break;
            }
        }
        bytes[t + 1] = sum;
    });
    for (unsigned t = 0; t < T; t++)
        if (err[t].at != (size_t)-1) throw err[t].e;  // the first bad record in file order: threads own ascending ranges
    for (unsigned t = 0; t < T; t++) bytes[t + 1] += bytes[t];
    tlap("pass A done");
    // pass B: the bytes, each thread behind its predecessors'
    const size_t base_bytes = m_bases.size(), base_seq = m_seq_off.size() - 1;
    m_bases.resize(base_bytes + bytes[T]);
    m_quals.resize(base_bytes + bytes[T]);
    m_seq_off.resize(base_seq + 1 + n * per);
    tlap("arrays sized");
    static const std::array<uint8_t, 256> up = [] {
        std::array<uint8_t, 256> t{};
        for (int c = 0; c < 256; c++) t[(size_t)c] = (uint8_t)toupper(c);
        return t;
    }();
    run([&](unsigned t) {
        uint64_t o = base_bytes + bytes[t];
        for (size_t r = n * t / T; r < n * (t + 1) / T; r++) {
            const char* l[4];
            size_t ln[4];
            for (size_t m = 0; m < per; m++) {
                (m ? f2 : f1).lines(r, l, ln);
                const size_t k = ln[1];
                if (!paired) {  // boost::to_upper_copy, :122 — pairs are NOT upper-cased, :197-198
                    for (size_t i = 0; i < k; i++) m_bases[o + i] = up[(uint8_t)l[1][i]];
                } else {
                    memcpy(&m_bases[o], l[1], k);
                }
                memcpy(&m_quals[o], l[3], k);
                o += k;
                m_seq_off[base_seq + 1 + r * per + m] = o;
            }
        }
    });
    tlap("pass B done");
    std::vector<Read>& vec = paired ? m_paired_vec : m_singles_vec;
    vec.reserve(vec.size() + n);
    m_first.reserve(m_first.size() + n);
    for (size_t r = 0; r < n; r++) {
        m_first.push_back(m_first.back() + (uint32_t)per);
        vec.emplace_back(this, 0u, paired, ids[r]);
    }
    return true;
}

void FastqStorage::read_singles(const std::string& path, unsigned long max_reads) {  // src/FastqStorage.cpp:92-152
    if (m_threads > 1 && read_mapped(path, nullptr, max_reads, m_threads)) return;
    LineFile f(path, 4ull * (unsigned int)max_reads);
    if (!f.is_open()) throw FatalError{HC_ERR_IO, "Unable to open fastq file " + path};
    m_bases.reserve(m_bases.size() + f.bytes() / 2);
    m_quals.reserve(m_quals.size() + f.bytes() / 2);
    const char* l[4];
    size_t n[4];
    while (f.next4(l, n)) {
        if (n[0] == 0 || l[0][0] != '@') throw FatalError{HC_ERR_FORMAT, "Read ID does not start with @. Exiting read_singles."};
        const char* tok;
        size_t tn;
        first_token(l[0], n[0], tok, tn);
        const read_id_t id = resolve_id(tok, tn);
        if (n[1] == 0) throw FatalError{HC_ERR_BAD_READ, "single read with ID " + std::to_string(id) + " has an empty sequence... exiting."};
        push_sequence(l[1], n[1], l[3], n[3], /*upper=*/true);  // boost::to_upper_copy, :122
        m_first.push_back(m_first.back() + 1);
        m_singles_vec.emplace_back(this, 0u, false, id);
    }
}

void FastqStorage::read_pairs(const std::string& p1, const std::string& p2, unsigned long max_reads) {  // :154-235
    if (m_threads > 1 && read_mapped(p1, &p2, max_reads, m_threads)) return;
    LineFile f1(p1, 4ull * (unsigned int)max_reads);
    if (!f1.is_open()) throw FatalError{HC_ERR_IO, "Unable to open fastq file " + p1};
    LineFile f2(p2, 4ull * (unsigned int)max_reads);
    if (!f2.is_open()) throw FatalError{HC_ERR_IO, "Unable to open fastq file " + p2};
    m_bases.reserve(m_bases.size() + (f1.bytes() + f2.bytes()) / 2);
    m_quals.reserve(m_quals.size() + (f1.bytes() + f2.bytes()) / 2);
    const char *l1[4], *l2[4];
    size_t n1[4], n2[4];
    while (f1.next4(l1, n1) && f2.next4(l2, n2)) {  // records up to the shorter file, :170
        if (n1[0] == 0 || l1[0][0] != '@') throw FatalError{HC_ERR_FORMAT, "Read ID does not start with @. Exiting read_pairs."};
        const char *t1, *t2;
        size_t tn1, tn2;
        first_token(l1[0], n1[0], t1, tn1);
        first_token(l2[0], n2[0], t2, tn2);
        if (tn1 != tn2 || memcmp(t1, t2, tn1) != 0) throw FatalError{HC_ERR_FORMAT, "Fastq files /1 /2 are not ordered identically. Exiting read_pairs."};
        const read_id_t id = resolve_id(t1, tn1);
        if (n1[1] == 0 || n2[1] == 0)
            throw FatalError{HC_ERR_BAD_READ, "paired read with ID " + std::to_string(id) + " has an empty sequence... exiting."};
        push_sequence(l1[1], n1[1], l1[3], n1[3], /*upper=*/false);  // pairs are NOT upper-cased, :197-198
        push_sequence(l2[1], n2[1], l2[3], n2[3], false);
        m_first.push_back(m_first.back() + 2);
        m_paired_vec.emplace_back(this, 0u, true, id);
    }
}

FastqStorage::FastqStorage(const ProgramSettings& ps) {  // src/FastqStorage.h:58-98
    const bool timing = getenv("HC_STAGE_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    m_threads = std::max(1u, std::min(16u, (unsigned)ps.n_threads));
    if (const char* e = getenv("HC_FASTQ_THREADS")) m_threads = (unsigned)std::max(1, atoi(e));
    if (!ps.id_correspondence.empty()) read_new_ids(ps.id_correspondence);
    if (!ps.singles_file.empty() && ps.singles_file != "None") read_singles(ps.singles_file, ps.max_reads);
    m_readcount_single = (unsigned int)m_singles_vec.size();
    if (!ps.paired1_file.empty() && ps.paired1_file != "None") read_pairs(ps.paired1_file, ps.paired2_file, ps.max_reads);
    m_readcount_paired = (unsigned int)m_paired_vec.size();
    const double t1 = now();
    struct Lap {
        bool on;
        double t1, t0;
        std::function<double()> now;
        ~Lap() {
            if (on) fprintf(stderr, "[hc stage] FastqStorage: files %.3f s, read vector + id index %.3f s\n", t1 - t0, now() - t1);
        }
    } lap{timing, t1, t0, now};
    if (ps.verbose) {
        printf("Singles: %u\n", m_readcount_single);
        printf("Pairs: %u\n", m_readcount_paired);
    }
    unsigned int count = 0;
    m_read_vec.reserve(m_singles_vec.size() + m_paired_vec.size());
    for (auto& r : m_singles_vec) {
        r = Read(this, count, false, r.get_read_id());
        m_read_vec.push_back(&r);
        m_ID_to_index.emplace_hint(m_ID_to_index.end(), r.get_read_id(), count);  // first occurrence wins, as insert(); ids mostly ascend
        count++;
    }
    for (auto& r : m_paired_vec) {
        r = Read(this, count, true, r.get_read_id());
        m_read_vec.push_back(&r);
        m_ID_to_index.emplace_hint(m_ID_to_index.end(), r.get_read_id(), count);
        count++;
    }
}

Read* FastqStorage::get_read(read_id_t ID) {
    auto it = m_ID_to_index.find(ID);
    if (it == m_ID_to_index.end()) throw FatalError{HC_ERR_BAD_OVERLAP, "read id not in the FASTQ input"};
    return m_read_vec[it->second];
}

// ------------------------------------------------------------------ Overlap
static void strip(std::string& s, const char* drop) {
    s.erase(std::remove_if(s.begin(), s.end(), [&](char c) { return strchr(drop, c) != nullptr; }), s.end());
}

static bool all_digits(const char* p, size_t n) {
    if (n == 0 || n > 18) return false;
    for (size_t i = 0; i < n; i++)
        if (p[i] < '0' || p[i] > '9') return false;
    return true;
}

unsigned long parse_id(const char* p, size_t n) {  // strtoul(s, NULL, 0), src/Types.h:99-102
    if (all_digits(p, n) && (p[0] != '0' || n == 1)) {
        unsigned long v = 0;
        for (size_t i = 0; i < n; i++) v = v * 10 + (unsigned long)(p[i] - '0');
        return v;
    }
    return strtoul(std::string(p, n).c_str(), nullptr, 0);
}

static int parse_int(const char* p, size_t n) {  // atoi
    if (n <= 9 && all_digits(p, n)) {
        int v = 0;
        for (size_t i = 0; i < n; i++) v = v * 10 + (p[i] - '0');
        return v;
    }
    return atoi(std::string(p, n).c_str());
}

static char one_char(const char* p, size_t n, const char* strip_set, const char* what) {
    if (n == 1) return p[0];
    std::string s(p, n);
    strip(s, strip_set);
    if (s.size() != 1) throw FatalError{HC_ERR_FORMAT, std::string("overlap field '") + what + "' is not a single character"};
    return s[0];
}

bool Overlap::from_plain_line(const char* s, size_t n, Overlap& o) {
    const char* p = s;
    const char* const e = s + n;
    // digits (at most max_digits, the range the general path converts by hand too) followed by a tab
    auto number = [&](unsigned max_digits, bool dash_ok, uint64_t& v, bool& dash) -> bool {
        dash = false;
        v = 0;
        if (dash_ok && p < e && *p == '-') {  // atoi("-") == 0
            dash = true;
            p++;
        } else {
            const char* b = p;
            while (p < e && (unsigned)(*p - '0') <= 9u) v = v * 10 + (uint64_t)(*p++ - '0');
            const size_t d = (size_t)(p - b);
            if (d == 0 || d > max_digits) return false;
            if (!dash_ok && b[0] == '0' && d > 1) return false;  // strtoul(.., 0) reads a leading 0 as octal
        }
        if (p >= e || *p != '\t') return false;
        p++;
        return true;
    };
    auto character = [&](char& c, bool last) -> bool {
        if (p >= e) return false;
        c = *p++;
        if (last) return p == e;
        if (p >= e || *p != '\t') return false;
        p++;
        return true;
    };
    uint64_t id1, id2, pos1, pos2, perc1, perc2, len1, len2;
    bool dash, dash_pos2;
    char ord, ori1, ori2, type1, type2;
    if (!number(18, false, id1, dash) || !number(18, false, id2, dash) || !number(9, true, pos1, dash) ||
        !number(9, true, pos2, dash_pos2) || !character(ord, false) || !character(ori1, false) || !character(ori2, false) ||
        !number(9, true, perc1, dash) || !number(9, true, perc2, dash) || !number(9, true, len1, dash) ||
        !number(9, true, len2, dash) || !character(type1, false) || !character(type2, true))
        return false;
    if (dash_pos2) perc2 = len2 = 0;  // src/Overlap.h:55-59
    if ((ori1 != '+' && ori1 != '-') || (ori2 != '+' && ori2 != '-')) return false;
    if (perc1 > 100 || perc2 > 100) return false;
    if ((type1 != 's' && type1 != 'p') || (type2 != 's' && type2 != 'p')) return false;
    if (type1 == 's' || type2 == 's' ? ord != '-' : (ord != '1' && ord != '2')) return false;
    o.m_id1 = id1;
    o.m_id2 = id2;
    o.m_pos1 = (unsigned int)pos1;
    o.m_pos2 = (unsigned int)pos2;
    o.m_perc1 = (unsigned int)perc1;
    o.m_perc2 = (unsigned int)perc2;
    o.m_len1 = (unsigned int)len1;
    o.m_len2 = (unsigned int)len2;
    o.m_ord = ord;
    o.m_ori1 = ori1;
    o.m_ori2 = ori2;
    o.m_type1 = type1;
    o.m_type2 = type2;
    return true;
}

Overlap Overlap::from_fields(const char* const f[13], const size_t n[13]) {  // src/Overlap.h:39-73
    Overlap o;
    o.m_id1 = parse_id(f[0], n[0]);
    o.m_id2 = parse_id(f[1], n[1]);
    o.m_pos1 = (unsigned int)parse_int(f[2], n[2]);
    o.m_pos2 = (unsigned int)parse_int(f[3], n[3]);
    o.m_perc1 = (unsigned int)parse_int(f[7], n[7]);
    o.m_perc2 = (unsigned int)parse_int(f[8], n[8]);
    o.m_len1 = (unsigned int)parse_int(f[9], n[9]);
    o.m_len2 = (unsigned int)parse_int(f[10], n[10]);
    if (n[3] == 1 && f[3][0] == '-') {  // :55-59
        o.m_pos2 = 0;
        o.m_perc2 = 0;
        o.m_len2 = 0;
    }
    if ((int)o.m_pos1 < 0 || (int)o.m_pos2 < 0) throw FatalError{HC_ERR_FORMAT, "overlap.m_pos < 0; Exiting."};  // :102-107
    o.m_ori1 = one_char(f[5], n[5], " ", "ori1");                                                                 // :121-130
    o.m_ori2 = one_char(f[6], n[6], " ", "ori2");
    if ((o.m_ori1 != '+' && o.m_ori1 != '-') || (o.m_ori2 != '+' && o.m_ori2 != '-'))
        throw FatalError{HC_ERR_FORMAT, "overlap.m_ori not of the right format (+, -). Exiting."};
    if ((int)o.m_perc1 < 0 || (int)o.m_perc1 > 100 || (int)o.m_perc2 < 0 || (int)o.m_perc2 > 100)               // :132-138
        throw FatalError{HC_ERR_FORMAT, "overlap.m_perc not of the right format (0 <= perc <= 100). Exiting."};
    if ((int)o.m_len1 < 0 || (int)o.m_len2 < 0) throw FatalError{HC_ERR_FORMAT, "overlap.m_len < 0. Exiting."};  // :140-145
    o.m_type1 = one_char(f[11], n[11], "\n\t ", "type1");                                                         // :147-158
    o.m_type2 = one_char(f[12], n[12], "\n\t ", "type2");
    if ((o.m_type1 != 's' && o.m_type1 != 'p') || (o.m_type2 != 's' && o.m_type2 != 'p'))
        throw FatalError{HC_ERR_FORMAT, "overlap type not of the form 's' or 'p'. Exiting."};
    o.m_ord = one_char(f[4], n[4], " ", "ord");                                                                   // :109-119
    if (o.m_ord != '1' && o.m_ord != '2' && o.m_ord != '-') throw FatalError{HC_ERR_FORMAT, "overlap ord must be 1, 2 or -"};
    if (o.m_type1 == 's' || o.m_type2 == 's') {
        if (o.m_ord != '-') throw FatalError{HC_ERR_FORMAT, "overlap ord must be - when a read is single-end"};
    } else if (o.m_ord == '-') {
        throw FatalError{HC_ERR_FORMAT, "overlap ord must be 1 or 2 for a paired-paired overlap"};
    }
    return o;
}

static char* put_u(char* p, unsigned long v) {
    char tmp[24];
    int k = 0;
    do {
        tmp[k++] = (char)('0' + v % 10);
        v /= 10;
    } while (v);
    while (k) *p++ = tmp[--k];
    return p;
}

size_t Overlap::write_line(char* buf) const {  // src/Overlap.h:234-237
    char* p = buf;
    p = put_u(p, m_id1); *p++ = '\t';
    p = put_u(p, m_id2); *p++ = '\t';
    p = put_u(p, m_pos1); *p++ = '\t';
    p = put_u(p, m_pos2); *p++ = '\t';
    *p++ = m_ord; *p++ = '\t';
    *p++ = m_ori1; *p++ = '\t';
    *p++ = m_ori2; *p++ = '\t';
    p = put_u(p, m_perc1); *p++ = '\t';
    p = put_u(p, m_perc2); *p++ = '\t';
    p = put_u(p, m_len1); *p++ = '\t';
    p = put_u(p, m_len2); *p++ = '\t';
    *p++ = m_type1; *p++ = '\t';
    *p++ = m_type2; *p++ = '\n';
    return (size_t)(p - buf);
}

std::string Overlap::get_overlap_line() const {
    char buf[192];
    return std::string(buf, write_line(buf));
}

// ------------------------------------------------------------------ OverlapGraph
OverlapGraph::~OverlapGraph() {
    adj_out.clear();  // the lists first: the borrowed ones point into the arenas
    adj_in.clear();
    free(out_arena);
    free(in_arena);
}

void OverlapGraph::ensure_slots() const {
    if (slots_valid) return;
    slots_valid = true;
    std::vector<uint64_t> keys;
    keys.reserve(edge_count);
    for (const auto& L : adj_out)
        for (const Edge& e : L)
            if (EdgeSlotIndex::representable(e.get_vertex(1), e.get_vertex(2)))
                keys.push_back(EdgeSlotIndex::key(e.get_vertex(1), e.get_vertex(2), e.get_ori(1) == e.get_ori(2)));
    slots.bulk_add(keys.data(), keys.size(), std::max(1u, std::min(program_settings.n_threads, 16u)));
}

void OverlapGraph::addEdge(const Edge& edge) {  // src/OverlapGraph.cpp:94-101
    ensure_slots();
    const node_id_t v = edge.get_vertex(1), w = edge.get_vertex(2);
    if (adj_out[v].capacity() == 0) adj_out[v].reserve(4);  // skips the 1 -> 2 -> 4 reallocations of nearly every vertex
    if (adj_in[w].capacity() == 0) adj_in[w].reserve(8);
    adj_out[v].push_back(edge);
    adj_in[w].push_back(v);
    edge_count++;
    if (EdgeSlotIndex::representable(v, w)) slots.add(EdgeSlotIndex::key(v, w, edge.get_ori(1) == edge.get_ori(2)));
}

static inline bool same_ori_class(const Edge& e, bool opposite_orientations) {
    return (e.get_ori(1) == e.get_ori(2)) == opposite_orientations;
}

Edge OverlapGraph::removeEdgeWithOri(node_id_t v, node_id_t w, bool opposite_orientations) {  // :150-194
    ensure_slots();
    auto& L = adj_out.at(v);
    Edge removed;
    bool found = false;
    for (auto it = L.begin(); it != L.end(); ++it) {
        if (it->get_vertex(2) == w && same_ori_class(*it, opposite_orientations)) {
            removed = *it;
            L.erase(it);
            edge_count--;
            found = true;
            if (EdgeSlotIndex::representable(v, w)) slots.remove(EdgeSlotIndex::key(v, w, opposite_orientations));
            break;
        }
    }
    if (!found) throw FatalError{HC_ERR_STATE, "Edge to be removed not found..."};
    auto& I = adj_in.at(w);
    for (auto it = I.begin(); it != I.end(); ++it) {
        if (*it == v) {
            I.erase(it);
            break;
        }
    }
    return removed;
}

Edge OverlapGraph::removeEdge(node_id_t v, node_id_t w) {  // :102-146: the first edge v -> w, the first v in w's in-list
    ensure_slots();
    auto& L = adj_out.at(v);
    Edge removed;
    bool found = false;
    for (auto it = L.begin(); it != L.end(); ++it) {
        if (it->get_vertex(2) == w) {
            removed = *it;
            if (EdgeSlotIndex::representable(v, w)) slots.remove(EdgeSlotIndex::key(v, w, it->get_ori(1) == it->get_ori(2)));
            L.erase(it);
            edge_count--;
            found = true;
            break;
        }
    }
    if (!found) throw FatalError{HC_ERR_STATE, "Edge to be removed not found..."};
    auto& I = adj_in.at(w);
    for (auto it = I.begin(); it != I.end(); ++it) {
        if (*it == v) {
            I.erase(it);
            break;
        }
    }
    return removed;
}

double OverlapGraph::checkEdge(node_id_t v, node_id_t w, bool reverse_allowed) const {  // :233-259
    for (const Edge& e : adj_out.at(v))
        if (e.get_vertex(2) == w) return e.get_score();
    if (reverse_allowed)
        for (const Edge& e : adj_out.at(w))
            if (e.get_vertex(2) == v) return e.get_score();
    return -1;
}

void OverlapGraph::addEquivalentEdges(unsigned int* n_built, unsigned int* n_doubles) {  // :608-719
    const size_t V = adj_out.size();
    std::vector<std::vector<Edge>> extra(V);
    for (size_t i = 0; i < V; i++) {  // :618-668
        for (const Edge& it : adj_out[i]) {
            int pos1 = it.get_extra_pos(1), pos2 = it.get_extra_pos(2);
            bool ori1, ori2;
            char ord;
            Read *read_1, *read_2;
            const bool no_ord = it.get_ord() == '-' || it.get_ord() == '0';
            if (pos1 < 0) {
                read_1 = it.get_read(2);
                read_2 = it.get_read(1);
                ori1 = !it.get_ori(2);
                ori2 = !it.get_ori(1);
                pos1 = -pos1;
                if (pos2 < 0) {
                    ord = '1';
                    pos2 = -pos2;
                } else {
                    ord = no_ord ? '-' : '2';
                }
            } else {
                read_1 = it.get_read(1);
                read_2 = it.get_read(2);
                ori1 = !it.get_ori(1);
                ori2 = !it.get_ori(2);
                if (pos2 < 0) {
                    pos2 = -pos2;
                    ord = '2';
                } else {
                    ord = no_ord ? '-' : '1';
                }
            }
            Edge e(it.get_score(), pos1, pos2, ori1, ori2, ord, read_1, read_2);
            const node_id_t node1 = read_1->get_vertex_id(ori1), node2 = read_2->get_vertex_id(ori2);
            e.set_vertices(node1, node2);
            e.set_len(it.get_len(1), it.get_len(2));
            e.set_perc(it.get_perc());
            extra.at(node1).push_back(e);
        }
    }
    unsigned int count = 0, doubles = 0;
    for (size_t i = 0; i < V; i++) {  // :670-709
        for (Edge it : extra[i]) {
            node_id_t v1 = it.get_vertex(1), v2 = it.get_vertex(2);
            if (it.get_pos(1) == 0 && v1 > v2) {  // either direction is possible: small id to large id
                std::swap(v1, v2);
                it.swap_reads();
            }
            const double score = checkEdge(v1, v2, /*reverse_allowed=*/false);
            if (score < 0) {
                addEdge(it);
                count++;
            } else if (it.get_score() > score) {
                removeEdge(v1, v2);
                addEdge(it);
                doubles++;
            } else {
                doubles++;
            }
        }
    }
    if (n_built) *n_built = count;
    if (n_doubles) *n_doubles = doubles;
}

double OverlapGraph::checkEdgeWithOri(node_id_t v, node_id_t w, bool opposite_orientations) const {  // :198-229
    ensure_slots();
    if (EdgeSlotIndex::representable(v, w) && !slots.contains(EdgeSlotIndex::key(v, w, opposite_orientations))) return -1;
    for (const Edge& e : adj_out.at(v))
        if (e.get_vertex(2) == w && same_ori_class(e, opposite_orientations)) return e.get_score();
    for (const Edge& e : adj_out.at(w))
        if (e.get_vertex(2) == v && same_ori_class(e, opposite_orientations)) return e.get_score();
    return -1;
}

Edge* OverlapGraph::getEdgeInfoWithOri(node_id_t v, node_id_t w, bool opposite_orientations, bool reverse_allowed) {  // :285-306
    for (Edge& e : adj_out.at(v))
        if (e.get_vertex(2) == w && same_ori_class(e, opposite_orientations)) return &e;
    if (reverse_allowed)
        for (Edge& e : adj_out.at(w))
            if (e.get_vertex(2) == v && same_ori_class(e, opposite_orientations)) return &e;
    throw FatalError{HC_ERR_STATE, "Edge not found. Exiting."};
}

// ------------------------------------------------------------------ pieces of EdgeCalculator without device dependencies
hc_settings to_hc_settings(const ProgramSettings& ps) {
    hc_settings s;
    memset(&s, 0, sizeof s);
    s.edge_threshold = ps.edge_threshold;
    s.ov_threshold = ps.ov_threshold;
    s.merge_contigs = ps.merge_contigs;
    s.mismatch = ps.mismatch;
    s.min_read_len = ps.min_read_len;
    s.min_overlap_len = ps.min_overlap_len;
    s.min_overlap_perc = ps.min_overlap_perc;
    s.flags = (ps.add_duplicates ? HC_FLAG_ADD_DUPLICATES : 0u) | (ps.resolve_orientations ? HC_FLAG_RESOLVE_ORIENTATIONS : 0u) |
              (ps.ignore_inclusions ? HC_FLAG_IGNORE_INCLUSIONS : 0u) | (ps.relax_PE_edges ? HC_FLAG_RELAX_PE_EDGES : 0u) |
              (ps.allow_spaces ? HC_FLAG_ALLOW_SPACES : 0u) | (ps.verbose ? HC_FLAG_VERBOSE : 0u);
    s.max_overlaps = ps.max_overlaps;
    s.device = ps.device;
    s.device_mask = ps.device_mask;
    s.n_threads = ps.n_threads;
    return s;
}

// src/EdgeCalculator.cpp:441-538
void insert_edge(OverlapGraph& g, const ProgramSettings& program_settings, Edge& e, InsertCounters& c) {
    node_id_t v1 = e.get_vertex(1), v2 = e.get_vertex(2);
    if (e.get_pos(1) == 0 && v1 > v2) {  // :443-448: undetermined direction => small id to large id
        std::swap(v1, v2);
        e.swap_reads();
    }
    if (e.get_perc() == 100) c.inclusion_count++;  // :449-451, before de-duplication
    const bool opposite_orientations = (e.get_ori(1) == e.get_ori(2));
    const double score = g.checkEdgeWithOri(v1, v2, opposite_orientations);
    if (score < 0) {  // :455-469
        g.addEdge(e);
        c.edges_added++;
        if (program_settings.ignore_inclusions && e.get_perc() == 100 && e.get_mismatch_rate() < 0.000001 &&
            e.get_mismatch_rate() >= 0) {
            if (e.get_extra_pos(1) < 0) {
                if (e.get_pos(1) == 0) g.inclusions[v1] = 1;  // otherwise only an effect of rounding the percentage
            } else {
                g.inclusions[v2] = 1;
            }
        }
        return;
    }
    c.dup_count++;  // `doubles++` on both remaining branches, :472,537
    if (!(e.get_score() >= score)) return;  // :535-538
    Edge* ex = g.getEdgeInfoWithOri(v1, v2, opposite_orientations, true);
    if (score == e.get_score()) {  // deterministic tie-break chain, :474-521
        if (ex->get_len(0) != e.get_len(0)) {
            if (ex->get_len(0) > e.get_len(0)) return;
        } else if (ex->get_mismatch_rate() != e.get_mismatch_rate()) {
            if (ex->get_mismatch_rate() < e.get_mismatch_rate()) return;
        } else if (ex->get_vertex(1) != e.get_vertex(1)) {
            if (ex->get_vertex(1) < e.get_vertex(1)) return;
        } else if (ex->get_ori(1) != e.get_ori(1)) {
            if (ex->get_ori(1)) return;
        } else if (ex->get_ori(2) != e.get_ori(2)) {
            if (ex->get_ori(2)) return;
        } else if (ex->get_pos(1) != e.get_pos(1)) {
            if (ex->get_pos(1) < e.get_pos(1)) return;
        } else if (ex->get_pos(2) != e.get_pos(2)) {
            if (ex->get_pos(2) < e.get_pos(2)) return;
        }
    }
    if (ex->get_vertex(1) == v1) g.removeEdgeWithOri(v1, v2, opposite_orientations);  // :523-528
    else g.removeEdgeWithOri(v2, v1, opposite_orientations);
    g.addEdge(e);  // :530
}

// true when the reference keeps the existing edge although the new one scores at least as high
// (src/EdgeCalculator.cpp:474-521; all-equal falls through to "replace")
static inline bool chain_keeps_existing(const Edge& ex, const Edge& e) {
    if (ex.get_len(0) != e.get_len(0)) return ex.get_len(0) > e.get_len(0);
    if (ex.get_mismatch_rate() != e.get_mismatch_rate()) return ex.get_mismatch_rate() < e.get_mismatch_rate();
    if (ex.get_vertex(1) != e.get_vertex(1)) return ex.get_vertex(1) < e.get_vertex(1);
    if (ex.get_ori(1) != e.get_ori(1)) return ex.get_ori(1);
    if (ex.get_ori(2) != e.get_ori(2)) return ex.get_ori(2);
    if (ex.get_pos(1) != e.get_pos(1)) return ex.get_pos(1) < e.get_pos(1);
    if (ex.get_pos(2) != e.get_pos(2)) return ex.get_pos(2) < e.get_pos(2);
    return false;
}

// A few workers for the bulk phases below; plain threads — each phase is entered once per overlaps file.
template <typename F>
static void run_workers(unsigned T, F&& body) {
    if (T <= 1) {
        body(0u);
        return;
    }
    std::vector<std::thread> th;
    th.reserve(T);
    for (unsigned t = 0; t < T; t++) th.emplace_back([&body, t] { body(t); });
    for (auto& x : th) x.join();
}

// addEdge(pool[order[0]]), addEdge(pool[order[1]]), ... — every adjacency list ends up exactly as those calls in that
// order leave it — done by vertex ranges: each worker owns a contiguous range of vertices, sizes the lists of its
// range exactly, then appends to them while streaming over the (v1, v2) columns; no two workers touch one list.
void OverlapGraph::bulk_add_edges(const Edge* pool, const std::vector<uint32_t>& order, unsigned n_threads) {
    const size_t m = order.size();
    if (m == 0) return;
    ensure_slots();
    const size_t V = adj_out.size();
    bool indexable = true;
    std::vector<node_id_t> c1(m), c2(m);
    std::vector<uint64_t> keys(m);
    const unsigned T = m < (1u << 14) ? 1u : std::max(1u, std::min(n_threads, 32u));
    std::vector<uint8_t> bad(T, 0);
    run_workers(T, [&](unsigned t) {
        for (size_t k = m * t / T; k < m * (t + 1) / T; k++) {
            const Edge& e = pool[order[k]];
            c1[k] = e.get_vertex(1);
            c2[k] = e.get_vertex(2);
            if (c1[k] >= V || c2[k] >= V) bad[t] = 1;
            else if (!EdgeSlotIndex::representable(c1[k], c2[k])) bad[t] = 2;
            else keys[k] = EdgeSlotIndex::key(c1[k], c2[k], e.get_ori(1) == e.get_ori(2));
        }
    });
    for (uint8_t b : bad) {
        if (b == 1) throw FatalError{HC_ERR_STATE, "bulk_add_edges: vertex out of range"};
        if (b == 2) indexable = false;
    }
    if (!indexable) {  // ids beyond the slot index's key space: nothing to gain, keep the plain path
        for (uint32_t k : order) addEdge(pool[k]);
        return;
    }
    static const bool timing = getenv("HC_STAGE_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tp = now();
    auto lap = [&](const char* what) {
        if (timing) {
            const double t = now();
            fprintf(stderr, "[hc stage] bulk fill: %s %.3f s\n", what, t - tp);
            tp = t;
        }
    };
    std::thread indexer([&] { slots.bulk_add(keys.data(), m, std::max(1u, T / 4)); });  // independent of the lists
    std::vector<uint32_t> out_deg(V, 0), in_deg(V, 0);
    run_workers(T, [&](unsigned t) {
        const size_t lo = V * t / T, hi = V * (t + 1) / T;
        for (size_t k = 0; k < m; k++) {
            if (c1[k] >= lo && c1[k] < hi) out_deg[c1[k]]++;
            if (c2[k] >= lo && c2[k] < hi) in_deg[c2[k]]++;
        }
    });
    lap("degrees");
    // Sizing the lists (two allocations per vertex with edges) stays on one thread: when 32 workers grow their
    // allocator arenas at once the calls serialise on the process's address-space lock — measured at C3 on the
    // first call of a process: 0.38 s against 0.05 s here.
    for (size_t v = 0; v < V; v++) {
        if (out_deg[v]) adj_out[v].reserve(adj_out[v].size() + out_deg[v]);
        if (in_deg[v]) adj_in[v].reserve(adj_in[v].size() + in_deg[v]);
    }
    lap("reserve");
    run_workers(T, [&](unsigned t) {
        const size_t lo = V * t / T, hi = V * (t + 1) / T;
        for (size_t k = 0; k < m; k++) {
            if (c1[k] >= lo && c1[k] < hi) adj_out[c1[k]].push_back(pool[order[k]]);
            if (c2[k] >= lo && c2[k] < hi) adj_in[c2[k]].push_back(c1[k]);
        }
    });
    lap("append");
    indexer.join();
    lap("slot index (rest)");
    edge_count += (unsigned int)m;
}

void OverlapGraph::adopt_csr(const hc_edge_rec* edges, const uint64_t* out_off, const uint32_t* in_nodes, const uint64_t* in_off,
                             const uint8_t* inclusion_bits, Read* const* reads, size_t n_reads, unsigned n_threads,
                             const std::atomic<size_t>* edges_arrived, const std::atomic<bool>* abandon) {
    if (edge_count != 0 || out_arena || in_arena) throw FatalError{HC_ERR_STATE, "adopt_csr: the graph already holds edges"};
    const size_t V = adj_out.size();
    const size_t E = (size_t)out_off[V];
    if ((size_t)in_off[V] != E || E > 0xFFFFFFFFull) throw FatalError{HC_ERR_STATE, "adopt_csr: inconsistent offsets"};
    if (E == 0) return;
    static_assert(std::is_trivially_copyable<Edge>::value, "Edge lives in malloc'd arenas");
    const size_t huge = (size_t)2 << 20;
    const size_t bytes = (E * sizeof(Edge) + huge - 1) & ~(huge - 1);
    if (posix_memalign((void**)&out_arena, huge, bytes) != 0) {
        out_arena = nullptr;
        throw FatalError{HC_ERR_NOMEM, "adopt_csr: out of memory"};
    }
    madvise((void*)out_arena, bytes, MADV_HUGEPAGE);  // a hint: 2 MiB pages where the system grants them (fewer first-touch faults)
    in_arena = (node_id_t*)malloc(E * sizeof(node_id_t));
    if (!in_arena) throw FatalError{HC_ERR_NOMEM, "adopt_csr: out of memory"};
    const unsigned T = E < (1u << 14) ? 1u : std::max(1u, std::min(n_threads, 32u));
    std::vector<uint8_t> bad(T, 0);
    // Behind a copy that is still running, the vertices are dealt in stretches of 4 096 taken in turn, so that every thread works near the
    // copy's front; otherwise in T contiguous ranges
    std::atomic<size_t> next_stretch{0};
    const size_t stretch = edges_arrived ? 4096 : (V + T - 1) / T;
    run_workers(T, [&](unsigned t) {
        uint8_t my_bad = 0;
        try {  // Edge's own checks throw: nothing may leave a worker thread
        size_t have = edges_arrived ? 0 : E;
        for (;;) {
        const size_t v0 = edges_arrived ? next_stretch.fetch_add(stretch) : stretch * t;
        if (v0 >= V || my_bad) break;
        const size_t v1 = std::min(V, v0 + stretch);
        for (size_t v = v0; v < v1; v++) {
            const size_t a = (size_t)out_off[v], b = (size_t)out_off[v + 1];
            if (b < a || b > E) { my_bad = 1; break; }
            while (have < b) {  // the copy has not reached this vertex's records yet
                have = edges_arrived->load(std::memory_order_acquire);
                if (have >= b) break;
                if (abandon && abandon->load(std::memory_order_acquire)) { my_bad = 2; break; }
                // (a short sleep, not a yield: up to 32 of these threads wait for ONE thread that copies, which needs a core of its own)
                std::this_thread::sleep_for(std::chrono::microseconds(20));
            }
            if (my_bad) break;
            for (size_t k = a; k < b; k++) {
                const hc_edge_rec& r = edges[k];
                if (r.read1 >= n_reads || r.read2 >= n_reads || r.v1 != v || r.v2 >= V) { my_bad = 1; break; }
                Edge* e = new ((void*)(out_arena + k)) Edge(r.score, r.pos1, r.pos2, r.ori1 != 0, r.ori2 != 0, (char)r.ord, reads[r.read1], reads[r.read2]);
                e->set_vertices(r.v1, r.v2);
                e->set_extra_pos(r.pos3, r.pos4);
                e->set_perc(r.perc);
                e->set_len(r.len1, r.len2);
                e->set_mismatch(r.mismatch_rate);
            }
            if (my_bad) break;
            adj_out[v].borrow(out_arena + a, b - a, b - a);
            const size_t ia = (size_t)in_off[v], ib = (size_t)in_off[v + 1];
            if (ib < ia || ib > E) { my_bad = 1; break; }
            for (size_t k = ia; k < ib; k++) {
                if (in_nodes[k] >= V) { my_bad = 1; break; }
                in_arena[k] = in_nodes[k];
            }
            if (my_bad) break;
            adj_in[v].borrow(in_arena + ia, ib - ia, ib - ia);
            if (inclusion_bits && inclusion_bits[v]) inclusions[v] = 1;
        }
        if (!edges_arrived) break;  // one contiguous range per thread
        }
        } catch (...) {
            my_bad = 1;
        }
        bad[t] = my_bad;
    });
    for (uint8_t b : bad)
        if (b) {  // leave the graph as it was found: empty, and fit for another adopt
            for (size_t v = 0; v < V; v++) {
                adj_out[v].borrow(nullptr, 0, 0);
                adj_in[v].borrow(nullptr, 0, 0);
                inclusions[v] = 0;
            }
            free((void*)out_arena);
            free((void*)in_arena);
            out_arena = nullptr;
            in_arena = nullptr;
            throw FatalError{HC_ERR_STATE, "adopt_csr: malformed adjacency"};
        }
    edge_count = (unsigned int)E;
    slots_valid = false;
}

void OverlapGraph::sort_out_list(node_id_t v, const uint32_t* len_by_read) {  // src/OverlapGraph.cpp:724-749
    ArenaList<Edge>& L = adj_out[v];
    if (L.size() < 2) return;
    struct Key {
        unsigned int nonoverlap;  // src/Edge.h:58-63, unsigned arithmetic
        node_id_t v2;
        uint32_t at;
    };
    std::vector<Key> keys;
    keys.reserve(L.size());
    for (size_t k = 0; k < L.size(); k++) {
        const Edge& e = L[k];
        keys.push_back(Key{(unsigned int)len_by_read[e.get_read(1)->get_index()] + (unsigned int)len_by_read[e.get_read(2)->get_index()] -
                               2u * (unsigned int)e.get_len(0),
                           e.get_vertex(2), (uint32_t)k});
    }
    // the comparator of :733-742 on the same sequence: std::sort's result depends on the comparisons only
    std::sort(keys.begin(), keys.end(), [](const Key& a, const Key& b) {
        if (a.nonoverlap == b.nonoverlap) return a.v2 < b.v2;
        return a.nonoverlap < b.nonoverlap;
    });
    std::vector<Edge> sorted;
    sorted.reserve(L.size());
    for (const Key& k : keys) sorted.push_back(L[k.at]);
    std::copy(sorted.begin(), sorted.end(), L.begin());
}

// adj_in, :751-762: for every vertex in order, for every edge of its out-list, vertex1 appended to the in-list of
// vertex2 — by ranges of vertex2, every worker walking the out-lists in that same order.  In-degrees do not change, so
// lists that point into the graph's arena are refilled in place.
void OverlapGraph::rebuild_in_lists(unsigned n_threads) {
    const size_t V = adj_out.size();
    const unsigned T = edge_count < (1u << 14) ? 1u : std::max(1u, std::min(n_threads, 32u));
    run_workers(T, [&](unsigned t) {
        const size_t lo = V * t / T, hi = V * (t + 1) / T;
        for (size_t v = lo; v < hi; v++) adj_in[v].clear();
        for (size_t v = 0; v < V; v++)
            for (const Edge& e : adj_out[v]) {
                const node_id_t w = e.get_vertex(2);
                if (w >= lo && w < hi) adj_in[w].push_back(e.get_vertex(1));
            }
    });
}

void OverlapGraph::sortEdges(const uint32_t* len_by_read, unsigned n_threads) {  // src/OverlapGraph.cpp:722-764
    const size_t V = adj_out.size();
    const unsigned T = edge_count < (1u << 14) ? 1u : std::max(1u, std::min(n_threads, 32u));
    run_workers(T, [&](unsigned t) {
        for (size_t v = V * t / T; v < V * (t + 1) / T; v++) sort_out_list(v, len_by_read);
    });
    rebuild_in_lists(n_threads);
}

void EdgeSlotIndex::bulk_add(const uint64_t* keys, size_t n, unsigned n_threads) {
    size_t cap = mask_ + 1;
    while ((filled_ + n) * 10 > cap * 6) cap *= 2;
    if (cap != mask_ + 1) rebuild(cap);
    const unsigned T = n < (1u << 14) ? 1u : std::max(1u, std::min(n_threads, 16u));
    std::vector<size_t> claimed(T, 0), revived(T, 0);
    // no slot is released while this runs, so a claimed key never moves: compare-and-swap on the key word is enough
    run_workers(T, [&](unsigned t) {
        size_t n_claimed = 0, n_revived = 0;  // locals: neighbouring vector elements would share a cache line
        for (size_t i = n * t / T; i < n * (t + 1) / T; i++) {
            if (i + 8 < n * (t + 1) / T) prefetch(keys[i + 8]);
            const uint64_t k = keys[i];
            for (size_t h = slot_of(k);; h = (h + 1) & mask_) {
                uint64_t cur = __atomic_load_n(&tab_[h].key, __ATOMIC_RELAXED);
                if (cur == kEmpty) {
                    if (__atomic_compare_exchange_n(&tab_[h].key, &cur, k, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
                        n_claimed++;
                        cur = k;
                    }
                }
                if (cur == k) {
                    if (__atomic_fetch_add(&tab_[h].count, 1u, __ATOMIC_RELAXED) == 0) n_revived++;
                    break;
                }
            }
        }
        claimed[t] = n_claimed;
        revived[t] = n_revived;
    });
    for (unsigned t = 0; t < T; t++) {
        filled_ += claimed[t];
        live_ += revived[t];
    }
}

void resolve_admitted_edges(OverlapGraph& g, const ProgramSettings& ps, Edge* admitted, size_t n, InsertCounters& c) {
    if (n == 0) return;
    if (n > 0xFFFFFFFFull) throw FatalError{HC_ERR_STATE, "resolve_admitted_edges: more than 2^32 admitted edges"};
    const unsigned T = n < (1u << 14) ? 1u : std::max(1u, std::min<unsigned>((unsigned)ps.n_threads, 32u));
    static const bool timing = getenv("HC_STAGE_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tp = now();
    auto lap = [&](const char* what) {
        if (timing) {
            const double t = now();
            fprintf(stderr, "[hc stage] resolve: %s %.3f s\n", what, t - tp);
            tp = t;
        }
    };
    struct Item {
        uint64_t key;  // smaller vertex << 33 | larger vertex << 1 | (ori1 == ori2)
        uint32_t seq;
    };
    struct Tally {
        uint64_t inclusion = 0, dups = 0, slots = 0;
        bool wide = false;
    };
    std::vector<Tally> tally(T);
    std::vector<uint64_t> keys(n);
    run_workers(T, [&](unsigned t) {
        Tally mine;  // a local: neighbouring vector elements would share a cache line
        for (size_t i = n * t / T; i < n * (t + 1) / T; i++) {
            Edge& e = admitted[i];
            if (e.get_pos(1) == 0 && e.get_vertex(1) > e.get_vertex(2)) e.swap_reads();  // :443-448
            if (e.get_perc() == 100) mine.inclusion++;                                    // :449-451
            if (!EdgeSlotIndex::representable(e.get_vertex(1), e.get_vertex(2))) mine.wide = true;
            keys[i] = EdgeSlotIndex::key(e.get_vertex(1), e.get_vertex(2), e.get_ori(1) == e.get_ori(2));
        }
        tally[t] = mine;
    });
    lap("normalise + keys");
    for (const Tally& x : tally)
        if (x.wide) throw FatalError{HC_ERR_STATE, "resolve_admitted_edges: vertex ids beyond 2^31"};
    // Slots are independent of each other, so a worker takes every slot whose key hashes to it — no global order
    // is needed, only the sequence order inside a slot.
    std::vector<uint8_t> keep(n, 0);
    run_workers(T, [&](unsigned t) {
        uint64_t my_dups = 0, my_slots = 0;
        std::vector<Item> items;
        items.reserve(n / T + n / (4 * T) + 16);
        for (size_t i = 0; i < n; i++)
            if (T == 1 || (unsigned)((keys[i] * 0x9E3779B97F4A7C15ull) >> 40) % T == t) items.push_back(Item{keys[i], (uint32_t)i});
        std::sort(items.begin(), items.end(), [](const Item& x, const Item& y) { return x.key != y.key ? x.key < y.key : x.seq < y.seq; });
        const size_t m = items.size();
        for (size_t i = 0; i < m;) {
            size_t j = i + 1;
            while (j < m && items[j].key == items[i].key) j++;
            const Edge& first = admitted[items[i].seq];  // the only record that is inserted into an empty slot: :455-469
            if (ps.ignore_inclusions && first.get_perc() == 100 && first.get_mismatch_rate() < 0.000001 &&
                first.get_mismatch_rate() >= 0) {
                if (first.get_extra_pos(1) < 0) {
                    if (first.get_pos(1) == 0) __atomic_store_n(&g.inclusions[first.get_vertex(1)], (uint8_t)1, __ATOMIC_RELAXED);
                } else {
                    __atomic_store_n(&g.inclusions[first.get_vertex(2)], (uint8_t)1, __ATOMIC_RELAXED);
                }
            }
            uint32_t ex = items[i].seq;
            for (size_t k = i + 1; k < j; k++) {  // the later records of the slot, in sequence order
                my_dups++;
                const Edge& e = admitted[items[k].seq];
                const Edge& cur = admitted[ex];
                if (!(e.get_score() >= cur.get_score())) continue;                                      // :535-538
                if (e.get_score() == cur.get_score() && chain_keeps_existing(cur, e)) continue;        // :474-521
                ex = items[k].seq;                                                                      // :523-530
            }
            keep[ex] = 1;
            my_slots++;  // edges_added counts first insertions (one per slot), as the sequential loop does
            i = j;
        }
        tally[t].dups = my_dups;
        tally[t].slots = my_slots;
    });
    lap("slots: partition, sort, replay");
    uint64_t n_slots = 0;
    for (const Tally& x : tally) {
        c.inclusion_count += x.inclusion;
        c.dup_count += x.dups;
        n_slots += x.slots;
    }
    c.edges_added += n_slots;
    std::vector<uint32_t> survivors;  // in sequence order: the order of their own addEdge calls
    survivors.reserve(n_slots);
    for (size_t i = 0; i < n; i++)
        if (keep[i]) survivors.push_back((uint32_t)i);
    lap("survivor list");
    g.bulk_add_edges(admitted, survivors, (unsigned)ps.n_threads);
    lap("bulk fill");
}

}  // namespace hc
