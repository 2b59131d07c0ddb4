// OverlapsParser.h — the tokenizer / validator / prefilter half of construct_edges
// (reference src/EdgeCalculator.cpp:581-635) over an mmap'ed file: no per-line allocations,
// same acceptance rules.
#pragma once
#include <cstdint>
#include <string>
#include <memory>
#include <mutex>
#include <vector>

#include "../../../include/hcedge.h"
#include "FastqStorage.h"
#include "Overlap.h"
#include "Types.h"
#include "WorkerPool.h"

namespace hc {

// One block of candidates, structure of arrays: lines[i] is what get_overlap_line() re-serialises, recs[i] what
// the device scores — the compact record (hc_cand_rec, 16 bytes: read ids resolved to m_read_vec indices, positions,
// orientations, ord); everything else of the line stays in lines[i] on the host.  The storage only grows (a block reuses the
// elements of the one before it), and the record array can live in memory the caller provides — page-locked
// memory in the stage, so that the parser's output is what the device reads, without a copy in between.
struct ParsedBatch {
    struct RecStorage {  // optional provider of the record array
        void* ctx = nullptr;
        hc_cand_rec* (*alloc)(void* ctx, size_t n) = nullptr;
        void (*release)(void* ctx, hc_cand_rec* p) = nullptr;
    };
    ParsedBatch() = default;
    explicit ParsedBatch(const RecStorage& st) : storage(st) {}
    ParsedBatch(const ParsedBatch&) = delete;
    ParsedBatch& operator=(const ParsedBatch&) = delete;
    ~ParsedBatch() {
        if (recs && storage.release) storage.release(storage.ctx, recs);
    }
    void ensure(size_t room) {  // contents are not kept
        if (lines.size() < room) lines.resize(room);
        if (cap >= room) return;
        const size_t want = room + room / 8;
        if (storage.alloc) {
            if (recs) storage.release(storage.ctx, recs);
            recs = storage.alloc(storage.ctx, want);
        } else {
            own.resize(want);
            recs = own.data();
        }
        cap = want;
    }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    void clear() { n = 0; }

    std::vector<Overlap> lines;
    hc_cand_rec* recs = nullptr;
    size_t n = 0;  // the first n lines / recs are this block

private:
    RecStorage storage;
    std::vector<hc_cand_rec> own;
    size_t cap = 0;
};

// the compact record of a line (read ids already resolved)
inline hc_cand_rec make_cand(const Overlap& o, uint32_t read1, uint32_t read2) {
    hc_cand_rec r;
    r.read1 = read1;
    r.read2 = read2;
    const uint32_t p1 = o.m_pos1 < HC_CAND_POS_MASK ? o.m_pos1 : HC_CAND_POS_MASK;
    const uint32_t p2 = o.m_pos2 < HC_CAND_POS_MASK ? o.m_pos2 : HC_CAND_POS_MASK;
    const uint32_t oc = o.m_ord == '-' ? 0u : (o.m_ord == '1' ? 1u : 2u);  // from_fields admits nothing else
    r.pos1_bits = p1 | (o.m_ori1 == '+' ? 1u << 28 : 0u) | (o.m_ori2 == '+' ? 1u << 29 : 0u) | (oc << 30);
    r.pos2_bits = p2;
    return r;
}
// the full record (include/hcedge.h) of a line, as the 32-byte ABI describes it
inline hc_overlap_rec make_overlap_rec(const Overlap& o, uint32_t read1, uint32_t read2) {
    hc_overlap_rec r;
    r.read1 = read1;
    r.read2 = read2;
    r.pos1 = o.m_pos1;
    r.pos2 = o.m_pos2;
    r.ori1 = o.m_ori1 == '+';
    r.ori2 = o.m_ori2 == '+';
    r.ord = (uint8_t)o.m_ord;
    r.flags = (uint8_t)((o.m_type1 == 'p') | ((o.m_type2 == 'p') << 1));
    r.len1 = o.m_len1;
    r.len2 = o.m_len2;
    r.perc = o.get_perc();
    return r;
}

struct ParseCounters {
    uint64_t lines_read = 0, malformed = 0, self_overlaps = 0, prefilter_rejected = 0, silently_dropped = 0;
};

// Splits one line exactly like src/EdgeCalculator.cpp:584-597: trim "\t ", then split on '\t'
// (or on "\t " with compression when allow_spaces).  Returns the number of tokens; tokens
// beyond `max_fields` are counted but not stored.  The line is not modified.
int split_overlap_line(const char* s, size_t n, bool allow_spaces, const char* field[], size_t len[], int max_fields);

// id -> m_read_vec index with the semantics of FastqStorage::m_ID_to_index (first occurrence of an id
// wins, FastqStorage.h:88-97) but O(1): a direct table when the ids are dense, else open addressing.
class IdIndex {
public:
    explicit IdIndex(const FastqStorage& fastq);
    // returns false when the id is unknown (std::map::at would throw, EdgeCalculator.cpp:170-171)
    bool find(read_id_t id, uint32_t& index) const {
        if (m_direct) {
            if (id >= m_table.size() || m_table[id] == kNone) return false;
            index = m_table[id];
            return true;
        }
        uint64_t h = (id * 0x9E3779B97F4A7C15ull) >> m_shift;
        for (;;) {
            const uint32_t v = m_table[h];
            if (v == kNone) return false;
            if (m_keys[h] == id) { index = v; return true; }
            h = (h + 1) & (m_table.size() - 1);
        }
    }

private:
    static constexpr uint32_t kNone = 0xFFFFFFFFu;
    bool m_direct = true;
    int m_shift = 0;
    std::vector<uint32_t> m_table;
    std::vector<read_id_t> m_keys;
};

class OverlapsParser {
public:
    // shared_pool: worker threads of the caller (at least ps.n_threads - 1), used instead of starting and joining a pool per file
    OverlapsParser(const std::string& path, const ProgramSettings& ps, const FastqStorage& fastq, WorkerPool* shared_pool = nullptr);
    // The same over the overlaps file's text in memory (kept alive by the parser): the reads -> graph call, whose
    // overlaps never become a file.
    OverlapsParser(std::shared_ptr<const std::string> text, const ProgramSettings& ps, const FastqStorage& fastq, WorkerPool* shared_pool = nullptr);
    ~OverlapsParser();
    bool is_open() const { return m_open; }
    // Fills `batch` with up to `max_batch` candidates that pass the prefilter (:612-635), in file
    // order; lines that fail the length/type test are appended to `rejected` (written to
    // nonedge_overlaps.txt at the end, :654-660).  Returns false when the input is exhausted
    // (or max_overlaps lines have been read, :581).
    bool next_batch(ParsedBatch& batch, size_t max_batch, std::vector<Overlap>& rejected, ParseCounters& c, bool print_malformed);
    // The same for the bytes [begin, end) of the file (begin at a line start, end behind a newline or at the end of the
    // file), whose first line is line number first_line of the file; returns the number of lines in the range.  The
    // device-parsing stage hands the blocks its kernels do not read (a line that is not plain) to this.
    uint64_t parse_range(size_t begin, size_t end, uint64_t first_line, ParsedBatch& batch, std::vector<Overlap>& rejected, ParseCounters& c,
                         bool print_malformed);
    // The device-parsing stage reads the file itself: its size, the bytes [begin, end) copied to dst on the parser's
    // threads (pread: no page of the mapping is touched) with the number of newlines among them, the end of the line
    // that holds byte `at` (offset behind its newline, or the size of the file), and the id table.
    // One line by itself (the per-line fallback of the device-parsing stage): what construct_edges does with it, :584-635
    enum class LineKind { Malformed, Self, Silent, Rejected, Pass };
    LineKind classify_line(const char* line, size_t n, Overlap& o, hc_cand_rec& rec) const;
    size_t size() const { return m_size; }
    const char* data() const { return m_data; }  // the file's mapping (PROT_READ, MAP_PRIVATE)
    void copy_range(char* dst, size_t begin, size_t end, uint64_t& newlines, bool count_newlines = true) const;  // false: dst is never read
    size_t line_end_at(size_t at) const;
    const IdIndex& ids() const { return m_ids; }

private:
    struct Segment;
    void parse_segment(Segment& seg) const;
    std::unique_ptr<WorkerPool> m_own_pool;  // the parser's worker threads, started once — unless the caller lends its pool
    WorkerPool* m_pool = nullptr;
    mutable std::mutex m_pool_mu;  // copy_range and parse_range may be called from two threads: one phase of the pool at a time
    struct Scratch {  // where one segment parses to before its place in the block is known; kept between blocks
        std::vector<Overlap> lines;
        std::vector<hc_cand_rec> recs;
    };
    std::vector<Scratch> m_scratch;

    const ProgramSettings& m_ps;
    IdIndex m_ids;
    unsigned int m_threads = 1;
    bool m_open = false;
    int m_fd = -1;
    std::shared_ptr<const std::string> m_text;  // the memory-backed form: m_data points into it, there is no file
    const char* m_data = nullptr;
    size_t m_size = 0, m_pos = 0;
    uint64_t m_line_no = 0;
    bool m_done = false;
};

}  // namespace hc
