// OverlapsParser.h — the tokenizer / validator / prefilter half of construct_edges
// (reference src/EdgeCalculator.cpp:581-635) over an mmap'ed file: no per-line allocations,
// same acceptance rules.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../../include/hcedge.h"
#include "FastqStorage.h"
#include "Overlap.h"
#include "Types.h"

namespace hc {

struct ParsedOverlap {
    Overlap line;        // what get_overlap_line() re-serialises
    hc_overlap_rec rec;  // what the device scores (read ids resolved to m_read_vec indices)
};

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
    OverlapsParser(const std::string& path, const ProgramSettings& ps, const FastqStorage& fastq);
    ~OverlapsParser();
    bool is_open() const { return m_open; }
    // Fills `batch` with up to `max_batch` candidates that pass the prefilter (:612-635), in file
    // order; lines that fail the length/type test are appended to `rejected` (written to
    // nonedge_overlaps.txt at the end, :654-660).  Returns false when the input is exhausted
    // (or max_overlaps lines have been read, :581).
    bool next_batch(std::vector<ParsedOverlap>& batch, size_t max_batch, std::vector<Overlap>& rejected,
                    ParseCounters& c, bool print_malformed);

private:
    struct Segment;
    void parse_segment(Segment& seg) const;

    const ProgramSettings& m_ps;
    const FastqStorage& m_fastq;
    IdIndex m_ids;
    unsigned int m_threads = 1;
    bool m_open = false;
    int m_fd = -1;
    const char* m_data = nullptr;
    size_t m_size = 0, m_pos = 0;
    uint64_t m_line_no = 0;
    bool m_done = false;
};

}  // namespace hc
