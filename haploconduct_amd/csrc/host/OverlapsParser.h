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
    const ProgramSettings& m_ps;
    const FastqStorage& m_fastq;
    bool m_open = false;
    int m_fd = -1;
    const char* m_data = nullptr;
    size_t m_size = 0, m_pos = 0;
    uint64_t m_line_no = 0;
    bool m_done = false;
};

}  // namespace hc
