// Overlap.h — one line of the 13-column overlaps file (reference src/Overlap.h:20-238).
// POD instead of 5 std::strings: the single-character fields are chars.
#pragma once
#include <string>

#include "Types.h"

namespace hc {

class Overlap {
public:
    Overlap() = default;
    // From the 13 tokens of a line (src/Overlap.h:39-73).  Throws FatalError where the
    // reference exits / asserts (negative pos/len, perc outside 0..100, bad ori/type/ord).
    static Overlap from_fields(const char* const field[13], const size_t len[13]);
    // One pass over a line of the plain form — 13 fields, single tabs, decimal numbers (or "-"), valid one-character
    // fields — filling `o` with exactly what the tokeniser + from_fields produce for it.  Returns false for every
    // other line (malformed, padded, out of range, ...): those take the general path, which also owns all errors.
    static bool from_plain_line(const char* line, size_t n, Overlap& o);

    read_id_t get_id(int i) const { return i == 1 ? m_id1 : m_id2; }
    int get_pos(int i) const { return (int)(i == 1 ? m_pos1 : m_pos2); }
    std::string get_ord() const { return std::string(1, m_ord); }
    std::string get_ori(int i) const { return std::string(1, i == 1 ? m_ori1 : m_ori2); }
    unsigned int get_perc() const {                       // src/Overlap.h:203-210
        if (m_perc2 > 0) return (unsigned int)(0.5 * (m_perc1 + m_perc2));
        return m_perc1;
    }
    unsigned int get_len(int i) const { return i == 1 ? m_len1 : m_len2; }
    std::string get_type(int i) const { return std::string(1, i == 1 ? m_type1 : m_type2); }
    std::string get_overlap_line() const;                 // src/Overlap.h:234-237
    size_t write_line(char* buf) const;                   // same text, no allocation (buf >= 192 bytes)

    read_id_t m_id1 = 0, m_id2 = 0;
    unsigned int m_pos1 = 0, m_pos2 = 0, m_perc1 = 0, m_perc2 = 0, m_len1 = 0, m_len2 = 0;
    char m_ord = '-', m_ori1 = '+', m_ori2 = '+', m_type1 = 's', m_type2 = 's';
};

}  // namespace hc
