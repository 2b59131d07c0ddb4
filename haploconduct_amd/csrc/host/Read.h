// Read.h — one read (single-end or a pair) of the FastqStorage (reference src/Read.h:22-336).
// The sequences live in FastqStorage's flat arenas (the layout hc_set_reads takes); a Read is
// a lightweight handle with the reference's getter names.
#pragma once
#include <string>

#include "Types.h"

namespace hc {

class FastqStorage;

class Read {
public:
    Read(const FastqStorage* store, unsigned int index, bool is_paired, read_id_t id)
        : m_store(store), m_index(index), m_is_paired(is_paired), m_read_id(id) {}

    bool is_paired() const { return m_is_paired; }                 // src/Read.h:136
    read_id_t get_read_id() const { return m_read_id; }            // src/Read.h:107
    unsigned int get_index() const { return m_index; }             // position in m_read_vec
    void set_vertex_id(bool normal, node_id_t id) {                // src/Read.h:92-101
        if (normal) { m_vertex_N = id; m_N_set = true; } else { m_vertex_R = id; m_R_set = true; }
    }
    bool has_vertex_id(bool normal) const { return normal ? m_N_set : m_R_set; }
    node_id_t get_vertex_id(bool normal) const;                    // src/Read.h:111-120 (asserts the id was set)
    // i = 0 for a single-end read, 1 / 2 for the mates of a pair (src/Read.h:144-201)
    std::string get_seq(int i) const;
    std::string get_phred(int i) const;
    std::string get_rev_comp(int i) const;
    std::string get_rev_phred(int i) const;
    unsigned int get_len() const;                                  // src/Read.h:203-212
    unsigned int get_seq_len(int i) const;

private:
    const FastqStorage* m_store;
    unsigned int m_index;
    bool m_is_paired;
    read_id_t m_read_id;
    node_id_t m_vertex_N = 0, m_vertex_R = 0;
    bool m_N_set = false, m_R_set = false;
};

}  // namespace hc
