// FastqStorage.h — reads the FASTQ inputs and owns the reads (reference src/FastqStorage.h:25-107,
// src/FastqStorage.cpp:38-235).  Same public members the hot path uses: m_read_vec (singles
// first, then pairs), m_ID_to_index, m_readcount_single/_paired.  Storage is flat: one byte
// arena for bases, one for qualities, offsets per sequence — exactly what hc_set_reads takes.
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <new>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "DefaultInit.h"
#include "Read.h"
#include "Types.h"

namespace hc {

// The flat base / quality arrays: bytes that resize() adds are left uninitialised (DefaultInit.h)
using ByteVec = std::vector<uint8_t, DefaultInitAllocator<uint8_t>>;

class FastqStorage {
public:
    explicit FastqStorage(const ProgramSettings& ps);

    std::vector<Read> m_singles_vec;
    std::vector<Read> m_paired_vec;
    std::vector<Read*> m_read_vec;                       // singles first, then pairs
    std::map<read_id_t, unsigned int> m_ID_to_index;    // first occurrence of an id wins (std::map::insert)
    unsigned int m_readcount_single = 0;
    unsigned int m_readcount_paired = 0;
    unsigned int m_largest_read_id = 0;

    Read* get_read(read_id_t ID);                        // src/FastqStorage.cpp:33-36
    unsigned int get_readcount() const { return (unsigned int)m_read_vec.size(); }

    // flat layout (hc_set_reads arguments)
    const ByteVec& bases() const { return m_bases; }
    const ByteVec& quals() const { return m_quals; }
    const std::vector<uint64_t>& seq_off() const { return m_seq_off; }
    const std::vector<uint32_t>& read_first_seq() const { return m_first; }
    // sequence index of mate i (0 = single, 1 = /1, 2 = /2) of read `index`
    uint32_t seq_index(unsigned int index, int i) const { return m_first[index] + (i == 2 ? 1u : 0u); }
    uint32_t seq_len(uint32_t q) const { return (uint32_t)(m_seq_off[q + 1] - m_seq_off[q]); }

private:
    void read_new_ids(const std::string& path);
    void read_singles(const std::string& path, unsigned long max_reads);
    void read_pairs(const std::string& p1, const std::string& p2, unsigned long max_reads);
    void push_sequence(const char* s, size_t ns, const char* q, size_t nq, bool upper);
    // the same records from read-only mappings of the files, on several threads (regular files of some size); false: not
    // applicable, nothing was touched, the sequential readers take over
    bool read_mapped(const std::string& p1, const std::string* p2, unsigned long max_reads, unsigned threads);
    unsigned m_threads = 1;
    read_id_t resolve_id(const std::string& token) const;
    read_id_t resolve_id(const char* token, size_t n) const;

    std::map<std::string, std::string> m_new_readIDs;    // fastq id -> overlaps-file id (--IDs)
    bool m_have_new_ids = false;
    ByteVec m_bases, m_quals;
    std::vector<uint64_t> m_seq_off{0};
    std::vector<uint32_t> m_first{0};
};

}  // namespace hc
