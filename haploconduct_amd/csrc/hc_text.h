// hc_text.h — launch interface of hc_text_kernels.hip (the overlaps file read on the device).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hcedge.h"

namespace hc {

struct IdTable {  // FastqStorage::m_ID_to_index on the device: direct table when the ids are dense, else open addressing
    const uint32_t* table;  // 0xFFFFFFFF = no such id
    const uint64_t* keys;   // open addressing only
    uint64_t size;          // entries (a power of two for open addressing)
    int shift;              // open addressing: hash >> shift
    int direct;
};

struct TextParams {
    uint64_t n_bytes;
    uint64_t first_line_no;  // number of the block's first line in the file ...
    const unsigned long long* first_line_ptr;  // ... plus *first_line_ptr when set (a line chain, hc_linechain)
    uint64_t max_overlaps;   // --max_ov: lines numbered >= this are not read (EdgeCalculator.cpp:581)
    uint32_t max_lines;      // room in the line arrays; more newlines than that: the block goes to the host
    uint32_t min_overlap_len, min_overlap_perc, relax_pe;
    uint32_t reject_cap;
    uint32_t nonplain_cap;   // hc_textblock_list_nonplain: room in the list of lines that are not plain (0: they are only counted)
};

// counters of one block (device memory, 16 x u64)
enum {
    kTextRead = 0,      // lines read (below --max_ov)
    kTextNonPlain = 1,  // lines that are not of the plain form: the host redoes the block
    kTextSelf = 2,      // id1 == id2, :605-607
    kTextSilent = 3,    // percentage below --min_overlap_perc: dropped without a trace
    kTextReject = 4,    // failed the length / type test: kept for nonedge_overlaps.txt, :633-635
    kTextPass = 5,      // candidates for process_overlaps
    kTextUnknownId = 6, // a passing line names a read that is not in the FASTQ input: the host reproduces the failure
    kTextRejectSlots = 7,  // slots handed out in the reject buffer
    kTextLines = 8,     // lines in the block (newlines + a last line without one)
    kTextOverflow = 9,  // more lines than max_lines
    kTextRows = 10,     // rows the scoring kernel appended
    kTextNonPlainSlots = 11,  // slots handed out in the list of lines that are not plain (hc_textblock_list_nonplain)
    kTextCounters = 16
};

// the line starts of a block in three launches: newlines per 4 KiB tile; the one-workgroup scan, which also ZEROES the block's
// counters before it sets kTextLines / kTextOverflow and, with a chain, writes *lines_before_next = *lines_before + lines; the starts
hipError_t launch_text_count(const char* text, uint64_t n_bytes, uint32_t* tile_cnt, hipStream_t s);
hipError_t launch_text_scan(const char* text, uint64_t n_bytes, const uint32_t* tile_cnt, uint32_t* tile_off, uint32_t max_lines, uint32_t* line_start,
                            unsigned long long* counters, const unsigned long long* lines_before, unsigned long long* lines_before_next,
                            hipStream_t s);
hipError_t launch_text_line_starts(const char* text, uint64_t n_bytes, const uint32_t* tile_off, uint32_t max_lines, uint32_t* line_start, hipStream_t s);
// (counters[0..6] are the sums of the workgroups' tallies: launch_kept_rows_flushed adds them up)
hipError_t launch_text_parse(const TextParams& prm, const char* text, const uint32_t* line_start, const IdTable& ids, hc_cand_rec* cands,
                             hc_line_rec* lines, hc_text_reject* rejects, unsigned long long* counters, uint32_t* tally /* [(max_lines + 255) / 256][8] */,
                             hc_text_nonplain* nonplain /* [prm.nonplain_cap], mapped host memory, or nullptr */, hipStream_t s);

// parsed lines instead of text (the device-resident stage a): the counters as the line scan leaves them (kTextLines = n_lines, or
// kTextOverflow when there is no room for them), then the parse kernel's second half on src[i]
hipError_t launch_lines_accept(const TextParams& prm, const hc_line_rec* src, uint32_t n_lines, const IdTable& ids, hc_cand_rec* cands, hc_line_rec* lines,
                               hc_text_reject* rejects, unsigned long long* counters, uint32_t* tally, hipStream_t s);

}  // namespace hc
