// hc_sfo_device.h — the SFO ingest's flip + sort on the device: the records of hc_find_overlaps are already there, and the
// script's `sort` (four numeric keys, then the rest of the line bytewise, LC_ALL=C) is a radix sort over a 192-bit key
// once every later column is mapped to the integer that orders like its decimal text.
#ifndef HC_SFO_DEVICE_H_
#define HC_SFO_DEVICE_H_
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/hcedge.h"
#include "hc_sfo_items.h"

namespace hc {

enum : unsigned long long {
    kSfoStatusId = 1,    // an SFO id outside --num_singles / --num_pairs: the host path reports it
    kSfoStatusRange = 2  // a number the keys do not hold (|overhang| or overlap length >= 10^7, K >= 10^4): the host path sorts
};
hipError_t sfo_flip(const hc_sfo_rec* in, uint64_t n, uint64_t ns, uint64_t np, SfoFlipped* out, uint64_t* k0, uint64_t* k1, uint64_t* k2,
                    uint32_t* iota, unsigned long long* status, hipStream_t s);
hipError_t sfo_gather(const SfoFlipped* in, const uint32_t* perm, uint64_t n, SfoFlipped* out, hipStream_t s);
// Of the sorted records, those the script's matching can see anything of (hc_sfo_kernels.hip): grouped[i] = a line that joins a
// group of one pair of reads, keep[i] = an output line by itself; then, for the grouped ones in order (idx), keep = its group has
// two lines or more, or it closes such a group; then the kept ones in order.
hipError_t sfo_classify(const SfoFlipped* sorted, uint64_t n, uint64_t ns, uint64_t np, uint8_t* grouped, uint8_t* keep, hipStream_t s);
hipError_t sfo_groups(const SfoFlipped* sorted, const uint32_t* idx, uint64_t m, uint64_t ns, uint64_t np, uint8_t* keep, hipStream_t s);
hipError_t sfo_gather_kept(const SfoFlipped* sorted, const uint32_t* idx, uint64_t k, SfoFlipped* out, hipStream_t s);

// The script's matching on the device (round 6; hc_sfo_kernels.hip): the overlap lines as records, in the script's order.
// kSfoStatusMatch (4) in *status: an assert / division by zero of the script's matching — the caller lets the host's matcher raise it.
// the SFO file's text -> records (hc_sfo_kernels.hip): one lane per line of a chunk whose line starts / count the overlaps file's kernels left
// (hc_text.h); bit 8 of *status: a line that is not canonical
hipError_t sfo_parse_text(const char* text, const uint32_t* line_start, uint32_t max_lines, const unsigned long long* counters,
                          const unsigned long long* lines_before, hc_sfo_rec* out, uint64_t out_cap, unsigned long long* status, hipStream_t s);
hipError_t sfo_group_starts(const SfoFlipped* sorted, const uint32_t* idx, uint64_t m, uint64_t ns, uint64_t np, uint8_t* start, hipStream_t s);
hipError_t sfo_match_groups(bool write, const SfoFlipped* sorted, const uint32_t* idx, const uint32_t* starts, uint64_t G, uint64_t ns, uint64_t np,
                            uint32_t* emit, const uint32_t* off, hc_line_rec* lines, unsigned long long* status, hipStream_t s);
hipError_t sfo_single_lines(bool write, const SfoFlipped* sorted, const uint8_t* keep, uint64_t n, uint64_t ns, uint64_t np, uint32_t* emit,
                            const uint32_t* off, hc_line_rec* lines, unsigned long long* status, hipStream_t s);
hipError_t sfo_uniq_lines(const hc_line_rec* lines, uint64_t n, uint8_t* keep, unsigned long long* n_dup, hipStream_t s);
hipError_t sfo_gather_lines(const hc_line_rec* in, const uint32_t* idx, uint64_t k, hc_line_rec* out, hipStream_t s);

}  // namespace hc
#endif
