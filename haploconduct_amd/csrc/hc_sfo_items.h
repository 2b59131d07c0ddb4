// hc_sfo_items.h — what the device form of the SFO ingest's sort hands back to the host (hc_sfo_kernels.hip,
// hc_api_finder.cpp: hc_found_to_overlaps).  Plain C++: host/Sfo2Overlaps.cpp builds without HIP.
#ifndef HC_SFO_ITEMS_H_
#define HC_SFO_ITEMS_H_
#include <cstdint>
#include <memory>
#include <string>

namespace hc {

// One SFO record after the flip of scripts/sfo2overlaps.py:112-122 (the read with the smaller ORIGINAL id first; the
// original ids follow from the SFO ids and --num_singles / --num_pairs), in the order of the script's
// `sort -k1,1n -k2,2n -k3,3n -k4,4n` with the whole line as last resort.
struct SfoFlipped {
    uint32_t s0, s1;   // SFO ids
    int32_t oha, ohb;
    uint32_t ola, olb, k;
    uint32_t inverted;  // 0 = "N", 1 = "I"
};
static_assert(sizeof(SfoFlipped) == 32, "SfoFlipped is 32 bytes");

// The matching half of the ingest over records in that order, fed in consecutive chunks (hc_found_to_overlaps copies
// them from the device through a ring of page-locked buffers while earlier chunks are matched).  A chunk is consumed
// before feed() returns.
class SfoSortedMatcher {
public:
    SfoSortedMatcher(long num_singles, long num_pairs);
    ~SfoSortedMatcher();
    void feed(const SfoFlipped* recs, uint64_t n);
    std::string finish(uint64_t& n_lines);  // the overlaps file's text

private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
};

}  // namespace hc
#endif
