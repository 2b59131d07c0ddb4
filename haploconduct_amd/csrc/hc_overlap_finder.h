// hc_overlap_finder.h — device steps of the overlap finder (hc_overlap_finder.hip), called from hc_api.cpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/hcedge.h"

namespace hc {

// One stored sequence: forward slot (symbols), length, and its id in the SFO numbering
// (singles, then all /1 mates, then all /2 mates: the s_p1_p2.fasta the pipelines feed to rust-overlaps).
struct SeqRef {
    uint64_t off;       // forward slot
    uint32_t len;
    uint32_t sfo_id;
    uint32_t rc_delta;  // reverse-complement slot = off + rc_delta (hc_device.h: store layout)
    uint32_t pad;
};

hipError_t finder_index(const void* sym, uint32_t symbytes, bool wide, const SeqRef* seqs, const uint64_t* pos_start, uint32_t n_seq,
                        uint32_t k, uint64_t* keys, uint64_t* vals, hipStream_t stream);
hipError_t finder_seeds(const void* sym, uint32_t symbytes, bool wide, const SeqRef* seqs, const uint64_t* seed_start, uint32_t n_seq,
                        uint32_t k, uint32_t s, uint32_t n_ori, const uint64_t* keys, uint64_t n_keys, uint64_t* seed_lo,
                        uint64_t* seed_cnt, hipStream_t stream);
hipError_t finder_count_valid(const SeqRef* seqs, const uint2* idlen /* (sfo id, length) per sequence */, const uint64_t* seed_start, uint32_t n_seq, uint32_t k, uint32_t s, uint32_t n_ori,
                              const uint64_t* vals, const uint64_t* seed_lo, const uint64_t* seed_cnt, uint32_t min_overlap, uint32_t flags,
                              uint64_t* seed_valid, hipStream_t stream);
hipError_t finder_expand(const SeqRef* seqs, const uint2* idlen, const uint64_t* seed_start, uint32_t q_begin, uint32_t q_end, uint64_t out_base, uint32_t k,
                         uint32_t s, uint32_t n_ori, const uint64_t* vals, const uint64_t* seed_lo, const uint64_t* seed_cnt,
                         const uint64_t* seed_out, uint32_t min_overlap, uint32_t flags, uint64_t* out_keys, hipStream_t stream);
hipError_t finder_verify(const void* sym, uint32_t symbytes, bool wide, const SeqRef* by_sfo, const uint64_t* keys, uint64_t n,
                         double err_rate, uint32_t min_overlap, uint32_t flags, uint32_t* kout, uint32_t* flag, hipStream_t stream);
hipError_t finder_emit(const SeqRef* by_sfo, const uint64_t* keys, const uint32_t* kout, const uint32_t* flag, const uint32_t* pos, uint64_t n,
                       hc_sfo_rec* out, hipStream_t stream);
hipError_t finder_boundaries(const uint64_t* off, const uint64_t* seed_start, uint32_t n, uint64_t* out, hipStream_t stream);
hipError_t finder_rekey(const hc_sfo_rec* recs, uint64_t n, uint64_t* keys, uint64_t* idx, hipStream_t stream);
hipError_t finder_gather(const hc_sfo_rec* recs, const uint64_t* idx, uint64_t n, hc_sfo_rec* out, hipStream_t stream);
hipError_t finder_sort_pairs(void* temp, size_t& temp_bytes, const uint64_t* k_in, uint64_t* k_out, const uint64_t* v_in, uint64_t* v_out,
                             uint64_t n, int end_bit, hipStream_t stream);
hipError_t finder_sort_keys(void* temp, size_t& temp_bytes, const uint64_t* k_in, uint64_t* k_out, uint64_t n, hipStream_t stream);
hipError_t finder_scan(void* temp, size_t& temp_bytes, const uint64_t* in, uint64_t* out, uint64_t n, hipStream_t stream);
hipError_t finder_unique(void* temp, size_t& temp_bytes, const uint64_t* in, uint64_t* out, unsigned long long* n_out, uint64_t n,
                         hipStream_t stream);
hipError_t finder_scan32(void* temp, size_t& temp_bytes, const uint32_t* in, uint32_t* out, uint64_t n, hipStream_t stream);

}  // namespace hc
