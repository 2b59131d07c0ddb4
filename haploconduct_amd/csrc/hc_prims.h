// hc_prims.h — the device-wide primitives the stage's kernels are built on (hc_prims.hip): a stable LSD radix sort, an
// exclusive prefix sum, ordered selection and unique.  Hand-written for gfx950 (wave64 ballot ranking, LDS histograms);
// no library underneath.  Every entry point is asynchronous on `stream`, takes caller-owned scratch (`temp`, at least
// the matching *_temp_bytes) and handles n < 2^32.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/hcedge.h"

namespace hc {
namespace prims {

// Stable radix sort by bits [begin_bit, end_bit) of the key, 8 bits a pass, least significant first.  k_in / v_in are
// left as they are; k_out / v_out receive the result; in and out must not overlap.
size_t sort_temp_bytes(uint64_t n, size_t key_bytes, size_t val_bytes);
hipError_t sort_pairs(void* temp, size_t temp_bytes, const uint32_t* k_in, uint32_t* k_out, const uint32_t* v_in, uint32_t* v_out, uint64_t n,
                      int begin_bit, int end_bit, hipStream_t stream);
hipError_t sort_pairs(void* temp, size_t temp_bytes, const uint64_t* k_in, uint64_t* k_out, const uint32_t* v_in, uint32_t* v_out, uint64_t n,
                      int begin_bit, int end_bit, hipStream_t stream);
hipError_t sort_pairs(void* temp, size_t temp_bytes, const uint64_t* k_in, uint64_t* k_out, const uint64_t* v_in, uint64_t* v_out, uint64_t n,
                      int begin_bit, int end_bit, hipStream_t stream);
hipError_t sort_keys(void* temp, size_t temp_bytes, const uint64_t* k_in, uint64_t* k_out, uint64_t n, int begin_bit, int end_bit,
                     hipStream_t stream);

// out[i] = in[0] + ... + in[i - 1]; in == out is allowed.
size_t scan_temp_bytes(uint64_t n, size_t elem_bytes);
hipError_t exclusive_sum(void* temp, size_t temp_bytes, const uint32_t* in, uint32_t* out, uint64_t n, hipStream_t stream);
hipError_t exclusive_sum(void* temp, size_t temp_bytes, const uint64_t* in, uint64_t* out, uint64_t n, hipStream_t stream);

// Ordered selection: the indices i (ascending) whose flag / record passes, *count = how many.
size_t select_temp_bytes(uint64_t n);
hipError_t select_flagged(void* temp, size_t temp_bytes, const uint8_t* flags, uint64_t n, uint32_t* idx_out, unsigned long long* count,
                          hipStream_t stream);
hipError_t select_not_dropped(void* temp, size_t temp_bytes, const hc_result_rec* res, uint64_t n, uint32_t* idx_out, unsigned long long* count,
                              hipStream_t stream);  // class != HC_CLS_DROP
// The first element of every run of equal neighbours (a sorted array's distinct values), in order.
hipError_t unique(void* temp, size_t temp_bytes, const uint64_t* in, uint64_t* out, unsigned long long* count, uint64_t n, hipStream_t stream);

}  // namespace prims
}  // namespace hc
