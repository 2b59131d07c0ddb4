// hc_device.h — device-side data layout shared by the kernels and the C-ABI glue.
//
// Read store in HBM (built once per hc_set_reads, see DESIGN.md "Data layout"):
//   every stored sequence q (a single read, or mate /1 or /2 of a pair) owns two
//   SLOTS of symbols: forward at seq_off[q], reverse-complement at
//   seq_off[q] + slot_stride(len).  One symbol per base:
//       sym = (qidx << 3) | code
//   qidx = index of the quality byte in the store's dense quality alphabet
//   (K distinct bytes in the read set, K <= 32 -> uint8 symbols, else uint16);
//   code = 0..3 for A,C,G,T (complement = 3 - code), 4 = N,
//          6 = quality byte outside [33,127]  (reference asserts, EdgeCalculator.cpp:61,97-98)
//          7 = base outside ACGTN             (reference asserts, EdgeCalculator.cpp:29-30)
//   Slots are padded with zero symbols to a multiple of 16 bytes plus 16 bytes,
//   so chunked loads may over-read safely.
#pragma once
#include <stdint.h>

namespace hc {

constexpr uint32_t kCodeN = 4;
constexpr uint32_t kCodeBadQual = 6;
constexpr uint32_t kCodeBadBase = 7;
constexpr uint32_t kSeqFlagBadBase = 1u;  // sequence holds a base outside ACGTN: build_rev_comp exits (Types.h:124-127)

// slot stride in SYMBOLS for a sequence of `len` symbols of `symbytes` bytes each
__host__ __device__ inline uint64_t slot_stride(uint32_t len, uint32_t symbytes) {
    uint64_t bytes = (uint64_t)len * symbytes;
    bytes = ((bytes + 15) & ~(uint64_t)15) + 16;
    return bytes / symbytes;
}

struct StoreView {
    const void* sym;                 // uint8_t* or uint16_t*
    const uint64_t* seq_off;         // [n_seq]  forward-slot offset, in symbols
    const uint32_t* seq_len;         // [n_seq]
    const uint8_t* seq_flags;        // [n_seq]
    const uint32_t* read_first_seq;  // [n_reads + 1]
    uint32_t n_reads;
    uint32_t n_seq;
    uint32_t K;                      // quality alphabet size
    uint32_t symbytes;               // 1 or 2
};

// x-space image of a score threshold T: exp(x) > T  <=>  x > hi ; x <= lo => exp(x) <= T;
// lo < x <= hi is the guard band the host libm decides (normally empty).
struct Band {
    double lo, hi;
};

struct ScoreParams {
    Band edge, ov;
    double merge_contigs;
    uint32_t min_read_len;
    uint32_t flags;
};

}  // namespace hc
