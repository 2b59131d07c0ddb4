// hc_device.h — device-side data layout shared by the kernels and the C-ABI glue.
//
// Read store in HBM (built once per hc_set_reads, see DESIGN.md "Data layout"):
//   every stored sequence q (a single read, or mate /1 or /2 of a pair) owns two
//   SLOTS of symbols, forward and reverse-complement, each slot_stride(len) symbols long.  A single read:
//   [fwd][rc].  A pair: [/1 fwd][/2 fwd][/1 rc][/2 rc] — the two windows a paired candidate reads of one read are the
//   same orientation of both mates, and next to each other they share a 128-byte line more often (3.0 instead of 3.3
//   lines per candidate on 2 x 150 bp reads; every miss moves a whole line).  rc slot = fwd slot + rc_delta
//   (ReadDesc / SeqRef), rc_delta = the slot(s) in between.
//   One symbol per base:
//       sym = (qidx << 3) | code
//   code = 0..3 for A,C,G,T (complement = 3 - code), 4 = N,
//          6 = quality byte outside [33,127]  (reference asserts, EdgeCalculator.cpp:61,97-98)
//          7 = base outside ACGTN             (reference asserts, EdgeCalculator.cpp:29-30)
//   qidx = index of the quality byte in the store's dense quality alphabet (K distinct
//          bytes in the read set), EXCEPT: an N base gets qidx = K (the all-zero row/column
//          of the log table: the position adds 0.0, EdgeCalculator.cpp:35-39,122-124) and an
//          invalid symbol gets qidx = K+1 (the NaN row: poisons the sum, the lane then
//          re-scans the overlap position by position to report exactly what the reference
//          would do).  Table dimension Kp = K + 2.
//   K <= 30 -> uint8 symbols as above (5-bit qidx, 3-bit code); with K <= 6 (3-bit qidx) bits 6-7 of the symbol repeat
//          the low two bits of qidx (they are address bits of the K <= 6 log table as they stand, see below);
//   31 <= K <= 48 -> "wide" uint8 symbols  sym = (qidx << 2) | base2  (6-bit qidx, A,C,G,T = 0..3): the quality values take 48 of the
//          indices 16..63 (dealt by frequency, kWideRankLabel below), the reserved indices are 0 = N, 1 = invalid quality,
//          2 = invalid base — recognisable by their two CLEAR top bits: (sym << 1 | sym) has bit 7 set for a base, one VALU op;
//   49 <= K <= 60 -> the same symbols with the quality values in indices 4..63 (kWide7RankLabel): a base has one of its top FOUR bits
//          set, two VALU ops per stream — one byte per position where round 5 spent two;
//   K > 60 -> uint16 symbols  (qidx << 3) | code.
//   Slots are padded with N symbols to a multiple of 16 bytes plus 32 bytes, so chunked
//   (16-symbol) loads may over-read safely.
#pragma once
#include <stdint.h>

namespace hc {

constexpr uint64_t kSinkMaxGroups = 4096;  // the largest grid that collects its rows in per-workgroup segments (n_cu x 16; hc_kernels.hip: RowSink)
constexpr uint32_t kCodeN = 4;
constexpr uint32_t kCodeBadQual = 6;
constexpr uint32_t kCodeBadBase = 7;

// per-read flags in ReadDesc
constexpr uint32_t kReadPaired = 1u;
constexpr uint32_t kReadBadBase1 = 2u;  // sequence /1 (or the single) holds a base outside ACGTN:
constexpr uint32_t kReadBadBase2 = 4u;  // build_rev_comp exits when it is reverse-complemented (Types.h:124-127)

// slot stride in SYMBOLS for a sequence of `len` symbols of `symbytes` bytes each: the symbols, N padding to a multiple
// of 16 bytes plus 32 bytes, rounded up to `align` bytes (a power of two >= 16; chosen per read set by hc_set_reads:
// slots that start on a 128-byte line put a partner read's window — the slot's first L symbols — into the fewest lines,
// worth 3 - 4 % on sets whose store stays inside the Infinity Cache; the 10^8-candidate set with its 384 MB store loses
// 9 % to the third more memory)
__host__ __device__ inline uint64_t slot_stride(uint32_t len, uint32_t symbytes, uint32_t align) {
    uint64_t bytes = (uint64_t)len * symbytes;
    bytes = ((bytes + 15) & ~(uint64_t)15) + 32;
    bytes = (bytes + align - 1) & ~(uint64_t)(align - 1);
    return bytes / symbytes;
}

// One 32-byte descriptor per read: everything compute_overlap needs to pick the oriented
// sequences (one dependent load instead of three levels of tables).
struct ReadDesc {
    uint64_t off1, off2;  // forward-slot offsets (symbols) of /1 (or the single sequence) and /2
    uint32_t len1, len2;
    uint32_t flags;
    uint32_t rc_delta;  // symbols from a forward slot of this read to the reverse-complement slot of the same mate
};

struct StoreView {
    const void* sym;        // uint8_t* or uint16_t*
    const ReadDesc* reads;  // [n_reads]
    uint32_t n_reads;
    uint32_t n_seq;
    uint32_t K;         // quality alphabet size; table dimension Kp = K + 2
    uint32_t symbytes;  // 1 or 2
    uint32_t lut_bytes; // bytes of the log table as laid out for this symbol width
    uint32_t balance;   // 1: sequence lengths differ widely -> block-local length balancing in the scoring kernel
    uint64_t store_bytes;  // bytes of the symbol store (the cooperative fetch addresses it through a buffer descriptor)
    // "Regular" store: every sequence has the same length, the single-end reads come before the paired-end ones and no
    // sequence holds a base outside ACGTN.  A read's descriptor is then arithmetic (hc_resolve.h: regular_desc) and the
    // scoring kernel does not look it up — the dependent, random 32-byte read per candidate costs a quarter of the
    // kernel's memory-side rate (tools/experiments/gather_shapes.hip).  0: descriptors are read from `reads`.
    uint32_t regular;
    uint32_t ulen;      // the common sequence length
    uint32_t n_single;  // reads [0, n_single) own one sequence, the others two
    uint32_t seq_syms;  // symbols per sequence in the store (both orientations): 2 * slot_stride(ulen)
    // Contig-length sequences (mean length > 600): a candidate streams its two windows through dozens of 64-byte steps, and a
    // 128-byte line fetched for one step is wanted again by the next.  With 16 waves on every CU the lines in use exceed an
    // XCD's 4 MB L2 (32 CUs x 16 waves x 64 candidates x 2 windows x 128 B = 8.4 MB) and half of them are fetched twice; the
    // launch therefore keeps 8 waves per CU (BASELINE config 5: 0.97 -> 0.75 ms; 12 waves 0.90, 4 waves 0.86;
    // profiles/r03_occupancy.txt).  Short-read sets want all 16 (10^8 x 2 x 150 bp: 8 waves cost +27 %).
    uint32_t long_rows;
    uint32_t inv_len;     // entries of inv_n: the longest sequence + 17
    const double* inv_n;  // inv_n[k] = 1.0 / k (host-built, IEEE division: the reference's `1.0/total_len`, :137), k < inv_len; [0] unused
};

// Log table layouts (doubles):
//   uint8 symbols, LG = 3 (Kp <= 8): SPARSE, 16 KiB of address space for 128 entries:
//                   byte address = qa << 11 | m << 10 | (qa & 3) << 6 | ((qb ^ qa) & 7) << 3
//                   — every field sits where the symbol byte (qidx << 3 | code, low index bits repeated in bits 6-7),
//                   the XOR of the two symbol bytes and the mismatch flag (bit 2) already have it: two AND-ORs per four
//                   positions build the four addresses' bytes, no shifts.  The (qa & 3) and XOR fields select the LDS
//                   bank, as in the dense layouts.
//   uint8 symbols, LG in {4,5}: two dense planes (match, mismatch) of 2^LG x 2^LG entries, LG = ceil(log2(Kp)):
//                   byte address = m * (8 << 2LG) + qa * (8 << LG) + ((qb ^ qa) & (2^LG - 1)) * 8.
//   uint8 symbols, LG = 6 (the wide encoding, 64 KiB): x = qa ^ qb,
//                   byte address = m << 15 | qa << 9 | (x >> 5) << 8 | (((x & 31) ^ (qa >> 1)) & 31) << 3
//                   — the low byte (the LDS bank) mixes the XOR with the ROW's upper bits: with the plain XOR every pair of EQUAL
//                   qualities (a sixth of all positions of real reads) met in bank 0 at different rows.  Both bytes cost what the
//                   plain layout's cost: the row bits are the first symbol's own bits 7..3.
//                   The address of a position is exactly the 16-bit value one v_perm_b32 assembles from two
//                   pre-masked symbol bytes (no shift, no multiply); the XOR of the column with the row
//                   spreads the few hot (qa, qb) pairs over the LDS banks (rows of a power-of-two table
//                   would otherwise alias bank for bank).
//   uint16 symbols: two planes (match, mismatch) holding the LOWER TRIANGLE of the Kp x Kp table — log p(Q1, Q2) is
//                   symmetric in the two qualities (hc_set_reads checks the table it built, entry by entry) —:
//                   byte address = (m*T + hi*(hi+1)/2 + lo) * 8, hi/lo = larger/smaller of (qa, qb), T = Kp*(Kp+1)/2
//                   (entry index < 2*4753 fits 16 bits: two positions per packed-16-bit VALU op).  Half the LDS of
//                   the square layout: 31 KiB for 60 quality values, 74 KiB for the full Phred range.
constexpr uint32_t kWideN = 0, kWideBadQual = 1, kWideBadBase = 2;  // reserved qidx of the wide 8-bit encodings (index 3 is never dealt)
constexpr uint32_t kWideFirst = 16, kWideMaxK = 48;                  // LG = 6: quality values take indices 16 .. 63 (a base: top TWO index bits not both clear)
constexpr uint32_t kWide7First = 4, kWide7MaxK = 60;                 // LG = 7: 49 .. 60 values take indices 4 .. 63 (a base: top FOUR index bits not all clear)
// The index of the r-th most frequent quality value of a read set (wide encoding).  Any one-to-one assignment gives the same results
// (symbols and table are built from the same map); this one was searched (round 6, tools/experiments/r06_wide_labels.py) so that the
// (qa, qb) pairs of the few values that make up most of real reads' qualities fall into different LDS banks under the table's address
// mix (lut_addr_u8): simulated LDS cycles per 32-lane group of table reads 2.3 against 4.2 (POLYTE example), 1.4 against 2.7 (SAVAGE).
constexpr uint8_t kWideRankLabel[kWideMaxK] = {33, 25, 45, 21, 28, 44, 48, 36, 39, 20, 50, 32, 16, 63, 26, 40, 34, 24, 35, 37, 62, 54, 27, 55,
                                               17, 59, 46, 42, 58, 43, 30, 41, 38, 19, 61, 56, 29, 53, 31, 47, 23, 49, 18, 60, 51, 52, 57, 22};
constexpr uint8_t kWide7RankLabel[kWide7MaxK] = {8,  15, 4,  19, 24, 35, 49, 31, 6,  42, 20, 9,  61, 37, 18, 58, 16, 22, 23, 21,
                                                 45, 33, 54, 14, 25, 5,  47, 12, 57, 7,  43, 38, 63, 34, 28, 50, 27, 51, 44, 10,
                                                 26, 29, 11, 32, 60, 40, 39, 56, 13, 46, 36, 55, 17, 62, 53, 59, 52, 48, 41, 30};
// LG: log2 of the 8-bit-symbol table dimension; 6 and 7 select the wide encodings (both: 64 x 64 x 2 entries, 64 KiB).  (More than 60 values:
// 16-bit symbols, for which the value is not used.)
__host__ __device__ inline uint32_t lut_lg(uint32_t K) {
    return K + 2 <= 8 ? 3u : (K + 2 <= 16 ? 4u : (K + 2 <= 32 ? 5u : (K <= kWideMaxK ? 6u : (K <= kWide7MaxK ? 7u : 6u))));
}
__host__ __device__ inline uint32_t wide_first(uint32_t lg) { return lg == 7 ? kWide7First : kWideFirst; }
__host__ __device__ inline uint32_t sym_bytes_for(uint32_t K) { return K <= kWide7MaxK ? 1u : 2u; }
__host__ __device__ inline uint32_t lut_tri(uint32_t Kp) { return Kp * (Kp + 1u) / 2u; }
__host__ __device__ inline uint32_t lut_addr_u16(uint32_t Kp, uint32_t qa, uint32_t qb, uint32_t m) {
    const uint32_t hi = qa > qb ? qa : qb, lo = qa > qb ? qb : qa;
    return (m * lut_tri(Kp) + hi * (hi + 1u) / 2u + lo) * 8u;
}
// doubles of the 8-bit-symbol table as laid out in LDS
__host__ __device__ inline uint32_t lut_doubles_u8(uint32_t lg) { return lg == 3 ? 2048u : (lg >= 6 ? 8192u : (2u << (2 * lg))); }
__host__ __device__ inline uint32_t lut_addr_u8(uint32_t lg, uint32_t qa, uint32_t qb, uint32_t m) {
    if (lg == 3) return (qa << 11) | (m << 10) | ((qa & 3u) << 6) | (((qb ^ qa) & 7u) << 3);
    if (lg >= 6) return (m << 15) | (qa << 9) | ((((qb ^ qa) >> 5) & 1u) << 8) | ((((qb ^ qa) ^ (qa >> 1)) & 31u) << 3);
    return m * (8u << (2 * lg)) + qa * (8u << lg) + ((qb ^ qa) & ((1u << lg) - 1u)) * 8u;
}

// x-space image of a score threshold T: exp(x) > T  <=>  x > hi ; x <= lo => exp(x) <= T;
// lo < x <= hi is the guard band the host libm decides (normally empty).
struct Band {
    double lo, hi;
};

struct ScoreParams {
    Band edge, ov;
    double merge_contigs;
    uint32_t min_read_len;
    uint32_t flags;  // bit0: edge_threshold < 0 (every score passes), bit1: ov_threshold < 0
    uint32_t rec_fmt;  // HC_REC_FULL / HC_REC_COMPACT: layout of the candidate records of this launch
    uint32_t pad;      // bits 8..11: 64-candidate steps per item of the wave queue (hc_kernels.hip: WQ), set by launch_score
    const unsigned long long* n_dev;  // nullptr, or where the device holds the number of records (<= the launch's n)
};

}  // namespace hc
