// hc_text_kernels.hip — the overlaps file read on the device (gfx950): what construct_edges does with every line
// before process_overlaps sees it (reference src/EdgeCalculator.cpp:581-635; Overlap's constructor, src/Overlap.h:39-73),
// as three HBM-bound byte kernels over a block of the file's text:
//   text_count_kernel   newlines per 4 KiB tile (16 bytes per lane, one coalesced load)
//   text_scan_kernel    exclusive scan of the tile counts (one workgroup; a block has a few thousand tiles); it also zeroes the
//                       block's counters, sets the line count and passes it down the chain of line counters across blocks
//   text_lines_kernel   start offset of every line
//   text_parse_kernel   one lane per line: the 13 fields of a PLAIN line (single tabs, decimal numbers or "-", valid
//                       one-character fields — what sfo2overlaps.py and FNO write), --max_ov, the self-overlap test, the
//                       prefilter (:612-635), id -> read index; out come the 16-byte candidate record the scoring kernel
//                       reads (or a "skip" record), the parsed line (for the few per cent of lines the host sees again)
//                       and the prefilter's rejects.
// A line that is not plain — padded, malformed, hexadecimal ids, ... — is not read here: it is counted and (round 5,
// hc_textblock_list_nonplain) LISTED with its number and span, so that the host takes just that line through its own tokeniser +
// Overlap constructor, which own every error the reference can raise, and splices the verdict in at its place in file order;
// without a list (or with more such lines than the list holds) the host takes the whole block.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hcedge.h"
#include "hc_text.h"

#ifndef HC_TEXT_ABLATE_LINES
#define HC_TEXT_ABLATE_LINES 0
#endif

namespace hc {

// bytes of w equal to '\n', as a mask with bit 7 of each such byte set (exact: no borrow artefacts)
__device__ __forceinline__ uint32_t newline_mask(uint32_t w) {
    const uint32_t v = w ^ 0x0A0A0A0Au;
    return ~(((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v | 0x7F7F7F7Fu);
}

// newline masks of the 16 bytes at `at` (< n_bytes).  The last piece of a block reaches past the text: what lies there — an earlier,
// longer block's bytes — is not part of it and counts for nothing (the buffer has room for the read; until round 4 a memset of 64
// bytes behind every block's text made sure instead: three fill kernels per block for an unaligned end).
__device__ __forceinline__ void newline_masks16(const char* text, uint64_t n_bytes, uint64_t at, uint32_t m[4]) {
    const uint4 v = *(const uint4*)(text + at);
    m[0] = newline_mask(v.x);
    m[1] = newline_mask(v.y);
    m[2] = newline_mask(v.z);
    m[3] = newline_mask(v.w);
    if (at + 16 > n_bytes) {
        const uint32_t valid = (uint32_t)(n_bytes - at);  // 1 .. 15 bytes of text in this piece
#pragma unroll
        for (uint32_t w = 0; w < 4; w++) {
            const uint32_t have = valid > 4u * w ? valid - 4u * w : 0u;  // bytes of word w that are text
            if (have < 4u) m[w] &= have ? (1u << (8u * have)) - 1u : 0u;
        }
    }
}

__global__ __launch_bounds__(256) void text_count_kernel(const char* __restrict__ text, uint64_t n_bytes, uint32_t* __restrict__ tile_cnt) {
    const uint64_t at = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 16;
    uint32_t c = 0;
    if (at < n_bytes) {
        uint32_t m[4];
        newline_masks16(text, n_bytes, at, m);
        c = __builtin_popcount(m[0]) + __builtin_popcount(m[1]) + __builtin_popcount(m[2]) + __builtin_popcount(m[3]);
    }
    for (int o = 32; o > 0; o >>= 1) c += (uint32_t)__shfl_down((int)c, o, 64);
    __shared__ uint32_t part[4];
    if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// tile_off[t] = newlines before tile t; tile_off[n_tiles] = all of them.  One workgroup of 1 024 lanes.  It is the first kernel of
// a block that touches the block's counters: it zeroes them (instead of a memset per block), and with a line chain it passes the
// count on: *lines_before_next = *lines_before + this block's lines (line numbers across blocks without the host counting newlines).
__global__ __launch_bounds__(1024) void text_scan_kernel(const char* __restrict__ text, uint64_t n_bytes, const uint32_t* __restrict__ tile_cnt,
                                                         uint32_t n_tiles, uint32_t max_lines, uint32_t* __restrict__ tile_off,
                                                         uint32_t* __restrict__ line_start, unsigned long long* __restrict__ counters,
                                                         const unsigned long long* __restrict__ lines_before,
                                                         unsigned long long* __restrict__ lines_before_next) {
    __shared__ uint32_t wave_sum[16];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    if (threadIdx.x < kTextCounters) counters[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n_tiles; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < n_tiles ? tile_cnt[i] : 0u;
        uint32_t incl = v;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, o, 64);
            if ((int)(threadIdx.x & 63u) >= o) incl += up;
        }
        if ((threadIdx.x & 63u) == 63u) wave_sum[threadIdx.x >> 6] = incl;
        __syncthreads();
        uint32_t before = carry;
        for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) before += wave_sum[w];
        if (i < n_tiles) tile_off[i] = before + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        tile_off[n_tiles] = carry;
        // a last piece without a newline is a line too (std::getline): its virtual newline sits at n_bytes
        const bool open_end = n_bytes > 0 && text[n_bytes - 1] != '\n';
        const uint64_t n_lines = (uint64_t)carry + (open_end ? 1u : 0u);
        counters[kTextLines] = n_lines;
        counters[kTextOverflow] = n_lines > max_lines ? 1 : 0;
        if (open_end && n_lines <= max_lines) line_start[n_lines] = (uint32_t)(n_bytes + 1);
        if (lines_before_next) *lines_before_next = *lines_before + n_lines;
    }
}

// line_start[k + 1] = offset of the byte after the k-th newline; line_start[0] = 0 (the host adds the end of a last
// line without a newline).  Lines beyond max_lines are not written (the block then goes to the host: see hc_text.h).
__global__ __launch_bounds__(256) void text_lines_kernel(const char* __restrict__ text, uint64_t n_bytes, const uint32_t* __restrict__ tile_off,
                                                         uint32_t max_lines, uint32_t* __restrict__ line_start) {
    const uint64_t at = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 16;
    uint32_t m[4] = {0, 0, 0, 0};
    if (at < n_bytes) newline_masks16(text, n_bytes, at, m);
    const uint32_t c = __builtin_popcount(m[0]) + __builtin_popcount(m[1]) + __builtin_popcount(m[2]) + __builtin_popcount(m[3]);
    uint32_t incl = c;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, o, 64);
        if ((int)(threadIdx.x & 63u) >= o) incl += up;
    }
    __shared__ uint32_t wave_sum[4];
    if ((threadIdx.x & 63u) == 63u) wave_sum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t rank = tile_off[blockIdx.x] + incl - c;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) rank += wave_sum[w];
    if (blockIdx.x == 0 && threadIdx.x == 0) line_start[0] = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        uint32_t mask = m[w];
        while (mask) {
            const int bit = __builtin_ctz(mask);  // bit 7 of byte (bit >> 3)
            mask &= mask - 1;
            rank++;
            if (rank <= max_lines) line_start[rank] = (uint32_t)(at + 4u * w + (uint32_t)(bit >> 3) + 1u);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// id -> m_read_vec index (FastqStorage::m_ID_to_index: the first occurrence of an id wins, FastqStorage.h:88-97)
__device__ __forceinline__ bool id_lookup(const IdTable& t, uint64_t id, uint32_t& index) {
    if (t.direct) {
        if (id >= t.size) return false;
        index = t.table[id];
        return index != 0xFFFFFFFFu;
    }
    uint64_t h = (id * 0x9E3779B97F4A7C15ull) >> t.shift;
    for (;;) {
        const uint32_t v = t.table[h];
        if (v == 0xFFFFFFFFu) return false;
        if (t.keys[h] == id) {
            index = v;
            return true;
        }
        h = (h + 1) & (t.size - 1);
    }
}

// The characters of a line through a 4-byte window: one aligned word read per four characters instead of one dependent
// byte read each (the line sits in LDS or in the block's text: either way a read is ~100 cycles of latency in the lane's
// serial walk; text_parse_kernel is bound by exactly that walk).  Reads stay inside [line & ~3, (end + 3) & ~3): the
// staged text and the block's text both have room on either side.
typedef const __attribute__((address_space(3))) char* lds_text;      // the workgroup's stretch of text staged in LDS
typedef const __attribute__((address_space(3))) uint32_t* lds_words;
template <bool LDS>
struct Cursor {
    uint32_t at, e;     // byte offsets from `base`
    uint32_t w;         // the aligned word `at` points into
    const char* base;   // global text (LDS: unused, offsets are LDS addresses)
    __device__ __forceinline__ uint32_t word(uint32_t off) const {
        if (LDS) return *(lds_words)(uintptr_t)off;
        return *(const uint32_t*)(base + off);
    }
    __device__ __forceinline__ void load() { w = word(at & ~3u); }
    __device__ __forceinline__ uint32_t cur() const { return (w >> (8u * (at & 3u))) & 0xFFu; }
    __device__ __forceinline__ void next() {
        at++;
        if ((at & 3u) == 0) w = word(at);
    }
};

// digits (at most max_digits) followed by a tab; dash_ok: a lone "-" reads as 0 (atoi("-"))
template <bool LDS>
__device__ __forceinline__ bool take_number(Cursor<LDS>& c, unsigned max_digits, bool dash_ok, uint64_t& v, bool& dash) {
    dash = false;
    v = 0;
    if (dash_ok && c.at < c.e && c.cur() == '-') {
        dash = true;
        c.next();
    } else {
        const uint32_t b = c.at;
        const uint32_t first = c.at < c.e ? c.cur() : 0u;
        while (c.at < c.e && (c.cur() - '0') <= 9u) {
            v = v * 10 + (uint64_t)(c.cur() - '0');
            c.next();
        }
        const unsigned d = c.at - b;
        if (d == 0 || d > max_digits) return false;
        if (!dash_ok && first == '0' && d > 1) return false;  // strtoul(.., 0) reads a leading 0 as octal: not plain
    }
    if (c.at >= c.e || c.cur() != '\t') return false;
    c.next();
    return true;
}

template <bool LDS>
__device__ __forceinline__ bool take_char(Cursor<LDS>& c, char& ch, bool last) {
    if (c.at >= c.e) return false;
    ch = (char)c.cur();
    c.next();
    if (last) return c.at == c.e;
    if (c.at >= c.e || c.cur() != '\t') return false;
    c.next();
    return true;
}

// Overlap::from_plain_line (host_model.cpp) on the device: the same acceptance rules, the same values.
// LDS: `begin` is the LDS byte address of the line; else its offset in `text`.
template <bool LDS>
__device__ __forceinline__ bool parse_plain_line(const char* text, uint32_t begin, uint32_t n, hc_line_rec& o) {
    Cursor<LDS> c{begin, begin + n, 0u, text};
    c.load();
    uint64_t id1, id2, pos1, pos2, perc1, perc2, len1, len2;
    bool dash, dash_pos2;
    char ord, ori1, ori2, type1, type2;
    if (!take_number(c, 18, false, id1, dash) || !take_number(c, 18, false, id2, dash) || !take_number(c, 9, true, pos1, dash) ||
        !take_number(c, 9, true, pos2, dash_pos2) || !take_char(c, ord, false) || !take_char(c, ori1, false) || !take_char(c, ori2, false) ||
        !take_number(c, 9, true, perc1, dash) || !take_number(c, 9, true, perc2, dash) || !take_number(c, 9, true, len1, dash) ||
        !take_number(c, 9, true, len2, dash) || !take_char(c, type1, false) || !take_char(c, type2, true))
        return false;
    if (dash_pos2) perc2 = len2 = 0;  // src/Overlap.h:55-59
    if ((ori1 != '+' && ori1 != '-') || (ori2 != '+' && ori2 != '-')) return false;
    if (perc1 > 100 || perc2 > 100) return false;
    if ((type1 != 's' && type1 != 'p') || (type2 != 's' && type2 != 'p')) return false;
    if (type1 == 's' || type2 == 's' ? ord != '-' : (ord != '1' && ord != '2')) return false;
    o.id1 = id1;
    o.id2 = id2;
    o.pos1 = (uint32_t)pos1;
    o.pos2 = (uint32_t)pos2;
    o.perc1 = (uint32_t)perc1;
    o.perc2 = (uint32_t)perc2;
    o.len1 = (uint32_t)len1;
    o.len2 = (uint32_t)len2;
    o.ord = (uint8_t)ord;
    o.ori1 = (uint8_t)ori1;
    o.ori2 = (uint8_t)ori2;
    o.type1 = (uint8_t)type1;
    o.type2 = (uint8_t)type2;
    o.pad[0] = o.pad[1] = o.pad[2] = 0;
    return true;
}

// What construct_edges does with a parsed line before process_overlaps (src/EdgeCalculator.cpp:605-635): self overlaps go, the prefilter
// passes, drops silently or rejects (kept for nonedge_overlaps.txt), the two ids are looked up (m_ID_to_index, :164-171); a line that passes
// becomes the candidate record cd.  Shared by the kernel that parses the file's text and the one that takes parsed lines as they are
// (lines_accept_kernel: the device-resident stage a).
__device__ __forceinline__ void accept_line(const TextParams& prm, const IdTable& ids, const hc_line_rec& o, uint32_t i, hc_cand_rec& cd,
                                            hc_text_reject* __restrict__ rejects, unsigned long long* __restrict__ counters, uint32_t& n_self,
                                            uint32_t& n_silent, uint32_t& n_reject, uint32_t& n_pass, uint32_t& n_unknown) {
    if (o.id1 == o.id2) {  // :605-607
        n_self = 1;
        return;
    }
    const uint32_t perc = o.perc2 > 0 ? (o.perc1 + o.perc2) >> 1 : o.perc1;  // Overlap::get_perc: (unsigned)(0.5 * (a + b))
    const bool ss = o.type1 == 's' && o.type2 == 's', anyp = !ss;
    const uint64_t M = prm.min_overlap_len;
    bool pass = false;
    if (o.len1 >= M && ss) {  // :612-617
        pass = perc >= prm.min_overlap_perc;
        n_silent = pass ? 0 : 1;
    } else if (2ull * o.len1 >= M && 2ull * o.len2 >= M && anyp) {  // :618-624: len >= 0.5 * M, exactly
        pass = perc >= prm.min_overlap_perc;
        n_silent = pass ? 0 : 1;
    } else if (prm.relax_pe && (uint64_t)o.len1 + o.len2 >= M && anyp) {  // :626-632 (unsigned int sum: no wrap below 2^32 here, both < 10^9)
        pass = perc >= prm.min_overlap_perc;
        n_silent = pass ? 0 : 1;
    } else {  // :633-635
        n_reject = 1;
        const unsigned long long slot = atomicAdd(&counters[kTextRejectSlots], 1ull);
        if (slot < prm.reject_cap) {
            hc_text_reject r;
            r.line_index = i;
            r.pad = 0;
            r.line = o;
            rejects[slot] = r;
        }
    }
    if (pass) {
        uint32_t r1, r2;
        if (!id_lookup(ids, o.id1, r1) || !id_lookup(ids, o.id2, r2)) {
            n_unknown = 1;  // std::map::at throws, :170-171: the host reproduces the failure
        } else {
            n_pass = 1;
            const uint32_t p1 = o.pos1 < HC_CAND_POS_MASK ? o.pos1 : HC_CAND_POS_MASK;
            const uint32_t p2 = o.pos2 < HC_CAND_POS_MASK ? o.pos2 : HC_CAND_POS_MASK;
            const uint32_t oc = o.ord == '-' ? 0u : (o.ord == '1' ? 1u : 2u);
            cd.read1 = r1;
            cd.read2 = r2;
            cd.pos1_bits = p1 | (o.ori1 == '+' ? 1u << 28 : 0u) | (o.ori2 == '+' ? 1u << 29 : 0u) | (oc << 30);
            cd.pos2_bits = p2;
        }
    }
}

constexpr uint32_t kStageBytes = 32 * 1024;  // text of the 256 lines of a workgroup staged in LDS when it fits (a plain line has ~45 bytes)

// counters: the enum of hc_text.h
__global__ __launch_bounds__(256) void text_parse_kernel(TextParams prm, const char* __restrict__ text, const uint32_t* __restrict__ line_start,
                                                         IdTable ids, hc_cand_rec* __restrict__ cands, hc_line_rec* __restrict__ lines,
                                                         hc_text_reject* __restrict__ rejects, unsigned long long* __restrict__ counters,
                                                         uint32_t* __restrict__ tally /* [workgroups][8] */, hc_text_nonplain* __restrict__ nonplain) {
    __shared__ __attribute__((aligned(16))) char stage[kStageBytes + 32];  // + the 16-byte pieces at both ends
    __shared__ uint32_t wave_tally[4][8];
    if (counters[kTextOverflow]) return;  // more lines than room: nothing is parsed here, the host takes the block
    const uint32_t n = (uint32_t)counters[kTextLines];
    const uint32_t first = blockIdx.x * 256u;
    if (first >= n) {  // slots behind the last line: "skip" records, the scoring kernel steps over them
        const uint32_t i = first + threadIdx.x;
        if (i < prm.max_lines) {
            hc_cand_rec cd;
            cd.read1 = cd.read2 = 0;
            cd.pos1_bits = 0;
            cd.pos2_bits = HC_CAND_SKIP;
            cands[i] = cd;
        }
        return;
    }
    const uint32_t last = first + 256u < n ? first + 256u : n;  // lines [first, last)
    const uint32_t lo = line_start[first], hi = line_start[last];  // bytes [lo, hi): hi counts the newline of the last line
    const bool staged = hi - lo <= kStageBytes;
    if (staged) {  // coalesced copy of the workgroup's stretch of text (16-byte pieces; lo is arbitrary: align down)
        const uint32_t a = lo & ~15u;
        for (uint32_t at = a + threadIdx.x * 16u; at < hi; at += 256u * 16u) *(uint4*)(stage + (at - a)) = *(const uint4*)(text + at);
        __syncthreads();
    }
    const uint32_t i = first + threadIdx.x;
    uint32_t n_read = 0, n_nonplain = 0, n_self = 0, n_silent = 0, n_reject = 0, n_pass = 0, n_unknown = 0;
    const uint64_t first_line = prm.first_line_no + (prm.first_line_ptr ? *prm.first_line_ptr : 0ull);
    if (i < n && first_line + i < prm.max_overlaps) {  // `&& i < max_overlaps`, :581
        n_read = 1;
        const uint32_t b = line_start[i];
        uint32_t e = line_start[i + 1] - 1u;  // the newline (or the virtual one behind a last line without)
        // the line: in the staged copy (an LDS address) or in the block's text (an offset)
        const uint32_t stage_at = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)stage + (b - (lo & ~15u));
        hc_cand_rec cd;
        cd.read1 = cd.read2 = 0;
        cd.pos1_bits = 0;
        cd.pos2_bits = HC_CAND_SKIP;
        hc_line_rec o;
        if (!(staged ? parse_plain_line<true>(text, stage_at, e - b, o) : parse_plain_line<false>(text, b, e - b, o))) {
            n_nonplain = 1;
            if (prm.nonplain_cap) {  // the host reads this line alone (src/EdgeCalculator.cpp:584-604)
                const unsigned long long slot = atomicAdd(&counters[kTextNonPlainSlots], 1ull);
                if (slot < prm.nonplain_cap) {
                    hc_text_nonplain np;
                    np.line_index = i;
                    np.begin = b;
                    np.length = e - b;
                    np.pad = 0;
                    nonplain[slot] = np;
                }
            }
        } else {
#if !HC_TEXT_ABLATE_LINES  // (experiment builds only, tools/experiments/r06_text_no_lines.sh: what NOT writing every parsed line would save at most)
            lines[i] = o;
#endif
            accept_line(prm, ids, o, i, cd, rejects, counters, n_self, n_silent, n_reject, n_pass, n_unknown);
        }
        cands[i] = cd;
    } else if (i < prm.max_lines) {
        hc_cand_rec cd;
        cd.read1 = cd.read2 = 0;
        cd.pos1_bits = 0;
        cd.pos2_bits = HC_CAND_SKIP;
        cands[i] = cd;
    }
    // tallies: summed over the workgroup and written to its own slot — no atomics (every wave adding to the same seven
    // counters of one cache line serialised there: the kernel then took as long as those ~40 000 atomics, 160 us a block)
    const uint32_t vals[7] = {n_read, n_nonplain, n_self, n_silent, n_reject, n_pass, n_unknown};
#pragma unroll
    for (int k = 0; k < 7; k++) {
        const uint32_t c = (uint32_t)__popcll(__ballot(vals[k] != 0));
        if ((threadIdx.x & 63u) == 0) wave_tally[threadIdx.x >> 6][k] = c;
    }
    __syncthreads();
    if (threadIdx.x < 7) tally[blockIdx.x * 8u + threadIdx.x] = wave_tally[0][threadIdx.x] + wave_tally[1][threadIdx.x] + wave_tally[2][threadIdx.x] + wave_tally[3][threadIdx.x];
}

// ------------------------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------------------------
// Parsed lines as they are (round 6: the device-resident stage a — the lines come from the SFO matcher's kernels, hc_sfo_kernels.hip, and
// no text exists): the block's counters as the line-start scan leaves them, then text_parse_kernel's second half on src[i].
__global__ void lines_prepare_kernel(unsigned long long* __restrict__ counters, uint32_t n_lines, uint32_t max_lines) {
    if (threadIdx.x < kTextCounters) counters[threadIdx.x] = 0;
    __syncthreads();
    if (threadIdx.x == 0) {
        counters[kTextLines] = n_lines <= max_lines ? n_lines : 0u;
        counters[kTextOverflow] = n_lines > max_lines ? 1u : 0u;
    }
}

__global__ __launch_bounds__(256) void lines_accept_kernel(TextParams prm, const hc_line_rec* __restrict__ src, IdTable ids,
                                                           hc_cand_rec* __restrict__ cands, hc_line_rec* __restrict__ lines,
                                                           hc_text_reject* __restrict__ rejects, unsigned long long* __restrict__ counters,
                                                           uint32_t* __restrict__ tally /* [workgroups][8] */) {
    __shared__ uint32_t wave_tally[4][8];
    if (counters[kTextOverflow]) return;
    const uint32_t n = (uint32_t)counters[kTextLines];
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    uint32_t n_read = 0, n_self = 0, n_silent = 0, n_reject = 0, n_pass = 0, n_unknown = 0;
    hc_cand_rec cd;
    cd.read1 = cd.read2 = 0;
    cd.pos1_bits = 0;
    cd.pos2_bits = HC_CAND_SKIP;
    if (i < n && prm.first_line_no + i < prm.max_overlaps) {  // `&& i < max_overlaps`, :581
        n_read = 1;
        const hc_line_rec o = src[i];
        lines[i] = o;
        accept_line(prm, ids, o, i, cd, rejects, counters, n_self, n_silent, n_reject, n_pass, n_unknown);
    }
    if (i < prm.max_lines) cands[i] = cd;
    const uint32_t vals[7] = {n_read, 0u, n_self, n_silent, n_reject, n_pass, n_unknown};
#pragma unroll
    for (int k = 0; k < 7; k++) {
        const uint32_t c = (uint32_t)__popcll(__ballot(vals[k] != 0));
        if ((threadIdx.x & 63u) == 0) wave_tally[threadIdx.x >> 6][k] = c;
    }
    __syncthreads();
    if (threadIdx.x < 7) tally[blockIdx.x * 8u + threadIdx.x] = wave_tally[0][threadIdx.x] + wave_tally[1][threadIdx.x] + wave_tally[2][threadIdx.x] + wave_tally[3][threadIdx.x];
}

hipError_t launch_text_count(const char* text, uint64_t n_bytes, uint32_t* tile_cnt, hipStream_t s) {
    const uint32_t n_tiles = (uint32_t)((n_bytes + 4095) / 4096);
    if (n_tiles == 0) return hipSuccess;
    hipLaunchKernelGGL(text_count_kernel, dim3(n_tiles), dim3(256), 0, s, text, n_bytes, tile_cnt);
    return hipGetLastError();
}
hipError_t launch_text_scan(const char* text, uint64_t n_bytes, const uint32_t* tile_cnt, uint32_t* tile_off, uint32_t max_lines, uint32_t* line_start,
                            unsigned long long* counters, const unsigned long long* lines_before, unsigned long long* lines_before_next,
                            hipStream_t s) {
    const uint32_t n_tiles = (uint32_t)((n_bytes + 4095) / 4096);
    hipLaunchKernelGGL(text_scan_kernel, dim3(1), dim3(1024), 0, s, text, n_bytes, tile_cnt, n_tiles, max_lines, tile_off, line_start, counters,
                       lines_before, lines_before_next);
    return hipGetLastError();
}
hipError_t launch_text_line_starts(const char* text, uint64_t n_bytes, const uint32_t* tile_off, uint32_t max_lines, uint32_t* line_start, hipStream_t s) {
    const uint32_t n_tiles = (uint32_t)((n_bytes + 4095) / 4096);
    if (n_tiles == 0) return hipSuccess;
    hipLaunchKernelGGL(text_lines_kernel, dim3(n_tiles), dim3(256), 0, s, text, n_bytes, tile_off, max_lines, line_start);
    return hipGetLastError();
}

hipError_t launch_text_parse(const TextParams& prm, const char* text, const uint32_t* line_start, const IdTable& ids, hc_cand_rec* cands,
                             hc_line_rec* lines, hc_text_reject* rejects, unsigned long long* counters, uint32_t* tally, hc_text_nonplain* nonplain,
                             hipStream_t s) {
    if (prm.max_lines == 0) return hipSuccess;
    hipLaunchKernelGGL(text_parse_kernel, dim3((prm.max_lines + 255) / 256), dim3(256), 0, s, prm, text, line_start, ids, cands, lines, rejects,
                       counters, tally, nonplain);
    return hipGetLastError();  // (the workgroups' tallies are summed by the one-workgroup scan of launch_kept_rows_flushed)
}

hipError_t launch_lines_accept(const TextParams& prm, const hc_line_rec* src, uint32_t n_lines, const IdTable& ids, hc_cand_rec* cands, hc_line_rec* lines,
                               hc_text_reject* rejects, unsigned long long* counters, uint32_t* tally, hipStream_t s) {
    hipLaunchKernelGGL(lines_prepare_kernel, dim3(1), dim3(64), 0, s, counters, n_lines, prm.max_lines);
    hipLaunchKernelGGL(lines_accept_kernel, dim3((prm.max_lines + 255u) / 256u), dim3(256), 0, s, prm, src, ids, cands, lines, rejects, counters, tally);
    return hipGetLastError();
}

}  // namespace hc
