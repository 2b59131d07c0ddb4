/*
 * hcedge.h — C ABI of libhcedge.so: the MI355X (gfx950) replacement for the
 * overlap-graph edge-calculation hot path of HaploConduct's ViralQuasispecies
 * binary.
 *
 * The reference has no FFI of its own (it is one statically linked C++ binary),
 * so this header *defines* the cut.  Every entry point names the reference
 * interface it replaces (file:line relative to the HaploConduct source tree).
 *
 *   reference                                        this ABI
 *   ------------------------------------------------ ---------------------------
 *   EdgeCalculator::EdgeCalculator (EdgeCalculator.h:43-53)   hc_create
 *     + ProgramSettings fields it reads (Types.h:19-67)       hc_settings
 *   FastqStorage::m_read_vec / Read::get_seq,get_phred,
 *     get_rev_comp,get_rev_phred (FastqStorage.h:48-55,
 *     Read.h:144-201)                                         hc_set_reads
 *   the `omp for` body of EdgeCalculator::process_overlaps
 *     = compute_overlap + 3-way classification
 *     (EdgeCalculator.cpp:400-414, 143-385, 67-139, 26-63)    hc_score_batch[_device]
 *   score = exp(...), 0.5*(ov1+ov2) / min(ov1,ov2)
 *     (EdgeCalculator.cpp:137-138, 254-261)                   hc_finalize
 *   EdgeCalculator::~EdgeCalculator                           hc_destroy
 *   EdgeCalculator::construct_edges() and the serial insert
 *     of process_overlaps (EdgeCalculator.cpp:427-545,
 *     561-666) over OverlapGraph (OverlapGraph.cpp:94-311)    hc_ec_* (host side, below)
 *
 * Conventions: plain pointers and sizes only; every function returns HC_OK (0)
 * or a negative hc_status and never calls exit()/abort(); the caller owns all
 * host buffers; one context is used from one thread at a time (the reference
 * has exactly one batch in flight, EdgeCalculator.cpp:636-644), and its asynchronous
 * entry points (the *_device calls) share grow-only scratch owned by the context:
 * at most ONE stream may have work of a context in flight — pass the same stream to
 * consecutive calls, or synchronise before switching streams.  hc_block objects
 * (below) own their buffers and stream and may be in flight side by side.
 */
#ifndef HCEDGE_H_
#define HCEDGE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum hc_status {
    HC_OK = 0,
    HC_ERR_ARG = -1,         /* null pointer / bad size / bad enum value                */
    HC_ERR_NOMEM = -2,       /* host or device allocation failed                        */
    HC_ERR_HIP = -3,         /* a HIP runtime call failed (hc_last_error has the text)  */
    HC_ERR_NO_DEVICE = -4,   /* no gfx950 device visible: there is NO CPU fallback      */
    HC_ERR_STATE = -5,       /* call order violated (e.g. score before set_reads)       */
    HC_ERR_BAD_READ = -6,    /* empty sequence / base+quality length mismatch           */
    HC_ERR_BAD_OVERLAP = -7, /* read index out of range, read1==read2, bad ord/ori      */
    HC_ERR_IO = -8,          /* file could not be opened / read / written               */
    HC_ERR_FORMAT = -9,      /* input the reference would exit(1)/assert on             */
    HC_ERR_DATA = -10,       /* an overlap touched a base/quality byte the reference
                                asserts on (EdgeCalculator.cpp:29-30,61)                */
    HC_ERR_NOT_ON_DEVICE = -11 /* hc_found_to_lines_device: an input the device does not decide (an assert of
                                scripts/sfo2overlaps.py, ids / numbers outside its sort keys): NOT an error of the
                                input — the caller takes the host's route, which raises what the script raises */
} hc_status;

/* ---- settings: the ProgramSettings fields the hot path reads (Types.h:19-67) ---- */
#define HC_FLAG_ADD_DUPLICATES       0x1u /* --add_duplicates                            */
#define HC_FLAG_RESOLVE_ORIENTATIONS 0x2u /* --resolve_orientations (default true)       */
#define HC_FLAG_IGNORE_INCLUSIONS    0x4u /* --ignore_inclusions                         */
#define HC_FLAG_RELAX_PE_EDGES       0x8u /* --relax_PE_edges                            */
#define HC_FLAG_ALLOW_SPACES         0x10u /* --allow_spaced_overlaps                    */
#define HC_FLAG_VERBOSE              0x20u /* --verbose                                  */

typedef struct hc_settings {
    double edge_threshold;     /* --edge_threshold  (default 0.99)                       */
    double ov_threshold;       /* --ov_threshold    (default 0.9)                        */
    double merge_contigs;      /* --merge_contigs   (default 0)                          */
    double mismatch;           /* --mismatch        (default 0)                          */
    uint32_t min_read_len;     /* --min_read_len    (default 0)                          */
    uint32_t min_overlap_len;  /* --min_overlap_len (default 150)  [host prefilter]      */
    uint32_t min_overlap_perc; /* --min_overlap_perc (default 0)   [host prefilter]      */
    uint32_t flags;            /* HC_FLAG_*                                              */
    uint64_t max_overlaps;     /* --max_ov          (default 1e8)  [host parser]         */
    int32_t device;            /* HIP device ordinal (hc_create; the stage's primary)    */
    uint32_t n_threads;        /* --threads: host-side worker threads (parser)           */
    uint32_t device_mask;      /* stage only (hc_ec_*): bit d = score blocks on device d too;
                                  0 = `device` alone.  hc_create always uses `device`.   */
    uint32_t reserved;         /* 0                                                      */
} hc_settings;

/* ---- one candidate overlap, as it enters process_overlaps (Overlap.h:20-73) ----
 * read1/read2 index FastqStorage::m_read_vec (singles first, then pairs,
 * FastqStorage.h:88-97); the id->index map lookup of EdgeCalculator.cpp:164-171
 * is done once by the host parser. */
typedef struct hc_overlap_rec {
    uint32_t read1, read2;
    uint32_t pos1, pos2;
    uint8_t ori1, ori2; /* 1 = "+", 0 = "-"                                               */
    uint8_t ord;        /* '-', '1' or '2'                                                */
    uint8_t flags;      /* bit0: type1=='p', bit1: type2=='p' as WRITTEN in the file
                           (prefilter only; scoring uses the reads' own types,
                           EdgeCalculator.cpp:186-187)                                    */
    uint32_t len1, len2;
    uint32_t perc;      /* Overlap::get_perc() (Overlap.h:203-210)                        */
} hc_overlap_rec;       /* 32 bytes */

/* ---- the same candidate, only what the DEVICE reads of it (compute_overlap takes ids, positions, orientations
 * and ord from the record and everything else from the reads, EdgeCalculator.cpp:143-385): the form that crosses
 * PCIe in the stage — 16 bytes instead of 32; LEN1/LEN2/PERC stay on the host, where the admitted edges are built.
 *   pos1_bits: bits 0..27 POS1, bit 28 ORI1 == '+', bit 29 ORI2 == '+', bits 30..31 ORD: 0 = '-', 1 = '1', 2 = '2'
 *   pos2_bits: bits 0..27 POS2, bits 28..30 zero, bit 31 (HC_CAND_SKIP): not a candidate — the scoring kernel writes a
 *              dropped result and reads nothing else of the record (the device's text parser marks the lines it
 *              dropped this way instead of compacting the block)
 * Positions saturate at 2^28-1: every stored sequence is shorter than that (hc_set_reads), so `pos >= length`
 * (EdgeCalculator.cpp:76-79) holds for the saturated value exactly when it holds for the original. */
#define HC_CAND_POS_MASK 0x0FFFFFFFu
#define HC_CAND_SKIP 0x80000000u
typedef struct hc_cand_rec {
    uint32_t read1, read2;
    uint32_t pos1_bits, pos2_bits;
} hc_cand_rec;          /* 16 bytes */
#define HC_REC_FULL 0u    /* hc_overlap_rec */
#define HC_REC_COMPACT 1u /* hc_cand_rec    */

/* ---- per-candidate result of the device pass --------------------------------
 * x_k = (1.0/total_len) * total_score of sub-overlap k, i.e. the argument of the
 * final exp() at EdgeCalculator.cpp:137-138, bit-identical to the reference's
 * value (same libm-built log table, same summation order, no FMA contraction).
 * x_k = -inf encodes every `return 0` of overlap_score (score 0, mismatch 1.0).
 * x2 = NaN for s-s overlaps (one sub-overlap).
 * (mm, n): mismatch_count / total_len of the sub-overlap with the larger
 * mismatch rate (max() at EdgeCalculator.cpp:254); 1/1 for early exits.
 * cls: the 3-way decision of EdgeCalculator.cpp:404-413 taken on the device in
 * x-space (thresholds pre-inverted through the host libm exp);
 * HC_CLS_AMBIG = x inside the (normally empty) guard band, host decides. */
#define HC_CLS_DROP 0u
#define HC_CLS_NONEDGE 1u /* goes to nonedge_overlaps.txt (EdgeCalculator.cpp:410-413)  */
#define HC_CLS_EDGE 2u    /* score > edge_threshold (EdgeCalculator.cpp:404)             */
#define HC_CLS_EDGE_MC 3u /* mismatch_rate <= merge_contigs (EdgeCalculator.cpp:407)     */
#define HC_CLS_AMBIG 4u
#define HC_CLS_ERROR 7u   /* touched an invalid base / quality byte                      */

typedef struct hc_result_rec {
    double x1;
    double x2;
    uint32_t mm;
    uint32_t n_cls; /* bits 0..27 total_len, bits 28..31 HC_CLS_*                         */
} hc_result_rec;    /* 24 bytes */

#define HC_RES_N(r) ((r).n_cls & 0x0FFFFFFFu)
#define HC_RES_CLS(r) ((r).n_cls >> 28)

typedef struct hc_ctx hc_ctx; /* opaque: device read store, log-prob table, streams */

/* Library / device ----------------------------------------------------------- */
const char* hc_version(void);
const char* hc_strerror(int status);
/* Text of the last failure on this thread ("" if none). */
const char* hc_last_error(void);
/* Number of visible HIP devices (0 if none); does not create a context. */
int hc_device_count(void);

/* Replaces EdgeCalculator's constructor (EdgeCalculator.h:43-53).  Builds the
 * pow/log table with the host libm exactly as EdgeCalculator.cpp:41,44,52,60 do. */
int hc_create(hc_ctx** out, const hc_settings* settings);
int hc_destroy(hc_ctx* ctx);
/* A context for another stage without making a new one (round 5, the resident process: a pipeline's iterations keep their contexts, streams,
 * scratch and text blocks): the read store goes, the settings are replaced — the context is then as hc_create(settings) had left it, on the
 * same device.  Blocks made on the context (hc_block, hc_textblock) stay valid.  Not while anything is in flight. */
int hc_reset(hc_ctx* ctx, const hc_settings* settings);

/* Replaces the Read getters used by compute_overlap (Read.h:144-201).
 *   bases, quals : concatenated sequences / quality strings (ASCII, 1 byte per base)
 *   seq_off      : n_seq+1 offsets into bases/quals
 *   read_first_seq : n_reads+1; read r owns sequences [read_first_seq[r], read_first_seq[r+1]):
 *                  one (single-end) or two (/1, /2 of a pair)
 * Buffers are borrowed until return; the store (both orientations, re-encoded)
 * lives in HBM until hc_destroy or the next hc_set_reads.  The grow-only scratch earlier calls of hc_find_overlaps and of the SFO ingest
 * left on the context (gigabytes at config 3's size) stays through hc_set_reads and hc_reset — it is capacity for the next read set; only
 * hc_destroy returns it (round 6: giving it back per read set made one hipMalloc in three take 0.8 s, profiles/r06_stage_a_from_store.md). */
int hc_set_reads(hc_ctx* ctx, const uint8_t* bases, const uint8_t* quals, const uint64_t* seq_off,
                 const uint32_t* read_first_seq, uint32_t n_reads);

/* Replaces the omp-for of process_overlaps (EdgeCalculator.cpp:400-414).
 * Synchronous; out[i] <-> in[i]; host buffers. */
int hc_score_batch(hc_ctx* ctx, const hc_overlap_rec* in, uint64_t n, hc_result_rec* out);

/* Candidate order.  The kernel is fastest when neighbouring candidates share a read, as they do in
 * overlap files written by sfo2overlaps (sorted by smaller/larger read id, scripts/sfo2overlaps.py:53)
 * and by FNO (a sorted std::set).  Results never depend on the order; out[i] <-> in[i] always.
 *   HC_REORDER_NEVER  score the batch as given
 *   HC_REORDER_ALWAYS sort candidate indices by min(read1, read2) on the device first
 *   HC_REORDER_AUTO   (default) hc_score_batch probes its host copy and reorders only unordered batches;
 *                     hc_score_batch_device scores as given (it cannot look without synchronising) */
#define HC_REORDER_NEVER 0
#define HC_REORDER_ALWAYS 1
#define HC_REORDER_AUTO 2
int hc_set_reorder(hc_ctx* ctx, int mode);

/* Page-locked host memory for the in/out arrays of hc_score_batch (hipHostMalloc): DMA at full
 * PCIe rate instead of staging through the driver's bounce buffers.  Optional; any host memory works. */
int hc_host_alloc(hc_ctx* ctx, void** ptr, uint64_t bytes);
int hc_host_free(hc_ctx* ctx, void* ptr);

/* Same, but in/out are DEVICE pointers (hipMalloc'd, or torch CUDA tensors) and
 * the launch is asynchronous on `hip_stream` (a hipStream_t, NULL = the context's
 * own stream).  This is the entry bench.py times: inputs already resident in HBM. */
int hc_score_batch_device(hc_ctx* ctx, const void* d_in, uint64_t n, void* d_out, void* hip_stream);
int hc_synchronize(hc_ctx* ctx);

/* The same two entry points for compact records (hc_cand_rec, 16 bytes): what the stage sends over PCIe. */
void hc_pack_cands(const hc_overlap_rec* in, uint64_t n, hc_cand_rec* out); /* host, pure: record -> compact record */
int hc_score_cands(hc_ctx* ctx, const hc_cand_rec* in, uint64_t n, hc_result_rec* out);
int hc_score_cands_device(hc_ctx* ctx, const void* d_cands, uint64_t n, void* d_out, void* hip_stream);

/* ---- one block of the stage in flight (EdgeCalculator.cpp:636-644: "flush every 1e6 overlaps") -------------------
 * construct_edges() hands the scoring loop one block of candidates at a time.  An hc_block owns what one block needs
 * on the device — candidate and result buffers, a stream, an event — plus a page-locked host buffer mapped into the
 * device's address space, into which the scoring kernel itself writes every record that is NOT dropped (a few per
 * cent of the block), tagged with its candidate index: nothing is compacted or copied back afterwards.
 *   hc_block_submit : asynchronous — H2D of the compact records, the scoring kernel, the row count
 *   hc_block_wait   : blocks until the block has been scored; *rows = the non-dropped records sorted by index
 *                     (index = base_index + position in the block), valid until the next submit on this block.
 * Several blocks of one context (or of several contexts on several devices) may be in flight at once. */
typedef struct hc_gather_row { /* a non-dropped result record tagged with its candidate index */
    uint64_t index;
    double x1, x2;
    uint32_t mm, n_cls;
} hc_gather_row; /* 32 bytes */
typedef struct hc_block hc_block;
int hc_block_create(hc_ctx* ctx, uint64_t max_candidates, hc_block** out);
int hc_block_submit(hc_block* b, const hc_cand_rec* cands, uint64_t n, uint64_t base_index);
int hc_block_wait(hc_block* b, const hc_gather_row** rows, uint64_t* n_rows);
int hc_block_destroy(hc_block* b);

/* Stream compaction of a scored batch: the indices (ascending = sequence order) of the records whose
 * class is not HC_CLS_DROP — admitted edges, non-edges kept for FNO, ambiguous and error records —
 * i.e. everything the host or the multi-GPU gather still has to look at (typically a few per cent).
 * d_results: n hc_result_rec on the device; d_indices: room for n uint32; d_count: one uint64 on the
 * device.  Asynchronous on `hip_stream` (NULL = the context's stream). */
int hc_compact_device(hc_ctx* ctx, const void* d_results, uint64_t n, void* d_indices, void* d_count, void* hip_stream);

/* The payload of the multi-GPU collection: for k < min(*d_count, cap),
 *   d_rows[k] = { uint64 base_index + d_indices[k]; double x1; double x2; uint32 mm; uint32 n_cls }   (32 bytes)
 * i.e. the compacted records of this rank's shard tagged with their global candidate index, ready for one
 * all-gather.  Rows at and beyond *d_count are left untouched.  Asynchronous on `hip_stream`. */
/* (hc_gather_row: defined with hc_block above) */
int hc_pack_rows_device(hc_ctx* ctx, const void* d_results, const void* d_indices, const void* d_count, uint64_t cap,
                        uint64_t base_index, void* d_rows, void* hip_stream);

/* ---- candidate generation ----------------------------------------------------------------
 * All suffix–prefix overlaps and inclusions between the sequences of the read store under a Hamming error rate:
 * the job the pipelines hand to the external tool `rust-overlaps -i -r <fasta> <out> <err_rate> <min_overlap>`
 * (savage.py:664,713; polyte.py:514,542), whose 8-column SFO records scripts/sfo2overlaps.py:36 reads:
 *      idA idB N|I OHA OHB OLA OLB K
 * Sequence ids are the SFO ids: singles 0..S-1, /1 mates S..S+P-1, /2 mates S+P..S+2P-1 (the s_p1_p2.fasta order;
 * the read set must list its single-end reads first, as FastqStorage does).  With B placed at offset d in A's
 * coordinates (B reverse-complemented when inverted): OHA = d, OHB = d + len(B) - len(A), OLA = OLB = L = length of
 * the common stretch, K = its mismatches.  Reported: every pair idA < idB, orientation and d with L >= min_overlap
 * and K <= floor(err_rate * L); a non-ACGT symbol matches nothing.  Exact (a seed filter with a pigeonhole
 * guarantee, then full verification), sorted by (idA, idB, orientation, d), no duplicates.
 * rust-overlaps itself is not part of the reference tree: its output is NOT available to compare with here. */
#define HC_FIND_REVERSALS  0x1u /* rust-overlaps -r: also B reverse-complemented ("I" records) */
#define HC_FIND_INCLUSIONS 0x2u /* rust-overlaps -i: also report a sequence lying entirely inside the other */
#define HC_FIND_RECOMPUTE  0x4u /* ignore the records kept from an identical earlier call (timing runs) */
typedef struct hc_sfo_rec {
    uint32_t idA, idB;
    int32_t OHA, OHB;
    uint32_t OLA, OLB, K;
    uint32_t inverted; /* 0 = "N", 1 = "I" */
} hc_sfo_rec; /* 32 bytes */
/* Runs on the device against the store of hc_set_reads.  *n_out = number of records found; the first
 * min(cap, *n_out) are copied to out (host memory; may be NULL when cap == 0).  The records of the last call stay on
 * the device until the next hc_set_reads / hc_find_overlaps with other parameters: asking for the count first and
 * fetching with a second, identical call computes once. */
int hc_find_overlaps(hc_ctx* ctx, double err_rate, uint32_t min_overlap, uint32_t flags, hc_sfo_rec* out, uint64_t cap,
                     uint64_t* n_out);
/* The SFO ingest (scripts/sfo2overlaps.py --in --out --num_singles --num_pairs; hcedge_host.h: hc_sfo_records_to_overlaps)
 * straight from the records the last hc_find_overlaps left on the device: the script's flip (:112-122) and its
 * `sort -k1,1n -k2,2n -k3,3n -k4,4n` (whole line as last resort, LC_ALL=C) run there as three radix sorts over a 192-bit
 * key, the sorted records come back once, the host threads match the paired candidates and write the 13-column overlaps
 * file.  Same bytes as hc_host_write_sfo + hc_sfo2overlaps on those records.  *n_lines = overlap lines written. */
int hc_found_to_overlaps(hc_ctx* ctx, const char* out_path, uint64_t num_singles, uint64_t num_pairs, uint64_t* n_lines);
/* The same ingest with nothing leaving the device (round 6): the script's flip, sort, matching (scripts/sfo2overlaps.py:63-103,150-329) and
 * both its `uniq`s on the records where hc_find_overlaps left them; *d_lines = the overlaps file's lines as hc_line_rec (hcedge.h below) in
 * DEVICE memory, in file order — what hc_textblock_submit_lines takes.  Valid until the next call or hc_set_reads.  HC_ERR_NOT_ON_DEVICE where
 * the device cannot decide (ids / numbers outside its sort keys, an assert of the script, thousands of lines for one pair of reads): the caller
 * takes hc_found_to_overlaps, which raises what the script raises.  Candidate generation itself: parity unpinned (rust-overlaps is absent). */
/* SFO records from elsewhere — a rust-overlaps output the caller has parsed (8 columns: /root/reference/scripts/sfo2overlaps.py:35-36) — in
 * the place of the finder's: hc_found_to_overlaps / hc_found_to_lines_device then run the ingest on them.  After hc_set_reads. */
int hc_set_found_records(hc_ctx* ctx, const hc_sfo_rec* recs, uint64_t n);
/* The same from the SFO FILE's text, read on the device (64 MiB chunks, one lane per line): a canonical file only — eight fields, single tabs,
 * "0" or [1-9][0-9]* with one '-' allowed in front of the two overhangs, `N` or `I`: what rust-overlaps writes.  Any other text:
 * HC_ERR_NOT_ON_DEVICE (hc_sfo2overlaps' general path takes such a file and owns its errors).  `text` is host memory as it is (a mapping of
 * the file will do). */
int hc_set_found_from_sfo_text(hc_ctx* ctx, const char* text, uint64_t n_bytes, uint64_t* n_records);
struct hc_line_rec;
/* n lines of hc_found_to_lines_device copied to the host */
int hc_found_lines_fetch(hc_ctx* ctx, const struct hc_line_rec* d_lines, uint64_t n, struct hc_line_rec* out);
int hc_found_to_lines_device(hc_ctx* ctx, uint64_t num_singles, uint64_t num_pairs, const struct hc_line_rec** d_lines, uint64_t* n_lines);

/* hc_compact_device + hc_pack_rows_device in one call, with the count travelling inside the payload: d_payload is
 * (cap + 1) rows; row 0 = { index = *d_count, x1 = x2 = 0, mm = 0, n_cls = 0 }, rows 1.. as hc_pack_rows_device writes
 * them.  One all-gather of this buffer is the whole all-gather-v.  d_indices: room for n uint32; d_count: one uint64. */
int hc_compact_pack_device(hc_ctx* ctx, const void* d_results, uint64_t n, void* d_indices, void* d_count, uint64_t cap,
                           uint64_t base_index, void* d_payload, void* hip_stream);

/* Scoring and collection payload in ONE kernel: like hc_score_batch_device, and every record that is not dropped is also
 * appended (tagged with base_index + its position) to d_payload in the layout of hc_compact_pack_device — row 0 counts
 * them — by the scoring kernel itself: no compaction pass over the results.  The rows arrive in no particular order
 * (sort by index if sequence order matters); the count is the number of kept rows exactly, so a count above cap means —
 * and only means — that there were more kept rows than cap and the surplus was dropped: rerun with a larger cap.
 * rec_fmt: HC_REC_FULL (d_in holds hc_overlap_rec) or HC_REC_COMPACT (hc_cand_rec).  Every read set takes this path.
 * Streams: the device entry points (this one, hc_score_batch_device, hc_score_cands_device, hc_score_compact_device) may be given
 * different streams on one context; launches that need the context's scratch (row collection, hc_set_reorder, length-bucketed
 * read sets) are then ordered behind one another on the device, others run side by side.  Calls on one context come from
 * one host thread at a time. */
int hc_score_pack_device(hc_ctx* ctx, uint32_t rec_fmt, const void* d_in, uint64_t n, void* d_out, uint64_t cap, uint64_t base_index,
                         void* d_payload, void* hip_stream);

/* The 24-byte form of a collection payload, for the multi-GPU exchange (three quarters of the bytes over every link).  d_payload: (cap + 1)
 * rows of 32 bytes as hc_score_pack_device / hc_compact_pack_device leave them.  d_payload24: (cap + 1) rows of three 64-bit words —
 * row 0 = { count, number of rows that did not fit, 0 }, row k = { x1 bits, x2 bits, index | mm << 32 | n << 46 | class << 60 }.  A row
 * fits when its index < 2^32 and mm, n < 2^14; a caller whose read set allows more overlapped positions, or whose job has 2^32 candidates
 * and more, exchanges the 32-byte rows.  No reference analogue (the reference is one process: src/ViralQuasispecies.cpp:279-283). */
int hc_narrow_payload_device(hc_ctx* ctx, const void* d_payload, uint64_t cap, void* d_payload24, void* hip_stream);

/* The multi-GPU step: the per-step exchange of the kept rows runs on a side stream BESIDE the next step's scoring
 * kernel, and the scoring kernel's large launches hold one workgroup per CU for the launch's whole duration (145 KiB of LDS, 480 of a
 * SIMD's 512 registers): a collective library's kernels find no room on a CU that holds one.  hc_set_comm_reserve(cus): launches of
 * this context leave `cus` CUs without a workgroup (0: none, the default) — the exchange's kernels run there.  hc_comm_gate_device:
 * enqueues on `hip_stream` (the exchange's stream) a one-lane kernel that returns once every workgroup of the hc_score_pack_device
 * launches enqueued on this context SO FAR has started (or after timeout_us; 0 = 2 000): put in front of the collective, it makes
 * the collective's kernels arrive when the scoring kernel beside them already sits on its CUs, so they take the free ones instead
 * of being spread over CUs the scoring workgroups then cannot use.  No reference analogue (single process:
 * src/ViralQuasispecies.cpp:279-283).  Without a reserve nothing is counted and the gate returns at once. */
int hc_set_comm_reserve(hc_ctx* ctx, uint32_t cus);
int hc_comm_gate_device(hc_ctx* ctx, void* hip_stream, uint32_t timeout_us);

/* hc_score_batch + compaction in one call for host callers: scores `in` on the device and copies back
 * only the non-DROP records: idx_out[k] (ascending) and res_out[k] = result of in[idx_out[k]].
 * cap = capacity of idx_out / res_out in records; *n_out = number of non-DROP records (if it exceeds
 * cap, HC_ERR_ARG is returned and nothing is copied). */
int hc_score_batch_compact(hc_ctx* ctx, const hc_overlap_rec* in, uint64_t n, uint32_t* idx_out, hc_result_rec* res_out,
                           uint64_t cap, uint64_t* n_out);

/* Times `iters` back-to-back launches of the scoring kernel on the context's own
 * stream with hipEvents recorded on THAT stream; returns the mean milliseconds
 * per launch.  Used for the roofline figure.  rec_fmt as for hc_score_pack_device. */
int hc_time_score_kernel(hc_ctx* ctx, uint32_t rec_fmt, const void* d_in, uint64_t n, void* d_out, int iters, float* ms_per_launch);

/* Sum over the batch of the overlapped positions sum_sub L_sub,
 * L_sub = min(L1 - pos, L2) (EdgeCalculator.cpp:86-88), computed on the device:
 * the exact multiplier for the algorithmic-bytes figure 32 + 16 + 4*L_sub. */
int hc_count_positions_device(hc_ctx* ctx, uint32_t rec_fmt, const void* d_in, uint64_t n, uint64_t* total_positions,
                              uint64_t* total_subs);

/* The scoring kernel hc_set_reads chose for the read set, as text: the kernel's symbol (template arguments included: what
 * rocprofv3's kernel trace lists), the symbol encoding, the log table's size and whether launches bucket their candidates
 * by length first (read sets of mixed sequence length).  Diagnostics: tests and bench.py name what they measured with it. */
int hc_get_kernel_info(hc_ctx* ctx, char* buf, uint32_t cap);
/* The same for a launch of n candidates: the library picks the kernel's form by the launch's size too (register-staged rows for
 * small launches, LDS-DMA rows from 5 * 10^5 candidates on, the waves' work queue where every wave gets 64 steps and more). */
int hc_get_kernel_info_for(hc_ctx* ctx, uint64_t n, char* buf, uint32_t cap);

/* ---- device primitives (csrc/hc_prims.hip), behind host-buffer entry points for their unit tests ------------------
 * The stage's device side sorts and compacts with its own kernels: a stable LSD radix sort (8 bits a pass; keys of 4 bytes
 * with 4-byte values, keys of 8 bytes with values of 4, 8 or 0 bytes), an exclusive prefix sum (elements of 4 or 8
 * bytes), the indices of the set flags in ascending order, and the first element of every run of equal neighbours.
 * Synchronous, host buffers in and out, n < 2^32; the sort orders by bits [begin_bit, end_bit) of the key. */
int hc_dev_radix_sort(uint32_t key_bytes, uint32_t val_bytes, const void* keys, const void* vals, void* keys_out, void* vals_out, uint64_t n,
                      int begin_bit, int end_bit);
int hc_dev_exclusive_sum(uint32_t elem_bytes, const void* in, void* out, uint64_t n);
int hc_dev_select_flagged(const uint8_t* flags, uint64_t n, uint32_t* idx_out, uint64_t* count);
int hc_dev_unique_u64(const uint64_t* in, uint64_t n, uint64_t* out, uint64_t* count);

/* Host finalisation of one record with the host libm exp(): score as the
 * reference returns it (EdgeCalculator.cpp:137-138, 254-261), mismatch_rate
 * (EdgeCalculator.cpp:132) and the class, AMBIG resolved. Pure function. */
int hc_finalize(const hc_settings* s, const hc_result_rec* r, double* score, double* mismatch_rate, uint32_t* cls);

/* hc_finalize over n records; any of the three output arrays may be NULL.  Returns
 * HC_ERR_DATA (after filling everything) if some record has class HC_CLS_ERROR. */
int hc_finalize_batch(const hc_settings* s, const hc_result_rec* r, uint64_t n, double* score, double* mismatch_rate,
                      uint32_t* cls);

/* ---- the overlaps file read on the device ("mmap + SIMD/GPU tokenizer") ----------------------
 * construct_edges spends its time in the serial text parser (EdgeCalculator.cpp:581-604: getline + stringstream + 13
 * strings per line); here a block of the file's TEXT goes to the device as it is and three byte kernels do what the
 * reference does with every line before process_overlaps: split it at its newlines, read the 13 fields of a PLAIN
 * line (single tabs, decimal numbers or "-", valid one-character fields: what scripts/sfo2overlaps.py and FNO write),
 * honour --max_ov, drop self overlaps (:605-607), apply the prefilter (:612-635), look the two ids up
 * (m_ID_to_index, :164-171) — and the scoring kernel runs on the result in the same stream.  What comes back is what
 * an hc_block returns, each row together with the parsed line it came from (the host needs LEN / PERC / the text of a
 * non-edge), plus the prefilter's rejects and the line counters.  A block holding a line that is not plain (padded,
 * malformed, hexadecimal ids, an id that is not in the FASTQ input, ...) is REPORTED, not guessed: the host then takes
 * that block through its own tokeniser and Overlap constructor, which own every error the reference can raise. */
typedef struct hc_line_rec { /* one parsed line: Overlap (src/Overlap.h:20-73) */
    uint64_t id1, id2;
    uint32_t pos1, pos2, perc1, perc2, len1, len2;
    uint8_t ord, ori1, ori2, type1, type2, pad[3]; /* the characters of the file: '1' '2' '-', '+' '-', 's' 'p' */
} hc_line_rec; /* 48 bytes */
typedef struct hc_text_row {
    hc_gather_row row; /* index = base_index + number of the line in the block */
    hc_line_rec line;
} hc_text_row; /* 80 bytes */
typedef struct hc_text_reject {
    uint32_t line_index, pad; /* number of the line in the block */
    hc_line_rec line;
} hc_text_reject; /* 56 bytes */
typedef struct hc_text_nonplain { /* a line the device does not read (not of the plain form): the host's tokeniser owns it */
    uint32_t line_index; /* number of the line in the block */
    uint32_t begin;      /* offset of its first byte in the block's text */
    uint32_t length;     /* bytes without the newline */
    uint32_t pad;
} hc_text_nonplain; /* 16 bytes */
typedef struct hc_text_result {
    uint64_t n_lines;          /* lines in the block (a last piece without a newline is a line)              */
    uint64_t lines_read;       /* of them below --max_ov                                                      */
    uint64_t needs_host;       /* != 0: lines the device does not read (n_nonplain), an unknown id, more lines than the
                                  block has room for, or more surviving records / rejects than its host buffers
                                  hold (an eighth of the lines): nothing else of this result may be used     */
    uint64_t n_nonplain, n_unknown_id;
    uint64_t self_overlaps, silently_dropped, prefilter_rejected, scored; /* the counters of construct_edges  */
    const hc_text_row* rows;   /* non-dropped records, sorted by index                                        */
    uint64_t n_rows;
    const hc_text_reject* rejected; /* sorted by line_index (the order they are written to nonedge_overlaps.txt) */
    uint64_t n_rejected;
    /* hc_textblock_list_nonplain: the n_nonplain lines the device did not read, sorted by line_index — the caller tokenises them
     * (src/EdgeCalculator.cpp:584-604), scores the ones that pass and splices rows, rejects and counters in at their places; every
     * other field of the result describes the OTHER lines (lines_read counts these lines too: they are below --max_ov).  NULL / 0
     * without a list; more such lines than the list holds: needs_host, as without one. */
    const hc_text_nonplain* nonplain;
    uint64_t n_nonplain_listed;
} hc_text_result;
typedef struct hc_textblock hc_textblock;
/* read_ids[r] = id of m_read_vec[r]; the first occurrence of an id wins (std::map::insert, FastqStorage.h:88-97). */
int hc_text_set_ids(hc_ctx* ctx, const uint64_t* read_ids, uint32_t n_reads);
/* hc_textblock_create reads nothing of the context but its device number and changes nothing of it (streams, events and
 * allocations of the block's own; errors go to the calling thread's hc_last_error): it may run on a second host thread beside
 * hc_set_reads / hc_text_set_ids on the same context (the stage's constructor does).  hc_textblock_submit hands the copy streams
 * of the context round: submits on one context come from one host thread at a time. */
int hc_textblock_create(hc_ctx* ctx, uint64_t max_bytes, hc_textblock** out);
/* The block's page-locked text buffer (max_bytes): the caller reads the file straight into it. */
char* hc_textblock_buffer(hc_textblock* b);
/* Asynchronous.  The first n_bytes of the buffer are one block of the file starting at a line start; first_line_no =
 * number of that line in the file; --max_ov, the prefilter settings and the flags come from the context's settings. */
int hc_textblock_submit(hc_textblock* b, uint64_t n_bytes, uint64_t first_line_no, uint64_t base_index);
/* The same without the caller copying or counting anything: `text` is host memory as it is — a mapping of the overlaps
 * file will do, page-locked or not — and the number of the block's first line comes from a chain of counters the blocks of
 * one file share: block k (0, 1, 2 ... in file order, one submit each) reads entry k, which block k - 1's launch sequence
 * wrote, and writes entry k + 1.  `prev` = the block object block k - 1 was submitted on (NULL for k == 0); it may belong
 * to another context / device.  hc_text_result.n_lines of every block tells the host the same numbers afterwards. */
typedef struct hc_linechain hc_linechain;
int hc_linechain_create(hc_ctx* ctx, uint64_t n_blocks, hc_linechain** out);
int hc_linechain_destroy(hc_linechain* chain);
int hc_textblock_submit_from(hc_textblock* b, const void* text, uint64_t n_bytes, hc_linechain* chain, uint64_t k, hc_textblock* prev,
                             uint64_t base_index);
/* Parsed lines instead of text (round 6: the device-resident stage a, hc_found_to_lines_device — no text exists): d_lines = n_lines
 * hc_line_rec in DEVICE memory of the block's device, in file order, numbered first_line_no... (--max_ov).  Everything behind the parse is the
 * text route's own: self overlaps, prefilter, id look-up (src/EdgeCalculator.cpp:605-635), scoring, rows.  n_lines <= hc_textblock_max_lines. */
int hc_textblock_submit_lines(hc_textblock* b, const hc_line_rec* d_lines, uint64_t n_lines, uint64_t first_line_no, uint64_t base_index);
uint64_t hc_textblock_max_lines(hc_textblock* b);
int hc_textblock_wait(hc_textblock* b, hc_text_result* out); /* valid until the next submit on this block */
/* Per-LINE fallback (round 5; the reference reads every line by itself, src/EdgeCalculator.cpp:581-604): with a list of `max_lines`
 * entries (0: none, the default) a block that holds up to that many lines which are not plain does NOT go to the host as a whole —
 * hc_text_result.nonplain names them and the caller handles just those.  Not while the block is in flight. */
int hc_textblock_list_nonplain(hc_textblock* b, uint32_t max_lines);
/* A block starts with row buffers for an eighth of its lines; a block with more surviving records (or prefilter rejects)
 * grows them inside hc_textblock_wait and runs its device half again — it does not go to the host.  How often so far: */
uint64_t hc_textblock_regrown(hc_textblock* b);
/* Row buffers for at least `rows` rows (at most the block's lines) ahead of a submit, for a caller that expects most lines to survive
 * (overlaps straight from hc_find_overlaps: the one-call route grows its blocks while the finder runs).  Not while the block is in
 * flight.  The buffers it replaces are freed with the block. */
int hc_textblock_reserve_rows(hc_textblock* b, uint64_t rows);
int hc_textblock_destroy(hc_textblock* b);

/* ---- duplicate resolution + adjacency on the device -----------------------------------------
 * The serial half of process_overlaps (EdgeCalculator.cpp:431-545) for a whole overlaps file at once, into an EMPTY
 * graph: the admitted candidates, in sequence order, become Edges (the tail of compute_overlap: pos3/pos4, lengths,
 * vertices, :219-232, :254-270, :292-308, :353-379), are normalised (:443-448), grouped by graph slot — unordered
 * vertex pair + (ori1 == ori2) — with a stable radix sort, and every slot replays the reference's replace / keep
 * decisions in sequence order (score, then the tie-break chain :470-521); the survivors come back as adjacency
 * lists, in the order the reference's addEdge calls leave them (HC_GRAPH_INSERTION_ORDER) or as
 * OverlapGraph::sortEdges (OverlapGraph.cpp:722-764) would re-order them right afterwards (HC_GRAPH_SORTED).
 * One Edge of OverlapGraph::adj_out (src/Edge.h:21-38), flattened; read1/read2 index m_read_vec. */
typedef struct hc_edge_rec {
    double score, mismatch_rate;
    int32_t pos1, pos2, pos3, pos4;
    uint8_t ori1, ori2, ord, pad;
    uint32_t read1, read2;
    uint64_t v1, v2;
    int32_t perc, len0, len1, len2;
} hc_edge_rec; /* 80 bytes */

/* An admitted candidate as the host hands it to the device: the record's own columns (the Edge takes POS, ORI, ORD,
 * PERC, LEN from the file, not from the geometry) + the host-finalised score (exp() is the host libm's)
 * + the mismatch ratio of the result record. */
typedef struct hc_admit_rec {
    double score;        /* hc_finalize                                       */
    uint32_t read1, read2;
    uint32_t pos1, pos2;
    uint32_t mm, n;      /* mismatch_rate = float(mm) / n, EdgeCalculator.cpp:132 */
    uint32_t len1, len2; /* LEN1, LEN2 columns                                */
    uint32_t perc;       /* Overlap::get_perc()                               */
    uint8_t ori1, ori2, ord, pad;
} hc_admit_rec; /* 48 bytes */

#define HC_GRAPH_INSERTION_ORDER 0u
#define HC_GRAPH_SORTED 1u /* as after OverlapGraph::sortEdges */
typedef struct hc_graph_counts {
    uint64_t n_admitted, n_edges; /* n_edges = occupied slots = survivors                              */
    uint64_t inclusion_count;     /* perc == 100 among the admitted, before de-duplication (:449-451)  */
    uint64_t dup_count;           /* `doubles`, :472,537                                               */
    uint64_t n_tied_lists;        /* HC_GRAPH_SORTED: out-lists longer than 16 that hold fully tied edges —
                                     std::sort's order of those depends on the insertion order; hc_graph_fetch
                                     reports them (tied_vertices) so the host can re-sort exactly those */
    int64_t first_bad;            /* index of the first admitted record the reference's Edge would reject
                                     (Edge::set_len, src/Edge.h:211-218), -1 = none                      */
} hc_graph_counts;
/* The admitted records may also be handed over piecewise while the file is still being scored (the stage does: the
 * copies hide behind the scoring of later blocks): hc_graph_begin forgets what was appended, hc_graph_append copies n
 * more records (sequence order = order of the calls) behind them; hc_graph_resolve with admitted == NULL resolves the
 * appended records (n must equal their number). */
int hc_graph_begin(hc_ctx* ctx);
int hc_graph_append(hc_ctx* ctx, const hc_admit_rec* admitted, uint64_t n);
/* vertex_of_read: n_reads vertex ids (Read::get_vertex_id(true)), NULL = identity; every id < n_vertices < 2^31.
 * admitted: host memory, sequence order (or NULL, see above).  Synchronous.  Results stay on the device until
 * hc_graph_fetch. */
int hc_graph_resolve(hc_ctx* ctx, const hc_admit_rec* admitted, uint64_t n, uint64_t n_vertices, const uint32_t* vertex_of_read,
                     uint32_t order, hc_graph_counts* counts);
/* edges: n_edges records, adj_out lists back to back in vertex order; out_off: n_vertices + 1 offsets into them;
 * in_nodes / in_off: adj_in the same way (vertex1 of every in-edge); seq: for every edge, the index of its admitted
 * record (its place in the reference's insertion sequence); inclusions: n_vertices bytes (OverlapGraph::inclusions,
 * only marked with HC_FLAG_IGNORE_INCLUSIONS); tied_vertices: room for counts.n_tied_lists ids.  Any may be NULL. */
int hc_graph_fetch(hc_ctx* ctx, hc_edge_rec* edges, uint64_t* out_off, uint32_t* in_nodes, uint64_t* in_off, uint32_t* seq,
                   uint8_t* inclusions, uint32_t* tied_vertices);
/* The edges of hc_graph_fetch in pieces: records [first, first + count) of the resolved graph's edge array into dst.  Synchronous.
 * The stage fetches the small arrays first (hc_graph_fetch with edges == NULL) and then the edges piece by piece, while its host
 * threads turn the pieces that have arrived into the graph's lists. */
int hc_graph_fetch_edges(hc_ctx* ctx, uint64_t first, uint64_t count, hc_edge_rec* dst);

/* PCI bus id of a device ("0000:c1:00.0"), for callers that place the host threads feeding it on its NUMA node
 * (/sys/bus/pci/devices/<id>/numa_node); the stage does (HC_NUMA=0 turns that off). */
int hc_device_bus_id(int32_t device, char* bus_id, uint32_t cap);

/* Introspection used by the tests: quality alphabet size K of the current store,
 * and the x-space guard band [lo, hi] of a threshold (x <= lo fails, x > hi passes). */
int hc_get_info(hc_ctx* ctx, uint32_t* qual_alphabet, uint64_t* store_bytes, double* x_edge_lo,
                double* x_edge_hi, double* x_ov_lo, double* x_ov_hi);

#ifdef __cplusplus
}
#endif
#endif /* HCEDGE_H_ */
