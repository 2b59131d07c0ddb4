/*
 * hc_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the reference algorithm for the edge-calculation hot
 * path of HaploConduct (src/EdgeCalculator.cpp, src/Overlap.h, src/Edge.h,
 * src/Read.h, src/Types.h, src/OverlapGraph.cpp).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * product library (libhcedge.so) never links or calls anything in oracle/.
 *
 * PINNING STATUS (see DESIGN.md "Oracle"):
 *   - The reference ships no tests / golden vectors for this path, and none of its translation
 *     units compiles here as a whole (Boost headers are absent; stand-ins are not allowed).
 *   - Boost is used in very few lines, though.  Everything else is executed AS IT IS through fragment probes
 *     (oracle/Makefile `ref`: reference lines piped verbatim into the compiler behind class shells that repeat
 *     declarations only; outputs in oracle/_ref, vectors + generating scripts in tests/golden/):
 *       * Types.h, Overlap.h, Read.h, Edge.h      whole headers                       (libhcref_headers.so)
 *       * EdgeCalculator.cpp:26-139                score / phred_to_prob / overlap_score (libhcref_fragment.so)
 *       * EdgeCalculator.cpp:26-557 + OverlapGraph.cpp:83-101,150-229,285-319
 *                                                  compute_overlap, process_overlaps incl. the serial insert and
 *                                                  the tie-break chain, the graph methods  (libhcref_edgecalc.so)
 *     tests/test_oracle_golden.py and tests/test_ec_golden.py hold this oracle to those outputs bit for bit.
 *   - "PARITY UNPINNED" (restated, no reference execution possible): construct_edges' line trimming / splitting
 *     and prefilter (EdgeCalculator.cpp:561-666, Boost at :584-587) and the FASTQ reader (FastqStorage.cpp).
 *
 * Every function cites the reference file:line it follows.
 */
#ifndef HC_ORACLE_H_
#define HC_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same layouts as include/hcedge.h (kept separate on purpose: the oracle must
 * not depend on product headers). */
typedef struct hco_settings {
    double edge_threshold, ov_threshold, merge_contigs, mismatch;
    uint32_t min_read_len, min_overlap_len, min_overlap_perc, flags;
    uint64_t max_overlaps;
    int32_t device;
    uint32_t n_threads;
} hco_settings;

#define HCO_FLAG_ADD_DUPLICATES 0x1u
#define HCO_FLAG_RESOLVE_ORIENTATIONS 0x2u
#define HCO_FLAG_IGNORE_INCLUSIONS 0x4u
#define HCO_FLAG_RELAX_PE_EDGES 0x8u
#define HCO_FLAG_ALLOW_SPACES 0x10u
#define HCO_FLAG_VERBOSE 0x20u

typedef struct hco_overlap {
    uint32_t read1, read2, pos1, pos2;
    uint8_t ori1, ori2, ord, flags;
    uint32_t len1, len2, perc;
} hco_overlap;

typedef struct hco_reads {
    const uint8_t* bases;
    const uint8_t* quals;
    const uint64_t* seq_off;        /* n_seq + 1 */
    const uint32_t* read_first_seq; /* n_reads + 1 */
    uint32_t n_reads;
} hco_reads;

/* What compute_overlap returns (an Edge, Edge.h:18-53) plus the intermediates
 * the parity tests compare against the device records. */
typedef struct hco_edge {
    double score;         /* Edge::score */
    double mismatch_rate; /* Edge::mismatch_rate */
    double x1, x2;        /* (1.0/total_len)*total_score per sub-overlap; -inf if it returned 0; x2 = NaN for s-s */
    double ov1, ov2;      /* overlap_score() return values */
    uint32_t mm, n;       /* of the sub-overlap with the larger mismatch rate; 1/1 for early exits */
    int32_t pos3, pos4;   /* Edge::pos3 / pos4 */
    uint32_t cls;         /* 0 drop, 1 non-edge, 2 edge (score), 3 edge (merge_contigs) */
    uint32_t n_subs;
    uint64_t positions;   /* sum of L over the sub-overlaps (for the algorithmic-bytes figure) */
    int32_t status;       /* 0, or <0 where the reference would assert/abort */
} hco_edge;

/* EdgeCalculator::phred_to_prob, EdgeCalculator.cpp:59-63 */
double hco_phred_to_prob(int phred);

/* EdgeCalculator::score, EdgeCalculator.cpp:26-56.
 * Returns log(p) (<=0), 1.0 for an N position, 2.0 for p < mismatch_setting;
 * -1000 if a base is not in ACGTN (the reference asserts). */
double hco_score(char nt1, char nt2, double p1, double p2, int* mismatch_count, double mismatch_setting);

/* EdgeCalculator::overlap_score, EdgeCalculator.cpp:67-139.
 * total_score/total_len/mismatch_count are also returned (may be NULL). */
double hco_overlap_score(const char* seq1, size_t len1, const char* seq2, size_t len2, const char* q1,
                         const char* q2, unsigned int pos, unsigned int min_read_len, double mismatch_setting,
                         double* mismatch_rate, double* x_out, uint32_t* mm_out, uint32_t* n_out,
                         uint64_t* positions_out, int* status);

/* build_rev_comp, Types.h:109-129; returns 0 or -1 on an invalid character. */
int hco_build_rev_comp(const char* seq, size_t len, char* out);

/* EdgeCalculator::compute_overlap (EdgeCalculator.cpp:143-385) followed by the
 * 3-way classification of process_overlaps (EdgeCalculator.cpp:404-413). */
int hco_compute_overlap(const hco_reads* reads, const hco_settings* s, const hco_overlap* ov, hco_edge* out);

/* The omp-for of process_overlaps over a batch (EdgeCalculator.cpp:395-414);
 * n_threads > 1 uses OpenMP like the reference.  Returns 0. */
int hco_score_batch(const hco_reads* reads, const hco_settings* s, const hco_overlap* in, uint64_t n,
                    hco_edge* out, int n_threads);

/* ---- overlaps-file record: Overlap.h:39-73 (constructor from 13 strings) ---- */
typedef struct hco_overlap_line {
    unsigned long id1, id2;
    unsigned int pos1, pos2;
    char ord, ori1, ori2, type1, type2;
    unsigned int perc1, perc2, len1, len2;
} hco_overlap_line;

/* Overlap(std::vector<std::string>) — returns 0, or <0 where the reference exits/asserts. */
int hco_overlap_from_fields(const char* const fields[13], hco_overlap_line* out);
/* Overlap::get_perc, Overlap.h:203-210 */
unsigned int hco_overlap_get_perc(const hco_overlap_line* o);
/* Overlap::get_overlap_line, Overlap.h:234-237; returns length written (buf >= 256). */
int hco_overlap_get_line(const hco_overlap_line* o, char* buf, size_t bufsz);
/* The line tokenizer of construct_edges, EdgeCalculator.cpp:584-603: trims "\t ",
 * splits on '\t' (or on "\t " with compression if allow_spaces).  fields[] point
 * into `scratch` (a modifiable copy of the line).  Returns the number of fields. */
int hco_split_line(char* scratch, int allow_spaces, char* fields[], int max_fields);

/* ---- graph side: process_overlaps' serial insert, EdgeCalculator.cpp:431-545,
 * over OverlapGraph (OverlapGraph.cpp:94-101,150-229,285-311). ---- */
typedef struct hco_gedge {
    double score, mismatch_rate;
    int32_t pos1, pos2, pos3, pos4;
    uint8_t ori1, ori2, ord, pad;
    uint32_t read1, read2;  /* index into m_read_vec */
    uint64_t v1, v2;        /* vertex1 / vertex2 */
    int32_t perc, len0, len1, len2;
} hco_gedge;

typedef struct hco_graph hco_graph;
hco_graph* hco_graph_new(uint64_t n_vertices);
void hco_graph_free(hco_graph* g);
uint64_t hco_graph_edge_count(const hco_graph* g);
/* Number of out-edges of v, and a pointer to them in list order (valid until the next insert). */
uint64_t hco_graph_out(const hco_graph* g, uint64_t v, const hco_gedge** edges);
int hco_graph_inclusion(const hco_graph* g, uint64_t v);

typedef struct hco_counters {
    uint64_t self_overlap_count, inclusion_count, dup_count;
    uint64_t edges_added, nonedges_written, prefilter_rejected, malformed_lines, lines_read, scored;
} hco_counters;

/* Insert one admitted edge exactly as EdgeCalculator.cpp:441-538 does. */
int hco_graph_insert(hco_graph* g, const hco_settings* s, hco_gedge* e, hco_counters* c);
/* OverlapGraph::addEquivalentEdges, OverlapGraph.cpp:608-719 */
int hco_graph_add_equivalent_edges(hco_graph* g, uint32_t n_reads);

/* EdgeCalculator::construct_edges, EdgeCalculator.cpp:561-666, single thread.
 * read_ids[i] is the read id of m_read_vec[i] (id->index as FastqStorage.h:88-97:
 * first occurrence wins); nonedge_path may be NULL (no file written).
 * Vertices are the m_read_vec indices; with HCO_FLAG_ADD_DUPLICATES the graph needs 2 * n_reads vertices (read r
 * reverse-complemented = vertex n_reads + r) and OverlapGraph::addEquivalentEdges runs at the end (:650-652). */
int hco_construct_edges(const hco_reads* reads, const unsigned long* read_ids, const hco_settings* s,
                        const char* overlaps_path, const char* nonedge_path, hco_graph* g, hco_counters* c);

#ifdef __cplusplus
}
#endif
#endif
