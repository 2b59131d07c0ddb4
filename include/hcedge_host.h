/*
 * hcedge_host.h — C ABI of the host side of the edge-calculation stage in libhcedge.so:
 * the mirror of the reference classes FastqStorage / OverlapGraph / EdgeCalculator
 * (haploconduct_amd/csrc/host/, C++) driven the way src/ViralQuasispecies.cpp:233-283 drives
 * them.  Everything here is plain C so that the Python tests, the CLI and a foreign caller
 * bind the same entry points.  Citations: reference file:line.
 */
#ifndef HCEDGE_HOST_H_
#define HCEDGE_HOST_H_

#include "hcedge.h"

#ifdef __cplusplus
extern "C" {
#endif

/* File arguments of the stage: --singles/--paired1/--paired2/--IDs/--overlaps/--output/--max_reads
 * (src/ViralQuasispecies.cpp:52-59).  NULL or "" = absent; "None" = absent for the FASTQ files
 * (src/FastqStorage.h:71,75). */
typedef struct hc_ec_paths {
    const char* singles_file;
    const char* paired1_file;
    const char* paired2_file;
    const char* id_correspondence;
    const char* overlaps_file;
    const char* output_dir;
    uint64_t max_reads;
} hc_ec_paths;

/* hc_edge_rec (one Edge of OverlapGraph::adj_out, flattened): include/hcedge.h */

typedef struct hc_ec_counters {
    uint64_t self_overlap_count, inclusion_count, dup_count; /* src/EdgeCalculator.h:40-42 */
    uint64_t edges_added, nonedges_written, prefilter_rejected, malformed_lines, lines_read, scored;
    uint64_t ambiguous, silently_dropped;
    double t_parse, t_score, t_insert, t_write; /* seconds, build-owned breakdown */
    /* blocks of the overlaps file's text: parsed on the device; taken over by the host's tokeniser (a line that is not
     * plain, an id that is not in the FASTQ input, more lines than a block has room for); device blocks whose row buffers
     * had to grow, so that their device half ran twice (more than an eighth of the lines survived scoring) */
    uint64_t device_blocks, host_blocks, regrown_blocks;
    /* lines of device-parsed blocks that the host's tokeniser read ONE BY ONE (round 5, per-line fallback: a line that is not plain
     * no longer sends its block to the host) */
    uint64_t host_lines;
} hc_ec_counters;

typedef struct hc_ec hc_ec; /* FastqStorage + OverlapGraph + EdgeCalculator */

/* new FastqStorage(ps); new OverlapGraph(readcount, ...); addVertex + set_vertex_id per read;
 * EdgeCalculator(fastq, graph, ps)  — src/ViralQuasispecies.cpp:233-279.  Needs a HIP device.
 * A process's first open of a device with 64 MiB of FASTQ and more starts the HIP runtime and loads the kernels on a thread beside
 * the FASTQ parsing (HC_WARM=0 / 1: never / always). */
int hc_ec_open(hc_ec** out, const hc_settings* settings, const hc_ec_paths* paths);
/* EdgeCalculator::construct_edges() — src/EdgeCalculator.cpp:561-666 */
int hc_ec_construct_edges(hc_ec* ec);
/* construct_edges() + OverlapGraph::sortEdges() (src/ViralQuasispecies.cpp:281,297: what every workflow calls next) as
 * one call: into an empty graph the adjacency lists come back from the device already in sortEdges order. */
int hc_ec_construct_edges_sorted(hc_ec* ec);
/* Reads -> graph with no files in between — the front of SAVAGE stage a (savage.py:643-717: rust-overlaps -> SFO file ->
 * scripts/sfo2overlaps.py -> overlaps.txt -> ViralQuasispecies) as one call on an open stage: hc_find_overlaps on the
 * stage's own read store (err_rate, min_overlap, HC_FIND_* flags as for rust-overlaps), the SFO ingest on the records where
 * they are (the script's flip and sort on the device, the matching on the host threads), the overlaps file's text from
 * memory into the device's text blocks, construct_edges (sorted != 0: + sortEdges).  The graph, nonedge_overlaps.txt and
 * the counters are those of writing the overlaps file (hc_found_to_overlaps) and calling hc_ec_construct_edges[_sorted]
 * on it; hc_paths.overlaps_file is not read.  *n_found: SFO records, *n_lines: overlap lines (either may be NULL). */
int hc_ec_construct_edges_from_reads(hc_ec* ec, double err_rate, uint32_t min_overlap, uint32_t find_flags, int sorted, uint64_t* n_found,
                                     uint64_t* n_lines);
/* The same call with nothing but device memory between the reads and the graph (round 6; SURVEY.md 8(f4): "would remove ... the text file
 * altogether"): the finder's records stay on the device, the SFO ingest — flip, sort, the script's matching (scripts/sfo2overlaps.py:63-103,
 * 150-329) and both its `uniq`s — runs there (hc_found_to_lines_device) and leaves the overlaps file's lines as parsed records, which the
 * stage's text blocks take as they are (hc_textblock_submit_lines); from there on it is construct_edges' own path (prefilter, scoring,
 * duplicate resolution, sortEdges order, nonedge_overlaps.txt from the kept rows).  Graph, inclusions, counters and nonedge_overlaps.txt:
 * hc_ec_construct_edges_from_reads', byte for byte.  Where the device cannot decide (an assert of the script's matching, an id or number
 * its sort keys do not hold) the call takes that route itself; *device_route (may be NULL) = 1 when the lines stayed on the device.
 * Candidate generation (hc_find_overlaps) stands in for rust-overlaps, which is not in the reference tree: parity unpinned there. */
int hc_ec_construct_edges_from_store(hc_ec* ec, double err_rate, uint32_t min_overlap, uint32_t find_flags, int sorted, uint64_t* n_found,
                                     uint64_t* n_lines, int* device_route);
/* The pipelines' own input — the SFO file `rust-overlaps` wrote (savage.py:664, polyte.py:514, 542) — straight to the graph: what
 * scripts/sfo2overlaps.py (--in sfo_path --out original_overlaps.txt --num_singles --num_pairs, savage.py:672), the overlaps file and the
 * binary's own text parser (src/EdgeCalculator.cpp:561-604) do between them.  A canonical file (eight fields, single tabs, plain decimal
 * numbers: what the tool writes) is read ON THE DEVICE (hc_set_found_from_sfo_text: 64 MiB chunks of the file's text, one lane per line) into
 * records that take the finder's place; the ingest and the stage are hc_ec_construct_edges_from_store's: no 13-column text is written, copied
 * or parsed.  Any other file, and any input the device does not
 * decide (HC_ERR_NOT_ON_DEVICE), goes through hc_sfo2overlaps' code with its text kept in memory — every error is the script's.  Same graph,
 * counters and nonedge_overlaps.txt as hc_sfo2overlaps + hc_ec_construct_edges[_sorted] on the file it writes.  Pinned end to end (the ingest by
 * the script's own outputs, the stage by the reference's own code); hc_paths.overlaps_file is not read.
 * *n_records: SFO lines read (0 on the general path), *n_lines: overlap lines, *device_route: 1 = the lines stayed on the device. */
int hc_ec_construct_edges_from_sfo(hc_ec* ec, const char* sfo_path, int sorted, uint64_t* n_records, uint64_t* n_lines, int* device_route);
/* number of device contexts the stage scores on (hc_settings.device_mask) */
uint32_t hc_ec_device_count(hc_ec* ec);
int hc_ec_get_counters(hc_ec* ec, hc_ec_counters* out);
uint64_t hc_ec_read_count(hc_ec* ec);
uint64_t hc_ec_vertex_count(hc_ec* ec); /* read count, or twice that with HC_FLAG_ADD_DUPLICATES (src/ViralQuasispecies.cpp:246-251) */
uint64_t hc_ec_edge_count(hc_ec* ec); /* OverlapGraph::getEdgeCount */
/* adj_out flattened in vertex order, each list in list order; *n_out = number of edges (may exceed cap). */
int hc_ec_get_edges(hc_ec* ec, hc_edge_rec* out, uint64_t cap, uint64_t* n_out);
int hc_ec_get_inclusions(hc_ec* ec, uint8_t* out, uint64_t cap); /* OverlapGraph::inclusions */
/* OverlapGraph::sortEdges() — src/OverlapGraph.cpp:722-764, called right after construct_edges in every workflow
 * (src/ViralQuasispecies.cpp:297,359,434): every out-list sorted by non-overlap length, then vertex2; adj_in rebuilt. */
int hc_ec_sort_edges(hc_ec* ec);
/* adj_in as offsets (vertex_count + 1) and vertex ids (edge_count), each list in list order.  Before hc_ec_sort_edges the
 * order INSIDE an in-list is that of the surviving edges' place in the insertion sequence; the reference's own order can
 * differ from it in one case — a vertex pair holding edges of both orientation classes in the same direction, one of
 * which was replaced later (removeEdgeWithOri erases the first `v` of the list, whichever edge it stood for) — and only
 * in where the other entries sit between the two equal ids.  sortEdges rebuilds every in-list (OverlapGraph.cpp:751-762),
 * after which the lists are the reference's in every case. */
int hc_ec_get_in_lists(hc_ec* ec, uint64_t* in_off, uint64_t* in_nodes, uint64_t cap);
/* EdgeCalculator::overlap_score on caller strings (src/EdgeCalculator.cpp:67-139), scored on the device. */
int hc_ec_overlap_score(hc_ec* ec, const char* seq1, const char* seq2, const char* phred1, const char* phred2,
                        uint32_t pos, double* score, double* mismatch_rate);
int hc_ec_close(hc_ec* ec);
/* A process that opens stage after stage (a pipeline's iterations in one process; the resident hc-edgecalc): with on != 0, hc_ec_close parks
 * the stage's devices — contexts, streams, scratch, text blocks with their page-locked buffers — and the next hc_ec_open on the same devices
 * takes them over (hc_reset gives the contexts the new settings; blocks of text at least as large as the new file needs).  Opening and
 * closing a stage then cost milliseconds instead of the 0.06 s + 0.1 s of making and freeing those.  on == 0 frees what is parked.
 * Results do not depend on it.  Process-wide; call it from the thread that opens and closes the stages. */
int hc_ec_keep_devices(int on);

/* ---- host-only pieces (no device needed): exercised by the CPU test-suite ---- */

/* The tokenizer of construct_edges (src/EdgeCalculator.cpp:584-597). Returns the token count;
 * off/len describe the first min(count, max_fields) tokens relative to `line`. */
int hc_host_split_line(const char* line, uint64_t n, int allow_spaces, uint32_t* off, uint32_t* len, int max_fields);

/* Overlap(std::vector<std::string>) + get_perc + get_overlap_line (src/Overlap.h:39-237) on one text
 * line.  Returns HC_OK, HC_ERR_ARG when the line does not have 13 fields, or HC_ERR_FORMAT where the
 * reference exits/asserts.  `text` (>= 192 bytes) receives the re-serialised line.
 * `allow_spaces`: bit 0 = --allow_spaced_overlaps; bit 1 = skip the one-pass reader the stage tries first on every
 * line (13 plain fields, single tabs) and go through the tokeniser + Overlap constructor steps only — both routes
 * must agree on every line, tests/test_host_logic.py. */
typedef struct hc_overlap_fields {
    uint64_t id1, id2;
    uint32_t pos1, pos2, perc1, perc2, len1, len2, perc;
    char ord, ori1, ori2, type1, type2, pad[3];
} hc_overlap_fields;
int hc_host_parse_overlap(const char* line, uint64_t n, int allow_spaces, hc_overlap_fields* out, char* text);

/* FastqStorage alone (src/FastqStorage.cpp:92-235): loads the files and exposes the flat layout
 * hc_set_reads takes.  Pointers stay valid until hc_host_fastq_free. */
typedef struct hc_fastq hc_fastq;
typedef struct hc_fastq_view {
    const uint8_t* bases;
    const uint8_t* quals;
    const uint64_t* seq_off;
    const uint32_t* read_first_seq;
    const uint64_t* read_ids; /* read id of m_read_vec[i] */
    uint32_t n_reads, n_seq, n_single, n_paired;
} hc_fastq_view;
int hc_host_fastq_load(hc_fastq** out, const hc_ec_paths* paths, hc_fastq_view* view);
int hc_host_fastq_free(hc_fastq* f);

/* Parser + prefilter of construct_edges (src/EdgeCalculator.cpp:581-635) over a whole file: the
 * candidates that would enter process_overlaps, in order.  *n_out may exceed cap. */
int hc_host_parse_file(const hc_settings* settings, hc_fastq* f, const char* overlaps_path, hc_overlap_rec* out,
                       uint64_t cap, uint64_t* n_out, hc_ec_counters* counters);
/* The same over the overlaps file's text in memory (what hc_ec_construct_edges_from_reads hands to the stage): identical
 * records, counters and errors as for a file holding those bytes. */
int hc_host_parse_text(const hc_settings* settings, hc_fastq* f, const char* text, uint64_t n_bytes, hc_overlap_rec* out, uint64_t cap,
                       uint64_t* n_out, hc_ec_counters* counters);

/* The inverse of the parser: n candidate records as 13-column overlaps-file lines (SURVEY.md Appendix A;
 * Overlap::get_overlap_line, src/Overlap.h:234-237, with "-" in the POS2/PERC2/LEN2 columns of a
 * single-single overlap as scripts/sfo2overlaps.py:153 writes them).  read_ids[r] / read_paired[r] describe
 * read index r.  Formats on n_threads threads (0 = all), writes sequentially. */
int hc_host_write_overlaps(const char* path, const hc_overlap_rec* recs, uint64_t n, const uint64_t* read_ids,
                           const uint8_t* read_paired, uint64_t n_reads, uint32_t n_threads);

/* SFO records as the text file rust-overlaps writes (tab separated, one record per line). */
int hc_host_write_sfo(const char* path, const hc_sfo_rec* recs, uint64_t n);

/* SFO ingest: rust-overlaps' 8-column SFO file -> SAVAGE's 13-column overlaps file with
 * the semantics of the reference's scripts/sfo2overlaps.py (--in, --out, --num_singles, --num_pairs),
 * including its sort / uniq passes.  *n_lines receives the number of overlap lines written. */
int hc_sfo2overlaps(const char* sfo_path, const char* out_path, uint64_t num_singles, uint64_t num_pairs, uint64_t* n_lines);

/* The same from the binary records of hc_find_overlaps: the result is what hc_sfo2overlaps gives for the file
 * hc_host_write_sfo writes for those records, without writing or parsing it. */
int hc_sfo_records_to_overlaps(const hc_sfo_rec* recs, uint64_t n, const char* out_path, uint64_t num_singles, uint64_t num_pairs,
                               uint64_t* n_lines);

/* The serial insert of process_overlaps (src/EdgeCalculator.cpp:441-538) on a bare graph. */
typedef struct hc_host_graph hc_host_graph;
int hc_host_graph_new(hc_host_graph** out, uint64_t n_vertices, const hc_settings* settings);
int hc_host_graph_insert(hc_host_graph* g, const hc_edge_rec* edge);
/* The sort-based equivalent on an EMPTY graph: resolves n edges given in sequence
 * order in one call; must leave the graph exactly as n hc_host_graph_insert calls would. */
int hc_host_graph_resolve(hc_host_graph* g, const hc_edge_rec* edges, uint64_t n);
/* The graph as the device's duplicate resolution hands it over (hc_graph_fetch: CSR of hc_edge_rec + in-lists), into an
 * EMPTY graph: the lists become views into two arrays owned by the graph (OverlapGraph::adopt_csr). */
int hc_host_graph_adopt(hc_host_graph* g, const hc_edge_rec* edges, const uint64_t* out_off, const uint32_t* in_nodes, const uint64_t* in_off,
                        const uint8_t* inclusions);
/* OverlapGraph::addEquivalentEdges (src/OverlapGraph.cpp:608-719); the graph must have been created with
 * HC_FLAG_ADD_DUPLICATES and twice the number of reads as vertices. */
int hc_host_graph_add_equivalent_edges(hc_host_graph* g);
int hc_host_graph_get(hc_host_graph* g, hc_edge_rec* out, uint64_t cap, uint64_t* n_out, uint8_t* inclusions,
                      hc_ec_counters* counters);
/* OverlapGraph::sortEdges on the bare graph; len_by_read[r] = Read::get_len() of read r (n_reads = n_vertices). */
int hc_host_graph_sort_edges(hc_host_graph* g, const uint32_t* len_by_read, uint64_t n_reads);
int hc_host_graph_get_in_lists(hc_host_graph* g, uint64_t* in_off, uint64_t* in_nodes, uint64_t cap);
int hc_host_graph_free(hc_host_graph* g);

#ifdef __cplusplus
}
#endif
#endif /* HCEDGE_HOST_H_ */
