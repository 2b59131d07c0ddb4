/*
 * hcfno.h — C ABI of the "find next overlaps" step in libhcedge.so: the pure-integer maps that turn the
 * edges of this iteration into the overlaps file of the next one.
 *
 *   FNO=1  SRBuilder::findNextOverlaps   src/FindNextOverlaps.cpp:890-958
 *            updateOverlap :25-327, findCliqueIndex :331-347, computeOverlapData :351-565,
 *            reconsiderEdgeOverlaps :605-631, reconsiderNonedgeOverlaps :635-813 (the checkEdge filter :702, the
 *            opposite overlaps of --add_duplicates :699-793),
 *            findInclusionOverlaps :816-887
 *   FNO=3  SRBuilder::findNextOverlaps3  src/FindNextOverlaps3.cpp:20-88, nodeDictApproach :90-173,
 *            deduceOverlap :176-406
 *
 * The reference reads its inputs out of SRBuilder / OverlapGraph / Read objects; this ABI takes the same
 * facts as flat arrays (what each getter would return), so that any owner of super-reads can call it.
 * What the reference writes to <output>/overlaps.txt comes back as one text buffer, byte for byte, plus the
 * counters the reference prints.  With a HIP device present, inputs of 2 * 10^5 edges and more run on it (FNO=1 whole; FNO=3's
 * deduceOverlap and text); smaller ones and HC_FNO=host on host threads — same bytes either way.
 *
 * Conventions as in hcedge.h: plain pointers and sizes, HC_OK or a negative hc_status, hc_last_error() has
 * the text.  Wherever the reference would assert/exit/throw (a node missing from a map, a percentage above
 * 100, a read of length 0 ...) the call returns HC_ERR_FORMAT and produces no output.
 */
#ifndef HCFNO_H_
#define HCFNO_H_

#include "hcedge.h"

#ifdef __cplusplus
extern "C" {
#endif

/* What FNO reads of a Read (src/Read.h): identifier, type and sequence lengths.
 * Used for the vertices of the overlap graph (original reads of this iteration) and for super-reads. */
typedef struct hc_fno_read {
    uint64_t id;         /* vertices: SRBuilder::nodes_to_new_IDs.at(vertex), read only when !visited;
                            super-reads: Read::get_read_id() */
    uint32_t len1;       /* single: get_seq(0).length(); paired: get_seq(1).length() */
    uint32_t len2;       /* paired: get_seq(2).length(); single: 0 */
    uint8_t paired;      /* Read::is_paired() */
    uint8_t visited;     /* vertices only: SRBuilder::visited[vertex] (merged into some super-read) */
    uint8_t orientation; /* vertices only: OverlapGraph::getOrientation(vertex) */
    uint8_t pad[5];
} hc_fno_read; /* 24 bytes */

/* What FNO reads of an Edge (src/Edge.h:123-218). */
typedef struct hc_fno_edge {
    uint64_t v1, v2; /* get_vertex(1), get_vertex(2) */
    double score;    /* get_score(); 0 marks a stored non-edge (FindNextOverlaps.cpp:35) */
    int32_t pos1, pos2;
    int32_t len1, len2; /* get_len(1), get_len(2) */
    int32_t perc;       /* get_perc() */
    uint8_t ord;        /* '-', '1', '2' */
    uint8_t ori1, ori2; /* get_ori(1), get_ori(2) */
    uint8_t pad;
} hc_fno_edge; /* 48 bytes */

/* One entry of a super-read's subreadMap: vertex -> SubreadInfo (src/Types.h:77-82). */
typedef struct hc_fno_subread {
    uint64_t node;
    int32_t index1, index2, startpos1, startpos2;
} hc_fno_subread; /* 24 bytes */

/* One entry of a super-read's original_read_indexes: original read id -> OriginalIndex.index1/index2
 * (src/Types.h:84-91; FNO=3 reads nothing else of it). */
typedef struct hc_fno_original {
    uint64_t original_id;
    int64_t index1, index2;
} hc_fno_original; /* 24 bytes */

#define HC_FNO_RESOLVE_ORIENTATIONS 0x1u /* program_settings.resolve_orientations */
#define HC_FNO_NO_INCLUSIONS        0x2u /* program_settings.no_inclusions */
#define HC_FNO_OPTIMIZE             0x4u /* program_settings.optimize: skip the stored non-edges (:914) */
/* program_settings.add_duplicates (src/FindNextOverlaps.cpp:672-675, 699-793): the overlap graph has a vertex per read AND strand —
 * `nodes` then lists every read twice, vertex r as it is and vertex r + n_nodes / 2 as its reverse complement
 * (src/ViralQuasispecies.cpp:246-270: same lengths and type, each with its own id / visited / orientation) — and
 * reconsiderNonedgeOverlaps takes the vertices of a stored non-edge by the line's orientations (`nonedges[i].v1` = read1's vertex on
 * strand ori1, likewise v2: anything else is HC_ERR_ARG) and, for every line that passes :702, adds the SAME overlap a second time as
 * seen from the other strand (positions from the read lengths, :703-792).  hc_fno1_run builds that second edge itself: `nonedges`
 * stays one record per line.  FNO=3 never reads the setting: hc_fno3_run accepts the bit and ignores it.
 * Bits nobody defined are refused with HC_ERR_ARG. */
#define HC_FNO_ADD_DUPLICATES       0x8u
#define HC_FNO_KNOWN_FLAGS          (HC_FNO_RESOLVE_ORIENTATIONS | HC_FNO_NO_INCLUSIONS | HC_FNO_OPTIMIZE | HC_FNO_ADD_DUPLICATES)

typedef struct hc_fno1_input {
    /* vertices of the overlap graph, index = vertex id (HC_FNO_ADD_DUPLICATES: 2 x reads, see above) */
    const hc_fno_read* nodes;
    uint64_t n_nodes;
    /* super-reads: single_SR_vec followed by paired_SR_vec (:893-906) */
    const hc_fno_read* srs;
    uint64_t n_srs;
    const uint64_t* clique_off;  /* [n_srs+1] into clique_nodes */
    const uint64_t* clique_nodes; /* get_sorted_clique(0) of a single / get_sorted_clique(1) of a paired super-read */
    const uint64_t* subread_off; /* [n_srs+1] into subreads */
    const hc_fno_subread* subreads; /* subreadMap of each super-read, any order */
    /* the edge sources, in the order the reference walks them */
    const hc_fno_edge* graph_edges; /* OverlapGraph::adj_out, vertex by vertex, each list front to back (:612-624) */
    uint64_t n_graph_edges;
    const hc_fno_edge* branching_edges; /* OverlapGraph::branching_edges (:627-630) */
    uint64_t n_branching_edges;
    const hc_fno_edge* nonedges; /* nonedge_overlaps.txt, one record per line, score 0 (:635-813); may be NULL */
    uint64_t n_nonedges;
    const uint64_t* inclusion_off; /* [n_inclusion_groups+1] into inclusion_edges */
    const hc_fno_edge* inclusion_edges; /* OverlapGraph::inclusion_edges, group by group (:822) */
    uint64_t n_inclusion_groups;
    uint64_t new_read_count; /* SRBuilder::new_read_count: every id is below it (:891) */
    double edge_threshold;   /* score given to inclusion-induced edges (:828) */
    uint32_t flags;          /* HC_FNO_* */
    uint32_t n_threads;      /* 0 = all */
} hc_fno1_input;

typedef struct hc_fno3_input {
    /* single_SR_vec, paired_SR_vec, trivial_SR_vec, concatenated in this order (FindNextOverlaps3.cpp:29-76) */
    const hc_fno_read* srs;
    uint64_t n_single, n_paired, n_trivial;
    const uint64_t* orig_off; /* [n_single+n_paired+n_trivial+1] into originals */
    const hc_fno_original* originals; /* get_original_reads() of each, in that map's iteration order */
    uint64_t new_read_count;      /* every super-read id is below it (:92) */
    uint64_t original_readcount;  /* program_settings.original_readcount: size of nodes_to_SR (SRBuilder.h:99-100) */
    uint32_t flags;               /* HC_FNO_NO_INCLUSIONS */
    uint32_t n_threads;
} hc_fno3_input;

typedef struct hc_fno_counters {
    uint64_t n_lines;                                 /* lines of overlaps.txt = SRBuilder::next_overlaps_count */
    uint64_t copied, u2sr, v2sr, sr2sr;               /* FNO=1: the four counters printed at :938-939 */
    uint64_t candidates;                              /* FNO=3: overlaps_list.size() (:156) */
} hc_fno_counters;

typedef struct hc_fno_output hc_fno_output; /* owns the text of overlaps.txt */

/* SRBuilder::findNextOverlaps(): lines sorted and unique as std::set<std::string> yields them. */
int hc_fno1_run(const hc_fno1_input* in, hc_fno_output** out);
/* SRBuilder::findNextOverlaps3(): lines in candidate order. */
int hc_fno3_run(const hc_fno3_input* in, hc_fno_output** out);
int hc_fno_output_text(const hc_fno_output* o, const char** text, uint64_t* n_bytes);
int hc_fno_output_counters(const hc_fno_output* o, hc_fno_counters* c);
int hc_fno_output_write(const hc_fno_output* o, const char* path); /* what the reference leaves in overlaps.txt */
void hc_fno_output_free(hc_fno_output* o);
/* 0: the host threads wrote this output.  1: the arithmetic, the ordering and the text were the device's, the walk and the
 * look-ups the host threads'.  2 (FNO=1): the walk — updateOverlap's case analysis, the first combination per pair of new ids
 * (:25-349) — and the look-ups were the device's too.  The device takes batches of 200 000 edges / combinations and more when a
 * HIP device is present; HC_FNO=host / HC_FNO=device force either, HC_FNO_WALK=host keeps the walk on the host threads.  The
 * device is the calling thread's current HIP device (hipSetDevice; device 0 unless the caller chose another). */
int hc_fno_output_on_device(const hc_fno_output* o);

/* computeOverlapData (:351-565) on its own: the overlap of two (super-)reads induced by one edge.
 * idx = {idx1l, idx1r, idx2l, idx2r}.  *ok = 0 is the reference's "failure" return.  out9 receives
 * new_pos1, new_pos2, ord1, ord2, type1 ('s'/'p'), type2, overlap_perc, overlap_len1, overlap_len2. */
int hc_fno_compute_overlap_data(const hc_fno_read* sr1, const hc_fno_read* sr2, const int32_t idx[4],
                                const hc_fno_edge* edge, int32_t* ok, int32_t out9[9]);

#ifdef __cplusplus
}
#endif
#endif /* HCFNO_H_ */
