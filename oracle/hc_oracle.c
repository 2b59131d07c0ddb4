/*
 * hc_oracle.c — CPU ORACLE (test infrastructure, NOT product code).
 * Plain-C restatement of HaploConduct's edge-calculation path; see hc_oracle.h
 * for scope and pinning status (compute_overlap / process_overlaps are held to outputs of the reference's
 * own code; "parity unpinned" only for construct_edges' tokeniser + prefilter and the FASTQ reader).
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp (no -ffast-math, no -march): the
 * reference is built with g++ -O2 for baseline x86-64, i.e. IEEE double, no FMA
 * contraction, glibc pow/log/exp (makefile:9-10).
 */
#define _GNU_SOURCE
#include "hc_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------- */
/* EdgeCalculator.cpp:59-63 — P = pow(10, -phred/10.0); assert 0<=P<=1 is checked by the caller */
double hco_phred_to_prob(int phred) { return pow(10, -phred / 10.0); }

static int valid_nt(char c) { return c == 'A' || c == 'T' || c == 'C' || c == 'G' || c == 'N'; }

/* EdgeCalculator.cpp:26-56 */
double hco_score(char nt1, char nt2, double p1, double p2, int* mismatch_count, double mismatch_setting) {
    if (!valid_nt(nt1) || !valid_nt(nt2)) return -1000.0; /* :29-30 assert */
    double p;
    if (nt1 == 'N' || nt2 == 'N') { /* :35-39 */
        return 1;
    } else if (nt1 == nt2) { /* :40-42 */
        p = (1 - p1) * (1 - p2) + (p1 * p2) / 3.0;
    } else { /* :43-46 — the counter moves before the --mismatch test */
        p = p1 * (1 - p2) / 3.0 + p2 * (1 - p1) / 3.0 + (2 / 9.0) * p1 * p2;
        (*mismatch_count)++;
    }
    if (!(p > 0 && p <= 1)) return -1000.0; /* :47 assert */
    if (p < mismatch_setting) return 2;     /* :49-51 */
    double lp = log(p);                     /* :52 */
    if (!(lp == lp) || !(lp <= 0)) return -1000.0; /* :53-54 asserts */
    return lp;
}

/* EdgeCalculator.cpp:67-139 */
double hco_overlap_score(const char* seq1, size_t len1, const char* seq2, size_t len2, const char* q1,
                         const char* q2, unsigned int pos, unsigned int min_read_len, double mismatch_setting,
                         double* mismatch_rate, double* x_out, uint32_t* mm_out, uint32_t* n_out,
                         uint64_t* positions_out, int* status) {
    *mismatch_rate = 1.0; /* :74 */
    if (x_out) *x_out = -INFINITY;
    if (mm_out) *mm_out = 1;
    if (n_out) *n_out = 1;
    if (positions_out) *positions_out = 0;
    if (status) *status = 0;
    if (len1 == 0 || len2 == 0) { /* :70-73 asserts */
        if (status) *status = -1;
        return 0;
    }
    if (pos >= len1) return 0;                                 /* :76-79 (the reference also prints pos / length) */
    if (len1 < min_read_len || len2 < min_read_len) return 0;  /* :82-84 */

    unsigned int L1 = (unsigned int)len1;
    unsigned int L2 = (unsigned int)len2;
    unsigned int Lu = L1 - pos < L2 ? L1 - pos : L2; /* :88 std::min<unsigned> */
    int L = (int)Lu;
    if (positions_out) *positions_out = (uint64_t)L;

    double* probs1 = (double*)malloc(sizeof(double) * (size_t)(L > 0 ? L : 1));
    double* probs2 = (double*)malloc(sizeof(double) * (size_t)(L > 0 ? L : 1));
    for (int i = 0; i < L; i++) { /* :92-101 */
        int Q1 = (int)(signed char)q1[i + pos] - 33;
        int Q2 = (int)(signed char)q2[i] - 33;
        double P1 = hco_phred_to_prob(Q1);
        double P2 = hco_phred_to_prob(Q2);
        if (!(P1 == P1 && P1 >= 0 && P1 <= 1) || !(P2 == P2 && P2 >= 0 && P2 <= 1)) { /* :61,97-98 asserts */
            if (status) *status = -2;
            free(probs1);
            free(probs2);
            return 0;
        }
        probs1[i] = P1;
        probs2[i] = P2;
    }

    double total_score = 0.0; /* :103-105 */
    double total_len = 0.0;
    int mismatch_count = 0;
    for (int i = 0; i < L; i++) { /* :106-128 */
        double s = hco_score(seq1[i + pos], seq2[i], probs1[i], probs2[i], &mismatch_count, mismatch_setting);
        if (s == -1000.0) {
            if (status) *status = -3;
            free(probs1);
            free(probs2);
            return 0;
        }
        if (s <= 0) {
            total_score += s;
            total_len += 1;
        } else if (s == 1) {
            continue;
        } else {
            free(probs1);
            free(probs2);
            return 0;
        }
    }
    free(probs1);
    free(probs2);
    if (total_len == 0) return 0; /* :129-131 */
    *mismatch_rate = (float)mismatch_count / total_len; /* :132 */
    total_score = (1.0 / total_len) * total_score;      /* :137 */
    if (x_out) *x_out = total_score;
    if (mm_out) *mm_out = (uint32_t)mismatch_count;
    if (n_out) *n_out = (uint32_t)total_len;
    return exp(total_score); /* :138 */
}

/* Types.h:109-129 */
int hco_build_rev_comp(const char* seq, size_t len, char* out) {
    for (size_t i = 0; i < len; i++) {
        char c = seq[len - 1 - i];
        char r;
        if (c == 'A') r = 'T';
        else if (c == 'T') r = 'A';
        else if (c == 'C') r = 'G';
        else if (c == 'G') r = 'C';
        else if (c == 'N') r = 'N';
        else return -1; /* "Invalid sequence character. Aborting." */
        out[i] = r;
    }
    return 0;
}

/* One oriented view of a stored sequence: Read::get_seq / get_phred (Read.h:144-170)
 * or get_rev_comp / get_rev_phred (Read.h:172-201). */
typedef struct view {
    char* seq;
    char* qual;
    size_t len;
} view;

static int make_view(const hco_reads* R, uint32_t seq_idx, int forward, view* v) {
    uint64_t a = R->seq_off[seq_idx], b = R->seq_off[seq_idx + 1];
    size_t len = (size_t)(b - a);
    v->len = len;
    v->seq = (char*)malloc(len + 1);
    v->qual = (char*)malloc(len + 1);
    if (forward) {
        memcpy(v->seq, R->bases + a, len);
        memcpy(v->qual, R->quals + a, len);
    } else {
        if (hco_build_rev_comp((const char*)R->bases + a, len, v->seq) != 0) return -1;
        for (size_t i = 0; i < len; i++) v->qual[i] = (char)R->quals[a + len - 1 - i];
    }
    return 0;
}
static void free_view(view* v) {
    free(v->seq);
    free(v->qual);
    v->seq = v->qual = NULL;
}

/* single: i=0; paired: i=1 -> /1, i=2 -> /2 (Read.h:144-156) */
static uint32_t seq_of(const hco_reads* R, uint32_t read, int i) {
    uint32_t f = R->read_first_seq[read];
    return i == 2 ? f + 1 : f;
}
static size_t seq_len(const hco_reads* R, uint32_t seq_idx) {
    return (size_t)(R->seq_off[seq_idx + 1] - R->seq_off[seq_idx]);
}

typedef struct sub_result {
    double ov, mismatch, x;
    uint32_t mm, n;
    uint64_t positions;
} sub_result;

static int run_sub(const hco_reads* R, const hco_settings* s, uint32_t seqA, int fwdA, uint32_t seqB, int fwdB,
                   unsigned int pos, sub_result* out) {
    view A, B;
    A.seq = A.qual = B.seq = B.qual = NULL;
    int st = 0;
    if (make_view(R, seqA, fwdA, &A) != 0 || make_view(R, seqB, fwdB, &B) != 0) {
        free_view(&A);
        free_view(&B);
        return -4; /* build_rev_comp exit(1) */
    }
    out->ov = hco_overlap_score(A.seq, A.len, B.seq, B.len, A.qual, B.qual, pos, s->min_read_len, s->mismatch,
                                &out->mismatch, &out->x, &out->mm, &out->n, &out->positions, &st);
    free_view(&A);
    free_view(&B);
    return st;
}

/* EdgeCalculator.cpp:404-413 */
static uint32_t classify(const hco_settings* s, double score, double mismatch_rate) {
    if (score > s->edge_threshold) return 2;
    if (mismatch_rate != -1 && mismatch_rate <= s->merge_contigs) return 3;
    if (score > s->ov_threshold && mismatch_rate != -1) return 1;
    return 0;
}

/* EdgeCalculator.cpp:143-385 */
int hco_compute_overlap(const hco_reads* R, const hco_settings* s, const hco_overlap* ov, hco_edge* e) {
    memset(e, 0, sizeof(*e));
    e->x1 = -INFINITY;
    e->x2 = NAN;
    e->mismatch_rate = -1;
    if (ov->read1 >= R->n_reads || ov->read2 >= R->n_reads || ov->read1 == ov->read2) { /* :170-171 .at(), :184 assert */
        e->status = -10;
        return e->status;
    }
    int ori1 = ov->ori1 ? 1 : 0, ori2 = ov->ori2 ? 1 : 0;
    if (!((s->flags & HCO_FLAG_ADD_DUPLICATES) || (s->flags & HCO_FLAG_RESOLVE_ORIENTATIONS))) { /* :153 */
        if (!(ori1 && ori2)) {
            e->status = -11;
            return e->status;
        }
    }
    unsigned int pos1 = ov->pos1, pos2 = ov->pos2;
    int paired1 = (R->read_first_seq[ov->read1 + 1] - R->read_first_seq[ov->read1]) == 2; /* :186-187 */
    int paired2 = (R->read_first_seq[ov->read2 + 1] - R->read_first_seq[ov->read2]) == 2;
    uint32_t r1 = ov->read1, r2 = ov->read2;
    sub_result a, b;
    memset(&a, 0, sizeof a);
    memset(&b, 0, sizeof b);
    int st;

    if (!paired1 && !paired2) { /* S-S :199-233 */
        st = run_sub(R, s, seq_of(R, r1, 0), ori1, seq_of(R, r2, 0), ori2, pos1, &a);
        if (st) { e->status = st; return st; }
        e->n_subs = 1;
        e->ov1 = a.ov;
        e->x1 = a.x;
        e->score = a.ov;
        e->mismatch_rate = a.mismatch;
        e->mm = a.mm;
        e->n = a.n;
        e->positions = a.positions;
        e->pos3 = (int)(seq_len(R, seq_of(R, r1, 0)) - pos1 - seq_len(R, seq_of(R, r2, 0))); /* :222 */
        e->pos4 = 0;                                                                          /* Edge.h:168 default */
    } else {
        uint32_t A1, B1, A2, B2;
        int fA1, fB1, fA2, fB2;
        if (!paired1 && paired2) { /* S-P :234-271 */
            A1 = seq_of(R, r1, 0); fA1 = ori1;
            A2 = A1; fA2 = ori1;
            if (ori2) { B1 = seq_of(R, r2, 1); fB1 = 1; B2 = seq_of(R, r2, 2); fB2 = 1; }
            else      { B1 = seq_of(R, r2, 2); fB1 = 0; B2 = seq_of(R, r2, 1); fB2 = 0; }
            e->pos3 = (int)(seq_len(R, seq_of(R, r1, 0)) - pos2 - seq_len(R, seq_of(R, r2, 2))); /* :262 */
            e->pos4 = (int)(seq_len(R, seq_of(R, r1, 0)) - pos1 - seq_len(R, seq_of(R, r2, 1))); /* :263 */
        } else if (paired1 && !paired2) { /* P-S :272-309 */
            if (ori1) { A1 = seq_of(R, r1, 1); fA1 = 1; B2 = seq_of(R, r1, 2); fB2 = 1; }
            else      { A1 = seq_of(R, r1, 2); fA1 = 0; B2 = seq_of(R, r1, 1); fB2 = 0; }
            B1 = seq_of(R, r2, 0); fB1 = ori2;
            A2 = seq_of(R, r2, 0); fA2 = ori2;
            e->pos3 = (int)(seq_len(R, seq_of(R, r1, 2)) + pos2 - seq_len(R, seq_of(R, r2, 0))); /* :300 */
            e->pos4 = (int)(seq_len(R, seq_of(R, r2, 0)) + pos1 - seq_len(R, seq_of(R, r1, 1))); /* :301 */
        } else { /* P-P :312-380 */
            uint32_t F1, K1, F2, K2;
            int fF1, fK1, fF2, fK2;
            if (ori1) { F1 = seq_of(R, r1, 1); fF1 = 1; K1 = seq_of(R, r1, 2); fK1 = 1; }
            else      { F1 = seq_of(R, r1, 2); fF1 = 0; K1 = seq_of(R, r1, 1); fK1 = 0; }
            if (ori2) { F2 = seq_of(R, r2, 1); fF2 = 1; K2 = seq_of(R, r2, 2); fK2 = 1; }
            else      { F2 = seq_of(R, r2, 2); fF2 = 0; K2 = seq_of(R, r2, 1); fK2 = 0; }
            A1 = F1; fA1 = fF1; B1 = F2; fB1 = fF2;
            if (ov->ord == '1')      { A2 = K1; fA2 = fK1; B2 = K2; fB2 = fK2; }
            else if (ov->ord == '2') { A2 = K2; fA2 = fK2; B2 = K1; fB2 = fK1; }
            else { e->status = -12; return e->status; } /* :369 assert (ov2/mismatch2 would be uninitialised) */
            if (ov->ord == '1')
                e->pos3 = (int)(seq_len(R, seq_of(R, r1, 2)) - pos2 - seq_len(R, seq_of(R, r2, 2))); /* :363 */
            else
                e->pos3 = (int)(seq_len(R, seq_of(R, r1, 2)) + pos2 - seq_len(R, seq_of(R, r2, 2))); /* :370 */
            e->pos4 = (int)(seq_len(R, seq_of(R, r1, 1)) - pos1 - seq_len(R, seq_of(R, r2, 1)));     /* :372 */
        }
        st = run_sub(R, s, A1, fA1, B1, fB1, pos1, &a);
        if (st) { e->status = st; return st; }
        st = run_sub(R, s, A2, fA2, B2, fB2, pos2, &b);
        if (st) { e->status = st; return st; }
        e->n_subs = 2;
        e->ov1 = a.ov;
        e->ov2 = b.ov;
        e->x1 = a.x;
        e->x2 = b.x;
        e->positions = a.positions + b.positions;
        double mismatch_rate = a.mismatch > b.mismatch ? a.mismatch : b.mismatch; /* std::max(m1,m2) :254 */
        if (a.mismatch < b.mismatch) { e->mm = b.mm; e->n = b.n; } else { e->mm = a.mm; e->n = a.n; }
        double score;
        if (a.ov > s->edge_threshold && b.ov > s->edge_threshold) score = 0.5 * (a.ov + b.ov); /* :256-258 */
        else score = b.ov < a.ov ? b.ov : a.ov;                                                 /* std::min :260 */
        e->score = score;
        e->mismatch_rate = mismatch_rate;
    }
    if (!(e->score == 0 || e->score == -1 || e->score > 0)) { /* Edge.h:47 */
        e->status = -13;
        return e->status;
    }
    e->cls = classify(s, e->score, e->mismatch_rate);
    return 0;
}

/* EdgeCalculator.cpp:395-414 */
int hco_score_batch(const hco_reads* R, const hco_settings* s, const hco_overlap* in, uint64_t n, hco_edge* out,
                    int n_threads) {
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel for num_threads(n_threads) schedule(static)
    for (uint64_t i = 0; i < n; i++) hco_compute_overlap(R, s, &in[i], &out[i]);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* Overlap.h:39-73 */
static void strip_chars(char* s, const char* drop) {
    char* w = s;
    for (char* r = s; *r; r++)
        if (!strchr(drop, *r)) *w++ = *r;
    *w = 0;
}

int hco_overlap_from_fields(const char* const f[13], hco_overlap_line* o) {
    char ord[64], ori1[64], ori2[64], t1[64], t2[64];
    if (strlen(f[4]) > 60 || strlen(f[5]) > 60 || strlen(f[6]) > 60 || strlen(f[11]) > 60 || strlen(f[12]) > 60)
        return -1;
    strcpy(ord, f[4]);
    strcpy(ori1, f[5]);
    strcpy(ori2, f[6]);
    strcpy(t1, f[11]);
    strcpy(t2, f[12]);
    o->id1 = strtoul(f[0], NULL, 0); /* Types.h:99-102 */
    o->id2 = strtoul(f[1], NULL, 0);
    o->pos1 = (unsigned int)atoi(f[2]);
    o->pos2 = (unsigned int)atoi(f[3]);
    o->perc1 = (unsigned int)atoi(f[7]);
    o->perc2 = (unsigned int)atoi(f[8]);
    o->len1 = (unsigned int)atoi(f[9]);
    o->len2 = (unsigned int)atoi(f[10]);
    if (strcmp(f[3], "-") == 0) { /* :55-59 */
        o->pos2 = 0;
        o->perc2 = 0;
        o->len2 = 0;
    }
    if ((int)o->pos1 < 0 || (int)o->pos2 < 0) return -2;                 /* check_pos :102-107 */
    if (strlen(ori1) != 1) strip_chars(ori1, " ");                        /* check_ori :121-130 */
    if (strlen(ori1) != 1) return -3;
    if (strcmp(ori1, "+") && strcmp(ori1, "-")) return -3;
    if (strlen(ori2) != 1) strip_chars(ori2, " ");
    if (strlen(ori2) != 1) return -3;
    if (strcmp(ori2, "+") && strcmp(ori2, "-")) return -3;
    if ((int)o->perc1 < 0 || (int)o->perc1 > 100) return -4;              /* check_perc :132-138 */
    if ((int)o->perc2 < 0 || (int)o->perc2 > 100) return -4;
    if ((int)o->len1 < 0 || (int)o->len2 < 0) return -5;                  /* check_len :140-145 */
    if (strlen(t1) != 1) strip_chars(t1, "\n\t ");                        /* check_type :147-158 */
    if (strlen(t1) != 1) return -6;
    if (strcmp(t1, "s") && strcmp(t1, "p")) return -6;
    if (strlen(t2) != 1) strip_chars(t2, "\n\t ");
    if (strlen(t2) != 1) return -6;
    if (strcmp(t2, "s") && strcmp(t2, "p")) return -6;
    if (strlen(ord) != 1) strip_chars(ord, " ");                          /* check_ord :109-119 */
    if (strlen(ord) != 1) return -7;
    if (strcmp(ord, "1") && strcmp(ord, "2") && strcmp(ord, "-")) return -7;
    if (t1[0] == 's' || t2[0] == 's') {
        if (ord[0] != '-') return -7;
    } else if (ord[0] != '1' && ord[0] != '2') return -7;
    o->ord = ord[0];
    o->ori1 = ori1[0];
    o->ori2 = ori2[0];
    o->type1 = t1[0];
    o->type2 = t2[0];
    return 0;
}

/* Overlap.h:203-210 */
unsigned int hco_overlap_get_perc(const hco_overlap_line* o) {
    if (o->perc2 > 0) return (unsigned int)(0.5 * (o->perc1 + o->perc2));
    return o->perc1;
}

/* Overlap.h:234-237 */
int hco_overlap_get_line(const hco_overlap_line* o, char* buf, size_t bufsz) {
    return snprintf(buf, bufsz, "%lu\t%lu\t%u\t%u\t%c\t%c\t%c\t%u\t%u\t%u\t%u\t%c\t%c\n", o->id1, o->id2, o->pos1,
                    o->pos2, o->ord, o->ori1, o->ori2, o->perc1, o->perc2, o->len1, o->len2, o->type1, o->type2);
}

/* EdgeCalculator.cpp:584-597 */
int hco_split_line(char* s, int allow_spaces, char* fields[], int max_fields) {
    /* boost::trim_if(line, is_any_of("\t ")) */
    size_t len = strlen(s);
    size_t a = 0;
    while (a < len && (s[a] == '\t' || s[a] == ' ')) a++;
    while (len > a && (s[len - 1] == '\t' || s[len - 1] == ' ')) len--;
    s[len] = 0;
    char* p = s + a;
    int n = 0;
    if (allow_spaces) {
        /* boost::split(..., is_any_of("\t "), token_compress_on): an empty input gives one empty token */
        for (;;) {
            if (n < max_fields) fields[n] = p;
            n++;
            char* q = p;
            while (*q && *q != '\t' && *q != ' ') q++;
            if (!*q) break;
            *q++ = 0;
            while (*q == '\t' || *q == ' ') q++;
            p = q;
        }
        return n;
    }
    /* while (getline(ss, tmp, '\t')): an empty input gives no token; a trailing
     * empty token cannot occur after the trim */
    if (!*p) return 0;
    for (;;) {
        if (n < max_fields) fields[n] = p;
        n++;
        char* q = strchr(p, '\t');
        if (!q) break;
        *q = 0;
        p = q + 1;
        if (!*p) break; /* getline yields no token for a trailing delimiter */
    }
    return n;
}

/* ------------------------------------------------------------------------- */
/* OverlapGraph: adj_out = vector<list<Edge>>, adj_in = vector<list<node_id_t>>
 * (OverlapGraph.h:44-45); lists become order-preserving arrays. */
typedef struct elist {
    hco_gedge* e;
    uint64_t n, cap;
} elist;
typedef struct vlist {
    uint64_t* v;
    uint64_t n, cap;
} vlist;
struct hco_graph {
    uint64_t V, edge_count;
    elist* out;
    vlist* in;
    uint8_t* inclusions;
};

hco_graph* hco_graph_new(uint64_t V) {
    hco_graph* g = (hco_graph*)calloc(1, sizeof(*g));
    g->V = V;
    g->out = (elist*)calloc(V ? V : 1, sizeof(elist));
    g->in = (vlist*)calloc(V ? V : 1, sizeof(vlist));
    g->inclusions = (uint8_t*)calloc(V ? V : 1, 1);
    return g;
}
void hco_graph_free(hco_graph* g) {
    if (!g) return;
    for (uint64_t i = 0; i < g->V; i++) {
        free(g->out[i].e);
        free(g->in[i].v);
    }
    free(g->out);
    free(g->in);
    free(g->inclusions);
    free(g);
}
uint64_t hco_graph_edge_count(const hco_graph* g) { return g->edge_count; }
uint64_t hco_graph_out(const hco_graph* g, uint64_t v, const hco_gedge** edges) {
    *edges = g->out[v].e;
    return g->out[v].n;
}
int hco_graph_inclusion(const hco_graph* g, uint64_t v) { return g->inclusions[v]; }

/* OverlapGraph::addEdge, OverlapGraph.cpp:94-101 */
static void g_add(hco_graph* g, const hco_gedge* e) {
    elist* L = &g->out[e->v1];
    if (L->n == L->cap) {
        L->cap = L->cap ? 2 * L->cap : 4;
        L->e = (hco_gedge*)realloc(L->e, L->cap * sizeof(hco_gedge));
    }
    L->e[L->n++] = *e;
    vlist* I = &g->in[e->v2];
    if (I->n == I->cap) {
        I->cap = I->cap ? 2 * I->cap : 4;
        I->v = (uint64_t*)realloc(I->v, I->cap * sizeof(uint64_t));
    }
    I->v[I->n++] = e->v1;
    g->edge_count++;
}

/* OverlapGraph::checkEdgeWithOri (OverlapGraph.cpp:198-229) and
 * getEdgeInfoWithOri (:285-306): v->w first, then w->v. */
static hco_gedge* g_find(hco_graph* g, uint64_t v, uint64_t w, int opp) {
    elist* L = &g->out[v];
    for (uint64_t i = 0; i < L->n; i++)
        if (L->e[i].v2 == w && ((L->e[i].ori1 == L->e[i].ori2) == opp)) return &L->e[i];
    L = &g->out[w];
    for (uint64_t i = 0; i < L->n; i++)
        if (L->e[i].v2 == v && ((L->e[i].ori1 == L->e[i].ori2) == opp)) return &L->e[i];
    return NULL;
}

/* OverlapGraph::removeEdgeWithOri, OverlapGraph.cpp:150-194 */
static int g_remove(hco_graph* g, uint64_t v, uint64_t w, int opp) {
    elist* L = &g->out[v];
    uint64_t i;
    for (i = 0; i < L->n; i++)
        if (L->e[i].v2 == w && ((L->e[i].ori1 == L->e[i].ori2) == opp)) break;
    if (i == L->n) return -1;
    memmove(&L->e[i], &L->e[i + 1], (L->n - i - 1) * sizeof(hco_gedge));
    L->n--;
    g->edge_count--;
    vlist* I = &g->in[w];
    for (i = 0; i < I->n; i++)
        if (I->v[i] == v) {
            memmove(&I->v[i], &I->v[i + 1], (I->n - i - 1) * sizeof(uint64_t));
            I->n--;
            break;
        }
    return 0;
}

/* OverlapGraph::checkEdge with reverse_allowed = false, OverlapGraph.cpp:233-259: the first edge v -> w of the list */
static double g_check_edge(const hco_graph* g, uint64_t v, uint64_t w) {
    const elist* L = &g->out[v];
    for (uint64_t i = 0; i < L->n; i++)
        if (L->e[i].v2 == w) return L->e[i].score;
    return -1;
}

/* OverlapGraph::removeEdge, OverlapGraph.cpp:102-146: the first edge v -> w, the first v in w's in-list */
static int g_remove_edge(hco_graph* g, uint64_t v, uint64_t w) {
    elist* L = &g->out[v];
    uint64_t i;
    for (i = 0; i < L->n; i++)
        if (L->e[i].v2 == w) break;
    if (i == L->n) return -1;
    memmove(&L->e[i], &L->e[i + 1], (L->n - i - 1) * sizeof(hco_gedge));
    L->n--;
    g->edge_count--;
    vlist* I = &g->in[w];
    for (i = 0; i < I->n; i++)
        if (I->v[i] == v) {
            memmove(&I->v[i], &I->v[i + 1], (I->n - i - 1) * sizeof(uint64_t));
            I->n--;
            break;
        }
    return 0;
}

/* Edge::swap_reads, Edge.h:74-88 */
static void edge_swap_reads(hco_gedge* e) {
    uint32_t r = e->read1; e->read1 = e->read2; e->read2 = r;
    uint64_t v = e->v1; e->v1 = e->v2; e->v2 = v;
    uint8_t o = e->ori1; e->ori1 = e->ori2; e->ori2 = o;
    if (e->ord == '1') e->ord = '2';
    else if (e->ord == '2') e->ord = '1';
    e->pos3 = -e->pos3;
    e->pos4 = -e->pos4;
}

/* EdgeCalculator.cpp:441-538 */
int hco_graph_insert(hco_graph* g, const hco_settings* s, hco_gedge* it1, hco_counters* c) {
    uint64_t v1 = it1->v1, v2 = it1->v2;
    if (it1->pos1 == 0) { /* :443-448 */
        if (v1 > v2) {
            uint64_t t = v1; v1 = v2; v2 = t;
            edge_swap_reads(it1);
        }
    }
    if (it1->perc == 100) c->inclusion_count++; /* :449-451 */
    int opp = (it1->ori1 == it1->ori2);          /* :453 */
    hco_gedge* ex = g_find(g, v1, v2, opp);
    if (!ex) { /* :455-469 (checkEdgeWithOri returned -1) */
        g_add(g, it1);
        c->edges_added++;
        if ((s->flags & HCO_FLAG_IGNORE_INCLUSIONS) && it1->perc == 100 && it1->mismatch_rate < 0.000001 &&
            it1->mismatch_rate >= 0) {
            if (it1->pos3 < 0) {
                if (it1->pos1 == 0) g->inclusions[v1] = 1;
            } else {
                g->inclusions[v2] = 1;
            }
        }
        return 0;
    }
    double score = ex->score;
    if (score < 0) return -1; /* cannot happen: found edges have score >= 0 */
    if (it1->score >= score) { /* :470-534 */
        c->dup_count++;
        if (score == it1->score) {
            if (ex->len0 != it1->len0) {
                if (ex->len0 > it1->len0) return 0;
            } else if (ex->mismatch_rate != it1->mismatch_rate) {
                if (ex->mismatch_rate < it1->mismatch_rate) return 0;
            } else if (ex->v1 != it1->v1) {
                if (ex->v1 < it1->v1) return 0;
            } else if (ex->ori1 != it1->ori1) {
                if (ex->ori1) return 0;
            } else if (ex->ori2 != it1->ori2) {
                if (ex->ori2) return 0;
            } else if (ex->pos1 != it1->pos1) {
                if (ex->pos1 < it1->pos1) return 0;
            } else if (ex->pos2 != it1->pos2) {
                if (ex->pos2 < it1->pos2) return 0;
            }
        }
        if (ex->v1 == v1) g_remove(g, v1, v2, opp); /* :523-528 */
        else g_remove(g, v2, v1, opp);
        g_add(g, it1); /* :530 */
    } else {
        c->dup_count++; /* :535-538 */
    }
    return 0;
}

/* OverlapGraph::addEquivalentEdges, OverlapGraph.cpp:608-719 (--add_duplicates): every edge once more between the
 * vertices of the reverse-complemented reads.  Vertex of read r in orientation o: r if o, n_reads + r otherwise
 * (ViralQuasispecies.cpp:259-271).  The mirrored Edge gets pos1/pos2 from the original's reverse offsets and never has
 * its own reverse offsets or mismatch rate set (uninitialised / -1 in the reference): pos3 = pos4 = 0, mismatch -1 here. */
int hco_graph_add_equivalent_edges(hco_graph* g, uint32_t n_reads) {
    uint64_t total = 0;
    for (uint64_t i = 0; i < g->V; i++) total += g->out[i].n;
    hco_gedge* extra = (hco_gedge*)malloc(sizeof(hco_gedge) * (total ? total : 1));
    uint64_t* cnt = (uint64_t*)calloc(g->V + 1, sizeof(uint64_t));
    uint64_t m = 0;
    for (uint64_t i = 0; i < g->V; i++) /* :618-668 */
        for (uint64_t k = 0; k < g->out[i].n; k++) {
            const hco_gedge* it = &g->out[i].e[k];
            hco_gedge e;
            memset(&e, 0, sizeof e);
            int pos1 = it->pos3, pos2 = it->pos4;
            if (pos1 < 0) {
                e.read1 = it->read2;
                e.read2 = it->read1;
                e.ori1 = !it->ori2;
                e.ori2 = !it->ori1;
                pos1 = -pos1;
                if (pos2 < 0) {
                    e.ord = '1';
                    pos2 = -pos2;
                } else {
                    e.ord = (it->ord == '-' || it->ord == '0') ? '-' : '2';
                }
            } else {
                e.read1 = it->read1;
                e.read2 = it->read2;
                e.ori1 = !it->ori1;
                e.ori2 = !it->ori2;
                if (pos2 < 0) {
                    pos2 = -pos2;
                    e.ord = '2';
                } else {
                    e.ord = (it->ord == '-' || it->ord == '0') ? '-' : '1';
                }
            }
            e.score = it->score;
            e.pos1 = pos1;
            e.pos2 = pos2;
            e.v1 = e.ori1 ? e.read1 : (uint64_t)n_reads + e.read1;
            e.v2 = e.ori2 ? e.read2 : (uint64_t)n_reads + e.read2;
            if (e.v1 >= g->V || e.v2 >= g->V) {
                free(extra);
                free(cnt);
                return -21;
            }
            e.len1 = it->len1; /* set_len(get_len(1), get_len(2)) */
            e.len2 = it->len2;
            e.len0 = it->len1 + it->len2;
            e.perc = it->perc;
            e.mismatch_rate = -1;
            extra[m++] = e;
            cnt[e.v1 + 1]++;
        }
    /* extra_edges.at(node1).push_back(edge): by vertex, in the order they were built */
    for (uint64_t v = 0; v < g->V; v++) cnt[v + 1] += cnt[v];
    hco_gedge* by_v = (hco_gedge*)malloc(sizeof(hco_gedge) * (total ? total : 1));
    uint64_t* at = (uint64_t*)malloc(sizeof(uint64_t) * (g->V + 1));
    memcpy(at, cnt, sizeof(uint64_t) * (g->V + 1));
    for (uint64_t k = 0; k < m; k++) by_v[at[extra[k].v1]++] = extra[k];
    for (uint64_t k = 0; k < m; k++) { /* :670-709 */
        hco_gedge* it = &by_v[k];
        uint64_t v1 = it->v1, v2 = it->v2;
        if (it->pos1 == 0 && v1 > v2) {
            uint64_t t = v1; v1 = v2; v2 = t;
            edge_swap_reads(it);
        }
        const double score = g_check_edge(g, v1, v2);
        if (score < 0) {
            g_add(g, it);
        } else if (it->score > score) {
            g_remove_edge(g, v1, v2);
            g_add(g, it);
        }
    }
    free(extra);
    free(by_v);
    free(cnt);
    free(at);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* id -> index: FastqStorage.h:88-97 builds a std::map by insert(): the first
 * occurrence of an id wins.  Sorted array + binary search here. */
typedef struct idmap_ent {
    unsigned long id;
    uint32_t idx;
} idmap_ent;
static int idmap_cmp(const void* a, const void* b) {
    const idmap_ent* x = (const idmap_ent*)a;
    const idmap_ent* y = (const idmap_ent*)b;
    if (x->id != y->id) return x->id < y->id ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx);
}
static long idmap_find(const idmap_ent* m, uint32_t n, unsigned long id) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        uint32_t mid = lo + (hi - lo) / 2;
        if (m[mid].id < id) lo = mid + 1;
        else hi = mid;
    }
    if (lo < n && m[lo].id == id) return (long)m[lo].idx; /* smallest idx among equal ids */
    return -1;
}

typedef struct pending {
    hco_overlap rec;
    hco_overlap_line line;
} pending;

/* EdgeCalculator::process_overlaps, EdgeCalculator.cpp:389-557 (one thread) */
static int process_batch(const hco_reads* R, const hco_settings* s, pending* batch, uint64_t n, hco_graph* g,
                         hco_counters* c, FILE* nonedge) {
    hco_edge e;
    char buf[512];
    /* the reference first scores the whole batch, then inserts, then writes: the
     * result is the same as doing all three per record in order (one thread). */
    for (uint64_t i = 0; i < n; i++) {
        int st = hco_compute_overlap(R, s, &batch[i].rec, &e);
        if (st) return st;
        c->scored++;
        if (e.cls == 2 || e.cls == 3) {
            hco_gedge ge;
            memset(&ge, 0, sizeof ge);
            const hco_overlap* o = &batch[i].rec;
            ge.score = e.score;
            ge.mismatch_rate = e.mismatch_rate;
            ge.pos1 = (int)o->pos1;
            ge.pos2 = (int)o->pos2;
            ge.pos3 = e.pos3;
            ge.pos4 = e.pos4;
            ge.ori1 = o->ori1;
            ge.ori2 = o->ori2;
            ge.ord = o->ord;
            ge.read1 = o->read1;
            ge.read2 = o->read2;
            if (s->flags & HCO_FLAG_ADD_DUPLICATES) { /* get_vertex_id(ori), :176-179 */
                ge.v1 = o->ori1 ? o->read1 : (uint64_t)R->n_reads + o->read1;
                ge.v2 = o->ori2 ? o->read2 : (uint64_t)R->n_reads + o->read2;
            } else { /* get_vertex_id(true), :181-182 */
                ge.v1 = o->read1;
                ge.v2 = o->read2;
            }
            ge.perc = (int)o->perc;
            if (e.n_subs == 1) { /* set_len(len1, 0) :227 ; set_len(len1, len2) :268 */
                ge.len0 = (int)o->len1;
                ge.len1 = (int)o->len1;
                ge.len2 = 0;
            } else {
                ge.len0 = (int)(o->len1 + o->len2);
                ge.len1 = (int)o->len1;
                ge.len2 = (int)o->len2;
            }
            if (ge.len1 <= 0 || ge.len2 < 0) return -20; /* Edge.h:213-214 asserts */
            st = hco_graph_insert(g, s, &ge, c);
            if (st) return st;
        } else if (e.cls == 1) {
            c->nonedges_written++;
            if (nonedge) {
                hco_overlap_get_line(&batch[i].line, buf, sizeof buf);
                fputs(buf, nonedge);
            }
        }
    }
    return 0;
}

/* EdgeCalculator.cpp:561-666 */
int hco_construct_edges(const hco_reads* R, const unsigned long* read_ids, const hco_settings* s,
                        const char* overlaps_path, const char* nonedge_path, hco_graph* g, hco_counters* c) {
    memset(c, 0, sizeof *c);
    FILE* f = fopen(overlaps_path, "r");
    if (!f) return -30; /* :662-665 */
    FILE* nonedge = nonedge_path ? fopen(nonedge_path, "a") : NULL;

    idmap_ent* map = (idmap_ent*)malloc(sizeof(idmap_ent) * (R->n_reads ? R->n_reads : 1));
    for (uint32_t i = 0; i < R->n_reads; i++) {
        map[i].id = read_ids[i];
        map[i].idx = i;
    }
    qsort(map, R->n_reads, sizeof(idmap_ent), idmap_cmp);

    const uint64_t per_vec = 1000000; /* :571 */
    pending* batch = (pending*)malloc(sizeof(pending) * per_vec);
    uint64_t nb = 0;
    hco_overlap_line* rejected = NULL;
    uint64_t nrej = 0, caprej = 0;
    char* line = NULL;
    size_t cap = 0;
    ssize_t got;
    uint64_t i = 0;
    int rc = 0;
    char* fields[16];
    while ((got = getline(&line, &cap, f)) >= 0 && i < s->max_overlaps) { /* :581 */
        i++;
        c->lines_read++;
        if (got > 0 && line[got - 1] == '\n') line[got - 1] = 0;
        int nf = hco_split_line(line, (s->flags & HCO_FLAG_ALLOW_SPACES) != 0, fields, 16);
        if (nf != 13) { /* :600-603 */
            c->malformed_lines++;
            continue;
        }
        hco_overlap_line o;
        rc = hco_overlap_from_fields((const char* const*)fields, &o);
        if (rc) { rc = -31; break; }
        if (o.id1 == o.id2) continue; /* :605-607 */
        unsigned int perc = hco_overlap_get_perc(&o);
        int pass = 0, store = 0;
        int ss = (o.type1 == 's' && o.type2 == 's');
        int anyp = (o.type1 == 'p' || o.type2 == 'p');
        if (o.len1 >= s->min_overlap_len && ss) { /* :612-617 */
            pass = perc >= s->min_overlap_perc;
        } else if (o.len1 >= 0.5 * s->min_overlap_len && o.len2 >= 0.5 * s->min_overlap_len && anyp) { /* :618-624 */
            pass = perc >= s->min_overlap_perc;
        } else if ((s->flags & HCO_FLAG_RELAX_PE_EDGES) && o.len1 + o.len2 >= s->min_overlap_len && anyp) { /* :626-632 */
            pass = perc >= s->min_overlap_perc;
        } else { /* :633-635 */
            store = 1;
        }
        if (store) {
            if (nrej == caprej) {
                caprej = caprej ? 2 * caprej : 1024;
                rejected = (hco_overlap_line*)realloc(rejected, caprej * sizeof(hco_overlap_line));
            }
            rejected[nrej++] = o;
            c->prefilter_rejected++;
        }
        if (pass) {
            long i1 = idmap_find(map, R->n_reads, o.id1);
            long i2 = idmap_find(map, R->n_reads, o.id2);
            if (i1 < 0 || i2 < 0) { rc = -32; break; } /* std::map::at throws, :170-171 */
            pending* p = &batch[nb++];
            p->line = o;
            p->rec.read1 = (uint32_t)i1;
            p->rec.read2 = (uint32_t)i2;
            p->rec.pos1 = o.pos1;
            p->rec.pos2 = o.pos2;
            p->rec.ori1 = o.ori1 == '+';
            p->rec.ori2 = o.ori2 == '+';
            p->rec.ord = (uint8_t)o.ord;
            p->rec.flags = (uint8_t)((o.type1 == 'p') | ((o.type2 == 'p') << 1));
            p->rec.len1 = o.len1;
            p->rec.len2 = o.len2;
            p->rec.perc = perc;
        }
        if (nb == per_vec) { /* :636-639 */
            rc = process_batch(R, s, batch, nb, g, c, nonedge);
            nb = 0;
            if (rc) break;
        }
    }
    if (!rc && nb > 0) rc = process_batch(R, s, batch, nb, g, c, nonedge); /* :641-644 */
    if (!rc && (s->flags & HCO_FLAG_ADD_DUPLICATES)) rc = hco_graph_add_equivalent_edges(g, R->n_reads); /* :650-652 */
    if (!rc && nonedge) { /* :654-660 */
        char buf[512];
        for (uint64_t k = 0; k < nrej; k++) {
            hco_overlap_get_line(&rejected[k], buf, sizeof buf);
            fputs(buf, nonedge);
        }
    }
    free(line);
    free(batch);
    free(rejected);
    free(map);
    fclose(f);
    if (nonedge) fclose(nonedge);
    return rc;
}
