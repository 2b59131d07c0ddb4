"""ctypes face of oracle/liboracle.so (the CPU restatement; test infrastructure only)."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "liboracle.so")
lib = C.CDLL(LIB)


class hco_settings(C.Structure):
    _fields_ = [
        ("edge_threshold", C.c_double), ("ov_threshold", C.c_double), ("merge_contigs", C.c_double),
        ("mismatch", C.c_double), ("min_read_len", C.c_uint32), ("min_overlap_len", C.c_uint32),
        ("min_overlap_perc", C.c_uint32), ("flags", C.c_uint32), ("max_overlaps", C.c_uint64),
        ("device", C.c_int32), ("n_threads", C.c_uint32),
    ]


class hco_reads(C.Structure):
    _fields_ = [("bases", C.c_void_p), ("quals", C.c_void_p), ("seq_off", C.c_void_p),
                ("read_first_seq", C.c_void_p), ("n_reads", C.c_uint32)]


EDGE_DTYPE = np.dtype(
    [("score", "<f8"), ("mismatch_rate", "<f8"), ("x1", "<f8"), ("x2", "<f8"), ("ov1", "<f8"), ("ov2", "<f8"),
     ("mm", "<u4"), ("n", "<u4"), ("pos3", "<i4"), ("pos4", "<i4"), ("cls", "<u4"), ("n_subs", "<u4"),
     ("positions", "<u8"), ("status", "<i4"), ("_pad", "<i4")], align=False)
assert EDGE_DTYPE.itemsize == 88

GEDGE_DTYPE = np.dtype(
    [("score", "<f8"), ("mismatch_rate", "<f8"), ("pos1", "<i4"), ("pos2", "<i4"), ("pos3", "<i4"), ("pos4", "<i4"),
     ("ori1", "u1"), ("ori2", "u1"), ("ord", "u1"), ("pad", "u1"), ("read1", "<u4"), ("read2", "<u4"), ("_p2", "<u4"),
     ("v1", "<u8"), ("v2", "<u8"), ("perc", "<i4"), ("len0", "<i4"), ("len1", "<i4"), ("len2", "<i4")], align=False)
assert GEDGE_DTYPE.itemsize == 80


class hco_counters(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in
                ("self_overlap_count", "inclusion_count", "dup_count", "edges_added", "nonedges_written",
                 "prefilter_rejected", "malformed_lines", "lines_read", "scored")]


class hco_overlap_line(C.Structure):
    _fields_ = [("id1", C.c_ulong), ("id2", C.c_ulong), ("pos1", C.c_uint), ("pos2", C.c_uint),
                ("ord", C.c_char), ("ori1", C.c_char), ("ori2", C.c_char), ("type1", C.c_char), ("type2", C.c_char),
                ("perc1", C.c_uint), ("perc2", C.c_uint), ("len1", C.c_uint), ("len2", C.c_uint)]


lib.hco_phred_to_prob.restype = C.c_double
lib.hco_phred_to_prob.argtypes = [C.c_int]
lib.hco_score.restype = C.c_double
lib.hco_score.argtypes = [C.c_char, C.c_char, C.c_double, C.c_double, C.POINTER(C.c_int), C.c_double]
lib.hco_overlap_score.restype = C.c_double
lib.hco_overlap_score.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_char_p, C.c_char_p, C.c_uint,
                                  C.c_uint, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                  C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64),
                                  C.POINTER(C.c_int)]
lib.hco_build_rev_comp.restype = C.c_int
lib.hco_build_rev_comp.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
lib.hco_score_batch.restype = C.c_int
lib.hco_score_batch.argtypes = [C.POINTER(hco_reads), C.POINTER(hco_settings), C.c_void_p, C.c_uint64, C.c_void_p, C.c_int]
lib.hco_overlap_from_fields.restype = C.c_int
lib.hco_overlap_from_fields.argtypes = [C.POINTER(C.c_char_p), C.POINTER(hco_overlap_line)]
lib.hco_overlap_get_perc.restype = C.c_uint
lib.hco_overlap_get_perc.argtypes = [C.POINTER(hco_overlap_line)]
lib.hco_overlap_get_line.restype = C.c_int
lib.hco_overlap_get_line.argtypes = [C.POINTER(hco_overlap_line), C.c_char_p, C.c_size_t]
lib.hco_split_line.restype = C.c_int
lib.hco_split_line.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_char_p), C.c_int]
lib.hco_graph_new.restype = C.c_void_p
lib.hco_graph_new.argtypes = [C.c_uint64]
lib.hco_graph_free.argtypes = [C.c_void_p]
lib.hco_graph_edge_count.restype = C.c_uint64
lib.hco_graph_edge_count.argtypes = [C.c_void_p]
lib.hco_graph_out.restype = C.c_uint64
lib.hco_graph_out.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]
lib.hco_graph_inclusion.restype = C.c_int
lib.hco_graph_inclusion.argtypes = [C.c_void_p, C.c_uint64]
lib.hco_graph_insert.restype = C.c_int
lib.hco_graph_insert.argtypes = [C.c_void_p, C.POINTER(hco_settings), C.c_void_p, C.POINTER(hco_counters)]
lib.hco_construct_edges.restype = C.c_int
lib.hco_construct_edges.argtypes = [C.POINTER(hco_reads), C.c_void_p, C.POINTER(hco_settings), C.c_char_p, C.c_char_p,
                                    C.c_void_p, C.POINTER(hco_counters)]


def settings_to_c(s):
    return hco_settings(s.edge_threshold, s.ov_threshold, s.merge_contigs, s.mismatch, s.min_read_len,
                        s.min_overlap_len, s.min_overlap_perc, s.flags, s.max_overlaps, s.device, s.n_threads)


def reads_to_c(reads):
    return hco_reads(reads.bases.ctypes.data, reads.quals.ctypes.data, reads.seq_off.ctypes.data,
                     reads.read_first_seq.ctypes.data, reads.n_reads)


def score_batch(reads, settings, overlaps, n_threads=1):
    """hco_score_batch -> structured array of hco_edge."""
    ov = np.ascontiguousarray(overlaps)
    out = np.zeros(ov.shape[0], dtype=EDGE_DTYPE)
    cr, cs = reads_to_c(reads), settings_to_c(settings)
    rc = lib.hco_score_batch(C.byref(cr), C.byref(cs), ov.ctypes.data, ov.shape[0], out.ctypes.data, n_threads)
    assert rc == 0
    return out


def overlap_score(seq1, seq2, q1, q2, pos, min_read_len=0, mismatch=0.0):
    mr, x = C.c_double(), C.c_double()
    mm, n = C.c_uint32(), C.c_uint32()
    positions, st = C.c_uint64(), C.c_int()
    sc = lib.hco_overlap_score(seq1, len(seq1), seq2, len(seq2), q1, q2, pos, min_read_len, mismatch, C.byref(mr),
                               C.byref(x), C.byref(mm), C.byref(n), C.byref(positions), C.byref(st))
    return {"score": sc, "mismatch_rate": mr.value, "x": x.value, "mm": mm.value, "n": n.value,
            "positions": positions.value, "status": st.value}


def parse_fields(fields):
    arr = (C.c_char_p * 13)(*[f.encode() if isinstance(f, str) else f for f in fields])
    o = hco_overlap_line()
    rc = lib.hco_overlap_from_fields(arr, C.byref(o))
    if rc:
        return rc, None
    buf = C.create_string_buffer(512)
    lib.hco_overlap_get_line(C.byref(o), buf, 512)
    return 0, {"id1": o.id1, "id2": o.id2, "pos1": o.pos1, "pos2": o.pos2, "ord": o.ord.decode(),
               "ori1": o.ori1.decode(), "ori2": o.ori2.decode(), "type1": o.type1.decode(), "type2": o.type2.decode(),
               "perc": lib.hco_overlap_get_perc(C.byref(o)), "len1": o.len1, "len2": o.len2,
               "line": buf.value.decode()}


def split_line(line, allow_spaces=False):
    buf = C.create_string_buffer(line.encode() if isinstance(line, str) else line)
    fields = (C.c_char_p * 32)()
    n = lib.hco_split_line(buf, 1 if allow_spaces else 0, fields, 32)
    return n, [fields[i].decode() for i in range(min(n, 32))]


class Graph:
    def __init__(self, n_vertices):
        self.V = n_vertices
        self.g = C.c_void_p(lib.hco_graph_new(n_vertices))

    def __del__(self):
        if self.g and lib is not None:
            lib.hco_graph_free(self.g)
            self.g = None

    def edge_count(self):
        return int(lib.hco_graph_edge_count(self.g))

    def out_edges(self, v):
        p = C.c_void_p()
        n = lib.hco_graph_out(self.g, v, C.byref(p))
        if n == 0:
            return np.zeros(0, dtype=GEDGE_DTYPE)
        buf = (C.c_char * (n * GEDGE_DTYPE.itemsize)).from_address(p.value)
        return np.frombuffer(buf, dtype=GEDGE_DTYPE).copy()

    def all_edges(self):
        parts = [self.out_edges(v) for v in range(self.V)]
        return np.concatenate(parts) if parts else np.zeros(0, dtype=GEDGE_DTYPE)

    def inclusions(self):
        return np.array([lib.hco_graph_inclusion(self.g, v) for v in range(self.V)], dtype=np.uint8)

    def add_equivalent_edges(self, n_reads):
        return lib.hco_graph_add_equivalent_edges(self.g, n_reads)

    def insert(self, settings, gedge, counters):
        e = np.array([gedge], dtype=GEDGE_DTYPE) if not isinstance(gedge, np.ndarray) else gedge
        cs = settings_to_c(settings)
        return lib.hco_graph_insert(self.g, C.byref(cs), e.ctypes.data, C.byref(counters))


def construct_edges(reads, settings, overlaps_path, nonedge_path=None):
    g = Graph(reads.n_reads * (2 if settings.flags & 1 else 1))  # --add_duplicates: a second vertex per read, ViralQuasispecies.cpp:246-251
    c = hco_counters()
    cr, cs = reads_to_c(reads), settings_to_c(settings)
    ids = np.ascontiguousarray(reads.read_ids, dtype=np.uint64)
    rc = lib.hco_construct_edges(C.byref(cr), ids.ctypes.data, C.byref(cs), overlaps_path.encode(),
                                 nonedge_path.encode() if nonedge_path else None, g.g, C.byref(c))
    return rc, g, c
