"""Python face of include/hcedge_host.h: the host-side stage (FastqStorage / OverlapGraph /
EdgeCalculator mirrors in csrc/host/) and its GPU-free pieces."""
import ctypes as C

import numpy as np

from . import _native as N
from .records import OVERLAP_DTYPE, Settings

_vp = C.c_void_p

EDGE_DTYPE = np.dtype(
    [("score", "<f8"), ("mismatch_rate", "<f8"), ("pos1", "<i4"), ("pos2", "<i4"), ("pos3", "<i4"), ("pos4", "<i4"),
     ("ori1", "u1"), ("ori2", "u1"), ("ord", "u1"), ("pad", "u1"), ("read1", "<u4"), ("read2", "<u4"), ("_p2", "<u4"),
     ("v1", "<u8"), ("v2", "<u8"), ("perc", "<i4"), ("len0", "<i4"), ("len1", "<i4"), ("len2", "<i4")], align=False)
assert EDGE_DTYPE.itemsize == 80


class hc_ec_paths(C.Structure):
    _fields_ = [("singles_file", C.c_char_p), ("paired1_file", C.c_char_p), ("paired2_file", C.c_char_p),
                ("id_correspondence", C.c_char_p), ("overlaps_file", C.c_char_p), ("output_dir", C.c_char_p),
                ("max_reads", C.c_uint64)]


class hc_ec_counters(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in
                ("self_overlap_count", "inclusion_count", "dup_count", "edges_added", "nonedges_written",
                 "prefilter_rejected", "malformed_lines", "lines_read", "scored", "ambiguous", "silently_dropped")] + \
               [(k, C.c_double) for k in ("t_parse", "t_score", "t_insert", "t_write")] + \
               [(k, C.c_uint64) for k in ("device_blocks", "host_blocks", "regrown_blocks", "host_lines")]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class hc_overlap_fields(C.Structure):
    _fields_ = [("id1", C.c_uint64), ("id2", C.c_uint64), ("pos1", C.c_uint32), ("pos2", C.c_uint32),
                ("perc1", C.c_uint32), ("perc2", C.c_uint32), ("len1", C.c_uint32), ("len2", C.c_uint32),
                ("perc", C.c_uint32), ("ord", C.c_char), ("ori1", C.c_char), ("ori2", C.c_char), ("type1", C.c_char),
                ("type2", C.c_char), ("pad", C.c_char * 3)]


class hc_fastq_view(C.Structure):
    _fields_ = [("bases", _vp), ("quals", _vp), ("seq_off", _vp), ("read_first_seq", _vp), ("read_ids", _vp),
                ("n_reads", C.c_uint32), ("n_seq", C.c_uint32), ("n_single", C.c_uint32), ("n_paired", C.c_uint32)]


_sig = {
    "hc_ec_open": (C.c_int, [C.POINTER(_vp), C.POINTER(N.hc_settings), C.POINTER(hc_ec_paths)]),
    "hc_ec_construct_edges": (C.c_int, [_vp]),
    "hc_ec_construct_edges_sorted": (C.c_int, [_vp]),
    "hc_ec_construct_edges_from_reads": (C.c_int, [_vp, C.c_double, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "hc_ec_construct_edges_from_sfo": (C.c_int, [_vp, C.c_char_p, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]),
    "hc_ec_construct_edges_from_store": (C.c_int, [_vp, C.c_double, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                                    C.POINTER(C.c_int)]),
    "hc_ec_device_count": (C.c_uint32, [_vp]),
    "hc_ec_get_counters": (C.c_int, [_vp, C.POINTER(hc_ec_counters)]),
    "hc_ec_read_count": (C.c_uint64, [_vp]),
    "hc_ec_vertex_count": (C.c_uint64, [_vp]),
    "hc_ec_edge_count": (C.c_uint64, [_vp]),
    "hc_ec_get_edges": (C.c_int, [_vp, _vp, C.c_uint64, C.POINTER(C.c_uint64)]),
    "hc_ec_get_inclusions": (C.c_int, [_vp, _vp, C.c_uint64]),
    "hc_ec_overlap_score": (C.c_int, [_vp, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint32,
                                      C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "hc_ec_close": (C.c_int, [_vp]),
    "hc_ec_keep_devices": (C.c_int, [C.c_int]),
    "hc_host_split_line": (C.c_int, [C.c_char_p, C.c_uint64, C.c_int, _vp, _vp, C.c_int]),
    "hc_host_parse_overlap": (C.c_int, [C.c_char_p, C.c_uint64, C.c_int, C.POINTER(hc_overlap_fields), C.c_char_p]),
    "hc_host_fastq_load": (C.c_int, [C.POINTER(_vp), C.POINTER(hc_ec_paths), C.POINTER(hc_fastq_view)]),
    "hc_host_fastq_free": (C.c_int, [_vp]),
    "hc_host_parse_file": (C.c_int, [C.POINTER(N.hc_settings), _vp, C.c_char_p, _vp, C.c_uint64, C.POINTER(C.c_uint64),
                                     C.POINTER(hc_ec_counters)]),
    "hc_host_parse_text": (C.c_int, [C.POINTER(N.hc_settings), _vp, C.c_char_p, C.c_uint64, _vp, C.c_uint64, C.POINTER(C.c_uint64),
                                     C.POINTER(hc_ec_counters)]),
    "hc_sfo2overlaps": (C.c_int, [C.c_char_p, C.c_char_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]),
    "hc_sfo_records_to_overlaps": (C.c_int, [_vp, C.c_uint64, C.c_char_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]),
    "hc_host_write_sfo": (C.c_int, [C.c_char_p, _vp, C.c_uint64]),
    "hc_host_write_overlaps": (C.c_int, [C.c_char_p, _vp, C.c_uint64, _vp, _vp, C.c_uint64, C.c_uint32]),
    "hc_host_graph_new": (C.c_int, [C.POINTER(_vp), C.c_uint64, C.POINTER(N.hc_settings)]),
    "hc_host_graph_insert": (C.c_int, [_vp, _vp]),
    "hc_host_graph_resolve": (C.c_int, [_vp, _vp, C.c_uint64]),
    "hc_host_graph_adopt": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "hc_host_graph_add_equivalent_edges": (C.c_int, [_vp]),
    "hc_host_graph_sort_edges": (C.c_int, [_vp, _vp, C.c_uint64]),
    "hc_host_graph_get_in_lists": (C.c_int, [_vp, _vp, _vp, C.c_uint64]),
    "hc_ec_sort_edges": (C.c_int, [_vp]),
    "hc_ec_get_in_lists": (C.c_int, [_vp, _vp, _vp, C.c_uint64]),
    "hc_host_graph_get": (C.c_int, [_vp, _vp, C.c_uint64, C.POINTER(C.c_uint64), _vp, C.POINTER(hc_ec_counters)]),
    "hc_host_graph_free": (C.c_int, [_vp]),
}
for _name, (_res, _args) in _sig.items():
    _f = getattr(N.lib, _name)
    _f.restype = _res
    _f.argtypes = _args


def _b(s):
    return None if s is None else (s if isinstance(s, bytes) else str(s).encode())


def make_paths(singles=None, paired1=None, paired2=None, ids=None, overlaps=None, output_dir="", max_reads=0):
    return hc_ec_paths(_b(singles), _b(paired1), _b(paired2), _b(ids), _b(overlaps), _b(output_dir), max_reads)


def split_line(line, allow_spaces=False, max_fields=32):
    raw = line if isinstance(line, bytes) else line.encode()
    off = np.zeros(max_fields, np.uint32)
    ln = np.zeros(max_fields, np.uint32)
    n = N.lib.hc_host_split_line(raw, len(raw), 1 if allow_spaces else 0, off.ctypes.data, ln.ctypes.data, max_fields)
    return n, [raw[int(off[i]):int(off[i] + ln[i])].decode() for i in range(min(n, max_fields))]


def parse_overlap(line, allow_spaces=False, general_only=False):
    raw = line if isinstance(line, bytes) else line.encode()
    o = hc_overlap_fields()
    text = C.create_string_buffer(256)
    rc = N.lib.hc_host_parse_overlap(raw, len(raw), (1 if allow_spaces else 0) | (2 if general_only else 0), C.byref(o), text)
    if rc:
        return rc, None
    return 0, {"id1": o.id1, "id2": o.id2, "pos1": o.pos1, "pos2": o.pos2, "ord": o.ord.decode(), "ori1": o.ori1.decode(),
               "ori2": o.ori2.decode(), "type1": o.type1.decode(), "type2": o.type2.decode(), "perc": o.perc,
               "len1": o.len1, "len2": o.len2, "line": text.value.decode()}


def write_overlaps(path, recs, reads, n_threads=0):
    """Candidate records -> 13-column overlaps file (hc_host_write_overlaps): what synth.records_to_lines does, natively."""
    recs = np.ascontiguousarray(recs)
    ids = np.ascontiguousarray(reads.read_ids, dtype=np.uint64)
    rfs = np.asarray(reads.read_first_seq)
    paired = np.ascontiguousarray((rfs[1:] - rfs[:-1]) == 2, dtype=np.uint8)
    N.check(N.lib.hc_host_write_overlaps(_b(path), recs.ctypes.data, recs.size, ids.ctypes.data, paired.ctypes.data, ids.size, n_threads),
            "hc_host_write_overlaps")


def write_sfo(path, recs):
    """SFO records (records.SFO_DTYPE) as the text file rust-overlaps writes."""
    recs = np.ascontiguousarray(recs)
    N.check(N.lib.hc_host_write_sfo(_b(path), recs.ctypes.data, recs.size), "hc_host_write_sfo")


def sfo_records_to_overlaps(recs, out_path, num_singles, num_pairs):
    """SFO records straight to the 13-column overlaps file (hc_sfo_records_to_overlaps): write_sfo + sfo2overlaps
    without the intermediate text."""
    recs = np.ascontiguousarray(recs)
    n = C.c_uint64()
    N.check(N.lib.hc_sfo_records_to_overlaps(recs.ctypes.data, recs.size, _b(out_path), num_singles, num_pairs, C.byref(n)),
            "hc_sfo_records_to_overlaps")
    return int(n.value)


def sfo2overlaps(sfo_path, out_path, num_singles, num_pairs):
    """scripts/sfo2overlaps.py --in --out --num_singles --num_pairs, natively (hc_sfo2overlaps)."""
    n = C.c_uint64()
    N.check(N.lib.hc_sfo2overlaps(_b(sfo_path), _b(out_path), num_singles, num_pairs, C.byref(n)), "hc_sfo2overlaps")
    return int(n.value)


class Fastq:
    """FastqStorage alone (no device)."""

    def __init__(self, singles=None, paired1=None, paired2=None, ids=None, max_reads=0):
        self._h = _vp()
        self._paths = make_paths(singles, paired1, paired2, ids, max_reads=max_reads)
        v = hc_fastq_view()
        N.check(N.lib.hc_host_fastq_load(C.byref(self._h), C.byref(self._paths), C.byref(v)), "hc_host_fastq_load")
        self.n_reads, self.n_seq, self.n_single, self.n_paired = v.n_reads, v.n_seq, v.n_single, v.n_paired

        def arr(ptr, n, dt):
            if n == 0:
                return np.zeros(0, dt)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(dt))), shape=(n,)).copy()

        self.seq_off = arr(v.seq_off, v.n_seq + 1, np.uint64)
        total = int(self.seq_off[-1]) if v.n_seq else 0
        self.bases = arr(v.bases, total, np.uint8)
        self.quals = arr(v.quals, total, np.uint8)
        self.read_first_seq = arr(v.read_first_seq, v.n_reads + 1, np.uint32)
        self.read_ids = arr(v.read_ids, v.n_reads, np.uint64)

    def readset(self):
        from .readstore import ReadSet

        return ReadSet(self.bases, self.quals, self.seq_off, self.read_first_seq, self.read_ids)

    def parse_file(self, settings: Settings, overlaps_path):
        cs = settings.to_c()
        n = C.c_uint64()
        c = hc_ec_counters()
        N.check(N.lib.hc_host_parse_file(C.byref(cs), self._h, _b(overlaps_path), None, 0, C.byref(n), C.byref(c)),
                "hc_host_parse_file")
        out = np.zeros(n.value, dtype=OVERLAP_DTYPE)
        N.check(N.lib.hc_host_parse_file(C.byref(cs), self._h, _b(overlaps_path), out.ctypes.data, n.value, C.byref(n),
                                         C.byref(c)), "hc_host_parse_file")
        return out, c.as_dict()

    def parse_text(self, settings: Settings, text):
        """hc_host_parse_text: the overlaps file's text in memory through the same parser."""
        raw = text if isinstance(text, bytes) else text.encode()
        cs = settings.to_c()
        n = C.c_uint64()
        c = hc_ec_counters()
        N.check(N.lib.hc_host_parse_text(C.byref(cs), self._h, raw, len(raw), None, 0, C.byref(n), C.byref(c)), "hc_host_parse_text")
        out = np.zeros(n.value, dtype=OVERLAP_DTYPE)
        N.check(N.lib.hc_host_parse_text(C.byref(cs), self._h, raw, len(raw), out.ctypes.data, n.value, C.byref(n), C.byref(c)), "hc_host_parse_text")
        return out, c.as_dict()

    def close(self):
        if self._h:
            N.lib.hc_host_fastq_free(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HostGraph:
    """The serial insert of process_overlaps on a bare graph (no device)."""

    def __init__(self, n_vertices, settings: Settings):
        self._h = _vp()
        self.V = n_vertices
        self._cs = settings.to_c()
        N.check(N.lib.hc_host_graph_new(C.byref(self._h), n_vertices, C.byref(self._cs)), "hc_host_graph_new")

    def insert(self, edge_rec):
        e = np.ascontiguousarray(edge_rec, dtype=EDGE_DTYPE).reshape(1)
        return N.lib.hc_host_graph_insert(self._h, e.ctypes.data)

    def resolve(self, edge_recs):
        e = np.ascontiguousarray(edge_recs, dtype=EDGE_DTYPE)
        return N.lib.hc_host_graph_resolve(self._h, e.ctypes.data, e.shape[0])

    def adopt(self, edges, out_off, in_nodes, in_off, inclusions=None):
        """OverlapGraph::adopt_csr: the CSR form hc_graph_fetch returns, into this (empty) graph."""
        e = np.ascontiguousarray(edges, dtype=EDGE_DTYPE)
        oo, io = np.ascontiguousarray(out_off, np.uint64), np.ascontiguousarray(in_off, np.uint64)
        inn = np.ascontiguousarray(in_nodes, np.uint32)
        inc = None if inclusions is None else np.ascontiguousarray(inclusions, np.uint8)
        return N.lib.hc_host_graph_adopt(self._h, e.ctypes.data, oo.ctypes.data, inn.ctypes.data, io.ctypes.data, None if inc is None else inc.ctypes.data)

    def add_equivalent_edges(self):
        return N.lib.hc_host_graph_add_equivalent_edges(self._h)

    def sort_edges(self, len_by_read):
        """OverlapGraph::sortEdges (src/OverlapGraph.cpp:722-764); len_by_read[r] = total length of read r."""
        L = np.ascontiguousarray(len_by_read, dtype=np.uint32)
        N.check(N.lib.hc_host_graph_sort_edges(self._h, L.ctypes.data, L.size), "hc_host_graph_sort_edges")

    def in_lists(self, n_edges):
        off = np.zeros(self.V + 1, np.uint64)
        nodes = np.zeros(max(n_edges, 1), np.uint64)
        N.check(N.lib.hc_host_graph_get_in_lists(self._h, off.ctypes.data, nodes.ctypes.data, n_edges), "hc_host_graph_get_in_lists")
        return off, nodes[:n_edges]

    def get(self):
        n = C.c_uint64()
        c = hc_ec_counters()
        inc = np.zeros(self.V, np.uint8)
        N.check(N.lib.hc_host_graph_get(self._h, None, 0, C.byref(n), None, None), "hc_host_graph_get")
        out = np.zeros(n.value, dtype=EDGE_DTYPE)
        N.check(N.lib.hc_host_graph_get(self._h, out.ctypes.data, n.value, C.byref(n), inc.ctypes.data, C.byref(c)),
                "hc_host_graph_get")
        return out, inc, c.as_dict()

    def __del__(self):
        try:
            if self._h:
                N.lib.hc_host_graph_free(self._h)
        except Exception:
            pass


def keep_devices(on=True):
    """hc_ec_keep_devices: closed stages park their devices (contexts, text blocks, page-locked buffers) for the next one of this process."""
    N.check(N.lib.hc_ec_keep_devices(1 if on else 0), "hc_ec_keep_devices")


class EdgeCalculatorStage:
    """FastqStorage + OverlapGraph + EdgeCalculator, driven like src/ViralQuasispecies.cpp:233-283."""

    def __init__(self, settings: Settings, singles=None, paired1=None, paired2=None, ids=None, overlaps=None,
                 output_dir="", max_reads=0):
        self._h = _vp()
        self._cs = settings.to_c()
        self._paths = make_paths(singles, paired1, paired2, ids, overlaps, output_dir, max_reads)
        N.check(N.lib.hc_ec_open(C.byref(self._h), C.byref(self._cs), C.byref(self._paths)), "hc_ec_open")

    def construct_edges(self):
        N.check(N.lib.hc_ec_construct_edges(self._h), "hc_ec_construct_edges")

    def construct_edges_sorted(self):
        """construct_edges() + sortEdges() as one call (the lists arrive from the device in sortEdges order)."""
        N.check(N.lib.hc_ec_construct_edges_sorted(self._h), "hc_ec_construct_edges_sorted")

    def construct_edges_from_reads(self, err_rate, min_overlap, reversals=True, inclusions=True, sorted_order=True):
        """hc_ec_construct_edges_from_reads: candidates found on the device, ingested and scored without an overlaps file.
        Returns (SFO records found, overlap lines)."""
        nf, nl = C.c_uint64(), C.c_uint64()
        flags = (1 if reversals else 0) | (2 if inclusions else 0)
        N.check(N.lib.hc_ec_construct_edges_from_reads(self._h, float(err_rate), int(min_overlap), flags, 1 if sorted_order else 0,
                                                       C.byref(nf), C.byref(nl)), "hc_ec_construct_edges_from_reads")
        return nf.value, nl.value

    def construct_edges_from_sfo(self, sfo_path, sorted_order=True):
        """hc_ec_construct_edges_from_sfo: the SFO file rust-overlaps wrote -> graph (scripts/sfo2overlaps.py + the overlaps file + the text
        parser in one call; nothing but device memory in between for a canonical file).  Returns (SFO records, overlap lines, on_device)."""
        nr, nl, dr = C.c_uint64(), C.c_uint64(), C.c_int()
        N.check(N.lib.hc_ec_construct_edges_from_sfo(self._h, _b(sfo_path), 1 if sorted_order else 0, C.byref(nr), C.byref(nl), C.byref(dr)),
                "hc_ec_construct_edges_from_sfo")
        return nr.value, nl.value, bool(dr.value)

    def construct_edges_from_store(self, err_rate, min_overlap, reversals=True, inclusions=True, sorted_order=True):
        """hc_ec_construct_edges_from_store: the same with nothing but device memory between the reads and the graph (no text is written,
        copied or parsed).  Returns (SFO records found, overlap lines, whether the lines stayed on the device)."""
        nf, nl, dr = C.c_uint64(), C.c_uint64(), C.c_int()
        flags = (1 if reversals else 0) | (2 if inclusions else 0)
        N.check(N.lib.hc_ec_construct_edges_from_store(self._h, float(err_rate), int(min_overlap), flags, 1 if sorted_order else 0,
                                                       C.byref(nf), C.byref(nl), C.byref(dr)), "hc_ec_construct_edges_from_store")
        return nf.value, nl.value, bool(dr.value)

    def device_count(self):
        return int(N.lib.hc_ec_device_count(self._h))

    def counters(self):
        c = hc_ec_counters()
        N.check(N.lib.hc_ec_get_counters(self._h, C.byref(c)), "hc_ec_get_counters")
        return c.as_dict()

    def read_count(self):
        return int(N.lib.hc_ec_read_count(self._h))

    def edge_count(self):
        return int(N.lib.hc_ec_edge_count(self._h))

    def edges(self):
        n = C.c_uint64()
        N.check(N.lib.hc_ec_get_edges(self._h, None, 0, C.byref(n)), "hc_ec_get_edges")
        out = np.zeros(n.value, dtype=EDGE_DTYPE)
        N.check(N.lib.hc_ec_get_edges(self._h, out.ctypes.data, n.value, C.byref(n)), "hc_ec_get_edges")
        return out

    def vertex_count(self):
        return int(N.lib.hc_ec_vertex_count(self._h))

    def inclusions(self):
        out = np.zeros(self.vertex_count(), np.uint8)
        N.check(N.lib.hc_ec_get_inclusions(self._h, out.ctypes.data, out.size), "hc_ec_get_inclusions")
        return out

    def sort_edges(self):
        """overlap_graph->sortEdges(), the call after construct_edges in src/ViralQuasispecies.cpp:297."""
        N.check(N.lib.hc_ec_sort_edges(self._h), "hc_ec_sort_edges")

    def in_lists(self):
        n = self.edge_count()
        off = np.zeros(self.vertex_count() + 1, np.uint64)
        nodes = np.zeros(max(n, 1), np.uint64)
        N.check(N.lib.hc_ec_get_in_lists(self._h, off.ctypes.data, nodes.ctypes.data, n), "hc_ec_get_in_lists")
        return off, nodes[:n]

    def overlap_score(self, seq1, seq2, phred1, phred2, pos):
        sc, mr = C.c_double(), C.c_double()
        N.check(N.lib.hc_ec_overlap_score(self._h, _b(seq1), _b(seq2), _b(phred1), _b(phred2), pos, C.byref(sc),
                                          C.byref(mr)), "hc_ec_overlap_score")
        return sc.value, mr.value

    def close(self):
        if self._h:
            N.lib.hc_ec_close(self._h)
            self._h = _vp()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
