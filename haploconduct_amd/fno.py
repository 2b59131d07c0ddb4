"""Find-next-overlaps (include/hcfno.h): ctypes view of hc_fno1_run / hc_fno3_run.

Mirrors SRBuilder::findNextOverlaps (src/FindNextOverlaps.cpp:890-958) and
SRBuilder::findNextOverlaps3 (src/FindNextOverlaps3.cpp:20-88) on flat numpy records.
The input builders here are shared with the tests, which feed the same structures to
the oracle (oracle/fno_oracle.cpp) — the structures are plain data, not the oracle.
"""
import time
import ctypes as C

import numpy as np

from . import _native as N

FNO_READ_DTYPE = np.dtype(
    [("id", "<u8"), ("len1", "<u4"), ("len2", "<u4"), ("paired", "u1"), ("visited", "u1"), ("orientation", "u1"), ("pad", "u1", (5,))]
)
FNO_EDGE_DTYPE = np.dtype(
    [("v1", "<u8"), ("v2", "<u8"), ("score", "<f8"), ("pos1", "<i4"), ("pos2", "<i4"), ("len1", "<i4"), ("len2", "<i4"), ("perc", "<i4"),
     ("ord", "u1"), ("ori1", "u1"), ("ori2", "u1"), ("pad", "u1")]
)
FNO_SUBREAD_DTYPE = np.dtype([("node", "<u8"), ("index1", "<i4"), ("index2", "<i4"), ("startpos1", "<i4"), ("startpos2", "<i4")])
FNO_ORIGINAL_DTYPE = np.dtype([("original_id", "<u8"), ("index1", "<i8"), ("index2", "<i8")])
assert FNO_READ_DTYPE.itemsize == 24 and FNO_EDGE_DTYPE.itemsize == 48
assert FNO_SUBREAD_DTYPE.itemsize == 24 and FNO_ORIGINAL_DTYPE.itemsize == 24

RESOLVE_ORIENTATIONS, NO_INCLUSIONS, OPTIMIZE = 1, 2, 4
ADD_DUPLICATES = 8  # program_settings.add_duplicates: a vertex per read and strand (include/hcfno.h)

_vp = C.c_void_p


class hc_fno1_input(C.Structure):
    _fields_ = [
        ("nodes", _vp), ("n_nodes", C.c_uint64),
        ("srs", _vp), ("n_srs", C.c_uint64),
        ("clique_off", _vp), ("clique_nodes", _vp),
        ("subread_off", _vp), ("subreads", _vp),
        ("graph_edges", _vp), ("n_graph_edges", C.c_uint64),
        ("branching_edges", _vp), ("n_branching_edges", C.c_uint64),
        ("nonedges", _vp), ("n_nonedges", C.c_uint64),
        ("inclusion_off", _vp), ("inclusion_edges", _vp), ("n_inclusion_groups", C.c_uint64),
        ("new_read_count", C.c_uint64),
        ("edge_threshold", C.c_double),
        ("flags", C.c_uint32), ("n_threads", C.c_uint32),
    ]


class hc_fno3_input(C.Structure):
    _fields_ = [
        ("srs", _vp), ("n_single", C.c_uint64), ("n_paired", C.c_uint64), ("n_trivial", C.c_uint64),
        ("orig_off", _vp), ("originals", _vp),
        ("new_read_count", C.c_uint64), ("original_readcount", C.c_uint64),
        ("flags", C.c_uint32), ("n_threads", C.c_uint32),
    ]


class hc_fno_counters(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("n_lines", "copied", "u2sr", "v2sr", "sr2sr", "candidates")]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


for _name, (_res, _args) in {
    "hc_fno1_run": (C.c_int, [C.POINTER(hc_fno1_input), C.POINTER(_vp)]),
    "hc_fno3_run": (C.c_int, [C.POINTER(hc_fno3_input), C.POINTER(_vp)]),
    "hc_fno_output_text": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(C.c_uint64)]),
    "hc_fno_output_counters": (C.c_int, [_vp, C.POINTER(hc_fno_counters)]),
    "hc_fno_output_write": (C.c_int, [_vp, C.c_char_p]),
    "hc_fno_output_free": (None, [_vp]),
    "hc_fno_output_on_device": (C.c_int, [_vp]),
    "hc_fno_compute_overlap_data": (C.c_int, [_vp, _vp, _vp, _vp, C.POINTER(C.c_int32), _vp]),
}.items():
    _f = getattr(N.lib, _name)
    _f.restype = _res
    _f.argtypes = _args


def _arr(a, dtype):
    a = np.ascontiguousarray(np.asarray(a, dtype=dtype))
    return a


def _ptr(a):
    return a.ctypes.data if a.size else None


def _csr(lists, dtype):
    off = np.zeros(len(lists) + 1, np.uint64)
    for i, l in enumerate(lists):
        off[i + 1] = off[i] + len(l)
    flat = np.concatenate([_arr(l, dtype) for l in lists]) if len(lists) and off[-1] else np.zeros(0, dtype)
    return off, _arr(flat, dtype)


class Fno1Input:
    """Owns the arrays behind an hc_fno1_input.

    nodes, srs: FNO_READ_DTYPE arrays; cliques: per super-read list of vertices; subreads: per super-read
    FNO_SUBREAD_DTYPE array; *_edges: FNO_EDGE_DTYPE arrays; inclusion_groups: list of FNO_EDGE_DTYPE arrays.
    """

    def __init__(self, nodes, srs, cliques, subreads, graph_edges, branching_edges=None, nonedges=None, inclusion_groups=(),
                 new_read_count=0, edge_threshold=0.97, flags=RESOLVE_ORIENTATIONS, n_threads=0):
        e0 = np.zeros(0, FNO_EDGE_DTYPE)
        self.nodes = _arr(nodes, FNO_READ_DTYPE)
        self.srs = _arr(srs, FNO_READ_DTYPE)
        self.clique_off, self.clique_nodes = _csr(list(cliques), np.uint64)
        self.subread_off, self.subreads = _csr(list(subreads), FNO_SUBREAD_DTYPE)
        self.graph_edges = _arr(graph_edges, FNO_EDGE_DTYPE)
        self.branching_edges = _arr(e0 if branching_edges is None else branching_edges, FNO_EDGE_DTYPE)
        self.nonedges = _arr(e0 if nonedges is None else nonedges, FNO_EDGE_DTYPE)
        self.inclusion_off, self.inclusion_edges = _csr(list(inclusion_groups), FNO_EDGE_DTYPE)
        self.n_inclusion_groups = len(inclusion_groups)
        self.new_read_count, self.edge_threshold, self.flags, self.n_threads = new_read_count, edge_threshold, flags, n_threads

    def struct(self):
        s = hc_fno1_input()
        s.nodes, s.n_nodes = _ptr(self.nodes), len(self.nodes)
        s.srs, s.n_srs = _ptr(self.srs), len(self.srs)
        s.clique_off, s.clique_nodes = self.clique_off.ctypes.data, _ptr(self.clique_nodes)
        s.subread_off, s.subreads = self.subread_off.ctypes.data, _ptr(self.subreads)
        s.graph_edges, s.n_graph_edges = _ptr(self.graph_edges), len(self.graph_edges)
        s.branching_edges, s.n_branching_edges = _ptr(self.branching_edges), len(self.branching_edges)
        s.nonedges, s.n_nonedges = _ptr(self.nonedges), len(self.nonedges)
        s.inclusion_off, s.inclusion_edges = self.inclusion_off.ctypes.data, _ptr(self.inclusion_edges)
        s.n_inclusion_groups = self.n_inclusion_groups
        s.new_read_count, s.edge_threshold = self.new_read_count, self.edge_threshold
        s.flags, s.n_threads = self.flags, self.n_threads
        return s


class Fno3Input:
    """Owns the arrays behind an hc_fno3_input.  srs = singles + paired + trivial super-reads, in that order."""

    def __init__(self, srs, n_single, n_paired, n_trivial, originals, new_read_count, original_readcount, flags=0, n_threads=0):
        self.srs = _arr(srs, FNO_READ_DTYPE)
        assert len(self.srs) == n_single + n_paired + n_trivial == len(originals)
        self.counts = (n_single, n_paired, n_trivial)
        self.orig_off, self.originals = _csr(list(originals), FNO_ORIGINAL_DTYPE)
        self.new_read_count, self.original_readcount, self.flags, self.n_threads = new_read_count, original_readcount, flags, n_threads

    def struct(self):
        s = hc_fno3_input()
        s.srs = _ptr(self.srs)
        s.n_single, s.n_paired, s.n_trivial = self.counts
        s.orig_off, s.originals = self.orig_off.ctypes.data, _ptr(self.originals)
        s.new_read_count, s.original_readcount = self.new_read_count, self.original_readcount
        s.flags, s.n_threads = self.flags, self.n_threads
        return s


last_on_device = False  # whether the latest output collected here came from the device form (hc_fno_output_on_device)
last_device_level = 0   # 0 host threads, 1 second half on the device, 2 the walk and the look-ups too (FNO=1)
last_run_s = 0.0        # seconds inside hc_fno1_run / hc_fno3_run alone (without this module's copy of the text into bytes)


def _collect(h, out_path):
    global last_on_device, last_device_level
    try:
        last_device_level = int(N.lib.hc_fno_output_on_device(h))
        last_on_device = last_device_level > 0
        text, n = _vp(), C.c_uint64()
        N.check(N.lib.hc_fno_output_text(h, C.byref(text), C.byref(n)), "hc_fno_output_text")
        data = C.string_at(text, n.value) if n.value else b""
        c = hc_fno_counters()
        N.check(N.lib.hc_fno_output_counters(h, C.byref(c)), "hc_fno_output_counters")
        if out_path is not None:
            N.check(N.lib.hc_fno_output_write(h, str(out_path).encode()), "hc_fno_output_write")
        return data, c.as_dict()
    finally:
        N.lib.hc_fno_output_free(h)


def find_next_overlaps(inp, out_path=None):
    """FNO=1.  Returns (text of overlaps.txt as bytes, counters dict); writes out_path when given."""
    global last_run_s
    s, h = inp.struct(), _vp()
    t0 = time.perf_counter()
    N.check(N.lib.hc_fno1_run(C.byref(s), C.byref(h)), "hc_fno1_run")
    last_run_s = time.perf_counter() - t0
    return _collect(h, out_path)


def find_next_overlaps3(inp, out_path=None):
    """FNO=3.  Returns (text of overlaps.txt as bytes, counters dict)."""
    global last_run_s
    s, h = inp.struct(), _vp()
    t0 = time.perf_counter()
    N.check(N.lib.hc_fno3_run(C.byref(s), C.byref(h)), "hc_fno3_run")
    last_run_s = time.perf_counter() - t0
    return _collect(h, out_path)


def compute_overlap_data(sr1, sr2, idx, edge):
    """computeOverlapData (src/FindNextOverlaps.cpp:351-565).  Returns (ok, out9 list)."""
    a, b = _arr([sr1], FNO_READ_DTYPE), _arr([sr2], FNO_READ_DTYPE)
    e = _arr([edge], FNO_EDGE_DTYPE)
    ix = _arr(idx, np.int32)
    out = np.zeros(9, np.int32)
    ok = C.c_int32()
    N.check(N.lib.hc_fno_compute_overlap_data(a.ctypes.data, b.ctypes.data, ix.ctypes.data, e.ctypes.data, C.byref(ok), out.ctypes.data),
            "hc_fno_compute_overlap_data")
    return int(ok.value), [int(x) for x in out]


def edges_from_records(recs, add_duplicates_reads=None):
    """hc_overlap_rec records (e.g. the lines of nonedge_overlaps.txt parsed by host.parse) -> FNO_EDGE_DTYPE
    with score 0: vertex = read index (Read::get_vertex_id(true) without --add_duplicates), perc = Overlap::get_perc.
    add_duplicates_reads = the number of reads: the vertices of --add_duplicates (src/FindNextOverlaps.cpp:672-675) — a read's own vertex
    for a '+' orientation, read index + number of reads for '-' (src/ViralQuasispecies.cpp:246-270)."""
    e = np.zeros(len(recs), FNO_EDGE_DTYPE)
    e["v1"], e["v2"] = recs["read1"], recs["read2"]
    if add_duplicates_reads is not None:
        e["v1"] += np.where(recs["ori1"] != 0, 0, add_duplicates_reads).astype(np.uint64)
        e["v2"] += np.where(recs["ori2"] != 0, 0, add_duplicates_reads).astype(np.uint64)
    e["pos1"], e["pos2"] = recs["pos1"], recs["pos2"]
    e["len1"], e["len2"], e["perc"] = recs["len1"], recs["len2"], recs["perc"]
    e["ori1"], e["ori2"] = recs["ori1"], recs["ori2"]
    e["ord"] = recs["ord"]
    return e
