#!/usr/bin/env python3
"""Generates tests/golden/compute_overlap.json with the reference's own compute_overlap (src/EdgeCalculator.cpp:143-385), through the one
probe of the edge calculation that holds NO substitute for anything the image lacks (oracle/_ref/libhcref_compute.so: lines 26-385 piped
verbatim behind an EdgeCalculator shell of two data members and four member-function declarations; genuine Types.h / Read.h / Edge.h /
Overlap.h / FastqStorage.h; no OverlapGraph shell, no std::vector<bool>, no build-owned trim — oracle/ref_compute_prelude.inc).

Runs only in the build container (needs /root/reference; `make -C oracle ref`).  Vectors: singles of 200..500 bp and 2 x 150 pairs over one
genome; candidates by geometry for all four type combinations (s-s, s-p, p-s, p-p with ord 1 and 2), each in all four orientation pairs, plus
positions at and beyond the end of read 1 (:76-79); five settings (the combination rule :254-261,292-299,353-360 on both sides of the threshold,
--min_read_len, --mismatch, --add_duplicates: vertices by orientation :176-179).  Stored: the 13 fields of every line and the Edge the reference
returned (score and mismatch rate as hex doubles, pos1..pos4, orientations, ord, vertices, perc, len0..len2).  Data only."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from haploconduct_amd import synth  # noqa: E402
from haploconduct_amd.records import OVERLAP_DTYPE  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "compute_overlap.json")
FIELDS = ["score", "mismatch_rate", "pos1", "pos2", "pos3", "pos4", "ori1", "ori2", "ord", "v1", "v2", "perc", "len0", "len1", "len2"]


class FragEdge(C.Structure):
    _fields_ = [("score", C.c_double), ("mismatch_rate", C.c_double), ("pos1", C.c_int32), ("pos2", C.c_int32), ("pos3", C.c_int32),
                ("pos4", C.c_int32), ("ori1", C.c_uint8), ("ori2", C.c_uint8), ("ord", C.c_uint8), ("pad", C.c_uint8), ("pad2", C.c_uint32),
                ("v1", C.c_uint64), ("v2", C.c_uint64), ("perc", C.c_int32), ("len0", C.c_int32), ("len1", C.c_int32), ("len2", C.c_int32)]


class FragSettings(C.Structure):
    _fields_ = [("edge_threshold", C.c_double), ("ov_threshold", C.c_double), ("merge_contigs", C.c_double), ("mismatch", C.c_double),
                ("min_read_len", C.c_uint32), ("flags", C.c_uint32)]


def candidates(reads, spos, ppos):
    ns = len(spos)
    rec = []

    def add(r1, r2, p1, p2, o, l1, l2):
        for o1 in (1, 0):
            for o2 in (1, 0):
                rec.append((r1, r2, p1, p2, o1, o2, ord(o), 0, l1, l2, 88))

    for i, (s, L) in enumerate(spos):  # s-s
        for j, (t, M) in enumerate(spos):
            if i != j and 0 <= t - s < L - 30:
                add(i, j, t - s, 0, "-", min(L - (t - s), M), 0)
    for i, (s, L) in enumerate(spos):  # s-p, p-s
        for j, (ps, ins) in enumerate(ppos):
            p1, p2 = ps - s, ps + ins - 150 - s
            if 0 <= p1 < L - 40 and 0 <= p2 < L - 40:
                add(i, ns + j, p1, p2, "-", min(L - p1, 150), min(L - p2, 150))
            q1, q2 = s - ps, ps + ins - 150 - s
            if 0 <= q1 < 110 and 0 <= q2 < L - 40:
                add(ns + j, i, q1, q2, "-", min(150 - q1, L), min(L - q2, 150))
    for i, (a, ia) in enumerate(ppos):  # p-p, ord by the sign of the /2 offset
        for j, (b, ib) in enumerate(ppos):
            d1, d2 = b - a, (b + ib) - (a + ia)
            if i != j and 0 <= d1 < 110 and abs(d2) < 110:
                add(ns + i, ns + j, d1, abs(d2), "1" if d2 >= 0 else "2", 150 - d1, 150 - abs(d2))
    # positions at and beyond the end of read 1 (overlap_score returns 0 and prints, :76-79), pos2 beyond a mate
    (s0, L0), (s1, L1) = spos[0], spos[1]
    for p in (L0 - 1, L0, L0 + 7):
        add(0, 1, p, 0, "-", 1, 0)
    add(ns, ns + 1, 10, 150, "1", 140, 1)
    add(ns, ns + 1, 149, 3, "2", 1, 147)
    return np.array(rec, dtype=OVERLAP_DTYPE)


def main():
    from test_gpu_parity import _mixed_reads

    lib_path = os.path.join(ROOT, "oracle", "_ref", "libhcref_compute.so")
    ref = C.CDLL(lib_path)
    ref.frag_compute_overlaps.restype = C.c_int
    ref.frag_compute_overlaps.argtypes = [C.POINTER(FragSettings), C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p]
    reads, spos, ppos = _mixed_reads(77, n_single=14, n_pair=16, glen=800)
    cand = candidates(reads, spos, ppos)
    lines = synth.records_to_lines(cand, reads)
    n_single = len(spos)
    seqs, quals = zip(*(reads.seq(q) for q in range(reads.n_seq)))
    S, Q = (C.c_char_p * len(seqs))(*seqs), (C.c_char_p * len(quals))(*quals)
    ids = np.ascontiguousarray(reads.read_ids, dtype=np.uint64)
    fields = [f.encode() for ln in lines for f in ln.split("\t")]
    L = (C.c_char_p * len(fields))(*fields)
    settings = {
        "default": dict(edge_threshold=0.97, ov_threshold=0.9, merge_contigs=0.0, mismatch=0.0, min_read_len=0, add_duplicates=0),
        "low_threshold": dict(edge_threshold=0.2, ov_threshold=0.05, merge_contigs=0.0, mismatch=0.0, min_read_len=0, add_duplicates=0),
        "min_read_len": dict(edge_threshold=0.97, ov_threshold=0.9, merge_contigs=0.0, mismatch=0.0, min_read_len=250, add_duplicates=0),
        "mismatch_setting": dict(edge_threshold=0.9, ov_threshold=0.3, merge_contigs=0.0, mismatch=0.02, min_read_len=0, add_duplicates=0),
        "add_duplicates": dict(edge_threshold=0.97, ov_threshold=0.9, merge_contigs=0.0, mismatch=0.0, min_read_len=0, add_duplicates=1),
    }
    out = {"source": "the reference's own compute_overlap (src/EdgeCalculator.cpp:26-385 piped verbatim, oracle/_ref/libhcref_compute.so: NO substitutes "
                     "in the probe) — tests/golden/make_golden_compute.py",
           "n_single": n_single, "n_paired": len(ppos), "read_ids": [int(x) for x in ids], "seqs": [s.decode() for s in seqs],
           "quals": [q.decode() for q in quals], "lines": lines,
           "record_fields": list(OVERLAP_DTYPE.names), "records": [[int(r[k]) for k in OVERLAP_DTYPE.names] for r in cand], "edge_fields": FIELDS, "settings": settings, "edges": {}}
    for name, st in settings.items():
        fs = FragSettings(st["edge_threshold"], st["ov_threshold"], st["merge_contigs"], st["mismatch"], st["min_read_len"], 2 if st["add_duplicates"] else 0)
        edges = (FragEdge * len(lines))()
        # the reference prints a line per overlap whose position lies at or beyond the end of read 1 (:77): not part of the vectors
        sys.stdout.flush()
        saved = os.dup(1)
        devnull = os.open(os.devnull, os.O_WRONLY)
        os.dup2(devnull, 1)
        try:
            rc = ref.frag_compute_overlaps(C.byref(fs), S, Q, ids.ctypes.data, n_single, len(ppos), L, len(lines), edges)
        finally:
            C.CDLL(None).fflush(None)
            os.dup2(saved, 1)
            os.close(saved)
            os.close(devnull)
        assert rc == 0
        out["edges"][name] = [[float(e.score).hex(), float(e.mismatch_rate).hex(), e.pos1, e.pos2, e.pos3, e.pos4, e.ori1, e.ori2, e.ord, int(e.v1), int(e.v2),
                               e.perc, e.len0, e.len1, e.len2] for e in edges]
    with open(OUT, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    sc = np.array([float.fromhex(e[0]) for e in out["edges"]["default"]])
    print(f"{len(lines)} lines x {len(settings)} settings -> {OUT} ({os.path.getsize(OUT)} bytes); default: {int((sc > 0.97).sum())} above the threshold, "
          f"{int((sc == 0).sum())} zero scores")


if __name__ == "__main__":
    main()
