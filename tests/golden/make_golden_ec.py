#!/usr/bin/env python3
"""Generates tests/golden/ec/*.json with the reference's own compute_overlap / process_overlaps.

Runs only in the build container (needs /root/reference): `make -C oracle ref` compiles the EDGE-CALCULATION FRAGMENT
PROBE oracle/_ref/libhcref_edgecalc.so — lines 26-557 of src/EdgeCalculator.cpp (score, phred_to_prob, overlap_score,
compute_overlap, process_overlaps) and the OverlapGraph methods they call (src/OverlapGraph.cpp:83-101,150-229,285-319),
piped verbatim into g++ behind build-owned class shells (oracle/ref_ec_prelude.inc).  This script feeds it seeded read
sets and candidate lines (the 13 fields construct_edges hands to Overlap) and stores inputs + what the reference left
behind: the adjacency lists (list order), the inclusions bits, nonedge_overlaps.txt, inclusion_count and dup_count.
The vectors are data; no reference source is stored.
"""
import ctypes as C
import json
import os
import random
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import haploconduct_amd as hc  # noqa: E402
from haploconduct_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "ec")
_vp = C.c_void_p
HQ = np.array([2, 12, 20, 30, 37, 37, 37, 40, 40], dtype=np.uint8) + 33


class FragEdge(C.Structure):
    _fields_ = [("score", C.c_double), ("mismatch_rate", C.c_double), ("pos1", C.c_int32), ("pos2", C.c_int32), ("pos3", C.c_int32),
                ("pos4", C.c_int32), ("ori1", C.c_uint8), ("ori2", C.c_uint8), ("ord", C.c_uint8), ("pad", C.c_uint8), ("pad2", C.c_uint32),
                ("v1", C.c_uint64), ("v2", C.c_uint64), ("perc", C.c_int32), ("len0", C.c_int32), ("len1", C.c_int32), ("len2", C.c_int32)]


class FragSettings(C.Structure):
    _fields_ = [("edge_threshold", C.c_double), ("ov_threshold", C.c_double), ("merge_contigs", C.c_double), ("mismatch", C.c_double),
                ("min_read_len", C.c_uint32), ("ignore_inclusions", C.c_uint32)]


def run_probe(ref, reads, lines, st):
    n_single = sum(1 for r in range(reads.n_reads) if not reads.is_paired(r))
    n_paired = reads.n_reads - n_single
    seqs, quals = [], []
    for q in range(reads.n_seq):
        s, ql = reads.seq(q)
        seqs.append(s)
        quals.append(ql)
    S = (C.c_char_p * len(seqs))(*seqs)
    Q = (C.c_char_p * len(quals))(*quals)
    ids = np.ascontiguousarray(reads.read_ids, dtype=np.uint64)
    fields = [f.encode() for ln in lines for f in ln.split("\t")]
    assert len(fields) == 13 * len(lines)
    L = (C.c_char_p * len(fields))(*fields)
    dup = 1 if st.get("add_duplicates") else 0
    cap = len(lines) * (2 if dup else 1)
    edges = (FragEdge * cap)()
    n_edges, nb = C.c_uint64(), C.c_uint64()
    incl = np.zeros(reads.n_reads * (2 if dup else 1), np.uint8)  # --add_duplicates: a second vertex per read (reverse complement)
    text = _vp()
    counters = (C.c_uint32 * 2)()
    fs = FragSettings(st["edge_threshold"], st["ov_threshold"], st["merge_contigs"], st["mismatch"], st["min_read_len"], st["ignore_inclusions"] | (2 * dup))
    with tempfile.TemporaryDirectory() as d:
        rc = ref.frag_process_overlaps(C.byref(fs), S, Q, ids.ctypes.data, n_single, n_paired, L, len(lines), d.encode(), edges, cap,
                                       C.byref(n_edges), incl.ctypes.data, C.byref(text), C.byref(nb), counters)
    assert rc == 0 and n_edges.value <= cap
    nonedge = C.string_at(text, nb.value).decode()
    ref.frag_ec_free(text)
    out = []
    for e in edges[: n_edges.value]:
        out.append([float(e.score).hex(), float(e.mismatch_rate).hex(), e.pos1, e.pos2, e.pos3, e.pos4, e.ori1, e.ori2, e.ord, int(e.v1), int(e.v2),
                    e.perc, e.len0, e.len1, e.len2])
    return out, incl.tolist(), nonedge, [int(counters[0]), int(counters[1])]


def scenarios():
    rng = random.Random(5)
    # 1. pairs, all orientations, duplicates shuffled in (replace / keep / tie-break chain)
    reads, meta = synth.make_paired_dataset(70, 600, flip_frac=0.3, seed=31)
    reads.quals = HQ[np.random.default_rng(1).integers(0, HQ.size, reads.quals.size)]
    cand = synth.paired_candidates(meta, n_candidates=600, seed=32)
    lines = synth.records_to_lines(cand, reads)
    for ln in rng.sample(lines, len(lines) // 2):
        lines.insert(rng.randrange(len(lines)), ln)
    yield "pairs_dups", reads, lines, dict(edge_threshold=0.97, ov_threshold=0.5, merge_contigs=0.0, mismatch=0.0, min_read_len=0, ignore_inclusions=0)
    yield "pairs_merge_contigs", reads, lines, dict(edge_threshold=0.995, ov_threshold=0.9, merge_contigs=0.01, mismatch=0.0, min_read_len=0,
                                                    ignore_inclusions=1)
    yield "pairs_mismatch_setting", reads, lines[:500], dict(edge_threshold=0.9, ov_threshold=0.1, merge_contigs=0.0, mismatch=0.02, min_read_len=0,
                                                            ignore_inclusions=0)
    # 2. singles of mixed length: inclusions, strand twins (the same overlap seen from the other read on the other strand)
    reads, meta = synth.make_single_dataset(90, 1500, len_lo=120, len_hi=500, flip_frac=0.4, seed=33, quals=HQ, log_uniform=True)
    cand = synth.single_candidates(meta, min_overlap=60, n_candidates=900)
    lines = synth.records_to_lines(cand, reads)
    lens, ids = meta["lens"], reads.read_ids
    for r in cand[::3]:
        la, lb = int(lens[r["read1"]]), int(lens[r["read2"]])
        Lo = la - int(r["pos1"])
        if 0 < Lo <= lb:
            lines.insert(rng.randrange(len(lines)), "\t".join([
                str(int(ids[r["read2"]])), str(int(ids[r["read1"]])), str(lb - Lo), "-", "-", "-" if r["ori2"] else "+",
                "-" if r["ori1"] else "+", str(int(r["perc"])), "-", str(int(r["len1"])), "-", "s", "s"]))
    yield "singles_inclusions", reads, lines, dict(edge_threshold=0.995, ov_threshold=0.9, merge_contigs=0.0, mismatch=0.0, min_read_len=0,
                                                   ignore_inclusions=1)
    yield "singles_threshold_one", reads, lines, dict(edge_threshold=1.0, ov_threshold=0.5, merge_contigs=0.0, mismatch=0.0, min_read_len=0,
                                                      ignore_inclusions=0)
    yield "singles_min_read_len", reads, lines[:700], dict(edge_threshold=0.97, ov_threshold=0.2, merge_contigs=0.0, mismatch=0.0, min_read_len=200,
                                                          ignore_inclusions=0)
    # 2b. --add_duplicates (never set by the pipelines): vertices by orientation (EdgeCalculator.cpp:176-179), then
    # OverlapGraph::addEquivalentEdges (OverlapGraph.cpp:608-719) mirrors every edge onto the reverse-complement vertices
    yield "singles_add_duplicates", reads, lines, dict(edge_threshold=0.995, ov_threshold=0.9, merge_contigs=0.0, mismatch=0.0, min_read_len=0,
                                                       ignore_inclusions=1, add_duplicates=1)
    preads, pmeta = synth.make_paired_dataset(90, 700, flip_frac=0.4, seed=35)
    preads.quals = HQ[np.random.default_rng(2).integers(0, HQ.size, preads.quals.size)]
    pcand = synth.paired_candidates(pmeta, seed=36)
    plines = synth.records_to_lines(pcand, preads)
    for ln in rng.sample(plines, len(plines) // 3):
        plines.insert(rng.randrange(len(plines)), ln)
    yield "pairs_add_duplicates", preads, plines, dict(edge_threshold=0.97, ov_threshold=0.5, merge_contigs=0.0, mismatch=0.0, min_read_len=0,
                                                      ignore_inclusions=0, add_duplicates=1)
    # 3. singles and pairs together: s-p and p-s candidates by geometry, right and wrong orientations
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_parity import _mixed_reads
    from haploconduct_amd.records import OVERLAP_DTYPE
    reads, spos, ppos = _mixed_reads(61, n_single=40, n_pair=40, glen=900)
    ns = len(spos)
    rec = []
    for i, (s, L) in enumerate(spos):
        for j, (ps, ins) in enumerate(ppos):
            p1, p2 = ps - s, ps + ins - 150 - s
            if 0 <= p1 < L - 40 and 0 <= p2 < L - 40:
                rec.append((i, ns + j, p1, p2, 1, 1, ord("-"), 2, min(L - p1, 150), min(L - p2, 150), 90))
            q1, q2 = s - ps, ps + ins - 150 - s
            if 0 <= q1 < 110 and 0 <= q2 < L - 40:
                rec.append((ns + j, i, q1, q2, 1, 1, ord("-"), 1, min(150 - q1, L), min(L - q2, 150), 90))
    cand = np.array(rec, dtype=OVERLAP_DTYPE)
    flip = cand.copy()
    flip["ori1"] = 0
    cand = np.concatenate([cand, flip[: len(flip) // 3]])
    lines = synth.records_to_lines(cand, reads)
    yield "mixed_sp_ps", reads, lines, dict(edge_threshold=0.97, ov_threshold=0.3, merge_contigs=0.0, mismatch=0.0, min_read_len=0, ignore_inclusions=0)


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], stdout=subprocess.DEVNULL)
    ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libhcref_edgecalc.so"))
    ref.frag_process_overlaps.restype = C.c_int
    ref.frag_process_overlaps.argtypes = [C.POINTER(FragSettings), _vp, _vp, _vp, C.c_uint32, C.c_uint32, _vp, C.c_uint64, C.c_char_p, _vp,
                                          C.c_uint64, C.POINTER(C.c_uint64), _vp, C.POINTER(_vp), C.POINTER(C.c_uint64), _vp]
    ref.frag_ec_free.argtypes = [_vp]
    os.makedirs(OUT, exist_ok=True)
    for name, reads, lines, st in scenarios():
        edges, incl, nonedge, counters = run_probe(ref, reads, lines, st)
        n_single = sum(1 for r in range(reads.n_reads) if not reads.is_paired(r))
        case = {"source": "fragment probe of src/EdgeCalculator.cpp:26-557 + src/OverlapGraph.cpp:83-147,150-229,233-259,285-319,608-719 "
                          "(process_overlaps; with add_duplicates followed by addEquivalentEdges, whose edges carry no reverse offsets: stored as 0)",
                "settings": st, "n_single": n_single, "n_paired": reads.n_reads - n_single,
                "read_ids": [int(x) for x in reads.read_ids],
                "seqs": [reads.seq(q)[0].decode() for q in range(reads.n_seq)], "quals": [reads.seq(q)[1].decode() for q in range(reads.n_seq)],
                "lines": lines,
                "edge_fields": ["score", "mismatch_rate", "pos1", "pos2", "pos3", "pos4", "ori1", "ori2", "ord", "v1", "v2", "perc", "len0", "len1", "len2"],
                "edges": edges, "inclusions": incl, "nonedge_overlaps": nonedge, "inclusion_count": counters[0], "dup_count": counters[1]}
        json.dump(case, open(os.path.join(OUT, name + ".json"), "w"), separators=(",", ":"))
        print(name, "lines", len(lines), "edges", len(edges), "nonedge lines", nonedge.count("\n"), "inclusions", sum(incl), "counters", counters)


if __name__ == "__main__":
    main()
