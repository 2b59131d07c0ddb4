#!/usr/bin/env python3
"""OverlapGraph::sortEdges — the call after construct_edges — on the graph of a bench workload: the product
(hc_ec_sort_edges, vertex ranges on --threads threads) beside the reference's own sortEdges (fragment probe
oracle/_ref/libhcref_edgecalc.so: src/OverlapGraph.cpp:722-764 on a genuine graph built by addEdge calls), and a check
that both leave the same lists."""
import argparse
import ctypes as C
import importlib.util
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import bench
    import haploconduct_amd as hc
    from haploconduct_amd import host

    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--threads", type=int, default=32)
    ap.add_argument("--candidates", type=int, default=0)
    a = ap.parse_args()
    spec = importlib.util.spec_from_file_location("make_golden_ec", os.path.join(ROOT, "tests", "golden", "make_golden_ec.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    reads, cand, cfg, st = bench.build_workload(a.workload, 0)
    if a.candidates:
        cand = cand[: a.candidates]
    st.n_threads = a.threads
    st.flags |= hc.records.FLAG_RESOLVE_ORIENTATIONS
    d = tempfile.mkdtemp(prefix="hc_sort_") + "/"
    host.write_overlaps(d + "overlaps.txt", cand, reads)
    kw = dict(overlaps=d + "overlaps.txt", output_dir=d)
    if reads.is_paired(0):
        kw.update(paired1=d + "p1.fastq", paired2=d + "p2.fastq")
        reads.write_fastq(None, kw["paired1"], kw["paired2"])
    else:
        kw.update(singles=d + "s.fastq")
        reads.write_fastq(kw["singles"], None, None)
    with host.EdgeCalculatorStage(st, **kw) as ec:
        ec.construct_edges()
        before = ec.edges()
        t0 = time.perf_counter()
        ec.sort_edges()
        ours = time.perf_counter() - t0
        after = ec.edges()
        off, nodes = ec.in_lists()
    V = reads.n_reads
    seq_len = np.diff(np.asarray(reads.seq_off)).astype(np.uint32)
    total = np.add.reduceat(seq_len, np.asarray(reads.read_first_seq)[:-1].astype(np.int64)).astype(np.uint32)
    ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libhcref_edgecalc.so"))
    ref.frag_sort_edges.restype = C.c_int
    ref.frag_sort_edges.argtypes = [C.c_void_p] + [C.c_uint64, C.c_uint32] + [C.c_void_p] * 4
    ref.frag_last_sort_seconds.restype = C.c_double
    n = before.size
    fin = np.zeros(n, dtype=np.dtype(mg.FragEdge))
    for k in ("score", "mismatch_rate", "pos1", "pos2", "pos3", "pos4", "ori1", "ori2", "ord", "v1", "v2", "perc", "len0", "len1", "len2"):
        fin[k] = before[k]
    fout = np.zeros(n, dtype=fin.dtype)
    roff = np.zeros(V + 1, np.uint64)
    rnodes = np.zeros(max(n, 1), np.uint64)
    rc = ref.frag_sort_edges(fin.ctypes.data, n, V, total.ctypes.data, fout.ctypes.data, roff.ctypes.data, rnodes.ctypes.data)
    assert rc == 0, rc
    same = all(np.array_equal(fout[k], after[k]) for k in ("v1", "v2", "pos1", "pos2", "pos3", "pos4", "ori1", "ori2", "ord", "perc", "len0", "len1", "len2")) and \
        np.array_equal(fout["score"].view(np.uint64), after["score"].view(np.uint64)) and np.array_equal(roff, off) and np.array_equal(rnodes[:n], nodes)
    print(json.dumps({"workload": cfg["workload"], "edges": int(n), "vertices": int(V), "moved": int((after["v2"] != before["v2"]).sum()),
                      "sortEdges_s": round(ours, 4), "threads": a.threads, "reference_sortEdges_s": round(ref.frag_last_sort_seconds(), 4),
                      "identical_lists": bool(same)}))
    import shutil
    shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
