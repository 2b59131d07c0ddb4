"""A whole overlaps file through the REFERENCE'S OWN construct_edges + sortEdges (src/EdgeCalculator.cpp:561-666, src/OverlapGraph.cpp:
722-764: the fragment probe's frag_stage_sorted — its parser, `i < max_overlaps`, prefilter, OpenMP loop, serial insert with the tie-break
chain, sortEdges; nothing of this build inside it) and through hc_ec_construct_edges_sorted: the two graphs must be the same — every
out-list in list order with scores and mismatch rates as bit patterns, every in-list, inclusions, nonedge_overlaps.txt, the counters."""
import copy
import ctypes as C
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def whole_file_against_the_references_own_stage(reads, st, d, overlaps_path, n_lines, edge_cap, fastq_kw, min_edges, threads=32):
    """fastq_kw: singles= / paired1= / paired2= paths (the files the stage reads; the probe takes the reads as arrays).  n_lines: lines read
    (--max_ov on a longer file).  Returns the number of edges."""
    from haploconduct_amd import host
    from tests.test_gpu_reference_patch import _stage

    lib_path = os.path.join(ROOT, "oracle", "_ref", "libhcref_edgecalc_omp.so")
    if not os.path.exists(lib_path):
        pytest.skip("oracle/_ref/libhcref_edgecalc_omp.so is built only where /root/reference exists")
    spec = importlib.util.spec_from_file_location("make_golden_ec", os.path.join(ROOT, "tests", "golden", "make_golden_ec.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    os.environ["HCREF_THREADS"] = str(min(threads, os.cpu_count() or 1))
    seqs, quals = zip(*(reads.seq(q) for q in range(reads.n_seq)))
    S, Q = (C.c_char_p * len(seqs))(*seqs), (C.c_char_p * len(quals))(*quals)
    ids = np.ascontiguousarray(reads.read_ids, dtype=np.uint64)
    n_single = sum(1 for r in range(reads.n_reads) if not reads.is_paired(r))
    flags = 1 if (st.flags & 4) else 0  # ignore_inclusions (records.FLAG_IGNORE_INCLUSIONS)
    fs = mg.FragSettings(st.edge_threshold, st.ov_threshold, st.merge_contigs, st.mismatch, st.min_read_len, flags)
    pre = (C.c_uint32 * 3)(st.min_overlap_len, st.min_overlap_perc, 0)
    os.mkdir(d + "ref_stage")
    fastq = [fastq_kw.get("singles") or "None", fastq_kw.get("paired1") or "None", fastq_kw.get("paired2") or "None"]
    n_ref, want = _stage(C.CDLL(lib_path), "frag_stage_sorted", mg, fs, pre, S, Q, ids, n_single, reads.n_reads - n_single, fastq, overlaps_path,
                         d + "ref_stage", edge_cap, reads.n_reads, max_overlaps=n_lines)
    assert min_edges < n_ref < edge_cap
    fe = np.dtype({"names": ["score", "mismatch_rate", "pos1", "pos2", "pos3", "pos4", "ori1", "ori2", "ord", "v1", "v2", "perc", "len0", "len1", "len2"],
                   "formats": ["<f8", "<f8", "<i4", "<i4", "<i4", "<i4", "u1", "u1", "u1", "<u8", "<u8", "<i4", "<i4", "<i4", "<i4"],
                   "offsets": [0, 8, 16, 20, 24, 28, 32, 33, 34, 40, 48, 56, 60, 64, 68], "itemsize": C.sizeof(mg.FragEdge)})
    ref_edges = np.frombuffer(want[0], fe)
    os.mkdir(d + "hip_stage")
    st2 = copy.copy(st)
    st2.max_overlaps = n_lines
    st2.n_threads = min(32, os.cpu_count() or 1)
    with host.EdgeCalculatorStage(st2, overlaps=overlaps_path, output_dir=d + "hip_stage/", **fastq_kw) as ec:
        ec.construct_edges_sorted()
        got, (in_off, in_nodes), incl, cnt = ec.edges(), ec.in_lists(), ec.inclusions(), ec.counters()
    assert got.size == n_ref, f"{got.size} edges, the reference built {n_ref}"
    for k in ("score", "mismatch_rate"):
        assert np.array_equal(np.ascontiguousarray(got[k]).view(np.uint64), np.ascontiguousarray(ref_edges[k]).view(np.uint64)), f"{k} not bit-identical"
    for k in ("pos1", "pos2", "pos3", "pos4", "ori1", "ori2", "ord", "v1", "v2", "perc", "len0", "len1", "len2"):
        assert np.array_equal(np.asarray(got[k]).astype(np.int64), ref_edges[k].astype(np.int64)), f"{k} differs"
    assert in_off.tobytes() == want[1] and in_nodes.tobytes() == want[2], "in-lists differ"
    assert incl.tobytes() == want[3], "inclusions differ"
    assert open(d + "hip_stage/nonedge_overlaps.txt", "rb").read() == want[4], "nonedge_overlaps.txt differs"
    assert [cnt["inclusion_count"], cnt["dup_count"], cnt["self_overlap_count"]] == want[5]
    return n_ref
