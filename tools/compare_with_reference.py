#!/usr/bin/env python3
"""The whole bench workload (BASELINE config 2: 2 000 000 candidates) through the REFERENCE'S OWN compute_overlap /
process_overlaps (fragment probe oracle/_ref/libhcref_edgecalc.so, one thread: the configuration with a defined insert
order) and through the HIP stage; compares the two graphs edge by edge, bit by bit.  Takes about a minute of CPU."""
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
    from haploconduct_amd import host, synth
    from tests.test_ec_golden import compare_edges

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
    workload = sys.argv[2] if len(sys.argv) > 2 else "c2"
    spec = importlib.util.spec_from_file_location("make_golden_ec", os.path.join(ROOT, "tests", "golden", "make_golden_ec.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    reads, cand, cfg, st = bench.build_workload(workload, 0)
    cand = cand[:n]
    dup_frac = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    if dup_frac > 0:  # duplicates of random candidates shuffled in: the replace / keep decisions and the tie-break chain at scale
        rng = np.random.default_rng(11)
        extra = cand[rng.integers(0, cand.size, int(cand.size * dup_frac))]
        cand = np.concatenate([cand, extra])[rng.permutation(cand.size + extra.size)]
    lines = synth.records_to_lines(cand, reads)
    ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libhcref_edgecalc.so"))
    ref.frag_process_overlaps.restype = C.c_int
    ref.frag_process_overlaps.argtypes = [C.POINTER(mg.FragSettings), C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p,
                                          C.c_uint64, C.c_char_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p,
                                          C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.c_void_p]
    ref.frag_ec_free.argtypes = [C.c_void_p]
    ign = int(os.environ.get("HC_CMP_IGNORE_INCLUSIONS", "0"))  # --ignore_inclusions
    st.merge_contigs = float(os.environ.get("HC_CMP_MERGE_CONTIGS", st.merge_contigs))  # --merge_contigs
    st.mismatch = float(os.environ.get("HC_CMP_MISMATCH", st.mismatch))  # --mismatch
    st.min_read_len = int(os.environ.get("HC_CMP_MIN_READ_LEN", st.min_read_len))  # --min_read_len
    if ign:
        st.flags |= hc.records.FLAG_IGNORE_INCLUSIONS
    t0 = time.perf_counter()
    edges, incl, nonedge, counters = mg.run_probe(ref, reads, lines, dict(edge_threshold=st.edge_threshold, ov_threshold=st.ov_threshold,
                                                                          merge_contigs=st.merge_contigs, mismatch=st.mismatch,
                                                                          min_read_len=st.min_read_len, ignore_inclusions=ign))
    t_ref = time.perf_counter() - t0
    names = ["score", "mismatch_rate", "pos1", "pos2", "pos3", "pos4", "ori1", "ori2", "ord", "v1", "v2", "perc", "len0", "len1", "len2"]
    want = {k: [e[i] for e in edges] for i, k in enumerate(names)}
    for k in ("score", "mismatch_rate"):
        want[k] = np.array([float.fromhex(x) for x in want[k]], np.float64)
    d = tempfile.mkdtemp(prefix="hccmp_") + "/"
    host.write_overlaps(d + "overlaps.txt", cand, reads)
    paired = reads.is_paired(0)
    fq = dict(paired1=d + "p1.fastq", paired2=d + "p2.fastq") if paired else dict(singles=d + "s.fastq")
    reads.write_fastq(fq.get("singles"), fq.get("paired1"), fq.get("paired2"))
    st.min_overlap_len, st.min_overlap_perc, st.n_threads = 0, 0, 32
    t0 = time.perf_counter()
    with host.EdgeCalculatorStage(st, overlaps=d + "overlaps.txt", output_dir=d, **fq) as ec:
        ec.construct_edges()
        got, cnt, got_incl = ec.edges(), ec.counters(), ec.inclusions()
    t_hip = time.perf_counter() - t0
    assert got_incl.tolist() == list(incl), "inclusions bits differ"

    compare_edges(got, want, "HIP stage vs the reference's own code")
    assert open(d + "nonedge_overlaps.txt").read() == nonedge and cnt["dup_count"] == counters[1] and cnt["inclusion_count"] == counters[0]
    print(json.dumps({"workload": cfg["workload"], "candidates": int(cand.size), "duplicates_resolved": int(counters[1]), "edges": len(edges), "identical_graph": True, "inclusion_bits_set": int(sum(incl)), "merge_contigs": st.merge_contigs, "mismatch": st.mismatch, "min_read_len": st.min_read_len, "ignore_inclusions": ign,
                      "reference_process_overlaps_1_thread_s": round(t_ref, 2), "hip_stage_open_plus_construct_edges_s": round(t_hip, 3)}))
    import shutil
    shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
