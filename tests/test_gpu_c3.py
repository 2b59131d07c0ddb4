"""BASELINE.json configs[2] at full size on the GPU box: 500 000 synthetic 2x150 bp read pairs, 10^8 p-p candidate
overlaps (--max_ov's default, src/ViralQuasispecies.cpp:58) — the configuration the north star's target is quoted on and
the one bench.py times.  Size-independent properties over the whole batch, bit comparison of EVERY one of the 10^8 records with
the oracle, a 150 000-line slice against the reference's own code, and the whole stage (text file -> sorted graph) through
both duplicate-resolution routes."""
import os

import numpy as np
import pytest

import haploconduct_amd as hc
from haploconduct_amd import host, synth
from haploconduct_amd.records import result_cls, result_n

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def c3():
    import bench

    reads, cand, cfg, st = bench.build_workload("c3", 0)
    assert cand.size == 100000000 and reads.n_reads == 500000
    return reads, cand, st


def test_full_size_c3_properties_and_every_record_against_the_oracle(oracle, c3):
    reads, cand, st = c3
    rng = np.random.default_rng(31)
    with hc.EdgeScorer(st) as sc:
        sc.set_reads(reads)
        cd = sc.pack_cands(cand)
        res = sc.score_cands(cd)
        # idempotence, and the two record formats agree, on a tenth of the batch (2.4 GB of results per full pass)
        lo = 37000000
        part = slice(lo, lo + 10000000)
        assert sc.score_cands(cd[part]).tobytes() == res[part].tobytes()
        assert sc.score_batch(cand[part]).tobytes() == res[part].tobytes()
        # order independence: every candidate is scored on its own
        perm = rng.permutation(10000000)
        assert sc.score_cands(cd[part][perm]).tobytes() == res[part][perm].tobytes()
        # the stage's device leg: blocks in flight deliver exactly the non-dropped records, in order
        rows = sc.score_blocks(cd[part], block=250000, in_flight=4)
        kept = np.nonzero(result_cls(res[part]) != 0)[0]
        assert np.array_equal(rows["index"], kept.astype(np.uint64))
        assert np.array_equal(rows["x1"].view(np.uint64), res["x1"][part][kept].view(np.uint64))
        assert np.array_equal(rows["n_cls"], res["n_cls"][part][kept])
        score, mrate, cls = sc.finalize(res)
    # structural invariants over all 10^8 records
    n, mm = result_n(res), res["mm"]
    assert (mm <= n).all() and (n >= 1).all() and (n <= 150).all()
    assert ((res["x1"] <= 0) & (res["x2"] <= 0)).all()
    assert ((score >= 0) & (score <= 1)).all() and ((mrate >= 0) & (mrate <= 1)).all()
    assert (cls[mrate == 0] >= 2).all(), "merge_contigs=0 admits every zero-mismatch overlap (EdgeCalculator.cpp:407)"
    assert (score[cls == 2] > st.edge_threshold).all() and (score[cls == 1] > st.ov_threshold).all()
    dev = result_cls(res)
    assert ((dev == cls) | (dev == 4)).all()
    n_adm = int(((cls == 2) | (cls == 3)).sum())
    assert 3000000 < n_adm < 5000000
    # bit comparison with the oracle over ALL 10^8 records (round 3 compared a 0.05 % sample): x1, x2, mm, n, the class, the score and
    # the mismatch rate as bit patterns, in pieces of 5 * 10^6 candidates on the host's threads (the oracle: ~2 * 10^6 candidates/s)
    threads = min(64, os.cpu_count() or 1)
    piece = 5000000
    for lo in range(0, cand.size, piece):
        hi = min(cand.size, lo + piece)
        ref = oracle.score_batch(reads, st, cand[lo:hi], n_threads=threads)
        where = f"candidates [{lo}, {hi})"
        assert (ref["status"] == 0).all(), where
        assert np.array_equal(ref["x1"].view(np.uint64), res["x1"][lo:hi].view(np.uint64)), where
        assert np.array_equal(ref["x2"].view(np.uint64), res["x2"][lo:hi].view(np.uint64)), where
        assert np.array_equal(ref["n"], n[lo:hi]) and np.array_equal(ref["mm"], mm[lo:hi]), where
        assert np.array_equal(ref["cls"], cls[lo:hi]), where
        assert np.array_equal(ref["score"].view(np.uint64), score[lo:hi].view(np.uint64)), where
        assert np.array_equal(ref["mismatch_rate"].view(np.uint64), mrate[lo:hi].view(np.uint64)), where
        del ref


def _the_file_against_the_references_own_stage(reads, st, d):
    """ALL 10^8 lines of config 3 (HC_C3_REFERENCE_LINES: fewer, through --max_ov on the same file) through the REFERENCE'S OWN
    construct_edges + sortEdges (32 OpenMP threads: a minute and a half) and through hc_ec_construct_edges_sorted: one graph
    (tests/_refstage.py)."""
    from tests._refstage import whole_file_against_the_references_own_stage

    n_lines = int(os.environ.get("HC_C3_REFERENCE_LINES", str(10 ** 8)))
    n = whole_file_against_the_references_own_stage(reads, st, d, d + "overlaps.txt", n_lines, n_lines // 8, dict(paired1=d + "p1.fastq", paired2=d + "p2.fastq"),
                                                    min_edges=n_lines * 0.03)
    if n_lines == 10 ** 8:
        assert n == 3750623


def test_c3_stage_both_resolution_routes_and_a_slice_against_the_references_own_code(c3, tmp_path):
    """The whole stage at 10^8 lines: hc_ec_construct_edges_sorted with the device's duplicate resolution (the default)
    and with the host threads' (HC_RESOLVE=host) leave the same sorted graph; 150 000 lines out of the middle of the
    file go through the REFERENCE'S OWN process_overlaps (fragment probe) and through the HIP stage: same graph."""
    import ctypes as C
    import importlib.util

    reads, cand, st = c3
    d = str(tmp_path) + "/"
    host.write_overlaps(d + "overlaps.txt", cand, reads)
    reads.write_fastq(None, d + "p1.fastq", d + "p2.fastq")
    st.n_threads = min(32, os.cpu_count() or 1)
    got = {}
    for route in ("device", "host"):
        if route == "host":
            os.environ["HC_RESOLVE"] = "host"
        try:
            with host.EdgeCalculatorStage(st, paired1=d + "p1.fastq", paired2=d + "p2.fastq", overlaps=d + "overlaps.txt", output_dir=d) as ec:
                ec.construct_edges_sorted()
                c = ec.counters()
                got[route] = (ec.edges(), ec.in_lists(), c["scored"], c["dup_count"], c["inclusion_count"], c["nonedges_written"])
        finally:
            os.environ.pop("HC_RESOLVE", None)
    a, b = got["device"], got["host"]
    assert a[2] == b[2] == cand.size and a[3:] == b[3:]
    assert a[0].size == b[0].size > 3000000
    assert a[0].tobytes() == b[0].tobytes(), "sorted adjacency lists differ between the device's and the host's resolution"
    assert np.array_equal(a[1][0], b[1][0]) and np.array_equal(a[1][1], b[1][1])
    _the_file_against_the_references_own_stage(reads, st, d)
    os.remove(d + "overlaps.txt")

    lib_path = os.path.join(ROOT, "oracle", "_ref", "libhcref_edgecalc.so")
    if not os.path.exists(lib_path):
        pytest.skip("oracle/_ref/libhcref_edgecalc.so is built only where /root/reference exists")
    spec = importlib.util.spec_from_file_location("make_golden_ec", os.path.join(ROOT, "tests", "golden", "make_golden_ec.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    from tests.test_ec_golden import compare_edges

    part = cand[50000000:50150000]
    lines = synth.records_to_lines(part, reads)
    ref = C.CDLL(lib_path)
    ref.frag_process_overlaps.restype = C.c_int
    ref.frag_process_overlaps.argtypes = [C.POINTER(mg.FragSettings), C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p,
                                          C.c_uint64, C.c_char_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p,
                                          C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.c_void_p]
    ref.frag_ec_free.argtypes = [C.c_void_p]
    settings = dict(edge_threshold=st.edge_threshold, ov_threshold=st.ov_threshold, merge_contigs=st.merge_contigs, mismatch=st.mismatch,
                    min_read_len=st.min_read_len, ignore_inclusions=0)
    edges, incl, nonedge, counters = mg.run_probe(ref, reads, lines, settings)
    assert len(edges) > 3000
    names = ["score", "mismatch_rate", "pos1", "pos2", "pos3", "pos4", "ori1", "ori2", "ord", "v1", "v2", "perc", "len0", "len1", "len2"]
    want = {k: [e[i] for e in edges] for i, k in enumerate(names)}
    for k in ("score", "mismatch_rate"):
        want[k] = np.array([float.fromhex(x) for x in want[k]], np.float64)
    open(d + "slice.txt", "w").write("\n".join(lines) + "\n")
    out = tmp_path / "out"
    out.mkdir()
    st.min_overlap_len, st.min_overlap_perc = 0, 0
    with host.EdgeCalculatorStage(st, paired1=d + "p1.fastq", paired2=d + "p2.fastq", overlaps=d + "slice.txt", output_dir=str(out) + "/") as ec:
        ec.construct_edges()
        g, cnt = ec.edges(), ec.counters()
    compare_edges(g, want, "HIP stage vs the reference's own code, C3 slice")
    assert (out / "nonedge_overlaps.txt").read_text() == nonedge
    assert cnt["inclusion_count"] == counters[0] and cnt["dup_count"] == counters[1]
