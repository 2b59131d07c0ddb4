"""BASELINE.json configs[3] and configs[4] at the size bench.py builds them, on the GPU box.

c4: POLYTE diploid, 64 000 reads of 250 bp as singles, 1.26 M s-s candidates, 35 distinct quality values — the WIDE 8-bit
    symbol encoding with its 64 KiB log table and 1 024-lane workgroups, --edge_threshold 1.
c5: SAVAGE stage b/c, 60 000 singles of log-uniform length 150..6 000 bp, 2 M s-s candidates — a read set of mixed sequence
    length: launches bucket their candidates by (tile, length class) and the waves take groups from a queue
    (hc::bucket_perm_kernel + score_kernel_coop<..., DYN = true>).

For each: which kernel the library picked, size-independent properties over every record, a 50 000-candidate sample
bit-compared with the oracle, 150 000 lines through the REFERENCE'S OWN process_overlaps (fragment probe) and through
the HIP stage.  Then the bucketed launch under random shapes: lengths up to 6 000, singles and pairs mixed, candidate
counts around the tile size."""
import ctypes as C
import importlib.util
import os

import numpy as np
import pytest

import haploconduct_amd as hc
from haploconduct_amd import host, synth
from haploconduct_amd.records import OVERLAP_DTYPE, result_cls, result_n

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _workload(name):
    import bench

    return bench.build_workload(name, 0)


@pytest.fixture(scope="module")
def c4():
    reads, cand, cfg, st = _workload("c4")
    assert reads.n_reads == 64000 and cand.size > 1200000 and np.unique(reads.quals).size == 35
    return reads, cand, st


@pytest.fixture(scope="module")
def c5():
    reads, cand, cfg, st = _workload("c5")
    assert reads.n_reads == 60000 and cand.size == 2000000
    return reads, cand, st


def _properties_and_oracle_sample(oracle, reads, cand, st, expect_in_info, max_len):
    rng = np.random.default_rng(91)
    with hc.EdgeScorer(st) as sc:
        sc.set_reads(reads)
        info = sc.kernel_info()
        for piece in expect_in_info:
            assert piece in info, f"{piece!r} not in the kernel the library picked: {info}"
        cd = sc.pack_cands(cand)
        res = sc.score_cands(cd)
        # idempotence; the two record formats agree; every candidate is scored on its own (order independence)
        assert sc.score_cands(cd).tobytes() == res.tobytes()
        assert sc.score_batch(cand).tobytes() == res.tobytes()
        perm = rng.permutation(cand.size)
        sc.set_reorder(0)
        assert sc.score_cands(cd[perm]).tobytes() == res[perm].tobytes()
        sc.set_reorder(1)  # the device's own re-ordering in front of the launch
        assert sc.score_cands(cd[perm]).tobytes() == res[perm].tobytes()
        sc.set_reorder(2)
        # the stage's device leg: blocks in flight deliver exactly the non-dropped records, in order
        rows = sc.score_blocks(cd, block=250000, in_flight=3)
        kept = np.nonzero(result_cls(res) != 0)[0]
        assert np.array_equal(rows["index"], kept.astype(np.uint64))
        assert np.array_equal(rows["x1"].view(np.uint64), res["x1"][kept].view(np.uint64))
        assert np.array_equal(rows["n_cls"], res["n_cls"][kept])
        score, mrate, cls = sc.finalize(res)
    n, mm = result_n(res), res["mm"]
    assert (mm <= n).all() and (n >= 1).all() and (n <= max_len).all()
    assert (res["x1"] <= 0).all() and np.isnan(res["x2"]).all(), "s-s candidates have one sub-overlap"
    assert ((score >= 0) & (score <= 1)).all() and ((mrate >= 0) & (mrate <= 1)).all()
    assert (cls[mrate == 0] >= 2).all(), "merge_contigs=0 admits every zero-mismatch overlap (EdgeCalculator.cpp:407)"
    assert (score[cls == 2] > st.edge_threshold).all() and (score[cls == 1] > st.ov_threshold).all()
    dev = result_cls(res)
    assert ((dev == cls) | (dev == 4)).all()
    # total_len never exceeds the overlap the geometry allows (the file's LEN1 column of these synthetic candidates)
    assert (n <= cand["len1"]).all()
    idx = np.sort(rng.choice(cand.size, 50000, replace=False))
    ref = oracle.score_batch(reads, st, cand[idx], n_threads=os.cpu_count() or 1)
    assert (ref["status"] == 0).all()
    assert np.array_equal(ref["x1"].view(np.uint64), res["x1"][idx].view(np.uint64))
    assert np.array_equal(ref["n"], n[idx]) and np.array_equal(ref["mm"], mm[idx])
    assert np.array_equal(ref["cls"], cls[idx]) and np.array_equal(ref["score"].view(np.uint64), score[idx].view(np.uint64))
    assert np.array_equal(ref["mismatch_rate"].view(np.uint64), mrate[idx].view(np.uint64))
    return cls


def _slice_against_the_reference(reads, cand, st, tmp_path, lo, n_lines):
    lib_path = os.path.join(ROOT, "oracle", "_ref", "libhcref_edgecalc.so")
    if not os.path.exists(lib_path):
        pytest.skip("oracle/_ref/libhcref_edgecalc.so is built only where /root/reference exists")
    spec = importlib.util.spec_from_file_location("make_golden_ec", os.path.join(ROOT, "tests", "golden", "make_golden_ec.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    from tests.test_ec_golden import compare_edges

    part = cand[lo:lo + n_lines]
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
    assert len(edges) > 1000
    names = ["score", "mismatch_rate", "pos1", "pos2", "pos3", "pos4", "ori1", "ori2", "ord", "v1", "v2", "perc", "len0", "len1", "len2"]
    want = {k: [e[i] for e in edges] for i, k in enumerate(names)}
    for k in ("score", "mismatch_rate"):
        want[k] = np.array([float.fromhex(x) for x in want[k]], np.float64)
    d = str(tmp_path) + "/"
    reads.write_fastq(d + "singles.fastq", None, None)
    open(d + "slice.txt", "w").write("\n".join(lines) + "\n")
    out = tmp_path / "out"
    out.mkdir()
    st.min_overlap_len, st.min_overlap_perc = 0, 0
    st.n_threads = min(32, os.cpu_count() or 1)
    with host.EdgeCalculatorStage(st, singles=d + "singles.fastq", overlaps=d + "slice.txt", output_dir=str(out) + "/") as ec:
        ec.construct_edges()
        g, cnt = ec.edges(), ec.counters()
    compare_edges(g, want, "HIP stage vs the reference's own code")
    assert (out / "nonedge_overlaps.txt").read_text() == nonedge
    assert cnt["inclusion_count"] == counters[0] and cnt["dup_count"] == counters[1]


def test_full_size_c5_bucketed_kernel_properties_and_oracle_sample(oracle, c5):
    reads, cand, st = c5
    cls = _properties_and_oracle_sample(oracle, reads, cand, st, ["score_kernel_coop<uint8_t", "length-bucketed", "true, true", "waves_per_cu=8"], 6000)
    assert 10000 < int(((cls == 2) | (cls == 3)).sum()) < cand.size


def test_c5_slice_against_the_references_own_code(c5, tmp_path):
    reads, cand, st = c5
    _slice_against_the_reference(reads, cand, st, tmp_path, 900000, 150000)


def test_full_size_c4_wide_symbols_properties_and_oracle_sample(oracle, c4):
    reads, cand, st = c4
    cls = _properties_and_oracle_sample(oracle, reads, cand, st, ["score_kernel_coop<uint8_t, 6, 768", "encoding=wide8", "table_bytes=65536"], 250)
    # --edge_threshold 1: an edge is an overlap without a mismatch (polyte.py:617-626)
    assert not (cls == 2).any() and (cls == 3).any()


def test_c4_slice_against_the_references_own_code(c4, tmp_path):
    reads, cand, st = c4
    _slice_against_the_reference(reads, cand, st, tmp_path, 500000, 150000)


@pytest.mark.parametrize("which", ["c4", "c5"])
def test_whole_file_against_the_references_own_construct_edges_and_sort_edges(which, c4, c5, tmp_path):
    """Config 4 (1.26 * 10^6 lines, 35 quality values, --edge_threshold 1) and config 5 (2 * 10^6 lines, lengths 150..6 000) WHOLE through the
    REFERENCE'S OWN construct_edges + sortEdges and through hc_ec_construct_edges_sorted: one graph (tests/_refstage.py)."""
    from tests._refstage import whole_file_against_the_references_own_stage

    reads, cand, st = c4 if which == "c4" else c5
    d = str(tmp_path) + "/"
    host.write_overlaps(d + "overlaps.txt", cand, reads)
    reads.write_fastq(d + "singles.fastq", None, None)
    whole_file_against_the_references_own_stage(reads, st, d, d + "overlaps.txt", int(cand.size), int(cand.size), dict(singles=d + "singles.fastq"), min_edges=10000)


@pytest.mark.parametrize("seed", range(int(os.environ.get("HC_FUZZ_BUCKET_SEEDS", "10"))))
def test_fuzz_bucketed_launch_random_shapes(oracle, monkeypatch, seed):
    """The length-bucketed launch under random shapes: sequences of 30..6 000 symbols (log-uniform), singles, pairs or
    both, candidate counts below / at / above one tile of 4 096 and not a multiple of 64, true geometry and random
    positions, an already permuted batch.  Bit for bit the oracle."""
    monkeypatch.setenv("HC_BALANCE", "1")  # the library buckets only read sets of contig-length sequences: here every shape takes that launch
    rng = np.random.default_rng(7700 + seed)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    hi = int(rng.choice([700, 2000, 6000]))
    glen = 4 * hi + 2000
    genome = acgt[rng.integers(0, 4, glen)]
    K = int(rng.choice([4, 6, 12, 25, 40, 60]))
    alphabet = rng.choice(np.arange(33, 127), size=K, replace=False).astype(np.uint8)

    def length():
        return int(np.exp(rng.uniform(np.log(30), np.log(hi))))

    def piece(s, L):
        seg = genome[s:s + L].copy()
        k = rng.random(L) < 0.002
        seg[k] = acgt[rng.integers(0, 4, int(k.sum()))]
        seg[rng.random(L) < 0.001] = ord("N")
        return seg.tobytes(), alphabet[rng.integers(0, K, L)].tobytes()

    mode = seed % 3
    n_s = 0 if mode == 1 else int(rng.integers(40, 200))
    n_p = 0 if mode == 0 else int(rng.integers(40, 200))
    singles, pairs, geo = [], [], []
    for _ in range(n_s):
        L = length()
        s = int(rng.integers(0, glen - L))
        singles.append(piece(s, L))
        geo.append((s, L, s, L))
    for _ in range(n_p):
        L1, L2 = length(), length()
        ins = L1 + L2 + int(rng.integers(0, 300))
        s = int(rng.integers(0, max(1, glen - ins)))
        s2 = min(s + ins - L2, glen - L2)
        pairs.append((piece(s, L1), piece(s2, L2)))
        geo.append((s, L1, s2, L2))
    reads = hc.ReadSet.from_lists(singles, pairs)
    n = reads.n_reads
    m = int(rng.choice([63, 1000, 4095, 4096, 4097, 9001, 20000]))
    cand = np.zeros(m, OVERLAP_DTYPE)
    a = rng.integers(0, n, m)
    b = (a + 1 + rng.integers(0, n - 1, m)) % n
    cand["read1"], cand["read2"] = a, b
    g = np.array(geo)
    pa, pb = a >= n_s, b >= n_s
    true_geo = rng.random(m) < 0.7
    d1 = g[b, 0] - g[a, 0]
    d2 = g[b, 2] - g[a, 2]
    cand["pos1"] = np.where(true_geo & (d1 >= 0), d1, rng.integers(0, hi + 20, m))
    cand["pos2"] = np.where(pa | pb, np.where(true_geo & (d2 >= 0), d2, rng.integers(0, hi + 20, m)), 0)
    cand["ori1"] = np.where(true_geo, 1, rng.integers(0, 2, m))
    cand["ori2"] = np.where(true_geo, 1, rng.integers(0, 2, m))
    cand["ord"] = np.where(pa & pb, np.where(rng.random(m) < 0.5, ord("1"), ord("2")), ord("-"))
    cand["flags"] = pa.astype(np.uint8) | (pb.astype(np.uint8) << 1)
    cand["len1"], cand["len2"], cand["perc"] = 100, 100, 90
    st = hc.Settings(edge_threshold=float(rng.choice([0.9, 0.97, 0.995])), ov_threshold=float(rng.choice([0.0, 0.5, 0.9])),
                     merge_contigs=float(rng.choice([0.0, 0.01])), min_read_len=int(rng.choice([0, 0, 60])))
    ref = oracle.score_batch(reads, st, cand, n_threads=os.cpu_count() or 1)
    assert (ref["status"] == 0).all()
    with hc.EdgeScorer(st) as sc:
        sc.set_reads(reads)
        assert "length-bucketed" in sc.kernel_info()
        for reorder in (0, 1):
            sc.set_reorder(reorder)
            res = sc.score_batch(cand)
            assert np.array_equal(res["x1"].view(np.uint64), ref["x1"].view(np.uint64)), (seed, reorder)
            x2_nan = np.isnan(ref["x2"])
            assert np.array_equal(np.isnan(res["x2"]), x2_nan)
            assert np.array_equal(res["x2"].view(np.uint64)[~x2_nan], ref["x2"].view(np.uint64)[~x2_nan])
            assert np.array_equal(res["mm"], ref["mm"]) and np.array_equal(result_n(res), ref["n"])
            score, mrate, cls = sc.finalize(res)
            assert np.array_equal(cls, ref["cls"]) and np.array_equal(score.view(np.uint64), ref["score"].view(np.uint64))
        # the per-lane kernel with block-local balancing (what a store of 4 GiB and more takes) on the same shapes
        for fetch in ("2", "4"):
            os.environ["HC_FETCH_GROUP"] = fetch
            try:
                with hc.EdgeScorer(st) as lane:
                    lane.set_reads(reads)
                    assert "hc::score_kernel<" in lane.kernel_info()
                    assert lane.score_batch(cand).tobytes() == res.tobytes(), (seed, fetch)
            finally:
                os.environ.pop("HC_FETCH_GROUP", None)
        # the launch that also collects its rows (the multi-GPU payload): the same records, whatever their order
        kept = np.nonzero(result_cls(res) != 0)[0]
        rows = sc.score_blocks(sc.pack_cands(cand), block=5000, in_flight=2)
        assert np.array_equal(rows["index"], kept.astype(np.uint64))
