"""Candidate generation on the device (hc_find_overlaps) against the brute-force oracle
(oracle/overlap_finder_oracle.py) on small seeded read sets, and through the whole front of the pipeline
(reads -> SFO records -> hc_sfo2overlaps -> overlaps file -> edge calculation) on a larger one."""
import os
import sys

import numpy as np
import pytest

import haploconduct_amd as hc
from haploconduct_amd import host, synth
from haploconduct_amd.records import SFO_DTYPE

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import overlap_finder_oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
COMP = np.zeros(256, np.uint8)
COMP[list(b"ACGTN")] = list(b"TGCAN")


def make_reads(seed, n_single, n_pair, glen, lo, hi, err, n_rate=0.0, rc_frac=0.5, repeat=False):
    rng = np.random.default_rng(seed)
    genome = ACGT[rng.integers(0, 4, glen)]
    if repeat:  # a tandem repeat: many seed hits per k-mer
        genome[glen // 3: glen // 3 + 120] = np.tile(ACGT[rng.integers(0, 4, 6)], 20)

    def piece(L):
        s = int(rng.integers(0, glen - L))
        seg = genome[s:s + L].copy()
        k = rng.random(L) < err
        seg[k] = ACGT[rng.integers(0, 4, int(k.sum()))]
        seg[rng.random(L) < n_rate] = ord("N")
        if rng.random() < rc_frac:
            seg = COMP[seg][::-1]
        return seg.tobytes(), b"I" * L

    singles = [piece(int(rng.integers(lo, hi))) for _ in range(n_single)]
    pairs = [(piece(int(rng.integers(lo, hi))), piece(int(rng.integers(lo, hi)))) for _ in range(n_pair)]
    return hc.ReadSet.from_lists(singles, pairs)


def as_tuples(recs):
    return [tuple(int(r[k]) for k in ("idA", "idB", "OHA", "OHB", "OLA", "OLB", "K", "inverted")) for r in recs]


@pytest.mark.parametrize("seed,err_rate,min_overlap,kw", [
    (1, 0.0, 40, {}),
    (2, 0.02, 50, {}),
    (3, 0.05, 60, dict(err=0.02)),
    (4, 0.0, 30, dict(n_rate=0.01)),
    (5, 0.02, 50, dict(repeat=True)),
    (6, 0.04, 80, dict(lo=80, hi=400, err=0.015)),
    (7, 0.0, 25, dict(lo=25, hi=60)),
])
def test_matches_brute_force(seed, err_rate, min_overlap, kw):
    args = dict(n_single=25, n_pair=20, glen=700, lo=60, hi=150, err=0.005)
    args.update(kw)
    reads = make_reads(100 + seed, **args)
    with hc.EdgeScorer(hc.Settings()) as sc:
        sc.set_reads(reads)
        for rev, inc in ((True, True), (False, True), (True, False)):
            got = as_tuples(sc.find_overlaps(err_rate, min_overlap, reversals=rev, inclusions=inc))
            want = O.find_overlaps(reads, err_rate, min_overlap, reversals=rev, inclusions=inc)
            assert got == want, (len(got), len(want), sorted(set(want) - set(got))[:5], sorted(set(got) - set(want))[:5])
            if rev and inc:
                assert len(want) > 50 and any(r[7] for r in want) and any(r[2] < 0 for r in want)


def test_wide_and_16_bit_symbol_stores_give_the_same_overlaps():
    reads = make_reads(31, n_single=30, n_pair=15, glen=600, lo=60, hi=140, err=0.004)
    rng = np.random.default_rng(3)
    want = None
    for n_qual in (5, 40, 58, 70):  # 8-bit, the two wide 8-bit encodings, 16-bit symbols: the finder reads the bases out of all four
        reads.quals = (33 + rng.integers(0, n_qual, reads.quals.size)).astype(np.uint8)
        with hc.EdgeScorer(hc.Settings()) as sc:
            sc.set_reads(reads)
            got = as_tuples(sc.find_overlaps(0.02, 50))
        if want is None:
            want = O.find_overlaps(reads, 0.02, 50)
        assert got == want and len(want) > 50


def test_one_context_takes_read_set_after_read_set(tmp_path):
    """A context keeps its grow-only scratch from read set to read set (hc_set_reads, round 6: a pipeline's stages on parked devices): a
    large set, a small one, a wide-alphabet one and the large one again on ONE context — finder records and the ingest's overlaps file each
    time equal to a fresh context's."""
    sets = [make_reads(41, n_single=400, n_pair=300, glen=4000, lo=80, hi=250, err=0.004),
            make_reads(42, n_single=12, n_pair=0, glen=500, lo=60, hi=120, err=0.0),
            make_reads(43, n_single=60, n_pair=90, glen=1500, lo=60, hi=140, err=0.01)]
    rng = np.random.default_rng(8)
    sets[2].quals = (33 + rng.integers(0, 40, sets[2].quals.size)).astype(np.uint8)
    order = [0, 1, 2, 0, 1]

    def run(sc, reads, out):
        sc.set_reads(reads)
        recs = as_tuples(sc.find_overlaps(0.02, 50))
        n_s = sum(1 for r in range(reads.n_reads) if not reads.is_paired(r))
        n_lines = sc.found_to_overlaps(out, n_s, reads.n_reads - n_s)
        return recs, n_lines, open(out, "rb").read()

    fresh = []
    for k in range(3):
        with hc.EdgeScorer(hc.Settings()) as sc:
            fresh.append(run(sc, sets[k], str(tmp_path / f"fresh{k}.txt")))
    assert len(fresh[0][0]) > 2000 and len(fresh[1][0]) < len(fresh[2][0]) < len(fresh[0][0])
    with hc.EdgeScorer(hc.Settings()) as sc:
        for i, k in enumerate(order):
            assert run(sc, sets[k], str(tmp_path / f"one{i}.txt")) == fresh[k], (i, k)


def test_argument_errors():
    reads = make_reads(5, n_single=4, n_pair=0, glen=300, lo=60, hi=100, err=0.0)
    with hc.EdgeScorer(hc.Settings()) as sc:
        with pytest.raises(hc.HcError):
            sc.find_overlaps(0.0, 30)  # no reads yet
        sc.set_reads(reads)
        with pytest.raises(hc.HcError):
            sc.find_overlaps(0.2, 20)  # an overlap of 20 with 4 errors need not hold 12 clean positions in a row
        with pytest.raises(hc.HcError):
            sc.find_overlaps(-0.1, 30)
        assert sc.find_overlaps(0.0, 5000).size == 0


def test_reads_to_graph_without_external_tools(oracle, tmp_path):
    """FASTQ -> hc_find_overlaps -> SFO text -> hc_sfo2overlaps -> overlaps file -> edge calculation: the front of the
    SAVAGE stage-a pipeline (savage.py:643-700) with nothing but this library.  Every overlap the generator planted
    between error-free reads must come out as an edge."""
    reads, meta = synth.make_paired_dataset(800, 2000, flip_frac=0.0, seed=41)
    reads.quals[:] = ord("I")
    n_pairs = reads.n_reads
    with hc.EdgeScorer(hc.Settings()) as sc:
        sc.set_reads(reads)
        recs = sc.find_overlaps(0.0, 75)
    assert recs.size > 2000
    # soundness on the host: every record is what it claims to be
    seqs = O.sfo_sequences(reads)
    for r in recs[:: max(1, recs.size // 500)]:
        A, B = seqs[r["idA"]], seqs[r["idB"]]
        if r["inverted"]:
            B = COMP[B][::-1]
        d = int(r["OHA"])
        start, end = max(0, d), min(A.size, d + B.size)
        assert end - start == r["OLA"] == r["OLB"] and r["OHB"] == d + B.size - A.size
        assert int(np.count_nonzero(A[start:end] != B[start - d:end - d])) == r["K"] == 0
    d_ = tmp_path
    host.write_sfo(str(d_ / "sfoverlaps.out"), recs)
    n_lines = host.sfo2overlaps(str(d_ / "sfoverlaps.out"), str(d_ / "overlaps.txt"), 0, n_pairs)
    assert n_lines > 200
    reads.write_fastq(None, str(d_ / "p1.fastq"), str(d_ / "p2.fastq"))
    st = hc.Settings(edge_threshold=0.97, min_overlap_len=150)
    out = d_ / "out"
    out.mkdir()
    with host.EdgeCalculatorStage(st, paired1=str(d_ / "p1.fastq"), paired2=str(d_ / "p2.fastq"), overlaps=str(d_ / "overlaps.txt"),
                                  output_dir=str(out) + "/") as ec:
        ec.construct_edges()
        edges = ec.edges()
    assert edges.size > 100
    # the same stage through the oracle on the same overlaps file
    rc, g, oc = oracle.construct_edges(reads, st, str(d_ / "overlaps.txt"), str(d_ / "ref_nonedge.txt"))
    want = g.all_edges()
    assert rc == 0 and want.size == edges.size
    assert np.array_equal(edges["score"].view(np.uint64), want["score"].view(np.uint64))
    assert np.array_equal(edges["v1"], want["v1"]) and np.array_equal(edges["v2"], want["v2"])


def test_batched_run_gives_the_same_records(monkeypatch):
    """HC_FIND_BATCH_HITS bounds the seed hits in flight: many small batches must give exactly the records of one."""
    reads = make_reads(77, n_single=60, n_pair=40, glen=900, lo=60, hi=160, err=0.004, repeat=True)
    with hc.EdgeScorer(hc.Settings()) as sc:
        sc.set_reads(reads)
        monkeypatch.delenv("HC_FIND_BATCH_HITS", raising=False)
        one = as_tuples(sc.find_overlaps(0.02, 50))
        monkeypatch.setenv("HC_FIND_BATCH_HITS", "1024")
        many = as_tuples(sc.find_overlaps(0.02, 50))
    assert one == many == O.find_overlaps(reads, 0.02, 50) and len(one) > 500


@pytest.mark.parametrize("seed", range(int(os.environ.get("HC_FUZZ_FINDER_SEEDS", "6"))))
def test_fuzz_against_brute_force(seed):
    """Random read sets (lengths, error and N rates, repeats, duplicated reads) and random (err_rate, min_overlap,
    flags) within what the exact seed filter accepts: the record lists must equal the brute-force oracle's."""
    rng = np.random.default_rng(5000 + seed)
    lo = int(rng.integers(30, 90))
    hi = lo + int(rng.integers(10, 200))
    reads = make_reads(6000 + seed, n_single=int(rng.integers(0, 40)), n_pair=int(rng.integers(1, 25)), glen=int(rng.integers(300, 1200)),
                       lo=lo, hi=hi, err=float(rng.choice([0.0, 0.004, 0.02])), n_rate=float(rng.choice([0.0, 0.005])),
                       rc_frac=float(rng.choice([0.0, 0.5])), repeat=bool(rng.integers(0, 2)))
    if rng.random() < 0.5:  # exact duplicates of some reads: d = 0 inclusions both ways, reported once
        k = reads.n_seq
        singles = [reads.seq(q) for q in range(min(5, k))]
        s0 = sum(1 for r in range(reads.n_reads) if not reads.is_paired(r))
        allsingles = [reads.seq(int(reads.read_first_seq[r])) for r in range(s0)] + singles
        pairs = [(reads.seq(int(reads.read_first_seq[r])), reads.seq(int(reads.read_first_seq[r]) + 1)) for r in range(s0, reads.n_reads)]
        reads = hc.ReadSet.from_lists(allsingles, pairs)
    min_overlap = int(rng.integers(20, lo + 1))
    err_rate = float(rng.choice([0.0, 0.01, 0.02, 0.04]))
    rev, inc = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    with hc.EdgeScorer(hc.Settings()) as sc:
        sc.set_reads(reads)
        try:
            got = as_tuples(sc.find_overlaps(err_rate, min_overlap, reversals=rev, inclusions=inc))
        except hc.HcError as e:
            assert "err_rate too high" in str(e)  # the filter refuses what it cannot do exactly
            return
    want = O.find_overlaps(reads, err_rate, min_overlap, reversals=rev, inclusions=inc)
    assert got == want, (len(got), len(want), sorted(set(want) - set(got))[:3], sorted(set(got) - set(want))[:3])


def test_real_reads_to_graph(oracle, tmp_path):
    """The excerpt of the reference's own SAVAGE example data (merged singles + 2x250 pairs with N and Q0 bases,
    tests/golden/savage_*.fastq.gz) from reads to graph with nothing but this library: device overlap finder at
    POLYTE's error rate, SFO ingest, overlaps file, edge-calculation stage — the stage compared with the oracle."""
    from tests.test_gpu_example_data import compare_stage, gunzip_to

    src = dict(singles=gunzip_to("savage_singles.fastq", str(tmp_path / "s0.fastq")),
               paired1=gunzip_to("savage_paired1.fastq", str(tmp_path / "a0.fastq")),
               paired2=gunzip_to("savage_paired2.fastq", str(tmp_path / "b0.fastq")))
    f = host.Fastq(**src)
    orig = f.readset()
    n_single, n_pairs = f.n_single, f.n_paired
    # the pipelines number the reads 0..n-1 (singles, then pairs) before rust-overlaps sees them (savage.py:643-664)
    reads = hc.ReadSet(orig.bases, orig.quals, orig.seq_off, orig.read_first_seq, np.arange(orig.n_reads, dtype=np.uint64))
    fq = dict(singles=str(tmp_path / "singles.fastq"), paired1=str(tmp_path / "paired1.fastq"), paired2=str(tmp_path / "paired2.fastq"))
    reads.write_fastq(fq["singles"], fq["paired1"], fq["paired2"])
    with hc.EdgeScorer(hc.Settings()) as sc:
        sc.set_reads(reads)
        recs = sc.find_overlaps(0.02, 100)
    assert recs.size > 3000 and recs["inverted"].any() and (recs["K"] > 0).any()
    n_lines = host.sfo_records_to_overlaps(recs, str(tmp_path / "overlaps.txt"), n_single, n_pairs)
    assert n_lines > 1000
    lines = (tmp_path / "overlaps.txt").read_text().split("\n")[:-1]
    types = {tuple(l.split("\t")[11:13]) for l in lines}
    assert ("s", "s") in types and len(types) >= 2  # single-single and overlaps involving pairs
    st = hc.Settings(edge_threshold=0.97, min_overlap_len=200)
    edges, c = compare_stage(oracle, tmp_path, st, fq, lines, "real")
    assert edges.size > 100 and c["scored"] > 500
    # and against the reference's own compute_overlap / process_overlaps where its probe library is present
    # (no prefilter on either side: process_overlaps sees every line)
    import ctypes as C
    import importlib.util

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib_path = os.path.join(root, "oracle", "_ref", "libhcref_edgecalc.so")
    if not os.path.exists(lib_path):
        return
    spec = importlib.util.spec_from_file_location("make_golden_ec", os.path.join(root, "tests", "golden", "make_golden_ec.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    from tests.test_ec_golden import compare_edges

    ref = C.CDLL(lib_path)
    ref.frag_process_overlaps.restype = C.c_int
    ref.frag_process_overlaps.argtypes = [C.POINTER(mg.FragSettings), C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p,
                                          C.c_uint64, C.c_char_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p,
                                          C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.c_void_p]
    ref.frag_ec_free.argtypes = [C.c_void_p]
    redges, rincl, rnonedge, rcounters = mg.run_probe(ref, reads, lines, dict(edge_threshold=0.97, ov_threshold=st.ov_threshold,
                                                                              merge_contigs=0.0, mismatch=0.0, min_read_len=0, ignore_inclusions=0))
    names = ["score", "mismatch_rate", "pos1", "pos2", "pos3", "pos4", "ori1", "ori2", "ord", "v1", "v2", "perc", "len0", "len1", "len2"]
    want = {k: [e[i] for e in redges] for i, k in enumerate(names)}
    for k in ("score", "mismatch_rate"):
        want[k] = np.array([float.fromhex(x) for x in want[k]], np.float64)
    st0 = hc.Settings(edge_threshold=0.97, ov_threshold=st.ov_threshold, min_overlap_len=0, min_overlap_perc=0)
    out = tmp_path / "out0"
    out.mkdir()
    with host.EdgeCalculatorStage(st0, overlaps=str(tmp_path / "overlaps.txt"), output_dir=str(out) + "/", **fq) as ec:
        ec.construct_edges()
        got, cnt = ec.edges(), ec.counters()
    compare_edges(got, want, "HIP stage vs the reference's own code on real reads")
    assert (out / "nonedge_overlaps.txt").read_text() == rnonedge
    assert cnt["inclusion_count"] == rcounters[0] and cnt["dup_count"] == rcounters[1] and len(redges) > 200


@pytest.mark.parametrize("case", range(6))
def test_ingest_from_the_device_records_writes_the_same_file(tmp_path, case):
    """hc_found_to_overlaps (flip + the script's sort on the device, matching on the host) against hc_sfo_records_to_overlaps
    on the fetched records (which tests/test_sfo_ingest.py pins to the reference's own scripts/sfo2overlaps.py): singles,
    pairs and both, reversals, mismatches, repeats; one, a few and many pieces of the sorted run."""
    kw = [dict(n_single=0, n_pair=700, glen=2500, lo=120, hi=150, err=0.0),
          dict(n_single=900, n_pair=0, glen=2500, lo=100, hi=300, err=0.01),
          dict(n_single=300, n_pair=500, glen=2500, lo=100, hi=200, err=0.01),
          dict(n_single=200, n_pair=600, glen=1500, lo=100, hi=160, err=0.02, repeat=True),
          dict(n_single=50, n_pair=900, glen=3000, lo=140, hi=150, err=0.0, rc_frac=0.0),
          dict(n_single=400, n_pair=400, glen=2000, lo=90, hi=250, err=0.02, n_rate=0.002)][case]
    reads = make_reads(100 + case, **kw)
    n_single, n_pairs = kw["n_single"], kw["n_pair"]
    err, t = (0.0, 60) if kw["err"] == 0.0 else (0.03, 70)
    with hc.EdgeScorer(hc.Settings()) as sc:
        sc.set_reads(reads)
        recs = sc.find_overlaps(err, t)
        assert recs.size > 500
        want_n = host.sfo_records_to_overlaps(recs, str(tmp_path / "want.txt"), n_single, n_pairs)
        want = (tmp_path / "want.txt").read_bytes()
        assert want_n == want.count(b"\n") and want_n > 50
        # pieces per chunk of the sorted run, records per chunk (the default takes these inputs in one chunk; 1 and 3: nearly
        # every group of paired candidates straddles a chunk border and waits in the carry)
        # ... and with / without the device's choice of the records the matching can see anything of (lines between unpaired reads,
        # groups of two lines and more, the lines that close them; HC_SFO_FILTER=0: every sorted record goes to the host)
        for pieces, chunk, filt in ((None, None, None), ("1", None, None), ("7", None, None), ("200", None, "0"), (None, "1", None), (None, "3", None),
                                    ("3", "1000", "0"), (None, "4096", None), (None, None, "0"), (None, "2", "0")):
            for k, v in (("HC_SFO_BUCKETS", pieces), ("HC_SFO_CHUNK", chunk), ("HC_SFO_FILTER", filt)):
                if v:
                    os.environ[k] = v
            try:
                got_n = sc.found_to_overlaps(tmp_path / "got.txt", n_single, n_pairs)
            finally:
                for k in ("HC_SFO_BUCKETS", "HC_SFO_CHUNK", "HC_SFO_FILTER"):
                    os.environ.pop(k, None)
            assert got_n == want_n and (tmp_path / "got.txt").read_bytes() == want, (pieces, chunk, filt)
        # ids that do not fit --num_singles / --num_pairs: the host path's diagnosis
        if n_pairs:
            with pytest.raises(hc.HcError) as ei:
                sc.found_to_overlaps(tmp_path / "bad.txt", n_single, n_pairs // 2)
            with pytest.raises(hc.HcError) as ej:
                host.sfo_records_to_overlaps(recs, str(tmp_path / "bad2.txt"), n_single, n_pairs // 2)
            assert ei.value.status == ej.value.status


def test_ingest_from_the_device_needs_found_records():
    with hc.EdgeScorer(hc.Settings()) as sc:
        sc.set_reads(make_reads(7, n_single=50, n_pair=0, glen=600, lo=100, hi=150, err=0.0))
        with pytest.raises(hc.HcError):
            sc.found_to_overlaps("/tmp/never_written.txt", 50, 0)


def test_ingest_from_the_device_with_nothing_found(tmp_path):
    rng = np.random.default_rng(5)
    n = 6
    bases = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=n * 120)
    reads = hc.ReadSet(bases, np.full(bases.size, ord("I"), np.uint8), np.arange(n + 1, dtype=np.uint64) * 120, np.arange(n + 1, dtype=np.uint32),
                       np.arange(n, dtype=np.uint64))
    with hc.EdgeScorer(hc.Settings()) as sc:
        sc.set_reads(reads)
        assert sc.find_overlaps(0.0, 60, count_only=True) == 0
        assert sc.found_to_overlaps(tmp_path / "none.txt", n, 0) == 0
    assert (tmp_path / "none.txt").read_bytes() == b""



@pytest.mark.parametrize("workload", ["c2", "mixed"])
def test_reads_to_graph_in_one_call_equals_the_file_route(tmp_path, workload):
    """hc_ec_construct_edges_from_reads — candidates found on the device, ingested where they are, the overlaps file's text
    from memory into the text blocks — against the route with files in between (hc_find_overlaps + hc_found_to_overlaps ->
    overlaps.txt -> hc_ec_construct_edges_sorted on a second stage): the same sorted graph, in-lists, inclusions,
    nonedge_overlaps.txt and counters.  C2 (50 000 read pairs of 2 x 150) and a small set of singles and pairs with
    reversals and mismatches."""
    if workload == "c2":
        import bench

        reads, cand, cfg, st = bench.build_workload("c2", 0)
        del cand
        n_single, n_pairs, err, t = 0, reads.n_reads, 0.0, 90
    else:
        kw = dict(n_single=300, n_pair=500, glen=2500, lo=100, hi=200, err=0.01)
        reads = make_reads(321, **kw)
        n_single, n_pairs, err, t = kw["n_single"], kw["n_pair"], 0.03, 70
        st = hc.Settings(edge_threshold=0.9, ov_threshold=0.5, min_overlap_len=0)
    st.n_threads = 8
    d = str(tmp_path) + "/"
    fq = dict(singles=d + "s.fastq" if n_single else None, paired1=d + "p1.fastq" if n_pairs else None, paired2=d + "p2.fastq" if n_pairs else None)
    reads.write_fastq(fq["singles"], fq["paired1"], fq["paired2"])
    out_a, out_b = tmp_path / "a", tmp_path / "b"
    out_a.mkdir()
    out_b.mkdir()
    with host.EdgeCalculatorStage(st, output_dir=str(out_a) + "/", **fq) as ec:  # no overlaps file at all
        n_found, n_lines = ec.construct_edges_from_reads(err, t)
        a = (ec.edges(), ec.in_lists(), ec.inclusions(), ec.counters())
    with hc.EdgeScorer(st) as sc:
        sc.set_reads(reads)
        assert sc.find_overlaps(err, t, count_only=True) == n_found
        assert sc.found_to_overlaps(d + "overlaps.txt", n_single, n_pairs) == n_lines
    with host.EdgeCalculatorStage(st, overlaps=d + "overlaps.txt", output_dir=str(out_b) + "/", **fq) as ec:
        ec.construct_edges_sorted()
        b = (ec.edges(), ec.in_lists(), ec.inclusions(), ec.counters())
    assert n_lines > 1000 and a[0].size == b[0].size > 500
    assert a[0].tobytes() == b[0].tobytes()
    assert np.array_equal(a[1][0], b[1][0]) and np.array_equal(a[1][1], b[1][1]) and np.array_equal(a[2], b[2])
    for k in ("inclusion_count", "dup_count", "edges_added", "nonedges_written", "prefilter_rejected", "lines_read", "scored", "self_overlap_count"):
        assert a[3][k] == b[3][k], k
    assert (out_a / "nonedge_overlaps.txt").read_bytes() == (out_b / "nonedge_overlaps.txt").read_bytes()
    # the one-call route gave its first text blocks row buffers for lines of 32 bytes while the finder ran (hc_textblock_reserve_rows):
    # even where every overlap is admitted no block had to grow inside its wait; the file route's blocks start with an eighth
    assert a[3]["host_blocks"] == 0 and a[3]["regrown_blocks"] == 0, a[3]
