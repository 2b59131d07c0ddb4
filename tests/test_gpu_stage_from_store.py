"""Stage a with nothing but device memory between the reads and the graph (round 6):
hc_ec_construct_edges_from_store — the finder's records stay on the device, the SFO ingest's flip, sort, MATCHING (scripts/sfo2overlaps.py:
63-103, 150-329) and both `uniq`s run there and leave the overlaps file's lines as parsed records, the text blocks take them as they are —
against today's routes, which are pinned against the reference's own code (hc_ec_construct_edges_from_reads: the same lines as text in
memory; hc_found_to_overlaps + hc_ec_construct_edges_sorted: as a file): graph in sortEdges order, in-lists, inclusions, counters and
nonedge_overlaps.txt, byte for byte.  Candidate generation itself (hc_find_overlaps for rust-overlaps): parity unpinned."""
import os

import numpy as np
import pytest

import haploconduct_amd as hc
from haploconduct_amd import host, records
from tests.test_gpu_example_data import gunzip_to
from tests.test_gpu_overlap_finder import make_reads

pytestmark = pytest.mark.gpu
COUNTERS = ("inclusion_count", "dup_count", "edges_added", "nonedges_written", "prefilter_rejected", "lines_read", "scored", "self_overlap_count",
            "silently_dropped")


def _both_routes(tmp_path, st, fq, err, min_overlap, tag, **find_kw):
    out = {}
    for route in ("store", "reads"):
        d = tmp_path / f"{tag}_{route}"
        d.mkdir()
        with host.EdgeCalculatorStage(st, output_dir=str(d) + "/", **fq) as ec:
            if route == "store":
                n_found, n_lines, on_device = ec.construct_edges_from_store(err, min_overlap, **find_kw)
                assert on_device, "the lines were expected to stay on the device"
            else:
                n_found, n_lines = ec.construct_edges_from_reads(err, min_overlap, **find_kw)
            out[route] = (n_found, n_lines, ec.edges(), ec.in_lists(), ec.inclusions(), ec.counters(), (d / "nonedge_overlaps.txt").read_bytes())
    a, b = out["store"], out["reads"]
    assert a[0] == b[0] and a[1] == b[1], (a[:2], b[:2])
    assert a[2].size == b[2].size and a[2].tobytes() == b[2].tobytes(), f"{tag}: the graphs differ"
    assert np.array_equal(a[3][0], b[3][0]) and np.array_equal(a[3][1], b[3][1]) and np.array_equal(a[4], b[4])
    for k in COUNTERS:
        if k in a[5]:
            assert a[5][k] == b[5][k], (tag, k, a[5][k], b[5][k])
    assert a[6] == b[6], f"{tag}: nonedge_overlaps.txt differs"
    assert a[5]["host_blocks"] == 0 and a[5]["host_lines"] == 0
    return a


def _write(reads, tmp_path, n_single, n_pairs):
    d = str(tmp_path) + "/"
    fq = dict(singles=d + "s.fastq" if n_single else None, paired1=d + "p1.fastq" if n_pairs else None, paired2=d + "p2.fastq" if n_pairs else None)
    reads.write_fastq(fq["singles"], fq["paired1"], fq["paired2"])
    return fq


def test_c2_pairs(tmp_path):
    import bench

    reads, cand, cfg, st = bench.build_workload("c2", 0)
    del cand
    st.n_threads = 8
    fq = _write(reads, tmp_path, 0, reads.n_reads)
    a = _both_routes(tmp_path, st, fq, 0.0, 90, "c2")
    assert a[1] > 20000 and a[2].size > 5000
    # ... and the route with a FILE in between (hc_found_to_overlaps -> overlaps.txt -> hc_ec_construct_edges_sorted), which the reference's own
    # construct_edges + sortEdges pin (tests/test_gpu_c3.py, test_gpu_golden_and_properties.py)
    d = str(tmp_path) + "/"
    with hc.EdgeScorer(st) as sc:
        sc.set_reads(reads)
        assert sc.find_overlaps(0.0, 90, count_only=True) == a[0]
        assert sc.found_to_overlaps(d + "overlaps.txt", 0, reads.n_reads) == a[1]
    os.mkdir(d + "file")
    with host.EdgeCalculatorStage(st, overlaps=d + "overlaps.txt", output_dir=d + "file/", **fq) as ec:
        ec.construct_edges_sorted()
        assert ec.edges().tobytes() == a[2].tobytes()
    assert open(d + "file/nonedge_overlaps.txt", "rb").read() == a[6]


@pytest.mark.parametrize("seed,kw,err,t,settings", [
    (321, dict(n_single=300, n_pair=500, glen=2500, lo=100, hi=200, err=0.01), 0.03, 70, dict(edge_threshold=0.9, ov_threshold=0.5, min_overlap_len=0)),
    # the prefilter at work (rejects go to nonedge_overlaps.txt), --max_ov cutting inside the lines, percentages
    (322, dict(n_single=400, n_pair=400, glen=2000, lo=90, hi=220, err=0.01), 0.03, 60, dict(edge_threshold=0.95, ov_threshold=0.6, min_overlap_len=150,
                                                                                               min_overlap_perc=40, max_overlaps=7001)),
    # singles only; pairs only with many candidates per pair of reads (short genome: groups of three lines and more)
    (323, dict(n_single=900, n_pair=0, glen=1500, lo=80, hi=160, err=0.005), 0.02, 50, dict(edge_threshold=0.97, min_overlap_len=100)),
    (324, dict(n_single=0, n_pair=700, glen=900, lo=100, hi=180, err=0.005, repeat=True), 0.04, 40, dict(edge_threshold=0.9, ov_threshold=0.3, min_overlap_len=60)),
    # relaxed paired-end prefilter, merge_contigs, inclusions ignored
    (325, dict(n_single=200, n_pair=600, glen=1800, lo=100, hi=200, err=0.01), 0.03, 60,
     dict(edge_threshold=0.995, min_overlap_len=220, merge_contigs=0.01, flags=records.FLAG_RESOLVE_ORIENTATIONS | records.FLAG_IGNORE_INCLUSIONS | records.FLAG_RELAX_PE_EDGES)),
])
def test_mixed_sets(tmp_path, seed, kw, err, t, settings):
    reads = make_reads(seed, **kw)
    st = hc.Settings(**settings)
    st.n_threads = 8
    fq = _write(reads, tmp_path, kw["n_single"], kw["n_pair"])
    a = _both_routes(tmp_path, st, fq, err, t, f"mixed{seed}")
    assert a[1] > 500
    # small blocks: the lines take several text blocks, --max_ov and the row buffers cross their borders
    old = os.environ.get("HC_TEXT_BLOCK")
    os.environ["HC_TEXT_BLOCK"] = "65536"
    try:
        b = _both_routes(tmp_path, st, fq, err, t, f"mixed{seed}_small")
    finally:
        if old is None:
            del os.environ["HC_TEXT_BLOCK"]
        else:
            os.environ["HC_TEXT_BLOCK"] = old
    assert b[2].tobytes() == a[2].tobytes() and b[6] == a[6]


def test_no_reversals_no_inclusions(tmp_path):
    kw = dict(n_single=300, n_pair=300, glen=2000, lo=100, hi=200, err=0.01)
    reads = make_reads(326, **kw)
    st = hc.Settings(edge_threshold=0.9, ov_threshold=0.5, min_overlap_len=0, n_threads=8)
    fq = _write(reads, tmp_path, 300, 300)
    _both_routes(tmp_path, st, fq, 0.02, 60, "norev", reversals=False)
    _both_routes(tmp_path, st, fq, 0.02, 60, "noinc", inclusions=False)


def test_add_duplicates_and_serial_insert(tmp_path):
    """--add_duplicates (vertices by orientation: the host resolves, addEquivalentEdges follows) and HC_INSERT_MODE=serial (the reference's
    per-edge insert) behind the device-resident lines: the routes behind the blocks do not care where the lines came from."""
    kw = dict(n_single=250, n_pair=350, glen=1800, lo=100, hi=200, err=0.01)
    reads = make_reads(327, **kw)
    fq = _write(reads, tmp_path, 250, 350)
    st = hc.Settings(edge_threshold=0.9, ov_threshold=0.5, min_overlap_len=100, flags=records.FLAG_ADD_DUPLICATES, n_threads=8)
    a = _both_routes(tmp_path, st, fq, 0.03, 60, "adddup")
    assert a[2].size > 200
    st = hc.Settings(edge_threshold=0.9, ov_threshold=0.5, min_overlap_len=100, n_threads=8)
    os.environ["HC_INSERT_MODE"] = "serial"
    try:
        b = _both_routes(tmp_path, st, fq, 0.03, 60, "serial")
    finally:
        del os.environ["HC_INSERT_MODE"]
    os.environ["HC_RESOLVE"] = "host"
    try:
        c = _both_routes(tmp_path, st, fq, 0.03, 60, "hostresolve")
    finally:
        del os.environ["HC_RESOLVE"]
    assert b[6] == c[6] and sorted(b[2].tolist()) == sorted(c[2].tolist()), "serial insert and bulk resolution: the same edges"


def test_savage_example_reads(tmp_path):
    """BASELINE config 1's reads (the whole savage/example/input_fas set: 2 000 merged singles + 200 pairs, 25 quality values), SAVAGE's
    stage a settings and finder arguments (savage.py:384,664: rust-overlaps ... 0.02 100)."""
    fq = dict(singles=gunzip_to("savage_singles.fastq", str(tmp_path / "singles.fastq")), paired1=gunzip_to("savage_paired1.fastq", str(tmp_path / "paired1.fastq")),
              paired2=gunzip_to("savage_paired2.fastq", str(tmp_path / "paired2.fastq")))
    for tag, st in (("a", hc.Settings(edge_threshold=0.97, min_overlap_len=200, n_threads=8)),
                    ("bc", hc.Settings(edge_threshold=0.995, min_overlap_len=100, merge_contigs=0.01, n_threads=8,
                                       flags=records.FLAG_RESOLVE_ORIENTATIONS | records.FLAG_IGNORE_INCLUSIONS))):
        a = _both_routes(tmp_path, st, fq, 0.02, 100, "savage_" + tag)
        assert a[1] > 20000 and a[2].size > 2000


def test_polyte_example_reads_as_singles(tmp_path):
    """The POLYTE example excerpt (35 quality values: the wide 8-bit table), every read a single (polyte.py:283-288), first-iteration settings."""
    recs = []
    for name in ("polyte_forward.fastq", "polyte_reverse.fastq"):
        L = open(gunzip_to(name, str(tmp_path / name))).read().split("\n")
        for i in range(0, len(L) - 3, 4):
            recs.append((L[i + 1], L[i + 3]))
    s = str(tmp_path / "singles.fastq")
    with open(s, "w") as o:
        for i, (seq, q) in enumerate(recs):
            o.write(f"@{i}\n{seq}\n+\n{q}\n")
    st = hc.Settings(edge_threshold=0.95, min_overlap_len=127, n_threads=8)
    a = _both_routes(tmp_path, st, dict(singles=s), 0.02, 80, "polyte")
    assert a[1] > 300


def _line_text(l):
    ss = l["type1"] == ord("s") and l["type2"] == ord("s")
    f = [str(int(l["id1"])), str(int(l["id2"])), str(int(l["pos1"])), "-" if ss else str(int(l["pos2"])), chr(l["ord"]), chr(l["ori1"]), chr(l["ori2"]),
         str(int(l["perc1"])), "-" if ss else str(int(l["perc2"])), str(int(l["len1"])), "-" if ss else str(int(l["len2"])), chr(l["type1"]), chr(l["type2"])]
    return "\t".join(f)


@pytest.mark.parametrize("seed,ns,npairs,n_rec", [(1, 6, 5, 3000), (2, 0, 8, 4000), (3, 9, 0, 1500), (4, 40, 60, 20000), (5, 3, 3, 200), (6, 2, 2, 2)])
def test_the_scripts_matching_on_the_device_equals_the_host_matcher(tmp_path, seed, ns, npairs, n_rec):
    """hc_found_to_lines_device on hand-made SFO records — few reads, so that one pair of reads has many lines (groups of dozens: every two
    of them are matched), repeated records (the script's first `uniq`), self overlaps, both orientations, every type combination, the
    closing line's types deciding a group's (scripts/sfo2overlaps.py:94), the last group never matched — against the host's matcher
    (hc_sfo_records_to_overlaps, pinned by the script itself: tests/golden/sfo), line for line."""
    from haploconduct_amd.records import SFO_DTYPE

    rng = np.random.default_rng(seed)
    n_seq = ns + 2 * npairs
    recs = np.zeros(n_rec, dtype=SFO_DTYPE)
    recs["idA"] = rng.integers(0, n_seq, n_rec)
    recs["idB"] = rng.integers(0, n_seq, n_rec)
    recs["OHA"] = rng.integers(-40, 41, n_rec)
    recs["OHB"] = rng.integers(-40, 41, n_rec)
    recs["OLA"] = rng.integers(30, 151, n_rec)
    recs["OLB"] = np.where(rng.random(n_rec) < 0.7, recs["OLA"], rng.integers(30, 151, n_rec))
    recs["K"] = rng.integers(0, 4, n_rec)
    recs["inverted"] = rng.integers(0, 2, n_rec)
    dup = rng.integers(0, n_rec, n_rec // 10)  # exact repeats
    recs[rng.integers(0, n_rec, dup.size)] = recs[dup]
    n_reads = ns + npairs
    singles = [(b"ACGT" * 10, b"I" * 40)] * ns
    pairs = [((b"ACGT" * 10, b"I" * 40), (b"TGCA" * 10, b"I" * 40))] * npairs
    reads = hc.ReadSet.from_lists(singles, pairs)
    assert reads.n_reads == n_reads
    want_path = str(tmp_path / "want.txt")
    n_want = host.sfo_records_to_overlaps(recs, want_path, ns, npairs)
    want = open(want_path).read().splitlines()
    assert len(want) == n_want
    with hc.EdgeScorer(hc.Settings()) as sc:
        sc.set_reads(reads)
        sc.set_found_records(recs)
        got = [_line_text(l) for l in sc.found_to_lines(ns, npairs)]
        # the records are where hc_found_to_overlaps looks for them too: the device sort + host matcher route writes the same file
        assert sc.found_to_overlaps(str(tmp_path / "mixed.txt"), ns, npairs) == n_want
    assert open(str(tmp_path / "mixed.txt")).read().splitlines() == want
    assert len(got) == len(want), (len(got), len(want))
    for k, (g, w) in enumerate(zip(got, want)):
        assert g == w, (k, g, w)
    if n_rec >= 1500:
        assert len(want) > 200


def test_an_assert_of_the_script_is_left_to_the_host(tmp_path):
    """A record with an empty read (the script divides by zero, scripts/sfo2overlaps.py:193): the device says so instead of guessing, and
    hc_ec_construct_edges_from_store takes the host's route for such an input."""
    from haploconduct_amd.records import SFO_DTYPE

    recs = np.zeros(3, dtype=SFO_DTYPE)
    recs["idA"], recs["idB"] = [0, 1, 0], [1, 2, 2]
    recs["OLA"] = recs["OLB"] = [50, 0, 60]
    recs["OHA"] = recs["OHB"] = [5, 0, 7]
    reads = hc.ReadSet.from_lists([(b"ACGT" * 10, b"I" * 40)] * 3, [])
    with hc.EdgeScorer(hc.Settings()) as sc:
        sc.set_reads(reads)
        sc.set_found_records(recs)
        with pytest.raises(Exception, match="not on the device"):
            sc.found_to_lines(3, 0)
        with pytest.raises(Exception):  # the host's matcher raises what the script raises
            sc.found_to_overlaps(str(tmp_path / "x.txt"), 3, 0)


@pytest.mark.parametrize("seed", range(int(os.environ.get("HC_FUZZ_STORE_SEEDS", "6"))))  # soak: HC_FUZZ_STORE_SEEDS=200
def test_fuzz_both_routes(tmp_path, seed):
    """Random read sets (singles and / or pairs, lengths, error rates, N bases, repeats), random finder arguments and stage settings: the
    device-resident route and the text-in-memory route leave the same graph, counters and nonedge_overlaps.txt."""
    rng = np.random.default_rng(7000 + seed)
    shape = int(rng.integers(0, 4))
    n_single = 0 if shape == 1 else int(rng.integers(50, 500))
    n_pair = 0 if shape == 2 else int(rng.integers(50, 500))
    lo = int(rng.integers(40, 120))
    kw = dict(n_single=n_single, n_pair=n_pair, glen=int(rng.integers(600, 3000)), lo=lo, hi=lo + int(rng.integers(10, 150)),
              err=float(rng.choice([0.0, 0.005, 0.02])), n_rate=float(rng.choice([0.0, 0.0, 0.01])), repeat=bool(rng.integers(0, 2)))
    reads = make_reads(9000 + seed, **kw)
    flags = records.FLAG_RESOLVE_ORIENTATIONS
    if rng.integers(0, 3) == 0:
        flags |= records.FLAG_IGNORE_INCLUSIONS
    if rng.integers(0, 3) == 0:
        flags |= records.FLAG_RELAX_PE_EDGES
    st = hc.Settings(edge_threshold=float(rng.choice([0.5, 0.9, 0.97, 0.995])), ov_threshold=float(rng.choice([0.1, 0.5, 0.9])),
                     min_overlap_len=int(rng.choice([0, 60, 150, 260])), min_overlap_perc=int(rng.choice([0, 0, 30, 60])),
                     merge_contigs=float(rng.choice([0.0, 0.0, 0.01, 0.05])), flags=flags,
                     max_overlaps=int(rng.choice([100000000, 100000000, 37, 2500])), n_threads=int(rng.choice([1, 4, 8])))
    fq = _write(reads, tmp_path, n_single, n_pair)
    old = os.environ.get("HC_TEXT_BLOCK")
    if rng.integers(0, 2):
        os.environ["HC_TEXT_BLOCK"] = str(int(rng.choice([16384, 65536, 1 << 20])))
    try:
        _both_routes(tmp_path, st, fq, float(rng.choice([0.0, 0.02, 0.05])), int(rng.integers(25, lo + 1)), f"fuzz{seed}",
                     reversals=bool(rng.integers(0, 4)), inclusions=bool(rng.integers(0, 4)))
    finally:
        if old is None:
            os.environ.pop("HC_TEXT_BLOCK", None)
        else:
            os.environ["HC_TEXT_BLOCK"] = old


def test_thousands_of_lines_for_one_pair_of_reads_go_to_the_host(tmp_path):
    """A group beyond what one lane matches (2 048 lines of ONE pair of reads: two million pairs to try) is not the device's: it says so."""
    from haploconduct_amd.records import SFO_DTYPE

    n = 2100
    recs = np.zeros(n + 1, dtype=SFO_DTYPE)
    recs["idA"][:n], recs["idB"][:n] = 0, 4          # pairs 0 and 1 (ns = 0, np = 3: SFO ids 0, 1, 2 are the /1 mates, 3, 4, 5 the /2 mates)
    recs["idA"][n], recs["idB"][n] = 0, 5            # pairs 0 and 2: the line that closes the group
    recs["OLA"] = recs["OLB"] = 40 + np.arange(n + 1) % 50
    recs["OHA"] = 1 + np.arange(n + 1) % 37
    recs["OHB"] = 2 + np.arange(n + 1) % 41
    recs["K"] = np.arange(n + 1) % 7
    reads = hc.ReadSet.from_lists([], [((b"ACGT" * 10, b"I" * 40), (b"TGCA" * 10, b"I" * 40))] * 3)
    with hc.EdgeScorer(hc.Settings()) as sc:
        sc.set_reads(reads)
        sc.set_found_records(recs)
        with pytest.raises(Exception, match="not on the device"):
            sc.found_to_lines(0, 3)
        assert sc.found_to_overlaps(str(tmp_path / "x.txt"), 0, 3) >= 0  # the host's matcher takes it


def _sfo_routes(tmp_path, st, fq, reads, n_single, n_pairs, err, t, tag, mangle=None):
    """The SFO file (what rust-overlaps writes) -> graph in one call, against scripts/sfo2overlaps.py's port + the overlaps file + the stage."""
    d = str(tmp_path) + "/"
    with hc.EdgeScorer(hc.Settings()) as sc:
        sc.set_reads(reads)
        recs = sc.find_overlaps(err, t)
    sfo = d + tag + ".sfo"
    host.write_sfo(sfo, recs)
    if mangle:
        text = mangle(open(sfo).read())
        open(sfo, "w").write(text)
    n_lines = host.sfo2overlaps(sfo, d + tag + "_overlaps.txt", n_single, n_pairs)
    out = {}
    for route in ("sfo", "file"):
        o = tmp_path / f"{tag}_{route}"
        o.mkdir()
        with host.EdgeCalculatorStage(st, output_dir=str(o) + "/", **(fq if route == "sfo" else dict(fq, overlaps=d + tag + "_overlaps.txt"))) as ec:
            if route == "sfo":
                n_rec, nl, on_device = ec.construct_edges_from_sfo(sfo)
                assert nl == n_lines and on_device == (mangle is None), (nl, n_lines, on_device)
                assert n_rec == (recs.size if mangle is None else 0)
            else:
                ec.construct_edges_sorted()
            out[route] = (ec.edges(), ec.in_lists(), ec.inclusions(), ec.counters(), (o / "nonedge_overlaps.txt").read_bytes())
    a, b = out["sfo"], out["file"]
    assert a[0].size == b[0].size and a[0].tobytes() == b[0].tobytes(), f"{tag}: the graphs differ"
    assert np.array_equal(a[1][0], b[1][0]) and np.array_equal(a[1][1], b[1][1]) and np.array_equal(a[2], b[2])
    for k in COUNTERS:
        if k in a[3]:
            assert a[3][k] == b[3][k], (tag, k)
    assert a[4] == b[4]
    return n_lines, a


def test_sfo_file_to_graph_savage_example(tmp_path):
    """hc_ec_construct_edges_from_sfo on the SAVAGE example's reads: the SFO file in, the sorted graph out — against the three-step route the
    pipeline runs (savage.py:664-717: the SFO file -> scripts/sfo2overlaps.py -> original_overlaps.txt -> the binary), each step of which is
    pinned against the reference's own code."""
    fq = dict(singles=gunzip_to("savage_singles.fastq", str(tmp_path / "singles.fastq")), paired1=gunzip_to("savage_paired1.fastq", str(tmp_path / "paired1.fastq")),
              paired2=gunzip_to("savage_paired2.fastq", str(tmp_path / "paired2.fastq")))
    f = host.Fastq(**fq)
    reads = f.readset()
    st = hc.Settings(edge_threshold=0.97, min_overlap_len=200, n_threads=8)
    n_lines, a = _sfo_routes(tmp_path, st, fq, reads, f.n_single, f.n_paired, 0.02, 100, "savage")
    assert n_lines > 20000 and a[0].size > 2000
    # a file that is not canonical (blanks for tabs: the script splits on any whitespace, but its `sort | uniq` sees other bytes — flipped lines
    # are re-joined with tabs, the others keep their blanks — so repeats survive that the tab-separated file loses): the host's ingest takes it
    n2, b = _sfo_routes(tmp_path, st, fq, reads, f.n_single, f.n_paired, 0.02, 100, "savage_blanks", mangle=lambda s: s.replace("\t", "  "))
    assert n2 >= n_lines and b[0].size > 2000


def test_sfo_file_to_graph_mixed_and_errors(tmp_path):
    kw = dict(n_single=300, n_pair=400, glen=2200, lo=100, hi=200, err=0.01)
    reads = make_reads(331, **kw)
    fq = _write(reads, tmp_path, 300, 400)
    st = hc.Settings(edge_threshold=0.9, ov_threshold=0.5, min_overlap_len=120, n_threads=8, max_overlaps=9000)
    n_lines, a = _sfo_routes(tmp_path, st, fq, reads, 300, 400, 0.03, 60, "mixed")
    assert n_lines > 2000
    # what the script raises is raised: a line with seven fields (assert len(sfo_line) == 8, scripts/sfo2overlaps.py:35)
    bad = str(tmp_path / "bad.sfo")
    open(bad, "w").write("0\t1\tN\t5\t5\t60\t60\n")
    with host.EdgeCalculatorStage(st, output_dir=str(tmp_path) + "/", **fq) as ec:
        with pytest.raises(Exception, match="8 fields"):
            ec.construct_edges_from_sfo(bad)
    with host.EdgeCalculatorStage(st, output_dir=str(tmp_path) + "/", **fq) as ec:
        with pytest.raises(Exception, match="cannot open"):
            ec.construct_edges_from_sfo(str(tmp_path / "missing.sfo"))


def test_hc_edgecalc_sfo_flag_equals_the_two_programs(tmp_path):
    """The process boundary: `hc-edgecalc --sfo sfoverlaps.out <the pipeline's other arguments>` against scripts/sfo2overlaps.py's port followed
    by `hc-edgecalc --overlaps original_overlaps.txt ...` (savage.py:664-717) on the SAVAGE example's reads: the four output files, byte for byte;
    --overlaps and --sfo together are refused."""
    import subprocess

    fq = dict(singles=gunzip_to("savage_singles.fastq", str(tmp_path / "singles.fastq")), paired1=gunzip_to("savage_paired1.fastq", str(tmp_path / "paired1.fastq")),
              paired2=gunzip_to("savage_paired2.fastq", str(tmp_path / "paired2.fastq")))
    f = host.Fastq(**fq)
    reads = f.readset()
    with hc.EdgeScorer(hc.Settings()) as sc:
        sc.set_reads(reads)
        recs = sc.find_overlaps(0.02, 100)
    d = str(tmp_path) + "/"
    host.write_sfo(d + "sfoverlaps.out", recs)
    host.sfo2overlaps(d + "sfoverlaps.out", d + "original_overlaps.txt", f.n_single, f.n_paired)
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "haploconduct_amd", "csrc", "hc-edgecalc")
    common = ["--singles", fq["singles"], "--paired1", fq["paired1"], "--paired2", fq["paired2"], "--threads", "8", "--edge_threshold", "0.97",
              "--min_overlap_len", "200", "--first_it", "true", "--original_readcount", str(reads.n_reads), "--verbose", "true"]
    outs = {}
    for name, src in (("sfo", ["--sfo", d + "sfoverlaps.out"]), ("overlaps", ["--overlaps", d + "original_overlaps.txt"])):
        o = d + name + "/"
        os.mkdir(o)
        r = subprocess.run([exe] + common + src + ["--output", o], capture_output=True, text=True, timeout=600, cwd=o)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        outs[name] = {fn: open(o + fn, "rb").read() for fn in ("edges.tsv", "edges_sorted.tsv", "edgecalc_stats.txt")}
        outs[name]["nonedge_overlaps.txt"] = open(o + "nonedge_overlaps.txt", "rb").read()  # (written in the working directory, src/EdgeCalculator.cpp:566,654)
        if name == "sfo":
            assert "ingest on the device" in r.stdout
    assert outs["sfo"] == outs["overlaps"] and len(outs["sfo"]["edges_sorted.tsv"]) > 10000
    r = subprocess.run([exe] + common + ["--sfo", d + "sfoverlaps.out", "--overlaps", d + "original_overlaps.txt"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "exclusive" in r.stderr


def _sfo_text(recs):
    return "".join(f"{r['idA']}\t{r['idB']}\t{'I' if r['inverted'] else 'N'}\t{r['OHA']}\t{r['OHB']}\t{r['OLA']}\t{r['OLB']}\t{r['K']}\n" for r in recs).encode()


def _random_sfo(rng, n_seq, n_rec):
    from haploconduct_amd.records import SFO_DTYPE

    recs = np.zeros(n_rec, dtype=SFO_DTYPE)
    recs["idA"] = rng.integers(0, n_seq, n_rec)
    recs["idB"] = rng.integers(0, n_seq, n_rec)
    recs["OHA"] = rng.integers(-40, 41, n_rec)
    recs["OHB"] = rng.integers(-40, 41, n_rec)
    recs["OLA"] = rng.integers(30, 151, n_rec)
    recs["OLB"] = np.where(rng.random(n_rec) < 0.7, recs["OLA"], rng.integers(30, 151, n_rec))
    recs["K"] = rng.integers(0, 4, n_rec)
    recs["inverted"] = rng.integers(0, 2, n_rec)
    return recs


def test_the_sfo_files_text_read_on_the_device(tmp_path):
    """hc_set_found_from_sfo_text: the text rust-overlaps writes -> records on the device, one lane per line — the same overlap lines as the
    records themselves give (hc_set_found_records), with and without a last newline, over several 64 MiB chunks; and every text that is not
    canonical (the host's parse_canonical_sfo has the same rules) is left to the host's general path."""
    rng = np.random.default_rng(11)
    ns, npairs = 40, 60
    reads = hc.ReadSet.from_lists([(b"ACGT" * 10, b"I" * 40)] * ns, [((b"ACGT" * 10, b"I" * 40), (b"TGCA" * 10, b"I" * 40))] * npairs)
    recs = _random_sfo(rng, ns + 2 * npairs, 20000)
    recs["OHA"][:4] = [-2147483647, 2147483647, 0, -1]  # the edges of the accepted ranges are accepted (the ingest's own sort keys then say no: 10^7 and more)
    text = _sfo_text(recs[4:])
    with hc.EdgeScorer(hc.Settings()) as sc:
        sc.set_reads(reads)
        sc.set_found_records(recs[4:])
        want = sc.found_to_lines(ns, npairs)
        for t in (text, text[:-1]):
            assert sc.set_found_from_sfo_text(t) == recs.size - 4
            got = sc.found_to_lines(ns, npairs)
            assert got.tobytes() == want.tobytes()
        assert sc.set_found_from_sfo_text(b"") == 0 and sc.found_to_lines(ns, npairs).size == 0
        line = b"3\t7\tN\t5\t-6\t60\t61\t2\n"
        assert sc.set_found_from_sfo_text(line) == 1
        for bad in (b"03\t7\tN\t5\t-6\t60\t61\t2\n", b"3\t7\tN\t+5\t-6\t60\t61\t2\n", b"3\t\t7\tN\t5\t-6\t60\t61\t2\n", b"3\t7\tN\t5\t-6\t60\t61\t2\r\n",
                    b"3\t7\tN\t5\t-6\t60\t61\t2 \n", b"3\t7\tN\t5\t-6\t60\t61\n", b"3\t7\tX\t5\t-6\t60\t61\t2\n", b"3\t7\tN\t5\t-0\t60\t61\t2\n",
                    b"3 7 N 5 -6 60 61 2\n", line + b"\n" + line, b"3\t7\tN\t5\t-6\t60\t61\t2\t9\n", b"4294967296\t7\tN\t5\t-6\t60\t61\t2\n",
                    b"3\t7\tN\t-2147483648\t-6\t60\t61\t2\n", b"3\t-7\tN\t5\t-6\t60\t61\t2\n"):
            with pytest.raises(Exception, match="not on the device"):
                sc.set_found_from_sfo_text(line * 3 + bad + line)
    # several chunks: 3 * 10^6 lines, ~90 MB of text (many reads: the groups stay small — with few reads the lines outnumber the records
    # eight to one and the device hands the input to the host, as it should)
    ns, npairs = 2000, 3000
    reads = hc.ReadSet.from_lists([(b"ACGT" * 10, b"I" * 40)] * ns, [((b"ACGT" * 10, b"I" * 40), (b"TGCA" * 10, b"I" * 40))] * npairs)
    with hc.EdgeScorer(hc.Settings()) as sc:
        sc.set_reads(reads)
        big = _random_sfo(rng, ns + 2 * npairs, 3000000)
        cols = [big[k].astype(str) for k in ("idA", "idB")] + [np.where(big["inverted"] != 0, "I", "N")] + [big[k].astype(str) for k in ("OHA", "OHB", "OLA", "OLB", "K")]
        big_text = "\n".join("\t".join(t) for t in zip(*cols)).encode() + b"\n"
        assert len(big_text) > (64 << 20)
        sc.set_found_records(big)
        want = sc.found_to_lines(ns, npairs)
        assert sc.set_found_from_sfo_text(big_text) == big.size
        assert sc.found_to_lines(ns, npairs).tobytes() == want.tobytes() and want.size > 100000
