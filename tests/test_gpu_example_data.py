"""BASELINE config 1 style parity on REAL reads: excerpts of the reference's example data
(tests/golden/*.fastq.gz, cut by tests/golden/make_example_excerpt.py): SAVAGE's merged singles and
2x250 pairs (25 distinct quality values, N and Q0 bases) and POLYTE's 2x250 reads treated as singles
(35 distinct quality values: the 16-bit-symbol path).  Candidates come from the build's own seed finder
(rust-overlaps is not available offline); the whole stage is compared with the oracle."""
import gzip
import os

import numpy as np
import pytest

import haploconduct_amd as hc
from haploconduct_amd import candidates, host, synth
from haploconduct_amd.records import FLAG_IGNORE_INCLUSIONS, FLAG_RESOLVE_ORIENTATIONS

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def gunzip_to(name, dst):
    with gzip.open(os.path.join(HERE, "golden", name + ".gz"), "rb") as f, open(dst, "wb") as o:
        o.write(f.read())
    return dst


def compare_stage(oracle, tmp_path, st, fq, lines, tag):
    d = str(tmp_path / tag) + "/"
    os.mkdir(d)
    ov = d + "overlaps.txt"
    open(ov, "w").write("\n".join(lines) + "\n")
    f = host.Fastq(**fq)
    reads = f.readset()
    rc, g, oc = oracle.construct_edges(reads, st, ov, d + "ref_nonedge.txt")
    assert rc == 0
    with host.EdgeCalculatorStage(st, overlaps=ov, output_dir=d, **fq) as ec:
        ec.construct_edges()
        edges, inc, c = ec.edges(), ec.inclusions(), ec.counters()
    want = g.all_edges()
    assert edges.size == want.size
    for k in ("score", "mismatch_rate"):
        assert np.array_equal(edges[k].view(np.uint64), want[k].view(np.uint64)), k
    for k in ("pos1", "pos2", "pos3", "pos4", "ori1", "ori2", "ord", "read1", "read2", "v1", "v2", "perc", "len0", "len1", "len2"):
        assert np.array_equal(edges[k], want[k]), k
    assert np.array_equal(inc, g.inclusions())
    assert open(d + "nonedge_overlaps.txt", "rb").read() == open(d + "ref_nonedge.txt", "rb").read()
    for k in ("inclusion_count", "dup_count", "edges_added", "nonedges_written", "prefilter_rejected", "scored"):
        assert c[k] == getattr(oc, k), k
    return edges, c


def reference_graph(reads, lines, st):
    """The REFERENCE'S OWN compute_overlap / process_overlaps on these candidate lines (fragment probe
    oracle/_ref/libhcref_edgecalc.so, built in the build container and shipped with the repository), or None."""
    import ctypes as C
    import importlib.util

    root = os.path.dirname(HERE)
    lib_path = os.path.join(root, "oracle", "_ref", "libhcref_edgecalc.so")
    if not os.path.exists(lib_path):
        return None
    spec = importlib.util.spec_from_file_location("make_golden_ec", os.path.join(root, "tests", "golden", "make_golden_ec.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    ref = C.CDLL(lib_path)
    ref.frag_process_overlaps.restype = C.c_int
    ref.frag_process_overlaps.argtypes = [C.POINTER(mg.FragSettings), C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p,
                                          C.c_uint64, C.c_char_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p,
                                          C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.c_void_p]
    ref.frag_ec_free.argtypes = [C.c_void_p]
    settings = dict(edge_threshold=st.edge_threshold, ov_threshold=st.ov_threshold, merge_contigs=st.merge_contigs, mismatch=st.mismatch,
                    min_read_len=st.min_read_len, ignore_inclusions=1 if st.flags & FLAG_IGNORE_INCLUSIONS else 0)
    edges, incl, nonedge, counters = mg.run_probe(ref, reads, lines, settings)
    names = ["score", "mismatch_rate", "pos1", "pos2", "pos3", "pos4", "ori1", "ori2", "ord", "v1", "v2", "perc", "len0", "len1", "len2"]
    want = {k: [e[i] for e in edges] for i, k in enumerate(names)}
    for k in ("score", "mismatch_rate"):
        want[k] = np.array([float.fromhex(x) for x in want[k]], np.float64)
    return want, incl, nonedge, counters


def test_savage_example_whole(oracle, tmp_path):
    """BASELINE config 1: the WHOLE savage/example/input_fas read set (2 000 merged singles + 200 pairs), candidates from the
    library's own overlap finder and SFO ingest (rust-overlaps + sfo2overlaps.py in the pipelines, savage.py:643-716),
    SAVAGE's stage a and stage b/c settings: the HIP stage against the oracle and — every candidate line — against the
    reference's own process_overlaps."""
    from tests.test_ec_golden import compare_edges

    s = gunzip_to("savage_singles.fastq", str(tmp_path / "singles.fastq"))
    p1 = gunzip_to("savage_paired1.fastq", str(tmp_path / "paired1.fastq"))
    p2 = gunzip_to("savage_paired2.fastq", str(tmp_path / "paired2.fastq"))
    fq = dict(singles=s, paired1=p1, paired2=p2)
    f = host.Fastq(**fq)
    reads = f.readset()
    assert (f.n_single, f.n_paired) == (2000, 200)
    ov = str(tmp_path / "overlaps.txt")
    with hc.EdgeScorer(hc.Settings()) as sc:
        sc.set_reads(reads)
        assert sc.info()["qual_alphabet"] == 25  # 8-bit symbols, 32x32 table planes
        sfo = sc.find_overlaps(0.02, 100)         # what `rust-overlaps -i -r s_p1_p2.fasta out 0.02 100` reports
    n_lines = host.sfo_records_to_overlaps(sfo, ov, f.n_single, f.n_paired)
    lines = open(ov).read().splitlines()
    assert n_lines == len(lines) > 20000
    types = {(ln.split("\t")[11], ln.split("\t")[12]) for ln in lines}
    assert ("s", "s") in types and ("p", "p") in types
    # savage stage a: --edge_threshold 0.97, M = 200 (savage/README.md:303-309, savage.py:384)
    st = hc.Settings(edge_threshold=0.97, min_overlap_len=200, n_threads=8)
    edges, c = compare_stage(oracle, tmp_path, st, fq, lines, "a")
    assert edges.size > 2000 and c["nonedges_written"] > 0 and c["prefilter_rejected"] > 0
    # stage b/c style: threshold 0.995, ignore inclusions, merge_contigs
    st = hc.Settings(edge_threshold=0.995, min_overlap_len=100, merge_contigs=0.01, n_threads=8,
                     flags=FLAG_RESOLVE_ORIENTATIONS | FLAG_IGNORE_INCLUSIONS)
    compare_stage(oracle, tmp_path, st, fq, lines, "bc")
    # every line through the reference's own code (no prefilter in front of process_overlaps: M = 0)
    for tag, st in (("ref_a", hc.Settings(edge_threshold=0.97, min_overlap_len=0, n_threads=8)),
                    ("ref_bc", hc.Settings(edge_threshold=0.995, min_overlap_len=0, merge_contigs=0.01, n_threads=8,
                                           flags=FLAG_RESOLVE_ORIENTATIONS | FLAG_IGNORE_INCLUSIONS))):
        got = reference_graph(reads, lines, st)
        if got is None:
            pytest.skip("oracle/_ref/libhcref_edgecalc.so is built only where /root/reference exists")
        want, incl, nonedge, counters = got
        d = str(tmp_path / tag) + "/"
        os.mkdir(d)
        with host.EdgeCalculatorStage(st, overlaps=ov, output_dir=d, **fq) as ec:
            ec.construct_edges()
            e2, inc2, c2 = ec.edges(), ec.inclusions(), ec.counters()
        compare_edges(e2, want, "HIP stage vs the reference's own code, savage example")
        assert inc2.tolist() == incl and open(d + "nonedge_overlaps.txt").read() == nonedge
        assert c2["inclusion_count"] == counters[0] and c2["dup_count"] == counters[1] and c2["scored"] == len(lines)


def test_polyte_example_excerpt_all_reads_as_singles(oracle, tmp_path):
    # POLYTE feeds every read as a single (polyte.py:283-288); write /1 and /2 into one singles file
    fwd = gunzip_to("polyte_forward.fastq", str(tmp_path / "f.fastq"))
    rev = gunzip_to("polyte_reverse.fastq", str(tmp_path / "r.fastq"))
    recs = []
    for path in (fwd, rev):
        L = open(path).read().split("\n")
        for i in range(0, len(L) - 3, 4):
            recs.append((L[i + 1], L[i + 3]))
    s = str(tmp_path / "singles.fastq")
    with open(s, "w") as o:
        for i, (seq, q) in enumerate(recs):
            o.write(f"@{i}\n{seq}\n+\n{q}\n")
    fq = dict(singles=s)
    f = host.Fastq(**fq)
    reads = f.readset()
    assert f.n_single == 1000
    cs = candidates.single_candidates(reads, list(range(1000)), k=20, min_overlap=80)
    assert cs.size > 500
    lines = synth.records_to_lines(cs, reads)
    with hc.EdgeScorer(hc.Settings()) as sc:
        sc.set_reads(reads)
        assert sc.info()["qual_alphabet"] > 30  # 16-bit symbols
    st = hc.Settings(edge_threshold=0.95, min_overlap_len=127)  # first POLYTE iteration (polyte.py:598-602)
    edges, _ = compare_stage(oracle, tmp_path, st, fq, lines, "it1")
    assert edges.size > 50
    st = hc.Settings(edge_threshold=1.0, merge_contigs=0.0, min_overlap_len=127)  # later iterations (polyte.py:617-626)
    compare_stage(oracle, tmp_path, st, fq, lines, "it2")
