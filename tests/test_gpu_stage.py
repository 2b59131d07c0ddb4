"""The whole edge-calculation stage on the GPU box: EdgeCalculator::construct_edges of the host
mirror (FASTQ + overlaps text file in; populated OverlapGraph, nonedge_overlaps.txt and counters
out) against the oracle's hco_construct_edges on the same files.  Edge set compared as the
flattened adjacency lists in list order (stronger than a set compare); doubles as bit patterns."""
import os
import random

import numpy as np
import pytest

import haploconduct_amd as hc
from haploconduct_amd import host, synth
from haploconduct_amd.records import FLAG_IGNORE_INCLUSIONS, FLAG_RELAX_PE_EDGES, FLAG_RESOLVE_ORIENTATIONS

pytestmark = pytest.mark.gpu

HQ = np.array([20, 30, 37, 37, 37, 40, 40, 40, 40], dtype=np.uint8) + 33
FIELDS = ("score", "mismatch_rate", "pos1", "pos2", "pos3", "pos4", "ori1", "ori2", "ord", "read1", "read2", "v1", "v2",
          "perc", "len0", "len1", "len2")


def run_both(oracle, tmp_path, reads, lines, st, tag):
    d = tmp_path / tag
    d.mkdir()
    ov = str(d / "overlaps.txt")
    with open(ov, "w") as f:
        f.write("\n".join(lines) + "\n")
    s = str(d / "singles.fastq") if any(not reads.is_paired(r) for r in range(reads.n_reads)) else None
    p1 = str(d / "paired1.fastq") if any(reads.is_paired(r) for r in range(reads.n_reads)) else None
    p2 = str(d / "paired2.fastq") if p1 else None
    reads.write_fastq(s, p1, p2)
    ref_nonedge = str(d / "ref_nonedge.txt")
    rc, g, oc = oracle.construct_edges(reads, st, ov, ref_nonedge)
    assert rc == 0
    out_dir = str(d / "out") + "/"
    os.mkdir(out_dir)
    with host.EdgeCalculatorStage(st, singles=s, paired1=p1, paired2=p2, overlaps=ov, output_dir=out_dir) as ec:
        ec.construct_edges()
        edges, inc, c = ec.edges(), ec.inclusions(), ec.counters()
        assert ec.edge_count() == edges.size
        ec.sort_edges()
        sorted_edges, sorted_in = ec.edges(), ec.in_lists()
    want = g.all_edges()
    assert edges.size == want.size, (edges.size, want.size)
    for k in FIELDS:
        a, b = edges[k], want[k]
        if a.dtype.kind == "f":
            a, b = a.view(np.uint64), b.view(np.uint64)
        assert np.array_equal(a, b), f"edge field {k} differs"
    assert np.array_equal(inc, g.inclusions())
    nonedge = open(out_dir + "nonedge_overlaps.txt", "rb").read()
    assert nonedge == open(ref_nonedge, "rb").read()
    COUNTERS = ("inclusion_count", "dup_count", "edges_added", "nonedges_written", "prefilter_rejected", "malformed_lines", "lines_read", "scored")
    for k in COUNTERS:
        assert c[k] == getattr(oc, k), k
    assert c["self_overlap_count"] == 0
    # The other routes to the same graph: above, the device read the file's text itself and resolved the duplicates;
    # the host threads' resolution, the per-edge serial insert, three contexts taking the blocks in turn, the host's
    # tokeniser instead of the device's (HC_PARSE=host), tiny text blocks, the text read in place from the file's mapping
    # (HC_TEXT_SOURCE=map), the host counting the lines itself (HC_TEXT_SOURCE=pread), write-combined text buffers, the fused construct + sortEdges call
    # (against construct_edges followed by sortEdges), and the resolved graph fetched in pieces with the host adopting behind the copy
    # (HC_FETCH_PIECE_BYTES: pieces of 50 and of 10 edges, and none) must all agree with it.
    for env, sorted_call in (({"HC_RESOLVE": "host"}, False), ({"HC_INSERT_MODE": "serial"}, False), ({"HC_DEVICE_LIST": "0,0,0"}, False),
                             ({"HC_PARSE": "host"}, False), ({"HC_PARSE": "host", "HC_DEVICE_LIST": "0,0,0"}, False),
                             ({"HC_TEXT_BLOCK": "4096"}, False), ({"HC_TEXT_BLOCK": "4096", "HC_TEXT_DEPTH": "1"}, False),
                             ({"HC_TEXT_BLOCK": "8192", "HC_TEXT_DEPTH": "3", "HC_COLLECTORS": "2"}, False), ({"HC_TEXT_SOURCE": "map"}, False), ({"HC_TEXT_SOURCE": "pread"}, False), ({"HC_TEXT_BUFFER": "wc"}, False),
                             ({"HC_TEXT_SOURCE": "pread", "HC_TEXT_BLOCK": "4096", "HC_DEVICE_LIST": "0,0,0"}, False),
                             ({"HC_TEXT_SOURCE": "map", "HC_TEXT_BLOCK": "4096", "HC_DEVICE_LIST": "0,0,0"}, False),
                             ({"HC_TEXT_BLOCK": "4096", "HC_DEVICE_LIST": "0,0,0"}, False), ({}, True), ({"HC_RESOLVE": "host"}, True),
                             ({"HC_FETCH_PIECE_BYTES": "4000"}, False), ({"HC_FETCH_PIECE_BYTES": "800"}, True), ({"HC_FETCH_PIECE_BYTES": "0"}, True)):
        os.environ.update(env)
        try:
            with host.EdgeCalculatorStage(st, singles=s, paired1=p1, paired2=p2, overlaps=ov, output_dir=out_dir) as ec:
                if "HC_DEVICE_LIST" in env:
                    assert ec.device_count() == 3
                os.remove(out_dir + "nonedge_overlaps.txt")
                if sorted_call:
                    ec.construct_edges_sorted()
                    e2, in2 = ec.edges(), ec.in_lists()
                    assert e2.tobytes() == sorted_edges.tobytes(), (env, "sorted adjacency differs")
                    assert np.array_equal(in2[0], sorted_in[0]) and np.array_equal(in2[1], sorted_in[1]), (env, "sorted in-lists differ")
                else:
                    ec.construct_edges()
                    assert ec.edges().tobytes() == edges.tobytes(), (env, "adjacency differs")
                assert np.array_equal(ec.inclusions(), inc), env
                c2 = ec.counters()
                for k in COUNTERS:
                    assert c2[k] == c[k], (env, k)
                assert open(out_dir + "nonedge_overlaps.txt", "rb").read() == nonedge, env
        finally:
            for k in env:
                os.environ.pop(k, None)
    return edges, c


def test_stage_paired_with_duplicates_and_junk_lines(oracle, tmp_path):
    reads, meta = synth.make_paired_dataset(1500, 3000, flip_frac=0.3, seed=3)
    reads.quals[:] = HQ[np.random.default_rng(1).integers(0, HQ.size, reads.quals.size)]
    cand = synth.paired_candidates(meta, n_candidates=20000, seed=4)
    lines = synth.records_to_lines(cand, reads)
    rng = random.Random(2)
    # a third of the candidates once more, shuffled in: exercises the replace / keep / tie-break chain
    for ln in rng.sample(lines, len(lines) // 3):
        lines.insert(rng.randrange(len(lines)), ln)
    for junk in ("", "only\tthree\tfields", "9\t9\t0\t0\t1\t+\t+\t100\t100\t150\t150\tp\tp", "   "):
        lines.insert(rng.randrange(len(lines)), junk)
    for st in (hc.Settings(edge_threshold=0.97, min_overlap_len=150),
               hc.Settings(edge_threshold=0.97, min_overlap_len=220, min_overlap_perc=60,
                           flags=FLAG_RESOLVE_ORIENTATIONS | FLAG_RELAX_PE_EDGES),
               hc.Settings(edge_threshold=0.995, merge_contigs=0.01, min_overlap_len=150,
                           flags=FLAG_RESOLVE_ORIENTATIONS | FLAG_IGNORE_INCLUSIONS)):
        edges, c = run_both(oracle, tmp_path, reads, lines, st, f"pp{st.edge_threshold}{st.min_overlap_len}")
        assert edges.size > 100 and c["dup_count"] > 50 and c["nonedges_written"] > 0


def test_stage_singles_contigs_inclusions(oracle, tmp_path):
    reads, meta = synth.make_single_dataset(1200, 5000, len_lo=150, len_hi=900, flip_frac=0.4, seed=5, quals=HQ,
                                            log_uniform=True)
    cand = synth.single_candidates(meta, min_overlap=80, n_candidates=20000)
    lines = synth.records_to_lines(cand, reads)
    # the same overlaps reported from the other read's point of view on the opposite strand:
    # (A, B, pos, oa, ob) ~ (B, A, lenB - L, !ob, !oa) when A's suffix lies inside B.  The two sum the same
    # terms in opposite order, so their scores differ in the last bits: the tie-break chain must agree.
    lens = meta["lens"]
    ids = reads.read_ids
    twin = []
    for r in cand[::3]:
        la, lb = int(lens[r["read1"]]), int(lens[r["read2"]])
        L = la - int(r["pos1"])
        if 0 < L <= lb:
            twin.append("\t".join([str(int(ids[r["read2"]])), str(int(ids[r["read1"]])), str(lb - L), "-", "-",
                                   "-" if r["ori2"] else "+", "-" if r["ori1"] else "+", str(int(r["perc"])), "-",
                                   str(int(r["len1"])), "-", "s", "s"]))
    assert len(twin) > 1000
    rng = random.Random(8)
    for t in twin:
        lines.insert(rng.randrange(len(lines)), t)
    st = hc.Settings(edge_threshold=0.995, ov_threshold=0.9, min_overlap_len=100,
                     flags=FLAG_RESOLVE_ORIENTATIONS | FLAG_IGNORE_INCLUSIONS)
    edges, c = run_both(oracle, tmp_path, reads, lines, st, "ss")
    assert edges.size > 100 and c["inclusion_count"] > 0 and c["dup_count"] > 200
    st = hc.Settings(edge_threshold=1.0, merge_contigs=0.0, min_overlap_len=100)
    run_both(oracle, tmp_path, reads, lines, st, "ss_polyte")


def test_stage_mixed_single_and_paired_reads(oracle, tmp_path):
    from tests.test_gpu_parity import _mixed_reads

    reads, spos, ppos = _mixed_reads(13, n_single=200, n_pair=200, glen=2000)
    ns = len(spos)
    ids = reads.read_ids
    lines = []
    for i, (s, L) in enumerate(spos):
        for j, (ps, ins) in enumerate(ppos):
            p1, p2 = ps - s, ps + ins - 150 - s
            if 0 <= p1 < L - 40 and 0 <= p2 < L - 40:
                l1, l2 = min(L - p1, 150), min(L - p2, 150)
                lines.append(f"{ids[i]}\t{ids[ns + j]}\t{p1}\t{p2}\t-\t+\t+\t{100 * l1 // 150}\t{100 * l2 // 150}\t{l1}\t{l2}\ts\tp")
            q1, q2 = s - ps, ps + ins - 150 - s
            if 0 <= q1 < 110 and 0 <= q2 < L - 40:
                l1, l2 = min(150 - q1, L), min(L - q2, 150)
                lines.append(f"{ids[ns + j]}\t{ids[i]}\t{q1}\t{q2}\t-\t+\t+\t{100 * l1 // 150}\t{100 * l2 // 150}\t{l1}\t{l2}\tp\ts")
    assert len(lines) > 300
    st = hc.Settings(edge_threshold=0.97, ov_threshold=0.5, min_overlap_len=100)
    edges, _ = run_both(oracle, tmp_path, reads, lines, st, "mixed")
    assert edges.size > 20


def test_stage_overlap_score_entry_point(oracle, tmp_path):
    reads = hc.ReadSet.from_lists([("ACGTACGTAC", "IIIIIIIIII"), ("ACGTACGTAC", "IIIIIIIIII")])
    reads.write_fastq(str(tmp_path / "s.fastq"))
    ov = str(tmp_path / "ov.txt")
    open(ov, "w").write("")
    with host.EdgeCalculatorStage(hc.Settings(), singles=str(tmp_path / "s.fastq"), overlaps=ov,
                                  output_dir=str(tmp_path) + "/") as ec:
        for a, b, qa, qb, pos in ((b"ACGTTGCAAGGCTA", b"TGCAAGGCTAAC", b"IIIII55555IIII", b"II5555IIII!I", 4),
                                  (b"ACGTNNNN", b"ACGAACGT", b"IIIIIIII", b"55555555", 0), (b"ACGT", b"ACGT", b"IIII", b"IIII", 9)):
            sc, mr = ec.overlap_score(a, b, qa, qb, pos)
            want = oracle.overlap_score(a, b, qa, qb, pos)
            assert sc.hex() == want["score"].hex() and mr.hex() == want["mismatch_rate"].hex()


def test_stage_errors(tmp_path):
    reads = hc.ReadSet.from_lists([("ACGTACGTAC", "IIIIIIIIII"), ("ACGTACGTAC", "IIIIIIIIII")])
    reads.write_fastq(str(tmp_path / "s.fastq"))
    bad = str(tmp_path / "bad.txt")
    open(bad, "w").write("0\t1\t0\t-\t-\t*\t+\t100\t-\t10\t-\ts\ts\n")  # ori '*': the reference exits
    with host.EdgeCalculatorStage(hc.Settings(min_overlap_len=5), singles=str(tmp_path / "s.fastq"), overlaps=bad,
                                  output_dir=str(tmp_path) + "/") as ec:
        with pytest.raises(hc.HcError):
            ec.construct_edges()
    unknown = str(tmp_path / "unknown.txt")
    open(unknown, "w").write("0\t7\t0\t-\t-\t+\t+\t100\t-\t10\t-\ts\ts\n")  # read 7 does not exist: map::at throws
    with host.EdgeCalculatorStage(hc.Settings(min_overlap_len=5), singles=str(tmp_path / "s.fastq"), overlaps=unknown,
                                  output_dir=str(tmp_path) + "/") as ec:
        with pytest.raises(hc.HcError):
            ec.construct_edges()
    with host.EdgeCalculatorStage(hc.Settings(), singles=str(tmp_path / "s.fastq"), overlaps=str(tmp_path / "nope.txt"),
                                  output_dir=str(tmp_path) + "/") as ec:
        with pytest.raises(hc.HcError):
            ec.construct_edges()  # "Unable to open overlaps file"


def test_cli_accepts_reference_argv_and_matches_oracle(oracle, tmp_path):
    """hc-edgecalc driven with the argv scripts/pipeline_per_stage.py:381-407 builds for ViralQuasispecies."""
    import subprocess

    reads, meta = synth.make_paired_dataset(800, 2000, flip_frac=0.2, seed=21)
    reads.quals[:] = HQ[np.random.default_rng(3).integers(0, HQ.size, reads.quals.size)]
    cand = synth.paired_candidates(meta, n_candidates=8000, seed=22)
    d = str(tmp_path) + "/"
    open(d + "overlaps.txt", "w").write("\n".join(synth.records_to_lines(cand, reads)) + "\n")
    reads.write_fastq(None, d + "paired1.fastq", d + "paired2.fastq")
    exe = os.path.join(os.path.dirname(hc.lib_path), "hc-edgecalc")
    argv = [exe, "--singles=None", "--paired1=" + d + "paired1.fastq", "--paired2=" + d + "paired2.fastq",
            "--overlaps=" + d + "overlaps.txt", "--threads=4", "--edge_threshold=0.970000", "--first_it=true",
            "--cliques=true", "--error_correction=true", "--keep_singletons=0", "--min_clique_size=4", "--min_overlap_perc=0",
            "--min_overlap_len=150", "--merge_contigs=0.000000", "--FNO=3", "--original_readcount=800", "--remove_trans=2",
            "--optimize=false", "--base_path=/nowhere", "--min_qual=0.9", "--verbose=true", "--output=" + d]
    r = subprocess.run(argv, cwd=d, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "edges have been constructed in" in r.stdout and "Number of inclusion edges:" in r.stdout
    st = hc.Settings(edge_threshold=0.97, min_overlap_len=150)
    rc, g, oc = oracle.construct_edges(reads, st, d + "overlaps.txt", d + "ref_nonedge.txt")
    want = g.all_edges()
    rows = [ln.split("\t") for ln in open(d + "edges.tsv").read().splitlines()]
    assert len(rows) == want.size
    for row, e in zip(rows, want):
        assert (int(row[0]), int(row[1]), int(row[4]), int(row[5]), int(row[6]), int(row[7])) == \
               (e["v1"], e["v2"], e["pos1"], e["pos2"], e["pos3"], e["pos4"])
        assert float(row[15]) == e["score"] and float(row[16]) == e["mismatch_rate"]
    assert open(d + "nonedge_overlaps.txt", "rb").read() == open(d + "ref_nonedge.txt", "rb").read()
    # edges_sorted.tsv: the same edges after sortEdges — per out-list non-decreasing (non-overlap length, vertex2)
    srows = [ln.split("\t") for ln in open(d + "edges_sorted.tsv").read().splitlines()]
    assert sorted(map(tuple, srows)) == sorted(map(tuple, rows))
    keys = [(int(r[0]), 600 - 2 * int(r[12]), int(r[1])) for r in srows]  # every read is 2 x 150 bp
    assert keys == sorted(keys) and srows != rows
    stats = dict(ln.split("\t") for ln in open(d + "edgecalc_stats.txt").read().splitlines())
    assert int(stats["dup_count"]) == oc.dup_count and int(stats["inclusion_count"]) == oc.inclusion_count


def test_inter_iteration_path_edge_calc_fno_edge_calc(oracle, tmp_path):
    """BASELINE config 4's shape: edge calculation -> find-next-overlaps -> edge calculation.  With no super-read
    merged in between, FNO=1 copies every edge and every stored non-edge into the next overlaps file (new id = old
    id), so the second edge calculation must rebuild the same graph, bit for bit, from FNO's text."""
    from haploconduct_amd import fno as F

    for tag, (reads, cand) in {
        "singles": (lambda rm: (rm[0], synth.single_candidates(rm[1], min_overlap=100, n_candidates=15000)))(
            synth.make_single_dataset(1500, 6000, len_lo=250, len_hi=250, flip_frac=0.0, seed=21, quals=HQ)),
        "pairs": (lambda rm: (rm[0], synth.paired_candidates(rm[1], n_candidates=15000, seed=23)))(
            synth.make_paired_dataset(1200, 2500, flip_frac=0.0, seed=22)),
    }.items():
        if tag == "pairs":
            reads.quals[:] = HQ[np.random.default_rng(5).integers(0, HQ.size, reads.quals.size)]
        st = hc.Settings(edge_threshold=0.97, ov_threshold=0.5, min_overlap_len=120 if tag == "singles" else 150)
        lines = synth.records_to_lines(cand, reads)
        edges_a, ca = run_both(oracle, tmp_path, reads, lines, st, tag + "_a")
        assert edges_a.size > 200 and ca["nonedges_written"] > 20
        d = tmp_path / (tag + "_a")
        s = str(d / "singles.fastq") if tag == "singles" else None
        p1, p2 = (None, None) if tag == "singles" else (str(d / "paired1.fastq"), str(d / "paired2.fastq"))
        # the facts FNO reads: every vertex unvisited, new id = read id, no super-reads
        n = reads.n_reads
        nodes = np.zeros(n, F.FNO_READ_DTYPE)
        nodes["id"] = reads.read_ids
        for r in range(n):
            q = int(reads.read_first_seq[r])
            nodes["len1"][r] = int(reads.seq_off[q + 1] - reads.seq_off[q])
            if reads.is_paired(r):
                nodes["len2"][r] = int(reads.seq_off[q + 2] - reads.seq_off[q + 1])
                nodes["paired"][r] = 1
        nodes["orientation"] = 1
        g = np.zeros(edges_a.size, F.FNO_EDGE_DTYPE)
        for k in ("v1", "v2", "score", "pos1", "pos2", "len1", "len2", "perc", "ord", "ori1", "ori2"):
            g[k] = edges_a[k]
        fq = host.Fastq(singles=s, paired1=p1, paired2=p2)
        recs, _ = fq.parse_file(hc.Settings(edge_threshold=0.97, min_overlap_len=0), str(d / "out" / "nonedge_overlaps.txt"))
        fq.close()
        # the file holds the scored non-edges and, at its end, the lines that failed the length/type prefilter (:654-660)
        assert recs.size == sum(1 for _ in open(d / "out" / "nonedge_overlaps.txt")) >= ca["nonedges_written"]
        inp = F.Fno1Input(nodes, np.zeros(0, F.FNO_READ_DTYPE), [], [], g, nonedges=F.edges_from_records(recs),
                          new_read_count=int(reads.read_ids.max()) + 1)
        text, cnt = F.find_next_overlaps(inp)
        assert cnt["copied"] == cnt["n_lines"] == edges_a.size + recs.size  # nothing merged, nothing shadowed, nothing duplicated
        lines_b = text.decode().split("\n")[:-1]
        edges_b, cb = run_both(oracle, tmp_path, reads, lines_b, st, tag + "_b")
        assert cb["dup_count"] == 0 and cb["nonedges_written"] == ca["nonedges_written"] and cb["edges_added"] == edges_a.size
        assert cb["prefilter_rejected"] == ca["prefilter_rejected"]
        key = lambda e: sorted(zip(*(e[k].view(np.uint64).tolist() if e[k].dtype.kind == "f" else e[k].tolist() for k in FIELDS)))
        assert key(edges_a) == key(edges_b)


def _ec_golden_cases():
    import glob

    return sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ec", "*.json")))


@pytest.mark.parametrize("path", _ec_golden_cases(), ids=[os.path.basename(p)[:-5] for p in _ec_golden_cases()])
def test_stage_reproduces_the_references_own_process_overlaps(tmp_path, path):
    """The device path against the REFERENCE ITSELF: tests/golden/ec/*.json hold what the genuine compute_overlap /
    process_overlaps (src/EdgeCalculator.cpp:26-557, run through the fragment probe) built from these reads and
    candidate lines — adjacency lists in list order, scores and mismatch rates as bit patterns, inclusions,
    nonedge_overlaps.txt, counters.  The HIP stage must leave exactly the same behind."""
    from tests.test_ec_golden import compare_edges, load_case

    c, reads, st, want = load_case(path)
    d = tmp_path
    (d / "overlaps.txt").write_text("\n".join(c["lines"]) + "\n")
    s = str(d / "singles.fastq") if c["n_single"] else None
    p1 = str(d / "paired1.fastq") if c["n_paired"] else None
    p2 = str(d / "paired2.fastq") if c["n_paired"] else None
    reads.write_fastq(s, p1, p2)
    out = d / "out"
    out.mkdir()
    st.n_threads = 4
    with host.EdgeCalculatorStage(st, singles=s, paired1=p1, paired2=p2, overlaps=str(d / "overlaps.txt"), output_dir=str(out) + "/") as ec:
        ec.construct_edges()
        edges, inc, cnt = ec.edges(), ec.inclusions(), ec.counters()
    compare_edges(edges, want, "HIP stage")
    assert inc.tolist() == c["inclusions"]
    assert (out / "nonedge_overlaps.txt").read_text() == c["nonedge_overlaps"]
    assert cnt["inclusion_count"] == c["inclusion_count"] and cnt["dup_count"] == c["dup_count"]


@pytest.mark.parametrize("seed", range(int(os.environ.get("HC_FUZZ_STAGE_SEEDS", "6"))))
def test_fuzz_stage_against_oracle(oracle, tmp_path, seed):
    """The whole stage on random inputs: read types, duplicates and strand twins shuffled in, junk lines, spaces,
    random prefilter and scoring settings, random block size and thread count — graph, inclusions, non-edge file and
    counters must equal the oracle's construct_edges."""
    rng = np.random.default_rng(7000 + seed)
    pyrng = random.Random(seed)
    mode = seed % 3
    if mode == 0:
        reads, meta = synth.make_paired_dataset(int(rng.integers(200, 800)), int(rng.integers(600, 2500)), flip_frac=float(rng.choice([0, 0.3])),
                                                seed=100 + seed)
        reads.quals[:] = HQ[rng.integers(0, HQ.size, reads.quals.size)]
        cand = synth.paired_candidates(meta, n_candidates=None, seed=seed)[: int(rng.integers(2000, 9000))]
    elif mode == 1:
        reads, meta = synth.make_single_dataset(int(rng.integers(200, 800)), int(rng.integers(1500, 6000)), len_lo=120, len_hi=int(rng.integers(200, 900)),
                                                flip_frac=float(rng.choice([0, 0.4])), seed=100 + seed, quals=HQ, log_uniform=bool(rng.integers(0, 2)))
        cand = synth.single_candidates(meta, min_overlap=int(rng.integers(40, 100)), n_candidates=None)[: int(rng.integers(2000, 9000))]
    else:
        from tests.test_gpu_parity import _mixed_reads
        from haploconduct_amd.records import OVERLAP_DTYPE

        reads, spos, ppos = _mixed_reads(300 + seed, n_single=120, n_pair=120, glen=1500)
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
    lines = synth.records_to_lines(cand, reads)
    for ln in pyrng.sample(lines, len(lines) // 4):  # duplicates: replace / keep / tie-break chain
        lines.insert(pyrng.randrange(len(lines)), ln)
    for junk in ("", "a\tb", "   ", "1\t2\t3", "\t".join(["7"] * 14)):
        lines.insert(pyrng.randrange(len(lines)), junk)
    allow_spaces = bool(rng.integers(0, 2))
    if allow_spaces:  # --allow_spaced_overlaps: spaces separate fields too
        for k in pyrng.sample(range(len(lines)), len(lines) // 10):
            if lines[k].count("\t") == 12:
                lines[k] = "  " + lines[k].replace("\t", " \t", 3) + " "
    from haploconduct_amd.records import FLAG_ALLOW_SPACES

    from haploconduct_amd.records import FLAG_ADD_DUPLICATES

    # every fourth scenario with --add_duplicates (vertices by orientation, addEquivalentEdges at the end) instead of its
    # exclusive twin --resolve_orientations (src/ViralQuasispecies.cpp:144-148)
    flags = (FLAG_ADD_DUPLICATES if seed % 4 == 3 else FLAG_RESOLVE_ORIENTATIONS) | (FLAG_IGNORE_INCLUSIONS if rng.integers(0, 2) else 0) | \
        (FLAG_RELAX_PE_EDGES if rng.integers(0, 2) else 0) | (FLAG_ALLOW_SPACES if allow_spaces else 0)
    st = hc.Settings(edge_threshold=float(rng.choice([0.9, 0.97, 0.995, 1.0])), ov_threshold=float(rng.choice([0.0, 0.5, 0.9])),
                     merge_contigs=float(rng.choice([0.0, 0.0, 0.01])), mismatch=float(rng.choice([0.0, 0.0, 0.02])),
                     min_read_len=int(rng.choice([0, 0, 100])), min_overlap_len=int(rng.choice([0, 100, 150, 220])),
                     min_overlap_perc=int(rng.choice([0, 0, 60])), flags=flags, max_overlaps=int(rng.choice([10 ** 8, 10 ** 8, len(lines) // 2])))
    st.n_threads = int(rng.choice([1, 3, 8]))
    os.environ["HC_STAGE_BLOCK"] = str(int(rng.choice([700, 5000, 250000])))
    os.environ["HC_TEXT_BLOCK"] = str(int(rng.choice([4096, 60000, 16 << 20])))
    try:
        edges, c = run_both(oracle, tmp_path, reads, lines, st, f"fz{seed}")
        if os.environ.get("HC_FUZZ_REPORT"):  # coverage of the soak: what the scenarios actually exercised
            with open(os.environ["HC_FUZZ_REPORT"], "a") as f:
                f.write(f"{seed} mode={mode} lines={len(lines)} edges={edges.size} " + " ".join(f"{k}={c[k]}" for k in (
                    "scored", "dup_count", "inclusion_count", "nonedges_written", "prefilter_rejected", "malformed_lines")) + "\n")
    finally:
        os.environ.pop("HC_STAGE_BLOCK", None)
        os.environ.pop("HC_TEXT_BLOCK", None)


def test_edge_invariant_violation_in_a_threaded_build_is_an_error_not_a_crash(tmp_path):
    """Edge::set_len asserts len1 > 0 (src/Edge.h:211-218).  With enough admitted edges the Edge objects are built on
    pool threads: the violation must come back as an error of construct_edges."""
    reads, meta = synth.make_paired_dataset(3000, 4000, seed=21)
    reads.quals[:] = ord("I")
    cand = synth.paired_candidates(meta, n_candidates=None, seed=3)[:60000]
    cand = cand.copy()
    cand["len1"] = 0
    lines = synth.records_to_lines(cand, reads)
    ov = str(tmp_path / "overlaps.txt")
    with open(ov, "w") as f:
        f.write("\n".join(lines) + "\n")
    p1, p2 = str(tmp_path / "p1.fastq"), str(tmp_path / "p2.fastq")
    reads.write_fastq(None, p1, p2)
    st = hc.Settings(edge_threshold=0.5, ov_threshold=0.0, min_overlap_len=0, flags=FLAG_RESOLVE_ORIENTATIONS)
    st.n_threads = 8
    os.mkdir(str(tmp_path / "out"))
    with host.EdgeCalculatorStage(st, paired1=p1, paired2=p2, overlaps=ov, output_dir=str(tmp_path / "out") + "/") as ec:
        with pytest.raises(hc.HcError):
            ec.construct_edges()


@pytest.mark.parametrize("kind", ["singles", "pairs"])
def test_sort_edges_after_construct_edges(tmp_path, kind):
    """construct_edges then sortEdges — the first two calls of every assembly iteration (src/ViralQuasispecies.cpp:281,297)
    — against the sortEdges oracle (itself pinned to the reference's own code, tests/test_sort_edges.py)."""
    import ctypes as C

    if kind == "singles":
        reads, meta = synth.make_single_dataset(6000, 9000, len_lo=120, len_hi=700, flip_frac=0.4, seed=41, quals=HQ, log_uniform=True)
        cand = synth.single_candidates(meta, min_overlap=60, n_candidates=None)[:120000]
        paths = dict(singles=str(tmp_path / "s.fastq"))
        reads.write_fastq(paths["singles"], None, None)
    else:
        reads, meta = synth.make_paired_dataset(4000, 5000, flip_frac=0.3, seed=42)
        reads.quals[:] = HQ[np.random.default_rng(1).integers(0, HQ.size, reads.quals.size)]
        cand = synth.paired_candidates(meta, n_candidates=None, seed=4)[:120000]
        paths = dict(paired1=str(tmp_path / "p1.fastq"), paired2=str(tmp_path / "p2.fastq"))
        reads.write_fastq(None, paths["paired1"], paths["paired2"])
    ov = str(tmp_path / "overlaps.txt")
    host.write_overlaps(ov, cand, reads)
    st = hc.Settings(edge_threshold=0.9, ov_threshold=0.5, min_overlap_len=0, flags=FLAG_RESOLVE_ORIENTATIONS)
    st.n_threads = 8
    os.mkdir(str(tmp_path / "out"))
    lib = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "libgraphoracle.so"))
    lib.hco_sort_edges.restype = C.c_int
    lib.hco_sort_edges.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    with host.EdgeCalculatorStage(st, overlaps=ov, output_dir=str(tmp_path / "out") + "/", **paths) as ec:
        ec.construct_edges()
        before = ec.edges()
        assert before.size > (1 << 14)  # the threaded path
        V = ec.read_count()
        rfs = np.asarray(reads.read_first_seq)
        seq_len = np.diff(np.asarray(reads.seq_off)).astype(np.uint32)
        total = np.add.reduceat(seq_len, rfs[:-1].astype(np.int64)).astype(np.uint32)
        want = np.zeros(before.size, dtype=host.EDGE_DTYPE)
        woff = np.zeros(V + 1, np.uint64)
        wnodes = np.zeros(before.size, np.uint64)
        assert lib.hco_sort_edges(before.ctypes.data, before.size, V, total.ctypes.data, want.ctypes.data, woff.ctypes.data, wnodes.ctypes.data) == 0
        ec.sort_edges()
        after = ec.edges()
        off, nodes = ec.in_lists()
    assert after.tobytes() == want.tobytes()
    assert not np.array_equal(after["v2"], before["v2"])  # it did reorder something
    assert np.array_equal(off, woff) and np.array_equal(nodes, wnodes)


def test_unordered_overlaps_file_over_many_blocks(oracle, tmp_path):
    """An overlaps file in no particular order (duplicates shuffled in), long enough for many pipeline blocks of
    different line counts: every block takes the device re-ordering before the kernel and the compaction after it."""
    reads, meta = synth.make_paired_dataset(12000, 16000, flip_frac=0.3, seed=61)
    reads.quals[:] = HQ[np.random.default_rng(2).integers(0, HQ.size, reads.quals.size)]
    cand = synth.paired_candidates(meta, n_candidates=None, seed=7)[:260000]
    rng = np.random.default_rng(5)
    cand = np.concatenate([cand, cand[rng.integers(0, cand.size, 40000)]])
    cand = cand[rng.permutation(cand.size)]
    lines = synth.records_to_lines(cand, reads)
    st = hc.Settings(edge_threshold=0.97, ov_threshold=0.9, min_overlap_len=0, flags=FLAG_RESOLVE_ORIENTATIONS)
    st.n_threads = 8
    os.environ["HC_STAGE_BLOCK"] = "37000"
    try:
        edges, c = run_both(oracle, tmp_path, reads, lines, st, "unordered")
    finally:
        os.environ.pop("HC_STAGE_BLOCK", None)
    assert c["dup_count"] > 1000 and edges.size > 5000


def test_prefilter_rejecting_half_of_the_lines_over_many_blocks(oracle, tmp_path):
    """Every block loses lines to the prefilter and to the line limit: the parser closes the gaps inside each block
    (through the workers' scratch), the rejected lines end up in nonedge_overlaps.txt after the scored non-edges."""
    reads, meta = synth.make_paired_dataset(9000, 12000, flip_frac=0.2, seed=63)
    reads.quals[:] = HQ[np.random.default_rng(4).integers(0, HQ.size, reads.quals.size)]
    cand = synth.paired_candidates(meta, n_candidates=None, seed=9)[:200000]
    lines = synth.records_to_lines(cand, reads)
    for junk_at in range(0, len(lines), 997):
        lines[junk_at] = "not\tan\toverlap"
    st = hc.Settings(edge_threshold=0.97, ov_threshold=0.5, min_overlap_len=230, min_overlap_perc=0, flags=FLAG_RESOLVE_ORIENTATIONS,
                     max_overlaps=len(lines) - 12345)
    st.n_threads = 16
    os.environ["HC_STAGE_BLOCK"] = "21000"
    try:
        edges, c = run_both(oracle, tmp_path, reads, lines, st, "prefilter")
    finally:
        os.environ.pop("HC_STAGE_BLOCK", None)
    assert c["prefilter_rejected"] > 50000 and c["scored"] > 30000 and c["malformed_lines"] > 100 and edges.size > 1000


def test_stage_device_mask(tmp_path):
    """hc_settings.device_mask: the stage deals its blocks to every device of the mask in turn (replicated read store, one
    context per device, results consumed in block order).  With one GPU visible the mask names it alone; with two or more,
    the first two: the reference's own process_overlaps vectors must come out either way."""
    from tests.test_ec_golden import compare_edges, load_case

    n_dev = hc.device_count()
    path = [p for p in _ec_golden_cases() if "pairs_dups" in p][0]
    c, reads, st, want = load_case(path)
    (tmp_path / "overlaps.txt").write_text("\n".join(c["lines"]) + "\n")
    reads.write_fastq(None, str(tmp_path / "p1.fastq"), str(tmp_path / "p2.fastq"))
    st.n_threads = 4
    os.environ["HC_TEXT_BLOCK"] = "8192"  # many blocks, so that every device gets some
    try:
        for mask in ([0b1] if n_dev < 2 else [0b1, 0b11, 0b10]):
            st.device_mask = mask
            out = tmp_path / f"out{mask}"
            out.mkdir()
            with host.EdgeCalculatorStage(st, paired1=str(tmp_path / "p1.fastq"), paired2=str(tmp_path / "p2.fastq"),
                                          overlaps=str(tmp_path / "overlaps.txt"), output_dir=str(out) + "/") as ec:
                assert ec.device_count() == bin(mask).count("1")
                ec.construct_edges()
                compare_edges(ec.edges(), want, f"HIP stage, device mask {mask:#b}")
                assert ec.inclusions().tolist() == c["inclusions"]
            assert (out / "nonedge_overlaps.txt").read_text() == c["nonedge_overlaps"]
        if n_dev < 2:
            st.device_mask = 0b10  # a device that is not there: an error, not a silent fallback
            with pytest.raises(hc.HcError):
                host.EdgeCalculatorStage(st, paired1=str(tmp_path / "p1.fastq"), paired2=str(tmp_path / "p2.fastq"),
                                         overlaps=str(tmp_path / "overlaps.txt"), output_dir=str(tmp_path) + "/")
    finally:
        os.environ.pop("HC_TEXT_BLOCK", None)


@pytest.mark.parametrize("what", ["survivors", "rejects"])
def test_blocks_where_most_lines_survive_stay_on_the_device(oracle, tmp_path, monkeypatch, what):
    """A text block starts with row buffers for an eighth of its lines (real files keep a few per cent).  Error-free reads:
    EVERY overlap is admitted (survivors), or a prefilter that rejects most lines into nonedge_overlaps.txt (rejects): the
    blocks must grow their buffers and run their device half again, not fall back to the host's tokeniser
    (hc_ec_counters.host_blocks == 0), and leave the oracle's graph, non-edge file and counters."""
    monkeypatch.setenv("HC_TEXT_BLOCK", str(1 << 20))
    reads, meta = synth.make_paired_dataset(2500, 1500, n_strains=1, divergence=0.0, err=0.0, n_rate=0.0, seed=77)
    cand = synth.paired_candidates(meta, n_candidates=120000, seed=78)
    lines = synth.records_to_lines(cand, reads)
    st = hc.Settings(edge_threshold=0.97, ov_threshold=0.9, min_overlap_len=150 if what == "survivors" else 260, n_threads=4)
    d = tmp_path
    ov = str(d / "overlaps.txt")
    open(ov, "w").write("\n".join(lines) + "\n")
    reads.write_fastq(None, str(d / "p1.fastq"), str(d / "p2.fastq"))
    rc, g, oc = oracle.construct_edges(reads, st, ov, str(d / "ref_nonedge.txt"))
    assert rc == 0
    out = d / "out"
    out.mkdir()
    with host.EdgeCalculatorStage(st, paired1=str(d / "p1.fastq"), paired2=str(d / "p2.fastq"), overlaps=ov, output_dir=str(out) + "/") as ec:
        ec.construct_edges()
        edges, c = ec.edges(), ec.counters()
    assert c["host_blocks"] == 0 and c["device_blocks"] >= 4, c
    assert c["regrown_blocks"] >= 1, "the test is meant to overflow the blocks' first row buffers"
    if what == "survivors":
        assert c["edges_added"] + c["dup_count"] > 0.5 * len(lines)
    else:
        assert c["prefilter_rejected"] > 0.5 * len(lines)
    want = g.all_edges()
    assert edges.size == want.size
    for k in FIELDS:
        a, b = edges[k], want[k]
        if a.dtype.kind == "f":
            a, b = a.view(np.uint64), b.view(np.uint64)
        assert np.array_equal(a, b), k
    assert (out / "nonedge_overlaps.txt").read_bytes() == (d / "ref_nonedge.txt").read_bytes()
    for k in ("inclusion_count", "dup_count", "edges_added", "nonedges_written", "prefilter_rejected", "lines_read", "scored"):
        assert c[k] == getattr(oc, k), k
