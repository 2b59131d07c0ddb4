"""Host logic of the stage (csrc/host/, through include/hcedge_host.h) against the oracle and the
golden vectors — no GPU needed: tokenizer, Overlap record parsing, FastqStorage, the parser +
prefilter of construct_edges, and the serial insert / duplicate resolution of process_overlaps."""
import json
import os
import random

import numpy as np
import pytest

import haploconduct_amd as hc
from haploconduct_amd import host, synth

HERE = os.path.dirname(os.path.abspath(__file__))


def test_tokenizer_matches_oracle(oracle):
    rng = random.Random(3)
    alphabet = ["\t", "\t", " ", "a", "12", "-", "+", "s", "p", ""]
    lines = ["", " ", "\t", "a", "a\tb", "a\t\tb", " a \t b ", "\ta\tb\t", "a b\tc", "1\t2\t3\t-\t-\t+\t+\t9\t-\t8\t-\ts\ts"]
    for _ in range(500):
        lines.append("".join(rng.choice(alphabet) for _ in range(rng.randrange(0, 30))))
    for ln in lines:
        for sp in (False, True):
            assert host.split_line(ln, sp) == oracle.split_line(ln, sp), repr(ln)


def test_overlap_parsing_matches_reference_golden_and_oracle(oracle):
    gold = json.load(open(os.path.join(HERE, "golden", "ref_headers.json")))
    for v in gold["overlap_parse"]:
        rc, o = host.parse_overlap("\t".join(v["fields"]))
        if any(f != f.strip(" ") or f == "" for f in v["fields"]):
            continue  # outer spaces are trimmed at line level before the split; covered by the fuzz below
        assert rc == 0
        for k in ("id1", "id2", "pos1", "pos2", "ord", "ori1", "ori2", "type1", "type2", "perc", "len1", "len2", "line"):
            assert o[k] == v[k], (k, v["fields"])
    rng = random.Random(5)
    pool = ["0", "7", "150", "-", "+", "1", "2", "s", "p", "-3", "101", "x", " 1", "0x1f", "", "s ", "12 ", "99999999999"]
    n_ok = n_bad = 0
    for _ in range(3000):
        f = [rng.choice(pool) for _ in range(13)]
        if rng.random() < 0.6:  # mostly well-formed
            t1, t2 = rng.choice("sp"), rng.choice("sp")
            f = [str(rng.randrange(1000)), str(rng.randrange(1000)), str(rng.randrange(300)), rng.choice(["-", "5", "0"]),
                 rng.choice("12") if t1 == t2 == "p" else "-", rng.choice("+-"), rng.choice("+-"), str(rng.randrange(101)),
                 rng.choice(["-", "0", "50"]), str(rng.randrange(500)), rng.choice(["-", "0", "80"]), t1, t2]
            if rng.random() < 0.2:
                f[rng.randrange(13)] = rng.choice(pool)
        line = "\t".join(f)
        n, toks = oracle.split_line(line, False)
        rc, o = host.parse_overlap(line)
        if n != 13:
            assert rc == -1
            continue
        orc, oo = oracle.parse_fields(toks)
        assert (rc == 0) == (orc == 0), (f, rc, orc)
        if rc == 0:
            n_ok += 1
            assert o == oo, f
        else:
            n_bad += 1
    assert n_ok > 1000 and n_bad > 100


def _write(path, text):
    with open(path, "w") as f:
        f.write(text)
    return path


def test_fastq_storage_semantics(tmp_path):
    s = _write(tmp_path / "s.fastq", "@10 extra words\nacgtNNac\n+\nIIIIIIII\n@0x11\nGGGG\n+anything\n!!!!\n@trailing\nAC\n")
    p1 = _write(tmp_path / "p1.fastq", "@20/1\nacgt\n+\nIIII\n@21\nTTTTT\n+\n55555\n")
    p2 = _write(tmp_path / "p2.fastq", "@20/1\nCCcc\n+\n####\n@21\nGG\n+\nII\n")
    f = host.Fastq(singles=s, paired1=p1, paired2=p2)
    assert (f.n_reads, f.n_single, f.n_paired, f.n_seq) == (4, 2, 2, 6)
    assert f.read_ids.tolist() == [10, 17, 20, 21]         # strtoul(base 0): "0x11" = 17, "20/1" stops at the slash
    rs = f.readset()
    assert rs.seq(0) == (b"ACGTNNAC", b"IIIIIIII")         # singles are upper-cased (FastqStorage.cpp:122)
    assert rs.seq(2) == (b"acgt", b"IIII")                  # pairs are not (FastqStorage.cpp:197-198)
    assert rs.seq(3) == (b"CCcc", b"####")
    assert f.read_first_seq.tolist() == [0, 1, 2, 4, 6]
    # "None" means absent (FastqStorage.h:71,75)
    g = host.Fastq(singles=s, paired1="None", paired2="None")
    assert (g.n_single, g.n_paired) == (2, 0)
    h = host.Fastq(singles=s, max_reads=1)
    assert h.n_single == 1


def test_fastq_ids_file_and_errors(tmp_path):
    s = _write(tmp_path / "s.fastq", "@readA\nACGT\n+\nIIII\n@readB\nAC\n+\nII\n")
    ids = _write(tmp_path / "ids.txt", "5\t>readA\n9\treadB\n")
    f = host.Fastq(singles=s, ids=ids)
    assert f.read_ids.tolist() == [5, 9]
    with pytest.raises(hc.HcError):
        host.Fastq(singles=s, ids=_write(tmp_path / "ids2.txt", "5\t>readA\n"))        # missing id: map::at throws
    with pytest.raises(hc.HcError):
        host.Fastq(singles=_write(tmp_path / "bad1.fastq", "readA\nACGT\n+\nIIII\n"))  # no '@': exit(1)
    with pytest.raises(hc.HcError):
        host.Fastq(singles=_write(tmp_path / "bad2.fastq", "@r\n\n+\n\n"))             # empty sequence: exit(1)
    with pytest.raises(hc.HcError):
        host.Fastq(singles=str(tmp_path / "missing.fastq"))                              # cannot open: exit(1)
    with pytest.raises(hc.HcError):
        host.Fastq(paired1=_write(tmp_path / "q1.fastq", "@a\nAC\n+\nII\n"),
                   paired2=_write(tmp_path / "q2.fastq", "@b\nAC\n+\nII\n"))            # /1 /2 ids differ: exit(1)


def _dataset(tmp_path, seed=7):
    reads, meta = synth.make_paired_dataset(300, 1200, flip_frac=0.2, seed=seed)
    reads.quals[:] = ord("I")
    sreads, smeta = synth.make_single_dataset(200, 1500, len_lo=150, len_hi=400, seed=seed + 1, quals=[ord("I"), ord("5")])
    return reads, meta, sreads, smeta


def test_parser_and_prefilter_match_oracle(oracle, tmp_path):
    reads, meta, _, _ = _dataset(tmp_path)
    cand = synth.paired_candidates(meta, n_candidates=4000, seed=3)
    lines = synth.records_to_lines(cand, reads)
    rng = random.Random(9)
    extra = ["", "garbage", "1\t2\t3", lines[0] + "\textra", "5\t5\t0\t0\t1\t+\t+\t90\t90\t100\t100\tp\tp",  # self overlap
             "1\t2\t0\t0\t1\t+\t+\t90\t90\t10\t10\tp\tp", "  " + lines[1] + "\t "]
    for e in extra:
        lines.insert(rng.randrange(len(lines)), e)
    path = str(tmp_path / "overlaps.txt")
    _write(path, "\n".join(lines) + "\n")
    reads.write_fastq(paired1_path=str(tmp_path / "p1.fastq"), paired2_path=str(tmp_path / "p2.fastq"))
    f = host.Fastq(paired1=str(tmp_path / "p1.fastq"), paired2=str(tmp_path / "p2.fastq"))
    for st in (hc.Settings(min_overlap_len=150), hc.Settings(min_overlap_len=200, min_overlap_perc=70),
               hc.Settings(min_overlap_len=260, flags=hc.records.FLAG_RELAX_PE_EDGES | hc.records.FLAG_RESOLVE_ORIENTATIONS),
               hc.Settings(min_overlap_len=150, max_overlaps=1000)):
        recs, c = f.parse_file(st, path)
        rc, g, oc = oracle.construct_edges(reads, st, path, None)
        assert rc == 0
        assert c["lines_read"] == oc.lines_read and c["malformed_lines"] == oc.malformed_lines
        assert c["prefilter_rejected"] == oc.prefilter_rejected and recs.size == oc.scored
    # the records themselves: what the parser hands to the device == what the generator meant
    recs, _ = f.parse_file(hc.Settings(min_overlap_len=150), path)
    key = lambda a: np.stack([a[k].astype(np.int64) for k in ("read1", "read2", "pos1", "pos2", "ori1", "ori2", "ord", "len1", "len2", "perc")], 1)
    want = cand[(cand["len1"] >= 75) & (cand["len2"] >= 75)]
    got = recs[: recs.size]
    assert got.size == want.size + 1  # + the trimmed duplicate of lines[1]
    assert set(map(tuple, key(want).tolist())) == set(map(tuple, key(got).tolist()))


def test_parallel_parser_equals_serial(tmp_path):
    reads, meta = synth.make_paired_dataset(1500, 1500, seed=17)
    cand = synth.paired_candidates(meta, n_candidates=30000, seed=5)
    lines = synth.records_to_lines(cand, reads)
    rng = random.Random(4)
    for junk in ["", "x\ty", "7\t7\t0\t0\t1\t+\t+\t90\t90\t100\t100\tp\tp", "1\t2\t0\t0\t1\t+\t+\t90\t90\t10\t10\tp\tp"] * 25:
        lines.insert(rng.randrange(len(lines)), junk)
    path = str(tmp_path / "overlaps.txt")
    _write(path, "\n".join(lines))  # no trailing newline on purpose
    reads.write_fastq(paired1_path=str(tmp_path / "p1.fastq"), paired2_path=str(tmp_path / "p2.fastq"))
    f = host.Fastq(paired1=str(tmp_path / "p1.fastq"), paired2=str(tmp_path / "p2.fastq"))
    for max_ov in (100000000, 12345, 1, len(lines), len(lines) - 1):
        base = None
        for threads in (1, 2, 7, 32):
            recs, c = f.parse_file(hc.Settings(min_overlap_len=150, max_overlaps=max_ov, n_threads=threads), path)
            got = (recs.tobytes(), c["lines_read"], c["malformed_lines"], c["prefilter_rejected"], c["scored"])
            if base is None:
                base = got
                assert c["lines_read"] == min(max_ov, len(lines))
            assert got == base, (max_ov, threads)


def _random_edge_stream(rng, V, n):
    """Edges with many duplicates of the same unordered pair / orientation class and engineered ties."""
    out = np.zeros(n, dtype=host.EDGE_DTYPE)
    scores = [0.97, 0.98, 0.98, 0.99, 0.5, 1.0]
    for i in range(n):
        a, b = rng.sample(range(V), 2)
        e = out[i]
        e["v1"], e["v2"], e["read1"], e["read2"] = a, b, a, b
        e["score"] = rng.choice(scores)
        e["mismatch_rate"] = rng.choice([0.0, 0.0, 0.01, 0.5])
        e["pos1"] = rng.choice([0, 0, 3, 7])
        e["pos2"] = rng.choice([0, 2, 5])
        e["pos3"] = rng.choice([-5, 0, 5])
        e["pos4"] = rng.choice([-2, 0, 2])
        e["ori1"], e["ori2"] = rng.randrange(2), rng.randrange(2)
        e["ord"] = ord(rng.choice("-12"))
        e["perc"] = rng.choice([100, 100, 80])
        e["len1"] = rng.choice([100, 120]); e["len2"] = rng.choice([0, 50]); e["len0"] = e["len1"] + e["len2"]
    return out


def test_serial_insert_and_tie_break_chain_match_oracle(oracle):
    rng = random.Random(11)
    for flags in (hc.records.FLAG_RESOLVE_ORIENTATIONS, hc.records.FLAG_RESOLVE_ORIENTATIONS | hc.records.FLAG_IGNORE_INCLUSIONS):
        st = hc.Settings(flags=flags)
        V = 12  # few vertices => most inserts hit an existing pair
        stream = _random_edge_stream(rng, V, 4000)
        g = host.HostGraph(V, st)
        og = oracle.Graph(V)
        oc = oracle.hco_counters()
        for e in stream:
            assert g.insert(e.copy()) == 0
            assert og.insert(st, np.array([e], dtype=oracle.GEDGE_DTYPE), oc) == 0
        edges, inc, c = g.get()
        want = og.all_edges()
        assert edges.size == want.size == og.edge_count()
        for k in ("score", "mismatch_rate", "pos1", "pos2", "pos3", "pos4", "ori1", "ori2", "ord", "read1", "read2", "v1",
                  "v2", "perc", "len0", "len1", "len2"):
            assert np.array_equal(edges[k], want[k]), k
        assert np.array_equal(inc, og.inclusions())
        assert c["dup_count"] == oc.dup_count > 1000 and c["inclusion_count"] == oc.inclusion_count
        assert c["edges_added"] == oc.edges_added
        # the sort-based resolution must leave the identical graph and counters
        g2 = host.HostGraph(V, st)
        assert g2.resolve(stream) == 0
        edges2, inc2, c2 = g2.get()
        assert edges2.tobytes() == edges.tobytes() and np.array_equal(inc2, inc)
        assert (c2["dup_count"], c2["inclusion_count"], c2["edges_added"]) == (c["dup_count"], c["inclusion_count"], c["edges_added"])


def test_parallel_resolution_and_bulk_fill_match_the_serial_insert():
    """Enough edges for the threaded phases (slot partitions, vertex-range fill, concurrent slot index): the graph after
    resolve() must equal the one the per-edge insert leaves, and must keep behaving like it for later inserts."""
    rng = random.Random(5)
    for flags, V, n_threads in ((hc.records.FLAG_RESOLVE_ORIENTATIONS, 900, 8),
                                (hc.records.FLAG_RESOLVE_ORIENTATIONS | hc.records.FLAG_IGNORE_INCLUSIONS, 5000, 3),
                                (hc.records.FLAG_RESOLVE_ORIENTATIONS, 40, 32)):
        st = hc.Settings(flags=flags, n_threads=n_threads)
        stream = _random_edge_stream(rng, V, 40000)
        later = _random_edge_stream(rng, V, 3000)
        g = host.HostGraph(V, st)
        for e in stream:
            assert g.insert(e.copy()) == 0
        g2 = host.HostGraph(V, st)
        assert g2.resolve(stream) == 0
        edges, inc, c = g.get()
        edges2, inc2, c2 = g2.get()
        assert edges2.tobytes() == edges.tobytes() and np.array_equal(inc2, inc)
        assert (c2["dup_count"], c2["inclusion_count"], c2["edges_added"]) == (c["dup_count"], c["inclusion_count"], c["edges_added"])
        for e in later:
            assert g.insert(e.copy()) == 0 and g2.insert(e.copy()) == 0
        edges, inc, c = g.get()
        edges2, inc2, c2 = g2.get()
        assert edges2.tobytes() == edges.tobytes() and np.array_equal(inc2, inc)
        assert (c2["dup_count"], c2["inclusion_count"], c2["edges_added"]) == (c["dup_count"], c["inclusion_count"], c["edges_added"])


def test_one_pass_line_reader_agrees_with_the_general_path(oracle):
    """The stage reads plain lines in one pass (Overlap::from_plain_line) and everything else through the reference's
    tokenise + construct steps; mutate valid lines character by character and require both routes — and the oracle —
    to agree on acceptance and on every field."""
    rng = random.Random(17)

    def valid():
        t1, t2 = rng.choice("sp"), rng.choice("sp")
        ss = t1 == t2 == "s"
        return "\t".join([str(rng.choice([0, 7, 10, 123456, 99999999, 123456789012345678])), str(rng.randrange(5000)),
                          str(rng.randrange(300)), "-" if ss else str(rng.randrange(300)), rng.choice("12") if t1 == t2 == "p" else "-",
                          rng.choice("+-"), rng.choice("+-"), str(rng.randrange(101)), "-" if ss else str(rng.randrange(101)),
                          str(rng.randrange(1, 999999999)), "-" if ss else str(rng.randrange(500)), t1, t2])

    junk = ["\t", " ", "0", "9", "-", "+", "x", "s", "p", "1", "\r", "00", "1234567890", "0x1f", "101", ""]
    n_plain = n_general = n_bad = 0
    for it in range(6000):
        line = valid()
        for _ in range(rng.choice([0, 0, 1, 1, 2, 4])):
            k = rng.randrange(len(line) + 1)
            line = line[:k] + rng.choice(junk) + line[k + rng.choice([0, 1, 1]):]
        for sp in (False, True):
            rc, o = host.parse_overlap(line, sp)
            rc2, o2 = host.parse_overlap(line, sp, general_only=True)
            assert rc == rc2 and o == o2, (line, sp, rc, rc2, o, o2)
            n, toks = oracle.split_line(line, sp)
            if n != 13:
                assert rc == -1, line
                n_bad += 1
                continue
            orc, oo = oracle.parse_fields(toks)
            assert (rc == 0) == (orc == 0), (line, rc, orc)
            if rc == 0:
                assert o == oo, line
                n_plain += 1
            else:
                n_general += 1
    assert n_plain > 4000 and n_general > 500 and n_bad > 500


def test_fastq_reader_line_semantics(tmp_path):
    """std::getline semantics of the reference's reader (src/FastqStorage.cpp:42-57) over the mapped file: lines end
    at '\\n' only, a last line without a newline counts, an incomplete record is ignored, the shorter mate file bounds
    the pairs, --max_reads counts records."""
    f = host.Fastq(singles=_write(tmp_path / "a.fastq", "@1\nACGT\n+\nIIII\n@2\nGG\n+\n55"))       # no newline at the end
    assert f.read_ids.tolist() == [1, 2] and f.readset().seq(1) == (b"GG", b"55")
    f = host.Fastq(singles=_write(tmp_path / "b.fastq", "@1\nACGT\n+\nIIII\n@2\nGG\n+\n"))          # fourth line missing
    assert f.read_ids.tolist() == [1]
    with pytest.raises(hc.HcError):                                                                   # ... but an empty fourth line is a line
        host.Fastq(singles=_write(tmp_path / "b2.fastq", "@1\nACGT\n+\nIIII\n@2\nGG\n+\n\n"))
    f = host.Fastq(singles=_write(tmp_path / "c.fastq", ""))
    assert f.n_reads == 0
    f = host.Fastq(singles=_write(tmp_path / "d.fastq", "@7\tx\nAC\r\n+\nII\r\n"))                   # '\\r' stays part of the line
    assert f.readset().seq(0) == (b"AC\r", b"II\r")
    f = host.Fastq(singles=_write(tmp_path / "e.fastq", "@  12 z\nAC\n+\nII\n"))                     # operator>> skips leading blanks
    assert f.read_ids.tolist() == [12]
    p1 = _write(tmp_path / "p1.fastq", "@1\nAC\n+\nII\n@2\nGG\n+\nII\n@3\nTT\n+\nII\n")
    p2 = _write(tmp_path / "p2.fastq", "@1\nCA\n+\nII\n@2\nCC\n+\nII\n@3\nAA\n+\n")                  # third record incomplete in /2
    f = host.Fastq(paired1=p1, paired2=p2)
    assert f.read_ids.tolist() == [1, 2]
    f = host.Fastq(paired1=p1, paired2=p2, max_reads=1)
    assert f.read_ids.tolist() == [1]
    with pytest.raises(hc.HcError):
        host.Fastq(singles=_write(tmp_path / "g.fastq", "@1\nACGT\n+\nIII\n"))                        # lengths differ


def test_id_lookup_sparse_and_duplicate_ids(tmp_path):
    """Overlaps-file ids -> read indices (std::map::at on m_ID_to_index in the reference): ids far beyond the read
    count take the hashed table, small ones the direct table; a duplicated id resolves to its first read."""
    for big in (10 ** 12, 900):
        ids = [7, big, 7, 41]
        s = _write(tmp_path / f"s{big}.fastq", "".join(f"@{i}\nACGTACGTAC\n+\nIIIIIIIIII\n" for i in ids))
        f = host.Fastq(singles=s)
        assert f.read_ids.tolist() == ids
        ov = _write(tmp_path / f"ov{big}.txt", f"7\t{big}\t3\t-\t-\t+\t+\t70\t-\t7\t-\ts\ts\n41\t7\t2\t-\t-\t+\t-\t80\t-\t8\t-\ts\ts\n")
        recs, c = f.parse_file(hc.Settings(min_overlap_len=0), ov)
        assert recs["read1"].tolist() == [0, 3] and recs["read2"].tolist() == [1, 0]
        bad = _write(tmp_path / f"bad{big}.txt", f"7\t{big + 1}\t3\t-\t-\t+\t+\t70\t-\t7\t-\ts\ts\n")
        with pytest.raises(hc.HcError):
            f.parse_file(hc.Settings(min_overlap_len=0), bad)


def _csr_of(g, V):
    """The graph in the form hc_graph_fetch hands over: edges by v1 (list order kept), offsets, in-lists."""
    edges, inc, _ = g.get()
    out_off = np.zeros(V + 1, np.uint64)
    np.add.at(out_off, edges["v1"].astype(np.int64) + 1, 1)
    out_off = np.cumsum(out_off).astype(np.uint64)
    in_off, in_nodes = g.in_lists(edges.size)
    return edges, out_off, in_nodes, in_off, inc


def test_adopted_csr_graph_behaves_like_an_owned_one():
    """OverlapGraph::adopt_csr makes every adjacency list a view into one array; later inserts (the tie-break chain may
    remove, swap and append), sortEdges and the read-out must behave exactly as on a graph built edge by edge."""
    rng = random.Random(23)
    for flags, V, n_threads in ((hc.records.FLAG_RESOLVE_ORIENTATIONS, 60, 1),
                                (hc.records.FLAG_RESOLVE_ORIENTATIONS | hc.records.FLAG_IGNORE_INCLUSIONS, 700, 4)):
        st = hc.Settings(flags=flags, n_threads=n_threads)
        stream = _random_edge_stream(rng, V, 6000)
        later = _random_edge_stream(rng, V, 1500)
        g = host.HostGraph(V, st)
        for e in stream:
            assert g.insert(e.copy()) == 0
        edges, out_off, in_nodes, in_off, inc = _csr_of(g, V)
        assert int(out_off[-1]) == edges.size == int(in_off[-1])
        g2 = host.HostGraph(V, st)
        assert g2.adopt(edges, out_off, in_nodes, in_off, inc) == 0
        edges2, out_off2, in_nodes2, in_off2, inc2 = _csr_of(g2, V)
        assert edges2.tobytes() == edges.tobytes() and np.array_equal(inc2, inc)
        assert np.array_equal(in_nodes2, in_nodes) and np.array_equal(in_off2, in_off)
        for e in later:  # grows some lists past their borrowed capacity, removes from others
            assert g.insert(e.copy()) == 0 and g2.insert(e.copy()) == 0
        a, b = _csr_of(g, V), _csr_of(g2, V)
        assert a[0].tobytes() == b[0].tobytes()
        for x, y in zip(a[1:], b[1:]):
            assert np.array_equal(x, y)
        lens = np.full(V, 150, np.uint32)
        g.sort_edges(lens)
        g2.sort_edges(lens)
        assert g.get()[0].tobytes() == g2.get()[0].tobytes()
        # adopting into a graph that already holds edges is refused, not merged
        assert g2.adopt(edges, out_off, in_nodes, in_off, inc) != 0
        # caller-supplied arrays are checked: an in-list entry that is no vertex, an edge filed under the wrong vertex; a
        # refused adopt leaves the graph empty and fit for the next one
        g3 = host.HostGraph(V, st)
        bad_nodes = in_nodes.copy()
        bad_nodes[bad_nodes.size // 2] = V
        assert g3.adopt(edges, out_off, bad_nodes, in_off, inc) != 0
        assert g3.get()[0].size == 0 and not g3.get()[1].any()
        bad_edges = edges.copy()
        bad_edges["v1"][edges.size // 3] += 1
        assert g3.adopt(bad_edges, out_off, in_nodes, in_off, inc) != 0
        assert g3.adopt(edges, out_off, in_nodes, in_off, inc) == 0
        assert g3.get()[0].tobytes() == edges.tobytes()


def test_add_equivalent_edges_matches_oracle(oracle):
    """--add_duplicates: OverlapGraph::addEquivalentEdges (src/OverlapGraph.cpp:608-719) on an owned and on an adopted graph."""
    rng = random.Random(3)
    R = 40
    flags = hc.records.FLAG_RESOLVE_ORIENTATIONS | hc.records.FLAG_ADD_DUPLICATES
    st = hc.Settings(flags=flags)
    stream = _random_edge_stream(rng, R, 1200)
    stream["v1"] = np.where(stream["ori1"] != 0, stream["read1"], R + stream["read1"].astype(np.int64))
    stream["v2"] = np.where(stream["ori2"] != 0, stream["read2"], R + stream["read2"].astype(np.int64))
    g = host.HostGraph(2 * R, st)
    og = oracle.Graph(2 * R)
    oc = oracle.hco_counters()
    for e in stream:
        assert g.insert(e.copy()) == 0
        assert og.insert(st, np.array([e], dtype=oracle.GEDGE_DTYPE), oc) == 0
    csr = _csr_of(g, 2 * R)
    g2 = host.HostGraph(2 * R, st)
    assert g2.adopt(*csr) == 0
    assert g.add_equivalent_edges() == 0 and g2.add_equivalent_edges() == 0
    assert og.add_equivalent_edges(R) == 0
    want = og.all_edges()
    for got in (g.get()[0], g2.get()[0]):
        assert got.size == want.size > csr[0].size
        for k in ("score", "pos1", "pos2", "ori1", "ori2", "ord", "read1", "read2", "v1", "v2"):
            assert np.array_equal(got[k], want[k]), k
    assert g.get()[0].tobytes() == g2.get()[0].tobytes()
    a, b = _csr_of(g, 2 * R), _csr_of(g2, 2 * R)
    for x, y in zip(a[1:], b[1:]):
        assert np.array_equal(x, y)


def test_threaded_fastq_reader_equals_the_sequential_one(tmp_path, monkeypatch):
    """Random single- and paired-end files (mixed case, CR-LF, lengths 1..300, header comments, a missing final newline, an
    incomplete last record, --max_reads in the middle): the mapped reader on five threads must leave exactly the arrays
    the sequential line reader leaves."""
    rng = random.Random(17)

    def rec(i, n, crlf):
        eol = "\r\n" if crlf else "\n"
        seq = "".join(rng.choice("ACGTNacgtn") for _ in range(n))
        qual = "".join(chr(rng.randrange(33, 110)) for _ in range(n))
        return f"@{i} c{rng.randrange(9)}{eol}{seq}{eol}+{eol}{qual}{eol}"

    for trial in range(6):
        crlf = trial == 2
        n_s, n_p = rng.randrange(50, 400), rng.randrange(50, 400)
        s = "".join(rec(i, rng.randrange(1, 300), crlf) for i in range(n_s))
        p1 = "".join(rec(1000 + i, rng.randrange(1, 300), crlf) for i in range(n_p))
        p2 = "".join(rec(1000 + i, rng.randrange(1, 300), crlf) for i in range(n_p))
        if trial == 3:
            s, p1 = s[:-1], p1[:-1]  # no final newline
        if trial == 4:
            s += "@77\nACGT\n+\n"  # incomplete record
            p2 += "@5000\nAC\n"
        paths = [str(tmp_path / f"{k}{trial}.fastq") for k in ("s", "p1", "p2")]
        for path, text in zip(paths, (s, p1, p2)):
            open(path, "w", newline="").write(text)
        for max_reads in (0, 73):
            got = []
            for threads, grain in (("1", None), ("5", "1")):
                monkeypatch.setenv("HC_FASTQ_THREADS", threads)
                monkeypatch.setenv("HC_FASTQ_PARALLEL_MIN", "0")
                if grain:
                    monkeypatch.setenv("HC_FASTQ_GRAIN", grain)
                else:
                    monkeypatch.delenv("HC_FASTQ_GRAIN", raising=False)
                f = host.Fastq(singles=paths[0], paired1=paths[1], paired2=paths[2], max_reads=max_reads)
                got.append((f.n_single, f.n_paired, f.bases.tobytes(), f.quals.tobytes(), f.seq_off.tobytes(), f.read_first_seq.tobytes(),
                            f.read_ids.tobytes()))
                f.close()
            assert got[0] == got[1], (trial, max_reads)
            assert got[0][0] > 0 and got[0][1] > 0
