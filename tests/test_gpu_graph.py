"""The device legs the stage is built from, each against its host / oracle counterpart on the GPU box:
compact candidate records (hc_cand_rec), blocks in flight (hc_block_*), and the duplicate resolution + adjacency
lists of hc_graph_resolve against the per-edge serial insert of the host mirror (which the
reference's own process_overlaps pins, tests/test_ec_golden.py) and its sortEdges."""
import os
import random

import numpy as np
import pytest

import haploconduct_amd as hc
from haploconduct_amd import host, synth
from haploconduct_amd.host import EDGE_DTYPE
from haploconduct_amd.records import ADMIT_DTYPE, FLAG_IGNORE_INCLUSIONS, FLAG_RESOLVE_ORIENTATIONS, RESULT_DTYPE, result_cls

pytestmark = pytest.mark.gpu


def _workload(name):
    if name == "pp":
        reads, meta = synth.make_paired_dataset(1200, 2500, flip_frac=0.3, seed=21)
        return reads, synth.paired_candidates(meta, n_candidates=20000, seed=22), hc.Settings(edge_threshold=0.97)
    if name == "ss":
        reads, meta = synth.make_single_dataset(900, 4000, len_lo=150, len_hi=1200, flip_frac=0.4, seed=23, log_uniform=True)
        return reads, synth.single_candidates(meta, min_overlap=80, n_candidates=30000), hc.Settings(edge_threshold=0.995, min_read_len=160)
    quals = (np.arange(1, 61) + 33).astype(np.uint8)  # 60 quality values: 16-bit symbols
    reads, meta = synth.make_single_dataset(600, 3000, len_lo=200, len_hi=260, flip_frac=0.5, seed=24, quals=quals)
    return reads, synth.single_candidates(meta, min_overlap=100, n_candidates=20000), hc.Settings(edge_threshold=1.0)


@pytest.mark.parametrize("name", ["pp", "ss", "ss16"])
def test_compact_records_score_exactly_like_full_records(name):
    """hc_score_cands (16-byte records, what crosses PCIe in the stage) and hc_score_batch (32-byte records) give the same
    bytes, malformed records and saturated positions included; the blocks deliver exactly the non-dropped ones."""
    reads, cand, st = _workload(name)
    cand = cand.copy()
    rng = np.random.default_rng(3)
    bad = rng.choice(cand.size, 60, replace=False)
    cand["read2"][bad[:15]] = cand["read1"][bad[:15]]          # self overlap: malformed
    cand["read1"][bad[15:30]] = reads.n_reads + 5               # out of range
    cand["pos1"][bad[30:45]] = 0xFFFFFFF0                       # beyond every sequence: saturates in the compact form
    cand["pos2"][bad[45:60]] = (1 << 28) + 7
    with hc.EdgeScorer(st) as sc:
        sc.set_reads(reads)
        full = sc.score_batch(cand)
        cd = sc.pack_cands(cand)
        compact = sc.score_cands(cd)
        assert full.tobytes() == compact.tobytes()
        assert (result_cls(full)[bad[:30]] == 7).all()
        kept = np.nonzero(result_cls(full) != 0)[0]
        assert kept.size > 100
        for block in (cand.size, 7000, 1111):
            rows = sc.score_blocks(cd, block=block, in_flight=3)
            assert np.array_equal(rows["index"], kept.astype(np.uint64)), block
            for k in ("x1", "x2"):
                assert np.array_equal(rows[k].view(np.uint64), full[k][kept].view(np.uint64)), (block, k)
            assert np.array_equal(rows["mm"], full["mm"][kept]) and np.array_equal(rows["n_cls"], full["n_cls"][kept])


def _admitted(seed, n_reads, m, paired_frac=0.0):
    """Random admitted candidates over a read set with engineered duplicates and ties: few distinct scores, lengths and
    positions, both orientation classes, pos1 == 0 (direction decided by the vertex ids)."""
    rng = np.random.default_rng(seed)
    n_single = n_reads - int(n_reads * paired_frac)

    def one():
        L = int(rng.integers(60, 200))
        return ("ACGT"[int(rng.integers(4))] * L, "I" * L)

    reads = hc.ReadSet.from_lists([one() for _ in range(n_single)], [(one(), one()) for _ in range(n_reads - n_single)])
    adm = np.zeros(m, ADMIT_DTYPE)
    hot = rng.integers(0, n_reads, 40)  # most records among few reads: deep slots
    a = np.where(rng.random(m) < 0.7, hot[rng.integers(0, hot.size, m)], rng.integers(0, n_reads, m))
    b = np.where(rng.random(m) < 0.7, hot[rng.integers(0, hot.size, m)], rng.integers(0, n_reads, m))
    b = np.where(a == b, (b + 1) % n_reads, b)
    adm["read1"], adm["read2"] = a, b
    adm["score"] = rng.choice([0.97, 0.98, 0.98, 0.99, 1.0], m)
    adm["pos1"] = rng.choice([0, 0, 3, 17], m)
    adm["pos2"] = rng.choice([0, 5], m)
    adm["mm"] = rng.choice([0, 0, 1, 2], m)
    adm["n"] = rng.choice([50, 100], m)
    adm["len1"] = rng.choice([40, 50, 50, 60], m)
    adm["len2"] = rng.choice([0, 30, 30], m)
    adm["perc"] = rng.choice([60, 100, 100], m)
    adm["ori1"], adm["ori2"] = rng.integers(0, 2, m), rng.integers(0, 2, m)
    adm["ord"] = np.where(rng.random(m) < 0.5, ord("1"), ord("2"))
    return reads, adm


def _host_edges(reads, adm):
    """The Edge compute_overlap builds (EdgeCalculator.cpp:219-232, :254-270, :292-308, :353-379) in numpy."""
    off = reads.seq_off.astype(np.int64)
    first = reads.read_first_seq.astype(np.int64)
    paired = (first[1:] - first[:-1]) == 2
    la = off[first[:-1] + 1] - off[first[:-1]]
    lb = np.where(paired, off[np.minimum(first[:-1] + 2, off.size - 1)] - off[first[:-1] + 1], 0)
    r1, r2 = adm["read1"].astype(np.int64), adm["read2"].astype(np.int64)
    p1, p2 = paired[r1], paired[r2]
    pos1, pos2 = adm["pos1"].astype(np.int64), adm["pos2"].astype(np.int64)
    one = adm["ord"] == ord("1")
    pos3 = np.where(~p1 & ~p2, la[r1] - pos1 - la[r2],
                    np.where(~p1 & p2, la[r1] - pos2 - lb[r2],
                             np.where(p1 & ~p2, lb[r1] + pos2 - la[r2], np.where(one, lb[r1] - pos2 - lb[r2], lb[r1] + pos2 - lb[r2]))))
    pos4 = np.where(~p1 & ~p2, 0, np.where(~p1 & p2, la[r1] - pos1 - la[r2], np.where(p1 & ~p2, la[r2] + pos1 - la[r1], la[r1] - pos1 - la[r2])))
    e = np.zeros(adm.size, EDGE_DTYPE)
    e["score"] = adm["score"]
    e["mismatch_rate"] = adm["mm"].astype(np.float32).astype(np.float64) / adm["n"].astype(np.float64)
    e["pos1"], e["pos2"], e["pos3"], e["pos4"] = pos1, pos2, pos3, pos4
    e["ori1"], e["ori2"], e["ord"] = adm["ori1"], adm["ori2"], adm["ord"]
    e["read1"], e["read2"], e["v1"], e["v2"] = r1, r2, r1, r2
    e["perc"] = adm["perc"]
    ss = ~p1 & ~p2
    e["len1"] = adm["len1"]
    e["len2"] = np.where(ss, 0, adm["len2"])
    e["len0"] = e["len1"] + e["len2"]
    return e, (la + lb).astype(np.uint32)


@pytest.mark.parametrize("seed,paired_frac", [(1, 0.0), (2, 0.5), (3, 1.0)])
def test_device_resolution_is_the_serial_insert(seed, paired_frac):
    """hc_graph_resolve on engineered duplicates / ties against hc_host_graph_insert, one call per record (the reference's
    serial half): identical adjacency lists in list order, in-lists, inclusions bits and counters; then the sortEdges
    order of both."""
    V, m = 300, 40000
    reads, adm = _admitted(seed, V, m, paired_frac)
    st = hc.Settings(edge_threshold=0.97, flags=FLAG_RESOLVE_ORIENTATIONS | FLAG_IGNORE_INCLUSIONS)
    edges, len_by_read = _host_edges(reads, adm)
    g = host.HostGraph(V, st)
    for k in range(m):
        assert g.insert(edges[k]) == 0
    want, want_inc, wc = g.get()
    with hc.EdgeScorer(st) as sc:
        sc.set_reads(reads)
        got = sc.graph_resolve(adm, V)
        assert got["counts"]["first_bad"] == -1
        assert got["edges"].tobytes() == want.tobytes()
        # handed over piecewise (what the stage does while later blocks are still being scored): same graph
        again = sc.graph_resolve(adm, V, pieces=37)
        assert again["edges"].tobytes() == want.tobytes() and again["counts"] == got["counts"]
        assert np.array_equal(again["in_nodes"], got["in_nodes"]) and np.array_equal(again["seq"], got["seq"])
        assert np.array_equal(got["inclusions"], want_inc) and want_inc.sum() > 0
        assert got["counts"]["dup_count"] == wc["dup_count"] > 1000 and got["counts"]["inclusion_count"] == wc["inclusion_count"]
        assert got["counts"]["n_edges"] == want.size == wc["edges_added"]
        # adj_in: the host's list is [order of addEdge calls, with erase-first-occurrence on replacement]; the device's is
        # the survivors' sequence order.  They agree as multisets per vertex always, and as lists unless a vertex pair holds
        # both orientation classes: compare as sorted lists per vertex.
        off, nodes = g.in_lists(want.size)
        assert np.array_equal(off, got["in_off"])
        for v in range(V):
            a, b = int(off[v]), int(off[v + 1])
            assert sorted(nodes[a:b]) == sorted(got["in_nodes"][a:b])
        # every edge knows its place in the insertion sequence
        assert np.array_equal(adm["read1"][got["seq"]] * 0 + got["edges"]["score"], adm["score"][got["seq"]])
        # sortEdges order
        g.sort_edges(len_by_read)
        swant, _, _ = g.get()
        soff, snodes = g.in_lists(want.size)
        sgot = sc.graph_resolve(adm, V, sorted_order=True)
        tied = set(int(v) for v in sgot["tied_vertices"])
        assert len(tied) == sgot["counts"]["n_tied_lists"]
        out_off = sgot["out_off"]
        for v in range(V):
            a, b = int(out_off[v]), int(out_off[v + 1])
            if v in tied:  # reported, not guessed: same edges, order left to the host's std::sort
                assert b - a > 16
                assert sorted(sgot["edges"][a:b].tobytes()[i * 80:(i + 1) * 80] for i in range(b - a)) == \
                       sorted(swant[a:b].tobytes()[i * 80:(i + 1) * 80] for i in range(b - a))
            else:
                assert sgot["edges"][a:b].tobytes() == swant[a:b].tobytes(), v
        if not tied:
            assert np.array_equal(soff, sgot["in_off"]) and np.array_equal(snodes, sgot["in_nodes"].astype(np.uint64))


def test_device_resolution_reports_what_the_edge_constructor_rejects():
    reads, adm = _admitted(5, 50, 500)
    adm["len1"][123] = 0  # Edge::set_len asserts len1 > 0 (src/Edge.h:211-218)
    adm["len1"][400] = 0
    with hc.EdgeScorer(hc.Settings()) as sc:
        sc.set_reads(reads)
        got = sc.graph_resolve(adm, 50)
        assert got["counts"]["first_bad"] == 123
        adm["len1"][[123, 400]] = 10
        adm["read2"][77] = 50  # not a read
        assert sc.graph_resolve(adm, 50)["counts"]["first_bad"] == 77
        assert sc.graph_resolve(adm[:0], 50)["counts"]["n_edges"] == 0


def test_stage_fused_sort_with_fully_tied_edges_in_long_lists(oracle, tmp_path):
    """sortEdges' order among edges that compare equal is std::sort's (an introsort beyond 16 elements): a hub with 40
    out-edges, every neighbour twice — once per orientation class, same lengths — so that every pair ties.  The fused
    construct + sort call must give what construct_edges followed by sortEdges gives."""
    rng = random.Random(4)
    n = 64
    seqs = [("".join(rng.choice("ACGT") for _ in range(150)), "I" * 150) for _ in range(n)]
    reads = hc.ReadSet.from_lists(seqs)
    ids = reads.read_ids
    lines = []
    for hub in (3, 40):
        nb = [v for v in range(n) if v != hub]
        rng.shuffle(nb)
        for v in nb[:25]:
            for o2 in "+-":
                lines.append(f"{ids[hub]}\t{ids[v]}\t{50}\t-\t-\t+\t{o2}\t66\t-\t100\t-\ts\ts")
    rng.shuffle(lines)
    d = tmp_path
    ov = str(d / "overlaps.txt")
    open(ov, "w").write("\n".join(lines) + "\n")
    reads.write_fastq(str(d / "singles.fastq"))
    st = hc.Settings(edge_threshold=-1.0, ov_threshold=-1.0, min_overlap_len=50)  # every candidate is admitted
    out_dir = str(d) + "/"
    with host.EdgeCalculatorStage(st, singles=str(d / "singles.fastq"), overlaps=ov, output_dir=out_dir) as ec:
        ec.construct_edges()
        assert ec.edge_count() == 100
        ec.sort_edges()
        want, want_in = ec.edges(), ec.in_lists()
    with host.EdgeCalculatorStage(st, singles=str(d / "singles.fastq"), overlaps=ov, output_dir=out_dir) as ec:
        ec.construct_edges_sorted()
        got, got_in = ec.edges(), ec.in_lists()
    assert got.tobytes() == want.tobytes()
    assert np.array_equal(got_in[0], want_in[0]) and np.array_equal(got_in[1], want_in[1])


def _text_case(seed):
    reads, meta = synth.make_paired_dataset(900, 2200, flip_frac=0.3, seed=seed)
    cand = synth.paired_candidates(meta, n_candidates=15000, seed=seed + 1)
    sreads_lines = synth.records_to_lines(cand, reads)
    return reads, cand, sreads_lines


@pytest.mark.parametrize("block_bytes", [1 << 20, 40000, 3000])
def test_device_text_parser_is_the_host_parser(block_bytes, tmp_path):
    """hc_textblock_*: the overlaps file's text split, parsed, prefiltered and scored on the device against the host
    parser (pinned by the reference's own prefilter lines, tests/test_prefilter_golden.py) followed by hc_score_cands: same
    non-dropped records with the same parsed lines, same rejects in file order, same counters — through --max_ov, self
    overlaps, "-" columns, percentages below --min_overlap_perc, a last line without a newline."""
    from haploconduct_amd.records import FLAG_RELAX_PE_EDGES, LINE_DTYPE

    reads, cand, lines = _text_case(41)
    rng = random.Random(3)
    ids = reads.read_ids
    # shorter overlaps (prefilter rejects / relaxed passes), self overlaps, low percentages
    for k in rng.sample(range(len(lines)), 600):
        f = lines[k].split("\t")
        f[9] = str(rng.choice([10, 70, 74, 75, 76, 149, 150]))
        f[10] = str(rng.choice([10, 70, 74, 75, 76, 149, 150]))
        f[7] = str(rng.choice([0, 59, 60, 61, 100]))
        lines[k] = "\t".join(f)
    for k in rng.sample(range(len(lines)), 30):
        f = lines[k].split("\t")
        f[1] = f[0]
        lines[k] = "\t".join(f)
    text = "\n".join(lines)  # no trailing newline: the last piece is a line too
    path = str(tmp_path / "ov.txt")
    open(path, "w").write(text)
    reads.write_fastq(None, str(tmp_path / "p1.fastq"), str(tmp_path / "p2.fastq"))
    fq = host.Fastq(paired1=str(tmp_path / "p1.fastq"), paired2=str(tmp_path / "p2.fastq"))
    for st in (hc.Settings(edge_threshold=0.97, min_overlap_len=150, min_overlap_perc=60),
               hc.Settings(edge_threshold=0.97, min_overlap_len=151, flags=FLAG_RESOLVE_ORIENTATIONS | FLAG_RELAX_PE_EDGES, max_overlaps=9000)):
        want_recs, wc = fq.parse_file(st, path)
        with hc.EdgeScorer(st) as sc:
            sc.set_reads(reads)
            sc.set_ids(ids)
            want_res = sc.score_batch(want_recs)
            blocks = sc.score_text(text, block_bytes=block_bytes)
            # the same text read in place by the device, two blocks in flight, line numbers (--max_ov) through the line chain
            chained = sc.score_text(text, block_bytes=block_bytes, chained=True)
        assert len(chained) == len(blocks)
        for x, y in zip(blocks, chained):
            for k in x:
                if k == "regrown":  # a count of the block object, not of the text
                    continue
                if k == "rows":  # the chained submit numbers a block's rows from 0 (the host learns the line counts afterwards)
                    yr = y[k].copy()
                    yr["row"]["index"] += y["base"]
                    assert x[k].tobytes() == yr.tobytes(), k
                elif k == "rejected":
                    assert x[k].tobytes() == y[k].tobytes(), k
                else:
                    assert x[k] == y[k], k
        assert all(b["needs_host"] == 0 for b in blocks)
        assert sum(b["lines_read"] for b in blocks) == wc["lines_read"] and sum(b["scored"] for b in blocks) == wc["scored"] == want_recs.size
        assert sum(b["prefilter_rejected"] for b in blocks) == wc["prefilter_rejected"] > 50
        assert sum(b["silently_dropped"] for b in blocks) == wc["silently_dropped"]
        assert sum(b["self_overlaps"] for b in blocks) > 0
        rows = np.concatenate([b["rows"] for b in blocks])
        assert (np.diff(rows["row"]["index"].astype(np.int64)) > 0).all()
        kept = np.nonzero(result_cls(want_res) != 0)[0]
        assert rows.size == kept.size > 300
        for k in ("x1", "x2"):
            assert np.array_equal(rows["row"][k].view(np.uint64), want_res[k][kept].view(np.uint64))
        assert np.array_equal(rows["row"]["mm"], want_res["mm"][kept]) and np.array_equal(rows["row"]["n_cls"], want_res["n_cls"][kept])
        # the parsed line travelling with every row is the host parser's record of that candidate
        ln, wr = rows["line"], want_recs[kept]
        assert np.array_equal(ln["id1"], ids[wr["read1"]]) and np.array_equal(ln["id2"], ids[wr["read2"]])
        assert np.array_equal(ln["pos1"], wr["pos1"]) and np.array_equal(ln["pos2"], wr["pos2"])
        assert np.array_equal(ln["len1"], wr["len1"]) and np.array_equal(ln["len2"], wr["len2"])
        assert np.array_equal(ln["ord"], wr["ord"]) and np.array_equal(ln["ori1"] == ord("+"), wr["ori1"] != 0)
        perc = np.where(ln["perc2"] > 0, (ln["perc1"] + ln["perc2"]) // 2, ln["perc1"])
        assert np.array_equal(perc, wr["perc"])
        # rejects: in file order, with the line number inside their block
        n_rej = sum(b["rejected"].size for b in blocks)
        assert n_rej == wc["prefilter_rejected"]
        for b in blocks:
            assert (np.diff(b["rejected"]["line_index"].astype(np.int64)) > 0).all()


def test_device_text_parser_hands_unusual_blocks_to_the_host():
    reads, cand, lines = _text_case(43)
    good = "\n".join(lines[:2000]) + "\n"
    with hc.EdgeScorer(hc.Settings()) as sc:
        sc.set_reads(reads)
        sc.set_ids(reads.read_ids)
        assert sc.score_text(good)[0]["needs_host"] == 0
        for bad in (lines[5] + " ", "  " + lines[5], "1\t2\t3", "", lines[5].replace("\t", " \t", 1), "0x10" + lines[5][lines[5].index("\t"):],
                    "007" + lines[5][lines[5].index("\t"):], lines[5] + "\tz", lines[5][:-1] + "q"):
            b = sc.score_text(good + bad + "\n" + good)[0]
            assert b["needs_host"] == 1 and b["n_nonplain"] == 1, bad
        unknown = lines[5].split("\t")
        unknown[1] = "99999999"
        b = sc.score_text(good + "\t".join(unknown) + "\n")[0]
        assert b["needs_host"] == 1 and b["n_unknown_id"] == 1 and b["n_nonplain"] == 0
        assert sc.score_text("\n" * 100000, block_bytes=1 << 20)[0]["needs_host"] == 1  # more lines than a block of plain lines can hold
        # sparse ids: the open-addressing table
        sparse = reads.read_ids * 1000003 + 17
        sc.set_ids(sparse)
        relabelled = []
        for ln in lines[:3000]:
            f = ln.split("\t")
            f[0], f[1] = str(int(f[0]) * 1000003 + 17), str(int(f[1]) * 1000003 + 17)
            relabelled.append("\t".join(f))
        sc.set_ids(reads.read_ids)
        a = sc.score_text("\n".join(lines[:3000]) + "\n")[0]
        sc.set_ids(sparse)
        b = sc.score_text("\n".join(relabelled) + "\n")[0]
        assert b["needs_host"] == 0 and a["rows"]["row"].tobytes() == b["rows"]["row"].tobytes()


def test_reserved_row_buffers_spare_the_block_its_second_run():
    """hc_textblock_reserve_rows (the one-call reads -> graph route grows its blocks while the finder runs): a block whose lines all
    survive grows its row buffers inside hc_textblock_wait and runs its device half again — unless they were reserved; same rows
    either way."""
    reads, meta = synth.make_paired_dataset(1500, 1200, n_strains=1, divergence=0.0, err=0.0, n_rate=0.0, seed=5)
    cand = synth.paired_candidates(meta, n_candidates=60000, seed=6)
    text = "\n".join(synth.records_to_lines(cand, reads)) + "\n"
    st = hc.Settings(edge_threshold=0.97, ov_threshold=0.9, min_overlap_len=150)
    with hc.EdgeScorer(st) as sc:
        sc.set_reads(reads)
        sc.set_ids(reads.read_ids)
        grown = sc.score_text(text, block_bytes=1 << 20)
        reserved = sc.score_text(text, block_bytes=1 << 20, reserve_rows=1 << 20)
    assert len(grown) == len(reserved) >= 2
    assert grown[0]["regrown"] >= 1, "every line survives: the first block is meant to overflow an eighth of its lines"
    assert all(b["regrown"] == 0 and b["needs_host"] == 0 for b in reserved)
    for x, y in zip(grown, reserved):
        assert x["rows"].tobytes() == y["rows"].tobytes() and x["n_rows"] == y["n_rows"] > 0.9 * x["n_lines"]
