"""construct_edges' line handling — tab tokeniser, 13-field check, Overlap constructor, self-overlap test, prefilter
(reference src/EdgeCalculator.cpp:590-596, 598-635) — pinned by vectors the reference's own lines produced
(tests/golden/prefilter.json, made by tests/golden/make_golden_prefilter.py through oracle/_ref/libhcref_prefilter.so):
the oracle's hco_construct_edges and the product's parser (hc_host_parse_file / hc_host_parse_overlap) must take the same
decision on every line, in sequence, and re-serialise the kept lines to the same bytes.  No GPU needed."""
import json
import os

import numpy as np
import pytest

import haploconduct_amd as hc
from haploconduct_amd import host
from haploconduct_amd.records import FLAG_RELAX_PE_EDGES, FLAG_RESOLVE_ORIENTATIONS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "prefilter.json")))


@pytest.fixture(scope="module")
def reads(tmp_path_factory):
    """Every id the vectors name, as a tiny single-end read (the prefilter looks at the types written in the file)."""
    n = 2200
    rs = hc.ReadSet.from_lists([("ACGTACGTAC" * 30, "I" * 300)] * n)
    d = tmp_path_factory.mktemp("pref")
    rs.write_fastq(str(d / "s.fastq"))
    return rs, str(d / "s.fastq")


def _settings(b, threads=1):
    return hc.Settings(edge_threshold=0.5, ov_threshold=1.0,  # no scored candidate is ever a "non-edge": the non-edge file holds the prefilter's rejects only
                       min_overlap_len=b["min_overlap_len"], min_overlap_perc=b["min_overlap_perc"], n_threads=threads,
                       flags=FLAG_RESOLVE_ORIENTATIONS | (FLAG_RELAX_PE_EDGES if b["relax_PE_edges"] else 0))


@pytest.mark.parametrize("k", range(len(GOLD["blocks"])))
def test_product_parser_takes_the_references_decisions(k, reads, tmp_path):
    b = GOLD["blocks"][k]
    rs, fq = reads
    path = str(tmp_path / "ov.txt")
    open(path, "w").write("\n".join(b["lines"]) + "\n")
    v = np.array(b["verdict"])
    f = host.Fastq(singles=fq)
    for threads in (1, 5):
        recs, c = f.parse_file(_settings(b, threads), path)
        assert c["lines_read"] == len(b["lines"]) and c["malformed_lines"] == int((v == 3).sum())
        assert c["prefilter_rejected"] == int((v == 2).sum()) and c["scored"] == recs.size == int((v == 1).sum())
        # which lines passed, in order: the id columns of the re-serialised lines
        want = [(int(t.split("\t")[0]), int(t.split("\t")[1])) for t, vv in zip(b["text"], b["verdict"]) if vv == 1]
        got = [(int(rs.read_ids[r["read1"]]), int(rs.read_ids[r["read2"]])) for r in recs]
        assert got == want
        # the same bytes in memory instead of a file (what hc_ec_construct_edges_from_reads hands to the parser): same records, same counters
        recs_m, c_m = f.parse_text(_settings(b, threads), "\n".join(b["lines"]) + "\n")
        assert recs_m.tobytes() == recs.tobytes() and c_m == c
    # the text the kept lines re-serialise to, both reader routes
    for ln, vv, t in zip(b["lines"], b["verdict"], b["text"]):
        for general in (False, True):
            rc, o = host.parse_overlap(ln, general_only=general)
            if vv == 3:
                assert rc != 0
            elif vv in (1, 2):
                assert rc == 0 and o["line"] == t, ln
    f.close()


@pytest.mark.parametrize("k", range(len(GOLD["blocks"])))
def test_oracle_takes_the_references_decisions(k, reads, tmp_path, oracle):
    b = GOLD["blocks"][k]
    rs, _ = reads
    st = _settings(b)
    path, non = str(tmp_path / "ov.txt"), str(tmp_path / "non.txt")
    open(path, "w").write("\n".join(b["lines"]) + "\n")
    v = np.array(b["verdict"])
    rc, g, c = oracle.construct_edges(rs, st, path, non)
    assert rc == 0
    assert c.lines_read == len(b["lines"]) and c.malformed_lines == int((v == 3).sum())
    assert c.prefilter_rejected == int((v == 2).sum()) and c.scored == int((v == 1).sum())
    assert open(non).read() == "".join(t for t, vv in zip(b["text"], b["verdict"]) if vv == 2)
    # line by line where only a count tells "dropped" from "scored"
    for ln, vv in zip(b["lines"], b["verdict"]):
        if vv in (0, 1):
            open(path, "w").write(ln + "\n")
            rc, g, c = oracle.construct_edges(rs, st, path, non)
            assert rc == 0 and c.scored == vv and c.prefilter_rejected == 0 and c.malformed_lines == 0, ln
