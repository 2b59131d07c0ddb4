"""Per-LINE fallback of the device's text parser (round 5; the reference reads every line by itself, src/EdgeCalculator.cpp:581-604).
An overlaps file of BASELINE config 2 in which one line in 10^4 is not of the plain form — padded with the blanks and tabs the reference
trims (:584), an id written in octal or hexadecimal (Overlap's constructor reads ids with strtoul(.., 0), src/Overlap.h:44-45), a line with
other than 13 fields (:598-603) — must give the graph the REFERENCE'S OWN construct_edges + sortEdges build from the same file
(tests/_refstage.py), with those lines — and only those — read by the host's tokeniser, and in about the time of the clean file."""
import copy
import json
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _odd_file(clean_path, odd_path, every=10000):
    """Every `every`-th line of the clean file in one of four odd forms (and a junk line in front of every fourth of them)."""
    n_odd = n_junk = 0
    with open(clean_path) as src, open(odd_path, "w") as dst:
        for k, line in enumerate(src):
            if k % every == every // 2:
                f = line.rstrip("\n").split("\t")
                kind = (k // every) % 4
                if kind == 0:
                    line = " \t" + "\t".join(f) + "\t  \n"  # padding on both sides (:584 trims it)
                elif kind == 1:
                    f[0] = "0" + oct(int(f[0]))[2:]  # the same id in octal: strtoul(.., 0)
                    line = "\t".join(f) + "\n"
                elif kind == 2:
                    f[1] = hex(int(f[1]))  # ... and in hexadecimal
                    line = "\t".join(f) + "\n"
                else:
                    dst.write("\t".join(f[:12]) + "\n")  # 12 fields: "incorrect overlap; skipping" (:598-603); then the line itself, padded
                    n_junk += 1
                    line = "\t".join(f) + " \n"
                n_odd += 1
            dst.write(line)
    return n_odd + n_junk, n_junk


def _stage_run(st, overlaps, out_dir, fastq_kw, env=None):
    from haploconduct_amd import host

    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        os.makedirs(out_dir, exist_ok=True)
        if os.path.exists(out_dir + "nonedge_overlaps.txt"):
            os.remove(out_dir + "nonedge_overlaps.txt")
        with host.EdgeCalculatorStage(st, overlaps=overlaps, output_dir=out_dir, **fastq_kw) as ec:
            t0 = time.perf_counter()
            ec.construct_edges_sorted()
            dt = time.perf_counter() - t0
            return ec.edges(), ec.counters(), open(out_dir + "nonedge_overlaps.txt", "rb").read(), dt
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_c2_file_with_odd_lines_equals_the_references_own_stage_and_costs_no_more(tmp_path):
    import bench
    from haploconduct_amd import host
    from tests._refstage import whole_file_against_the_references_own_stage

    reads, cand, cfg, st = bench.build_workload("c2", 0)
    d = str(tmp_path) + "/"
    host.write_overlaps(d + "clean.txt", cand, reads)
    reads.write_fastq(None, d + "p1.fastq", d + "p2.fastq")
    fastq_kw = dict(paired1=d + "p1.fastq", paired2=d + "p2.fastq")
    n_host_lines, n_junk = _odd_file(d + "clean.txt", d + "odd.txt")
    n_lines = int(cand.size) + n_junk
    assert n_host_lines >= 200
    # (1) the reference's own construct_edges + sortEdges on the odd file = ours (graph, in-lists, inclusions, non-edge file, counters)
    whole_file_against_the_references_own_stage(reads, st, d, d + "odd.txt", n_lines, int(cand.size), fastq_kw, min_edges=50000)
    # (2) who read what: every block stayed on the device, exactly the odd lines went through the host's tokeniser; the routes agree
    st2 = copy.copy(st)
    st2.max_overlaps = n_lines
    st2.n_threads = min(32, os.cpu_count() or 1)
    e_line, c_line, ne_line, _ = _stage_run(st2, d + "odd.txt", d + "o_line/", fastq_kw)
    assert c_line["host_lines"] == n_host_lines and c_line["host_blocks"] == 0 and c_line["device_blocks"] >= 4
    assert c_line["malformed_lines"] == n_junk and c_line["lines_read"] == n_lines
    e_block, c_block, ne_block, _ = _stage_run(st2, d + "odd.txt", d + "o_block/", fastq_kw, env={"HC_PARSE_FALLBACK": "block"})
    assert c_block["host_lines"] == 0 and c_block["host_blocks"] >= 4, "round 4's route: every block holds an odd line and goes to the host"
    e_clean, c_clean, ne_clean, _ = _stage_run(st2, d + "clean.txt", d + "o_clean/", fastq_kw)
    for other, what in ((e_block, "whole-block fallback"), (e_clean, "the clean file")):
        assert e_line.tobytes() == other.tobytes(), f"per-line fallback and {what} build different graphs"
    assert ne_line == ne_block == ne_clean
    for k in ("edges_added", "nonedges_written", "prefilter_rejected", "scored", "silently_dropped"):
        assert c_line[k] == c_block[k] == c_clean[k], k
    # a list that is too short sends the block to the host as before
    e_short, c_short, _, _ = _stage_run(st2, d + "odd.txt", d + "o_short/", fastq_kw, env={"HC_PARSE_FALLBACK": "8"})
    assert e_short.tobytes() == e_line.tobytes() and c_short["host_blocks"] >= 4
    # (3) time: the odd file within 1.2 x of the clean one (medians of 7; + 1 ms: the whole stage is a few ms at this size)
    runs = {"clean": [], "odd": [], "odd_whole_block_fallback": []}
    for _ in range(7):
        runs["clean"].append(_stage_run(st2, d + "clean.txt", d + "t/", fastq_kw)[3])
        runs["odd"].append(_stage_run(st2, d + "odd.txt", d + "t/", fastq_kw)[3])
        runs["odd_whole_block_fallback"].append(_stage_run(st2, d + "odd.txt", d + "t/", fastq_kw, env={"HC_PARSE_FALLBACK": "block"})[3])
    med = {k: float(np.median(v)) for k, v in runs.items()}
    rec = {"workload": "c2 (2 000 000 lines, 81 MB)", "odd_lines": n_host_lines, "construct_edges_sorted_s_median_of_7": med,
           "odd_over_clean": med["odd"] / med["clean"], "whole_block_over_clean": med["odd_whole_block_fallback"] / med["clean"], "runs": runs}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r05_odd_lines.json"), "w") as f:
        json.dump(rec, f, indent=1)
    assert med["odd"] <= 1.2 * med["clean"] + 5e-4, rec


def test_odd_lines_errors_are_the_host_parsers(tmp_path):
    """A line the device does not read and whose Overlap the reference's constructor refuses (a percentage above 100: src/Overlap.h's
    assert) fails the stage exactly as the host-parsed route does; an odd line naming an unknown read fails as std::map::at would."""
    import haploconduct_amd as hc
    from haploconduct_amd import host, synth

    reads, meta = synth.make_paired_dataset(600, 1500, seed=3)
    cand = synth.paired_candidates(meta, n_candidates=3000, seed=4)
    d = str(tmp_path) + "/"
    host.write_overlaps(d + "clean.txt", cand, reads)
    reads.write_fastq(None, d + "p1.fastq", d + "p2.fastq")
    lines = open(d + "clean.txt").read().split("\n")
    st = hc.Settings(edge_threshold=0.97, min_overlap_len=150)
    st.n_threads = 4
    fastq_kw = dict(paired1=d + "p1.fastq", paired2=d + "p2.fastq")
    for name, edit in (("unknown_id", lambda f: [" " + f[0]] + ["999999"] + f[2:]), ("bad_ori", lambda f: [" " + f[0]] + f[1:5] + ["x"] + f[6:])):
        bad = list(lines)
        bad[1500] = "\t".join(edit(bad[1500].split("\t")))
        open(d + name + ".txt", "w").write("\n".join(bad))
        msgs = []
        for env in ({}, {"HC_PARSE": "host"}):
            with pytest.raises(hc.HcError) as ei:
                _stage_run(st, d + name + ".txt", d + "o_" + name + "/", fastq_kw, env=env)
            msgs.append(str(ei.value))
        assert msgs[0] == msgs[1], name


def _small_case(tmp_path, n_pairs=1200, n_cand=20000):
    import haploconduct_amd as hc
    from haploconduct_amd import host, synth

    reads, meta = synth.make_paired_dataset(n_pairs, 2500, flip_frac=0.2, seed=41)
    cand = synth.paired_candidates(meta, n_candidates=n_cand, seed=42)
    d = str(tmp_path) + "/"
    host.write_overlaps(d + "clean.txt", cand, reads)
    reads.write_fastq(None, d + "p1.fastq", d + "p2.fastq")
    st = hc.Settings(edge_threshold=0.97, min_overlap_len=150)
    st.n_threads = 4
    return d, open(d + "clean.txt").read().split("\n")[:-1], st, dict(paired1=d + "p1.fastq", paired2=d + "p2.fastq")


def _routes_agree(st, path, d, fastq_kw, tag, env=None):
    """The per-line route, round 4's whole-block route and the host-parsed route: one graph, one non-edge file, the same counters."""
    got = {}
    for name, e in (("line", {}), ("block", {"HC_PARSE_FALLBACK": "block"}), ("host", {"HC_PARSE": "host"})):
        got[name] = _stage_run(st, path, f"{d}{tag}_{name}/", fastq_kw, env=dict(env or {}, **e))
    for name in ("block", "host"):
        assert got["line"][0].tobytes() == got[name][0].tobytes(), f"{tag}: per-line and {name} routes build different graphs"
        assert got["line"][2] == got[name][2], f"{tag}: nonedge_overlaps.txt differs ({name})"
        for k in ("edges_added", "nonedges_written", "prefilter_rejected", "malformed_lines", "lines_read", "scored", "silently_dropped", "self_overlap_count",
                  "inclusion_count", "dup_count"):
            assert got["line"][1][k] == got[name][1][k], f"{tag}: {k} ({name})"
    return got["line"]


def test_max_ov_cuts_between_and_on_odd_lines(tmp_path):
    """`&& i < max_overlaps` (:581) counts odd lines like any other: cuts right before, on and right behind an odd line, and behind a junk line."""
    d, lines, st, fastq_kw = _small_case(tmp_path)
    odd = list(lines)
    for k in (100, 101, 5000, 12345):
        odd[k] = " " + odd[k] + "\t"
    odd.insert(7000, "not\tan\toverlap")
    open(d + "odd.txt", "w").write("\n".join(odd) + "\n")
    for cut in (100, 101, 102, 5001, 7000, 7001, 7002, 12346, len(odd)):
        st2 = copy.copy(st)
        st2.max_overlaps = cut
        e, c, _, _ = _routes_agree(st2, d + "odd.txt", d, fastq_kw, f"cut{cut}")
        assert c["lines_read"] == cut and c["host_lines"] == sum(1 for k in (100, 101, 5000, 7000, 12346) if k < cut)


def test_odd_last_line_without_a_newline_and_odd_lines_across_small_blocks(tmp_path):
    """A file whose LAST line is odd and has no newline (std::getline still reads it); blocks of 64 KiB so that odd lines sit first and last in
    blocks; a rejected odd line (prefilter, :633-635) lands in nonedge_overlaps.txt at its place among the device's rejects."""
    d, lines, st, fastq_kw = _small_case(tmp_path)
    odd = list(lines)
    rng = np.random.default_rng(9)
    for k in rng.choice(len(odd), 300, replace=False):
        f = odd[k].split("\t")
        if k % 3 == 0:  # a line the prefilter rejects (LEN below 0.5 M for both mates), padded
            f[9], f[10] = "20", "30"
        odd[k] = "\t".join(f) + "  "
    for k in rng.choice(len(odd), 300, replace=False):  # plain rejected lines too: the two kinds interleave in the rejects' file
        f = odd[k].split("\t")
        if not odd[k].endswith(" "):
            f[9], f[10] = "21", "31"
            odd[k] = "\t".join(f)
    odd[-1] = "\t" + odd[-1]
    open(d + "odd.txt", "w").write("\n".join(odd))  # no trailing newline
    e, c, ne, _ = _routes_agree(st, d + "odd.txt", d, fastq_kw, "tail", env={"HC_TEXT_BLOCK": "65536"})
    assert c["host_lines"] >= 300 and c["host_blocks"] == 0 and c["device_blocks"] >= 10 and c["prefilter_rejected"] >= 300
    assert c["lines_read"] == len(odd)


def test_every_line_odd_falls_back_block_by_block(tmp_path):
    """--allow_spaced_overlaps input (blanks between the fields): no line is plain, the blocks' lists overflow, every block goes to the host's
    tokeniser as a whole — the same graph as the tab-separated file."""
    d, lines, st, fastq_kw = _small_case(tmp_path, n_cand=12000)
    open(d + "spaced.txt", "w").write("\n".join(ln.replace("\t", " \t ") for ln in lines) + "\n")
    st2 = copy.copy(st)
    from haploconduct_amd.records import FLAG_ALLOW_SPACES

    st2.flags |= FLAG_ALLOW_SPACES
    e_sp, c_sp, ne_sp, _ = _stage_run(st2, d + "spaced.txt", d + "sp/", fastq_kw, env={"HC_PARSE_FALLBACK": "64"})
    e_cl, c_cl, ne_cl, _ = _stage_run(st, d + "clean.txt", d + "cl/", fastq_kw)
    assert e_sp.tobytes() == e_cl.tobytes() and ne_sp == ne_cl
    assert c_sp["host_blocks"] >= 1 and c_sp["device_blocks"] == 0 and c_sp["host_lines"] == 0
    e_ln, c_ln, _, _ = _stage_run(st2, d + "spaced.txt", d + "ln/", fastq_kw, env={"HC_PARSE_FALLBACK": "1000000"})  # a list long enough: line by line
    assert e_ln.tobytes() == e_cl.tobytes() and c_ln["host_lines"] == len(lines) and c_ln["host_blocks"] == 0
