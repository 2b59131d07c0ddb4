"""`hc-edgecalc --resident` on the device: a loop of stage calls with CHANGING inputs and settings (another read set, another quality alphabet,
other thresholds, singles instead of pairs, a file with odd lines) through ONE resident process — whose contexts, text blocks and page-locked
buffers are taken over from call to call (hc_reset, keep_devices_resident) — must write, call for call, the bytes a process of its own writes."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "haploconduct_amd", "csrc", "hc-edgecalc")
FILES = ("edges.tsv", "edges_sorted.tsv", "nonedge_overlaps.txt", "edgecalc_stats.txt")


def _case(d, name, reads, cand, extra):
    from haploconduct_amd import host

    os.makedirs(d + name)
    p = d + name + "/"
    host.write_overlaps(p + "overlaps.txt", cand, reads)
    paired = reads.is_paired(reads.n_reads - 1)
    if paired:
        reads.write_fastq(None, p + "p1.fastq", p + "p2.fastq")
        inputs = ["--paired1", p + "p1.fastq", "--paired2", p + "p2.fastq"]
    else:
        reads.write_fastq(p + "s.fastq", None, None)
        inputs = ["--singles", p + "s.fastq"]
    return inputs + ["--overlaps", p + "overlaps.txt", "--original_readcount", str(reads.n_reads), "--threads", "8", "--verbose", "true"] + extra


def test_a_pipeline_of_changing_stages_through_one_resident_process(tmp_path):
    from haploconduct_amd import synth

    d = str(tmp_path) + "/"
    env = dict(os.environ, HC_RESIDENT_DIR=d + "res", HC_RESIDENT_IDLE_S="60")
    cases = []
    r1, m1 = synth.make_paired_dataset(4000, 6000, flip_frac=0.25, seed=21)
    cases.append(_case(d, "a", r1, synth.paired_candidates(m1, n_candidates=100000, seed=22), ["--edge_threshold", "0.97", "--min_overlap_len", "150"]))
    r2, m2 = synth.make_single_dataset(3000, 9000, len_lo=150, len_hi=900, n_strains=3, divergence=0.01, flip_frac=0.5, seed=5, log_uniform=True)
    cases.append(_case(d, "b", r2, synth.single_candidates(m2, min_overlap=100, n_candidates=150000),
                       ["--edge_threshold", "0.995", "--min_overlap_len", "100", "--merge_contigs", "0.01", "--ignore_inclusions", "true"]))
    quals = (np.arange(1, 36) + 33).astype(np.uint8)  # 35 quality values: the wide table, another kernel
    r3, m3 = synth.make_single_dataset(4000, 20000, len_lo=250, len_hi=250, n_strains=2, divergence=0.001, flip_frac=0.5, seed=4, quals=quals)
    cases.append(_case(d, "c", r3, synth.single_candidates(m3, min_overlap=127), ["--edge_threshold", "1", "--min_overlap_len", "127"]))
    cases.append(cases[0][:-4] + ["--edge_threshold", "0.9", "--min_overlap_len", "100", "--max_ov", "61234"])  # the first inputs again, other settings
    try:
        for rnd in range(2):  # the whole loop twice: every case also FOLLOWS every other kind of case
            for k, args in enumerate(cases):
                outs = {}
                for mode in ("own", "resident"):
                    o = f"{d}out_{rnd}_{k}_{mode}/"
                    os.makedirs(o)
                    r = subprocess.run([EXE] + (["--resident"] if mode == "resident" else []) + args + ["--output", o], env=env, capture_output=True, text=True,
                                       timeout=600)
                    assert r.returncode == 0, (mode, k, r.stdout[-1500:], r.stderr[-1500:])
                    outs[mode] = ({f: open(o + f, "rb").read() for f in FILES}, [ln for ln in r.stdout.splitlines() if "edges have been constructed" in ln])
                assert outs["own"][0] == outs["resident"][0], f"round {rnd}, case {k}: the resident process wrote other files"
                assert len(outs["own"][0]["edges_sorted.tsv"]) > 10000
                assert [ln.split(" in ")[0] for ln in outs["own"][1]] == [ln.split(" in ")[0] for ln in outs["resident"][1]]
        # the HC_* environment of a job is its CLIENT's, call by call: HC_STAGE_TIMING (the stage's lap prints on stderr) with, without, with
        o = d + "env_probe/"
        os.makedirs(o)
        laps = []
        subprocess.run([EXE, "--resident_stop"], env=env, timeout=60)  # the next call STARTS a resident process — with the variable in its environment
        for e in (dict(env, HC_STAGE_TIMING="1"), env, dict(env, HC_STAGE_TIMING="1")):
            for fn in FILES:
                if os.path.exists(o + fn):
                    os.remove(o + fn)
            r = subprocess.run([EXE, "--resident"] + cases[0] + ["--output", o], env=e, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0
            laps.append("[hc stage]" in r.stderr)
        assert laps == [True, False, True], laps
        # a failing job (a FASTQ file that is not there) answers 1 and the resident process goes on
        bad = subprocess.run([EXE, "--resident", "--singles", d + "nope.fastq", "--overlaps", d + "a/overlaps.txt", "--original_readcount", "3", "--output", d], env=env,
                             capture_output=True, text=True, timeout=120)
        assert bad.returncode == 1 and "nope.fastq" in bad.stderr
        o = d + "after_failure/"
        os.makedirs(o)
        ok = subprocess.run([EXE, "--resident"] + cases[1] + ["--output", o], env=env, capture_output=True, text=True, timeout=600)
        assert ok.returncode == 0 and open(o + "edges.tsv", "rb").read() == open(f"{d}out_1_1_own/edges.tsv", "rb").read()
    finally:
        subprocess.run([EXE, "--resident_stop"], env=env, timeout=60)


def test_stages_of_one_process_take_over_each_others_devices(tmp_path):
    """hc_ec_keep_devices (what the resident process does, for a caller that opens stage after stage itself): with it on, a closed stage's
    contexts and text blocks serve the next one — other reads, other settings, another kernel — and every graph equals the one a stage with
    devices of its own builds; opening gets cheaper from the second stage on."""
    import time

    import haploconduct_amd as hc
    from haploconduct_amd import host, synth

    d = str(tmp_path) + "/"
    cases = []
    r1, m1 = synth.make_paired_dataset(4000, 6000, flip_frac=0.25, seed=21)
    cases.append((r1, synth.paired_candidates(m1, n_candidates=100000, seed=22), hc.Settings(edge_threshold=0.97, min_overlap_len=150)))
    r2, m2 = synth.make_single_dataset(3000, 9000, len_lo=150, len_hi=900, n_strains=3, divergence=0.01, flip_frac=0.5, seed=5, log_uniform=True)
    cases.append((r2, synth.single_candidates(m2, min_overlap=100, n_candidates=150000), hc.Settings(edge_threshold=0.995, min_overlap_len=100, merge_contigs=0.01)))
    cases.append((r1, cases[0][1][:60000], hc.Settings(edge_threshold=0.9, min_overlap_len=100)))
    files = []
    for k, (reads, cand, st) in enumerate(cases):
        p = f"{d}c{k}/"
        os.makedirs(p)
        host.write_overlaps(p + "overlaps.txt", cand, reads)
        if reads.is_paired(0):
            reads.write_fastq(None, p + "p1.fastq", p + "p2.fastq")
            kw = dict(paired1=p + "p1.fastq", paired2=p + "p2.fastq")
        else:
            reads.write_fastq(p + "s.fastq", None, None)
            kw = dict(singles=p + "s.fastq")
        st.n_threads = 8
        files.append((st, dict(kw, overlaps=p + "overlaps.txt", output_dir=p)))

    def run_all():
        out, opens = [], []
        for st, kw in files * 2:
            if os.path.exists(kw["output_dir"] + "nonedge_overlaps.txt"):
                os.remove(kw["output_dir"] + "nonedge_overlaps.txt")
            t0 = time.perf_counter()
            ec = host.EdgeCalculatorStage(st, **kw)
            opens.append(time.perf_counter() - t0)
            ec.construct_edges_sorted()
            out.append((ec.edges().tobytes(), ec.counters(), open(kw["output_dir"] + "nonedge_overlaps.txt", "rb").read()))
            ec.close()
        return out, opens

    own, opens_own = run_all()
    host.keep_devices(True)
    try:
        kept, opens_kept = run_all()
    finally:
        host.keep_devices(False)
    for k, (a, b) in enumerate(zip(own, kept)):
        assert a[0] == b[0] and a[2] == b[2], f"stage {k}: another graph on taken-over devices"
        for key in ("edges_added", "nonedges_written", "scored", "lines_read", "dup_count", "inclusion_count"):
            assert a[1][key] == b[1][key], (k, key)
    assert len(own[0][0]) > 100000
    # (opening on taken-over devices is the cheaper one — contexts, streams, page-locked text buffers exist already —, but in a warm process
    # both are a few milliseconds at this size and the box's noise decides single runs: no bar on the clock here; profiles/r05_c1_process.json
    # has the resident process's figures)
    print("open_s with devices of their own:", [round(x, 4) for x in opens_own], "taken over:", [round(x, 4) for x in opens_kept])
