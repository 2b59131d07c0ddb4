#!/usr/bin/env python3
"""BASELINE config 1 as a PROCESS: `hc-edgecalc` end to end — process start, FASTQ, HIP runtime start, contexts and
read store, construct_edges, sortEdges, output files — on the reference's SAVAGE example reads (tests/golden/savage_*.fastq.gz: the
whole savage/example/input_fas set, 2 000 merged singles of 400..490 bp + 200 2x250 pairs) with the argv list of
scripts/pipeline_per_stage.py:272-298 at SAVAGE's stage a and stage b/c values, beside the reference's own
construct_edges + sortEdges (fragment probe, oracle/_ref/libhcref_edgecalc_omp.so) on the same overlaps file.  The overlaps file comes
from the library's own finder + SFO ingest (rust-overlaps is not in the image), written once, outside every timed figure.

Every pipeline iteration starts ViralQuasispecies anew, so what a drop-in process pays per call is the whole wall time here.

    python tools/c1_process.py [--reps 5] > gpurun_out/r04_c1_process.json
"""
import argparse
import ctypes as C
import gzip
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def gunzip_to(name, dst):
    with gzip.open(os.path.join(ROOT, "tests", "golden", name + ".gz"), "rb") as f, open(dst, "wb") as o:
        o.write(f.read())
    return dst


def run_cli(argv, reps, env=None):
    walls, outs = [], []
    for _ in range(reps):
        t0 = time.perf_counter()
        r = subprocess.run(argv, capture_output=True, text=True, env=env)
        walls.append(time.perf_counter() - t0)
        if r.returncode != 0:
            raise SystemExit(f"hc-edgecalc failed ({r.returncode}): {r.stdout[-1500:]} {r.stderr[-1500:]}")
        outs.append(r.stdout)
    return walls, outs


def breakdown(stdout):
    """The verbose lines of one run -> seconds by phase."""
    b = {}
    m = re.search(r"FastqStorage ready! Construction took ([0-9.e+-]+) seconds", stdout)
    if m:
        b["fastq_s"] = float(m.group(1))
    m = re.search(r"HIP runtime start ([0-9.]+) s .*EdgeCalculator \(contexts, read store, text blocks\) ([0-9.]+) s", stdout)
    if m:
        b["hip_runtime_start_s"], b["edge_calculator_ctor_s"] = float(m.group(1)), float(m.group(2))
    m = re.search(r"(\d+) edges have been constructed in ([0-9.e+-]+) seconds", stdout)
    if m:
        b["edges"], b["construct_edges_s"] = int(m.group(1)), float(m.group(2))
    m = re.search(r"\[hc-edgecalc\] ([0-9.]+) s since main\(\) started", stdout)
    if m:
        b["main_total_s"] = float(m.group(1))
    return b


def reference_probe(reads, path, settings, pre, threads_list):
    lib = os.path.join(ROOT, "oracle", "_ref", "libhcref_edgecalc_omp.so")
    if not os.path.exists(lib):
        return None
    import bench

    ref = C.CDLL(lib)
    vp = C.c_void_p
    ref.frag_time_construct_edges.restype = C.c_int
    ref.frag_time_construct_edges.argtypes = [C.POINTER(bench._FragSettings), vp, C.c_uint64, vp, vp, vp, C.c_uint32, C.c_uint32, C.c_char_p, C.c_char_p,
                                              C.c_int, C.c_int, vp, vp, C.POINTER(C.c_uint64), vp]
    seqs, quals = zip(*(reads.seq(q) for q in range(reads.n_seq)))
    S, Q = (C.c_char_p * len(seqs))(*seqs), (C.c_char_p * len(quals))(*quals)
    ids = np.ascontiguousarray(reads.read_ids, dtype=np.uint64)
    n_single = sum(1 for r in range(reads.n_reads) if not reads.is_paired(r))
    fs = bench._FragSettings(settings["edge_threshold"], 0.9, settings.get("merge_contigs", 0.0), 0.0, 0, 1 if settings.get("ignore_inclusions") else 0)
    p3 = (C.c_uint32 * 3)(*pre)
    out = {}
    d = tempfile.mkdtemp(prefix="hcc1ref_")
    try:
        for t in threads_list:
            reps = 5
            cs, ss = np.zeros(reps, np.float64), np.zeros(reps, np.float64)
            edges, counters = C.c_uint64(), (C.c_uint32 * 3)()
            rc = ref.frag_time_construct_edges(C.byref(fs), p3, 10 ** 8, S, Q, ids.ctypes.data, n_single, reads.n_reads - n_single, path.encode(), d.encode(), t,
                                               reps, cs.ctypes.data, ss.ctypes.data, C.byref(edges), counters)
            assert rc == 0
            out[str(t)] = {"construct_edges_s_median": float(np.median(cs)), "sort_edges_s_median": float(np.median(ss)), "edges": int(edges.value)}
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    import haploconduct_amd as hc
    from haploconduct_amd import host

    d = tempfile.mkdtemp(prefix="hcc1_") + "/"
    try:
        s = gunzip_to("savage_singles.fastq", d + "singles.fastq")
        p1 = gunzip_to("savage_paired1.fastq", d + "paired1.fastq")
        p2 = gunzip_to("savage_paired2.fastq", d + "paired2.fastq")
        f = host.Fastq(singles=s, paired1=p1, paired2=p2)
        reads = f.readset()
        with hc.EdgeScorer(hc.Settings()) as sc:  # the candidates: what `rust-overlaps -i -r ... 0.02 100` + sfo2overlaps.py would write
            sc.set_reads(reads)
            sfo = sc.find_overlaps(0.02, 100)
        n_lines = host.sfo_records_to_overlaps(sfo, d + "overlaps.txt", f.n_single, f.n_paired)
        exe = os.path.join(ROOT, "haploconduct_amd", "csrc", "hc-edgecalc")
        def argv(o, edge_threshold, min_overlap_len, error_rate, remove_inclusions):
            # the list scripts/pipeline_per_stage.py:272-298 (run_first_it_noEC) hands to subprocess.check_call, flag for flag and in its
            # `--name=value` spelling, with SAVAGE's values for the stage; tests/golden/pipeline_argv.json holds the templates
            return [exe, f"--singles={s}", f"--paired1={p1}", f"--paired2={p2}", f"--overlaps={d}overlaps.txt", "--threads=8",
                    "--edge_threshold=%f" % edge_threshold, "--first_it=true", "--min_clique_size=2", "--keep_singletons=0", "--remove_branches=true",
                    "--min_overlap_perc=0", "--min_overlap_len=%d" % min_overlap_len, "--merge_contigs=%f" % error_rate, "--FNO=1",
                    "--original_readcount=%d" % (f.n_single + f.n_paired), "--error_correction=false", "--remove_trans=1", "--optimize=false",
                    "--verbose=true", "--diploid=false", f"--base_path={ROOT}", "--min_read_len=0", "--max_tip_len=150", "--separate_tips=false",
                    "--ignore_inclusions=%s" % remove_inclusions, f"--output={o}"]

        stages = {"stage_a": ((0.97, 200, 0.0, "false"), {"edge_threshold": 0.97}, (200, 0, 0)),
                  "stage_bc": ((0.995, 100, 0.01, "true"), {"edge_threshold": 0.995, "merge_contigs": 0.01, "ignore_inclusions": 1}, (100, 0, 0))}
        help_walls, _ = run_cli([exe, "--help"], 5)  # loads every library the program links (the HIP runtime's among them) and leaves: no HIP call
        out = {"process_start_and_exit_s": {"median": sorted(help_walls)[2], "min": min(help_walls),
                                            "what": "`hc-edgecalc --help`: exec + dynamic loading of libhcedge.so and the HIP runtime's libraries + exit, no HIP call"},
               "workload": f"savage/example/input_fas whole: {f.n_single} singles + {f.n_paired} pairs, {n_lines} overlap lines "
                           f"({os.path.getsize(d + 'overlaps.txt')} bytes) from the library's own finder + SFO ingest",
               "host": f"{os.cpu_count()} hardware threads", "reps": args.reps, "stages": {}}
        for name, (vals, rs, pre) in stages.items():
            o = d + name + "/"
            os.mkdir(o)
            flags = argv(o, *vals)[1:]
            walls, outs = run_cli(argv(o, *vals), args.reps)
            runs = [dict(breakdown(t), process_wall_s=w) for w, t in zip(walls, outs)]
            med = sorted(runs, key=lambda r: r["process_wall_s"])[(len(runs) - 1) // 2]
            rec = {"argv_flags": flags, "process_wall_s": {"median": med["process_wall_s"], "min": min(walls), "first": walls[0]},
                   "median_run": med, "runs": runs,
                   "outside_main_s": med["process_wall_s"] - med.get("main_total_s", 0.0),
                   "reference_construct_edges_probe": reference_probe(reads, d + "overlaps.txt", rs, pre, (1, 8, 32))}
            # one more run with the stage's own lap prints (stderr): what the constructor and construct_edges spend their time on
            tr = subprocess.run(argv(o, *vals), capture_output=True, text=True, env=dict(os.environ, HC_STAGE_TIMING="1"))
            rec["stage_timing_lines"] = [ln for ln in tr.stderr.splitlines() if ln.startswith("[hc stage]")][:40]
            ctor = med.get("edge_calculator_ctor_s", 0.0) + med.get("hip_runtime_start_s", 0.0)
            rec["what_a_resident_context_would_save_s"] = ctor
            rec["note"] = ("a process per call pays the HIP runtime's start and the context / store / text-block set-up (hip_runtime_start_s + "
                           "edge_calculator_ctor_s) every time; a stage kept open across the iterations of a pipeline (hc_ec_open once, "
                           "hc_ec_construct_edges[_from_reads] per iteration: include/hcedge_host.h) pays them once")
            out["stages"][name] = rec
        # The same calls through the RESIDENT process (round 5): `hc-edgecalc --resident <the same arguments>` — a thin client that forwards argv, cwd
        # and its stdout / stderr to a per-user process which keeps the HIP runtime, the code object and the library loaded.  A SAVAGE stage-b/c loop:
        # the first call starts the resident process, the later ones are what every further iteration of a pipeline pays.  Outputs byte for byte.
        renv = dict(os.environ, HC_RESIDENT_DIR=d + "resident", HC_RESIDENT_IDLE_S="120")
        names = ("edges.tsv", "edges_sorted.tsv", "nonedge_overlaps.txt", "edgecalc_stats.txt")
        res = {}
        try:
            for name, (vals, rs, pre) in stages.items():
                o, o2 = d + name + "/", d + name + "_resident/"
                os.mkdir(o2)
                for fn in names:  # the process-per-call outputs of a fresh directory, to compare with
                    if os.path.exists(o + fn):
                        os.remove(o + fn)
                run_cli(argv(o, *vals), 1)
                want = {fn: open(o + fn, "rb").read() for fn in names}
                a = argv(o2, *vals)
                walls = []
                for k in range(args.reps + 3):
                    for fn in names:
                        if os.path.exists(o2 + fn):
                            os.remove(o2 + fn)
                    w, outs = run_cli([a[0], "--resident"] + a[1:], 1, env=renv)
                    walls.append(w[0])
                    got = {fn: open(o2 + fn, "rb").read() for fn in names}
                    assert got == want, f"{name}: the resident process wrote other files than a process of its own (call {k})"
                    last_out = outs[0]
                later = sorted(walls[1:])
                res[name] = {"first_call_s": walls[0], "later_calls_s": walls[1:], "later_median_s": later[(len(later) - 1) // 2], "later_min_s": later[0],
                             "outputs_identical_to_a_process_per_call": True, "breakdown_of_the_last_call": breakdown(last_out)}
        finally:
            subprocess.run([exe, "--resident_stop"], env=renv)
        out["resident"] = dict(res, what="`hc-edgecalc --resident` + the stage's argv: wall time of the client process per call; first_call_s includes starting the "
                                         "resident process (the first stage's) — later calls are a pipeline's further iterations")
        print(json.dumps(out, indent=1))
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
