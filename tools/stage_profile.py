#!/usr/bin/env python3
"""Only the stage (text overlaps file + FASTQ -> sorted graph), `--reps` files of one workload, nothing else on the device: the program
a rocprofv3 kernel trace of the STAGE is taken of (bench.py's own launches of the scoring kernel would share its symbol).
    rocprofv3 --kernel-trace --stats -d <dir> -- python3 tools/stage_profile.py --workload c3 --reps 4
prints one JSON line (the runs as in bench.py's stage_end_to_end).  `--summarize <dir> --reps N` reads the trace's kernel stats and
prints the device time PER FILE by kernel (opening the stage — store upload, id table — is listed apart)."""
import argparse
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def summarize(d, reps):
    f = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    if not f:
        raise SystemExit("no kernel_stats.csv under " + d)
    rows = list(csv.DictReader(open(f[0])))
    per_block = ("text_", "score_kernel", "kept_", "scan_tiles", "flush_rows", "bucket", "sink_")
    total = blocks = 0.0
    out = []
    for r in rows:
        ns, calls = float(r["TotalDurationNs"]), int(r["Calls"])
        name = r["Name"].split("(")[0].replace("void ", "")
        ms = ns / 1e6 / reps
        total += ms
        if any(k in name for k in per_block):
            blocks += ms
        out.append((ms, calls / reps, float(r["AverageNs"]) / 1e3, name))
    out.sort(reverse=True)
    print("device time per file (%d files traced): %.2f ms; of it the per-block launches (lines, parse, scoring, rows in order, flush): %.2f ms" % (reps, total, blocks))
    print("%10s %10s %10s  %s" % ("ms/file", "calls/file", "avg us", "kernel"))
    for ms, calls, avg, name in out:
        if ms >= 0.02:
            print("%10.3f %10.1f %10.2f  %s" % (ms, calls, avg, name[:150]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--summarize", default=None)
    a = ap.parse_args()
    if a.summarize:
        return summarize(a.summarize, a.reps)
    import bench

    reads, cand, cfg, st = bench.build_workload(a.workload, 0)
    threads = a.threads or min(32, os.cpu_count() or 1)
    r = bench.stage_end_to_end(reads, cand, st, threads, reps=a.reps)
    print(json.dumps({"workload": cfg["workload"], "stage": r}))


if __name__ == "__main__":
    main()
