#!/usr/bin/env python3
"""End-to-end timing of the edge-calculation stage (text overlaps file + FASTQ in -> populated
OverlapGraph + nonedge_overlaps.txt out) through the host mirror, with its parse / score / insert /
write breakdown, next to the oracle's construct_edges on the same files (1 thread) for reference."""
import argparse
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--threads", type=int, default=16, help="--threads of the stage (parser workers)")
    ap.add_argument("--oracle-lines", type=int, default=200000, help="lines of the file the 1-thread oracle is timed on")
    ap.add_argument("--reps", type=int, default=2)
    args = ap.parse_args()
    import bench
    import haploconduct_amd as hc
    from haploconduct_amd import host, synth
    from tests import _oracle

    reads, cand, cfg, st = bench.build_workload(args.workload, 0)
    st.n_threads = args.threads
    d = tempfile.mkdtemp(prefix="hcstage_") + "/"
    t0 = time.time()
    host.write_overlaps(d + "overlaps.txt", cand, reads)
    n_lines = int(cand.size)
    paired = reads.is_paired(0)
    reads.write_fastq(None if paired else d + "singles.fastq", d + "paired1.fastq" if paired else None,
                      d + "paired2.fastq" if paired else None)
    print(f"files written in {time.time()-t0:.1f} s: {os.path.getsize(d+'overlaps.txt')/1e6:.1f} MB overlaps, {n_lines} lines")
    try:  # the host side shares its cores with whatever else runs on the machine: say how busy it is
        print(f"host: {os.cpu_count()} hardware threads, load average {open('/proc/loadavg').read().split()[0]} before the runs")
    except OSError:
        pass
    kw = dict(singles=None if paired else d + "singles.fastq", paired1=d + "paired1.fastq" if paired else None,
              paired2=d + "paired2.fastq" if paired else None, overlaps=d + "overlaps.txt", output_dir=d)
    for rep in range(args.reps):
        if os.path.exists(d + "nonedge_overlaps.txt"):
            os.remove(d + "nonedge_overlaps.txt")
        t0 = time.time()
        ec = host.EdgeCalculatorStage(st, **kw)
        t1 = time.time()
        ec.construct_edges()
        t2 = time.time()
        c = ec.counters()
        print(f"run {rep}: open (FASTQ + store upload) {t1-t0:.3f} s, construct_edges {t2-t1:.3f} s -> "
              f"{c['scored']/(t2-t1)/1e6:.2f} M candidates/s end-to-end; parse {c['t_parse']:.3f} score {c['t_score']:.3f} "
              f"insert {c['t_insert']:.3f} write {c['t_write']:.3f}; edges {ec.edge_count()} dups {c['dup_count']} "
              f"nonedges {c['nonedges_written']}")
        t3 = time.time()
        ec.sort_edges()  # the next call of every assembly iteration (src/ViralQuasispecies.cpp:297)
        print(f"        sortEdges {time.time()-t3:.3f} s")
        ec.close()
    n = min(args.oracle_lines, n_lines)
    host.write_overlaps(d + "head.txt", cand[:n], reads)
    t0 = time.time()
    rc, g, oc = _oracle.construct_edges(reads, st, d + "head.txt", d + "ref_nonedge.txt")
    dt = time.time() - t0
    print(f"oracle construct_edges (1 thread) on the first {n} lines: {dt:.2f} s -> {oc.scored/dt/1e3:.1f} k candidates/s")
    import shutil
    shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
