#!/usr/bin/env python3
"""FASTQ -> overlap graph with nothing but this library (the front of SAVAGE stage a, savage.py:643-700 + the
ViralQuasispecies edge calculation): read store, hc_find_overlaps (instead of rust-overlaps), hc_sfo2overlaps
(instead of scripts/sfo2overlaps.py), edge-calculation stage.  Prints the time of every step."""
import argparse
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--err", type=float, default=0.0)
    ap.add_argument("--min-overlap", type=int, default=90)
    ap.add_argument("--threads", type=int, default=32)
    ap.add_argument("--host-ingest", action="store_true", help="fetch the SFO records and run the whole ingest on the host threads")
    ap.add_argument("--one-call", action="store_true", help="hc_ec_construct_edges_from_reads: one stage, no overlaps file, the FASTQ files read once")
    a = ap.parse_args()
    import bench
    import haploconduct_amd as hc
    from haploconduct_amd import host

    reads, cand, cfg, st = bench.build_workload(a.workload, 0)
    del cand
    st.n_threads = a.threads
    d = tempfile.mkdtemp(prefix="hcr2g_") + "/"
    paired = reads.is_paired(0)
    n_single = 0 if paired else reads.n_reads
    n_pairs = reads.n_reads if paired else 0
    reads.write_fastq(None if paired else d + "singles.fastq", d + "p1.fastq" if paired else None, d + "p2.fastq" if paired else None)
    t = {}
    if a.one_call:
        kw = dict(singles=None if paired else d + "singles.fastq", paired1=d + "p1.fastq" if paired else None,
                  paired2=d + "p2.fastq" if paired else None, output_dir=d)
        runs = []
        for _ in range(3):  # a process's first call pays the runtime's start and the finder's allocations
            t0 = time.perf_counter()
            with host.EdgeCalculatorStage(st, **kw) as ec:
                t_open = time.perf_counter() - t0
                t1 = time.perf_counter()
                n_recs, n_lines = ec.construct_edges_from_reads(a.err, a.min_overlap)
                t_call = time.perf_counter() - t1
                n_edges = ec.edge_count()
            runs.append({"open_s": round(t_open, 4), "construct_edges_from_reads_s": round(t_call, 4), "total_s": round(t_open + t_call, 4)})
        print(json.dumps({"workload": cfg["workload"], "route": "one call (hc_ec_construct_edges_from_reads): FASTQ -> sorted graph, no overlaps file",
                          "sequences": int(reads.n_seq), "sfo_records": int(n_recs), "overlap_lines": int(n_lines), "edges": int(n_edges), "runs": runs,
                          "best_total_s": min(r["total_s"] for r in runs)}))
        return
    t0 = time.perf_counter()
    with hc.EdgeScorer(st) as sc:
        sc.set_reads(reads)
        t["store"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        if a.host_ingest:  # the records fetched, flipped and sorted by the host threads
            recs = sc.find_overlaps(a.err, a.min_overlap)
            n_recs = recs.size
            t["find_overlaps"] = time.perf_counter() - t0
            t0 = time.perf_counter()
            n_lines = host.sfo_records_to_overlaps(recs, d + "overlaps.txt", n_single, n_pairs)  # == write_sfo + sfo2overlaps
            t["sfo_records_to_overlaps"] = time.perf_counter() - t0
        else:  # the records stay on the device: flip + sort there, the sorted records come back once for the matching
            n_recs = sc.find_overlaps(a.err, a.min_overlap, count_only=True)
            t["find_overlaps"] = time.perf_counter() - t0
            t0 = time.perf_counter()
            n_lines = sc.found_to_overlaps(d + "overlaps.txt", n_single, n_pairs)
            t["found_to_overlaps"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    kw = dict(singles=None if paired else d + "singles.fastq", paired1=d + "p1.fastq" if paired else None,
              paired2=d + "p2.fastq" if paired else None, overlaps=d + "overlaps.txt", output_dir=d)
    with host.EdgeCalculatorStage(st, **kw) as ec:
        t["stage_open"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        ec.construct_edges()
        t["construct_edges"] = time.perf_counter() - t0
        edges = ec.edge_count()
    print(json.dumps({"workload": cfg.get("workload", a.workload), "sequences": int(reads.n_seq), "sfo_records": int(n_recs),
                      "overlap_lines": int(n_lines), "edges": int(edges), "seconds": {k: round(v, 4) for k, v in t.items()},
                      "total_s": round(sum(t.values()), 3)}))
    import shutil
    shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
