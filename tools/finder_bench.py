#!/usr/bin/env python3
"""Times hc_find_overlaps (candidate generation on the device) on the read sets of the bench
workloads and checks a sample of the records on the host.  One JSON line per run."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--err", type=float, default=0.0, help="rust-overlaps err_rate (SAVAGE stage a passes 1/200: exact matching below 200 bp; de novo / POLYTE 0.02)")
    ap.add_argument("--min-overlap", type=int, default=90, help="rust-overlaps threshold (SAVAGE: 60 %% of the read length)")
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    import bench
    import haploconduct_amd as hc

    reads, cand, cfg, st = bench.build_workload(a.workload, 0)
    del cand
    comp = np.zeros(256, np.uint8)
    comp[list(b"ACGTN")] = list(b"TGCAN")
    with hc.EdgeScorer(st) as sc:
        t0 = time.perf_counter()
        sc.set_reads(reads)
        t_store = time.perf_counter() - t0
        times = []
        for _ in range(a.reps):  # device work only: index, seeds, expand, sort/unique, verify, compaction
            t0 = time.perf_counter()
            n_found = sc.find_overlaps(a.err, a.min_overlap, count_only=True)
            times.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        recs = sc.find_overlaps(a.err, a.min_overlap)  # recomputes once more and copies the records to the host
        t_fetch = time.perf_counter() - t0
        assert recs.size == n_found
    # host check of a sample: each record is what it says
    n_single = sum(1 for r in range(reads.n_reads) if not reads.is_paired(r))
    n_pairs = reads.n_reads - n_single

    def seq_of(sfo):
        if sfo < n_single:
            q = int(reads.read_first_seq[sfo])
        elif sfo < n_single + n_pairs:
            q = int(reads.read_first_seq[sfo])
        else:
            q = int(reads.read_first_seq[sfo - n_pairs]) + 1
        return np.frombuffer(reads.seq(q)[0], np.uint8)

    rng = np.random.default_rng(1)
    acgt = np.zeros(256, bool)
    acgt[list(b"ACGT")] = True
    for r in recs[rng.integers(0, recs.size, 2000)] if recs.size else []:
        A, B = seq_of(int(r["idA"])), seq_of(int(r["idB"]))
        if r["inverted"]:
            B = comp[B][::-1]
        d = int(r["OHA"])
        s, e = max(0, d), min(A.size, d + B.size)
        x, y = A[s:e], B[s - d:e - d]
        k_host = int(np.count_nonzero((x != y) | ~acgt[x]))  # a non-ACGT symbol matches nothing
        assert e - s == r["OLA"] >= a.min_overlap and k_host == r["K"] <= int(a.err * (e - s)), (r, k_host)
    best = min(times)
    print(json.dumps({"workload": cfg.get("workload", a.workload), "sequences": int(reads.n_seq), "bases": int(reads.bases.size),
                      "err_rate": a.err, "min_overlap": a.min_overlap, "overlaps_found": int(recs.size),
                      "inverted": int(recs["inverted"].sum()), "seconds": [round(t, 4) for t in times], "best_s": round(best, 4),
                      "sequences_per_s": round(reads.n_seq / best), "overlaps_per_s": round(recs.size / best), "store_build_s": round(t_store, 3), "compute_and_fetch_s": round(t_fetch, 3),
                      "sample_checked_on_host": 2000 if recs.size else 0}))


if __name__ == "__main__":
    main()
