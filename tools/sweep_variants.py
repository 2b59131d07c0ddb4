#!/usr/bin/env python3
"""In-process A/B of the scoring-kernel variants (HC_SCORE_VARIANT 0..7): interleaved rounds,
median and min kernel time per variant (methodology: one process, same data, same device)."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--variants", default="0,1,2,3,4,5,6,7")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--order", default="sfo")
    ap.add_argument("--reorder", type=int, default=0, help="hc_set_reorder mode for the device entry (0 never, 1 always)")
    args = ap.parse_args()
    import torch

    import bench
    import haploconduct_amd as hc

    reads, cand, cfg, st = bench.build_workload(args.workload, 0)
    if args.order == "grouped":
        cand = cand[np.argsort(cand["read1"], kind="stable")]
    elif args.order == "shuffled":
        cand = cand[np.random.default_rng(5).permutation(cand.size)]
    elif args.order.startswith("sfo-w"):
        # experiment: inside windows of W consecutive candidates, order by overlap length (longest first)
        W = int(args.order[5:])
        lens = (reads.seq_off[1:] - reads.seq_off[:-1]).astype(np.int64)
        f = reads.read_first_seq.astype(np.int64)
        paired = (f[1:] - f[:-1]) == 2
        if paired.all():
            L = np.maximum(lens[f[cand["read1"]]] - cand["pos1"], lens[f[cand["read1"]] + 1] - cand["pos2"])
        else:
            L = np.minimum(lens[f[cand["read1"]]] - cand["pos1"], lens[f[cand["read2"]]])
        ch = (L + 15) // 16
        win = np.arange(cand.size) // W
        cand = cand[np.lexsort((-ch, win))]
    n = cand.size
    d_in = torch.from_numpy(cand.view(np.uint8).reshape(-1)).cuda()
    d_out = torch.empty(n * 24, dtype=torch.uint8, device="cuda")
    scorers, outs = {}, {}
    for v in [int(x) for x in args.variants.split(",")]:
        os.environ["HC_SCORE_VARIANT"] = str(v)
        sc = hc.EdgeScorer(st)
        sc.set_reads(reads)
        sc.set_reorder(args.reorder)
        scorers[v] = sc
        sc.time_kernel(d_in.data_ptr(), n, d_out.data_ptr(), 3)
        outs[v] = d_out.cpu().numpy().copy()
    ref = next(iter(outs.values()))
    for v, o in outs.items():
        assert np.array_equal(o, ref), f"variant {v} produced different bytes"
    times = {v: [] for v in scorers}
    for _ in range(args.rounds):
        for v, sc in scorers.items():
            times[v].append(sc.time_kernel(d_in.data_ptr(), n, d_out.data_ptr(), args.iters))
    for v, t in times.items():
        t = np.array(t)
        print(f"variant {v}: median {np.median(t)*1e3:8.1f} us  min {t.min()*1e3:8.1f} us   ({n/np.median(t)/1e3:.0f} Mcand/s)")


if __name__ == "__main__":
    main()
