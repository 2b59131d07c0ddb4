#!/usr/bin/env python3
"""Does the kernel's time follow the number of 128-byte LINES a partner window touches?  A read set small enough for the Infinity Cache either
way (200 000 pairs: 154 MB packed, 205 MB with 128-byte-aligned slots) scored with HC_SLOT_ALIGN=16 (slots of 192 B: a pair's /1 prefix starts
on a line, its /2 prefix in the middle of one: 3.3 lines per candidate) and HC_SLOT_ALIGN=128 (256-byte slots: 2.6 lines).  If the model of
profiles/r05_kernel_why_not.md holds, the aligned store is faster by about the ratio of the lines although it is a third larger."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    import numpy as np
    import torch

    import haploconduct_amd as hc
    from haploconduct_amd import synth
    from haploconduct_amd.records import REC_COMPACT

    reads, meta = synth.make_paired_dataset(200000, 36000, seed=1)
    cand = synth.paired_candidates(meta, n_candidates=40000000, seed=2)
    st = hc.Settings(edge_threshold=0.97, ov_threshold=0.9, merge_contigs=0.0, min_overlap_len=150)
    with hc.EdgeScorer(st) as sc:
        sc.set_reads(reads)
        d_in = torch.from_numpy(sc.pack_cands(cand).view(np.uint8).reshape(-1)).cuda()
        d_out = torch.empty(cand.size * 24, dtype=torch.uint8, device="cuda")
        ms = sc.time_kernel(d_in.data_ptr(), cand.size, d_out.data_ptr(), 20, REC_COMPACT)
        print(json.dumps({"HC_SLOT_ALIGN": os.environ.get("HC_SLOT_ALIGN"), "kernel_ms": ms, "store_bytes": sc.info()["store_bytes"], "candidates": int(cand.size)}))
else:
    for a in ("16", "128", "16", "128"):
        subprocess.run([sys.executable, __file__, "run"], env=dict(os.environ, HC_SLOT_ALIGN=a))
