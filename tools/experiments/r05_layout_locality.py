#!/usr/bin/env python3
"""r05 experiment: how much of the C3 kernel's time is the store's LAYOUT?  The read store holds the reads in file order, i.e. in random
genome order: a candidate's partner window is a random 150-byte access into 384 MB (fabric traffic 45.7 GB per 1e8 candidates for 4.4 GB of
unique bytes, L2 hit rate 0.43).  Here the same reads are stored (a) as they come, (b) sorted by genome position (the generator knows it) with
the candidates in their old order, (c) as (b) with the candidates re-sorted into sfo order of the new numbering, (d) sorted by a key computed
from the reads alone: the minimiser (smallest hashed 16-mer) of mate /1.  Kernel time by hipEvents; results must be the same multiset."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import haploconduct_amd as hc  # noqa: E402
from haploconduct_amd import synth  # noqa: E402
from haploconduct_amd.readstore import ReadSet  # noqa: E402
from haploconduct_amd.records import REC_COMPACT  # noqa: E402

n_pairs, glen, n_cand = (500000, 90000, int(os.environ.get("N_CAND", "100000000")))
reads, meta = synth.make_paired_dataset(n_pairs, glen, seed=1)
cand = synth.paired_candidates(meta, n_candidates=n_cand, seed=2)
st = hc.Settings(edge_threshold=0.97, ov_threshold=0.9, merge_contigs=0.0, min_overlap_len=150)
B = np.asarray(reads.bases).reshape(n_pairs, 2, 150)
Q = np.asarray(reads.quals).reshape(n_pairs, 2, 150)


def minimiser_key(b1, k=16):
    code = np.searchsorted(np.frombuffer(b"ACGNT", np.uint8), b1).astype(np.uint64) & np.uint64(3)  # N -> some base: a key, not a result
    h = np.zeros((b1.shape[0], 150 - k + 1), np.uint64)
    for j in range(k):
        h = (h << np.uint64(2)) | code[:, j:150 - k + 1 + j]
    h = (h * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(20)
    return h.min(axis=1)


def run(name, order, resort):
    inv = np.empty(n_pairs, np.int64)
    inv[order] = np.arange(n_pairs)
    r = ReadSet(B[order].reshape(-1).copy(), Q[order].reshape(-1).copy(), np.arange(2 * n_pairs + 1, dtype=np.uint64) * 150,
                np.arange(n_pairs + 1, dtype=np.uint32) * 2, np.arange(n_pairs, dtype=np.uint64))
    c = cand.copy()
    c["read1"] = inv[cand["read1"]]
    c["read2"] = inv[cand["read2"]]
    if resort:
        lo = np.minimum(c["read1"], c["read2"]).astype(np.uint64)
        hi = np.maximum(c["read1"], c["read2"]).astype(np.uint64)
        c = c[np.argsort((lo << np.uint64(32)) | hi, kind="stable")]
    with hc.EdgeScorer(st) as sc:
        sc.set_reads(r)
        d_in = torch.from_numpy(sc.pack_cands(c).view(np.uint8).reshape(-1)).cuda()
        d_out = torch.empty(c.size * 24, dtype=torch.uint8, device="cuda")
        ms = sc.time_kernel(d_in.data_ptr(), c.size, d_out.data_ptr(), 20, REC_COMPACT)
        w = d_out.view(torch.int64).view(-1, 3)
        digest = int((w[:, 0] ^ (w[:, 1] * 31) ^ (w[:, 2] * 1000003)).sum().item()) & 0xFFFFFFFFFFFFFFFF  # order-independent
    print(json.dumps({"layout": name, "candidates_resorted": resort, "kernel_ms": ms, "digest": f"{digest:#x}"}), flush=True)


ident = np.arange(n_pairs)
run("file order (as shipped)", ident, False)
by_pos = np.argsort(meta["s"], kind="stable")
run("sorted by genome position", by_pos, False)
run("sorted by genome position", by_pos, True)
by_min = np.argsort(minimiser_key(B[:, 0]), kind="stable")
run("sorted by the minimiser of /1", by_min, False)
run("sorted by the minimiser of /1", by_min, True)
