#!/usr/bin/env python3
"""What the row collection adds to a step at the N = 8 shard size (1.25e7 candidates of config 3): the plain scoring launch against
hc_score_pack_device (the same kernel appending its kept rows to per-workgroup segments + the compaction kernel), 40 launches each on one
stream; run under `rocprofv3 --kernel-trace --stats` for the kernels' own durations.
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cc -- python3 tools/experiments/r06_collection_cost.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

import bench
import haploconduct_amd as hc
from haploconduct_amd import parallel
from haploconduct_amd.records import REC_COMPACT

reads, cand, cfg, st = bench.build_workload("c3", 0)
lo, hi = parallel.shard_range(cand.size, 7, 8)
n = hi - lo
sc = hc.EdgeScorer(st)
sc.set_reads(reads)
cd = sc.pack_cands(cand[lo:hi])
d_in = torch.from_numpy(cd.view(np.uint8).reshape(-1)).cuda()
d_out = torch.empty(n * 24, dtype=torch.uint8, device="cuda")
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
s = stream.cuda_stream
sc.score_cands_device(d_in.data_ptr(), n, d_out.data_ptr(), s)
torch.cuda.synchronize()
kept = int((((d_out.view(torch.int64).view(-1, 3)[:, 2] >> 60) & 0xF) != 0).sum().item())
cap = kept * 5 // 4 + 1024
payload = torch.zeros((cap + 1, 4), dtype=torch.int64, device="cuda")
pay24 = torch.zeros((cap + 1, 3), dtype=torch.int64, device="cuda")
out = {}
for name in ("plain", "pack", "pack+narrow"):
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40):
            if name == "plain":
                sc.score_cands_device(d_in.data_ptr(), n, d_out.data_ptr(), s)
            else:
                sc.score_pack_device(d_in.data_ptr(), n, d_out.data_ptr(), cap, lo, payload.data_ptr(), s, REC_COMPACT)
                if name == "pack+narrow":
                    sc.narrow_payload_device(payload.data_ptr(), cap, pay24.data_ptr(), s)
        torch.cuda.synchronize()
        out[name] = (time.perf_counter() - t0) / 40 * 1e3
print({k: round(v, 4) for k, v in out.items()}, "kept", kept, "n", n)
