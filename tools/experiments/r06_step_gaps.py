#!/usr/bin/env python3
"""The N-rank step at the N = 8 shard size on one GPU (world size 1 over RCCL): run under `rocprofv3 --kernel-trace` to see what sits between
two consecutive scoring kernels.  python3 tools/experiments/r06_step_gaps.py [mode] [narrow 0/1]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist

import bench
import haploconduct_amd as hc
from haploconduct_amd import parallel
from haploconduct_amd.records import REC_COMPACT

mode = sys.argv[1] if len(sys.argv) > 1 else "ring"
narrow = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29673", rank=0, world_size=1, device_id=torch.device("cuda", 0))
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
g = parallel.StreamedGather(sc, n, base_index=lo, cap_rows=600000, rec_fmt=REC_COMPACT, mode=mode, narrow=narrow)
for _ in range(5):
    g.score_step(d_in.data_ptr(), d_out)
g.finish()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(40):
    g.score_step(d_in.data_ptr(), d_out)
stream.synchronize()
g.finish()
torch.cuda.synchronize()
print("step_ms", round((time.perf_counter() - t0) / 40 * 1e3, 4), mode, "narrow" if narrow else "wide")
dist.destroy_process_group()
