#!/usr/bin/env python3
"""Can RCCL run two ranks on ONE device (so that its real kernels could be measured beside the scoring kernel on a one-GPU box)?
Launched with torch.distributed.run --nproc-per-node 2; every rank takes cuda:0."""
import os
import sys
import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    x = torch.full((1 << 20,), rank + 1, dtype=torch.int64, device="cuda")
    out = torch.zeros(world << 20, dtype=torch.int64, device="cuda")
    dist.all_gather_into_tensor(out, x)
    torch.cuda.synchronize()
    print(f"rank {rank}: all_gather ok, sum {int(out.sum().item())}", flush=True)
    dist.destroy_process_group()
except Exception as e:
    print(f"rank {rank}: FAILED {type(e).__name__}: {str(e)[:400]}", flush=True)
    sys.exit(3)
