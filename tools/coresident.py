#!/usr/bin/env python3
"""What does a communication kernel cost BESIDE the scoring kernel?  (VERDICT r4, next #1d; one GPU.)

The N-rank step of bench.py (parallel.StreamedGather.score_step) with the collective replaced by a stand-in kernel of the same
stream structure: a device-to-device copy on the side stream, issued right behind the launch of the next step's scoring kernel,
in two shapes (tools/experiments/comm_like.hip): "thin" (10 registers, no LDS: fits on a CU beside a scoring workgroup) and "fat"
(128 registers, 32 KiB of LDS: the shape of a collective library's generic kernel — needs a CU without a scoring workgroup).  The
scoring launch is the strong split's shard at N = 8 (1.25e7 of config 3's 1e8 candidates) or any --candidates; the copy moves
--comm-bytes (105 MiB: what a rank sends and receives per step at N = 8, 7 x 15 MB) — as many bytes out of and into the rank's HBM as
the real exchange moves; how long it takes alone depends on the shape (reported).

Per scenario (reserve = CUs the scoring launch leaves free, hc_set_comm_reserve; gate = hc_comm_gate_device in front of the
stand-in): ms per step over --steps steps, against the step without any exchange; the stand-in's own duration (events on the side
stream).  One JSON object per line on stdout."""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def comm_lib():
    d = os.path.join(ROOT, "tools", "experiments")
    so = os.path.join(d, "libcommlike.so")
    if not os.path.exists(so):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(d, "comm_like.hip")])
    lib = C.CDLL(so)
    lib.comm_like_launch.restype = C.c_int
    lib.comm_like_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_void_p]
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--candidates", type=int, default=12500000)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--comm-bytes", type=int, default=105 << 20)
    ap.add_argument("--reserves", default="0,8,16,32")
    ap.add_argument("--only", default=None, help="reserve:shape-name:gate(0|1)[,...] — just these scenarios (for a kernel trace)")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist

    import bench
    import haploconduct_amd as hc
    from haploconduct_amd import parallel
    from haploconduct_amd.records import REC_COMPACT

    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29671", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    lib = comm_lib()
    reads, cand, cfg, settings = bench.build_workload(args.workload, 0)
    cand = cand[: args.candidates]
    n = int(cand.size)
    sc = hc.EdgeScorer(settings)
    sc.set_reads(reads)
    d_in = torch.from_numpy(sc.pack_cands(cand).view(np.uint8).reshape(-1)).cuda()
    d_out = torch.empty(n * 24, dtype=torch.uint8, device="cuda")
    launch = torch.cuda.Stream()
    torch.cuda.set_stream(launch)
    sc.score_cands_device(d_in.data_ptr(), n, d_out.data_ptr(), launch.cuda_stream)
    torch.cuda.synchronize()
    kept = int((((d_out.view(torch.int64).view(-1, 3)[:, 2] >> 60) & 0xF) != 0).sum().item())
    src = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    dst = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")

    def alone_ms(nbytes, blocks, lds, fat, reps=20):
        s = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        lib.comm_like_launch(dst.data_ptr(), src.data_ptr(), nbytes, blocks, lds, fat, s.cuda_stream)
        torch.cuda.synchronize()
        e0.record(s)
        for _ in range(reps):
            lib.comm_like_launch(dst.data_ptr(), src.data_ptr(), nbytes, blocks, lds, fat, s.cuda_stream)
        e1.record(s)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    shapes = [("none", 0, 0, 0), ("thin x56", 56, 0, 0), ("fat x16", 16, 32768, 1), ("fat x32", 32, 32768, 1), ("fat x64", 64, 32768, 1)]
    sized = {}
    for name, blocks, lds, fat in shapes[1:]:
        nbytes = args.comm_bytes  # what a rank moves per step at N = 8: 7 x 15 MB out of and into its HBM
        sized[name] = (nbytes, alone_ms(nbytes, blocks, lds, fat))
    print(json.dumps({"what": "stand-in kernels alone", "sized": {k: {"bytes": v[0], "alone_ms": v[1]} for k, v in sized.items()},
                      "scoring": sc.kernel_info(n), "candidates": n, "kept_rows": kept}), flush=True)

    def run(reserve, shape, gate):
        name, blocks, lds, fat = shape
        g = parallel.StreamedGather(sc, n, base_index=0, cap_rows=kept * 5 // 4 + 1024, rec_fmt=REC_COMPACT, mode="ring", reserve_cus=reserve)
        if not gate:
            g.reserve_gate_off = True
            g.flush_orig = g.flush
            g.flush = lambda gate=None: g.flush_orig(None)
        if name == "none":
            g._ring = lambda b: []
        else:
            nbytes = sized[name][0]
            g._ring = lambda b: lib.comm_like_launch(dst.data_ptr(), src.data_ptr(), nbytes, blocks, lds, fat, torch.cuda.current_stream().cuda_stream) and []
        for _ in range(3):
            g.score_step(d_in.data_ptr(), d_out)
        g.finish()
        g.reset_timings()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            g.score_step(d_in.data_ptr(), d_out)
        launch.synchronize()
        t_scored = time.perf_counter()
        g.finish()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        rec = {"reserve_cus": reserve, "comm": name, "gate": gate, "ms_per_step": (t1 - t0) / args.steps * 1e3,
               "scoring_done_ms_per_step": (t_scored - t0) / args.steps * 1e3, "comm_ms_beside": g.gather_ms(),
               "comm_ms_alone": sized[name][1] if name != "none" else 0.0}
        sc.set_comm_reserve(0)
        return rec

    print(json.dumps({"what": "scoring kernel alone (hipEvents, back-to-back launches)", "kernel_ms": sc.time_kernel(d_in.data_ptr(), n, d_out.data_ptr(), 50, REC_COMPACT)}), flush=True)
    if args.only:
        for item in args.only.split(","):
            r, nm, gt = item.split(":")
            print(json.dumps(run(int(r), next(sh for sh in shapes if sh[0] == nm), gt == "1")), flush=True)
        sc.close()
        dist.destroy_process_group()
        return
    for reserve in [int(x) for x in args.reserves.split(",")]:
        for shape in shapes:
            for gate in ((False,) if (shape[0] == "none" or reserve == 0) else (False, True)):
                print(json.dumps(run(reserve, shape, gate)), flush=True)
    print(json.dumps(dict(run(0, shapes[0], False), note="the first scenario once more (drift check)")), flush=True)
    sc.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
