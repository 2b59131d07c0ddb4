#!/usr/bin/env python3
"""What a rank of an N-GPU run will see, measured on ONE GPU: the scoring kernel and the N-rank step (kernel +
row payload + one RCCL all-gather, world size 1) at the shard sizes of the strong split of the 10^8-candidate set — 10^8 / N for
N = 1, 2, 4, 8 — and at the full size (the weak figure); round 6: the rows travel in the 24-byte form (hc_narrow_payload_device).  From these, a PREDICTED curve: strong(N) = 10^8 / step_ms(10^8 / N),
weak(N) = N * 10^8 / step_ms(10^8), both assuming that the all-gather stays hidden behind the next step's kernel as it is at world
size 1 (the payload a rank contributes: ~3.75 * 10^6 / N rows of 32 B strong, 120 MB weak; over xGMI's 7 links at ~150 GB/s each,
MI355X_MICROARCH.md, that is 0.075 ms strong at N = 8 and 0.6 ms weak against kernels of 0.9 and 6.5 ms; the "root" form moves the same bytes per link —
every rank's rows over ONE link to rank 0 — but nothing INTO the memory of the seven ranks that are scoring).  Nothing here is a
measurement of N > 1: the output says "predicted" in every figure derived that way.

    python tools/predict_scaling.py [--workload c3] [--steps 20] > gpurun_out/r04_predicted_scaling.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist

    import bench
    import haploconduct_amd as hc
    from haploconduct_amd import parallel
    from haploconduct_amd.records import REC_COMPACT

    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29671", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    reads, cand, cfg, settings = bench.build_workload(args.workload, 0)
    n_all = int(cand.size)
    sc = hc.EdgeScorer(settings)
    sc.set_reads(reads)
    cd = sc.pack_cands(cand)
    d_all = torch.from_numpy(cd.view(np.uint8).reshape(-1)).cuda()
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    rows = []
    for N in (1, 2, 4, 8):
        lo, hi = parallel.shard_range(n_all, N - 1, N)  # the last rank's shard (the largest index range; sizes differ by one at most)
        n = hi - lo
        d_in = d_all[lo * 16:hi * 16]
        d_out = torch.empty(n * 24, dtype=torch.uint8, device="cuda")
        sc.score_cands_device(d_in.data_ptr(), n, d_out.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize()
        kept = int((((d_out.view(torch.int64).view(-1, 3)[:, 2] >> 60) & 0xF) != 0).sum().item())
        kern_ms = sc.time_kernel(d_in.data_ptr(), n, d_out.data_ptr(), args.steps, REC_COMPACT)
        by_reserve = {}
        for reserve in (0, 16, 32):  # round 5: CUs the scoring launches leave to the exchange's kernels (hc_set_comm_reserve; profiles/r05_coresident.md)
            g = parallel.StreamedGather(sc, n, base_index=lo, cap_rows=kept * 5 // 4 + 1024, rec_fmt=REC_COMPACT, reserve_cus=reserve, narrow=True)
            for _ in range(3):
                g.score_step(d_in.data_ptr(), d_out)
            g.finish()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            last = None
            for _ in range(args.steps):
                last = g.score_step(d_in.data_ptr(), d_out)
            stream.synchronize()
            g.finish()
            torch.cuda.synchronize()
            by_reserve[reserve] = (time.perf_counter() - t0) / args.steps * 1e3
            out_rows, counts = g.collect(last)
            assert counts == [kept]
            g.close()
            del g
        sc.set_comm_reserve(0)
        step_ms = by_reserve[0]
        rows.append({"N": N, "shard_candidates": n, "kept_rows": kept, "payload_MB_per_rank": (kept + 1) * 24 / 1e6, "row_bytes": 24, "kernel_ms": kern_ms,
                     "step_ms_world1": step_ms, "step_ms_world1_by_reserved_cus": {str(k): v for k, v in by_reserve.items()},
                     "collection_on_critical_path_ms_world1": max(0.0, step_ms - kern_ms)})
        del d_out
        torch.cuda.empty_cache()
    full = rows[0]
    out = {"workload": cfg["workload"], "measured_on": "one MI355X, world size 1 over RCCL (HC_BENCH_FORCE_GATHER's code path)", "steps": args.steps,
           "measured": rows,
           "predicted": [{"N": r["N"],
                          "strong_candidates_per_s_PREDICTED": n_all / (r["step_ms_world1"] * 1e-3),
                          "strong_speedup_over_1_PREDICTED": full["step_ms_world1"] / r["step_ms_world1"],
                          # round 5 (profiles/r05_coresident.md): the exchange runs BESIDE the kernel only on CUs the launch leaves free; without them it
                          # is serialised behind the kernel.  xgmi estimate x 2: beside the kernel a transfer shares the memory side (measured with stand-ins)
                          "strong_speedup_PREDICTED_32_cus_reserved_exchange_overlapped": full["step_ms_world1"] / r["step_ms_world1_by_reserved_cus"]["32"],
                          "strong_speedup_PREDICTED_no_reserve_exchange_serialised": full["step_ms_world1"] / (
                              r["step_ms_world1"] + (2.0 * (r["N"] - 1) * r["payload_MB_per_rank"] / 1e3 / (min(r["N"] - 1, 7) * 150.0) * 1e3 if r["N"] > 1 else 0.0)),
                          "weak_candidates_per_s_PREDICTED": r["N"] * n_all / (full["step_ms_world1"] * 1e-3),
                          "xgmi_allgather_ms_ESTIMATE_strong": (r["N"] - 1) * r["payload_MB_per_rank"] / 1e3 / (min(r["N"] - 1, 7) * 150.0) * 1e3 if r["N"] > 1 else 0.0,
                          "xgmi_allgather_ms_ESTIMATE_weak": (r["N"] - 1) * full["payload_MB_per_rank"] / 1e3 / (min(r["N"] - 1, 7) * 150.0) * 1e3 if r["N"] > 1 else 0.0}
                         for r in rows],
           "assumptions": "predicted = a rank's own step time at its shard size, measured at world size 1; the all-gather over xGMI is assumed to stay "
                          "behind the next step's kernel (estimates above: bytes a rank receives / (links used x 150 GB/s)); no N > 1 run has been "
                          "measured by the builder"}
    dist.destroy_process_group()
    sys.stdout.flush()
    import ctypes
    ctypes.CDLL(None).fflush(None)  # RCCL's banner sits in C stdio: it goes to stderr with the rest
    os.dup2(real_stdout, 1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
