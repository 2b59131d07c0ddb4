#!/usr/bin/env python3
"""The PCIe-inclusive scoring rate (never bench.py's `value`): candidate records start in page-locked HOST
memory and what the host needs of the results ends there —
  blocks : hc_block_submit / hc_block_wait, the stage's device leg: 16-byte records over PCIe, the scoring kernel
           writing the non-dropped rows straight into host memory, several blocks in flight;
  full   : hc_score_cands / hc_score_batch, one synchronous call: records in, all 24-byte result records out."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import bench
    import haploconduct_amd as hc
    from haploconduct_amd import _native as N

    workload = sys.argv[1] if len(sys.argv) > 1 else "c2"
    reads, cand, cfg, st = bench.build_workload(workload, 0)
    n = int(cand.size)
    res = {}
    with hc.EdgeScorer(st) as sc:
        sc.set_reads(reads)
        cd = sc.pack_cands(cand)
        pin = C.c_void_p()
        N.check(N.lib.hc_host_alloc(sc._ctx, C.byref(pin), n * 16), "hc_host_alloc")
        C.memmove(pin, cd.ctypes.data, n * 16)
        for block, depth in ((1 << 20, 4), (250000, 4), (1 << 22, 3)):
            blocks = []
            for _ in range(depth):
                b = C.c_void_p()
                N.check(N.lib.hc_block_create(sc._ctx, block, C.byref(b)), "hc_block_create")
                blocks.append(b)
            rows, k = C.c_void_p(), C.c_uint64()

            spent = {"submit": 0.0, "wait": 0.0}

            def run():
                pending, kept = [], 0
                for i, at in enumerate(range(0, n, block)):
                    if len(pending) == depth:
                        t = time.perf_counter()
                        N.check(N.lib.hc_block_wait(pending.pop(0), C.byref(rows), C.byref(k)), "hc_block_wait")
                        spent["wait"] += time.perf_counter() - t
                        kept += k.value
                    b = blocks[i % depth]
                    t = time.perf_counter()
                    N.check(N.lib.hc_block_submit(b, C.c_void_p(pin.value + at * 16), min(block, n - at), at), "hc_block_submit")
                    spent["submit"] += time.perf_counter() - t
                    pending.append(b)
                while pending:
                    t = time.perf_counter()
                    N.check(N.lib.hc_block_wait(pending.pop(0), C.byref(rows), C.byref(k)), "hc_block_wait")
                    spent["wait"] += time.perf_counter() - t
                    kept += k.value
                return kept

            run()
            reps = 5 if n > 10000000 else 30
            spent["submit"] = spent["wait"] = 0.0
            t0 = time.perf_counter()
            for _ in range(reps):
                kept = run()
            dt = (time.perf_counter() - t0) / reps
            res[f"blocks_{block}_x{depth}"] = {"ms": round(dt * 1e3, 3), "candidates_per_s": round(n / dt), "rows_back": kept,
                                               "in_submit_ms": round(spent["submit"] / reps * 1e3, 3), "in_wait_ms": round(spent["wait"] / reps * 1e3, 3),
                                               "pcie_GB_per_s": round((n * 16 + kept * 32) / dt / 1e9, 1)}
            for b in blocks:
                N.lib.hc_block_destroy(b)
        if n <= 20000000:
            pin_out = C.c_void_p()
            N.check(N.lib.hc_host_alloc(sc._ctx, C.byref(pin_out), n * 24), "hc_host_alloc")
            for name, fn, rb in (("full_cands16", N.lib.hc_score_cands, 16),):
                for _ in range(3):
                    N.check(fn(sc._ctx, pin, n, pin_out), name)
                t0 = time.perf_counter()
                for _ in range(20):
                    N.check(fn(sc._ctx, pin, n, pin_out), name)
                dt = (time.perf_counter() - t0) / 20
                res[name] = {"ms": round(dt * 1e3, 3), "candidates_per_s": round(n / dt), "pcie_GB_per_s": round(n * (rb + 24) / dt / 1e9, 1)}
            N.lib.hc_host_free(sc._ctx, pin_out)
        N.lib.hc_host_free(sc._ctx, pin)
    print(json.dumps({"workload": cfg["workload"], "candidates": n, **res}))


if __name__ == "__main__":
    main()
