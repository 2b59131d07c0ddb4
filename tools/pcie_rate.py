#!/usr/bin/env python3
"""The PCIe-inclusive scoring rate (never bench.py's `value`): hc_score_batch on host buffers —
H2D of the candidate records, kernel, D2H of the result records — with page-locked (hc_host_alloc) and with pageable
memory, on the bench workload."""
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

    reads, cand, cfg, st = bench.build_workload("c2", 0)
    n = int(cand.size)
    with hc.EdgeScorer(st) as sc:
        sc.set_reads(reads)
        pin_in, pin_out = C.c_void_p(), C.c_void_p()
        N.check(N.lib.hc_host_alloc(sc._ctx, C.byref(pin_in), n * 32), "hc_host_alloc")
        N.check(N.lib.hc_host_alloc(sc._ctx, C.byref(pin_out), n * 24), "hc_host_alloc")
        C.memmove(pin_in, cand.ctypes.data, n * 32)
        out_pageable = np.zeros(n * 24, np.uint8)
        res = {}
        for name, pi, po in (("page_locked", pin_in, pin_out), ("pageable", C.c_void_p(cand.ctypes.data), C.c_void_p(out_pageable.ctypes.data))):
            for _ in range(3):
                N.check(N.lib.hc_score_batch(sc._ctx, pi, n, po), "hc_score_batch")
            t0 = time.perf_counter()
            reps = 20
            for _ in range(reps):
                N.check(N.lib.hc_score_batch(sc._ctx, pi, n, po), "hc_score_batch")
            dt = (time.perf_counter() - t0) / reps
            res[name] = {"ms_per_batch": round(dt * 1e3, 3), "candidates_per_s": round(n / dt), "pcie_GB_per_s": round(n * 56 / dt / 1e9, 1)}
        same = bytes((C.c_char * (n * 24)).from_address(pin_out.value)) == out_pageable.tobytes()
        N.lib.hc_host_free(sc._ctx, pin_in)
        N.lib.hc_host_free(sc._ctx, pin_out)
    print(json.dumps({"workload": cfg["workload"], "candidates": n, "bytes_over_pcie_per_candidate": 56, "identical_results": same, **res}))


if __name__ == "__main__":
    main()
