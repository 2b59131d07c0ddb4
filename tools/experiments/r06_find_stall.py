"""Round 6 — how often a reads -> graph call stalls (0.6 s instead of 0.075 in the finder's batch phase) when stage follows stage in one
process on parked devices: stage_a_from_reads's alternating routes, `reps` calls of each behind the first.   python tools/experiments/r06_find_stall.py [reps]"""
import json
import os
import sys

sys.path.insert(0, os.getcwd())
import bench

reads, cand, cfg, st = bench.build_workload("c3", 0)
r = bench.stage_a_from_reads(reads, st, 32, reps=int(sys.argv[1]) if len(sys.argv) > 1 else 1)
for route, runs in r["runs"].items():
    print(route, "call_s", [x["call_s"] for x in runs], "open_s", [x["open_s"] for x in runs])
