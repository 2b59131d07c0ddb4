#!/usr/bin/env python3
"""hc_sfo2overlaps (SFO text file -> overlaps file) on the SFO records of a bench workload written as text: the route
through the records path (plain tab-separated files) beside the general line-keeping path."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import bench
import haploconduct_amd as hc
from haploconduct_amd import host
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
reads, cand, cfg, st = bench.build_workload(wl, 0)
sc = hc.EdgeScorer(st); sc.set_reads(reads)
recs = sc.find_overlaps(0.0, 90)
n_pairs = reads.n_reads
d = "/tmp/sfotxt/"
import shutil
shutil.rmtree(d, ignore_errors=True)
os.makedirs(d)
host.write_sfo(d + "a.sfo", recs)
print(wl, "records", recs.size, "file MB", os.path.getsize(d + "a.sfo") / 1e6)
for mode in ("auto", "general"):
    if mode == "general":
        if recs.size > 5e6: break
        os.environ["HC_SFO_TEXT_GENERAL"] = "1"
    for rep in range(2):
        t = time.time(); k = host.sfo2overlaps(d + "a.sfo", d + f"o_{mode}.txt", 0, n_pairs); print(mode, "lines", k, round(time.time() - t, 3), "s")
if os.path.exists(d + "o_general.txt"):
    print("identical", open(d + "o_auto.txt", "rb").read() == open(d + "o_general.txt", "rb").read())
