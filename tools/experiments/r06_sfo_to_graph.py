#!/usr/bin/env python3
"""The pipelines' input to stage a — the SFO file rust-overlaps writes — to the sorted graph: one call (hc_ec_construct_edges_from_sfo) against
the three steps the pipeline runs (scripts/sfo2overlaps.py -> original_overlaps.txt -> the binary; here their ports: hc_sfo2overlaps +
hc_ec_construct_edges_sorted).  The SFO file is this library's own finder's output on the workload's reads (written once, untimed)."""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import haploconduct_amd as hc
from haploconduct_amd import host

w = sys.argv[1] if len(sys.argv) > 1 else "c3"
reads, cand, cfg, st = bench.build_workload(w, 0)
del cand
st.n_threads = 32
d = tempfile.mkdtemp(prefix="hcsfo_") + "/"
reads.write_fastq(None, d + "p1.fastq", d + "p2.fastq")
with hc.EdgeScorer(st) as sc:
    sc.set_reads(reads)
    recs = sc.find_overlaps(0.0, 90)
host.write_sfo(d + "sfoverlaps.out", recs)
n_rec = int(recs.size)
del recs
fq = dict(paired1=d + "p1.fastq", paired2=d + "p2.fastq")
host.keep_devices(True)
out = {"workload": cfg["workload"], "sfo_records": n_rec, "sfo_file_bytes": os.path.getsize(d + "sfoverlaps.out"), "one_call": [], "three_steps": []}
for rep in range(3):
    with host.EdgeCalculatorStage(st, output_dir=d, **fq) as ec:
        t0 = time.perf_counter()
        nr, nl, dev = ec.construct_edges_from_sfo(d + "sfoverlaps.out")
        t1 = time.perf_counter()
        a = (ec.edge_count(), hash(ec.edges().tobytes()))
    out["one_call"].append({"s": round(t1 - t0, 4), "lines": nl, "on_device": dev})
    t0 = time.perf_counter()
    n_lines = host.sfo2overlaps(d + "sfoverlaps.out", d + "original_overlaps.txt", 0, reads.n_reads)
    t1 = time.perf_counter()
    with host.EdgeCalculatorStage(st, output_dir=d, overlaps=d + "original_overlaps.txt", **fq) as ec:
        t2 = time.perf_counter()
        ec.construct_edges_sorted()
        t3 = time.perf_counter()
        b = (ec.edge_count(), hash(ec.edges().tobytes()))
    assert a == b and nl == n_lines
    out["three_steps"].append({"sfo2overlaps_s": round(t1 - t0, 4), "construct_edges_sorted_s": round(t3 - t2, 4), "s": round(t1 - t0 + t3 - t2, 4)})
host.keep_devices(False)
out["edges"] = a[0]
print(json.dumps(out))
