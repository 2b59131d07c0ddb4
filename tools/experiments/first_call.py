"""Why is a process's first construct_edges slower than its later ones?  C2-sized stage, several fresh EdgeCalculators in one
process, optionally with the GPU kept busy right before each call."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
import haploconduct_amd as hc
from haploconduct_amd import host, synth

reads, cand, cfg, st = bench.build_workload("c2", 0)
d = tempfile.mkdtemp() + "/"
reads.write_fastq(None, d + "p1.fastq", d + "p2.fastq")
open(d + "ov.txt", "w").write("\n".join(synth.records_to_lines(cand, reads)) + "\n")
st = hc.Settings(**{**st.__dict__, "n_threads": 32}) if hasattr(st, "__dict__") else st
busy = len(sys.argv) > 1 and sys.argv[1] == "busy"
with hc.EdgeScorer(st) as sc:
    sc.set_reads(reads)
    for k in range(5):
        ec = host.EdgeCalculatorStage(st, paired1=d + "p1.fastq", paired2=d + "p2.fastq", overlaps=d + "ov.txt", output_dir=d)
        if busy:
            for _ in range(30):
                sc.score_batch(cand[:200000])
        t = time.time()
        ec.construct_edges_sorted()
        print("call", k, "busy" if busy else "idle", round(time.time() - t, 4), flush=True)
        ec.close()
        if k == 2:
            time.sleep(1.0)  # let the GPU idle
