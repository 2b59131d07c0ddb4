"""Stage at C3 with and without its threads bound to the device's NUMA node (HC_NUMA=0), alternating in one process:
open (FASTQ -> store) and construct_edges_sorted seconds."""
import json, os, sys, tempfile, time, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from haploconduct_amd import host

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
reads, cand, cfg, st = bench.build_workload(wl, 0)
d = tempfile.mkdtemp(prefix="hcnuma_") + "/"
host.write_overlaps(d + "ov.txt", cand, reads)
reads.write_fastq(None, d + "p1.fastq", d + "p2.fastq")
del cand
st.n_threads = 32
kw = dict(paired1=d + "p1.fastq", paired2=d + "p2.fastq", overlaps=d + "ov.txt", output_dir=d)
for rnd in range(3):
    for numa in ("1", "0"):
        os.environ["HC_NUMA"] = numa
        opens, cons = [], []
        for rep in range(4):
            if os.path.exists(d + "nonedge_overlaps.txt"):
                os.remove(d + "nonedge_overlaps.txt")
            t = time.perf_counter()
            ec = host.EdgeCalculatorStage(st, **kw)
            t1 = time.perf_counter()
            ec.construct_edges_sorted()
            t2 = time.perf_counter()
            ec.close()
            opens.append(round(t1 - t, 4))
            cons.append(round(t2 - t1, 4))
        print(json.dumps({"bound_to_device_node": numa == "1", "open_s": opens, "construct_s": cons, "open_median": sorted(opens)[2],
                          "construct_median": sorted(cons)[2]}), flush=True)
shutil.rmtree(d, ignore_errors=True)
