"""hc_find_overlaps twice in one process (the scratch slots only grow: the second call shows the steady state)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import haploconduct_amd as hc
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
reads, cand, cfg, st = bench.build_workload(wl, 0)  # c3-lite: the reads of c3 with fewer candidates to build
del cand
if os.environ.get("FT_HOST"):
    from haploconduct_amd import host
if os.environ.get("FT_FASTQ"):
    import tempfile
    d = tempfile.mkdtemp() + "/"
    reads.write_fastq(None, d + "p1.fastq", d + "p2.fastq")
if os.environ.get("FT_THREADS"):
    st.n_threads = 64
with hc.EdgeScorer(st) as sc:
    sc.set_reads(reads)
    for k in range(3):
        t = time.perf_counter()
        recs = sc.find_overlaps(0.0, 90)
        print("call", k, recs.size, round(time.perf_counter() - t, 4), flush=True)
        del recs
