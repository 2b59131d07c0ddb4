"""Stage at C3 over its pipeline knobs (read in the EdgeCalculator's constructor): text block size, blocks in flight,
collector threads.  One process, the files written once; per setting the construct_edges_sorted times of several runs."""
import json, os, sys, tempfile, time, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from haploconduct_amd import host

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
reads, cand, cfg, st = bench.build_workload(wl, 0)
d = tempfile.mkdtemp(prefix="hcsweep_") + "/"
host.write_overlaps(d + "ov.txt", cand, reads)
reads.write_fastq(None, d + "p1.fastq", d + "p2.fastq")
del cand
st.n_threads = min(64, os.cpu_count() or 8)
kw = dict(paired1=d + "p1.fastq", paired2=d + "p2.fastq", overlaps=d + "ov.txt", output_dir=d)
settings = [{}] + [{"HC_TEXT_BLOCK": str(b << 20), "HC_TEXT_DEPTH": str(dp), "HC_COLLECTORS": str(c)}
                   for b in (8, 16, 32) for dp in (4, 6, 10, 16) for c in (4, 8) if not (b == 16 and dp == 6 and c == 4)]
for env in settings:
    for k in ("HC_TEXT_BLOCK", "HC_TEXT_DEPTH", "HC_COLLECTORS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    ts = []
    for rep in range(4):
        if os.path.exists(d + "nonedge_overlaps.txt"):
            os.remove(d + "nonedge_overlaps.txt")
        ec = host.EdgeCalculatorStage(st, **kw)
        t = time.perf_counter()
        ec.construct_edges_sorted()
        ts.append(round(time.perf_counter() - t, 4))
        ec.close()
    print(json.dumps({"settings": env or "default (16 MiB, 6, 4)", "construct_s": ts, "best": min(ts), "median": sorted(ts)[len(ts) // 2]}), flush=True)
shutil.rmtree(d, ignore_errors=True)
