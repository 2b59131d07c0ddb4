"""Does warming the device code inside hc_ec_open (beside the FASTQ parsing) take the kernel code loading out of a
process's first construct_edges?  Fresh process per measurement: HC_WARM=0 / 1, workload c2 / c3."""
import os, subprocess, sys, tempfile, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

if len(sys.argv) > 2 and sys.argv[1] == "child":
    import haploconduct_amd as hc
    from haploconduct_amd import host
    import pickle
    d = sys.argv[2]
    st = pickle.load(open(d + "st.pkl", "rb"))
    t0 = time.time()
    ec = host.EdgeCalculatorStage(st, paired1=d + "p1.fastq", paired2=d + "p2.fastq", overlaps=d + "ov.txt", output_dir=d)
    t1 = time.time()
    ec.construct_edges_sorted()
    t2 = time.time()
    ec.close()
    ec = host.EdgeCalculatorStage(st, paired1=d + "p1.fastq", paired2=d + "p2.fastq", overlaps=d + "ov.txt", output_dir=d)
    t3 = time.time()
    ec.construct_edges_sorted()
    t4 = time.time()
    ec.close()
    print(json.dumps({"warm": os.environ.get("HC_WARM", "1"), "open_s": round(t1 - t0, 4), "first_construct_s": round(t2 - t1, 4),
                      "second_open_s": round(t3 - t2, 4), "second_construct_s": round(t4 - t3, 4)}), flush=True)
    sys.exit(0)

import bench, pickle
import haploconduct_amd as hc
from haploconduct_amd import host
for wl in sys.argv[1:] or ["c2", "c3"]:
    reads, cand, cfg, st = bench.build_workload(wl, 0)
    d = tempfile.mkdtemp() + "/"
    reads.write_fastq(None, d + "p1.fastq", d + "p2.fastq")
    host.write_overlaps(d + "ov.txt", cand, reads)
    st.n_threads = os.cpu_count() or 8
    pickle.dump(st, open(d + "st.pkl", "wb"))
    for rep in range(int(os.environ.get("REPS", "2"))):
        for warm in ("0", "1"):
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "child", d], env={**os.environ, "HC_WARM": warm},
                                 capture_output=True, text=True)
            line = [l for l in out.stdout.splitlines() if l.startswith("{")]
            print(wl, line[-1] if line else out.stderr[-400:], flush=True)
