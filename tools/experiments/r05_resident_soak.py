#!/usr/bin/env python3
"""A resident process under a long pipeline: N calls alternating between the SAVAGE example's stage-a and stage-b/c argv (tools/c1_process.py's
inputs) through ONE `hc-edgecalc --resident` process; every call's outputs must equal the first call's of its stage, and the resident
process's memory (VmRSS from /proc, its pid from the socket directory) must not grow call by call."""
import argparse
import gzip
import json
import glob
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=300)
    a = ap.parse_args()
    import haploconduct_amd as hc
    from haploconduct_amd import host

    d = tempfile.mkdtemp(prefix="hcsoak_") + "/"
    paths = {}
    for name in ("savage_singles", "savage_paired1", "savage_paired2"):
        paths[name] = d + name + ".fastq"
        with gzip.open(os.path.join(ROOT, "tests", "golden", name + ".fastq.gz"), "rb") as f, open(paths[name], "wb") as o:
            o.write(f.read())
    f = host.Fastq(singles=paths["savage_singles"], paired1=paths["savage_paired1"], paired2=paths["savage_paired2"])
    with hc.EdgeScorer(hc.Settings()) as sc:
        sc.set_reads(f.readset())
        sfo = sc.find_overlaps(0.02, 100)
    host.sfo_records_to_overlaps(sfo, d + "overlaps.txt", f.n_single, f.n_paired)
    exe = os.path.join(ROOT, "haploconduct_amd", "csrc", "hc-edgecalc")
    env = dict(os.environ, HC_RESIDENT_DIR=d + "res", HC_RESIDENT_IDLE_S="120")
    base = [exe, "--resident", f"--singles={paths['savage_singles']}", f"--paired1={paths['savage_paired1']}", f"--paired2={paths['savage_paired2']}",
            f"--overlaps={d}overlaps.txt", "--threads=8", f"--original_readcount={f.n_single + f.n_paired}"]
    stages = [base + ["--edge_threshold=0.97", "--min_overlap_len=200"],
              base + ["--edge_threshold=0.995", "--min_overlap_len=100", "--merge_contigs=0.01", "--ignore_inclusions=true"]]
    names = ("edges.tsv", "edges_sorted.tsv", "nonedge_overlaps.txt", "edgecalc_stats.txt")
    want, rss, walls = [None, None], [], []
    try:
        for k in range(a.calls):
            o = d + "out/"
            os.makedirs(o, exist_ok=True)
            for fn in names:
                if os.path.exists(o + fn):
                    os.remove(o + fn)
            t0 = time.perf_counter()
            r = subprocess.run(stages[k % 2] + [f"--output={o}"], env=env, capture_output=True, text=True)
            walls.append(time.perf_counter() - t0)
            assert r.returncode == 0, r.stderr[-1000:]
            got = {fn: open(o + fn, "rb").read() for fn in names}
            if want[k % 2] is None:
                want[k % 2] = got
            assert got == want[k % 2], f"call {k}: other outputs than the first call of its stage"
            pid = int(open((glob.glob(d + "res/vis-*/pid") + [d + "res/pid"])[0]).read())  # (round 6: a sub-directory per device-visibility setting)
            for ln in open(f"/proc/{pid}/status"):
                if ln.startswith("VmRSS"):
                    rss.append(int(ln.split()[1]))
    finally:
        subprocess.run([exe, "--resident_stop"], env=env)
    q = len(rss) // 4
    print(json.dumps({"calls": a.calls, "outputs_identical": True, "wall_s_median_later": sorted(walls[2:])[len(walls[2:]) // 2],
                      "resident_VmRSS_kB": {"after_call_4": rss[3], "first_quarter_max": max(rss[4:q]), "last_quarter_max": max(rss[-q:]), "last": rss[-1]},
                      "growth_kB_per_call_last_three_quarters": (rss[-1] - rss[q]) / max(1, len(rss) - q)}))


if __name__ == "__main__":
    main()
