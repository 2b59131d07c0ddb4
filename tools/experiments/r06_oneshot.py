"""Round 6 — what a one-shot hc-edgecalc process spends on a small stage (4 000 pairs, 10^5 candidates): wall per process and the stage's own
timing lines.   python tools/experiments/r06_oneshot.py"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from haploconduct_amd import host, synth  # noqa: E402

EXE = os.path.join(ROOT, "haploconduct_amd", "csrc", "hc-edgecalc")
d = tempfile.mkdtemp(prefix="hconeshot_") + "/"
reads, meta = synth.make_paired_dataset(4000, 6000, flip_frac=0.25, seed=21)
cand = synth.paired_candidates(meta, n_candidates=100000, seed=22)
host.write_overlaps(d + "overlaps.txt", cand, reads)
reads.write_fastq(None, d + "p1.fastq", d + "p2.fastq")
args = ["--paired1", d + "p1.fastq", "--paired2", d + "p2.fastq", "--overlaps", d + "overlaps.txt", "--original_readcount", str(reads.n_reads), "--threads", "8",
        "--edge_threshold", "0.97", "--min_overlap_len", "150", "--output", d]
for k in range(4):
    env = dict(os.environ, **({"HC_STAGE_TIMING": "1"} if k == 3 else {}))
    t0 = time.perf_counter()
    r = subprocess.run([EXE] + args, env=env, capture_output=True, text=True)
    dt = time.perf_counter() - t0
    print("process", k, "rc", r.returncode, "wall %.3f s" % dt)
    if k == 3:
        print(r.stderr[-6000:])
t0 = time.perf_counter()
subprocess.run([EXE, "--help"], capture_output=True)
print("--help: %.3f s" % (time.perf_counter() - t0))
