"""FastqStorage load time, sequential reader against the mapped multi-threaded one (C3's reads: 2 x 500 000 x 150 bp)."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from haploconduct_amd import synth, host
reads, meta = synth.make_paired_dataset(500000, 90000, seed=1)
d = tempfile.mkdtemp()
reads.write_fastq(None, d + "/p1.fastq", d + "/p2.fastq")
os.environ["HC_STAGE_TIMING"] = "1"
for thr in ("1", "1", "4", "8", "16", "16"):
    os.environ["HC_FASTQ_THREADS"] = thr
    t = time.time()
    f = host.Fastq(paired1=d + "/p1.fastq", paired2=d + "/p2.fastq")
    print("threads", thr, "load", round(time.time() - t, 3), f.n_paired, flush=True)
    f.close()
