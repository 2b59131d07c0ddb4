"""Deep-coverage check of hc_find_overlaps: 70 k reads over a 5 kb genome (~2 100x), one batch vs several
(HC_FIND_BATCH_HITS): same records, sorted, unique."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import haploconduct_amd as hc
from haploconduct_amd import synth
reads, meta = synth.make_paired_dataset(35000, 5000, flip_frac=0.0, seed=3)   # 70k x 150 bp over 5 kb: ~2100x
res = {}
for env in (None, "5000000"):
    if env: os.environ["HC_FIND_BATCH_HITS"] = env
    with hc.EdgeScorer(hc.Settings()) as sc:
        sc.set_reads(reads)
        t = time.perf_counter(); n = sc.find_overlaps(0.0, 90, count_only=True); dt = time.perf_counter() - t
        t = time.perf_counter(); n2 = sc.find_overlaps(0.0, 90, count_only=True); dt2 = time.perf_counter() - t
        recs = sc.find_overlaps(0.0, 90)
    key = (recs["idA"].astype(np.uint64) << 40) | (recs["idB"].astype(np.uint64) << 16) | (recs["inverted"].astype(np.uint64) << 15) | (recs["OHA"].astype(np.int64) + 16384).astype(np.uint64)
    assert (np.diff(key.astype(np.int64)) > 0).all(), "not sorted / not unique"
    res[env] = (n, recs.tobytes())
    print("batch_hits", env, "overlaps", n, "first call %.3f s, second %.3f s" % (dt, dt2))
assert res[None] == res["5000000"]
print("identical")
