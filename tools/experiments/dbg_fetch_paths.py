"""Debug helper: the cooperative fetch against the per-lane fetch on single-end candidates of every window length."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import haploconduct_amd as hc
from haploconduct_amd.records import OVERLAP_DTYPE, result_n
def run(fetch, reads, cand, st):
    os.environ["HC_FETCH_GROUP"] = fetch
    with hc.EdgeScorer(st) as sc:
        sc.set_reads(reads)
        return sc.score_batch(cand).copy()
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(1)
alphabet = rng.choice(np.arange(33, 127), size=nq, replace=False).astype(np.uint8)
L = 150
seq = rng.choice(np.frombuffer(b"ACGT", np.uint8), L)
singles = [(seq.tobytes(), alphabet[rng.integers(0, nq, L)].tobytes()) for _ in range(4)]
reads = hc.ReadSet.from_lists(singles, [])
cand = np.zeros(L, OVERLAP_DTYPE)
cand["read1"], cand["read2"] = 0, 1
cand["pos1"] = np.arange(L)
cand["ori1"] = cand["ori2"] = 1
cand["ord"] = ord("-")
st = hc.Settings(edge_threshold=0.97, ov_threshold=0.5)
a = run("2", reads, cand, st); b = run("coop", reads, cand, st)
for i in range(L):
    same = a[i]["x1"] == b[i]["x1"] and a[i]["mm"] == b[i]["mm"] and a[i]["n_cls"] == b[i]["n_cls"]
    if not same or i % 25 == 0:
        print("pos", i, "L", L - i, "same" if same else "DIFF", "lane mm,n", a[i]["mm"], result_n(a[i:i+1])[0], "coop", b[i]["mm"], result_n(b[i:i+1])[0])
