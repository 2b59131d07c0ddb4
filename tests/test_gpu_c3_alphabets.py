"""Config 3's size with the quality alphabets real reads have (round 6): the same 500 000 pairs and 10^8 candidates with
25 / 35 / 60 / 70 distinct quality values — the LG = 5 table (LDS-DMA form), the two wide 8-bit tables and the 16-bit-symbol kernel at the
size the north star's target is quoted on (with the skewed 35-value histogram of the reference's polyte/example reads: HC_TEST_C3Q35R=1).  Every launch scores all
10^8 candidates (the launch form of that size); 1.2 * 10^7 records of each are compared with the oracle bit for bit (x1, x2, mm, n, class,
score, mismatch rate: /root/reference/src/EdgeCalculator.cpp:92-101,106-138 restated in oracle/hc_oracle.c), the rest through the
size-independent invariants."""
import os

import numpy as np
import pytest

import haploconduct_amd as hc
from haploconduct_amd.records import result_cls, result_n

pytestmark = pytest.mark.gpu

CASES = [("c3q25", 25, "score_kernel_coop<uint8_t, 5, 1024"), ("c3q35", 35, "score_kernel_coop<uint8_t, 6, 768"),
         ("c3q60", 60, "score_kernel_coop<uint8_t, 7, 768"), ("c3q70", 70, "score_kernel_coop<uint16_t, 5, 1024")]
# (c3q35r — the 35 values drawn from the POLYTE example's histogram: same kernel as c3q35, indices dealt by frequency — is checked by bench.py's
# in-run parity on every default run: 2^20 records against the oracle and the digest of all 10^8; HC_TEST_C3Q35R=1 adds it here, +30 s)
if os.environ.get("HC_TEST_C3Q35R") == "1":
    CASES.append(("c3q35r", 35, "score_kernel_coop<uint8_t, 6, 768"))


@pytest.mark.parametrize("workload,nq,kernel", CASES)
def test_ten_million_records_of_every_alphabet_against_the_oracle(oracle, workload, nq, kernel):
    import bench

    reads, cand, cfg, st = bench.build_workload(workload, 0)
    assert cand.size == 100000000 and reads.n_reads == 500000 and cfg["quality_alphabet"] == nq
    with hc.EdgeScorer(st) as sc:
        sc.set_reads(reads)
        assert kernel in sc.kernel_info(cand.size), sc.kernel_info(cand.size)
        cd = sc.pack_cands(cand)
        res = sc.score_cands(cd)
        score, mrate, cls = sc.finalize(res)
    n, mm = result_n(res), res["mm"]
    assert (mm <= n).all() and (n >= 1).all() and (n <= 150).all()
    assert ((res["x1"] <= 0) & (res["x2"] <= 0)).all()
    assert (cls[mrate == 0] >= 2).all(), "merge_contigs=0 admits every zero-mismatch overlap (EdgeCalculator.cpp:407)"
    dev = result_cls(res)
    assert ((dev == cls) | (dev == 4)).all()
    threads = min(64, os.cpu_count() or 1)
    for lo in (0, 61000000):  # two stretches of 6 * 10^6
        hi = lo + 6000000
        ref = oracle.score_batch(reads, st, cand[lo:hi], n_threads=threads)
        where = f"{workload}: candidates [{lo}, {hi})"
        assert (ref["status"] == 0).all(), where
        assert np.array_equal(ref["x1"].view(np.uint64), res["x1"][lo:hi].view(np.uint64)), where
        assert np.array_equal(ref["x2"].view(np.uint64), res["x2"][lo:hi].view(np.uint64)), where
        assert np.array_equal(ref["n"], n[lo:hi]) and np.array_equal(ref["mm"], mm[lo:hi]), where
        assert np.array_equal(ref["cls"], cls[lo:hi]), where
        assert np.array_equal(ref["score"].view(np.uint64), score[lo:hi].view(np.uint64)), where
        assert np.array_equal(ref["mismatch_rate"].view(np.uint64), mrate[lo:hi].view(np.uint64)), where
        del ref
