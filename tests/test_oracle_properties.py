"""Known-answer and property tests of the CPU oracle, independent of the reference build:
closed-form values computed with Python's math (same libm), symmetry/linearity properties
the algorithm has (reference src/EdgeCalculator.cpp:26-139), and the classification rule."""
import math

import numpy as np

import haploconduct_amd as hc
from haploconduct_amd import synth
from haploconduct_amd.records import OVERLAP_DTYPE


def test_known_answer_uniform_quality_perfect_match(oracle):
    # all positions match at Q40/Q40: every term is log((1-p)^2 + p^2/3); n terms summed in order
    p = math.pow(10, -40 / 10.0)
    term = math.log((1 - p) * (1 - p) + (p * p) / 3.0)
    for n in (1, 7, 150):
        s = 0.0
        for _ in range(n):
            s += term
        want = math.exp((1.0 / n) * s)
        r = oracle.overlap_score(b"ACGT" * 50, (b"ACGT" * 50)[:n], b"I" * 200, b"I" * n, 0)
        assert r["score"] == want and r["n"] == n and r["mm"] == 0 and r["mismatch_rate"] == 0.0


def test_known_answer_single_mismatch(oracle):
    p1, p2 = math.pow(10, -30 / 10.0), math.pow(10, -20 / 10.0)
    pm = p1 * (1 - p2) / 3.0 + p2 * (1 - p1) / 3.0 + (2 / 9.0) * p1 * p2
    r = oracle.overlap_score(b"A", b"C", b"?", b"5", 0)
    assert r["score"] == math.exp((1.0 / 1.0) * (0.0 + math.log(pm))) and r["mm"] == 1 and r["mismatch_rate"] == 1.0


def test_n_positions_are_ignored(oracle):
    a = oracle.overlap_score(b"ACGTNACGT", b"ACGTAACGT", b"IIII!IIII", b"IIIIIIIII", 0)
    b = oracle.overlap_score(b"ACGTACGT", b"ACGTACGT", b"IIIIIIII", b"IIIIIIII", 0)
    assert a["n"] == 8 and a["score"] == b["score"]
    z = oracle.overlap_score(b"NNNN", b"ACGT", b"IIII", b"IIII", 0)
    assert z["score"] == 0 and z["mismatch_rate"] == 1.0


def test_early_exits_leave_mismatch_rate_one(oracle):
    assert oracle.overlap_score(b"ACGT", b"ACGT", b"IIII", b"IIII", 4)["score"] == 0            # pos >= len (:76)
    assert oracle.overlap_score(b"ACGT", b"ACGT", b"IIII", b"IIII", 0, min_read_len=5)["score"] == 0  # (:82)
    r = oracle.overlap_score(b"ACGT", b"ACGA", b"IIII", b"IIII", 0, mismatch=0.5)             # (:125-127)
    assert r["score"] == 0 and r["mismatch_rate"] == 1.0


def test_swapping_roles_at_pos0_same_length_gives_same_match_terms(oracle):
    # match probability is symmetric in (p1, p2) bit-for-bit; with zero mismatches A-vs-B == B-vs-A
    rng = np.random.default_rng(3)
    s = bytes(rng.choice(list(b"ACGT"), 120).tolist())
    q1 = bytes((rng.integers(0, 41, 120) + 33).astype(np.uint8).tolist())
    q2 = bytes((rng.integers(0, 41, 120) + 33).astype(np.uint8).tolist())
    assert oracle.overlap_score(s, s, q1, q2, 0)["score"] == oracle.overlap_score(s, s, q2, q1, 0)["score"]


def test_classification_rule(oracle):
    # EdgeCalculator.cpp:404-413 on a perfect, a one-mismatch and a bad s-s overlap
    good = "ACGTTGCAAGCTTAGGCATCGATCGGATCCTAGACGTTGCAAGCTTAGGCATCGATCGGATCCTAG" * 2
    one = good[:40] + ("A" if good[40] != "A" else "C") + good[41:]
    bad = "T" * len(good)
    q = "I" * len(good)
    reads = hc.ReadSet.from_lists([(good, q), (good, q), (one, q), (bad, q)])
    cand = np.array([(0, 1, 0, 0, 1, 1, ord("-"), 0, len(good), 0, 100), (0, 2, 0, 0, 1, 1, ord("-"), 0, len(good), 0, 100),
                     (0, 3, 0, 0, 1, 1, ord("-"), 0, len(good), 0, 100)], dtype=OVERLAP_DTYPE)
    r = oracle.score_batch(reads, hc.Settings(edge_threshold=0.97, ov_threshold=0.9), cand)
    assert r["cls"].tolist() == [2, 1, 0]
    r = oracle.score_batch(reads, hc.Settings(edge_threshold=1.0, ov_threshold=0.9, merge_contigs=0.0), cand)
    assert r["cls"].tolist() == [3, 1, 0]  # zero mismatches admitted by the merge_contigs clause (:407)
    r = oracle.score_batch(reads, hc.Settings(edge_threshold=0.97, ov_threshold=0.9, merge_contigs=0.01), cand)
    assert r["cls"].tolist() == [2, 3, 0]


def test_pp_combination_rule(oracle):
    # both sub-overlaps above the threshold -> mean; otherwise min (:254-261)
    reads, meta = synth.make_paired_dataset(300, 1000, seed=9)
    reads.quals[:] = ord("I")
    cand = synth.paired_candidates(meta, n_candidates=2000, seed=10)
    st = hc.Settings(edge_threshold=0.97)
    r = oracle.score_batch(reads, st, cand)
    both = (r["ov1"] > 0.97) & (r["ov2"] > 0.97)
    assert both.any() and (~both).any()
    assert np.array_equal(r["score"][both], 0.5 * (r["ov1"][both] + r["ov2"][both]))
    assert np.array_equal(r["score"][~both], np.minimum(r["ov1"][~both], r["ov2"][~both]))
    assert (r["mismatch_rate"] == np.maximum(r["ov1"] * 0 + (r["mm"] / r["n"]).astype(np.float32) * 0 + r["mismatch_rate"], 0)).all()


def test_openmp_batch_equals_serial(oracle):
    reads, meta = synth.make_paired_dataset(300, 1000, flip_frac=0.3, seed=12)
    cand = synth.paired_candidates(meta, n_candidates=3000, seed=13)
    st = hc.Settings(edge_threshold=0.97)
    a = oracle.score_batch(reads, st, cand, n_threads=1)
    b = oracle.score_batch(reads, st, cand, n_threads=4)
    assert a.tobytes() == b.tobytes()
