"""bench.py's synthetic workloads (CPU): the q<K>[r] variants of a paired configuration — K distinct quality values, uniform or drawn from the
histogram of the reference's example reads (tests/golden/quality_histograms.json) — keep the configuration's bases, fragments and
candidates; only the quality bytes change."""
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_quality_histograms_fixture():
    h = json.load(open(os.path.join(ROOT, "tests", "golden", "quality_histograms.json")))
    assert {v["distinct"] for v in h.values()} >= {25, 35}
    for v in h.values():
        assert v["distinct"] == len(v["counts"]) and sum(v["counts"].values()) == v["bases"]
        assert all(33 <= int(b) <= 126 for b in v["counts"])  # Phred + 33 (src/EdgeCalculator.cpp:92-101 accepts [33, 127])


@pytest.mark.parametrize("name,k", [("c2-smallq25", 25), ("c2-smallq35", 35), ("c2-smallq60", 60), ("c2-smallq35r", 35), ("c2-smallq25r", 25)])
def test_alphabet_variants_keep_the_configurations_candidates(name, k):
    import bench

    base_reads, base_cand, _, base_st = bench.build_workload("c2-small", 0)
    reads, cand, cfg, st = bench.build_workload(name, 0)
    assert cfg["quality_alphabet"] == k == np.unique(reads.quals).size
    assert np.array_equal(reads.bases, base_reads.bases) and np.array_equal(reads.seq_off, base_reads.seq_off)
    assert cand.tobytes() == base_cand.tobytes()
    assert st.edge_threshold == base_st.edge_threshold
    if name.endswith("r"):  # skewed: the most frequent value covers far more than 1 / K of the bases
        top = np.bincount(reads.quals).max() / reads.quals.size
        assert top > 3.0 / k


def test_unknown_workloads_are_refused():
    import bench

    for bad in ("c3q", "c3q17r", "c9", "c2q0"):
        with pytest.raises(SystemExit):
            bench.build_workload(bad, 0)
