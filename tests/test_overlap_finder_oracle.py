"""CPU checks around candidate generation: the brute-force oracle finds what was planted, the SFO text writer
and the SFO ingest agree on the record format (no device work here; the device finder is tested with -m gpu)."""
import os
import sys

import numpy as np
import pytest

import haploconduct_amd as hc
from haploconduct_amd import host
from haploconduct_amd.records import SFO_DTYPE

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import overlap_finder_oracle as O  # noqa: E402
import sfo2overlaps_oracle as S  # noqa: E402


def test_oracle_finds_planted_overlaps_in_both_orientations():
    rng = np.random.default_rng(1)
    g = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 400)]
    comp = np.zeros(256, np.uint8)
    comp[list(b"ACGTN")] = list(b"TGCAN")
    a, b, c = g[0:120], g[50:170], comp[g[100:220]][::-1]
    b = b.copy()
    b[20] = ord("A") if b[20] != ord("A") else ord("C")  # one substitution and one N inside the a/b overlap (genome 50..120),
    b[10] = ord("N")                                      # none inside the b/c overlap (genome 100..170)
    reads = hc.ReadSet.from_lists([(x.tobytes(), b"I" * x.size) for x in (a, b, c)], [])
    got = O.find_overlaps(reads, 0.0, 40)
    assert (0, 2, 100, 100, 20, 20, 0, 1) not in got  # 20 < min_overlap
    assert (1, 2, 50, 50, 70, 70, 0, 1) in got  # b = g[50:170] against rc(c) = g[100:220]: 70 exact positions
    assert not any(r[:2] == (0, 1) for r in got)
    assert not any(r[:2] == (0, 1) for r in O.find_overlaps(reads, 0.02, 40))  # floor(0.02 * 70) = 1 < 2
    got = O.find_overlaps(reads, 0.03, 40)
    assert (0, 1, 50, 50, 70, 70, 2, 0) in got  # the N matches nothing: 2 mismatches <= floor(0.03 * 70)
    assert all(r[7] == 0 for r in O.find_overlaps(reads, 0.03, 40, reversals=False))


def test_sfo_text_writer_roundtrip(tmp_path):
    recs = np.zeros(3, SFO_DTYPE)
    recs["idA"], recs["idB"] = [0, 1, 2], [5, 6, 7]
    recs["OHA"], recs["OHB"] = [10, -4, 0], [12, -3, 0]
    recs["OLA"] = recs["OLB"] = [100, 90, 150]
    recs["K"], recs["inverted"] = [0, 2, 1], [0, 1, 0]
    p = tmp_path / "x.sfo"
    host.write_sfo(str(p), recs)
    assert p.read_text() == "0\t5\tN\t10\t12\t100\t100\t0\n1\t6\tI\t-4\t-3\t90\t90\t2\n2\t7\tN\t0\t0\t150\t150\t1\n"
    # the ingest (native and oracle) reads that text: 8 singles, no pairs
    out = tmp_path / "ov.txt"
    n = host.sfo2overlaps(str(p), str(out), 8, 0)
    want = "".join(l + "\n" for l in S.sfo2overlaps(p.read_text().splitlines(), 8, 0))
    assert n == 3 and out.read_text() == want


def test_records_path_equals_text_path(tmp_path, monkeypatch):
    """hc_sfo_records_to_overlaps == hc_host_write_sfo + hc_sfo2overlaps, byte for byte (singles, pairs, both)."""
    monkeypatch.setenv("HC_SFO_TEXT_GENERAL", "1")  # the text side through the line-keeping path, not through the records path
    rng = np.random.default_rng(7)
    for ns, npairs in ((40, 0), (0, 30), (15, 20)):
        n_ids = ns + 2 * npairs
        n = 4000
        recs = np.zeros(n, SFO_DTYPE)
        a = rng.integers(0, n_ids, n)
        b = (a + 1 + rng.integers(0, n_ids - 1, n)) % n_ids
        recs["idA"], recs["idB"] = np.minimum(a, b), np.maximum(a, b)
        recs["OHA"], recs["OHB"] = rng.integers(-60, 60, n), rng.integers(-60, 60, n)
        recs["OLA"] = recs["OLB"] = rng.integers(40, 150, n)
        recs["K"], recs["inverted"] = rng.integers(0, 3, n), rng.integers(0, 2, n)
        recs = recs[rng.permutation(n)]  # any order: the ingest sorts
        host.write_sfo(str(tmp_path / "a.sfo"), recs)
        n_text = host.sfo2overlaps(str(tmp_path / "a.sfo"), str(tmp_path / "text.txt"), ns, npairs)
        n_rec = host.sfo_records_to_overlaps(recs, str(tmp_path / "rec.txt"), ns, npairs)
        assert n_text == n_rec > 0
        assert (tmp_path / "text.txt").read_bytes() == (tmp_path / "rec.txt").read_bytes()
        want = "".join(l + "\n" for l in S.sfo2overlaps((tmp_path / "a.sfo").read_text().splitlines(), ns, npairs))
        assert (tmp_path / "rec.txt").read_text() == want


@pytest.mark.parametrize("buckets", ["1", "7", "64"])
def test_bucketed_records_path_equals_text_path(tmp_path, monkeypatch, buckets):
    """The records path partitions by id0, sorts and matches the buckets independently and stitches the group that is
    open at a bucket border into the next bucket: with few reads and many buckets nearly every border carries one.
    Byte for byte the text path (hc_host_write_sfo + hc_sfo2overlaps) and the Python oracle of the script."""
    monkeypatch.setenv("HC_SFO_BUCKETS", buckets)
    monkeypatch.setenv("HC_SFO_TEXT_GENERAL", "1")  # the text side through the line-keeping path, not through the records path
    rng = np.random.default_rng(int(buckets))
    for ns, npairs, n in ((0, 25, 30000), (12, 18, 30000), (300, 0, 20000), (3, 400, 40000)):
        n_ids = ns + 2 * npairs
        recs = np.zeros(n, SFO_DTYPE)
        a = rng.integers(0, n_ids, n)
        b = (a + 1 + rng.integers(0, n_ids - 1, n)) % n_ids
        recs["idA"], recs["idB"] = a, b                      # both id orders: the flip of :112-122
        recs["OHA"], recs["OHB"] = rng.integers(-12, 120, n), rng.integers(-12, 12, n)   # few values: ties down to the text order
        recs["OLA"] = rng.integers(95, 105, n)
        recs["OLB"] = recs["OLA"] + rng.integers(0, 2, n)
        recs["K"], recs["inverted"] = rng.integers(0, 12, n), rng.integers(0, 2, n)      # K = 9 / 10 / 11: string order != numeric order
        recs[: n // 10] = recs[n // 2: n // 2 + n // 10]     # exact duplicates for uniq
        recs = recs[rng.permutation(n)]
        host.write_sfo(str(tmp_path / "a.sfo"), recs)
        n_text = host.sfo2overlaps(str(tmp_path / "a.sfo"), str(tmp_path / "text.txt"), ns, npairs)
        n_rec = host.sfo_records_to_overlaps(recs, str(tmp_path / "rec.txt"), ns, npairs)
        assert (tmp_path / "text.txt").read_bytes() == (tmp_path / "rec.txt").read_bytes()
        assert n_text == n_rec > 0
    want = "".join(l + "\n" for l in S.sfo2overlaps((tmp_path / "a.sfo").read_text().splitlines(), ns, npairs))
    assert (tmp_path / "rec.txt").read_text() == want
