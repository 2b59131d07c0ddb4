"""Which scoring kernel a read set takes (hc_set_reads decides by the set's shape, launch_score by the launch's size), asserted by
symbol, with the results compared bit for bit with the oracle and with the kernel the library did NOT pick.  The shapes and the
measurements behind the rule: profiles/r04_dispatch.txt, r04_dispatch_pairs.txt (the SAVAGE example's merged
singles of 400..500 bp are one of them — the cooperative kernel is the faster one there, 0.265 against 0.304 ms)."""
import gzip
import os

import numpy as np
import pytest

import haploconduct_amd as hc
from haploconduct_amd import host, synth
from haploconduct_amd.records import result_cls, result_n

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _check(oracle, reads, cand, st, expect, monkeypatch, other_env):
    with hc.EdgeScorer(st) as sc:
        sc.set_reads(reads)
        info_small, info_large = sc.kernel_info(cand.size), sc.kernel_info(10 ** 8)
        for piece in expect:
            assert piece in info_small or piece in info_large, f"{piece!r} not in the kernels the library picked: {info_small} / {info_large}"
        res = sc.score_batch(cand)
        score, mrate, cls = sc.finalize(res)
    ref = oracle.score_batch(reads, st, cand, n_threads=min(32, os.cpu_count() or 1))
    assert (ref["status"] == 0).all()
    assert np.array_equal(ref["x1"].view(np.uint64), res["x1"].view(np.uint64)) and np.array_equal(ref["x2"].view(np.uint64), res["x2"].view(np.uint64))
    assert np.array_equal(ref["n"], result_n(res)) and np.array_equal(ref["mm"], res["mm"]) and np.array_equal(ref["cls"], cls)
    assert np.array_equal(ref["score"].view(np.uint64), score.view(np.uint64))
    for k, v in other_env.items():
        monkeypatch.setenv(k, v)
    with hc.EdgeScorer(st) as sc:  # the kernel the rule did not pick: same records
        sc.set_reads(reads)
        assert sc.kernel_info(cand.size) != info_small
        assert sc.score_batch(cand).tobytes() == res.tobytes()
    return info_small, info_large


def test_savage_example_shape_takes_the_cooperative_kernel(oracle, monkeypatch):
    reads, meta = synth.make_single_dataset(20000, 60000, len_lo=400, len_hi=500, n_strains=3, divergence=0.01, flip_frac=0.5, seed=8)
    cand = synth.single_candidates(meta, min_overlap=200, n_candidates=300000)
    st = hc.Settings(edge_threshold=0.97, min_overlap_len=200)
    small, large = _check(oracle, reads, cand, st, ["hc::score_kernel_coop<uint8_t, 3, 1024, true, false, 0, true>"], monkeypatch, {"HC_FETCH_GROUP": "2"})
    # 300 000 candidates and 10^8 alike: LDS-DMA rows, one workgroup per CU, the waves take their items by ticket; below 150 000: 256-lane workgroups
    assert small.split(" encoding=")[0] == large.split(" encoding=")[0] == "hc::score_kernel_coop<uint8_t, 3, 1024, true, false, 0, true>"
    assert "length-bucketed" not in small
    monkeypatch.delenv("HC_FETCH_GROUP")  # (_check left it set for its second scorer)
    with hc.EdgeScorer(st) as sc:
        sc.set_reads(reads)
        assert sc.kernel_info(100000).startswith("hc::score_kernel_coop<uint8_t, 3, 256, true, false, 1>")


def test_the_savage_example_reads_themselves_take_the_cooperative_kernel(tmp_path):
    paths = {}
    for name in ("savage_singles", "savage_paired1", "savage_paired2"):
        paths[name] = str(tmp_path / (name + ".fastq"))
        with gzip.open(os.path.join(HERE, "golden", name + ".fastq.gz"), "rb") as f, open(paths[name], "wb") as o:
            o.write(f.read())
    f = host.Fastq(singles=paths["savage_singles"], paired1=paths["savage_paired1"], paired2=paths["savage_paired2"])
    with hc.EdgeScorer(hc.Settings()) as sc:
        sc.set_reads(f.readset())
        info = sc.kernel_info(88289)  # the example's overlap lines
    assert info.startswith("hc::score_kernel_coop<uint8_t, 5, 256, true, false, 1>"), info  # 25 quality values: the 32 x 32 table planes


def test_short_singles_of_mixed_length_take_the_per_lane_kernel(oracle, monkeypatch):
    reads, meta = synth.make_single_dataset(30000, 80000, len_lo=100, len_hi=400, n_strains=3, divergence=0.01, flip_frac=0.5, seed=6, log_uniform=True)
    cand = synth.single_candidates(meta, min_overlap=60, n_candidates=300000)
    st = hc.Settings(edge_threshold=0.995, min_overlap_len=60)
    small, large = _check(oracle, reads, cand, st, ["hc::score_kernel<uint8_t, 2, 3, false>"], monkeypatch, {"HC_FETCH_GROUP": "coop"})
    assert small == large  # one lane, one fetch at every size


def test_trimmed_pairs_take_the_cooperative_kernel(oracle, monkeypatch):
    reads, meta = synth.make_paired_dataset(12000, 11000, seed=1, trim_lo=60)
    cand = synth.paired_candidates(meta, min_len=50, seed=2)[:300000]
    assert cand.size > 100000
    st = hc.Settings(edge_threshold=0.97, min_overlap_len=100)
    _check(oracle, reads, cand, st, ["hc::score_kernel_coop<uint8_t, 3, 1024, true, false, 0, true>"], monkeypatch, {"HC_FETCH_GROUP": "4"})


def test_contig_length_mixed_singles_take_the_bucketed_launch(oracle, monkeypatch):
    reads, meta = synth.make_single_dataset(8000, 60000, len_lo=150, len_hi=3000, n_strains=3, divergence=0.01, flip_frac=0.5, seed=5, log_uniform=True)
    cand = synth.single_candidates(meta, min_overlap=100, n_candidates=200000)
    st = hc.Settings(edge_threshold=0.995, min_overlap_len=100)
    small, _ = _check(oracle, reads, cand, st, ["hc::score_kernel_coop<uint8_t, 3, 256, true, true, 2, true>", "length-bucketed"], monkeypatch, {"HC_BALANCE": "0"})


def test_a_few_short_outliers_do_not_flip_the_kernel_of_a_uniform_set(oracle, monkeypatch):
    """the dispatch went by the shortest and the longest sequence, so ONE short read (routine after quality trimming) sent an
    otherwise uniform 250-bp set to the per-lane kernel.  It goes by the 5th / 95th percentile of the lengths now: 1 % of 60-bp reads among
    250-bp singles keep the cooperative kernel (and the results are the oracle's, and the per-lane kernel's)."""
    reads, meta = synth.make_single_dataset(20000, 60000, len_lo=250, len_hi=250, n_strains=2, divergence=0.001, flip_frac=0.5, seed=14)
    # cut 1 % of the reads down to 60 bp (the read set is rebuilt: sequences of two lengths)
    rng = np.random.default_rng(3)
    short = set(rng.choice(reads.n_reads, reads.n_reads // 100, replace=False).tolist())
    singles = []
    for r in range(reads.n_reads):
        b, q = reads.seq(r)
        singles.append((b[:60], q[:60]) if r in short else (b, q))
    reads2 = hc.ReadSet.from_lists(singles, [])
    cand = synth.single_candidates(meta, min_overlap=127, n_candidates=300000)
    cand = cand[~np.isin(cand["read1"], list(short)) | (cand["pos1"] < 40)]  # (a short read 1 overlaps only near its start)
    st = hc.Settings(edge_threshold=1.0, min_overlap_len=0)
    small, large = _check(oracle, reads2, cand, st, ["hc::score_kernel_coop<uint8_t, 3, "], monkeypatch, {"HC_FETCH_GROUP": "2"})
    assert "hc::score_kernel<" not in small and "length-bucketed" not in small
