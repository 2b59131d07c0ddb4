"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs.  Bit-exact bar: x1/x2 (the argument of the final exp), mismatch counts,
total_len, the 3-way admission class, and — after host finalisation with the same
libm — score and mismatch_rate as IEEE bit patterns."""
import numpy as np
import pytest

import haploconduct_amd as hc
from haploconduct_amd import synth
from haploconduct_amd.records import OVERLAP_DTYPE, result_cls, result_n

pytestmark = pytest.mark.gpu

HQ = np.array([20, 30, 37, 37, 37, 40, 40, 40, 40], dtype=np.uint8) + 33


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def check_parity(oracle, reads, settings, cand, expect_classes=None):
    ref = oracle.score_batch(reads, settings, cand)
    assert (ref["status"] == 0).all(), "oracle rejected an input the test meant to be valid"
    with hc.EdgeScorer(settings) as sc:
        sc.set_reads(reads)
        res = sc.score_batch(cand)
        score, mrate, cls = sc.finalize(res)
    assert np.array_equal(bits(res["x1"]), bits(ref["x1"])), "x1 (sub-overlap 1 mean log-prob) not bit-exact"
    x2_nan = np.isnan(ref["x2"])
    assert np.array_equal(np.isnan(res["x2"]), x2_nan)
    assert np.array_equal(bits(res["x2"])[~x2_nan], bits(ref["x2"])[~x2_nan]), "x2 not bit-exact"
    assert np.array_equal(res["mm"], ref["mm"]), "mismatch_count differs"
    assert np.array_equal(result_n(res), ref["n"]), "total_len differs"
    assert np.array_equal(cls, ref["cls"]), "admission class differs (host-finalised)"
    dev = result_cls(res)
    assert ((dev == ref["cls"]) | (dev == 4)).all(), "device-side admission class differs"
    assert (dev == 4).mean() < 1e-3, "guard band should be (nearly) empty"
    assert np.array_equal(bits(score), bits(ref["score"])), "score not bit-exact"
    assert np.array_equal(bits(mrate), bits(ref["mismatch_rate"])), "mismatch_rate not bit-exact"
    # the north star's tolerance, stated: |score - ref| <= 1e-6 (we are at 0)
    assert np.max(np.abs(score - ref["score"]), initial=0.0) <= 1e-6
    if expect_classes:
        present = set(np.unique(ref["cls"]).tolist())
        assert set(expect_classes) <= present, f"test data should exercise classes {expect_classes}, has {present}"
    return ref, res


def test_pp_all_orientations(oracle):
    reads, meta = synth.make_paired_dataset(1500, 3000, flip_frac=0.3, seed=21)
    reads.quals[:] = HQ[np.random.default_rng(5).integers(0, HQ.size, reads.quals.size)]
    cand = synth.paired_candidates(meta, n_candidates=20000, seed=22)
    assert set(np.unique(cand["ord"]).tolist()) == {ord("1"), ord("2")}
    assert (cand["ori1"] == 0).any() and (cand["ori2"] == 0).any()
    st = hc.Settings(edge_threshold=0.97, ov_threshold=0.9)
    check_parity(oracle, reads, st, cand, expect_classes=[0, 1, 2])


def test_pp_survey_quality_mix_merge_contigs(oracle):
    # SURVEY §8(d) quality set {2,...,40}: clause 1 almost never fires, clause 2 (0 mismatches) does
    reads, meta = synth.make_paired_dataset(1200, 2500, flip_frac=0.25, seed=31)
    cand = synth.paired_candidates(meta, n_candidates=15000, seed=32)
    for mc in (0.0, 0.01):
        st = hc.Settings(edge_threshold=0.97, merge_contigs=mc)
        ref, _ = check_parity(oracle, reads, st, cand)
        assert (ref["cls"] == 3).any()


def test_ss_mixed_lengths_both_orientations(oracle):
    reads, meta = synth.make_single_dataset(1500, 6000, len_lo=150, len_hi=1200, flip_frac=0.4, seed=41, quals=HQ,
                                            log_uniform=True)
    cand = synth.single_candidates(meta, min_overlap=60, n_candidates=20000)
    st = hc.Settings(edge_threshold=0.995, ov_threshold=0.9, min_overlap_len=100)
    ref, _ = check_parity(oracle, reads, st, cand, expect_classes=[0, 2])
    assert np.isnan(ref["x2"]).all()


def test_polyte_threshold_one(oracle):
    # POLYTE iterations run with --edge_threshold=1: admission is "zero mismatches" only (polyte.py:620)
    reads, meta = synth.make_single_dataset(800, 3000, len_lo=250, len_hi=250, flip_frac=0.5, seed=51, quals=HQ)
    cand = synth.single_candidates(meta, min_overlap=127)
    st = hc.Settings(edge_threshold=1.0, ov_threshold=0.9, merge_contigs=0.0, min_overlap_len=127)
    ref, _ = check_parity(oracle, reads, st, cand)
    assert not (ref["cls"] == 2).any() and (ref["cls"] == 3).any()


def _mixed_reads(seed, n_single=300, n_pair=300, glen=2500):
    """Singles (contig-like, 200-500 bp) and pairs (2x150) over one genome."""
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    genome = acgt[rng.integers(0, 4, glen)]
    singles, pairs, spos, ppos = [], [], [], []

    def noisy(seg):
        seg = seg.copy()
        k = rng.random(seg.size) < 0.004
        seg[k] = acgt[rng.integers(0, 4, int(k.sum()))]
        seg[rng.random(seg.size) < 0.002] = ord("N")
        return seg.tobytes()

    def q(n):
        return HQ[rng.integers(0, HQ.size, n)].tobytes()

    for _ in range(n_single):
        L = int(rng.integers(200, 500))
        s = int(rng.integers(0, glen - L))
        singles.append((noisy(genome[s:s + L]), q(L)))
        spos.append((s, L))
    for _ in range(n_pair):
        ins = int(rng.integers(350, 600))
        s = int(rng.integers(0, glen - ins))
        pairs.append(((noisy(genome[s:s + 150]), q(150)), (noisy(genome[s + ins - 150:s + ins]), q(150))))
        ppos.append((s, ins))
    return hc.ReadSet.from_lists(singles, pairs), spos, ppos


def test_sp_ps_mixed_types(oracle):
    reads, spos, ppos = _mixed_reads(61)
    ns = len(spos)
    rec = []
    # s-p: single A covers both mates of pair B.  p-s: /1 of pair A starts before single B ... enumerate by geometry
    for i, (s, L) in enumerate(spos):
        for j, (ps, ins) in enumerate(ppos):
            p1 = ps - s                  # /1 of the pair starts at p1 inside the single
            p2 = ps + ins - 150 - s      # /2 of the pair starts at p2 inside the single
            if 0 <= p1 < L - 40 and 0 <= p2 < L - 40:
                rec.append((i, ns + j, p1, p2, 1, 1, ord("-"), 2, min(L - p1, 150), min(L - p2, 150), 90))
            q1 = s - ps                  # single starts at q1 inside /1 of the pair
            q2 = ps + ins - 150 - s      # /2 of the pair starts at q2 inside the single
            if 0 <= q1 < 110 and 0 <= q2 < L - 40:
                rec.append((ns + j, i, q1, q2, 1, 1, ord("-"), 1, min(150 - q1, L), min(L - q2, 150), 90))
    cand = np.array(rec, dtype=OVERLAP_DTYPE)
    assert cand.size > 500
    # add wrong-orientation variants too (they score badly but must agree)
    flip = cand.copy()
    flip["ori1"] = 0
    flip2 = cand.copy()
    flip2["ori2"] = 0
    both = cand.copy()
    both["ori1"] = 0
    both["ori2"] = 0
    cand = np.concatenate([cand, flip[:400], flip2[:400], both[:400]])
    st = hc.Settings(edge_threshold=0.97, ov_threshold=0.5)
    ref, _ = check_parity(oracle, reads, st, cand, expect_classes=[0, 2])
    assert (ref["n_subs"] == 2).all()


def test_n_rich_q0_and_wide_alphabet_uint16_symbols(oracle):
    # > 32 distinct quality bytes forces 16-bit symbols; Q0 ('!') and 'N' runs as in savage/example read @2000
    rng = np.random.default_rng(71)
    reads, meta = synth.make_single_dataset(600, 2500, len_lo=200, len_hi=400, flip_frac=0.3, seed=72, n_rate=0.05)
    reads.quals[:] = (rng.integers(0, 60, reads.quals.size) + 33).astype(np.uint8)
    reads.quals[rng.random(reads.quals.size) < 0.05] = ord("!")
    cand = synth.single_candidates(meta, min_overlap=50, n_candidates=8000)
    st = hc.Settings(edge_threshold=0.5, ov_threshold=0.1)
    with hc.EdgeScorer(st) as sc:
        sc.set_reads(reads)
        assert sc.info()["qual_alphabet"] > 32
    check_parity(oracle, reads, st, cand)


def test_all_n_overlap_and_early_exits(oracle):
    singles = [("N" * 120, "I" * 120), ("ACGT" * 30, "I" * 120), ("ACGT" * 10, "5" * 40), ("ACGT" * 30, "I" * 120)]
    reads = hc.ReadSet.from_lists(singles)
    rows = [
        (0, 1, 0, 0, 1, 1, ord("-"), 0, 120, 0, 100),    # all-N: total_len == 0 -> score 0
        (1, 3, 0, 0, 1, 1, ord("-"), 0, 120, 0, 100),    # identical: perfect
        (1, 3, 4, 0, 1, 1, ord("-"), 0, 116, 0, 96),     # in phase: perfect
        (1, 3, 1, 0, 1, 1, ord("-"), 0, 119, 0, 99),     # out of phase: all mismatches
        (1, 3, 120, 0, 1, 1, ord("-"), 0, 1, 0, 1),      # pos == len: early exit (:76)
        (1, 3, 500, 0, 1, 1, ord("-"), 0, 1, 0, 1),      # pos > len
        (1, 2, 100, 0, 1, 1, ord("-"), 0, 20, 0, 50),    # short second read
        (2, 1, 0, 0, 0, 0, ord("-"), 0, 40, 0, 100),     # both reverse-complemented
    ]
    cand = np.array(rows, dtype=OVERLAP_DTYPE)
    for st in (hc.Settings(edge_threshold=0.9), hc.Settings(edge_threshold=0.9, min_read_len=100),
               hc.Settings(edge_threshold=0.9, merge_contigs=1.0), hc.Settings(edge_threshold=-1.0, ov_threshold=-1.0),
               hc.Settings(edge_threshold=0.0, ov_threshold=0.0)):
        check_parity(oracle, reads, st, cand)


def test_mismatch_setting_rejects(oracle):
    reads, meta = synth.make_single_dataset(500, 2000, len_lo=150, len_hi=300, flip_frac=0.3, seed=81, quals=HQ)
    cand = synth.single_candidates(meta, min_overlap=60, n_candidates=6000)
    for mm_setting in (1e-4, 0.05, 0.5):
        st = hc.Settings(edge_threshold=0.97, ov_threshold=0.5, mismatch=mm_setting)
        ref, _ = check_parity(oracle, reads, st, cand)
        assert np.isinf(ref["x1"]).any(), "--mismatch should reject some overlaps"


def test_threshold_sweep_decisions(oracle):
    reads, meta = synth.make_paired_dataset(800, 2000, flip_frac=0.2, seed=91)
    reads.quals[:] = HQ[np.random.default_rng(6).integers(0, HQ.size, reads.quals.size)]
    cand = synth.paired_candidates(meta, n_candidates=8000, seed=92)
    for et, ot, mc in ((0.97, 0.9, 0.0), (0.995, 0.9, 0.0), (1.0, 0.9, 0.0), (0.9, 0.95, 0.01), (0.99, 0.0, 0.0)):
        check_parity(oracle, reads, hc.Settings(edge_threshold=et, ov_threshold=ot, merge_contigs=mc), cand)


@pytest.mark.parametrize("n_quals", [3, 40, 55, 70])  # 8-bit symbols, the two wide 8-bit encodings (31..48 / 49..60 values), 16-bit symbols
def test_invalid_bases_and_quals_are_errors(oracle, n_quals):
    # lower-case bases in a PAIRED read abort the reference (EdgeCalculator.cpp:29-30); quality < '!' too (:61)
    filler = "".join(chr(33 + (i % n_quals)) for i in range(70))  # forces the size of the quality alphabet
    singles = [("ACGTACGTACGTACGTACGT", "IIIIIIIIIIIIIIIIIIII"), ("ACGTACGTacGTACGTACGT", "IIIIIIIIIIIIIIIIIIII"),
               ("ACGTACGTACGTACGTACGT", "IIIIIIII IIIIIIIIIII"), ("ACGTACGTACGTACGTACGT", "IIIIIIIIIIIIIIIIIIII"),
               ("ACGTNACGTN" * 7, filler), ("ACGTNACGTN" * 7, filler[::-1])]
    singles = [(s_, q_[: len(s_)]) for s_, q_ in singles]
    reads = hc.ReadSet.from_lists(singles)
    rows = [(0, 3, 0, 0, 1, 1, ord("-"), 0, 20, 0, 100), (0, 1, 0, 0, 1, 1, ord("-"), 0, 20, 0, 100),
            (0, 2, 0, 0, 1, 1, ord("-"), 0, 20, 0, 100), (0, 1, 12, 0, 1, 1, ord("-"), 0, 8, 0, 40),
            (3, 1, 0, 0, 1, 0, ord("-"), 0, 20, 0, 100), (4, 5, 0, 0, 1, 1, ord("-"), 0, 70, 0, 100),
            (4, 5, 3, 0, 1, 0, ord("-"), 0, 67, 0, 95), (5, 4, 10, 0, 0, 0, ord("-"), 0, 60, 0, 85)]
    cand = np.array(rows, dtype=OVERLAP_DTYPE)
    st = hc.Settings(edge_threshold=0.9, ov_threshold=0.1)
    with hc.EdgeScorer(st) as sc:
        sc.set_reads(reads)
        k = sc.info()["qual_alphabet"]
        assert (k <= 30) == (n_quals == 3) and (k > 60) == (n_quals == 70) and (48 < k <= 60) == (n_quals == 55)
        res = sc.score_batch(cand)
        cls = result_cls(res)
        # row 3 overlaps read 1 at positions 0..7 only (the lower-case bases sit at 8,9): valid
        assert cls[:5].tolist() == [2, 7, 7, 2, 7]
        with pytest.raises(hc.HcError):
            sc.finalize(res)
        ok = np.array([0, 3, 5, 6, 7])
        ref = oracle.score_batch(reads, st, cand[ok])
        assert (ref["status"] == 0).all()
        assert np.array_equal(res["x1"][ok].view(np.uint64), ref["x1"].view(np.uint64))
        assert np.array_equal(result_n(res)[ok], ref["n"]) and np.array_equal(res["mm"][ok], ref["mm"])


def test_api_errors_and_empty_batch():
    with hc.EdgeScorer(hc.Settings()) as sc:
        with pytest.raises(hc.HcError):
            sc.score_batch(np.zeros(1, dtype=OVERLAP_DTYPE))  # no reads yet
        reads = hc.ReadSet.from_lists([("ACGT", "IIII"), ("ACGT", "IIII")])
        sc.set_reads(reads)
        assert sc.score_batch(np.zeros(0, dtype=OVERLAP_DTYPE)).size == 0
        bad = np.zeros(2, dtype=OVERLAP_DTYPE)
        bad[0]["read1"], bad[0]["read2"] = 0, 0     # self overlap
        bad[1]["read1"], bad[1]["read2"] = 0, 9     # out of range
        assert result_cls(sc.score_batch(bad)).tolist() == [7, 7]
    empty = hc.ReadSet(np.zeros(0, np.uint8), np.zeros(0, np.uint8), np.zeros(2, np.uint64), np.array([0, 1], np.uint32),
                       np.zeros(1, np.uint64))
    with hc.EdgeScorer(hc.Settings()) as sc:
        with pytest.raises(hc.HcError):
            sc.set_reads(empty)  # empty sequence: FastqStorage exits (FastqStorage.cpp:143-146)


def test_device_resident_entry_matches_host_entry(oracle):
    import torch

    reads, meta = synth.make_paired_dataset(1000, 2500, seed=101)
    cand = synth.paired_candidates(meta, n_candidates=10000, seed=102)
    st = hc.Settings(edge_threshold=0.97)
    with hc.EdgeScorer(st) as sc:
        sc.set_reads(reads)
        host = sc.score_batch(cand)
        d_in = torch.from_numpy(cand.view(np.uint8).reshape(-1)).cuda()
        d_out = torch.empty(cand.size * 24, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        sc.score_batch_device(d_in.data_ptr(), cand.size, d_out.data_ptr())
        sc.synchronize()
        dev = d_out.cpu().numpy().view(hc.RESULT_DTYPE)
        assert np.array_equal(dev.view(np.uint8), host.view(np.uint8))
        pos, subs = sc.count_positions_device(d_in.data_ptr(), cand.size)
    ref = oracle.score_batch(reads, st, cand)
    assert pos == int(ref["positions"].sum()) and subs == int(ref["n_subs"].sum())


def test_reorder_modes_give_identical_records(oracle):
    """hc_set_reorder: results never depend on the candidate order or on the reorder policy."""
    import torch

    reads, meta = synth.make_paired_dataset(3000, 4000, flip_frac=0.2, seed=111)
    cand = synth.paired_candidates(meta, n_candidates=60000, seed=112)
    shuf = cand[np.random.default_rng(1).permutation(cand.size)]
    st = hc.Settings(edge_threshold=0.97)
    ref = oracle.score_batch(reads, st, shuf)
    outs = []
    with hc.EdgeScorer(st) as sc:
        sc.set_reads(reads)
        for mode in (0, 1, 2):
            sc.set_reorder(mode)
            outs.append(sc.score_batch(shuf))          # host entry: AUTO probes and reorders the shuffled batch
            d_in = torch.from_numpy(shuf.view(np.uint8).reshape(-1)).cuda()
            d_out = torch.zeros(shuf.size * 24, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            sc.score_batch_device(d_in.data_ptr(), shuf.size, d_out.data_ptr())
            sc.synchronize()
            outs.append(d_out.cpu().numpy().view(hc.RESULT_DTYPE))
    for o in outs:
        assert o.tobytes() == outs[0].tobytes()
    assert np.array_equal(outs[0]["x1"].view(np.uint64), ref["x1"].view(np.uint64))
    assert np.array_equal(result_n(outs[0]), ref["n"])


def test_compaction_entry_points(oracle):
    import torch

    reads, meta = synth.make_paired_dataset(2000, 3000, flip_frac=0.2, seed=121)
    reads.quals[:] = HQ[np.random.default_rng(2).integers(0, HQ.size, reads.quals.size)]
    cand = synth.paired_candidates(meta, n_candidates=50000, seed=122)
    st = hc.Settings(edge_threshold=0.97, ov_threshold=0.9)
    with hc.EdgeScorer(st) as sc:
        sc.set_reads(reads)
        full = sc.score_batch(cand)
        idx, res = sc.score_batch_compact(cand)
        keep = np.nonzero(result_cls(full) != 0)[0]
        assert 0 < keep.size < cand.size
        assert np.array_equal(idx, keep) and res.tobytes() == full[keep].tobytes()
        # device-resident form
        d_in = torch.from_numpy(cand.view(np.uint8).reshape(-1)).cuda()
        d_out = torch.empty(cand.size * 24, dtype=torch.uint8, device="cuda")
        d_idx = torch.zeros(cand.size, dtype=torch.int32, device="cuda")
        d_cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        sc.score_batch_device(d_in.data_ptr(), cand.size, d_out.data_ptr())
        sc.compact_device(d_out.data_ptr(), cand.size, d_idx.data_ptr(), d_cnt.data_ptr())
        sc.synchronize()
        k = int(d_cnt.item())
        assert k == keep.size and np.array_equal(d_idx[:k].cpu().numpy().view(np.uint32), keep)
        empty_idx, empty_res = sc.score_batch_compact(cand[:0])
        assert empty_idx.size == 0


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("HC_FUZZ_SEEDS", "40"))))
def test_fuzz_random_read_sets_settings_and_geometry(oracle, seed):
    """Everything at once, seeded: singles / pairs / both, sequence lengths from 1 to a few hundred, quality
    alphabets of 1..70 symbols (all three symbol encodings), N runs, random orientations and ord, positions
    from the true geometry and at random (including pos >= length), random thresholds / --mismatch /
    min_read_len / merge_contigs.  The HIP path must reproduce the oracle bit for bit on all of it."""
    rng = np.random.default_rng(9000 + seed)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    glen = int(rng.integers(400, 3000))
    genome = acgt[rng.integers(0, 4, glen)]
    K = int(rng.choice([1, 2, 5, 9, 20, 29, 30, 31, 40, 47, 48, 49, 60, 70]))
    alphabet = (rng.choice(np.arange(33, 127), size=min(K, 94), replace=False)).astype(np.uint8)
    err, nrate = float(rng.choice([0.0, 0.003, 0.02])), float(rng.choice([0.0, 0.002, 0.05]))
    lo, hi = (1, 40) if seed % 5 == 0 else (30, int(rng.integers(120, 700)))
    if glen < 3 * hi + 300:  # room for a pair with its insert
        glen = 3 * hi + 300
        genome = acgt[rng.integers(0, 4, glen)]

    def piece(s, L, rc):
        seg = genome[s:s + L].copy()
        k = rng.random(L) < err
        seg[k] = acgt[rng.integers(0, 4, int(k.sum()))]
        if nrate:
            seg[rng.random(L) < nrate] = ord("N")
            if L > 20 and rng.random() < 0.3:
                a = int(rng.integers(0, L - 10))
                seg[a:a + int(rng.integers(1, 10))] = ord("N")
        if rc:  # stored reverse-complemented: the candidate then carries ori '-'
            comp = np.zeros(256, np.uint8)
            comp[list(b"ACGTN")] = list(b"TGCAN")
            seg = comp[seg][::-1]
        return seg.tobytes(), alphabet[rng.integers(0, alphabet.size, L)].tobytes()

    mode = seed % 3  # 0 singles, 1 pairs, 2 both
    singles, pairs, geo = [], [], []  # geo: (start, len1, start2, len2, rc)
    n_s = 0 if mode == 1 else int(rng.integers(20, 120))
    n_p = 0 if mode == 0 else int(rng.integers(20, 120))
    for _ in range(n_s):
        L = int(rng.integers(lo, hi))
        s = int(rng.integers(0, glen - L))
        rc = bool(rng.random() < 0.3)
        singles.append(piece(s, L, rc))
        geo.append((s, L, 0, 0, rc))
    for _ in range(n_p):
        L1, L2 = int(rng.integers(lo, hi)), int(rng.integers(lo, hi))
        ins = L1 + L2 + int(rng.integers(0, 200))
        s = int(rng.integers(0, max(1, glen - ins)))
        s2 = min(s + ins - L2, glen - L2)
        pairs.append((piece(s, L1, False), piece(s2, L2, False)))
        geo.append((s, L1, s2, L2, False))
    reads = hc.ReadSet.from_lists(singles, pairs)
    n = reads.n_reads
    m = 1500
    cand = np.zeros(m, OVERLAP_DTYPE)
    a = rng.integers(0, n, m)
    b = (a + 1 + rng.integers(0, n - 1, m)) % n
    cand["read1"], cand["read2"] = a, b
    for i in range(m):
        ga, gb = geo[a[i]], geo[b[i]]
        pa, pb = a[i] >= n_s, b[i] >= n_s
        true_geo = rng.random() < 0.6
        cand["pos1"][i] = max(0, gb[0] - ga[0]) if true_geo else int(rng.integers(0, hi + 20))
        cand["pos2"][i] = (max(0, gb[2] - ga[2]) if true_geo else int(rng.integers(0, hi + 20))) if (pa or pb) else 0
        cand["ori1"][i] = (0 if ga[4] else 1) if true_geo else int(rng.integers(0, 2))
        cand["ori2"][i] = (0 if gb[4] else 1) if true_geo else int(rng.integers(0, 2))
        cand["ord"][i] = ord("12"[int(rng.integers(2))]) if (pa and pb) else ord("-")
        cand["flags"][i] = int(pa) | (int(pb) << 1)
    cand["len1"], cand["len2"], cand["perc"] = rng.integers(1, 300, m), rng.integers(0, 300, m), rng.integers(0, 101, m)
    st = hc.Settings(edge_threshold=float(rng.choice([0.5, 0.9, 0.97, 0.995, 1.0])), ov_threshold=float(rng.choice([0.0, 0.3, 0.9])),
                     merge_contigs=float(rng.choice([0.0, 0.0, 0.01, 0.2])), mismatch=float(rng.choice([0.0, 0.0, 1e-4, 0.05])),
                     min_read_len=int(rng.choice([0, 0, 20, 60])))
    with hc.EdgeScorer(st) as sc:
        sc.set_reads(reads)
        assert sc.info()["qual_alphabet"] == np.unique(reads.quals).size
    check_parity(oracle, reads, st, cand)


@pytest.mark.parametrize("seed", [1, 2, 4, 7, 10, 13, 16, 22, 31, 37])
def test_fuzz_through_the_lds_dma_form(oracle, monkeypatch, seed):
    """The LDS-DMA form of the cooperative fetch serves launches of 5 * 10^5 candidates and more; here it is made to take
    the fuzz scenarios' 1 500 (read sets of equal-length sequences only: mixed lengths take the bucketed launch)."""
    monkeypatch.setenv("HC_COOP_DMA_MIN", "1")
    monkeypatch.setenv("HC_BALANCE", "0")  # every read set through the plain launch, whatever its lengths
    test_fuzz_random_read_sets_settings_and_geometry(oracle, seed)


@pytest.mark.parametrize("order", ["frequency", "value"])
@pytest.mark.parametrize("form", ["dma", "registers"])
@pytest.mark.parametrize("n_quals", [31, 40, 48, 49, 60])
def test_the_wide_table_in_both_forms_and_both_index_orders(oracle, monkeypatch, n_quals, form, order):
    """31..48 / 49..60 quality values take the wide 8-bit encodings (hc_device.h): the quality indices dealt by frequency (kWideRankLabel) or in byte
    order (HC_QIDX_ORDER=value), the table read by the 768-lane LDS-DMA form (whose scratch words sit in an unaddressed row of the table)
    or by the register-staged 1 024-lane form (HC_WIDE_DMA=0).  All four must reproduce the oracle, with and without invalid symbols."""
    monkeypatch.setenv("HC_COOP_DMA_MIN", "1")
    monkeypatch.setenv("HC_WIDE_DMA", "1" if form == "dma" else "0")
    if order == "value":
        monkeypatch.setenv("HC_QIDX_ORDER", "value")
    test_fetch_and_descriptor_paths_agree(oracle, monkeypatch, "coop", "1", n_quals)
    with hc.EdgeScorer(hc.Settings()) as sc:
        from haploconduct_amd import synth
        reads, _ = synth.make_paired_dataset(n_pairs=300, genome_len=3000, seed=5, quals=np.arange(40, 40 + n_quals, dtype=np.uint8))
        sc.set_reads(reads)
        lg = "6" if n_quals <= 48 else "7"
        want = lg + (", 768, true, false, 0, true" if form == "dma" else ", 1024, true, false, 1, true")
        assert sc.kernel_info(10 ** 6).startswith("hc::score_kernel_coop<uint8_t, " + want), sc.kernel_info(10 ** 6)
    test_invalid_bases_and_quals_are_errors(oracle, 40 if n_quals <= 48 else 55)


@pytest.mark.parametrize("align", ["16", "64", "256"])
@pytest.mark.parametrize("seed", [1, 2, 3, 7, 10])
def test_slot_alignment_of_the_store_changes_nothing(oracle, monkeypatch, align, seed):
    """hc_set_reads starts every read slot on a 128-byte line while the store stays inside the Infinity Cache and packs the slots
    (16 bytes) beyond that; HC_SLOT_ALIGN forces a stride.  The fuzz scenarios (regular and irregular stores, all three encodings)
    under the packed layout the 10^8-candidate set gets, and under two more."""
    monkeypatch.setenv("HC_SLOT_ALIGN", align)
    test_fuzz_random_read_sets_settings_and_geometry(oracle, seed)


@pytest.mark.parametrize("fetch", ["coop", "4", "2"])
@pytest.mark.parametrize("regular", ["1", "0"])
@pytest.mark.parametrize("n_quals", [5, 12, 25, 40, 60, 75])  # the three dense / sparse 8-bit tables, the two wide 8-bit encodings (per lane: 512-lane workgroups), 16-bit symbols
def test_fetch_and_descriptor_paths_agree(oracle, monkeypatch, fetch, regular, n_quals):
    """A store of equal-length sequences, singles first ("regular": read descriptors by arithmetic) scored with the
    cooperative fetch and with both per-lane fetch groups, each with and without descriptor look-ups: every combination
    must reproduce the oracle.  Singles, pairs, all four type combinations, window lengths 1..150, random orientations."""
    monkeypatch.setenv("HC_FETCH_GROUP", fetch)
    monkeypatch.setenv("HC_REGULAR_STORE", regular)
    rng = np.random.default_rng(77 + n_quals)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    genome = acgt[rng.integers(0, 4, 4000)]
    alphabet = rng.choice(np.arange(33, 127), size=n_quals, replace=False).astype(np.uint8)
    L = 150

    def piece(s):
        seg = genome[s:s + L].copy()
        k = rng.random(L) < 0.004
        seg[k] = acgt[rng.integers(0, 4, int(k.sum()))]
        seg[rng.random(L) < 0.003] = ord("N")
        return seg.tobytes(), alphabet[rng.integers(0, n_quals, L)].tobytes()

    n_s, n_p = 150, 250
    starts = rng.integers(0, 4000 - 600, n_s + n_p)
    ins = rng.integers(300, 600, n_s + n_p)
    singles = [piece(int(starts[i])) for i in range(n_s)]
    pairs = [(piece(int(starts[i])), piece(int(starts[i] + ins[i] - L))) for i in range(n_s, n_s + n_p)]
    reads = hc.ReadSet.from_lists(singles, pairs)
    n, m = reads.n_reads, 6000
    cand = np.zeros(m, OVERLAP_DTYPE)
    a = rng.integers(0, n, m)
    b = (a + 1 + rng.integers(0, n - 1, m)) % n
    cand["read1"], cand["read2"] = a, b
    pa, pb = a >= n_s, b >= n_s
    true_geo = rng.random(m) < 0.7
    d1 = starts[b] - starts[a]
    d2 = (starts[b] + ins[b]) - (starts[a] + ins[a])
    cand["pos1"] = np.where(true_geo & (d1 >= 0) & (d1 < L), d1, rng.integers(0, L + 10, m))
    cand["pos2"] = np.where(pa | pb, np.where(true_geo & (d2 >= 0) & (d2 < L), d2, rng.integers(0, L + 10, m)), 0)
    cand["ori1"] = np.where(true_geo, 1, rng.integers(0, 2, m))
    cand["ori2"] = np.where(true_geo, 1, rng.integers(0, 2, m))
    cand["ord"] = np.where(pa & pb, np.where(rng.random(m) < 0.5, ord("1"), ord("2")), ord("-"))
    cand["flags"] = pa.astype(np.uint8) | (pb.astype(np.uint8) << 1)
    cand["len1"], cand["len2"], cand["perc"] = 100, 100, 90
    st = hc.Settings(edge_threshold=0.97, ov_threshold=0.5, min_read_len=0)
    with hc.EdgeScorer(st) as sc:
        sc.set_reads(reads)
        assert sc.info()["qual_alphabet"] == n_quals
    check_parity(oracle, reads, st, cand)


@pytest.mark.gpu
def test_unordered_batches_of_growing_size_through_the_compact_path():
    """Regression: an unordered batch takes the device re-ordering (its scratch grows with the batch) inside
    hc_score_batch_compact, after the compaction scratch was sized — growing the one must not disturb the other.
    Shuffled candidates in batches of increasing size; indices and records must be those of the plain call."""
    from haploconduct_amd import synth

    reads, meta = synth.make_paired_dataset(20000, 30000, seed=71)
    cand = synth.paired_candidates(meta, n_candidates=None, seed=5)
    rng = np.random.default_rng(3)
    cand = cand[rng.permutation(cand.size)]
    assert cand.size > 480000
    st = hc.Settings(edge_threshold=0.97, ov_threshold=0.9, min_overlap_len=0)
    with hc.EdgeScorer(st) as sc:
        sc.set_reads(reads)
        at = 0
        for n in (5000, 90000, 150000, 230000):
            batch = np.ascontiguousarray(cand[at:at + n])
            at += n
            idx, res = sc.score_batch_compact(batch)
            full = sc.score_batch(batch)
            keep = np.nonzero((full["n_cls"] >> 28) != 0)[0]
            assert np.array_equal(idx, keep.astype(idx.dtype)), n
            assert res.tobytes() == full[keep].tobytes(), n
