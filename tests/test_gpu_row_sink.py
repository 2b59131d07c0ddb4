"""The row sink of hc_score_pack_device (hc_kernels.hip: RowSink, per-workgroup segments + spill area) and the context's scratch
under launches on two streams.

Contract (include/hcedge.h): row 0 of the payload counts the kept rows EXACTLY; a count above cap means, and only means, that
there were more kept rows than cap.  A launch whose kept rows all sit in one stretch of the batch (one workgroup's share far
beyond the mean) must neither lose rows nor report an overflow when cap = the number of kept rows."""
import numpy as np
import pytest

import haploconduct_amd as hc
from haploconduct_amd import synth
from haploconduct_amd.records import REC_FULL, result_cls

pytestmark = pytest.mark.gpu


def _payload_rows(payload):
    p = payload.cpu().numpy()
    count = int(p[0, 0])
    return count, p[1:]


def _expect_rows(host, kept, base):
    want = np.zeros((kept.size, 4), np.int64)
    want[:, 0] = kept + base
    want[:, 1] = host["x1"][kept].view(np.int64)
    want[:, 2] = host["x2"][kept].view(np.int64)
    want[:, 3] = host["mm"][kept].astype(np.int64) | (host["n_cls"][kept].astype(np.int64) << 32)
    return want


def _skewed(n_total, seed, singles=False):
    """Candidates whose kept rows are all at the front: real overlaps first, then pairs of unrelated reads (dropped)."""
    if singles:
        reads, meta = synth.make_single_dataset(6000, 30000, len_lo=150, len_hi=3000, flip_frac=0.3, seed=seed, log_uniform=True)
        good = synth.single_candidates(meta, min_overlap=60, n_candidates=n_total // 8)
    else:
        reads, meta = synth.make_paired_dataset(4000, 6000, flip_frac=0.25, seed=seed)
        good = synth.paired_candidates(meta, n_candidates=n_total // 8, seed=seed + 1)
    rng = np.random.default_rng(seed + 2)
    junk = good[rng.integers(0, good.size, n_total - good.size)].copy()
    junk["read2"] = rng.integers(0, reads.n_reads, junk.size)  # a partner from anywhere: mismatches everywhere, dropped
    junk = junk[junk["read1"] != junk["read2"]]
    return reads, np.concatenate([good, junk])


@pytest.mark.parametrize("singles", [False, True], ids=["pairs_plain_launch", "mixed_lengths_bucketed_launch"])
def test_kept_rows_in_one_stretch_fill_a_tight_payload(singles):
    import torch

    reads, cand = _skewed(600000 if not singles else 300000, 31, singles)
    st = hc.Settings(edge_threshold=0.97, min_overlap_len=60 if singles else 150)
    with hc.EdgeScorer(st) as sc:
        sc.set_reads(reads)
        if singles:
            assert "length-bucketed" in sc.kernel_info()
        host = sc.score_batch(cand)
        kept = np.nonzero(result_cls(host) != 0)[0]
        assert kept.size > 1000 and kept.size < cand.size // 4
        front = (kept < cand.size // 8).mean()
        assert front > 0.9, "the workload is meant to keep its rows at the front"
        d_in = torch.from_numpy(cand.view(np.uint8).reshape(-1)).cuda()
        d_out = torch.empty(cand.size * 24, dtype=torch.uint8, device="cuda")
        for cap, base in ((kept.size, 0), (kept.size + 5, 1 << 33), (3 * kept.size, 17)):
            payload = torch.full((cap + 1, 4), -1, dtype=torch.int64, device="cuda")
            for _ in range(3):  # consecutive launches take the two spill counters in turn
                sc.score_pack_device(d_in.data_ptr(), cand.size, d_out.data_ptr(), cap, base, payload.data_ptr(), None, REC_FULL)
                sc.synchronize()
                count, rows = _payload_rows(payload)
                assert count == kept.size, f"cap {cap}: the payload counts {count} rows, {kept.size} were kept"
                rows = rows[:count]
                rows = rows[np.argsort(rows[:, 0])]
                assert np.array_equal(rows, _expect_rows(host, kept, base))
            assert d_out.cpu().numpy().view(hc.RESULT_DTYPE).tobytes() == host.tobytes()
        # a payload that is too small: the count still says how many there were, nothing is written behind the payload
        cap = kept.size // 3
        payload = torch.full((cap + 2, 4), -1, dtype=torch.int64, device="cuda")
        sc.score_pack_device(d_in.data_ptr(), cand.size, d_out.data_ptr(), cap, 0, payload.data_ptr(), None, REC_FULL)
        sc.synchronize()
        count, rows = _payload_rows(payload)
        assert count == kept.size and (rows[cap] == -1).all()
        got = rows[:cap]
        assert np.unique(got[:, 0]).size == cap and np.isin(got[:, 0], kept).all()
        # and the launch after an overflow starts from clean counters
        payload = torch.full((kept.size + 1, 4), -1, dtype=torch.int64, device="cuda")
        sc.score_pack_device(d_in.data_ptr(), cand.size, d_out.data_ptr(), kept.size, 0, payload.data_ptr(), None, REC_FULL)
        sc.synchronize()
        count, rows = _payload_rows(payload)
        assert count == kept.size and np.array_equal(np.sort(rows[:count, 0]), kept)


def test_launches_on_two_streams_of_one_context():
    """Collecting launches given different streams share the context's segments: the library orders them on the device."""
    import torch

    reads, meta = synth.make_paired_dataset(10000, 8000, flip_frac=0.25, seed=77)
    a = synth.paired_candidates(meta, n_candidates=400000, seed=78)
    b = synth.paired_candidates(meta, n_candidates=250000, seed=79)
    st = hc.Settings(edge_threshold=0.97)
    with hc.EdgeScorer(st) as sc:
        sc.set_reads(reads)
        ha, hb = sc.score_batch(a), sc.score_batch(b)
        ka, kb = np.nonzero(result_cls(ha) != 0)[0], np.nonzero(result_cls(hb) != 0)[0]
        da = torch.from_numpy(a.view(np.uint8).reshape(-1)).cuda()
        db = torch.from_numpy(b.view(np.uint8).reshape(-1)).cuda()
        oa = torch.empty(a.size * 24, dtype=torch.uint8, device="cuda")
        ob = torch.empty(b.size * 24, dtype=torch.uint8, device="cuda")
        pa = torch.zeros((ka.size + 1, 4), dtype=torch.int64, device="cuda")
        pb = torch.zeros((kb.size + 1, 4), dtype=torch.int64, device="cuda")
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        torch.cuda.synchronize()
        for _ in range(6):
            sc.score_pack_device(da.data_ptr(), a.size, oa.data_ptr(), ka.size, 0, pa.data_ptr(), s1.cuda_stream, REC_FULL)
            sc.score_pack_device(db.data_ptr(), b.size, ob.data_ptr(), kb.size, 5, pb.data_ptr(), s2.cuda_stream, REC_FULL)
        s1.synchronize()
        s2.synchronize()
        ca, ra = _payload_rows(pa)
        cb, rb = _payload_rows(pb)
        assert ca == ka.size and cb == kb.size
        ra, rb = ra[np.argsort(ra[:, 0])], rb[np.argsort(rb[:, 0])]
        assert np.array_equal(ra, _expect_rows(ha, ka, 0)) and np.array_equal(rb, _expect_rows(hb, kb, 5))
