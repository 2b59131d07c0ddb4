"""The HIP path against the reference's own compute_overlap, pinned WITHOUT substitutes (tests/golden/compute_overlap.json, see
tests/test_compute_golden.py): every candidate of every setting — score and mismatch rate as the host finalises the device's record, bit for
bit; and, for the candidates the serial insert does not re-orient (pos1 > 0, src/EdgeCalculator.cpp:443-448), the whole Edge as the device's
edge builder (hc_graph_resolve: pos3 / pos4, lengths, vertices, :219-232,254-270,292-308,353-379) makes it."""
import numpy as np
import pytest

import haploconduct_amd as hc
from haploconduct_amd.records import ADMIT_DTYPE, result_n
from tests.test_compute_golden import load, settings_of, wanted

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["default", "low_threshold", "min_read_len", "mismatch_setting", "add_duplicates"])
def test_device_reproduces_the_references_compute_overlap(name):
    c, reads, cand = load()
    want = wanted(c, name)
    st = settings_of(c["settings"][name])
    with hc.EdgeScorer(st) as sc:
        sc.set_reads(reads)
        res = sc.score_batch(cand)
        score, mrate, cls = sc.finalize(res)
        assert np.array_equal(score.view(np.uint64), want["score"].view(np.uint64)), "score not bit-identical"
        assert np.array_equal(mrate.view(np.uint64), want["mismatch_rate"].view(np.uint64)), "mismatch rate not bit-identical"
        if c["settings"][name]["add_duplicates"]:
            return  # (vertices by orientation: the stage's route, tests/golden/ec/*_add_duplicates.json)
        # the Edge itself: rounds of candidates that share no graph slot (unordered vertex pair + equal / opposite orientations)
        todo = [i for i in range(cand.size) if cand["pos1"][i] > 0]
        assert len(todo) > 500
        n_checked = 0
        while todo:
            seen, now, later = set(), [], []
            for i in todo:
                key = (min(int(cand["read1"][i]), int(cand["read2"][i])), max(int(cand["read1"][i]), int(cand["read2"][i])), int(cand["ori1"][i] == cand["ori2"][i]))
                (later if key in seen else now).append(i)
                seen.add(key)
            adm = np.zeros(len(now), ADMIT_DTYPE)
            for k in ("read1", "read2", "pos1", "pos2", "len1", "len2", "perc", "ori1", "ori2", "ord"):
                adm[k] = cand[k][now]
            adm["score"], adm["mm"], adm["n"] = score[now], res["mm"][now], result_n(res)[now]
            g = sc.graph_resolve(adm, reads.n_reads)
            assert g["counts"]["n_edges"] == len(now)
            by_slot = {(min(int(e["v1"]), int(e["v2"])), max(int(e["v1"]), int(e["v2"])), int(e["ori1"] == e["ori2"])): e for e in g["edges"]}
            for i in now:
                e = by_slot[(min(int(cand["read1"][i]), int(cand["read2"][i])), max(int(cand["read1"][i]), int(cand["read2"][i])), int(cand["ori1"][i] == cand["ori2"][i]))]
                for k in ("pos1", "pos2", "pos3", "pos4", "ori1", "ori2", "ord", "v1", "v2", "perc", "len0", "len1", "len2"):
                    assert int(e[k]) == int(want[k][i]), f"candidate {i} ({c['lines'][i]!r}): {k} = {int(e[k])}, the reference's compute_overlap says {int(want[k][i])}"
                assert np.float64(e["score"]).view(np.uint64) == want["score"][i:i + 1].view(np.uint64)[0]
                assert np.float64(e["mismatch_rate"]).view(np.uint64) == want["mismatch_rate"][i:i + 1].view(np.uint64)[0]
                n_checked += 1
            todo = later
        assert n_checked > 500
