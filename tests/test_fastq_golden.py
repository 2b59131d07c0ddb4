"""The host FASTQ reader (FastqStorage mirror: line reader, paired-end records, --IDs file, id index) against vectors from
the reference's own FastqStorage constructor / read_pairs / fastq_to_stream / read_new_ids (fragment probe
oracle/_ref/hcref_fastq; generator tests/golden/make_golden_fastq.py).  CPU only."""
import json
import os

import numpy as np
import pytest

from haploconduct_amd import host

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "fastq_pairs.json")))["cases"]


@pytest.mark.parametrize("parallel", [False, True], ids=["sequential", "threads"])
@pytest.mark.parametrize("case", GOLD, ids=[c["name"] for c in GOLD])
def test_paired_fastq_reader_matches_reference_vectors(case, parallel, tmp_path, monkeypatch):
    if parallel:  # the mapped, multi-threaded reader on the same tiny inputs: three threads, one record each at least
        monkeypatch.setenv("HC_FASTQ_THREADS", "3")
        monkeypatch.setenv("HC_FASTQ_PARALLEL_MIN", "0")
        monkeypatch.setenv("HC_FASTQ_GRAIN", "1")
    else:
        monkeypatch.setenv("HC_FASTQ_THREADS", "1")
    p1, p2, ids = str(tmp_path / "p1.fastq"), str(tmp_path / "p2.fastq"), str(tmp_path / "ids.txt")
    open(p1, "w", newline="").write(case["p1"])
    if case.get("p2") is not None:
        open(p2, "w", newline="").write(case["p2"])
    if case.get("ids") is not None:
        open(ids, "w", newline="").write(case["ids"])
    want = case["expect"]
    kw = dict(paired1=p1, paired2=p2, ids=ids if case.get("ids") is not None else None, max_reads=case.get("max_reads", 100000))
    if want["exit"] != 0:
        # exit(1) with a message, or an uncaught std::out_of_range: the library reports an error instead of ending the process
        with pytest.raises(Exception) as e:
            host.Fastq(**kw)
        msg = want["stderr"].strip().split("\n")[0]
        if want["exit"] == 1 and "<dir>" not in msg:
            assert msg.rstrip(".") .split(" ... ")[0][:40] in str(e.value), (msg, str(e.value))
        return
    f = host.Fastq(**kw)
    assert (f.n_single, f.n_paired) == (want["singles"], want["pairs"])
    assert f.read_ids.tolist() == [r["id"] for r in want["reads"]]
    for i, r in enumerate(want["reads"]):
        q = int(f.read_first_seq[i])
        assert int(f.read_first_seq[i + 1]) - q == 2
        for m, (sk, pk) in enumerate((("seq1", "phred1"), ("seq2", "phred2"))):
            a, b = int(f.seq_off[q + m]), int(f.seq_off[q + m + 1])
            assert f.bases[a:b].tobytes().decode("latin1") == r[sk], (i, sk)
            assert f.quals[a:b].tobytes().decode("latin1") == r[pk], (i, pk)
    # m_ID_to_index: the first read with an id owns it
    first = {}
    for i, rid in enumerate(f.read_ids.tolist()):
        first.setdefault(str(rid), i)
    assert first == want["id_to_index"]
    f.close()
