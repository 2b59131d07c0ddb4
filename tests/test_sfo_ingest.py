"""SFO ingest: the native hc_sfo2overlaps and the Python oracle restatement against
outputs of the reference's own scripts/sfo2overlaps.py (tests/golden/sfo/, see make_golden_sfo.py),
plus native-vs-oracle on fresh seeded inputs.  No GPU needed."""
import glob
import os
import sys

import pytest

import haploconduct_amd as hc
from haploconduct_amd import host

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
sys.path.insert(0, os.path.join(HERE, "golden"))
from sfo2overlaps_oracle import sfo2overlaps as oracle_sfo2overlaps  # noqa: E402

CASES = sorted(glob.glob(os.path.join(HERE, "golden", "sfo", "*.sfo")))


@pytest.fixture(autouse=True, params=["auto", "general", "matcher 1", "matcher 7", "matcher 1000"])
def ingest_route(request, monkeypatch):
    """Files of the plain shape (eight fields, single tabs, canonical numbers) are parsed into records and run through
    the partitioned records path; every other file — and every file under HC_SFO_TEXT_GENERAL — through the line-keeping
    general path.  "matcher k": the records path only sorts and hands the sorted run, k records at a time, to the chunk-fed
    matcher hc_found_to_overlaps uses behind the device's sort (HC_SFO_VIA_MATCHER).  Every test below runs all ways."""
    if request.param == "general":
        monkeypatch.setenv("HC_SFO_TEXT_GENERAL", "1")
    if request.param.startswith("matcher"):
        monkeypatch.setenv("HC_SFO_VIA_MATCHER", request.param.split()[1])
    monkeypatch.setenv("HC_SFO_BUCKETS", "5")  # several buckets even on the small fixtures
    return request.param


@pytest.mark.parametrize("sfo", CASES, ids=[os.path.basename(c) for c in CASES])
def test_against_reference_script_outputs(sfo, tmp_path):
    s, p = map(int, open(sfo[:-4] + ".args").read().split())
    want = open(sfo[:-4] + ".expected", "rb").read()
    assert len(want) > 1000
    got_oracle = "".join(l + "\n" for l in oracle_sfo2overlaps(open(sfo).read().splitlines(), s, p)).encode()
    assert got_oracle == want, "oracle restatement differs from the reference script"
    out = str(tmp_path / "overlaps.txt")
    n = host.sfo2overlaps(sfo, out, s, p)
    assert open(out, "rb").read() == want, "native ingest differs from the reference script"
    assert n == want.count(b"\n")


def test_native_equals_oracle_on_fresh_inputs(tmp_path):
    from make_golden_sfo import synth_sfo

    for seed, s, p, n, sep in ((11, 50, 0, 2000, "\t"), (12, 0, 60, 3000, " "), (13, 40, 40, 4000, "\t"), (14, 5, 200, 6000, "\t")):
        lines = synth_sfo(seed, s, p, n, sep)
        sfo = str(tmp_path / f"in{seed}.sfo")
        open(sfo, "w").write("\n".join(lines))  # no trailing newline
        want = "".join(l + "\n" for l in oracle_sfo2overlaps(lines, s, p))
        out = str(tmp_path / f"out{seed}.txt")
        host.sfo2overlaps(sfo, out, s, p)
        assert open(out).read() == want
        assert len(want) > 5000


def _lines_the_matching_can_see(lines, s, p):
    """What hc_found_to_overlaps lets leave the device since round 3 (hc_sfo_kernels.hip: sfo_classify_kernel, sfo_groups_kernel), restated
    on text lines: after the script's flip, sort and uniq, the lines between unpaired reads, the lines of groups (one pair of reads) of two
    and more, and the line that closes such a group."""
    from sfo2overlaps_oracle import flip, get_original_id, is_paired

    tmp = []
    for raw in lines:
        f = raw.split()
        na, nb = get_original_id(int(f[0]), s, p), get_original_id(int(f[1]), s, p)
        t = [str(nb), str(na)] + flip(f) if na > nb else [str(na), str(nb)] + f
        tmp.append(("\t".join(t), raw))
    tmp.sort(key=lambda x: (int(x[0].split()[0]), int(x[0].split()[1]), int(x[0].split()[2]), int(x[0].split()[3]), (x[0] + "\n").encode()))
    uniq = [t for i, t in enumerate(tmp) if i == 0 or t[0] != tmp[i - 1][0]]
    seen = [t for t in uniq if t[0].split()[0] != t[0].split()[1]]
    single = lambda t: not is_paired(int(t[0].split()[0]), s, p) and not is_paired(int(t[0].split()[1]), s, p)
    grouped = [t for t in seen if not single(t)]
    pair = lambda t: t[0].split()[:2]
    keep = [t[1] for t in seen if single(t)]
    for j, t in enumerate(grouped):
        same_prev = j > 0 and pair(grouped[j - 1]) == pair(t)
        same_next = j + 1 < len(grouped) and pair(grouped[j + 1]) == pair(t)
        closes = j > 1 and not same_prev and pair(grouped[j - 2]) == pair(grouped[j - 1])
        if same_prev or same_next or closes:
            keep.append(t[1])
    return keep, len(uniq)


def test_the_lines_the_device_keeps_give_the_same_file(ingest_route):
    """The argument behind the device's choice of records, on adversarial line sets (few reads, so that groups of one, two and more
    lines, repeated lines, self-overlaps and lines between unpaired reads all neighbour each other): the oracle on the kept lines
    writes what it writes on all of them — including the closing line's read types (:94) and the last group that is never matched."""
    if ingest_route != "auto":
        return  # the oracle alone: one route is enough
    import random

    def adversarial(seed, s, p, n):  # ids from a handful of reads, numbers that keep the script's asserts quiet
        rng = random.Random(seed)
        nseq, out = s + 2 * p, []
        while len(out) < n:
            a, b = rng.randrange(nseq), rng.randrange(nseq)
            if a == b and rng.random() < 0.9:
                continue
            out.append("\t".join(map(str, (a, b, rng.choice("NNNI"), rng.randrange(-40, 41), rng.randrange(-40, 41), rng.randrange(50, 151),
                                           rng.randrange(50, 151), rng.randrange(0, 3)))))
            if rng.random() < 0.1:
                out.append(out[-1])  # a repeated line
        return out

    dropped = written = 0
    for seed, s, p, n in ((21, 6, 5, 400), (22, 0, 8, 600), (23, 3, 12, 900), (24, 10, 3, 500), (25, 2, 30, 1500), (26, 0, 3, 200), (27, 1, 1, 60),
                          (28, 0, 40, 300), (29, 5, 60, 500)):
        lines = adversarial(seed, s, p, n)
        kept, n_uniq = _lines_the_matching_can_see(lines, s, p)
        want = oracle_sfo2overlaps(lines, s, p)
        assert oracle_sfo2overlaps(kept, s, p) == want, seed
        dropped += n_uniq - len(kept)
        written += len(want)
    assert written > 500, "the line sets are meant to produce matches"
    assert dropped > 50, "the line sets are meant to hold groups of one line"


def test_the_output_feeds_the_overlaps_parser(tmp_path):
    # every line the ingest writes is a well-formed 13-column record for the stage's parser
    sfo = CASES[0]
    s, p = map(int, open(sfo[:-4] + ".args").read().split())
    out = str(tmp_path / "overlaps.txt")
    host.sfo2overlaps(sfo, out, s, p)
    for line in open(out).read().splitlines():
        rc, o = host.parse_overlap(line)
        assert rc == 0 and o["type1"] in "sp" and o["type2"] in "sp"


def test_errors(tmp_path):
    bad = str(tmp_path / "bad.sfo")
    open(bad, "w").write("1 2 N 3 4 5\n")
    with pytest.raises(hc.HcError):
        host.sfo2overlaps(bad, str(tmp_path / "o.txt"), 10, 0)
    open(bad, "w").write("1 2 N x 4 5 6 0\n")
    with pytest.raises(hc.HcError):
        host.sfo2overlaps(bad, str(tmp_path / "o.txt"), 10, 0)
    open(bad, "w").write("1 99 N 3 4 5 6 0\n")
    with pytest.raises(hc.HcError):
        host.sfo2overlaps(bad, str(tmp_path / "o.txt"), 5, 5)  # id out of range
    with pytest.raises(hc.HcError):
        host.sfo2overlaps(str(tmp_path / "missing.sfo"), str(tmp_path / "o.txt"), 1, 0)
