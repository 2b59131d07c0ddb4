"""overlap_finder_oracle.py — TEST INFRASTRUCTURE, not product code.

Brute-force statement of what hc_find_overlaps (include/hcedge.h) promises: for every pair of sequences with
SFO ids idA < idB, both orientations of B (with reversals) and every diagonal d, the common stretch of A and B
placed at offset d: length L >= min_overlap and Hamming mismatches K <= floor(err_rate * L), a non-ACGT symbol
matching nothing.  O(n^2 * len^2): small inputs only.

PARITY UNPINNED: the tool this replaces in the pipelines, rust-overlaps (savage.py:664), is an external program
that is not in the reference tree, so there is no reference output to compare with; the SFO record semantics
(OHA, OHB, OLA, OLB) follow how the reference's scripts/sfo2overlaps.py:139-200 reads them back.
"""
import numpy as np

_COMP = np.zeros(256, np.uint8)
_COMP[list(b"ACGTN")] = list(b"TGCAN")


def sfo_sequences(reads):
    """The s_p1_p2.fasta order (savage.py:643-664): singles, then every /1 mate, then every /2 mate."""
    singles, m1, m2 = [], [], []
    for r in range(reads.n_reads):
        q = int(reads.read_first_seq[r])
        if reads.is_paired(r):
            m1.append(reads.seq(q)[0])
            m2.append(reads.seq(q + 1)[0])
        else:
            assert not m1, "single-end reads must come first"
            singles.append(reads.seq(q)[0])
    return [np.frombuffer(s, np.uint8) for s in singles + m1 + m2]


def find_overlaps(reads, err_rate, min_overlap, reversals=True, inclusions=True):
    seqs = sfo_sequences(reads)
    acgt = np.zeros(256, bool)
    acgt[list(b"ACGT")] = True
    out = []
    for a in range(len(seqs)):
        A = seqs[a]
        la = A.size
        for b in range(a + 1, len(seqs)):
            for inv in ((0, 1) if reversals else (0,)):
                B = _COMP[seqs[b]][::-1] if inv else seqs[b]
                lb = B.size
                for d in range(-(lb - min_overlap), la - min_overlap + 1):
                    start, end = max(0, d), min(la, d + lb)
                    L = end - start
                    if L < min_overlap:
                        continue
                    inclusion = (d >= 0 and d + lb <= la) or (d <= 0 and d + lb >= la)
                    if inclusion and not inclusions:
                        continue
                    x, y = A[start:end], B[start - d:end - d]
                    K = int(np.count_nonzero((x != y) | ~acgt[x]))
                    if K <= int(err_rate * L):
                        out.append((a, b, d, d + lb - la, L, L, K, inv))
    # the product's order: (idA, idB, orientation, d)
    out.sort(key=lambda r: (r[0], r[1], r[7], r[2]))
    return out
