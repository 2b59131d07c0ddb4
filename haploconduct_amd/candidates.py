"""A small exact-seed suffix–prefix candidate finder for REAL reads (test/bench support).

rust-overlaps / bwa / blast, which produce the overlaps file in the reference workflows
(savage.py:664,713), are not available offline, so parity tests on the reference's example reads
need their own candidates: read B (in either orientation) is a candidate partner of read A at
offset p when the first k bases of B occur in A at p.  Whether the pair really overlaps is exactly
what the edge-calculation stage then decides.  Output: hc_overlap_rec arrays in sfo2overlaps order
and the matching 13-column lines."""
import numpy as np

from .records import OVERLAP_DTYPE

_COMP = bytes.maketrans(b"ACGTN", b"TGCAN")


def _rc(s):
    return s.translate(_COMP)[::-1]


def _index(seqs, k):
    idx = {}
    for r, s in enumerate(seqs):
        for p in range(0, len(s) - k + 1):
            km = s[p:p + k]
            if b"N" in km:
                continue
            idx.setdefault(km, []).append((r, p))
    return idx


def _sfo_sort(rec):
    lo = np.minimum(rec["read1"], rec["read2"]).astype(np.uint64)
    hi = np.maximum(rec["read1"], rec["read2"]).astype(np.uint64)
    return rec[np.argsort((lo << np.uint64(32)) | hi, kind="stable")]


def single_candidates(reads, read_indices, k=20, min_overlap=50, max_hits=64):
    """s-s candidates among the single-end reads `read_indices` (indices into m_read_vec order)."""
    seqs = [reads.seq(int(reads.read_first_seq[r]))[0].upper() for r in read_indices]
    idx = _index(seqs, k)
    rows = []
    for jb, sb in enumerate(seqs):
        for ob, oriented in ((1, sb), (0, _rc(sb))):
            seed = oriented[:k]
            if len(seed) < k or b"N" in seed:
                continue
            hits = idx.get(seed, ())
            if len(hits) > max_hits:
                continue
            for ja, p in hits:
                if ja == jb:
                    continue
                la, lb = len(seqs[ja]), len(sb)
                ovl = min(la - p, lb)
                if ovl < min_overlap:
                    continue
                perc = min(int(100 * ovl // min(la, lb)), 100)
                rows.append((read_indices[ja], read_indices[jb], p, 0, 1, ob, ord("-"), 0, ovl, 0, perc))
    return _sfo_sort(np.array(rows, dtype=OVERLAP_DTYPE)) if rows else np.zeros(0, OVERLAP_DTYPE)


def paired_candidates(reads, read_indices, k=20, min_overlap=50, max_hits=64):
    """p-p candidates (both reads '+') among the paired reads `read_indices`: /1 of B seeds into /1 of A,
    and the /2 mates overlap in either order (ord 1 / 2)."""
    s1 = [reads.seq(int(reads.read_first_seq[r]))[0].upper() for r in read_indices]
    s2 = [reads.seq(int(reads.read_first_seq[r]) + 1)[0].upper() for r in read_indices]
    idx1 = _index(s1, k)
    rows = []
    for jb in range(len(s1)):
        seed = s1[jb][:k]
        if len(seed) < k or b"N" in seed:
            continue
        hits = idx1.get(seed, ())
        if len(hits) > max_hits:
            continue
        for ja, p1 in hits:
            if ja == jb:
                continue
            l1 = min(len(s1[ja]) - p1, len(s1[jb]))
            if l1 < min_overlap // 2:
                continue
            for order, (first, second) in ((ord("1"), (s2[ja], s2[jb])), (ord("2"), (s2[jb], s2[ja]))):
                sd = second[:k]
                if len(sd) < k or b"N" in sd:
                    continue
                p2 = first.find(sd)
                if p2 < 0:
                    continue
                l2 = min(len(first) - p2, len(second))
                if l2 < min_overlap // 2:
                    continue
                perc1 = min(int(100 * l1 // min(len(s1[ja]), len(s1[jb]))), 100)
                perc2 = min(int(100 * l2 // min(len(first), len(second))), 100)
                perc = int(0.5 * (perc1 + perc2)) if perc2 > 0 else perc1
                rows.append((read_indices[ja], read_indices[jb], p1, p2, 1, 1, order, 3, l1, l2, perc))
                break
    return _sfo_sort(np.array(rows, dtype=OVERLAP_DTYPE)) if rows else np.zeros(0, OVERLAP_DTYPE)
