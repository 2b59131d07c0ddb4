"""Deterministic synthetic workloads for the edge-calculation path.

genome = uniform random ACGT; K strains differing from it at `divergence` of the
positions; fragments at uniform random start, insert size uniform [350, 600]; reads
2 x read_len forward-forward (SAVAGE convention: both mates on the fragment's strand);
per-base substitution error 0.5 %, 0.1 % N; qualities i.i.d. from
{2,12,20,30,37,37,37,40,40} (+33 ASCII).  Candidates: all ordered pairs whose /1
offsets give len1 >= min_len and whose /2 offsets give len2 >= min_len, `ord` by the
sign of the /2 offset, sub-sampled with a seeded RNG to the requested count.

`flip_frac` of the read pairs are STORED reverse-complemented (/1' = rc(/2),
/2' = rc(/1)); candidates touching them carry ori '-' so the true overlap is recovered
(exercises get_rev_comp / get_rev_phred, reference src/Read.h:172-201).
"""
import numpy as np

from .readstore import ReadSet
from .records import OVERLAP_DTYPE

QUAL_SET = np.array([2, 12, 20, 30, 37, 37, 37, 40, 40], dtype=np.uint8) + 33
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTN", b"TGCAN"):
    _COMP[_a] = _b


def _mutate(rng, seq, rate):
    """Substitute `rate` of the positions by a different base."""
    n = seq.size
    k = rng.random(n) < rate
    idx = np.nonzero(k)[0]
    if idx.size:
        code = np.searchsorted(_ACGT, seq[idx])  # A,C,G,T are sorted in ASCII
        code = (code + rng.integers(1, 4, idx.size)) % 4
        seq = seq.copy()
        seq[idx] = _ACGT[code]
    return seq


def make_paired_dataset(n_pairs, genome_len, n_strains=3, divergence=0.01, read_len=150, ins_lo=350, ins_hi=600,
                        err=0.005, n_rate=0.001, flip_frac=0.0, seed=1, trim_lo=None, quals=None, qual_p=None):
    """Returns (ReadSet, meta) where meta has frag start `s`, /2 start `e`, `flipped` per pair.
    quals: the quality bytes (ASCII, +33 applied) the bases draw from, i.i.d. — uniformly, or with the probabilities `qual_p` (a histogram
    of real reads); default: QUAL_SET.  The bases, the fragments and hence the candidates do not depend on it (qualities are drawn last).
    trim_lo: every mate keeps only its first U[trim_lo, read_len] bases (quality-trimmed reads: sequences of mixed length; needs
    flip_frac == 0); meta then also has the mates' lengths `l1`, `l2`."""
    rng = np.random.default_rng(seed)
    base = _ACGT[rng.integers(0, 4, genome_len)]
    strains = [_mutate(rng, base, divergence) for _ in range(n_strains)]
    strain = rng.integers(0, n_strains, n_pairs)
    ins = rng.integers(ins_lo, ins_hi + 1, n_pairs)
    s = (rng.random(n_pairs) * (genome_len - ins)).astype(np.int64)
    e = s + ins - read_len
    G = np.stack(strains)  # [K, genome_len]
    ar = np.arange(read_len)
    r1 = G[strain[:, None], s[:, None] + ar[None, :]]
    r2 = G[strain[:, None], e[:, None] + ar[None, :]]

    def noise(r):
        r = r.copy()
        k = rng.random(r.shape) < err
        idx = np.nonzero(k)
        code = np.searchsorted(_ACGT, r[idx])
        r[idx] = _ACGT[(code + rng.integers(1, 4, code.size)) % 4]
        r[rng.random(r.shape) < n_rate] = ord("N")
        return r

    r1, r2 = noise(r1), noise(r2)
    qset = QUAL_SET if quals is None else np.asarray(quals, dtype=np.uint8)
    if qual_p is None:
        q1 = qset[rng.integers(0, qset.size, r1.shape)]
        q2 = qset[rng.integers(0, qset.size, r2.shape)]
    else:  # inverse-CDF draw (rng.choice with p= needs 8 bytes per draw)
        cdf = np.cumsum(np.asarray(qual_p, dtype=np.float64))
        cdf /= cdf[-1]
        q1 = qset[np.minimum(np.searchsorted(cdf, rng.random(r1.shape, dtype=np.float32)), qset.size - 1)]
        q2 = qset[np.minimum(np.searchsorted(cdf, rng.random(r2.shape, dtype=np.float32)), qset.size - 1)]
    flipped = rng.random(n_pairs) < flip_frac
    if flipped.any():
        f = np.nonzero(flipped)[0]
        a1, a2, b1, b2 = r1[f].copy(), r2[f].copy(), q1[f].copy(), q2[f].copy()
        r1[f] = _COMP[a2[:, ::-1]]
        r2[f] = _COMP[a1[:, ::-1]]
        q1[f] = b2[:, ::-1]
        q2[f] = b1[:, ::-1]
    # interleave /1,/2 per pair: seq 2i = /1, 2i+1 = /2
    bases = np.empty((n_pairs, 2, read_len), np.uint8)
    quals = np.empty((n_pairs, 2, read_len), np.uint8)
    bases[:, 0], bases[:, 1] = r1, r2
    quals[:, 0], quals[:, 1] = q1, q2
    first = np.arange(n_pairs + 1, dtype=np.uint32) * 2
    if trim_lo is not None:
        assert flip_frac == 0.0, "trimmed pairs are generated unflipped"
        lens = rng.integers(trim_lo, read_len + 1, (n_pairs, 2))
        keep = (np.arange(read_len)[None, None, :] < lens[:, :, None])
        seq_off = np.zeros(2 * n_pairs + 1, dtype=np.uint64)
        seq_off[1:] = np.cumsum(lens.reshape(-1))
        reads = ReadSet(bases[keep], quals[keep], seq_off, first, np.arange(n_pairs, dtype=np.uint64))
        return reads, {"s": s, "e": e, "flipped": flipped, "read_len": read_len, "strain": strain, "l1": lens[:, 0].astype(np.int32),
                       "l2": lens[:, 1].astype(np.int32)}
    seq_off = np.arange(2 * n_pairs + 1, dtype=np.uint64) * read_len
    reads = ReadSet(bases.reshape(-1), quals.reshape(-1), seq_off, first, np.arange(n_pairs, dtype=np.uint64))
    return reads, {"s": s, "e": e, "flipped": flipped, "read_len": read_len, "strain": strain}


def _sfo_order(read1, read2):
    """Order real overlap files have: scripts/sfo2overlaps.py:53 sorts the lines by (smaller read id,
    larger read id) before writing them, and FNO writes a sorted std::set of lines."""
    lo = np.minimum(read1, read2).astype(np.uint64)
    hi = np.maximum(read1, read2).astype(np.uint64)
    return np.argsort((lo << np.uint64(32)) | hi, kind="stable")


def paired_candidates(meta, n_candidates=None, min_len=75, seed=2, max_window=100000):
    """All p-p candidates (len1 >= min_len and len2 >= min_len), optionally sub-sampled to n_candidates,
    in sfo2overlaps order.  int32 intermediates: 1e8 candidates need ~8 GB of host memory."""
    s, e, flipped, rl = meta["s"], meta["e"], meta["flipped"], meta["read_len"]
    order = np.argsort(s, kind="stable").astype(np.int32)
    ss, ee = s[order].astype(np.int32), e[order].astype(np.int32)
    n = ss.size
    maxd = rl - min_len
    out_i, out_j, out_p1, out_d2 = [], [], [], []
    for k in range(1, min(n, max_window)):
        ds = ss[k:] - ss[:-k]
        m = ds <= maxd
        if not m.any():
            break
        d2 = ee[k:] - ee[:-k]
        m &= np.abs(d2) <= maxd
        idx = np.nonzero(m)[0]
        out_i.append(order[idx])
        out_j.append(order[idx + k])
        out_p1.append(ds[idx].astype(np.int16))
        out_d2.append(d2[idx].astype(np.int16))
    cat = lambda xs, dt: np.concatenate(xs) if xs else np.zeros(0, dt)
    i, j, p1, d2 = cat(out_i, np.int32), cat(out_j, np.int32), cat(out_p1, np.int16), cat(out_d2, np.int16)
    del out_i, out_j, out_p1, out_d2
    rng = np.random.default_rng(seed)
    if n_candidates is not None:
        if i.size < n_candidates:
            raise ValueError(f"only {i.size} candidates exist, {n_candidates} requested: lower genome_len")
        if i.size > n_candidates:
            pick = np.sort(rng.choice(i.size, n_candidates, replace=False))
            i, j, p1, d2 = i[pick], j[pick], p1[pick], d2[pick]
            del pick
    o = _sfo_order(i, j)
    i, j, p1, d2 = i[o], j[o], p1[o].astype(np.int32), d2[o].astype(np.int32)
    del o
    rec = np.zeros(i.size, dtype=OVERLAP_DTYPE)
    rec["read1"], rec["read2"] = i, j
    rec["pos1"] = p1
    rec["pos2"] = np.abs(d2)
    rec["ord"] = np.where(d2 >= 0, ord("1"), ord("2"))
    rec["ori1"] = ~flipped[i]
    rec["ori2"] = ~flipped[j]
    rec["len1"] = rl - p1
    rec["len2"] = rl - np.abs(d2)
    rec["perc"] = (0.5 * (np.floor(100.0 * rec["len1"] / rl) + np.floor(100.0 * rec["len2"] / rl))).astype(np.uint32)
    rec["flags"] = 3
    if "l1" in meta:  # trimmed mates: the overlap of a suffix of one mate with the other's prefix, from their own lengths; short ones go
        l1, l2 = meta["l1"], meta["l2"]
        a1 = np.minimum(l1[i] - p1, l1[j])
        ord1 = d2 >= 0
        a2 = np.where(ord1, np.minimum(l2[i] - np.abs(d2), l2[j]), np.minimum(l2[j] - np.abs(d2), l2[i]))
        ok = (a1 >= min_len) & (a2 >= min_len)
        rec["len1"], rec["len2"] = np.maximum(a1, 0), np.maximum(a2, 0)
        rec["perc"] = (0.5 * (np.floor(100.0 * np.maximum(a1, 0) / np.minimum(l1[i], l1[j])) + np.floor(100.0 * np.maximum(a2, 0) / np.minimum(l2[i], l2[j])))).astype(np.uint32)
        rec = rec[ok]
    return rec


def make_single_dataset(n_reads, genome_len, len_lo=150, len_hi=150, n_strains=2, divergence=0.001, err=0.005,
                        n_rate=0.001, flip_frac=0.25, seed=3, quals=QUAL_SET, log_uniform=False):
    """Single-end reads of (log-)uniform length in [len_lo, len_hi] (configs 4/5 style)."""
    rng = np.random.default_rng(seed)
    base = _ACGT[rng.integers(0, 4, genome_len)]
    strains = np.stack([_mutate(rng, base, divergence) for _ in range(n_strains)])
    if log_uniform:
        lens = np.exp(rng.uniform(np.log(len_lo), np.log(len_hi + 1), n_reads)).astype(np.int64)
    else:
        lens = rng.integers(len_lo, len_hi + 1, n_reads)
    lens = np.clip(lens, len_lo, min(len_hi, genome_len))
    s = (rng.random(n_reads) * (genome_len - lens)).astype(np.int64)
    strain = rng.integers(0, n_strains, n_reads)
    flipped = rng.random(n_reads) < flip_frac
    off = np.zeros(n_reads + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens)
    total = int(off[-1])
    bases = np.empty(total, np.uint8)
    for r in range(n_reads):
        seg = strains[strain[r], s[r]:s[r] + lens[r]]
        bases[int(off[r]):int(off[r + 1])] = _COMP[seg[::-1]] if flipped[r] else seg
    k = rng.random(total) < err
    idx = np.nonzero(k)[0]
    code = np.searchsorted(_ACGT, bases[idx])
    bases[idx] = _ACGT[(code + rng.integers(1, 4, idx.size)) % 4]
    bases[rng.random(total) < n_rate] = ord("N")
    q = np.asarray(quals, dtype=np.uint8)[rng.integers(0, len(quals), total)]
    first = np.arange(n_reads + 1, dtype=np.uint32)
    reads = ReadSet(bases, q, off, first, np.arange(n_reads, dtype=np.uint64))
    return reads, {"s": s, "lens": lens, "flipped": flipped, "strain": strain}


def single_candidates(meta, min_overlap=100, n_candidates=None, seed=4, max_window=100000):
    """All s-s suffix-prefix candidates with overlap >= min_overlap, expressed in the STORED orientation."""
    s, lens, flipped = meta["s"], meta["lens"], meta["flipped"]
    order = np.argsort(s, kind="stable")
    ss, ll = s[order], lens[order]
    n = ss.size
    oi, oj, op, ol = [], [], [], []
    for k in range(1, min(n, max_window)):
        ds = ss[k:] - ss[:-k]
        ovl = np.minimum(ll[:-k] - ds, ll[k:])
        m = ovl >= min_overlap
        if not (ds < ll[:-k]).any():
            break
        idx = np.nonzero(m)[0]
        oi.append(order[idx]); oj.append(order[idx + k]); op.append(ds[idx]); ol.append(ovl[idx])
    i = np.concatenate(oi) if oi else np.zeros(0, np.int64)
    j = np.concatenate(oj) if oj else np.zeros(0, np.int64)
    p = np.concatenate(op) if op else np.zeros(0, np.int64)
    ovl = np.concatenate(ol) if ol else np.zeros(0, np.int64)
    rng = np.random.default_rng(seed)
    if n_candidates is not None and i.size > n_candidates:
        pick = np.sort(rng.choice(i.size, n_candidates, replace=False))
        i, j, p, ovl = i[pick], j[pick], p[pick], ovl[pick]
    rec = np.zeros(i.size, dtype=OVERLAP_DTYPE)
    rec["read1"], rec["read2"], rec["pos1"] = i, j, p
    rec["ord"] = ord("-")
    rec["ori1"] = ~flipped[i]
    rec["ori2"] = ~flipped[j]
    rec["len1"] = ovl
    rec["perc"] = np.minimum(np.floor(100.0 * ovl / np.minimum(lens[i], lens[j])), 100).astype(np.uint32)
    return rec[_sfo_order(rec["read1"], rec["read2"])]


def records_to_lines(rec, reads):
    """13-column overlaps-file lines for a record array."""
    ids = reads.read_ids
    lines = []
    for r in rec:
        p1 = reads.is_paired(int(r["read1"]))
        p2 = reads.is_paired(int(r["read2"]))
        ss = not p1 and not p2
        lines.append("\t".join([
            str(int(ids[r["read1"]])), str(int(ids[r["read2"]])), str(int(r["pos1"])),
            "-" if ss else str(int(r["pos2"])), chr(r["ord"]), "+" if r["ori1"] else "-", "+" if r["ori2"] else "-",
            str(int(r["perc"])), "-" if ss else str(int(r["perc"])), str(int(r["len1"])),
            "-" if ss else str(int(r["len2"])), "p" if p1 else "s", "p" if p2 else "s"]))
    return lines
