"""Flat read set: the layout hc_set_reads takes (include/hcedge.h).

Mirrors what FastqStorage holds after reading the FASTQ files (reference
src/FastqStorage.h:58-98): m_read_vec order = all single-end reads, then all pairs;
each read has an id (the `@id` of the FASTQ record) and one or two sequences."""
import numpy as np


class ReadSet:
    def __init__(self, bases, quals, seq_off, read_first_seq, read_ids):
        self.bases = np.ascontiguousarray(bases, dtype=np.uint8)
        self.quals = np.ascontiguousarray(quals, dtype=np.uint8)
        self.seq_off = np.ascontiguousarray(seq_off, dtype=np.uint64)
        self.read_first_seq = np.ascontiguousarray(read_first_seq, dtype=np.uint32)
        self.read_ids = np.ascontiguousarray(read_ids, dtype=np.uint64)
        assert self.bases.shape == self.quals.shape
        assert self.seq_off[-1] == self.bases.size

    @property
    def n_reads(self):
        return int(self.read_first_seq.size - 1)

    @property
    def n_seq(self):
        return int(self.seq_off.size - 1)

    @classmethod
    def from_lists(cls, singles=(), pairs=(), single_ids=None, pair_ids=None):
        """singles: [(seq, qual)], pairs: [((seq1, qual1), (seq2, qual2))] with str/bytes members."""
        def b(x):
            return x.encode() if isinstance(x, str) else bytes(x)

        seqs, quals, first = [], [], [0]
        for s, q in singles:
            seqs.append(b(s)); quals.append(b(q)); first.append(first[-1] + 1)
        for (s1, q1), (s2, q2) in pairs:
            seqs += [b(s1), b(s2)]; quals += [b(q1), b(q2)]; first.append(first[-1] + 2)
        for s, q in zip(seqs, quals):
            assert len(s) == len(q), "sequence / quality length mismatch"
        off = np.zeros(len(seqs) + 1, dtype=np.uint64)
        if seqs:
            off[1:] = np.cumsum([len(s) for s in seqs])
        bases = np.frombuffer(b"".join(seqs), dtype=np.uint8) if seqs else np.zeros(0, np.uint8)
        qual = np.frombuffer(b"".join(quals), dtype=np.uint8) if seqs else np.zeros(0, np.uint8)
        n = len(singles) + len(pairs)
        ids = list(single_ids if single_ids is not None else range(len(singles)))
        ids += list(pair_ids if pair_ids is not None else range(len(singles), n))
        return cls(bases, qual, off, np.array(first, dtype=np.uint32), np.array(ids, dtype=np.uint64))

    def seq(self, q):
        a, b = int(self.seq_off[q]), int(self.seq_off[q + 1])
        return self.bases[a:b].tobytes(), self.quals[a:b].tobytes()

    def is_paired(self, r):
        return int(self.read_first_seq[r + 1] - self.read_first_seq[r]) == 2

    def write_fastq(self, singles_path=None, paired1_path=None, paired2_path=None):
        """Write the set in the file layout the reference reads (src/FastqStorage.cpp:92-235)."""
        fs = open(singles_path, "wb") if singles_path else None
        f1 = open(paired1_path, "wb") if paired1_path else None
        f2 = open(paired2_path, "wb") if paired2_path else None
        for r in range(self.n_reads):
            q0 = int(self.read_first_seq[r])
            rid = str(int(self.read_ids[r])).encode()
            if self.is_paired(r):
                for f, q in ((f1, q0), (f2, q0 + 1)):
                    s, ql = self.seq(q)
                    f.write(b"@" + rid + b"\n" + s + b"\n+\n" + ql + b"\n")
            else:
                s, ql = self.seq(q0)
                fs.write(b"@" + rid + b"\n" + s + b"\n+\n" + ql + b"\n")
        for f in (fs, f1, f2):
            if f:
                f.close()
