#!/usr/bin/env python3
"""Golden vectors for the FASTQ reader: small paired FASTQ inputs (+ --IDs files) through the reference's own
FastqStorage constructor / read_pairs / fastq_to_stream / read_new_ids (fragment probe oracle/_ref/hcref_fastq, built by
`make -C oracle ref` from /root/reference/src where it lies).  Runs in the build container only; writes
tests/golden/fastq_pairs.json (inputs and what the reference made of them: data, not source)."""
import json
import os
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
PROBE = os.path.join(ROOT, "oracle", "_ref", "hcref_fastq")


def rec(i, seq, qual=None, extra=""):
    return f"@{i}{extra}\n{seq}\n+\n{qual if qual is not None else 'I' * len(seq)}\n"


def fq(records):
    return "".join(records)


BASE = [("ACGTACGTAC", "IIIIIHHHHH"), ("TTTTGGGGCC", "5555566666"), ("NACGTNACGT", "!!!!!IIIII"), ("GATTACAGAT", "ABCDEFGHIJ"), ("CCCCCCCCCC", "JJJJJJJJJJ")]
P1 = [rec(i, s, q) for i, (s, q) in enumerate(BASE)]
P2 = [rec(i, s[::-1], q[::-1]) for i, (s, q) in enumerate(BASE)]

CASES = [
    dict(name="plain", p1=fq(P1), p2=fq(P2)),
    dict(name="header_tokens", p1=rec(7, "ACGT", extra=" some comment/1") + rec(8, "ACGA", extra="\tx") + "@  9 z\nAAAA\n+\nIIII\n",
         p2=rec(7, "TTTT", extra=" other/2") + rec(8, "TTTA", extra="\ty") + "@  9\nCCCC\n+\nIIII\n"),
    dict(name="no_trailing_newline", p1=fq(P1)[:-1], p2=fq(P2)[:-1]),
    dict(name="second_file_shorter", p1=fq(P1), p2=fq(P2[:3])),
    dict(name="incomplete_last_record", p1=fq(P1[:2]) + "@2\nACGT\n", p2=fq(P2[:2]) + "@2\nACGT\n+\n"),
    dict(name="crlf", p1=fq(P1[:2]).replace("\n", "\r\n"), p2=fq(P2[:2]).replace("\n", "\r\n")),
    dict(name="lower_case_kept", p1=rec(1, "acgtnACGT"), p2=rec(1, "ttttaaaac")),
    dict(name="max_reads", p1=fq(P1), p2=fq(P2), max_reads=2),
    dict(name="id_bases", p1=rec("0x10", "ACGT") + rec("010", "ACGA") + rec("12abc", "ACGC") + rec("abc", "ACGG"),
         p2=rec("0x10", "TTTT") + rec("010", "TTTA") + rec("12abc", "TTTC") + rec("abc", "TTTG")),
    dict(name="duplicate_ids", p1=rec(3, "AAAA") + rec(5, "CCCC") + rec(3, "GGGG"), p2=rec(3, "TTTT") + rec(5, "TTTA") + rec(3, "TTTC")),
    dict(name="blank_plus_line_content", p1="@4\nACGT\n+anything here\nIIII\n", p2="@4\nTTTT\n+\nIIII\n"),
    dict(name="ids_file", p1=rec("read_a", "ACGT") + rec("read_b", "ACGA"), p2=rec("read_a", "TTTT") + rec("read_b", "TTTA"),
         ids="5\tread_a\n6\t>read_b\n"),
    dict(name="ids_file_three_fields", p1=rec("ra", "ACGT") + rec("rb", "ACGA") + rec("rc", "ACGC"),
         p2=rec("ra", "TTTT") + rec("rb", "TTTA") + rec("rc", "TTTC"), ids="5\tra\t1\n6\trb\tleft over\n7\trc\n"),
    dict(name="ids_file_largest", p1=rec("q", "ACGT"), p2=rec("q", "TTTT"), ids="3\tz\n900\tq\n17\tw\n"),
    dict(name="ids_file_missing_id", p1=rec("nobody", "ACGT"), p2=rec("nobody", "TTTT"), ids="5\tread_a\n"),
    dict(name="ids_file_one_field", p1=rec("a", "ACGT"), p2=rec("a", "TTTT"), ids="5\n"),
    dict(name="err_no_at", p1="7\nACGT\n+\nIIII\n", p2="@7\nTTTT\n+\nIIII\n"),
    dict(name="err_order", p1=rec(1, "ACGT") + rec(2, "ACGA"), p2=rec(1, "TTTT") + rec(3, "TTTA")),
    dict(name="err_empty_sequence", p1=rec(1, "ACGT") + "@2\n\n+\n\n", p2=rec(1, "TTTT") + rec(2, "TTTA")),
    dict(name="err_missing_file", p1=fq(P1), p2=None),
    dict(name="empty_files", p1="", p2=""),
]


def run(case, d):
    p1, p2, ids = os.path.join(d, "p1.fastq"), os.path.join(d, "p2.fastq"), os.path.join(d, "ids.txt")
    open(p1, "w", newline="").write(case["p1"])
    if case.get("p2") is not None:
        open(p2, "w", newline="").write(case["p2"])
    elif os.path.exists(p2):
        os.remove(p2)
    if case.get("ids") is not None:
        open(ids, "w", newline="").write(case["ids"])
    r = subprocess.run([PROBE, p1, p2, ids if case.get("ids") is not None else "-", str(case.get("max_reads", 100000))],
                       capture_output=True, text=True)
    out = {"exit": r.returncode if r.returncode >= 0 else "signal", "stderr": r.stderr.replace(d, "<dir>")}
    if r.returncode == 0:
        reads, index = [], {}
        for line in r.stdout.splitlines():
            f = line.split()
            if f[0] == "R":
                unhex = lambda h: "" if h == "-" else bytes.fromhex(h).decode("latin1")
                reads.append({"id": int(f[2]), "seq1": unhex(f[3]), "seq2": unhex(f[4]), "phred1": unhex(f[5]), "phred2": unhex(f[6])})
            elif f[0] == "M":
                index[f[1]] = int(f[2])
            elif f[0] == "C":
                out.update(singles=int(f[1]), pairs=int(f[2]), largest_read_id=int(f[3]))
        out.update(reads=reads, id_to_index=index)
    return out


def main():
    res = []
    with tempfile.TemporaryDirectory() as d:
        for c in CASES:
            c = dict(c)
            c["expect"] = run(c, d)
            res.append(c)
            print(c["name"], c["expect"]["exit"], c["expect"].get("pairs"), c["expect"]["stderr"].strip()[:80])
    json.dump({"source": "reference src/FastqStorage.{h,cpp} via oracle/_ref/hcref_fastq (fragment probe)", "cases": res},
              open(os.path.join(HERE, "fastq_pairs.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
