"""Counts the quality bytes of the reference's example reads (data: a histogram, no source text) — the alphabets and their skew that
SAVAGE / POLYTE users' reads have (polyte/example forward.fastq carries 35 distinct quality bytes, the SAVAGE example's
singles 25).  bench.py's c3q35r / c3q25r workloads draw synthetic qualities i.i.d. from these histograms.
Run in the build container (needs /root/reference):  python tests/golden/make_quality_histograms.py"""
import collections
import json
import os

REF = "/root/reference"
FILES = {
    "polyte_forward": "polyte/example/input/forward.fastq",
    "polyte_reverse": "polyte/example/input/reverse.fastq",
    "savage_singles": "savage/example/input_fas/singles.fastq",
    "savage_paired1": "savage/example/input_fas/paired1.fastq",
}


def histogram(path):
    c = collections.Counter()
    with open(path, "rb") as f:
        for k, line in enumerate(f):
            if k % 4 == 3:
                c.update(line.rstrip(b"\r\n"))
    return c


def main():
    out = {}
    for name, rel in FILES.items():
        p = os.path.join(REF, rel)
        if not os.path.exists(p):
            continue
        c = histogram(p)
        out[name] = {"source": rel, "distinct": len(c), "bases": sum(c.values()),
                     "counts": {str(b): n for b, n in sorted(c.items())}}  # key: the ASCII byte (Phred + 33)
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "quality_histograms.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    for k, v in out.items():
        top = sorted(v["counts"].items(), key=lambda kv: -kv[1])[:5]
        print(k, v["distinct"], "distinct;", "top:", [(chr(int(b)), round(n / v["bases"], 3)) for b, n in top])


if __name__ == "__main__":
    main()
