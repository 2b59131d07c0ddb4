#!/usr/bin/env python3
"""Copies the reference's example READ DATA (not source) for the C1 parity tests: savage/example/input_fas WHOLE
(BASELINE config 1: 2 000 merged singles of ~400-490 bp + 200 2x250 pairs, 25 distinct quality values, N and Q0 bases)
and an excerpt of polyte/example/input (2x250, 35 distinct quality values -> exercises the wide-symbol path), gzipped.
Run in the build container:  python tests/golden/make_example_excerpt.py"""
import gzip
import os

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def head(path, n_records):
    with open(path, "rb") as f:
        lines = f.read().split(b"\n")
    if n_records is None:  # the whole file
        return b"\n".join(lines).rstrip(b"\n") + b"\n"
    return b"\n".join(lines[: 4 * n_records]) + b"\n"


def main():
    out = {
        "savage_singles.fastq": head(f"{REF}/savage/example/input_fas/singles.fastq", None),
        "savage_paired1.fastq": head(f"{REF}/savage/example/input_fas/paired1.fastq", None),
        "savage_paired2.fastq": head(f"{REF}/savage/example/input_fas/paired2.fastq", None),
        "polyte_forward.fastq": head(f"{REF}/polyte/example/input/forward.fastq", 500),
        "polyte_reverse.fastq": head(f"{REF}/polyte/example/input/reverse.fastq", 500),
    }
    for name, data in out.items():
        with gzip.GzipFile(os.path.join(HERE, name + ".gz"), "wb", mtime=0) as f:
            f.write(data)
        print(name, len(data), "bytes")


if __name__ == "__main__":
    main()
