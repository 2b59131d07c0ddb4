#!/usr/bin/env python3
"""Cuts small excerpts out of the reference's example READ DATA (not source) for the C1-style
parity tests: savage/example/input_fas (merged singles ~400-490 bp + 2x250 pairs, 25 distinct
quality values, N and Q0 bases) and polyte/example/input (2x250, 35 distinct quality values ->
exercises the 16-bit-symbol path).  Run in the build container:  python tests/golden/make_example_excerpt.py"""
import gzip
import os

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def head(path, n_records):
    with open(path, "rb") as f:
        lines = f.read().split(b"\n")
    return b"\n".join(lines[: 4 * n_records]) + b"\n"


def main():
    out = {
        "savage_singles.fastq": head(f"{REF}/savage/example/input_fas/singles.fastq", 450),
        "savage_paired1.fastq": head(f"{REF}/savage/example/input_fas/paired1.fastq", 200),
        "savage_paired2.fastq": head(f"{REF}/savage/example/input_fas/paired2.fastq", 200),
        "polyte_forward.fastq": head(f"{REF}/polyte/example/input/forward.fastq", 500),
        "polyte_reverse.fastq": head(f"{REF}/polyte/example/input/reverse.fastq", 500),
    }
    for name, data in out.items():
        with gzip.GzipFile(os.path.join(HERE, name + ".gz"), "wb", mtime=0) as f:
            f.write(data)
        print(name, len(data), "bytes")


if __name__ == "__main__":
    main()
