#!/usr/bin/env python3
"""Golden vectors for the SFO ingest: rust-overlaps' 8-column SFO lines ->
SAVAGE's 13-column overlaps file, as the reference's scripts/sfo2overlaps.py does it.

The reference script is Python 2 (print statements, xrange) and cannot be imported by this
container's Python 3.  It is translated IN MEMORY by lib2to3 (print / xrange only; nothing is written
to disk) and executed with one shim: `round` is bound to Python 2's round-half-away-from-zero, which
the script relies on for the overlap percentage (sfo2overlaps.py:189).  Its sort/uniq subprocess calls
run unchanged under LC_ALL=C.  Inputs are seeded synthetic SFO files (tests/golden/sfo/*.sfo); the
expected outputs are stored next to them (*.expected)."""
import math
import os
import random
import subprocess
import sys
import tempfile
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "sfo")
REF = "/root/reference/scripts/sfo2overlaps.py"


def py2_round(x):
    """Python 2's round(): nearest integer as a float, halves away from zero (C round())."""
    f = math.floor(abs(x))
    r = f + 1.0 if abs(x) - f >= 0.5 else float(f)
    return r if x >= 0 else -r


def load_reference_module():
    warnings.filterwarnings("ignore")
    from lib2to3 import refactor

    rt = refactor.RefactoringTool(refactor.get_fixers_from_package("lib2to3.fixes"))
    code = str(rt.refactor_string(open(REF).read(), "sfo2overlaps.py"))
    ns = {"__name__": "sfo2overlaps_ref", "round": py2_round}
    exec(compile(code, REF, "exec"), ns)
    return ns


def run_reference(ns, sfo_path, out_path, num_singles, num_pairs):
    cwd = os.getcwd()
    env_lc = os.environ.get("LC_ALL")
    os.environ["LC_ALL"] = "C"
    with tempfile.TemporaryDirectory() as d:  # the script writes tmp_overlaps.txt into the cwd
        os.chdir(d)
        argv = sys.argv
        sys.argv = ["sfo2overlaps.py", "--in", sfo_path, "--out", out_path, "--num_singles", str(num_singles), "--num_pairs", str(num_pairs)]
        try:
            stdout = sys.stdout
            sys.stdout = open(os.devnull, "w")
            ns["main"]()
        finally:
            sys.stdout = stdout
            sys.argv = argv
            os.chdir(cwd)
            if env_lc is None:
                del os.environ["LC_ALL"]
            else:
                os.environ["LC_ALL"] = env_lc


def synth_sfo(seed, n_singles, n_pairs, n_lines, sep="\t"):
    """Plausible SFO lines: sequences at random genome positions; overlaps reported for (idA, idB) in SFO id space."""
    rng = random.Random(seed)
    nseq = n_singles + 2 * n_pairs
    start, length = [], []
    for q in range(nseq):
        L = rng.randrange(150, 400) if q < n_singles else 150
        start.append(rng.randrange(0, 3000))
        length.append(L)
    for i in range(n_pairs):  # /2 lies 200..450 behind /1
        start[n_singles + n_pairs + i] = start[n_singles + i] + rng.randrange(200, 450)
    lines = []
    while len(lines) < n_lines:
        a, b = rng.randrange(nseq), rng.randrange(nseq)
        if a == b:
            continue
        sa, sb, la, lb = start[a], start[b], length[a], length[b]
        ovl = min(sa + la, sb + lb) - max(sa, sb)
        if ovl < 20:
            continue
        ori = "N" if rng.random() < 0.8 else "I"
        oha, ohb = sb - sa, (sb + lb) - (sa + la)
        ola = ovl
        olb = ovl if rng.random() < 0.9 else max(1, ovl + rng.randrange(-2, 3))  # indel in the overlap
        lines.append(sep.join(map(str, (a, b, ori, oha, ohb, ola, olb, rng.randrange(0, 3)))))
        if rng.random() < 0.05:
            lines.append(lines[-1])  # exact duplicate
        if rng.random() < 0.05:     # a second report for the same pair
            lines.append(sep.join(map(str, (a, b, ori, oha + 1, ohb + 1, ola - 1, olb - 1, 1))))
    # make sure paired matches exist: /1-/1 and /2-/2 (and single - /1, single - /2) of neighbouring reads
    for _ in range(n_lines // 4):
        if n_pairs >= 2:
            i, j = rng.sample(range(n_pairs), 2)
            d1, d2 = rng.randrange(0, 75), rng.randrange(-75, 75)
            o = rng.choice("NNNI")
            lines.append(sep.join(map(str, (n_singles + i, n_singles + j, o, d1, d1, 150 - d1, 150 - d1, 0))))
            lines.append(sep.join(map(str, (n_singles + n_pairs + i, n_singles + n_pairs + j, o, d2, d2, 150 - abs(d2), 150 - abs(d2), 0))))
        if n_pairs >= 1 and n_singles >= 1:
            k, i = rng.randrange(n_singles), rng.randrange(n_pairs)
            p1, p2 = rng.randrange(0, 100), rng.randrange(100, 240)
            lines.append(sep.join(map(str, (k, n_singles + i, "N", p1, p1 + 150 - length[k], min(150, length[k] - p1), min(150, length[k] - p1), 0))))
            lines.append(sep.join(map(str, (k, n_singles + n_pairs + i, "N", p2, p2 + 150 - length[k], max(1, min(150, length[k] - p2)), max(1, min(150, length[k] - p2)), 0))))
    rng.shuffle(lines)
    return lines


CASES = [  # name, seed, singles, pairs, lines, separator
    ("singles_only", 1, 60, 0, 600, "\t"),
    ("pairs_only", 2, 0, 40, 600, "\t"),
    ("mixed", 3, 30, 30, 900, "\t"),
    ("mixed_space_separated", 4, 20, 20, 500, " "),
]


def main():
    ns = load_reference_module()
    os.makedirs(OUT, exist_ok=True)
    for name, seed, s, p, n, sep in CASES:
        lines = synth_sfo(seed, s, p, n, sep)
        sfo = os.path.join(OUT, name + ".sfo")
        with open(sfo, "w") as f:
            f.write("\n".join(lines) + "\n")
        exp = os.path.join(OUT, name + ".expected")
        run_reference(ns, sfo, exp, s, p)
        with open(os.path.join(OUT, name + ".args"), "w") as f:
            f.write(f"{s} {p}\n")
        print(name, len(lines), "SFO lines ->", sum(1 for _ in open(exp)), "overlap lines")


if __name__ == "__main__":
    main()
