#!/usr/bin/env python3
"""Round 6 — which quality index the r-th most frequent quality value takes in the wide 8-bit encoding (hc_device.h: kWideRankLabel).

The 64 KiB table of the wide encoding is read with ds_read_b64: a wave's 64 reads are served in two groups of 32 lanes, one LDS cycle per
group when the 32 addresses fall into different banks or are equal (MI355X_MICROARCH.md, LDS), one more for every further DISTINCT address
in a bank.  The bank is the address's low byte; round 5's layout put (qa ^ qb) & 31 there, so every pair of equal qualities met in bank 0
at different rows.  This script simulates the cycles per group for a layout's low byte and an assignment of indices to frequency ranks,
over quality distributions of real reads (tests/golden/quality_histograms.json) and a few synthetic shapes, and searches the assignment
by pair swaps.  Low bytes that cost two VALU ops per four positions, as the plain XOR does:
    F1  x & 31                      (round 5)
    F3  ((x >> 1) ^ la) & 31        (the row's LOW five bits against x5..x1; the high byte then carries x0)
    F4  (x ^ (la >> 1)) & 31        (the row's HIGH five bits against x4..x0; the high byte carries x5 as before)   <- taken
Prints the held-out cycles per group and the assignment.  The one in hc_device.h is F4's from seed 7.
"""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
h = json.load(open(os.path.join(HERE, "..", "..", "tests", "golden", "quality_histograms.json")))


def cycles(slot, addr):
    n = slot.shape[0]
    key = np.sort(slot.astype(np.int64) * (1 << 20) + addr, axis=1)
    new = np.ones_like(key, bool)
    new[:, 1:] = key[:, 1:] != key[:, :-1]
    s = (key >> 20) + np.arange(n)[:, None] * 32
    cnt = np.bincount(s[new], minlength=n * 32).reshape(n, 32)
    return cnt.max(axis=1).mean()


def dists():
    out = {}
    for nm in ("polyte_forward", "polyte_reverse", "savage_singles", "savage_paired1"):
        v = np.array(sorted(h[nm]["counts"].values(), reverse=True), float)
        out[nm] = v / v.sum()
    for nm, z in (("zipf1", 1 / np.arange(1, 41) ** 1.0), ("zipf2", 1 / np.arange(1, 41) ** 2.0), ("uniform35", np.ones(35)),
                  ("geo0.7", 0.7 ** np.arange(42)), ("geo0.85", 0.85 ** np.arange(42))):
        out[nm] = z / z.sum()
    return out


def sample(p, n, seed):  # ranks of the two qualities of 32 lanes' positions, and their mismatch flags
    rng = np.random.default_rng(seed)
    return rng.choice(len(p), size=(n, 32), p=p), rng.choice(len(p), size=(n, 32), p=p), (rng.random((n, 32)) < 0.012).astype(int)


FORMS = {"F1": lambda la, lb, x: x, "F3": lambda la, lb, x: (x >> 1) ^ la, "F4": lambda la, lb, x: x ^ (la >> 1)}


def score(T, f, S):
    r = []
    for ra, rb, m in S.values():
        la, lb = T[ra], T[rb]
        x = la ^ lb
        r.append(cycles(f(la, lb, x) & 31, (la * 64 + x) * 2 + m))
    return np.array(r)


if __name__ == "__main__":
    D = dists()
    S = {nm: sample(p, 3000, 1) for nm, p in D.items()}
    S2 = {nm: sample(p, 8000, 99) for nm, p in D.items()}
    print("distributions:", list(D))
    print("round 5 (index = 16 + rank by byte value is what it amounted to; here by frequency rank), F1:", np.round(score(np.arange(16, 64), FORMS["F1"], S2), 2))
    rng = np.random.default_rng(7)
    for fn in ("F3", "F4", "F1"):
        f, best = FORMS[fn], None
        for restart in range(4):
            T = rng.permutation(np.arange(16, 64))
            s = score(T, f, S).sum()
            for it in range(1500):
                i = rng.integers(0, 16) if it % 3 else rng.integers(0, 48)
                j = rng.integers(0, 48)
                if i == j:
                    continue
                T2 = T.copy()
                T2[i], T2[j] = T2[j], T2[i]
                s2 = score(T2, f, S).sum()
                if s2 <= s:
                    T, s = T2, s2
            if best is None or s < best[0]:
                best = (s, T.copy())
        s, T = best
        print(fn, "held-out cycles per 32-lane group:", np.round(score(T, f, S2), 2), "sum", round(float(score(T, f, S2).sum()), 3))
        print("   assignment by frequency rank:", list(map(int, T)))
