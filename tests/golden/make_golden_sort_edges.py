#!/usr/bin/env python3
"""Generates tests/golden/sort_edges.json with the reference's own OverlapGraph::sortEdges.

Runs only in the build container (needs /root/reference): `make -C oracle ref` compiles the fragment probe
oracle/_ref/libhcref_edgecalc.so, which also holds src/OverlapGraph.cpp:722-764 (sortEdges) piped verbatim behind
build-owned class shells.  The probe's frag_sort_edges builds a genuine graph by addEdge calls in the given order, calls
sortEdges and returns adj_out (list order) and adj_in.  Cases: random graphs with few distinct lengths (ties on the
non-overlap length, then on vertex2 for the two orientation classes of one pair), vertices with more than 16 and more
than 100 out-edges (std::sort leaves insertion sort there: its tie order is part of the behaviour), non-overlap lengths
that wrap below zero in unsigned arithmetic.  The vectors are data; no reference source is stored.
"""
import ctypes as C
import importlib.util
import json
import os
import random
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("make_golden_ec", os.path.join(ROOT, "tests", "golden", "make_golden_ec.py"))
mg = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mg)
FragEdge = mg.FragEdge


def random_case(seed, V, n, lens, len_choices, hubs):
    rng = random.Random(seed)
    read_len = [rng.choice(lens) for _ in range(V)]
    edges = []
    seen = set()
    while len(edges) < n:
        a = rng.choice(hubs) if hubs and rng.random() < 0.6 else rng.randrange(V)
        b = rng.randrange(V)
        o1, o2 = rng.randrange(2), rng.randrange(2)
        if a == b or (a, b, o1 == o2) in seen or (b, a, o1 == o2) in seen:  # one edge per slot, as the insert leaves it
            continue
        seen.add((a, b, o1 == o2))
        l1 = rng.choice(len_choices)
        l2 = rng.choice([0, 0, rng.choice(len_choices)])
        # pos1 != 0: the insert of a product graph leaves such edges as given (no swap towards the smaller vertex id)
        edges.append([float(rng.choice([0.97, 0.99, 1.0])).hex(), float(rng.choice([0.0, 0.01])).hex(), rng.choice([3, 12, 40]), rng.choice([0, 7]),
                      rng.choice([-5, 0, 9]), rng.choice([-2, 0, 4]), o1, o2, ord(rng.choice("-12")), a, b, rng.choice([100, 77]), l1, l2])
    return read_len, edges


def run(ref, V, read_len, edges):
    n = len(edges)
    arr = (FragEdge * max(n, 1))()
    for k, e in enumerate(edges):
        arr[k] = FragEdge(float.fromhex(e[0]), float.fromhex(e[1]), e[2], e[3], e[4], e[5], e[6], e[7], e[8], 0, 0, e[9], e[10], e[11], e[12] + e[13],
                          e[12], e[13])
    out = (FragEdge * max(n, 1))()
    rl = np.ascontiguousarray(read_len, dtype=np.uint32)
    off = np.zeros(V + 1, np.uint64)
    nodes = np.zeros(max(n, 1), np.uint64)
    rc = ref.frag_sort_edges(arr, n, V, rl.ctypes.data, out, off.ctypes.data, nodes.ctypes.data)
    assert rc == 0, rc
    flat = [[float(o.score).hex(), float(o.mismatch_rate).hex(), o.pos1, o.pos2, o.pos3, o.pos4, o.ori1, o.ori2, o.ord, int(o.v1), int(o.v2), o.perc,
             o.len1, o.len2] for o in out[:n]]
    return flat, off.tolist(), nodes[:n].tolist()


def main():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True)
    ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libhcref_edgecalc.so"))
    ref.frag_sort_edges.restype = C.c_int
    ref.frag_sort_edges.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    cases = []
    for name, seed, V, n, lens, len_choices, hubs in (
            ("small_ties", 1, 8, 40, [300, 300, 450], [100, 120], []),
            ("hubs_over_16", 2, 40, 420, [300, 600], [90, 100, 110], [3, 17]),
            ("hub_over_100", 3, 160, 520, [250, 300, 1000], [80, 100], [5]),
            ("wrapping_lengths", 4, 25, 200, [100, 150], [60, 90, 140, 200], [0, 1]),  # 2 * overlap_len may exceed len1 + len2
            ("sparse", 5, 400, 300, [300], [100, 150, 151], []),
            ("empty", 6, 5, 0, [300], [100], [])):
        read_len, edges = random_case(seed, V, n, lens, len_choices, hubs)
        out, off, nodes = run(ref, V, read_len, edges)
        cases.append(dict(name=name, V=V, read_len=read_len, edges_in=edges, edges_out=out, in_off=off, in_nodes=nodes))
        deg = max([off[v + 1] - off[v] for v in range(V)], default=0)
        print(name, "edges", len(edges), "max in-degree", deg)
    fields = ["score(hex)", "mismatch_rate(hex)", "pos1", "pos2", "pos3", "pos4", "ori1", "ori2", "ord", "v1", "v2", "perc", "len1", "len2"]
    with open(os.path.join(ROOT, "tests", "golden", "sort_edges.json"), "w") as f:
        json.dump(dict(note="OverlapGraph::sortEdges of the reference (src/OverlapGraph.cpp:722-764) through the fragment probe; "
                            "edges_in = addEdge order, edges_out = adj_out vertex by vertex in list order, in_off/in_nodes = adj_in",
                       edge_fields=fields, cases=cases), f, separators=(",", ":"))


if __name__ == "__main__":
    main()
