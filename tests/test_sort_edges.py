"""OverlapGraph::sortEdges (reference src/OverlapGraph.cpp:722-764, the call that follows construct_edges in every
workflow): the oracle restatement and the product's host implementation against vectors produced by the reference's own
code (tests/golden/sort_edges.json, made by tests/golden/make_golden_sort_edges.py through the fragment probe)."""
import ctypes as C
import json
import os
import random

import numpy as np
import pytest

import haploconduct_amd as hc
from haploconduct_amd import host

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = json.load(open(os.path.join(HERE, "golden", "sort_edges.json")))


def _recs(rows):
    out = np.zeros(len(rows), dtype=host.EDGE_DTYPE)
    for k, e in enumerate(rows):
        r = out[k]
        r["score"], r["mismatch_rate"] = float.fromhex(e[0]), float.fromhex(e[1])
        r["pos1"], r["pos2"], r["pos3"], r["pos4"], r["ori1"], r["ori2"], r["ord"] = e[2:9]
        r["v1"], r["v2"], r["read1"], r["read2"] = e[9], e[10], e[9], e[10]
        r["perc"], r["len1"], r["len2"], r["len0"] = e[11], e[12], e[13], e[12] + e[13]
    return out


@pytest.fixture(scope="module")
def graph_oracle():
    lib = C.CDLL(os.path.join(ROOT, "oracle", "libgraphoracle.so"))
    lib.hco_sort_edges.restype = C.c_int
    lib.hco_sort_edges.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]

    def sort_edges(edges, V, read_len):
        edges = np.ascontiguousarray(edges)
        out = np.zeros(max(edges.size, 1), dtype=host.EDGE_DTYPE)
        off = np.zeros(V + 1, np.uint64)
        nodes = np.zeros(max(edges.size, 1), np.uint64)
        rl = np.ascontiguousarray(read_len, dtype=np.uint32)
        assert lib.hco_sort_edges(edges.ctypes.data, edges.size, V, rl.ctypes.data, out.ctypes.data, off.ctypes.data, nodes.ctypes.data) == 0
        return out[: edges.size], off, nodes[: edges.size]

    return sort_edges


def _grouped(recs):
    """adj_out order of a graph built by addEdge calls in the given order: by v1, list order = call order"""
    return recs[np.argsort(recs["v1"], kind="stable")]


@pytest.mark.parametrize("case", GOLD["cases"], ids=[c["name"] for c in GOLD["cases"]])
def test_oracle_matches_the_reference_sort_edges(graph_oracle, case):
    got, off, nodes = graph_oracle(_grouped(_recs(case["edges_in"])), case["V"], case["read_len"])
    assert got.tobytes() == _recs(case["edges_out"]).tobytes()
    assert off.tolist() == case["in_off"] and nodes.tolist() == case["in_nodes"]


@pytest.mark.parametrize("threads", [1, 8])
@pytest.mark.parametrize("case", GOLD["cases"], ids=[c["name"] for c in GOLD["cases"]])
def test_host_graph_sort_edges_matches_the_reference(case, threads):
    st = hc.Settings(flags=hc.records.FLAG_RESOLVE_ORIENTATIONS, n_threads=threads)
    g = host.HostGraph(case["V"], st)
    for e in _recs(case["edges_in"]):
        assert g.insert(e.copy()) == 0  # the vectors hold one edge per slot and no pos1 == 0: the insert is a plain addEdge
    before, _, _ = g.get()
    assert before.tobytes() == _grouped(_recs(case["edges_in"])).tobytes()
    g.sort_edges(case["read_len"])
    after, _, _ = g.get()
    assert after.tobytes() == _recs(case["edges_out"]).tobytes()
    off, nodes = g.in_lists(after.size)
    assert off.tolist() == case["in_off"] and nodes.tolist() == case["in_nodes"]


def test_sort_edges_on_large_random_graphs_matches_the_oracle(graph_oracle):
    """Threaded path (>= 2^14 edges): hubs with hundreds of out-edges and many ties, against the oracle."""
    rng = random.Random(9)
    V, n = 3000, 40000
    read_len = [rng.choice([250, 300, 300, 500]) for _ in range(V)]
    rows, seen = [], set()
    hubs = [5, 77, 1500]
    while len(rows) < n:
        a = rng.choice(hubs) if rng.random() < 0.05 else rng.randrange(V)
        b = rng.randrange(V)
        o1, o2 = rng.randrange(2), rng.randrange(2)
        if a == b or (min(a, b), max(a, b), o1 == o2) in seen:
            continue
        seen.add((min(a, b), max(a, b), o1 == o2))
        rows.append([float(0.99).hex(), float(0.0).hex(), 3, 0, rng.choice([-4, 6]), 0, o1, o2, ord("-"), a, b, 90, rng.choice([100, 110, 120]), 0])
    recs = _recs(rows)
    for threads in (1, 8):
        g = host.HostGraph(V, hc.Settings(flags=hc.records.FLAG_RESOLVE_ORIENTATIONS, n_threads=threads))
        assert g.resolve(recs) == 0
        before, _, _ = g.get()
        assert before.size == n
        want, woff, wnodes = graph_oracle(before, V, read_len)
        g.sort_edges(read_len)
        after, _, _ = g.get()
        assert after.tobytes() == want.tobytes()
        off, nodes = g.in_lists(n)
        assert np.array_equal(off, woff) and np.array_equal(nodes, wnodes)
        # and the graph keeps answering slot questions correctly afterwards: a duplicate of an existing edge is a duplicate
        e = after[123].copy()
        assert g.insert(e) == 0
        assert g.get()[2]["dup_count"] == 1


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libhcref_edgecalc.so")),
                    reason="the reference fragment probe is built only where /root/reference exists (it ships with the repository's built files)")
def test_sort_edges_against_the_live_reference_probe(graph_oracle):
    """Beyond the committed vectors: seeded random graphs — hubs with up to a thousand out-edges, two or three distinct
    lengths so that nearly everything ties — through the reference's own sortEdges (fragment probe), the oracle and the
    product."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("make_golden_sort_edges", os.path.join(HERE, "golden", "make_golden_sort_edges.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libhcref_edgecalc.so"))
    ref.frag_sort_edges.restype = C.c_int
    ref.frag_sort_edges.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    for seed in range(12):
        V = [30, 300, 1500][seed % 3]
        n = [400, 3000, 2500][seed % 3]
        read_len, rows = mg.random_case(100 + seed, V, n, [300, 300, 450][: 1 + seed % 3], [100, 101][: 1 + seed % 2], [1, 2] if seed % 2 else [0])
        want_rows, want_off, want_nodes = mg.run(ref, V, read_len, rows)
        want = _recs(want_rows)
        got, off, nodes = graph_oracle(_grouped(_recs(rows)), V, read_len)
        assert got.tobytes() == want.tobytes() and off.tolist() == want_off and nodes.tolist() == want_nodes, f"oracle, seed {seed}"
        g = host.HostGraph(V, hc.Settings(flags=hc.records.FLAG_RESOLVE_ORIENTATIONS, n_threads=4))
        assert g.resolve(_recs(rows)) == 0
        g.sort_edges(read_len)
        after, _, _ = g.get()
        assert after.tobytes() == want.tobytes(), f"product, seed {seed}"
        poff, pnodes = g.in_lists(after.size)
        assert poff.tolist() == want_off and pnodes.tolist() == want_nodes
