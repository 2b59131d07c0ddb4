#!/usr/bin/env python3
"""Times find-next-overlaps (FNO=1 and FNO=3, include/hcfno.h) on a synthetic iteration of SAVAGE-like size and,
beside it, the sequential oracle (oracle/fno_oracle.cpp: std::set<std::string> like the reference) on the same input.
Prints one JSON line per mode.  Host code only (no GPU work)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from haploconduct_amd import fno as F  # noqa: E402


def big_fno1(n_nodes, n_srs, n_edges, seed=1, dup=False):
    """dup: the vertices of --add_duplicates (vertex r and r + n_nodes / 2 are the two strands of read r; the stored non-edges carry
    the orientation of the strand their vertices lie on); the caller sets F.ADD_DUPLICATES in flags."""
    rng = np.random.default_rng(seed)
    paired = rng.random(n_nodes) < 0.4
    if dup:
        paired[n_nodes // 2:] = paired[:n_nodes // 2]
    k = 4
    base = rng.integers(0, n_nodes, size=n_srs)
    cl = (base[:, None] + np.arange(k)[None, :] * 7919) % n_nodes  # k distinct vertices per super-read
    visited = np.zeros(n_nodes, bool)
    visited[cl.ravel()] = True
    nodes = np.zeros(n_nodes, F.FNO_READ_DTYPE)
    nodes["len1"], nodes["len2"] = rng.integers(100, 300, n_nodes), np.where(paired, rng.integers(100, 300, n_nodes), 0)
    if dup:
        for f in ("len1", "len2"):
            nodes[f][n_nodes // 2:] = nodes[f][:n_nodes // 2]
    nodes["paired"], nodes["visited"], nodes["orientation"] = paired, visited, rng.integers(0, 2, n_nodes)
    nodes["id"] = np.where(visited, 0, np.cumsum(~visited) - 1)
    n_unvisited = int((~visited).sum())
    srs = np.zeros(n_srs, F.FNO_READ_DTYPE)
    sp = np.sort(rng.random(n_srs) < 0.4)  # single-end super-reads first, then paired (the API's contract)
    srs["id"] = n_unvisited + np.arange(n_srs)
    srs["len1"], srs["len2"], srs["paired"] = rng.integers(300, 900, n_srs), np.where(sp, rng.integers(300, 900, n_srs), 0), sp
    sub = np.zeros(n_srs * k, F.FNO_SUBREAD_DTYPE)
    sub["node"] = cl.ravel()
    trimmed = rng.random(n_srs * k) < 0.2
    sub["index1"] = np.where(trimmed, 0, rng.integers(0, 400, n_srs * k))
    sub["startpos1"] = np.where(trimmed, rng.integers(0, 50, n_srs * k), 0)
    sub["index2"], sub["startpos2"] = sub["index1"], sub["startpos1"]

    def edges(n, score):
        e = np.zeros(n, F.FNO_EDGE_DTYPE)
        e["v1"] = rng.integers(0, n_nodes, n)
        e["v2"] = (e["v1"] + 1 + rng.integers(0, n_nodes - 1, n)) % n_nodes
        pp = paired[e["v1"]] & paired[e["v2"]]
        anyp = paired[e["v1"]] | paired[e["v2"]]
        e["pos1"], e["pos2"] = rng.integers(0, 200, n), np.where(anyp, rng.integers(0, 200, n), 0)
        e["len1"], e["len2"] = rng.integers(1, 250, n), np.where(anyp, rng.integers(1, 250, n), 0)
        e["perc"], e["score"] = rng.integers(0, 100, n), score
        e["ord"] = np.where(pp, np.where(rng.random(n) < 0.5, ord("1"), ord("2")), ord("-"))
        e["ori1"], e["ori2"] = rng.integers(0, 2, n), rng.integers(0, 2, n)
        return e[np.argsort(e["v1"], kind="stable")]

    inp = F.Fno1Input.__new__(F.Fno1Input)
    inp.nodes, inp.srs = nodes, srs
    inp.clique_off, inp.clique_nodes = (np.arange(n_srs + 1) * k).astype(np.uint64), cl.ravel().astype(np.uint64)
    inp.subread_off, inp.subreads = (np.arange(n_srs + 1) * k).astype(np.uint64), sub
    inp.graph_edges, inp.branching_edges, inp.nonedges = edges(n_edges, 0.99), edges(n_edges // 20, 1.0), edges(n_edges // 2, 0.0)
    if dup:
        ne, half = inp.nonedges, n_nodes // 2
        ne = ne[(ne["v1"] % half) != (ne["v2"] % half)]
        ne["ori1"], ne["ori2"] = ne["v1"] < half, ne["v2"] < half
        inp.nonedges = ne
    inp.inclusion_off, inp.inclusion_edges, inp.n_inclusion_groups = np.zeros(1, np.uint64), np.zeros(0, F.FNO_EDGE_DTYPE), 0
    inp.new_read_count, inp.edge_threshold, inp.flags, inp.n_threads = n_unvisited + n_srs, 0.97, F.RESOLVE_ORIENTATIONS, 0
    return inp


def big_fno3(n_srs, n_originals, seed=2):
    rng = np.random.default_rng(seed)
    k = 6
    srs = np.zeros(n_srs, F.FNO_READ_DTYPE)
    n_single = n_srs // 2
    srs["id"] = rng.permutation(n_srs)
    srs["paired"] = np.arange(n_srs) >= n_single
    srs["len1"], srs["len2"] = rng.integers(200, 900, n_srs), np.where(srs["paired"] != 0, rng.integers(200, 900, n_srs), 0)
    base = rng.integers(0, n_originals, n_srs)
    o = np.zeros(n_srs * k, F.FNO_ORIGINAL_DTYPE)
    o["original_id"] = ((base[:, None] + np.arange(k)[None, :] * 7919) % n_originals).ravel()
    o["index1"], o["index2"] = rng.integers(0, 700, n_srs * k), rng.integers(0, 700, n_srs * k)
    inp = F.Fno3Input.__new__(F.Fno3Input)
    inp.srs, inp.counts = srs, (n_single, n_srs - n_single, 0)
    inp.orig_off, inp.originals = (np.arange(n_srs + 1) * k).astype(np.uint64), o
    inp.new_read_count, inp.original_readcount, inp.flags, inp.n_threads = n_srs, n_originals, 0, 0
    return inp


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=1_000_000)
    ap.add_argument("--srs", type=int, default=250_000)
    ap.add_argument("--edges", type=int, default=4_000_000)
    ap.add_argument("--threads", type=int, nargs="*", default=[1, 8, 0])
    ap.add_argument("--no-oracle", action="store_true")
    a = ap.parse_args()
    inp = big_fno1(a.nodes, a.srs, a.edges)
    res = {"mode": "fno1", "nodes": a.nodes, "super_reads": a.srs, "graph_edges": a.edges, "nonedges": a.edges // 2, "product_s": {}}
    ref_text = None
    os.environ["HC_FNO"] = "host"  # the host threads' form at several thread counts ...
    for th in a.threads:
        inp.n_threads = th
        t = time.perf_counter()
        text, cnt = F.find_next_overlaps(inp)
        res["product_s"][str(th or os.cpu_count())] = round(time.perf_counter() - t, 3)
        assert ref_text is None or text == ref_text
        ref_text = text
    os.environ.pop("HC_FNO", None)
    res["lines"], res["bytes"] = cnt["n_lines"], len(ref_text)
    inp.n_threads = 0  # ... and the default routing: the device takes the second half when there is one
    runs, lib_runs = [], []
    for _ in range(3):
        t = time.perf_counter()
        text, dcnt = F.find_next_overlaps(inp)
        runs.append(round(time.perf_counter() - t, 3))
        lib_runs.append(round(F.last_run_s, 3))
        if not F.last_on_device:
            break
        assert text == ref_text and dcnt == cnt
    if F.last_on_device:
        res["device_form_s"] = runs                  # with this harness's copy of the text into a bytes object
        res["device_form_library_s"] = lib_runs      # hc_fno1_run alone
        res["device_level"] = F.last_device_level    # 2: walk and look-ups on the device too
        os.environ["HC_FNO_WALK"] = "host"           # round 2's split: walk and look-ups with the host threads
        F.find_next_overlaps(inp)
        res["walk_on_host_library_s"] = round(F.last_run_s, 3)
        os.environ.pop("HC_FNO_WALK", None)
    if not a.no_oracle:
        from tests import _fno as T
        lib = T.load_oracle()
        t = time.perf_counter()
        want, wc = T.oracle_fno1(lib, inp)
        res["oracle_s"] = round(time.perf_counter() - t, 3)
        res["identical_to_oracle"] = bool(want == ref_text and wc == cnt)
    # the reference's own findNextOverlaps() (fragment probe, single thread as SRBuilder forces it, SRBuilder.h:92) on the
    # same input without the stored non-edges (the one branch the probe cannot run), and the product on that input
    import ctypes as C
    import tempfile

    ref_path = os.path.join(ROOT, "oracle", "_ref", "libhcref_fno.so")
    if os.path.exists(ref_path):
        ref = C.CDLL(ref_path)
        vp = C.c_void_p
        ref.frag_fno1_run.argtypes = [C.POINTER(F.hc_fno1_input), C.c_char_p, C.POINTER(vp), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        ref.frag_fno_free.argtypes = [vp]
        saved, inp.nonedges = inp.nonedges, np.zeros(0, F.FNO_EDGE_DTYPE)
        inp.flags |= F.OPTIMIZE
        inp.n_threads = 0
        t = time.perf_counter()
        ptext, pcnt = F.find_next_overlaps(inp)
        tp = time.perf_counter() - t
        s_, text, nb, nl = inp.struct(), vp(), C.c_uint64(), C.c_uint64()
        with tempfile.TemporaryDirectory() as d:
            t = time.perf_counter()
            ref.frag_fno1_run(C.byref(s_), d.encode(), C.byref(text), C.byref(nb), C.byref(nl))
            tr = time.perf_counter() - t
        same = C.string_at(text, nb.value) == ptext
        ref.frag_fno_free(text)
        res["without_nonedges"] = {"lines": pcnt["n_lines"], "product_s": round(tp, 3), "product_library_s": round(F.last_run_s, 3), "reference_own_code_s": round(tr, 3),
                                   "identical_to_reference": bool(same)}
        inp.nonedges = saved
        inp.flags &= ~F.OPTIMIZE
    print(json.dumps(res))

    inp3 = big_fno3(a.srs, a.nodes)
    res = {"mode": "fno3", "super_reads": a.srs, "originals": a.nodes, "product_s": {}}
    ref_text = None
    for th in a.threads:
        inp3.n_threads = th
        t = time.perf_counter()
        text, cnt = F.find_next_overlaps3(inp3)
        res["product_s"][str(th or os.cpu_count())] = round(time.perf_counter() - t, 3)
        assert ref_text is None or text == ref_text
        ref_text = text
    res["lines"], res["candidates"] = cnt["n_lines"], cnt["candidates"]
    if not a.no_oracle:
        t = time.perf_counter()
        want, wc = T.oracle_fno3(lib, inp3)
        res["oracle_s"] = round(time.perf_counter() - t, 3)
        res["identical_to_oracle"] = bool(want == ref_text and wc == cnt)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
