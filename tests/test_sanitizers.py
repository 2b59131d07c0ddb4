"""AddressSanitizer + UBSan over the host-only code (FASTQ reader, overlaps tokenizer / record parser /
parallel block parser / prefilter, serial insert) and over the oracle: a CPU build with g++/gcc
-fsanitize=address,undefined driven through the same C entry points on hostile inputs.  (GPU
sanitizers are not available on the pool; the device code is covered by the parity tests.)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "build", "asan")

DRIVER = r'''
import ctypes as C, os, random, sys
host = C.CDLL(os.environ["HC_ASAN_HOST"])
orc = C.CDLL(os.environ["HC_ASAN_ORACLE"])
rng = random.Random(1)
d = os.environ["HC_TMP"]

class S(C.Structure):
    _fields_ = [("edge_threshold", C.c_double), ("ov_threshold", C.c_double), ("merge_contigs", C.c_double),
                ("mismatch", C.c_double), ("min_read_len", C.c_uint32), ("min_overlap_len", C.c_uint32),
                ("min_overlap_perc", C.c_uint32), ("flags", C.c_uint32), ("max_overlaps", C.c_uint64),
                ("device", C.c_int32), ("n_threads", C.c_uint32), ("device_mask", C.c_uint32), ("reserved", C.c_uint32)]
class P(C.Structure):
    _fields_ = [(k, C.c_char_p) for k in ("singles", "p1", "p2", "ids", "ov", "out")] + [("max_reads", C.c_uint64)]
class V(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("bases", "quals", "seq_off", "first", "ids")] + [(k, C.c_uint32) for k in ("n_reads", "n_seq", "n_single", "n_paired")]

# 1. tokenizer + record parser on hostile lines
host.hc_host_split_line.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
host.hc_host_parse_overlap.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.c_void_p, C.c_char_p]
alphabet = ["\t", "\t", " ", "0", "12", "-", "+", "s", "p", "", "99999999999999999999", "-5", "0x1F", "\x00", "\xff", "1" * 80]
out = C.create_string_buffer(256); text = C.create_string_buffer(256)
off = (C.c_uint32 * 64)(); ln = (C.c_uint32 * 64)()
for _ in range(20000):
    line = "".join(rng.choice(alphabet) for _ in range(rng.randrange(0, 40))).encode("latin1")
    for sp in (0, 1):
        host.hc_host_split_line(line, len(line), sp, off, ln, rng.choice([0, 1, 13, 14, 64]))
        host.hc_host_parse_overlap(line, len(line), sp, out, text)
good = b"7\t9\t12\t30\t1\t+\t-\t80\t70\t120\t105\tp\tp"
assert host.hc_host_parse_overlap(good, len(good), 0, out, text) == 0

# 2. FASTQ reader on hostile files (both the sequential line reader and the mapped multi-threaded one), then parser + prefilter with many threads
def w(name, data):
    open(d + name, "wb").write(data); return (d + name).encode()
fq = b"".join(b"@%d\nACGTNACGTN%s\n+\nIIIIIIIIII%s\n" % (i, b"ACGT" * (i % 7), b"5555" * (i % 7)) for i in range(300))
s = w("s.fastq", fq)
for mode in (0, 1):
    if mode:  # the mapped reader on tiny inputs: three threads, one record each at least
        os.environ.update(HC_FASTQ_THREADS="3", HC_FASTQ_PARALLEL_MIN="0", HC_FASTQ_GRAIN="1")
    for bad in (b"", b"@1\n", b"@1\nACGT\n+\nII\n", b"x\nACGT\n+\nIIII\n", b"@1\n\n+\n\n", b"@1\nACGT\n+\nIIII", b"\n\n\n\n",
                b"@1\nAC\n+\nII\n@2\nACG\n+\nIII\n@3\nA\n+\nI\n@4\nAC\n+\nI", b"@1\r\nAC\r\n+\r\nII\r\n" * 7):
        h = C.c_void_p(); v = V(); p = P(w("bad.fastq", bad), None, None, None, None, None, 0)
        rc = host.hc_host_fastq_load(C.byref(h), C.byref(p), C.byref(v))
        if rc == 0: host.hc_host_fastq_free(h)
        h = C.c_void_p(); v = V(); p = P(None, w("bad1.fastq", bad), w("bad2.fastq", bad[: len(bad) // 2 * 2]), None, None, None, 0)
        rc = host.hc_host_fastq_load(C.byref(h), C.byref(p), C.byref(v))
        if rc == 0: host.hc_host_fastq_free(h)
h = C.c_void_p(); v = V(); p = P(s, None, None, None, None, None, 0)
assert host.hc_host_fastq_load(C.byref(h), C.byref(p), C.byref(v)) == 0 and v.n_reads == 300
lines = []
for _ in range(30000):
    a, b = rng.randrange(300), rng.randrange(300)
    lines.append("%d\t%d\t%d\t-\t-\t%s\t%s\t%d\t-\t%d\t-\ts\ts" % (a, b, rng.randrange(12), rng.choice("+-"), rng.choice("+-"), rng.randrange(101), rng.randrange(1, 40)))
for junk in ("", "x", "1\t2", "\t\t\t", lines[0] + "\tz"):
    lines.insert(rng.randrange(len(lines)), junk)
ov = w("ov.txt", ("\n".join(lines)).encode())
n = C.c_uint64()
host.hc_host_parse_file.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
for threads in (1, 3, 16):
    for mo in (10**8, 1, 777):
        st = S(0.97, 0.9, 0, 0, 0, 10, 0, 2, mo, 0, threads)
        buf = (C.c_char * (32 * 40000))()
        assert host.hc_host_parse_file(C.byref(st), h, ov, buf, 40000, C.byref(n), None) == 0
# unknown read id -> error, not a crash
ov2 = w("ov2.txt", b"0\t9999\t0\t-\t-\t+\t+\t100\t-\t30\t-\ts\ts\n")
st = S(0.97, 0.9, 0, 0, 0, 10, 0, 2, 10**8, 0, 4)
assert host.hc_host_parse_file(C.byref(st), h, ov2, None, 0, C.byref(n), None) != 0
host.hc_host_parse_text.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
for raw in (b"", b"\n", open(ov, "rb").read(), open(ov, "rb").read()[:-7], b"\t\t\n" * 5000):  # the memory-backed parser on the same hostile bytes
    host.hc_host_parse_text(C.byref(st), h, raw, len(raw), None, 0, C.byref(n), None)
host.hc_host_fastq_free(h)

# 3. serial insert under duplicates
g = C.c_void_p(); st = S(0.97, 0.9, 0, 0, 0, 10, 0, 2 | 4, 10**8, 0, 1)
assert host.hc_host_graph_new(C.byref(g), 10, C.byref(st)) == 0
import struct
for _ in range(5000):
    a, b = rng.sample(range(10), 2)
    rec = struct.pack("<ddiiiiBBBBIIIQQiiii", rng.choice([0.97, 0.98, 1.0]), rng.choice([0.0, 0.1]), rng.choice([0, 3]), 0, rng.choice([-2, 2]), 0,
                      rng.randrange(2), rng.randrange(2), ord(rng.choice("-12")), 0, a, b, 0, a, b, rng.choice([100, 50]), 100, 100, 0)
    assert host.hc_host_graph_insert(g, rec) == 0
host.hc_host_graph_free(g)

# 3a. the threaded resolution + bulk fill (slot partitions, vertex-range fill, concurrent slot index), then serial inserts on top
g = C.c_void_p(); st = S(0.97, 0.9, 0, 0, 0, 10, 0, 2 | 4, 10**8, 0, 8)
assert host.hc_host_graph_new(C.byref(g), 400, C.byref(st)) == 0
recs = []
for _ in range(30000):
    a, b = rng.sample(range(400), 2)
    recs.append(struct.pack("<ddiiiiBBBBIIIQQiiii", rng.choice([0.97, 0.98, 1.0]), rng.choice([0.0, 0.1]), rng.choice([0, 3]), 0, rng.choice([-2, 2]), 0,
                            rng.randrange(2), rng.randrange(2), ord(rng.choice("-12")), 0, a, b, 0, a, b, rng.choice([100, 50]), 100, 100, 0))
host.hc_host_graph_resolve.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64]
assert host.hc_host_graph_resolve(g, b"".join(recs), len(recs)) == 0
for r in recs[:2000]:
    assert host.hc_host_graph_insert(g, r) == 0
lens = (C.c_uint32 * 400)(*[rng.choice([100, 150, 400]) for _ in range(400)])
host.hc_host_graph_sort_edges.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
assert host.hc_host_graph_sort_edges(g, lens, 400) == 0
assert host.hc_host_graph_sort_edges(g, lens, 399) != 0
host.hc_host_graph_free(g)

# 3a'. the device's hand-over (CSR -> list views into one array), inserts on top of the views, addEquivalentEdges, sortEdges
import numpy as np
R = 150; V2 = 2 * R
st = S(0.97, 0.9, 0, 0, 0, 10, 0, 2 | 1, 10**8, 0, 3)
def rec2():
    a, b = rng.sample(range(R), 2); o1, o2 = rng.randrange(2), rng.randrange(2)
    return struct.pack("<ddiiiiBBBBIIIQQiiii", rng.choice([0.97, 0.98, 1.0]), rng.choice([0.0, 0.1]), rng.choice([0, 3]), rng.choice([0, 2]), rng.choice([-2, 0, 2]),
                       rng.choice([-1, 0, 1]), o1, o2, ord(rng.choice("-12")), 0, a, b, 0, a if o1 else R + a, b if o2 else R + b, rng.choice([100, 50]), 100, 100, 0)
ga = C.c_void_p(); gb = C.c_void_p()
assert host.hc_host_graph_new(C.byref(ga), V2, C.byref(st)) == 0 and host.hc_host_graph_new(C.byref(gb), V2, C.byref(st)) == 0
for _ in range(5000):
    assert host.hc_host_graph_insert(ga, rec2()) == 0
host.hc_host_graph_get.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
host.hc_host_graph_get_in_lists.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
host.hc_host_graph_adopt.argtypes = [C.c_void_p] * 6
def dump(g):
    ne = C.c_uint64()
    assert host.hc_host_graph_get(g, None, 0, C.byref(ne), None, None) == 0
    e = np.zeros((ne.value, 80), np.uint8); inc = np.zeros(V2, np.uint8)
    assert host.hc_host_graph_get(g, e.ctypes.data, ne.value, C.byref(ne), inc.ctypes.data, None) == 0
    io = np.zeros(V2 + 1, np.uint64); inn = np.zeros(max(ne.value, 1), np.uint64)
    assert host.hc_host_graph_get_in_lists(g, io.ctypes.data, inn.ctypes.data, ne.value) == 0
    return e, inc, io, inn[:ne.value]
e, inc, io, inn = dump(ga)
v1 = e[:, 48:56].copy().view(np.uint64).ravel()
oo = np.zeros(V2 + 1, np.uint64); np.add.at(oo, v1.astype(np.int64) + 1, 1); oo = np.cumsum(oo).astype(np.uint64)
inn32 = inn.astype(np.uint32)
assert host.hc_host_graph_adopt(gb, e.ctypes.data, oo.ctypes.data, inn32.ctypes.data, io.ctypes.data, inc.ctypes.data) == 0
assert host.hc_host_graph_adopt(gb, e.ctypes.data, oo.ctypes.data, inn32.ctypes.data, io.ctypes.data, inc.ctypes.data) != 0  # not empty any more
del e, inn32, oo  # the graph owns copies; nothing may point into the caller's arrays
for _ in range(3000):
    r = rec2()
    assert host.hc_host_graph_insert(ga, r) == 0 and host.hc_host_graph_insert(gb, r) == 0
assert host.hc_host_graph_add_equivalent_edges(ga) == 0 and host.hc_host_graph_add_equivalent_edges(gb) == 0
lens2 = (C.c_uint32 * R)(*[rng.choice([100, 150, 400]) for _ in range(R)])
assert host.hc_host_graph_sort_edges(ga, lens2, R) == 0 and host.hc_host_graph_sort_edges(gb, lens2, R) == 0
da, db = dump(ga), dump(gb)
assert all(np.array_equal(x, y) for x, y in zip(da, db)) and da[0].shape[0] > 1000
host.hc_host_graph_free(ga); host.hc_host_graph_free(gb)

# 3b. SFO ingest on plausible and hostile files
host.hc_sfo2overlaps.argtypes = [C.c_char_p, C.c_char_p, C.c_uint64, C.c_uint64, C.c_void_p]
sfo = []
for _ in range(5000):
    a, b = rng.randrange(120), rng.randrange(120)
    sfo.append("%d %d %s %d %d %d %d %d" % (a, b, rng.choice("NNI"), rng.randrange(-100, 100), rng.randrange(-100, 100), rng.randrange(1, 150), rng.randrange(1, 150), rng.randrange(3)))
assert host.hc_sfo2overlaps(w("a.sfo", "\n".join(sfo).encode()), (d + "a.out").encode(), 40, 40, None) == 0
for bad in (b"", b"\n\n", b"1 2 N 3\n", b"1 2 N 0 0 0 0 0\n", b"1 2 N a b c d e\n", b"999 2 N 1 1 1 1 0\n", b"1 2 " + b"N" * 5000 + b" 1 1 1 1 0\n"):
    host.hc_sfo2overlaps(w("b.sfo", bad), (d + "b.out").encode(), 40, 40, None)

# 3b'. the records path of the same ingest (buckets sorted and matched on threads, open groups stitched across borders)
host.hc_sfo_records_to_overlaps.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64, C.c_uint64, C.c_void_p]
os.environ["HC_SFO_BUCKETS"] = "9"
raw = b"".join(struct.pack("<IIiiIIII", rng.randrange(120), rng.randrange(120), rng.randrange(-100, 100), rng.randrange(-100, 100),
                           rng.randrange(1, 150), rng.randrange(1, 150), rng.randrange(12), rng.randrange(2)) for _ in range(30000))
nl = C.c_uint64()
host.hc_sfo_records_to_overlaps(raw, 30000, (d + "c.out").encode(), 40, 40, C.byref(nl))
assert host.hc_sfo_records_to_overlaps(raw, 30000, (d + "c.out").encode(), 10, 10, C.byref(nl)) != 0  # ids out of range: an error
# ... and the chunk-fed matcher behind a sorted run (what hc_found_to_overlaps uses after the device's sort): same file
want_c = open(d + "c.out", "rb").read() if host.hc_sfo_records_to_overlaps(raw, 30000, (d + "c.out").encode(), 40, 40, C.byref(nl)) == 0 else None
for chunk in ("1", "5", "4096"):
    os.environ["HC_SFO_VIA_MATCHER"] = chunk
    assert host.hc_sfo_records_to_overlaps(raw, 30000, (d + "c2.out").encode(), 40, 40, C.byref(nl)) == 0
    assert want_c is not None and open(d + "c2.out", "rb").read() == want_c
del os.environ["HC_SFO_VIA_MATCHER"]
del os.environ["HC_SFO_BUCKETS"]

# 3c. find-next-overlaps on the scenarios the parent process saved (well-formed and hostile ones)
import glob
import numpy as np
vp = C.c_void_p
class F1(C.Structure):
    _fields_ = [("nodes", vp), ("n_nodes", C.c_uint64), ("srs", vp), ("n_srs", C.c_uint64), ("clique_off", vp), ("clique_nodes", vp),
                ("subread_off", vp), ("subreads", vp), ("graph_edges", vp), ("n_graph_edges", C.c_uint64), ("branching_edges", vp),
                ("n_branching_edges", C.c_uint64), ("nonedges", vp), ("n_nonedges", C.c_uint64), ("inclusion_off", vp), ("inclusion_edges", vp),
                ("n_inclusion_groups", C.c_uint64), ("new_read_count", C.c_uint64), ("edge_threshold", C.c_double), ("flags", C.c_uint32),
                ("n_threads", C.c_uint32)]
class F3(C.Structure):
    _fields_ = [("srs", vp), ("n_single", C.c_uint64), ("n_paired", C.c_uint64), ("n_trivial", C.c_uint64), ("orig_off", vp), ("originals", vp),
                ("new_read_count", C.c_uint64), ("original_readcount", C.c_uint64), ("flags", C.c_uint32), ("n_threads", C.c_uint32)]
def P(a):
    return a.ctypes.data if a.size else None
n_fno = 0
for f in sorted(glob.glob(d + "fno1_*.npz")):
    z = dict(np.load(f))
    s1 = F1(P(z["nodes"]), len(z["nodes"]) // 24, P(z["srs"]), len(z["srs"]) // 24, P(z["clique_off"]), P(z["clique_nodes"]), P(z["subread_off"]),
            P(z["subreads"]), P(z["graph_edges"]), len(z["graph_edges"]) // 48, P(z["branching_edges"]), len(z["branching_edges"]) // 48,
            P(z["nonedges"]), len(z["nonedges"]) // 48, P(z["inclusion_off"]), P(z["inclusion_edges"]), len(z["inclusion_off"]) - 1,
            int(z["scalars"][0]), 0.97, int(z["scalars"][1]), int(z["scalars"][2]))
    out = vp()
    rc = host.hc_fno1_run(C.byref(s1), C.byref(out))
    assert int(z["scalars"][3]) < 0 or (rc == 0) == (int(z["scalars"][3]) == 1), (f, rc)  # (< 0: either outcome, only the sanitizers judge)
    if rc == 0:
        host.hc_fno_output_free(out)
    n_fno += 1
for f in sorted(glob.glob(d + "fno3_*.npz")):
    z = dict(np.load(f))
    s3 = F3(P(z["srs"]), int(z["scalars"][0]), int(z["scalars"][1]), int(z["scalars"][2]), P(z["orig_off"]), P(z["originals"]), int(z["scalars"][3]),
            int(z["scalars"][4]), int(z["scalars"][5]), int(z["scalars"][6]))
    out = vp()
    rc = host.hc_fno3_run(C.byref(s3), C.byref(out))
    assert (rc == 0) == (int(z["scalars"][7]) == 1), (f, rc)
    if rc == 0:
        host.hc_fno_output_free(out)
    n_fno += 1
assert n_fno >= 14

# 4. oracle on odd inputs
orc.hco_overlap_score.restype = C.c_double
mr = C.c_double(); x = C.c_double(); mm = C.c_uint32(); nn = C.c_uint32(); pos = C.c_uint64(); stt = C.c_int()
for a, b, qa, qb, p in ((b"ACGT", b"ACGT", b"IIII", b"IIII", 0), (b"ACGT", b"AC", b"IIII", b"II", 3), (b"NNNN", b"ACGT", b"!!!!", b"IIII", 0),
                        (b"ACGT", b"ACGT", b"IIII", b"IIII", 77), (b"acgt", b"ACGT", b"IIII", b"IIII", 0), (b"ACGT", b"ACGT", b"  II", b"IIII", 0)):
    orc.hco_overlap_score(a, C.c_size_t(len(a)), b, C.c_size_t(len(b)), qa, qb, C.c_uint(p), C.c_uint(0), C.c_double(0.0), C.byref(mr), C.byref(x),
                          C.byref(mm), C.byref(nn), C.byref(pos), C.byref(stt))
print("sanitizer driver finished")
'''


def _save_fno_scenarios(d):
    """FNO inputs as raw bytes for the sanitizer child (which must not import the package: that would load the real library)."""
    import numpy as np

    from tests import _fno as T

    def raw(a):
        return np.frombuffer(np.ascontiguousarray(a).tobytes(), np.uint8)

    def save1(name, inp, ok):
        np.savez(d + name, nodes=raw(inp.nodes), srs=raw(inp.srs), clique_off=inp.clique_off, clique_nodes=inp.clique_nodes,
                 subread_off=inp.subread_off, subreads=raw(inp.subreads), graph_edges=raw(inp.graph_edges),
                 branching_edges=raw(inp.branching_edges), nonedges=raw(inp.nonedges), inclusion_off=inp.inclusion_off,
                 inclusion_edges=raw(inp.inclusion_edges), scalars=np.array([inp.new_read_count, inp.flags, inp.n_threads, ok], np.int64))

    for seed in range(6):
        save1(f"fno1_{seed}.npz", T.fno1_scenario(seed, n_nodes=80, n_srs=30, n_edges=800, with_extras=True, flags=[1, 3, 5][seed % 3],
                                                  n_threads=[1, 4][seed % 2]), 1)
    bad = T.fno1_scenario(3)
    bad.subreads["node"][0] = 10 ** 6
    save1("fno1_bad_a.npz", bad, 0)
    bad = T.fno1_scenario(3)
    bad.new_read_count = 3
    save1("fno1_bad_b.npz", bad, 0)
    bad = T.fno1_scenario(3)
    bad.graph_edges["v2"][5] = 10 ** 9
    save1("fno1_bad_c.npz", bad, 0)
    empty = T.fno1_scenario(3, n_edges=0)
    save1("fno1_empty.npz", empty, 1)
    for seed in range(3):  # --add_duplicates (flag 8): every kept stored non-edge and its opposite overlap
        dup = T.fno1_scenario(40 + seed, n_nodes=80, n_srs=30, n_edges=600, with_extras=True, flags=[8, 9, 10][seed], dup=True, n_threads=[1, 4, 3][seed],
                              paired_frac=[0.4, 1.0, 0.0][seed])
        dup.nonedges, _ = T.dup_nonedges(np.random.default_rng(seed), dup, 300)
        save1(f"fno1_dup_{seed}.npz", dup, 1)
    bad = T.fno1_scenario(41, n_nodes=80, n_srs=30, n_edges=600, with_extras=True, flags=9, dup=True, paired_frac=1.0)
    bad.nonedges, _ = T.dup_nonedges(np.random.default_rng(1), bad, 300)
    bad.nonedges["ord"][:] = ord("-")  # two paired reads: the reference's assert at src/FindNextOverlaps.cpp:759
    save1("fno1_dup_bad_a.npz", bad, 0)
    for seed in range(4):
        inp = T.fno3_scenario(seed, n_single=30, n_paired=20, n_trivial=25, n_originals=90, n_threads=[1, 4][seed % 2])
        ok = 1
        if seed == 3:
            inp.original_readcount, ok = 2, 0
        np.savez(d + f"fno3_{seed}.npz", srs=raw(inp.srs), orig_off=inp.orig_off, originals=raw(inp.originals),
                 scalars=np.array([*inp.counts, inp.new_read_count, inp.original_readcount, inp.flags, inp.n_threads, ok], np.int64))


def test_host_code_and_oracle_under_asan_ubsan(tmp_path):
    os.makedirs(BUILD, exist_ok=True)
    host_so = os.path.join(BUILD, "libhchost_asan.so")
    orc_so = os.path.join(BUILD, "liboracle_asan.so")
    hd = os.path.join(ROOT, "haploconduct_amd", "csrc", "host")
    stub = os.path.join(BUILD, "stub.cpp")
    # what the host files expect from the HIP side of the library: error text, and find-next-overlaps' device form (not present here)
    open(stub, "w").write('#include <cstdint>\n#include <functional>\n#include <string>\n'
                          'namespace hc { int set_last_error(int s, const std::string&) { return s; }\n'
                          'struct FnoItem; bool fno_device_wanted(uint64_t) { return false; }\n'
                          'bool fno_lines_on_device(const FnoItem*, uint64_t, bool, const std::function<char*(uint64_t)>&, uint64_t*, double*) { return false; }\n'
                          'bool fno3_lines_on_device(const FnoItem*, uint64_t, bool, const std::function<char*(uint64_t)>&, uint64_t*, double*) { return false; }\n'
                          'struct FnoWalkHost; bool fno1_walk_on_device(const FnoWalkHost&, const std::function<char*(uint64_t)>&, uint64_t*, uint64_t*, double*) { return false; } }\n'
                          'extern "C" { const char* hc_strerror(int) { return ""; } const char* hc_last_error(void) { return ""; } }\n')
    san = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g", "-O1", "-fPIC", "-shared"]
    r = subprocess.run(["g++", "-std=c++17", *san, "-pthread", "-o", host_so, os.path.join(hd, "host_model.cpp"),
                        os.path.join(hd, "OverlapsParser.cpp"), os.path.join(hd, "hc_host_api.cpp"),
                        os.path.join(hd, "Sfo2Overlaps.cpp"), os.path.join(hd, "FindNextOverlaps.cpp"), stub],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run(["gcc", "-std=gnu11", "-ffp-contract=off", "-fopenmp", *san, "-o", orc_so,
                        os.path.join(ROOT, "oracle", "hc_oracle.c"), "-lm"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    _save_fno_scenarios(str(tmp_path) + "/")
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    libstdcpp = subprocess.run(["gcc", "-print-file-name=libstdc++.so.6"], capture_output=True, text=True).stdout.strip()
    # libstdc++ must be loaded before the ASan runtime initialises, or its __cxa_throw interceptor has no target
    env = dict(os.environ, LD_PRELOAD=libasan + " " + libstdcpp, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", HC_ASAN_HOST=host_so,
               HC_ASAN_ORACLE=orc_so, HC_TMP=str(tmp_path) + "/", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-c", DRIVER], env=env, capture_output=True, text=True, timeout=600)
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
    assert r.returncode == 0 and "sanitizer driver finished" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])
