"""INTEGRATION.md level A, compiled and run: the REFERENCE'S OWN src/EdgeCalculator.cpp:26-557 with the seven pieces of
integration/reference_patch/ inserted at the lines INTEGRATION.md names — the class members and constructor lines, the short cut
at the head of overlap_score, the three insertions in process_overlaps, the two new member functions — compiled against
include/hcedge.h and linked to libhcedge.so (oracle/Makefile: _ref/libhcref_edgecalc_patched.so, built in the build container,
shipped with the repository).  So here the reference's own Overlap / Read / Edge types, its compute_overlap (sequence choice,
sub-overlap combination, reverse offsets), its 3-way class, its serial insert with the tie-break chain and its OverlapGraph run
around scores that come from the MI355X.  The nine scenarios of tests/golden/ec/ (outputs of the UNPATCHED reference lines) must
come out byte for byte: edges in list order with score and mismatch rate as bit patterns, inclusions, nonedge_overlaps.txt,
inclusion_count, dup_count."""
import ctypes as C
import glob
import importlib.util
import os

import numpy as np
import pytest

from tests.test_ec_golden import CASES, load_case

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "_ref", "libhcref_edgecalc_patched.so")


@pytest.fixture(scope="module")
def patched():
    if not os.path.exists(LIB):
        pytest.skip("oracle/_ref/libhcref_edgecalc_patched.so is built only where /root/reference exists")
    spec = importlib.util.spec_from_file_location("make_golden_ec", os.path.join(ROOT, "tests", "golden", "make_golden_ec.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    ref = C.CDLL(LIB)
    vp = C.c_void_p
    ref.frag_process_overlaps.restype = C.c_int
    ref.frag_process_overlaps.argtypes = [C.POINTER(mg.FragSettings), vp, vp, vp, C.c_uint32, C.c_uint32, vp, C.c_uint64, C.c_char_p, vp,
                                          C.c_uint64, C.POINTER(C.c_uint64), vp, C.POINTER(vp), C.POINTER(C.c_uint64), vp]
    ref.frag_ec_free.argtypes = [vp]
    return mg, ref


def test_the_patch_files_are_what_integration_md_quotes():
    """Every line of every piece under integration/reference_patch/ appears in INTEGRATION.md, in order."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    pieces = sorted(glob.glob(os.path.join(ROOT, "integration", "reference_patch", "*.inc")))
    assert len(pieces) == 8  # seven pieces of level A, one of level B
    for p in pieces:
        text = open(p).read().rstrip("\n")
        assert text in doc, f"INTEGRATION.md does not quote {os.path.basename(p)} verbatim"


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[:-5] for p in CASES])
def test_reference_code_with_hip_scoring_reproduces_the_reference(patched, path):
    mg, ref = patched
    c, reads, st, want = load_case(path)
    edges, incl, nonedge, counters = mg.run_probe(ref, reads, c["lines"], c["settings"])
    assert len(edges) == len(c["edges"]), f"{len(edges)} edges, the unpatched reference built {len(c['edges'])}"
    assert edges == c["edges"], "an Edge differs (scores and mismatch rates compared as hex floats)"
    assert incl == c["inclusions"]
    assert nonedge == c["nonedge_overlaps"]
    assert counters == [c["inclusion_count"], c["dup_count"]]


def test_patched_construct_edges_on_a_file(patched, tmp_path):
    """The reference's own construct_edges (parser, prefilter, batches) over the patched process_overlaps, against the unpatched
    probe on the same file: 60 000 lines of 2x150 pairs."""
    from haploconduct_amd import host, synth

    mg, ref = patched
    plain_path = os.path.join(ROOT, "oracle", "_ref", "libhcref_edgecalc.so")
    if not os.path.exists(plain_path):
        pytest.skip("no unpatched probe")
    plain = C.CDLL(plain_path)
    reads, meta = synth.make_paired_dataset(3000, 4000, flip_frac=0.25, seed=41)
    cand = synth.paired_candidates(meta, n_candidates=60000, seed=42)
    path = str(tmp_path / "overlaps.txt")
    host.write_overlaps(path, cand, reads)
    seqs, quals = zip(*(reads.seq(q) for q in range(reads.n_seq)))
    S, Q = (C.c_char_p * len(seqs))(*seqs), (C.c_char_p * len(quals))(*quals)
    ids = np.ascontiguousarray(reads.read_ids, dtype=np.uint64)
    fs = mg.FragSettings(0.97, 0.9, 0.0, 0.0, 0, 0)
    pre = (C.c_uint32 * 3)(150, 0, 0)
    vp = C.c_void_p
    out = {}
    for name, lib in (("plain", plain), ("patched", ref)):
        lib.frag_construct_edges.restype = C.c_int
        lib.frag_construct_edges.argtypes = [C.POINTER(mg.FragSettings), vp, C.c_uint64, vp, vp, vp, C.c_uint32, C.c_uint32, C.c_char_p, C.c_char_p, vp,
                                             C.c_uint64, C.POINTER(C.c_uint64), vp, C.POINTER(vp), C.POINTER(C.c_uint64), vp]
        lib.frag_ec_free.argtypes = [vp]
        cap = cand.size
        edges = (mg.FragEdge * cap)()
        n_edges, nb = C.c_uint64(), C.c_uint64()
        incl = np.zeros(reads.n_reads, np.uint8)
        text = vp()
        counters = (C.c_uint32 * 3)()
        d = tmp_path / name
        d.mkdir()
        rc = lib.frag_construct_edges(C.byref(fs), pre, 10 ** 8, S, Q, ids.ctypes.data, 0, reads.n_reads, path.encode(), str(d).encode(), edges, cap,
                                      C.byref(n_edges), incl.ctypes.data, C.byref(text), C.byref(nb), counters)
        assert rc == 0
        out[name] = (bytes(C.string_at(C.addressof(edges), int(n_edges.value) * C.sizeof(mg.FragEdge))), incl.tobytes(),
                     C.string_at(text, nb.value), list(counters))
        lib.frag_ec_free(text)
    assert len(out["plain"][0]) > 1000 * 80
    assert out["plain"] == out["patched"]


def _stage(lib, name, mg, fs, pre, S, Q, ids, n_single, n_paired, fastq, ov_path, out_dir, cap, n_vertices, max_overlaps=10 ** 8):
    vp = C.c_void_p
    fn = getattr(lib, name)
    fn.restype = C.c_int
    fn.argtypes = [C.POINTER(mg.FragSettings), vp, C.c_uint64, vp, vp, vp, C.c_uint32, C.c_uint32, vp, C.c_char_p, C.c_char_p, vp, C.c_uint64,
                   C.POINTER(C.c_uint64), vp, vp, vp, C.POINTER(vp), C.POINTER(C.c_uint64), vp]
    edges = (mg.FragEdge * cap)()
    n_edges, nb = C.c_uint64(), C.c_uint64()
    in_off = np.zeros(n_vertices + 1, np.uint64)
    in_nodes = np.zeros(cap, np.uint64)
    incl = np.zeros(n_vertices, np.uint8)
    text = vp()
    counters = (C.c_uint32 * 3)()
    F = (C.c_char_p * 3)(*[f.encode() for f in fastq])
    rc = fn(C.byref(fs), pre, max_overlaps, S, Q, ids.ctypes.data, n_single, n_paired, F, ov_path.encode(), out_dir.encode(), edges, cap, C.byref(n_edges),
            in_off.ctypes.data, in_nodes.ctypes.data, incl.ctypes.data, C.byref(text), C.byref(nb), counters)
    assert rc == 0, f"{name} returned {rc}"
    n = int(n_edges.value)
    out = (bytes(C.string_at(C.addressof(edges), n * C.sizeof(mg.FragEdge))), in_off.tobytes(), in_nodes[:n].tobytes(), incl.tobytes(),
           C.string_at(text, nb.value), list(counters))
    lib.frag_ec_free.argtypes = [vp]
    lib.frag_ec_free(text)
    return n, out


@pytest.mark.parametrize("what", ["pairs", "savage_example"])
def test_level_b_adapter_fills_the_references_graph(patched, tmp_path, what):
    """INTEGRATION.md level B, compiled (integration/reference_patch/ViralQuasispecies.cpp.level_b.inc inside the patched probe): the whole
    stage inside libhcedge.so, its edges copied into the REFERENCE'S OWN OverlapGraph with the reference's Edge and addEdge — against the
    reference's own construct_edges + sortEdges on the same files: adj_out in list order (scores and mismatch rates as bits), adj_in,
    inclusions, nonedge_overlaps.txt, the three counters main() logs."""
    import gzip

    from haploconduct_amd import host, synth
    import haploconduct_amd as hc

    mg, ref = patched
    d = str(tmp_path) + "/"
    if what == "pairs":
        reads, meta = synth.make_paired_dataset(3000, 4000, flip_frac=0.25, seed=41)
        cand = synth.paired_candidates(meta, n_candidates=60000, seed=42)
        host.write_overlaps(d + "overlaps.txt", cand, reads)
        reads.write_fastq(None, d + "p1.fastq", d + "p2.fastq")
        fastq = ["None", d + "p1.fastq", d + "p2.fastq"]
        fs, pre = mg.FragSettings(0.97, 0.9, 0.0, 0.0, 0, 0), (C.c_uint32 * 3)(150, 0, 0)
    else:  # the reference's example reads, SAVAGE stage b/c settings (ignore_inclusions, merge_contigs)
        fastq = []
        for name in ("savage_singles", "savage_paired1", "savage_paired2"):
            fastq.append(d + name + ".fastq")
            with gzip.open(os.path.join(ROOT, "tests", "golden", name + ".fastq.gz"), "rb") as f, open(fastq[-1], "wb") as o:
                o.write(f.read())
        f = host.Fastq(singles=fastq[0], paired1=fastq[1], paired2=fastq[2])
        reads = f.readset()
        with hc.EdgeScorer(hc.Settings()) as sc:
            sc.set_reads(reads)
            sfo = sc.find_overlaps(0.02, 100)
        host.sfo_records_to_overlaps(sfo, d + "overlaps.txt", f.n_single, f.n_paired)
        fs, pre = mg.FragSettings(0.995, 0.9, 0.01, 0.0, 0, 1), (C.c_uint32 * 3)(100, 0, 0)
    seqs, quals = zip(*(reads.seq(q) for q in range(reads.n_seq)))
    S, Q = (C.c_char_p * len(seqs))(*seqs), (C.c_char_p * len(quals))(*quals)
    ids = np.ascontiguousarray(reads.read_ids, dtype=np.uint64)
    n_single = sum(1 for r in range(reads.n_reads) if not reads.is_paired(r))
    n_lines = sum(1 for _ in open(d + "overlaps.txt"))
    plain_path = os.path.join(ROOT, "oracle", "_ref", "libhcref_edgecalc.so")
    if not os.path.exists(plain_path):
        pytest.skip("no unpatched probe")
    plain = C.CDLL(plain_path)  # the reference's stage from the UNPATCHED probe: nothing of this build inside it
    got = {}
    for name, lib in (("frag_stage_sorted", plain), ("frag_stage_level_b", ref)):
        o = d + name
        os.mkdir(o)
        got[name] = _stage(lib, name, mg, fs, pre, S, Q, ids, n_single, reads.n_reads - n_single, fastq, d + "overlaps.txt", o, n_lines, reads.n_reads)
    n_ref, a = got["frag_stage_sorted"]
    n_b, b = got["frag_stage_level_b"]
    assert n_ref == n_b > 1000
    for k, part in enumerate(("adj_out", "in_off", "in_nodes", "inclusions", "nonedge_overlaps.txt", "counters")):
        assert a[k] == b[k], f"{part} differs between the reference's stage and the level-B adapter"
