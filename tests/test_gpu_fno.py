"""Find-next-overlaps with its second half on the device (csrc/hc_fno_kernels.hip): computeOverlapData
per combination, four radix sorts over the lines' 256-bit text-order keys, unique, text.  Must write what the host form
writes — which tests/test_fno.py pins to the reference's own findNextOverlaps() (fragment-probe goldens) — byte for byte,
report the same counters, and leave the reference's stops to the host form's diagnosis."""
import os

import numpy as np
import pytest

from haploconduct_amd import HcError
from haploconduct_amd import fno as F
from tests import _fno as T
from tests import test_fno as host_tests

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def olib():
    return T.load_oracle()


@pytest.fixture(params=["walk on the device", "walk on the host threads"])
def on_device(request):
    """Both device routes of FNO=1: everything behind the edge list on the device (level 2), or the walk and the look-ups with
    the host threads and the second half on the device (level 1, HC_FNO_WALK=host).  FNO=3 has the one device route."""
    os.environ["HC_FNO"] = "device"
    if request.param.endswith("host threads"):
        os.environ["HC_FNO_WALK"] = "host"
    yield 1 if request.param.endswith("host threads") else 2
    os.environ.pop("HC_FNO", None)
    os.environ.pop("HC_FNO_WALK", None)


def test_goldens_of_the_reference_through_the_device(olib, on_device):
    # the same golden files, the same assertions, the other route
    host_tests.test_golden_update_overlap_oracle_and_product(olib)
    assert F.last_device_level == on_device
    host_tests.test_golden_whole_find_next_overlaps_runs(olib)
    assert F.last_device_level == on_device
    host_tests.test_golden_whole_runs_with_stored_nonedges(olib)
    assert F.last_device_level == on_device
    host_tests.test_golden_whole_runs_under_add_duplicates(olib)  # --add_duplicates: the opposite overlaps are the device's (level 2) too
    assert F.last_device_level == on_device
    host_tests.test_fno1_sections_contribute(olib)
    host_tests.test_fno1_first_edge_wins_per_superread_pair(olib)
    host_tests.test_fno1_nonedge_behind_existing_edge_is_skipped(olib)


@pytest.mark.parametrize("seed", range(int(os.environ.get("HC_FUZZ_FNO_SEEDS", "24"))))  # soak: HC_FUZZ_FNO_SEEDS=400
def test_device_matches_oracle_on_random_scenarios(olib, on_device, seed):
    host_tests.test_fno1_product_matches_oracle(olib, seed)
    assert F.last_device_level == on_device


@pytest.mark.parametrize("seed", range(12))
def test_device_matches_oracle_under_add_duplicates(olib, on_device, seed):
    host_tests.test_fno1_add_duplicates_product_matches_oracle(olib, seed)
    assert F.last_device_level == on_device


def test_add_duplicates_contract_through_the_device(olib, on_device):
    host_tests.test_add_duplicates_contract(olib)


def test_large_iteration_under_add_duplicates_device_equals_host():
    """2 x 10^5 reads on both strands, 8 x 10^5 edges and 4 x 10^5 stored non-edges with their opposites: default routing picks the
    device for the walk, HC_FNO=host the host threads; one file."""
    inp = _big(400000, 90000, 800000, 5, F.RESOLVE_ORIENTATIONS | F.ADD_DUPLICATES, dup=True)
    assert len(inp.nonedges) > 390000
    os.environ["HC_FNO"] = "host"
    try:
        want, wc = F.find_next_overlaps(inp)
        assert not F.last_on_device
    finally:
        os.environ.pop("HC_FNO", None)
    got, gc = F.find_next_overlaps(inp)
    assert F.last_device_level == 2
    assert gc == wc and got == want
    inp.flags = F.RESOLVE_ORIENTATIONS
    other, oc = F.find_next_overlaps(inp)
    assert oc["n_lines"] < wc["n_lines"], "the opposite overlaps add lines"


@pytest.mark.parametrize("seed", range(6))
def test_device_takes_graph_edges_in_any_order(olib, on_device, seed):
    """adj_out not vertex by vertex: the device's offsets need it sorted, so the walk is the host threads' (which build adj_out by
    counting) whatever was asked for, and the second half the device's."""
    host_tests.test_fno1_graph_edges_in_any_order(olib, seed)
    assert F.last_device_level == 1


def test_stops_of_the_reference_are_reported_as_by_the_host_form(olib, on_device):
    host_tests.test_fno1_aborts_where_the_reference_does(olib)


def _big(n_nodes, n_srs, n_edges, seed, flags, dup=False):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fno_bench
    inp = fno_bench.big_fno1(n_nodes, n_srs, n_edges, seed=seed, dup=dup)
    inp.flags = flags
    return inp


@pytest.mark.parametrize("seed,flags", [(1, F.RESOLVE_ORIENTATIONS), (2, F.RESOLVE_ORIENTATIONS | F.NO_INCLUSIONS), (3, 0)])
def test_large_iteration_device_equals_host(seed, flags):
    """3·10^5 vertices, 7.5·10^4 super-reads, 1.2·10^6 edges and stored non-edges: millions of combinations, repeated lines
    among them; default routing must pick the device, HC_FNO=host the host threads, and both write the same file."""
    inp = _big(300000, 75000, 1200000, seed, flags)
    os.environ["HC_FNO"] = "host"
    try:
        want, wc = F.find_next_overlaps(inp)
        assert not F.last_on_device
    finally:
        os.environ.pop("HC_FNO", None)
    got, gc = F.find_next_overlaps(inp)
    assert F.last_device_level == 2, "a batch of this size goes to the device by default, walk and all"
    assert gc == wc
    assert got == want
    os.environ["HC_FNO_WALK"] = "host"
    try:
        got1, gc1 = F.find_next_overlaps(inp)
        assert F.last_device_level == 1
    finally:
        os.environ.pop("HC_FNO_WALK", None)
    assert gc1 == wc and got1 == want
    assert wc["n_lines"] > 10 ** 6
    lines = got.split(b"\n")[:-1]
    assert len(lines) == gc["n_lines"] and all(a < b for a, b in zip(lines[:200000], lines[1:200001]))


def test_numbers_beyond_the_keys_fall_to_the_host_form(on_device):
    """ids of eleven digits and a four-digit percentage of a copied edge do not fit the device's keys: the call still
    succeeds, on the host threads, with the lines the host form writes."""
    nodes = np.zeros(3, F.FNO_READ_DTYPE)
    nodes["id"] = [20000000000, 5, 7]
    nodes["len1"] = 100
    edges = np.zeros(2, F.FNO_EDGE_DTYPE)
    edges["v1"], edges["v2"], edges["score"], edges["pos1"], edges["len1"], edges["perc"], edges["ord"] = [0, 1], [1, 2], 1.0, 10, 90, [90, 1234], ord("-")
    inp = F.Fno1Input(nodes, np.zeros(0, F.FNO_READ_DTYPE), [], [], edges, new_read_count=30000000000)
    got, cnt = F.find_next_overlaps(inp)
    assert not F.last_on_device
    assert got == b"20000000000\t5\t10\t0\t-\t+\t+\t90\t0\t90\t0\ts\ts\n5\t7\t10\t0\t-\t+\t+\t1234\t0\t90\t0\ts\ts\n" and cnt["copied"] == 2


# ---- FNO=3 on the device (deduceOverlap per candidate pair of the host's walk, the lines' places by a scan, their text) ----
def test_fno3_goldens_of_the_reference_through_the_device(olib, on_device):
    host_tests.test_golden_whole_find_next_overlaps3_runs(olib)
    assert F.last_on_device
    for flags in (0, F.NO_INCLUSIONS):  # the 800 deduceOverlap calls of the reference, one device launch each
        host_tests.test_golden_deduce_overlap(olib, flags)
        assert F.last_on_device


@pytest.mark.parametrize("seed", range(16))
def test_fno3_device_matches_oracle_on_random_scenarios(olib, on_device, seed):
    host_tests.test_fno3_product_matches_oracle(olib, seed)
    assert F.last_on_device


def test_fno3_stops_of_the_reference_are_reported_as_by_the_host_form(olib, on_device):
    host_tests.test_fno3_aborts_where_the_reference_does(olib)


@pytest.mark.parametrize("flags", [0, F.NO_INCLUSIONS])
def test_fno3_large_iteration_device_equals_host(flags):
    """2.5·10^5 super-reads over 10^6 originals: hundreds of thousands of candidate pairs; the default routing picks the device
    for a batch of this size, HC_FNO=host the host threads, and both write the same file in the same (walk) order."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fno_bench
    inp = fno_bench.big_fno3(250000, 1000000)
    inp.flags = flags
    os.environ["HC_FNO"] = "host"
    try:
        want, wc = F.find_next_overlaps3(inp)
        assert not F.last_on_device
    finally:
        os.environ.pop("HC_FNO", None)
    got, gc = F.find_next_overlaps3(inp)
    assert wc["candidates"] >= 200000, "the test means to exceed the device threshold"
    assert F.last_on_device
    assert gc == wc and got == want and wc["n_lines"] > 100000


@pytest.mark.parametrize("dup", [False, True], ids=["plain", "add_duplicates"])
def test_large_iteration_against_the_references_own_find_next_overlaps(dup, tmp_path):
    """3 * 10^5 vertices, 7.5 * 10^4 super-reads, 1.2 * 10^6 edges and 6 * 10^5 stored non-edges through the REFERENCE'S OWN
    findNextOverlaps() (fragment probe: updateOverlap, computeOverlapData, reconsiderNonedgeOverlaps reading nonedge_overlaps.txt, its
    checkEdge, the std::set<std::string>; single thread as SRBuilder forces it) and through hc_fno1_run on the device: the same
    overlaps.txt, byte for byte — also under --add_duplicates (vertices by orientation, every kept line followed by its opposite)."""
    import ctypes as C

    import numpy as np

    ref_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libhcref_fno.so")
    if not os.path.exists(ref_path):
        pytest.skip("oracle/_ref/libhcref_fno.so is built only where /root/reference exists")
    inp = _big(300000, 75000, 1200000, 11, F.RESOLVE_ORIENTATIONS | (F.ADD_DUPLICATES if dup else 0), dup=dup)
    ne, nodes = inp.nonedges, inp.nodes
    half = len(nodes) // 2 if dup else len(nodes)
    t = np.where(nodes["paired"] != 0, "p", "s")
    with open(tmp_path / "nonedge_overlaps.txt", "w") as f:  # reads named by their number (the probe's convention)
        for e in ne:
            v1, v2 = int(e["v1"]), int(e["v2"])
            f.write(f"{v1 % half}\t{v2 % half}\t{e['pos1']}\t{e['pos2']}\t{chr(e['ord'])}\t{'+' if e['ori1'] else '-'}\t{'+' if e['ori2'] else '-'}\t"
                    f"{e['perc']}\t0\t{e['len1']}\t{e['len2']}\t{t[v1]}\t{t[v2]}\n")
    got, gc = F.find_next_overlaps(inp)
    assert F.last_device_level == 2
    ref = C.CDLL(ref_path)
    vp = C.c_void_p
    ref.frag_fno1_run.argtypes = [C.POINTER(F.hc_fno1_input), C.c_char_p, C.POINTER(vp), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    ref.frag_fno_free.argtypes = [vp]
    s_, text, nb, nl = inp.struct(), vp(), C.c_uint64(), C.c_uint64()
    ref.frag_fno1_run(C.byref(s_), str(tmp_path).encode(), C.byref(text), C.byref(nb), C.byref(nl))
    want = C.string_at(text, nb.value)
    ref.frag_fno_free(text)
    assert nl.value == gc["n_lines"] > 10 ** 6
    assert got == want, "overlaps.txt differs from the reference's own"
