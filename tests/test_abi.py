"""C-ABI checks that need no GPU: the library loads, exports every function that
include/*.h declares, record layouts match, and every device entry point fails loudly
(no CPU fallback) when no HIP device is present."""
import ctypes as C
import glob
import os
import re

import numpy as np
import pytest

import haploconduct_amd as hc
from haploconduct_amd import _native as N
from haploconduct_amd.records import OVERLAP_DTYPE, RESULT_DTYPE

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    names = []
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        src = open(h).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names += re.findall(r"\b(hc_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    fns = declared_functions()
    assert len(fns) >= 14
    for name in fns:
        assert hasattr(N.lib, name), f"{name} is declared in include/*.h but not exported by libhcedge.so"


def test_every_export_cites_the_reference_in_the_header():
    src = open(os.path.join(ROOT, "include", "hcedge.h")).read()
    assert src.count("EdgeCalculator.cpp:") >= 10 and "Read.h:" in src and "Overlap.h:" in src


def test_record_layouts(tmp_path):
    """The numpy / ctypes views against what a C compiler makes of include/*.h."""
    import subprocess

    from haploconduct_amd.host import EDGE_DTYPE
    from haploconduct_amd.records import ADMIT_DTYPE, CAND_DTYPE, LINE_DTYPE, ROW_DTYPE, SFO_DTYPE, TEXT_REJECT_DTYPE, TEXT_ROW_DTYPE

    assert OVERLAP_DTYPE.itemsize == 32 and RESULT_DTYPE.itemsize == 24
    assert OVERLAP_DTYPE.fields["ord"][1] == 18 and OVERLAP_DTYPE.fields["perc"][1] == 28
    assert RESULT_DTYPE.fields["n_cls"][1] == 20
    views = {"hc_overlap_rec": OVERLAP_DTYPE, "hc_cand_rec": CAND_DTYPE, "hc_result_rec": RESULT_DTYPE, "hc_gather_row": ROW_DTYPE,
             "hc_admit_rec": ADMIT_DTYPE, "hc_edge_rec": EDGE_DTYPE, "hc_sfo_rec": SFO_DTYPE, "hc_line_rec": LINE_DTYPE,
             "hc_text_row": TEXT_ROW_DTYPE, "hc_text_reject": TEXT_REJECT_DTYPE}
    probes = []
    for name, dt in views.items():
        tag = "struct hc_gather_row" if name == "hc_gather_row" else name
        probes.append(f'printf("{name} %zu\\n", sizeof({tag}));')
        for f in dt.names:
            if not f.startswith("_"):
                probes.append(f'printf("{name}.{f} %zu\\n", offsetof({tag}, {f}));')
    structs = (("hc_settings", N.hc_settings), ("hc_graph_counts", N.hc_graph_counts), ("hc_text_result", N.hc_text_result))
    for name, st in structs:
        probes.append(f'printf("{name} %zu\\n", sizeof({name}));')
        for f, _ in st._fields_:
            probes.append(f'printf("{name}.{f} %zu\\n", offsetof({name}, {f}));')
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "hcedge.h"\n#include "hcedge_host.h"\nint main(void) {\n' + "\n".join(probes) + "\nreturn 0; }\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)], check=True)  # the headers are plain C
    got = dict(ln.split() for ln in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for name, dt in views.items():
        assert int(got[name]) == dt.itemsize, name
        for f in dt.names:
            if not f.startswith("_"):
                assert int(got[f"{name}.{f}"]) == dt.fields[f][1], (name, f)
    for name, st in structs:
        assert int(got[name]) == C.sizeof(st), name
        for f, _ in st._fields_:
            assert int(got[f"{name}.{f}"]) == getattr(st, f).offset, (name, f)


def test_pack_cands_is_the_compact_form_of_a_record():
    """hc_pack_cands (host, pure): ids as they are, positions saturating at 2^28-1, orientation and ord bits."""
    rec = np.zeros(6, OVERLAP_DTYPE)
    rec["read1"], rec["read2"] = [1, 2, 3, 4, 5, 6], [9, 8, 7, 6, 5, 4]
    rec["pos1"] = [0, 7, (1 << 28) - 1, 1 << 28, 0xFFFFFFFF, 12]
    rec["pos2"] = [5, 0, 1 << 30, 3, 2, 1]
    rec["ori1"], rec["ori2"] = [1, 0, 1, 0, 1, 1], [1, 1, 0, 0, 1, 0]
    rec["ord"] = [ord(c) for c in "-12-1x"]
    rec["len1"], rec["len2"], rec["perc"] = 77, 88, 99  # not part of the compact form
    cd = hc.EdgeScorer.pack_cands(rec)
    assert np.array_equal(cd["read1"], rec["read1"]) and np.array_equal(cd["read2"], rec["read2"])
    sat = (1 << 28) - 1
    assert np.array_equal(cd["pos1_bits"] & sat, np.minimum(rec["pos1"], sat)) and np.array_equal(cd["pos2_bits"], np.minimum(rec["pos2"], sat))
    assert np.array_equal((cd["pos1_bits"] >> 28) & 1, rec["ori1"]) and np.array_equal((cd["pos1_bits"] >> 29) & 1, rec["ori2"])
    assert list(cd["pos1_bits"] >> 30) == [0, 1, 2, 0, 1, 3]


def test_strerror_and_version():
    assert "gfx950" in hc.version()
    assert N.lib.hc_strerror(0) == b"ok"
    assert b"no CPU fallback" in N.lib.hc_strerror(-4)


@pytest.mark.skipif(hc.device_count() > 0, reason="this box has a GPU")
def test_no_device_means_loud_failure_not_a_cpu_fallback():
    with pytest.raises(hc.HcError) as e:
        hc.EdgeScorer(hc.Settings())
    assert e.value.status == -4


def test_null_arguments_are_rejected():
    assert N.lib.hc_create(None, None) == -1
    assert N.lib.hc_set_reads(None, None, None, None, None, 0) == -1
    assert N.lib.hc_score_batch(None, None, 0, None) == -1
    assert N.lib.hc_destroy(None) == 0


def test_finalize_is_the_reference_tail_arithmetic(oracle):
    """hc_finalize (host libm exp, EdgeCalculator.cpp:137-138,254-261,404-413) against the oracle's
    compute_overlap on the same (x1, x2, mm, n): pure host function, no GPU needed."""
    from haploconduct_amd import synth

    reads, meta = synth.make_paired_dataset(400, 1200, flip_frac=0.2, seed=5)
    cand = synth.paired_candidates(meta, n_candidates=3000, seed=6)
    sreads, smeta = synth.make_single_dataset(300, 1500, len_lo=150, len_hi=300, seed=7)
    scand = synth.single_candidates(smeta, min_overlap=60, n_candidates=2000)
    for rd, cd in ((reads, cand), (sreads, scand)):
        for st in (hc.Settings(edge_threshold=0.97), hc.Settings(edge_threshold=0.5, ov_threshold=0.2, merge_contigs=0.02),
                   hc.Settings(edge_threshold=1.0), hc.Settings(edge_threshold=-1, ov_threshold=-1)):
            ref = oracle.score_batch(rd, st, cd)
            res = np.zeros(cd.size, dtype=RESULT_DTYPE)
            res["x1"], res["x2"], res["mm"] = ref["x1"], ref["x2"], ref["mm"]
            res["n_cls"] = ref["n"] | (np.uint32(4) << 28)  # class AMBIG: the host must decide everything
            cs = st.to_c()
            score = np.empty(cd.size); mrate = np.empty(cd.size); cls = np.empty(cd.size, np.uint32)
            rc = N.lib.hc_finalize_batch(C.byref(cs), res.ctypes.data, cd.size, score.ctypes.data, mrate.ctypes.data,
                                         cls.ctypes.data)
            assert rc == 0
            assert np.array_equal(score.view(np.uint64), ref["score"].view(np.uint64))
            assert np.array_equal(mrate.view(np.uint64), ref["mismatch_rate"].view(np.uint64))
            assert np.array_equal(cls, ref["cls"])
