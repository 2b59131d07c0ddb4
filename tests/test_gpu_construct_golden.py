"""The HIP stage against the reference's OWN construct_edges (src/EdgeCalculator.cpp:561-666 through the fragment probe,
tests/golden/make_golden_construct.py) on the regenerated 2.5-million-line overlaps file: every --max_ov case of the golden —
before, one line before, at and one line after the line that triggers the flush of the first 1 000 000 accepted overlaps,
ten lines before the end, unlimited — must leave the reference's adjacency lists, inclusions, nonedge_overlaps.txt and
counters.  The device's parser numbers the lines of a block through a chain of counters (hc_linechain): --max_ov cuts in the
middle of a block here."""
import os

import pytest

from haploconduct_amd import host
from tests.test_construct_golden import check_case, construct_case, settings_of  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def test_stage_reproduces_the_references_construct_edges(construct_case, tmp_path):
    m, golden, reads, ov, d = construct_case
    reads.write_fastq(None, d + "/p1.fastq", d + "/p2.fastq")
    for case in golden["cases"]:
        out = tmp_path / ("out_%d" % case["max_ov"])
        out.mkdir()
        st = settings_of(golden, case["max_ov"], n_threads=min(32, os.cpu_count() or 1))
        with host.EdgeCalculatorStage(st, paired1=d + "/p1.fastq", paired2=d + "/p2.fastq", overlaps=ov, output_dir=str(out) + "/") as ec:
            ec.construct_edges()
            edges, inc, cnt = ec.edges(), ec.inclusions(), ec.counters()
        assert cnt["self_overlap_count"] == case["self_overlap_count"]
        check_case(m, case, edges, inc, (out / "nonedge_overlaps.txt").read_bytes(), cnt)
