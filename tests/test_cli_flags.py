"""hc-edgecalc keeps ViralQuasispecies' flag surface and argument validation
(reference src/ViralQuasispecies.cpp:49-154): checks that need no GPU."""
import os
import subprocess

import haploconduct_amd as hc

EXE = os.path.join(os.path.dirname(hc.lib_path), "hc-edgecalc")


def run(*args):
    return subprocess.run([EXE, *args], capture_output=True, text=True)


def test_help_lists_every_reference_flag():
    r = run("--help")  # savage.py:84-90 / haploconduct.py:46-54 use `--help` as the install check
    assert r.returncode == 0
    for flag in ("fastq", "singles", "paired1", "paired2", "overlaps", "output", "IDs", "max_ov", "max_reads", "threads",
                 "min_clique_size", "min_qual", "min_overlap_perc", "min_overlap_len", "edge_threshold", "ov_threshold",
                 "allow_spaced_overlaps", "first_it", "add_duplicates", "resolve_orientations", "keep_singletons",
                 "error_correction", "cliques", "ignore_inclusions", "graph_only", "FNO", "original_readcount", "mismatch",
                 "optimize", "no_inclusion_overlaps", "merge_contigs", "remove_multi_occ", "remove_trans", "remove_branches",
                 "remove_tips", "min_read_len", "max_tip_len", "separate_tips", "base_path", "diploid", "relax_PE_edges",
                 "original_fastq", "branch_reduction", "branch_SE_c", "branch_PE_c", "careful_diploid", "verbose"):
        assert "--" + flag in r.stdout, flag


def test_validation_messages_and_exit_codes():
    r = run("--singles", "x.fastq")
    assert r.returncode == 1 and "No overlaps file provided." in r.stderr
    r = run("--singles", "x.fastq", "--overlaps", "o.txt")
    assert r.returncode == 1 and "No original readcount provided." in r.stderr
    r = run("--singles=x", "--overlaps=o", "--original_readcount=1", "--add_duplicates=true")
    assert r.returncode == 1 and "exclusive options" in r.stderr
    r = run("--singles=x", "--overlaps=o", "--original_readcount=1", "--error_correction=true")
    assert r.returncode == 1 and "Error correction requires clique enumeration" in r.stderr
    r = run("--fastq=dir", "--singles=x", "--overlaps=o", "--original_readcount=1")
    assert r.returncode == 1 and "Cannot combine --fastq" in r.stderr
    r = run("--no_such_flag=1")
    assert r.returncode == 1
    r = run("--threads=abc", "--singles=x", "--overlaps=o", "--original_readcount=1")
    assert r.returncode == 1
    r = run("-s", "missing.fastq", "--overlaps", "o.txt", "--original_readcount", "5", "-t", "2", "-v", "false")
    assert r.returncode == 1 and "Unable to open fastq file" in r.stderr  # FastqStorage.cpp:53-56


def test_workflow_argument_lists_are_accepted(tmp_path):
    """The process boundary: every argument list the reference's workflows build for
    ViralQuasispecies — scripts/pipeline_per_stage.py's five run_* functions and polyte.py's call, flag names and
    printf templates extracted into tests/golden/pipeline_argv.json — parses and validates in hc-edgecalc: the run
    then stops where the reference would too, at the FASTQ file that does not exist."""
    import json

    here = os.path.dirname(os.path.abspath(__file__))
    sites = json.load(open(os.path.join(here, "golden", "pipeline_argv.json")))["sites"]
    assert [s["function"] for s in sites] == ["run_first_it_merge", "run_first_it_noEC", "run_merging_it", "run_error_correction",
                                              "run_clique_it", "run_viralquasispecies"]
    paths = {"singles", "paired1", "paired2", "overlaps", "base_path", "original_fastq", "fastq", "output", "IDs"}
    booleans = {"first_it", "remove_branches", "remove_tips", "verbose", "diploid", "separate_tips", "ignore_inclusions",
                "error_correction", "cliques", "optimize", "branch_reduction", "careful_diploid", "relax_PE_edges"}
    for s in sites:
        assert len(s["flags"]) >= 22
        argv = []
        for f in s["flags"]:
            name, t = f["flag"], f["template"]
            if name in paths:
                v = t.replace("%s", str(tmp_path / "nowhere"))
            elif name in booleans:
                v = t.replace("%s", "false" if name == "error_correction" else "true")
            else:
                v = t.replace("%d", "3").replace("%f", "%f" % 0.97).replace("%s", "5")
            assert "%" not in v, (name, t)
            argv += [f"--{name}", v] if f["separate"] else [f"--{name}={v}"]
        r = run(*argv)
        assert r.returncode == 1, (s["function"], r.stderr)
        assert "unrecognised option" not in r.stderr and "is invalid" not in r.stderr, (s["function"], r.stderr)
        assert "Unable to open fastq file" in r.stderr, (s["function"], r.stderr)
