"""More than one GPU, when the box has them: bench.py's N-rank step (candidate shards against a replicated read store, one
RCCL all-gather of the non-dropped records per step) launched the way the driver launches it — torch.distributed.run, one
process per GPU — as a child process, with every form of the per-step exchange (ring: one all-gather of the fixed-capacity payload; direct:
all-gather-v by counts + grouped per-peer send / recv; root: the same towards rank 0 only), 24-byte rows where the job allows them.  RCCL must have seen every rank, and the rows every rank ends up with must be the
non-dropped records of the whole candidate set, i.e. what ONE device computes for it.  Skipped on a one-GPU box (no curve
has been measured by the builder: the first real 2/4/8-GPU numbers are the driver's)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _n_devices():
    import torch  # device_count() does not start the HIP runtime on this image

    return torch.cuda.device_count()


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("scaling,gather", [("strong", "ring"), ("strong", "direct"), ("strong", "root"), ("weak", "ring"), ("weak", "direct"), ("weak", "root")])
def test_ranks_over_rccl_collect_what_one_device_computes(tmp_path, scaling, gather, world):
    n_dev = _n_devices()
    if n_dev < world:
        pytest.skip(f"{n_dev} GPU(s) visible: this case needs {world}")
    rows_file = str(tmp_path / "rows.npy")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1",
           "--workload", "c2", "--scaling", scaling, "--one-mode", "--gather", gather, "--no-stage", "--no-cpu-baseline", "--also", "none",
           "--dump-rows", rows_file]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == world and line["scaling"] == scaling and line["config"]["gather"] == gather and line["ranks"]["gather"] == gather
    assert all(p["gather_ms"] > 0 for p in line["ranks"]["per_rank"]) and line["summary"]["gather"] == gather
    ranks = line["ranks"]
    assert ranks["n_ranks_seen"] == world and len(ranks["per_rank"]) == world, "RCCL did not see every rank"
    assert all(p["n_ranks_seen"] == world and p["kernel_ms"] > 0 for p in ranks["per_rank"])
    rows = np.load(rows_file)
    # the same candidates on ONE device, in this process
    import bench
    import haploconduct_amd as hc
    from haploconduct_amd.records import result_cls

    per_rank = []
    for rank in range(world):
        reads, cand, cfg, st = bench.build_workload("c2", 0 if scaling == "strong" else rank)
        per_rank.append(cand)
        if scaling == "strong":
            break
    want = []
    with hc.EdgeScorer(st) as sc:
        sc.set_reads(reads)
        for rank, cand in enumerate(per_rank):
            res = sc.score_cands(sc.pack_cands(cand))
            kept = np.nonzero(result_cls(res) != 0)[0]
            base = 0 if scaling == "strong" else rank * cand.size
            want.append(np.stack([(kept + base).astype(np.int64), res["x1"][kept].view(np.int64), res["x2"][kept].view(np.int64),
                                  res["mm"][kept].astype(np.int64) | (res["n_cls"][kept].astype(np.int64) << 32)], axis=1))
    want = np.concatenate(want)
    assert rows.shape == want.shape and np.array_equal(rows, want), "the gathered rows are not the one-device result"
    expected_total = sum(c.size for c in per_rank) if scaling == "weak" else per_rank[0].size
    assert sum(p["candidates"] for p in ranks["per_rank"]) == expected_total


def _check_default_line(line, world):
    """The driver's command line (no --scaling, no --one-mode, no --gather): the headline is the STRONG split of the one candidate set
    (BASELINE configs[2]), the weak figure rides along under "weak" with its own per-rank record, both forms of the exchange ran as full
    legs ("gather_modes") and the faster one is the headline; every leg passed its in-run parity checks; "summary" closes the line."""
    assert line["n_gpus"] == world and line["scaling"] == "strong"
    assert line["config"]["candidates_per_step"] == 2000000 and f"split over {world} ranks" in line["config"]["workload"]
    assert sum(p["candidates"] for p in line["ranks"]["per_rank"]) == line["config"]["candidates_per_step"]
    assert line["ranks"]["n_ranks_seen"] == world and len(line["ranks"]["per_rank"]) == world
    wk = line["weak"]
    assert wk["scaling"] == "weak" and wk["candidates_per_step"] == world * 2000000 and wk["ranks"]["n_ranks_seen"] == world
    assert sum(p["candidates"] for p in wk["ranks"]["per_rank"]) == wk["candidates_per_step"]
    assert line["parity"]["digest_matches_untimed_launch"] and wk["parity"]["digest_matches_untimed_launch"] and line["parity"]["parity_checked_records"] > 0
    gm = line["gather_modes"]
    assert set(gm) == {"ring", "direct", "root"} and all(v["value"] > 0 and v["parity"]["digest_matches_untimed_launch"] for v in gm.values())
    assert gm["ring"]["parity"]["digest"] == gm["direct"]["parity"]["digest"] == gm["root"]["parity"]["digest"], "the forms of the exchange must leave the same results"
    assert all(v["gather_row_bytes"] == 24 for v in gm.values()), "config 2's rows fit the 24-byte form"
    best = max(gm, key=lambda m: gm[m]["value"])
    assert line["config"]["gather"] == best and line["value"] == gm[best]["value"]
    assert list(line)[-1] == "summary" and abs(line["summary"]["value"] / line["value"] - 1.0) < 1e-4 and line["summary"]["weak"]["value"] == wk["value"]
    assert line["summary_first"] == line["summary"] and list(line).index("summary_first") < list(line).index("config"), "the summary also sits behind the contract's keys"


def test_one_pass_yields_both_curves():
    """The driver's command line at N = 2 over RCCL (see _check_default_line)."""
    n_dev = _n_devices()
    if n_dev < 2:
        pytest.skip(f"{n_dev} GPU(s) visible: the N > 1 path needs two")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--workload", "c2", "--no-stage", "--no-cpu-baseline", "--also", "none"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    _check_default_line(json.loads(r.stdout.strip().splitlines()[-1]), 2)


def test_hc_edgecalc_device_mask_over_real_devices(tmp_path):
    """The process boundary on more than one GPU: hc-edgecalc --device_mask 3 deals the blocks of the overlaps file to two devices;
    graph and non-edge file equal the one-device run's."""
    n_dev = _n_devices()
    if n_dev < 2:
        pytest.skip(f"{n_dev} GPU(s) visible: --device_mask 3 needs two")
    from haploconduct_amd import host, synth

    reads, meta = synth.make_paired_dataset(4000, 6000, flip_frac=0.25, seed=21)
    cand = synth.paired_candidates(meta, n_candidates=600000, seed=22)
    d = str(tmp_path) + "/"
    host.write_overlaps(d + "overlaps.txt", cand, reads)
    reads.write_fastq(None, d + "p1.fastq", d + "p2.fastq")
    exe = os.path.join(ROOT, "haploconduct_amd", "csrc", "hc-edgecalc")
    outs = {}
    for name, mask in (("one", "1"), ("two", "3")):
        o = d + name + "/"
        os.mkdir(o)
        r = subprocess.run([exe, "--paired1", d + "p1.fastq", "--paired2", d + "p2.fastq", "--overlaps", d + "overlaps.txt", "--output", o,
                            "--edge_threshold", "0.97", "--min_overlap_len", "150", "--threads", "8", "--device_mask", mask, "--graph_only", "true",
                            "--original_readcount", str(reads.n_reads)],
                           capture_output=True, text=True, timeout=600, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs[name] = {f: open(o + f, "rb").read() for f in sorted(os.listdir(o)) if f in ("edges.tsv", "edges_sorted.tsv", "nonedge_overlaps.txt", "edgecalc_stats.txt")}
    assert len(outs["one"]) == 4 and len(outs["one"]["edges_sorted.tsv"]) > 10000 and outs["one"] == outs["two"]


@pytest.mark.parametrize("world", [2, 4, 8])
def test_n_rank_bench_path_on_one_gpu_over_gloo(tmp_path, world):
    """The N-rank path of bench.py WITHOUT N GPUs: every rank on device 0, the all-gather staged through the host over gloo
    (HC_BENCH_BACKEND=gloo HC_BENCH_ONE_DEVICE=1; the line says "test_run").  Everything but RCCL itself is the driver's path: torch.distributed.run,
    the shards, both scaling modes in one line, the per-rank records, the in-run parity of every rank, the gathered rows.  Runs on a one-GPU box."""
    if _n_devices() < 1:
        pytest.skip("no GPU")
    env = dict(os.environ, HC_BENCH_BACKEND="gloo", HC_BENCH_ONE_DEVICE="1", MASTER_ADDR="127.0.0.1")
    base = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1"]
    tail = ["--gpus", str(world), "--steps", "3", "--warmup", "1", "--workload", "c2", "--no-stage", "--no-cpu-baseline", "--also", "none"]
    # (1) the driver's command line: both curves in one line
    r = subprocess.run(base + ["--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")] + tail, cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert "test_run" in line["config"]
    _check_default_line(line, world)
    # (2) the strong split's gathered rows are the one-device result, by either form of the exchange
    rows = {}
    for gather, row_bytes in (("ring", "24"), ("direct", "24"), ("root", "24"), ("root", "32")):  # (32: the rows as the kernels write them)
        rows_file = str(tmp_path / f"rows_{gather}_{row_bytes}.npy")
        r = subprocess.run(base + ["--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")] + tail +
                           ["--scaling", "strong", "--one-mode", "--gather", gather, "--dump-rows", rows_file], cwd=ROOT,
                           env=dict(env, HC_BENCH_ROW_BYTES=row_bytes), capture_output=True, text=True, timeout=1200)
        assert r.returncode == 0, r.stderr[-3000:]
        assert json.loads(r.stdout.strip().splitlines()[-1])["config"]["gather_row_bytes"] == int(row_bytes)
        rows[(gather, row_bytes)] = np.load(rows_file)
    assert all(np.array_equal(rows[("ring", "24")], v) for v in rows.values()), "every form and row width collects the same set"
    rows = rows[("ring", "24")]
    import bench
    import haploconduct_amd as hc
    from haploconduct_amd.records import result_cls

    reads, cand, cfg, stt = bench.build_workload("c2", 0)
    with hc.EdgeScorer(stt) as sc:
        sc.set_reads(reads)
        res = sc.score_cands(sc.pack_cands(cand))
    kept = np.nonzero(result_cls(res) != 0)[0]
    want = np.stack([kept.astype(np.int64), res["x1"][kept].view(np.int64), res["x2"][kept].view(np.int64),
                     res["mm"][kept].astype(np.int64) | (res["n_cls"][kept].astype(np.int64) << 32)], axis=1)
    assert rows.shape == want.shape and np.array_equal(rows, want)


@pytest.mark.parametrize("workload,kernel", [("c5", "true, true, 2"), ("c4", "uint8_t, 6, 768")])
def test_every_launch_form_through_the_n_rank_path_over_gloo(tmp_path, workload, kernel):
    """BASELINE configs[4] ("... length-bucketed LDS tiling, 8 MI355X") and configs[3] through the N-rank step (round 6):
    the length-bucketed launch with its row sink (c5) and the wide-table 768-lane LDS-DMA form (c4) under StreamedGather — two
    ranks on device 0 over gloo, the strong split's gathered rows against the one-device result, by the ring and the root form."""
    if _n_devices() < 1:
        pytest.skip("no GPU")
    import bench
    import haploconduct_amd as hc
    from haploconduct_amd.records import result_cls

    env = dict(os.environ, HC_BENCH_BACKEND="gloo", HC_BENCH_ONE_DEVICE="1", MASTER_ADDR="127.0.0.1")
    base = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1"]
    tail = ["--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", workload, "--no-stage", "--no-cpu-baseline", "--also", "none",
            "--scaling", "strong", "--one-mode"]
    got = {}
    for gather in ("ring", "root"):
        rows_file = str(tmp_path / f"rows_{gather}.npy")
        r = subprocess.run(base + ["--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")] + tail + ["--gather", gather, "--dump-rows", rows_file],
                           cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
        assert r.returncode == 0, r.stderr[-3000:]
        line = json.loads(r.stdout.strip().splitlines()[-1])
        assert kernel in line["roofline"]["kernel"], line["roofline"]["kernel"]
        assert line["config"]["gather_row_bytes"] == 24 and line["ranks"]["n_ranks_seen"] == 2
        got[gather] = np.load(rows_file)
    assert np.array_equal(got["ring"], got["root"])
    reads, cand, cfg, stt = bench.build_workload(workload, 0)
    with hc.EdgeScorer(stt) as sc:
        sc.set_reads(reads)
        res = sc.score_cands(sc.pack_cands(cand))
    kept = np.nonzero(result_cls(res) != 0)[0]
    want = np.stack([kept.astype(np.int64), res["x1"][kept].view(np.int64), res["x2"][kept].view(np.int64),
                     res["mm"][kept].astype(np.int64) | (res["n_cls"][kept].astype(np.int64) << 32)], axis=1)
    assert kept.size > 1000 and got["ring"].shape == want.shape and np.array_equal(got["ring"], want)
