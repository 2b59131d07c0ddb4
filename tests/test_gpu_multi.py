"""More than one GPU, when the box has them: bench.py's N-rank step (candidate shards against a replicated read store, one
RCCL all-gather of the non-dropped records per step) launched the way the driver launches it — torch.distributed.run, one
process per GPU — as a child process.  RCCL must have seen every rank, and the rows every rank ends up with must be the
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


@pytest.mark.parametrize("scaling", ["strong", "weak"])
def test_two_ranks_over_rccl_collect_what_one_device_computes(tmp_path, scaling):
    n_dev = _n_devices()
    if n_dev < 2:
        pytest.skip(f"{n_dev} GPU(s) visible: the N > 1 path needs two")
    world = 2
    rows_file = str(tmp_path / "rows.npy")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1",
           "--workload", "c2", "--scaling", scaling, "--no-stage", "--no-cpu-baseline", "--also", "none", "--dump-rows", rows_file]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == world and line["scaling"] == scaling
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
