"""World-size-2/4/8 test of the multi-GPU path on CPU (gloo): ONE candidate set split into contiguous, order-preserving
shards (parallel.shard_range — what `bench.py --gpus N --scaling strong` does with the 1e8 candidates; an odd count, so
the shards differ in size), one all-gather-v of the admitted records, same admitted set as the single-process run.
The scoring itself is stood in for by the oracle here (no GPU in this container); on the GPU box the same functions
run over RCCL from bench.py --gpus N."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["HC_ROOT"])
import numpy as np, torch, torch.distributed as dist
import haploconduct_amd as hc
from haploconduct_amd import synth, parallel
from haploconduct_amd.records import RESULT_DTYPE
from tests import _oracle

dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + os.environ["HC_PORT"],
                        rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
reads, meta = synth.make_paired_dataset(500, 1500, flip_frac=0.2, seed=5)
reads.quals[:] = ord("I")
cand = synth.paired_candidates(meta, n_candidates=5001, seed=6)
st = hc.Settings(edge_threshold=0.97)
lo, hi = parallel.shard_range(cand.size, rank, world)
ref = _oracle.score_batch(reads, st, cand[lo:hi])          # stands in for hc_score_batch on this rank's GPU
res = np.zeros(hi - lo, dtype=RESULT_DTYPE)
res["x1"], res["x2"], res["mm"] = ref["x1"], ref["x2"], ref["mm"]
res["n_cls"] = ref["n"] | (ref["cls"].astype(np.uint32) << 28)
rows, counts = parallel.gather_admitted(res, lo)
np.save(os.path.join(os.environ["HC_OUT"], f"rows{rank}.npy"), rows.numpy())
np.save(os.path.join(os.environ["HC_OUT"], f"counts{rank}.npy"), np.array(counts))
# the form bench.py --gpus N runs: ONE all-gather per batch of a fixed-capacity payload (count in row 0), double
# buffered; here the payload is packed on the host in the layout the device kernels write (rows unordered, as the
# fused scoring kernel emits them), several batches in a row so that the buffers are reused
kept = int((res["n_cls"] >> 28 != 0).sum())
cap = torch.tensor([kept])
dist.all_reduce(cap, op=dist.ReduceOp.MAX)
lowest = torch.tensor([kept])
dist.all_reduce(lowest, op=dist.ReduceOp.MIN)  # one capacity for all ranks: the payloads must have one size
for mode in parallel.GATHER_MODES:  # "ring": one all-gather of the fixed-capacity payload; "direct": counts + per-peer send / recv of 1 + count rows; "root": the same towards rank 0 only
  for width in (4, 3):  # 32-byte rows; 24-byte rows (index | mm << 32 | n << 46 | class << 60 in one word)
    tag = f"{mode}_{width}_"
    pg = parallel.PayloadGather(int(cap.item()) + 5, depth=2, mode=mode, width=width)
    last = None
    for it in range(5):
        b = pg.next_buffers()
        b["payload"].copy_(parallel.pack_payload(res, lo, pg.cap, shuffle_seed=100 * rank + it, width=width))
        b["unordered"] = True
        last = pg.submit(b)
    prows, pcounts = pg.collect(last)
    pg.finish()
    assert pg.gather_ms() > 0.0 and pg.steps_timed == 5
    assert (prows is None) == (mode == "root" and rank != 0), "root: the rows are on rank 0 only"
    # lag mode (what StreamedGather.score_step does): the exchange of batch i is issued behind the hand-over of batch i + 1
    lg = parallel.PayloadGather(int(cap.item()) + 5, depth=2, mode=mode, lag=True, width=width)
    for it in range(5):
        b = lg.next_buffers()
        b["payload"].copy_(parallel.pack_payload(res, lo, lg.cap, shuffle_seed=7 * rank + it, width=width))
        b["unordered"] = True
        lg.flush()
        assert not lg._pending
        last = lg.submit(b)
        assert last["pending"] and len(lg._pending) == 1
    lrows, lcounts = lg.collect(last)
    lg.finish()
    assert (lrows is None and prows is None) or torch.equal(lrows, prows)
    assert lcounts == pcounts and lg.steps_timed == 5
    if prows is not None:
        np.save(os.path.join(os.environ["HC_OUT"], f"prows_{tag}{rank}.npy"), prows.numpy())
    np.save(os.path.join(os.environ["HC_OUT"], f"pcounts_{tag}{rank}.npy"), np.array(pcounts))
    small = parallel.PayloadGather(max(1, int(lowest.item()) // 3), mode=mode, width=width)
    b = small.next_buffers()
    b["payload"].copy_(parallel.pack_payload(res, lo, small.cap, width=width))
    try:
        small.collect(small.submit(b))
        overflow = False
    except OverflowError:
        overflow = True
    small.finish()
    np.save(os.path.join(os.environ["HC_OUT"], f"overflow_{tag}{rank}.npy"), np.array([overflow]))
    dist.barrier()
# a row that does not fit the 24-byte form is refused, not truncated: 20 000 overlapped positions
big = res.copy()
big["n_cls"][np.nonzero(big["n_cls"] >> 28 != 0)[0][:1]] = 20000 | (2 << 28)
nb = parallel.PayloadGather(int(cap.item()) + 5, mode="ring", width=3)
b = nb.next_buffers()
pl = parallel.pack_payload(big, lo, nb.cap, width=3)
pl[0, 1] = 1  # what hc_narrow_payload_device counts
b["payload"].copy_(pl)
try:
    nb.collect(nb.submit(b))
    refused = False
except OverflowError:
    refused = True
nb.finish()
assert refused
dist.destroy_process_group()
'''


def test_shard_range_partitions_exactly():
    from haploconduct_amd.parallel import shard_range

    for n in (0, 1, 7, 64, 1000003):
        for w in (1, 2, 3, 8):
            edges = [shard_range(n, r, w) for r in range(w)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("world", [2, 4, 8])
def test_n_rank_gather_equals_single_process(oracle, world):
    import haploconduct_amd as hc
    from haploconduct_amd import parallel, synth

    with tempfile.TemporaryDirectory() as d:
        script = os.path.join(d, "worker.py")
        open(script, "w").write(WORKER)
        import socket

        with socket.socket() as sk:  # a port nobody holds right now (not derived from the pid: parallel CI runs collide)
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), HC_PORT=port, HC_ROOT=ROOT, HC_OUT=d,
                       OMP_NUM_THREADS="1")
            procs.append(subprocess.Popen([sys.executable, script], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
        for p in procs:
            out, _ = p.communicate(timeout=300)
            assert p.returncode == 0, out.decode()[-2000:]
        rows = [np.load(os.path.join(d, f"rows{r}.npy")) for r in range(world)]
        rows0 = rows[0]
        counts = np.load(os.path.join(d, "counts0.npy"))
        forms = [f"{m}_{w}_" for m in parallel.GATHER_MODES for w in (4, 3)]
        holders = {f: (range(1) if f.startswith("root") else range(world)) for f in forms}  # root: rank 0 alone holds the rows
        prows = {f: [np.load(os.path.join(d, f"prows_{f}{r}.npy")) for r in holders[f]] for f in forms}
        pcounts = {f: [np.load(os.path.join(d, f"pcounts_{f}{r}.npy")) for r in range(world)] for f in forms}
        overflow = {f: [bool(np.load(os.path.join(d, f"overflow_{f}{r}.npy"))[0]) for r in range(world)] for f in forms}
        assert not any(os.path.exists(os.path.join(d, f"prows_root_{w}_{r}.npy")) for w in (4, 3) for r in range(1, world))
    assert all(np.array_equal(rows0, r) for r in rows), "every rank must hold the same gathered set"
    reads, meta = synth.make_paired_dataset(500, 1500, flip_frac=0.2, seed=5)
    reads.quals[:] = ord("I")
    cand = synth.paired_candidates(meta, n_candidates=5001, seed=6)
    ref = oracle.score_batch(reads, hc.Settings(edge_threshold=0.97), cand)
    want = np.nonzero((ref["cls"] == 2) | (ref["cls"] == 3))[0]
    assert want.size > 50
    assert np.array_equal(rows0[:, 0], want), "admitted set / order differs from the single-process run"
    assert np.array_equal(rows0[:, 1].view(np.float64).view(np.uint64), ref["x1"][want].view(np.uint64))
    assert counts.sum() == want.size and len(counts) == world
    # the payload forms (ring: one all-gather of the fixed-capacity payload; direct: counts, then 1 + count rows per pair of ranks): every
    # non-dropped record (edges and non-edges), on every rank, in global order — and the two forms bit-identical
    kept = np.nonzero(ref["cls"] != 0)[0]
    for m in forms:
        assert all(np.array_equal(prows[m][0], r) for r in prows[m]) and all(np.array_equal(pcounts[m][0], c) for c in pcounts[m])
        assert pcounts[m][0].sum() == kept.size and len(pcounts[m][0]) == world
        assert np.array_equal(prows[m][0][:, 0], kept)
        assert np.array_equal(prows[m][0][:, 1].view(np.float64).view(np.uint64), ref["x1"][kept].view(np.uint64))
        assert np.array_equal(prows[m][0][:, 3] >> 60, ref["cls"][kept].astype(np.int64))
        assert overflow[m] == [True] * world
        assert np.array_equal(prows[forms[0]][0], prows[m][0]), f"{m}: another set than {forms[0]}"


def test_narrow_rows_round_trip():
    import torch

    from haploconduct_amd import parallel

    rng = np.random.default_rng(3)
    k = 1000
    rows = np.zeros((k, 4), np.int64)
    rows[:, 0] = rng.integers(0, 1 << 32, k)
    rows[:, 1:3] = rng.integers(-(1 << 62), 1 << 62, (k, 2))
    n = rng.integers(1, 1 << 14, k)
    mm = rng.integers(0, n + 1)
    cls = rng.integers(0, 16, k)
    rows[:, 3] = (mm.astype(np.uint64) | ((n.astype(np.uint64) | (cls.astype(np.uint64) << np.uint64(28))) << np.uint64(32))).view(np.int64)
    rows[0, 0], rows[1, 0] = 0, (1 << 32) - 1
    rows[2, 3] = np.array([((1 << 14) - 1) | ((((1 << 14) - 1) | (15 << 28)) << 32)], dtype=np.uint64).view(np.int64)[0]
    t = torch.from_numpy(rows)
    assert torch.equal(parallel.widen_rows(parallel.narrow_rows(t)), t)
    assert parallel.rows_fit_narrow(10**8, 150) and parallel.rows_fit_narrow((1 << 32) - 1, 16367)
    assert not parallel.rows_fit_narrow(1 << 32, 150) and not parallel.rows_fit_narrow(10**8, 16368)
