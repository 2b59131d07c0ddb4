"""Multi-GPU sharding of the edge-calculation path.

Candidates are independent given the read store, so the path shards with NO data-path
collective: rank r scores the contiguous slice shard_range(n, r, world) of the candidate
array against a replicated read store.  The only exchange is the collection of the admitted
edge records (class EDGE / EDGE_MC / AMBIG): one all-gather of the per-rank counts, then one
all-gather of the payload padded to the largest count (RCCL has no native all-gather-v).
Works on any torch.distributed backend: "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU
tests.  Records travel as int64 rows [global_index, x1_bits, x2_bits, mm | n_cls << 32]."""
import numpy as np
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous, balanced, order-preserving partition: rank r gets [lo, hi)."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def admitted_rows(results_i64, global_offset):
    """results_i64: [n, 3] int64 view of hc_result_rec rows on any device -> [k, 4] rows of admitted records."""
    cls = (results_i64[:, 2] >> 60) & 0xF
    idx = torch.nonzero((cls >= 2) & (cls <= 4)).squeeze(1)
    return torch.cat([(idx + global_offset).unsqueeze(1), results_i64[idx]], dim=1).contiguous()


def all_gather_v(rows, group=None):
    """rows: [k_r, C] int64 on this rank -> [sum k_r, C] on every rank, in rank order."""
    world = dist.get_world_size(group)
    cnt = torch.tensor([rows.shape[0]], dtype=torch.int64, device=rows.device)
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt, group=group)
    counts = [int(c.item()) for c in cnts]
    kmax = max(counts) if counts else 0
    pad = torch.zeros((kmax, rows.shape[1]), dtype=rows.dtype, device=rows.device)
    pad[: rows.shape[0]] = rows
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0), counts


def gather_admitted(results_np_or_tensor, global_offset, group=None, device=None):
    """Convenience: hc_result_rec array (numpy structured or uint8/int64 tensor) of this rank's shard ->
    all ranks' admitted rows, ordered by global candidate index."""
    if isinstance(results_np_or_tensor, np.ndarray):
        t = torch.from_numpy(results_np_or_tensor.view(np.int64).reshape(-1, 3).copy())
        if device is not None:
            t = t.to(device)
    else:
        t = results_np_or_tensor.view(torch.int64).view(-1, 3)
    rows, counts = all_gather_v(admitted_rows(t, global_offset), group)
    return rows, counts


class StreamedGather:
    """The same collection without a host round trip per batch — what `bench.py --gpus N` runs.

    Per batch, on torch's current HIP stream: device stream compaction of the non-dropped records
    (`hc_compact_device`), one kernel that tags them with their global candidate index (`hc_pack_rows_device`,
    32-byte rows), then two asynchronous all-gathers over RCCL: the counts and the rows padded to a fixed
    capacity.  Nothing synchronises with the host; the buffers are double-buffered, so the all-gather of batch i
    overlaps the scoring kernel of batch i+1.  `collect()` is the host side: waits, checks the capacity, trims.
    """

    def __init__(self, scorer, n_local, base_index, cap_rows, group=None, depth=2):
        self.sc, self.n, self.base, self.cap, self.group, self.depth = scorer, int(n_local), int(base_index), int(cap_rows), group, depth
        self.world = dist.get_world_size(group)
        dev = torch.device("cuda", torch.cuda.current_device())
        # zero-initialised: entries beyond the count of a batch are stale but always valid indices
        self.idx = torch.zeros(max(self.n, 1), dtype=torch.int32, device=dev)
        self.bufs = [{"count": torch.zeros(1, dtype=torch.int64, device=dev),
                      "rows": torch.zeros((self.cap, 4), dtype=torch.int64, device=dev),
                      "counts_all": torch.zeros(self.world, dtype=torch.int64, device=dev),
                      "rows_all": torch.zeros((self.world * self.cap, 4), dtype=torch.int64, device=dev),
                      "work": ()} for _ in range(depth)]
        self.i = 0

    def step(self, d_results):
        """d_results: uint8/int64 CUDA tensor holding this rank's n hc_result_rec.  Enqueues; returns the buffer set."""
        b = self.bufs[self.i % self.depth]
        self.i += 1
        for w in b["work"]:
            w.wait()  # the batch that used these buffers `depth` batches ago has been gathered
        stream = torch.cuda.current_stream().cuda_stream
        self.sc.compact_device(d_results.data_ptr(), self.n, self.idx.data_ptr(), b["count"].data_ptr(), stream)
        self.sc.pack_rows_device(d_results.data_ptr(), self.idx.data_ptr(), b["count"].data_ptr(), self.cap, self.base,
                                 b["rows"].data_ptr(), stream)
        b["work"] = (dist.all_gather_into_tensor(b["counts_all"], b["count"], group=self.group, async_op=True),
                     dist.all_gather_into_tensor(b["rows_all"], b["rows"], group=self.group, async_op=True))
        return b

    def collect(self, b):
        """Host side of one batch: (rows [sum k_r, 4] int64 ordered by global index, counts per rank)."""
        for w in b["work"]:
            w.wait()
        torch.cuda.current_stream().synchronize()
        counts = [int(c) for c in b["counts_all"].tolist()]
        if max(counts) > self.cap:
            raise OverflowError(f"a rank produced {max(counts)} records, capacity is {self.cap}: rerun the batch with a larger cap_rows")
        rows = torch.cat([b["rows_all"][r * self.cap: r * self.cap + counts[r]] for r in range(self.world)], dim=0)
        return rows, counts

    def finish(self):
        for b in self.bufs:
            for w in b["work"]:
                w.wait()
            b["work"] = ()
        torch.cuda.current_stream().synchronize()
